"""Row sharding of one frame over the GPUs of a node and the gather of the accumulation buffer.

The reference parallelises by handing image rows to worker threads (src/main.cpp:1001-1020);
each pixel is written by exactly one worker.  Here a rank plays the worker: rows are dealt in
blocks of PT_ROW_BLOCK, block b to rank b % world (PtFrame.shard_index / shard_count), so each
pixel's samples are still summed by one owner in ascending order and the assembled buffer is
byte-identical to a single-GPU render.  The only exchange is the final gather of each rank's
rows to rank 0 (RCCL over xGMI on GPUs, gloo on CPU): 32 B per pixel, once per frame, or 4 B per
pixel when only the tone-mapped image is wanted (`gather_image8`).
"""
import torch
import torch.distributed as dist

from .ctypes_abi import PT_ROW_BLOCK


def shard_rows(height, rank, world, row_begin=0):
    """Row indices (ascending) that `rank` owns out of rows [row_begin, height)."""
    rows = torch.arange(row_begin, height)
    block = (rows - row_begin) // PT_ROW_BLOCK
    return rows[block % world == rank]


def pack_rows(accum, rows, length, out=None):
    """The send buffer of one rank: its rows of `accum` (row indices `rows`, a tensor on accum's device),
    packed in ascending order into `length` rows (every rank sends the same size; the tail is zero).
    `out`, if given, is a buffer from an earlier call for the same arguments: nothing is allocated then."""
    if out is None:
        out = torch.zeros((length,) + tuple(accum.shape[1:]), dtype=accum.dtype, device=accum.device)
    torch.index_select(accum, 0, rows, out=out[: len(rows)])
    return out


def unpack_rows(accum, rows, received):
    """Puts the rows another rank packed with `pack_rows` back at their places (`rows`) in `accum`."""
    accum.index_copy_(0, rows, received[: len(rows)])
    return accum


class RowGather:
    """The gather of one frame size, with its index tensors and its send and receive buffers made once: a frame loop
    calls it every frame, and nothing is allocated inside the loop."""

    def __init__(self, height, width, dtype, device, rank, world, dst=0, group=None):
        self.shape, self.rank, self.world, self.dst, self.group = (height, width, 4), rank, world, dst, group
        self.rows_of = [shard_rows(height, r, world).to(device) for r in range(world)]
        longest = max(len(rows) for rows in self.rows_of)
        self.send = torch.zeros((longest, width, 4), dtype=dtype, device=device)
        self.received = [torch.empty_like(self.send) for _ in range(world)] if rank == dst else None

    def __call__(self, accum):
        assert tuple(accum.shape) == self.shape and accum.dtype == self.send.dtype and accum.device == self.send.device
        pack_rows(accum, self.rows_of[self.rank], len(self.send), out=self.send)
        dist.gather(self.send, self.received, dst=self.dst, group=self.group)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst:
                    unpack_rows(accum, self.rows_of[r], self.received[r])
        return accum


_gathers = {}


def gather_rows(accum, height, width, rank, world, dst=0, group=None):
    """accum: (height, width, 4) tensor (float64 sums, or uint8 image) holding this rank's rows
    (others are ignored).  After the call rank `dst` holds every rank's rows in place; other ranks
    are unchanged.  The buffers of a (size, type, device, rank, world) are made at the first call and kept."""
    if world == 1:
        return accum
    assert accum.shape == (height, width, 4)
    key = (height, width, accum.dtype, accum.device, rank, world, dst, id(group))
    if key not in _gathers:
        if len(_gathers) >= 16:
            _gathers.clear()
        _gathers[key] = RowGather(height, width, accum.dtype, accum.device, rank, world, dst, group)
    return _gathers[key](accum)


def gather_image8(accum, sample_count, height, width, rank, world, dst=0, group=None):
    """Tone-maps this rank's rows where they are (src/main.cpp:1024-1034; on the GPU when `accum` is
    there) and gathers the BGRA8 image to rank `dst`: an eighth of the bytes of `gather_rows`."""
    from . import api
    if accum.is_cuda:
        image = torch.empty((height, width, 4), dtype=torch.uint8, device=accum.device)
        api.tonemap_device(accum.device.index, accum.data_ptr(), height * width, sample_count, image.data_ptr(),
                           torch.cuda.current_stream(accum.device).cuda_stream)
    else:
        image = torch.from_numpy(api.tonemap_bgra8(accum.contiguous().numpy(), sample_count)).reshape(height, width, 4)
    return gather_rows(image, height, width, rank, world, dst=dst, group=group)
