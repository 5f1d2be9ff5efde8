"""Row sharding of one frame over the GPUs of a node and the gather of the accumulation buffer.

The reference parallelises by handing image rows to worker threads (src/main.cpp:1001-1020);
each pixel is written by exactly one worker.  Here a rank plays the worker: rows are dealt in
blocks of PT_ROW_BLOCK, block b to rank b % world (PtFrame.shard_index / shard_count), so each
pixel's samples are still summed by one owner in ascending order and the assembled buffer is
byte-identical to a single-GPU render.  The only exchange is the final gather of each rank's
rows to rank 0 (RCCL over xGMI on GPUs, gloo on CPU): 32 B per pixel, once per frame, or 4 B per
pixel when only the tone-mapped image is wanted (`gather_image8`).
"""
import functools

import torch
import torch.distributed as dist

from .ctypes_abi import PT_ROW_BLOCK


def shard_rows(height, rank, world, row_begin=0):
    """Row indices (ascending) that `rank` owns out of rows [row_begin, height)."""
    rows = torch.arange(row_begin, height)
    block = (rows - row_begin) // PT_ROW_BLOCK
    return rows[block % world == rank]


@functools.lru_cache(maxsize=64)
def _rows_on_device(height, rank, world, device):
    """shard_rows(...) as an index tensor on `device`, made once: a frame loop calls the gather every frame."""
    return shard_rows(height, rank, world).to(device)


def pack_rows(accum, rows, length):
    """The send buffer of one rank: its rows of `accum` (row indices `rows`, a tensor on accum's device),
    packed in ascending order into `length` rows (every rank sends the same size; the tail is zero)."""
    send = torch.zeros((length,) + tuple(accum.shape[1:]), dtype=accum.dtype, device=accum.device)
    send[: len(rows)] = accum.index_select(0, rows)
    return send


def unpack_rows(accum, rows, received):
    """Puts the rows another rank packed with `pack_rows` back at their places (`rows`) in `accum`."""
    accum.index_copy_(0, rows, received[: len(rows)])
    return accum


def gather_rows(accum, height, width, rank, world, dst=0, group=None):
    """accum: (height, width, 4) tensor (float64 sums, or uint8 image) holding this rank's rows
    (others are ignored).  After the call rank `dst` holds every rank's rows in place; other ranks
    are unchanged."""
    if world == 1:
        return accum
    assert accum.shape == (height, width, 4)
    rows_of = [_rows_on_device(height, r, world, accum.device) for r in range(world)]
    longest = max(len(rows) for rows in rows_of)
    send = pack_rows(accum, rows_of[rank], longest)
    if rank == dst:
        received = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, received, dst=dst, group=group)
        for r in range(world):
            if r != dst:
                unpack_rows(accum, rows_of[r], received[r])
    else:
        dist.gather(send, None, dst=dst, group=group)
    return accum


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
