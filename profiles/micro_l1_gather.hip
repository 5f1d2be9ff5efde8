// Micro-benchmark: how fast does a CU fetch 128-byte records at per-lane random indices?
//   own      every lane reads 7 x 16 bytes of ITS record (what the node step of trace_paths_kernel does)
//   coop     the 8 lanes of a group read one record together, 16 bytes each, for each of the group's 8 records
//            (8 loads per lane, every load touches 8 lines per wave instead of 64); data not redistributed
//   coop_lds the same, with the records passed through LDS so that every lane ends up with its own record
//            (8 x ds_write_b128, 7 x ds_read_b128)
//   glds     the same with loads that write LDS directly (global_load_lds_dwordx4), then 7 x ds_read_b128
// The next index depends on the data read, as a tree search's does.  Table sizes: 27 MB (L2 + Infinity Cache)
// build: hipcc -O3 --offload-arch=gfx950 -o micro_l1_gather micro_l1_gather.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float Float4 __attribute__((ext_vector_type(4)));
typedef unsigned int Uint4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15; return x; }

template <int MODE>
__global__ __launch_bounds__(256, 3) void gather_kernel(const char* __restrict__ table, unsigned records, int iterations, unsigned* __restrict__ out) {
    __shared__ Uint4 stage[4][8][64];       // per wave: 8 rounds x 64 lanes x 16 bytes = 8 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned index = mix(blockIdx.x * 256u + threadIdx.x) % records;
    unsigned sum = 0;
    for (int it = 0; it < iterations; ++it) {
        Uint4 piece[8];
        if (MODE == 0) {
            const char* rec = table + ((size_t)index << 7);
#pragma unroll
            for (int k = 0; k < 7; ++k) piece[k] = *reinterpret_cast<const Uint4*>(rec + 16 * k);
            piece[7] = Uint4{0, 0, 0, 0};
        } else {
            const int member = lane & 7, group0 = lane & ~7;
            unsigned their[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) their[k] = (unsigned)__shfl((int)index, group0 + k);
            if (MODE == 3) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const char* src = table + ((size_t)their[k] << 7) + 16 * member;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&stage[wave][k][0], 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) piece[k] = *reinterpret_cast<const Uint4*>(table + ((size_t)their[k] << 7) + 16 * member);
                if (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) stage[wave][k][lane] = piece[k];
                }
            }
            if (MODE >= 2) {
                // the record of this lane is round `member`, lanes group0 .. group0+7
#pragma unroll
                for (int k = 0; k < 7; ++k) piece[k] = stage[wave][member][group0 + k];
                piece[7] = Uint4{0, 0, 0, 0};
            }
        }
        unsigned h = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) h += piece[k].x ^ piece[k].y ^ piece[k].z ^ piece[k].w;
        sum += h;
        index = mix(index + h + it) % records;
    }
    out[blockIdx.x * 256u + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const size_t megabytes = argc > 1 ? std::atoi(argv[1]) : 27;
    const int iterations = argc > 2 ? std::atoi(argv[2]) : 2000;
    const unsigned records = (unsigned)(megabytes * 1024 * 1024 / 128);
    std::vector<unsigned> host(records * 32);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (unsigned)(i * 2654435761u);
    char* table = nullptr;
    unsigned* out = nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 3;
    CHECK(hipMalloc(&table, (size_t)records * 128));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(table, host.data(), (size_t)records * 128, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const char* names[4] = {"own 7x16B", "coop 8x16B (no redistribution)", "coop + LDS transpose", "global_load_lds + ds_read"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, out);
            if (mode == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, out);
            if (mode == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, out);
            if (mode == 3) hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, out);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep == 1) {
                const double visits = (double)blocks * 256 * iterations;
                std::printf("%-34s %8.3f ms  %7.2f G records/s  %6.2f cycles per record per CU (2.4 GHz)\n", names[mode], ms, visits / ms / 1e6,
                            ms * 1e-3 * 2.4e9 * prop.multiProcessorCount / visits);
            }
        }
    }
    return 0;
}
