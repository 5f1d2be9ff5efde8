// Micro-benchmark: can the first records of a table live in LDS and be read by the SAME load instruction as the rest?
// A search visits the top of its hierarchy in every ray; the vector L1 serves one 16-byte piece per lane per cycle, LDS
// eight times that.  A flat load goes to LDS or to memory by its address (the shared aperture), so one instruction can
// serve both kinds of lane.
//   global    every lane reads 7 x 16 bytes of its record with global loads (what the node step does)
//   flat      the same through flat loads, all addresses in memory (what the flat form itself costs)
//   flat+lds  flat loads; a lane whose record is one of the first `top` reads the copy in LDS
//   branch    two paths: ds_read for the lanes in the top, global loads for the others
//   awaited   global loads, the first piece awaited before the other six are issued (are loads of a line that is on its
//             way dearer than loads of a line that is there?)
// `percent` of the visits go to the first `top` records.  The next index depends on the data read.
// build: hipcc -O3 --offload-arch=gfx950 -o micro_flat_lds micro_flat_lds.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef unsigned int Uint4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15; return x; }

constexpr int TOP = 85;

template <int MODE>
__global__ __launch_bounds__(256, 4) void gather_kernel(const char* __restrict__ table, unsigned records, int iterations, unsigned percent,
                                                        unsigned* __restrict__ out) {
    __shared__ Uint4 top[TOP * 8];
    for (int i = threadIdx.x; i < TOP * 8; i += 256) top[i] = reinterpret_cast<const Uint4*>(table)[i];
    __syncthreads();
    unsigned index = mix(blockIdx.x * 256u + threadIdx.x) % records;
    unsigned sum = 0;
    for (int it = 0; it < iterations; ++it) {
        Uint4 piece[7];
        if (MODE == 0) {
            const char* rec = table + ((size_t)index << 7);
#pragma unroll
            for (int k = 0; k < 7; ++k) piece[k] = *reinterpret_cast<const Uint4*>(rec + 16 * k);
        } else if (MODE == 1 || MODE == 2) {
            const char* in_memory = table + ((size_t)index << 7);
            const char* in_lds = reinterpret_cast<const char*>(top) + (index << 7);     // generic pointer into the shared aperture
            const char* rec = (MODE == 2 && index < (unsigned)TOP) ? in_lds : in_memory;
            asm volatile("" : "+v"(rec));      // the compiler may not prove which address space this is
#pragma unroll
            for (int k = 0; k < 7; ++k) piece[k] = *reinterpret_cast<const Uint4*>(rec + 16 * k);
        } else if (MODE == 4) {
            // the record's line is fetched by one load; the other six are issued once it has arrived
            const char* rec = table + ((size_t)index << 7);
            piece[0] = *reinterpret_cast<const Uint4*>(rec);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(piece[0]) :: "memory");
#pragma unroll
            for (int k = 1; k < 7; ++k) piece[k] = *reinterpret_cast<const Uint4*>(rec + 16 * k);
        } else {
            if (index < (unsigned)TOP) {
#pragma unroll
                for (int k = 0; k < 7; ++k) piece[k] = top[index * 8 + k];
            } else {
                const char* rec = table + ((size_t)index << 7);
#pragma unroll
                for (int k = 0; k < 7; ++k) piece[k] = *reinterpret_cast<const Uint4*>(rec + 16 * k);
            }
        }
        unsigned h = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) h += piece[k].x ^ piece[k].y ^ piece[k].z ^ piece[k].w;
        sum += h;
        const unsigned next = mix(index + h + it);
        index = (next % 100u < percent) ? (next >> 8) % (unsigned)TOP : (next >> 8) % records;
    }
    out[blockIdx.x * 256u + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const size_t megabytes = argc > 1 ? std::atoi(argv[1]) : 27;
    const int iterations = argc > 2 ? std::atoi(argv[2]) : 2000;
    const unsigned records = (unsigned)(megabytes * 1024 * 1024 / 128);
    std::vector<unsigned> host((size_t)records * 32);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (unsigned)(i * 2654435761u);
    char* table = nullptr;
    unsigned* out = nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 4;
    CHECK(hipMalloc(&table, (size_t)records * 128));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(table, host.data(), (size_t)records * 128, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const char* names[5] = {"global", "flat (all in memory)", "flat + lds", "branch: ds_read | global", "global, first piece awaited"};
    std::printf("table %zu MB, %d CUs x 16 waves, %d dependent visits per lane, first %d records also in LDS\n", megabytes, prop.multiProcessorCount,
                iterations, TOP);
    const unsigned shares[6] = {0, 20, 40, 60, 80, 100};
    for (unsigned share : shares) {
        for (int mode = 0; mode < 5; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, share, out);
                if (mode == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, share, out);
                if (mode == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, share, out);
                if (mode == 3) hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, share, out);
                if (mode == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(256), 0, 0, table, records, iterations, share, out);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep == 1) {
                    const double visits = (double)blocks * 256 * iterations;
                    std::printf("%2u %% in the top  %-26s %8.3f ms  %6.2f cycles per record per CU (2.4 GHz)\n", share, names[mode], ms,
                                ms * 1e-3 * 2.4e9 * prop.multiProcessorCount / visits);
                }
            }
        }
    }
    return 0;
}
