// Micro-benchmark: what a CU pays per 16-byte piece when every lane reads pieces of its own 128-byte record
// (a tree search's node fetch), by number of pieces and by cache policy of the load.
//   plain   global_load_dwordx4
//   nt      the same, nontemporal (slc/nt bit)
//   sc1     agent-scope relaxed atomic loads of 8 bytes x 2 (bypass the vector L1)
// The next index depends on the data read.  build: hipcc -O3 --offload-arch=gfx950 -o micro_l1_pieces micro_l1_pieces.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned int Uint4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15; return x; }

template <int PIECES, int POLICY, int STRIDE>
__global__ __launch_bounds__(256, 3) void gather_kernel(const char* __restrict__ table, unsigned records, int iterations, unsigned* __restrict__ out) {
    unsigned index = mix(blockIdx.x * 256u + threadIdx.x) % records;
    unsigned sum = 0;
    for (int it = 0; it < iterations; ++it) {
        Uint4 piece[PIECES];
        const char* rec = table + (size_t)index * STRIDE;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) {
            const Uint4* p = reinterpret_cast<const Uint4*>(rec + 16 * k);
            if (POLICY == 0) piece[k] = *p;
            else if (POLICY == 1) piece[k] = __builtin_nontemporal_load(p);
            else {
                const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
                const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                piece[k] = Uint4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
            }
        }
        unsigned h = 0;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) h += piece[k].x ^ piece[k].y ^ piece[k].z ^ piece[k].w;
        sum += h;
        index = mix(index + h + it) % records;
    }
    out[blockIdx.x * 256u + threadIdx.x] = sum;
}

template <int PIECES, int POLICY, int STRIDE>
void run(const char* name, const char* table, size_t table_bytes, int iterations, unsigned* out, int blocks, int cus) {
    const unsigned records = (unsigned)(table_bytes / STRIDE);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((gather_kernel<PIECES, POLICY, STRIDE>), dim3(blocks), dim3(256), 0, 0, table, records, iterations, out);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
    }
    const double visits = (double)blocks * 256 * iterations;
    const double cycles = ms * 1e-3 * 2.4e9 * cus / visits;
    std::printf("%-28s %d pieces of a %3d-byte record  %8.3f ms  %6.2f cycles per record per CU  %5.2f per piece\n", name, PIECES, STRIDE, ms, cycles, cycles / PIECES);
}

int main(int argc, char** argv) {
    const size_t megabytes = argc > 1 ? std::atoi(argv[1]) : 27;
    const int iterations = argc > 2 ? std::atoi(argv[2]) : 2000;
    const size_t bytes = megabytes * 1024 * 1024;
    std::vector<unsigned> host(bytes / 4);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (unsigned)(i * 2654435761u);
    char* table = nullptr; unsigned* out = nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 3;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(table, host.data(), bytes, hipMemcpyHostToDevice));
    std::printf("table %zu MB, %d CUs x 12 waves, %d dependent visits per lane\n", megabytes, cus, iterations);
    run<7, 0, 128>("plain", table, bytes, iterations, out, blocks, cus);
    run<5, 0, 128>("plain", table, bytes, iterations, out, blocks, cus);
    run<4, 0, 128>("plain", table, bytes, iterations, out, blocks, cus);
    run<4, 0, 64>("plain", table, bytes, iterations, out, blocks, cus);
    run<3, 0, 48>("plain", table, bytes, iterations, out, blocks, cus);
    run<2, 0, 32>("plain", table, bytes, iterations, out, blocks, cus);
    run<1, 0, 16>("plain", table, bytes, iterations, out, blocks, cus);
    run<7, 1, 128>("nontemporal", table, bytes, iterations, out, blocks, cus);
    run<3, 1, 48>("nontemporal", table, bytes, iterations, out, blocks, cus);
    run<7, 2, 128>("agent-scope (sc1, 8 B x 2)", table, bytes, iterations, out, blocks, cus);
    run<3, 2, 48>("agent-scope (sc1, 8 B x 2)", table, bytes, iterations, out, blocks, cus);
    return 0;
}
