// placement_probe.hip -- does the PLACEMENT of a device block (which physical pages the driver backs it with) show in a
// micro-probe?  DESIGN section 4: the prefilter's scattering kernels run in one of two states per set of allocations, with
// identical virtual addresses, request counts and UTCL1 miss counts (profiles/r05_placement_states.md).  ROUNDS times: a
// filler of F GiB is allocated (F cycles through a few sizes, to move the driver's free lists), then a block of B GiB;
// random 4-byte writes, random 16-byte reads and a streaming write over the block are timed; everything is released.
// usage: placement_probe <block GiB> <rounds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
__global__ void k_random4w(uint32_t* __restrict__ pool, uint64_t pool_n, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) pool[mix(i) % pool_n] = (uint32_t)i;
}
__global__ void k_random16(const uint4* __restrict__ pool, uint64_t pool_n, uint64_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = pool[mix(i) % pool_n]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_fill(uint4* __restrict__ pool, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) pool[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0; const int rounds = argc > 2 ? atoi(argv[2]) : 12;
    const uint64_t bytes = (uint64_t)(gib * 1073741824.0) & ~15ULL, n = 1000000000ULL;
    uint32_t* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double fillers[] = { 0, 40, 0, 120, 7, 0, 200, 63, 0, 1.5, 90, 0 };
    for (int r = 0; r < rounds; ++r) {
        void* filler = nullptr; const double f = fillers[r % 12];
        if (f > 0) CK(hipMalloc(&filler, (size_t)(f * 1073741824.0)));
        void* pool; CK(hipMalloc(&pool, bytes));
        float ms[3]; const int grid = 256 * 16, block = 256;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fill, dim3(grid), dim3(block), 0, 0, (uint4*)pool, bytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_random4w, dim3(grid), dim3(block), 0, 0, (uint32_t*)pool, bytes / 4, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_random16, dim3(grid), dim3(block), 0, 0, (const uint4*)pool, bytes / 16, n, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[2], e0, e1));
        printf("round %2d filler %5.1f GiB block %p: fill %.2f ms (%.0f GB/s)  random 4-byte writes %.2f ms (%.1f G/s)  random 16-byte reads %.2f ms (%.1f G/s)\n",
               r, f, pool, ms[0], bytes / ms[0] / 1e6, ms[1], n / ms[1] / 1e6, ms[2], n / ms[2] / 1e6);
        fflush(stdout);
        CK(hipFree(pool)); if (filler) CK(hipFree(filler));
    }
    return 0;
}
