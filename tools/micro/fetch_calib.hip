// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this
// repository (developer tool; MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
//   k_stream16   every lane reads consecutive 16 bytes (the guide's calibrated case: FETCH_SIZE = 1/2 of the bytes)
//   k_random16   every lane reads 16 bytes at a random 16-byte aligned address of the pool (the LZ parse's probes:
//                query chunk, bucket bounds, index entries, reference chunk)
//   k_random4w   every lane writes 4 bytes at a random address (the row-pointer scatter of the bucket kernel)
//   k_random4r   every lane reads 4 bytes at a random address (genome lookups, row-pointer gathers)
// usage: fetch_calib <pool GiB> <accesses (millions)>; run under rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --kernel-trace
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
__global__ void k_stream16(const uint4* __restrict__ pool, uint64_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = pool[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_random16(const uint4* __restrict__ pool, uint64_t pool_n, uint64_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = pool[mix(i) % pool_n]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_random4r(const uint32_t* __restrict__ pool, uint64_t pool_n, uint64_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc ^= pool[mix(i) % pool_n];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_random4w(uint32_t* __restrict__ pool, uint64_t pool_n, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) pool[mix(i) % pool_n] = (uint32_t)i;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 40.0;
    const uint64_t n = (uint64_t)((argc > 2 ? atof(argv[2]) : 1000.0) * 1e6);
    const uint64_t bytes = (uint64_t)(gib * 1073741824.0) & ~15ULL;
    void* pool; uint32_t* sink;
    CK(hipMalloc(&pool, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(pool, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 16, block = 256; float ms;
    const uint64_t n_stream = bytes / 16;
#define RUN(name, launch, useful) CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-12s accesses %llu useful_bytes %llu ms %.3f  useful GB/s %.1f\n", name, (unsigned long long)(useful / 16), (unsigned long long)(useful), ms, (double)(useful) / ms / 1e6);
    RUN("k_stream16", hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(block), 0, 0, (const uint4*)pool, n_stream, sink), n_stream * 16)
    RUN("k_random16", hipLaunchKernelGGL(k_random16, dim3(grid), dim3(block), 0, 0, (const uint4*)pool, bytes / 16, n, sink), n * 16)
    RUN("k_random4r", hipLaunchKernelGGL(k_random4r, dim3(grid), dim3(block), 0, 0, (const uint32_t*)pool, bytes / 4, n, sink), n * 4)
    RUN("k_random4w", hipLaunchKernelGGL(k_random4w, dim3(grid), dim3(block), 0, 0, (uint32_t*)pool, bytes / 4, n), n * 4)
    return 0;
}
