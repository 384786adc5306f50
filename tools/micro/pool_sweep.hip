// pool_sweep.hip -- does the size of a randomly read working set move the request rate?  (developer tool, round 6)
// The LZ parse is bound by the NUMBER of random 64-byte requests it sends past the L2 (1.93 G per launch at phage-100k,
// 50 G/s: DESIGN.md section 4); its resident working set -- ~900 reference indexes of 0.4 MB plus the sequences of 8 192
// pairs -- is ~600 MB against 4 MB of L2 per XCD and the 256 MB Infinity Cache.  Before building a more compact index
// "so that it fits the Infinity Cache", this measures what fitting would buy: one random 16-byte read per lane out of a pool
// of 8 MB ... 8 GB (one pool for the whole device, like reference indexes shared by the waves of an XCD group), and the same
// with every XCD reading its OWN eighth of the pool (block b runs on XCD b % 8).  Prints G requests/s per pool size.
// usage: pool_sweep [accesses, millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
// lines = 64-byte lines of the pool; per_xcd: the pool is cut into eight parts, a workgroup reads the part of its XCD
template <bool PER_XCD>
__global__ void __launch_bounds__(256) k_random_lines(const uint4* __restrict__ pool, uint64_t lines, uint64_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const uint64_t part = PER_XCD ? lines / 8 : lines, base = PER_XCD ? (uint64_t)(blockIdx.x % 8) * part : 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix(i);
        const uint4 v = pool[(base + h % part) * 4 + ((h >> 60) & 3)];      // one 16-byte piece of a random line
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const uint64_t n = (uint64_t)((argc > 1 ? atof(argv[1]) : 2000.0) * 1e6);
    const uint64_t max_bytes = 8ULL << 30;
    void* pool; uint32_t* sink;
    CK(hipMalloc(&pool, max_bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(pool, 1, max_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 32, block = 256; float ms;
    printf("%10s  %22s  %22s\n", "pool", "one pool: G req/s", "a part per XCD: G req/s");
    for (uint64_t mb = 8; mb <= 8192; mb *= 2) {
        const uint64_t lines = (mb << 20) / 64;
        double rate[2];
        for (int v = 0; v < 2; ++v) {
            for (int rep = 0; rep < 2; ++rep) {      // (the first repetition warms the caches the pool fits)
                CK(hipEventRecord(e0));
                if (v == 0) hipLaunchKernelGGL(k_random_lines<false>, dim3(grid), dim3(block), 0, 0, (const uint4*)pool, lines, n, sink);
                else hipLaunchKernelGGL(k_random_lines<true>, dim3(grid), dim3(block), 0, 0, (const uint4*)pool, lines, n, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            rate[v] = (double)n / ms / 1e6;
        }
        printf("%7llu MB  %22.1f  %22.1f\n", (unsigned long long)mb, rate[0], rate[1]);
    }
    return 0;
}
