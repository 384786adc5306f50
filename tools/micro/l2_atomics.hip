// l2_atomics.hip -- what a "seen twice" bit filter between the two partition levels would cost (developer tool, round 6).
// After level 1 the records of one level-1 bucket (2 M of them at 100 k genomes) are contiguous; a bit table of ~10 bits per
// record (2.5 MB) fits the 4 MB L2 of an XCD.  Pass A of the filter: every record sets bit h(k-mer) in table T1 and, when it was
// set already, the same bit in T2; a record is kept by level 2 only when its T2 bit is set (equal k-mers always meet; a
// singleton survives with the table's load factor).  This measures the rate of that pass in its intended shape: the 32 CUs
// of an XCD work on ONE bucket at a time (block b runs on XCD b % 8), every thread takes its records' hashes from a counter
// (the record stream itself is a coalesced read: not the question here) and issues atomicOr WITH RETURN on T1 + a
// conditional atomicOr on T2.  Also: pass B's bit TEST (a random 4-byte read of T2) alone.
// usage: l2_atomics [records per bucket, millions] [buckets] [table MB per bucket]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
// grid = 8 * wg_per_xcd; the workgroups of XCD x take buckets x, x + 8, ...; within a bucket the XCD's threads stride over its records
template <int MODE>   // 0 = set (T1 with return, T2 conditional), 1 = test T2
__global__ void __launch_bounds__(256) k_filter(uint32_t* __restrict__ t1, uint32_t* __restrict__ t2, uint64_t words_per_bucket, int n_buckets,
                                                uint64_t rec_per_bucket, double dup_frac, unsigned long long* __restrict__ kept) {
    const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8, wgs = gridDim.x / 8;
    const uint64_t bits = words_per_bucket * 32;
    unsigned long long mine = 0;
    for (int b = xcd; b < n_buckets; b += 8) {
        uint32_t* a1 = t1 + (uint64_t)b * words_per_bucket; uint32_t* a2 = t2 + (uint64_t)b * words_per_bucket;
        for (uint64_t i = (uint64_t)wg * 256 + threadIdx.x; i < rec_per_bucket; i += (uint64_t)wgs * 256) {
            // a fraction dup_frac of the records repeat an earlier record's k-mer (families of ~5): i -> i / 5 * 5
            uint64_t id = ((double)(mix(i ^ 0x1234) >> 11) * (1.0 / 9007199254740992.0) < dup_frac) ? i / 5 * 5 + 1000000007ULL : i;
            const uint64_t h = mix(id * 2048 + b) % bits;
            const uint32_t bit = 1u << (h & 31);
            if (MODE == 0) {
                const uint32_t old = atomicOr(a1 + (h >> 5), bit);
                if (old & bit) atomicOr(a2 + (h >> 5), bit);
            } else mine += (a2[h >> 5] & bit) ? 1 : 0;
        }
    }
    if (MODE == 1 && mine) atomicAdd(kept, mine);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const uint64_t rec = (uint64_t)((argc > 1 ? atof(argv[1]) : 2.0) * 1e6);
    const int nb = argc > 2 ? atoi(argv[2]) : 256;
    unsigned long long* kept; CK(hipMalloc(&kept, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d buckets x %.1f M records; dup fraction 0.27\n", nb, rec / 1e6);
    for (double mb : { 1.25, 2.5, 5.0 }) {
        const uint64_t wpb = (uint64_t)(mb * 1048576.0 / 4.0);
        uint32_t *t1, *t2; CK(hipMalloc(&t1, wpb * 4 * nb)); CK(hipMalloc(&t2, wpb * 4 * nb));
        for (int wg_per_xcd : { 32, 64, 128 }) {
            CK(hipMemset(t1, 0, wpb * 4 * nb)); CK(hipMemset(t2, 0, wpb * 4 * nb)); CK(hipMemset(kept, 0, 8));
            float ms_a, ms_b;
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_filter<0>, dim3(8 * wg_per_xcd), dim3(256), 0, 0, t1, t2, wpb, nb, rec, 0.27, kept);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_a, e0, e1));
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_filter<1>, dim3(8 * wg_per_xcd), dim3(256), 0, 0, t1, t2, wpb, nb, rec, 0.27, kept);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_b, e0, e1));
            unsigned long long k = 0; CK(hipMemcpy(&k, kept, 8, hipMemcpyDeviceToHost));
            const double n = (double)rec * nb;
            printf("table %.2f MB/bucket, %3d workgroups per XCD: set %.2f ms = %.1f G records/s; test %.2f ms = %.1f G/s; kept %.1f %%  (4 G records: set %.1f ms, test %.1f ms)\n",
                   mb, wg_per_xcd, ms_a, n / ms_a / 1e6, ms_b, n / ms_b / 1e6, 100.0 * (double)k / n, 4.0e9 / (n / ms_a / 1e-3) , 4.0e9 / (n / ms_b / 1e-3));
        }
        CK(hipFree(t1)); CK(hipFree(t2));
    }
    return 0;
}
