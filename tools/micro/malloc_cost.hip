// malloc_cost.hip -- developer tool: what a cold allocation of tens of GB costs on this box (the CLI's wall time was
// dominated by it).  One strategy per process: malloc_cost <GiB> <mode>, mode = big | pieces | async | vmm
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    const char* mode = argc > 2 ? argv[2] : "big";
    const size_t bytes = (size_t)(gib * 1073741824.0);
    double t0 = now(); CK(hipFree(0)); double t1 = now(); printf("%-7s context %.3f s", mode, t1 - t0);
    t0 = now();
    if (!strcmp(mode, "big")) {
        void* p; CK(hipMalloc(&p, bytes)); t1 = now(); printf("  alloc %.3f s", t1 - t0);
        t0 = now(); CK(hipMemset(p, 1, bytes)); CK(hipDeviceSynchronize()); t1 = now(); printf("  memset %.3f s", t1 - t0);
    } else if (!strcmp(mode, "pieces")) {
        const int n = argc > 3 ? atoi(argv[3]) : 32; std::vector<void*> v(n);
        for (auto& q : v) CK(hipMalloc(&q, bytes / n));
        t1 = now(); printf("  alloc %d pieces %.3f s", n, t1 - t0);
        t0 = now(); for (auto& q : v) CK(hipMemset(q, 1, bytes / n)); CK(hipDeviceSynchronize()); t1 = now(); printf("  memset %.3f s", t1 - t0);
    } else if (!strcmp(mode, "async")) {
        hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0)); uint64_t thr = ~0ULL; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        void* p; CK(hipMallocAsync(&p, bytes, 0)); CK(hipStreamSynchronize(0)); t1 = now(); printf("  alloc %.3f s", t1 - t0);
        t0 = now(); CK(hipMemset(p, 1, bytes)); CK(hipDeviceSynchronize()); t1 = now(); printf("  memset %.3f s", t1 - t0);
    }
    printf("\n");
    return 0;
}
