// malloc_cost.hip -- developer tool: what a cold allocation of tens of GB costs on this box (the CLI's wall time was
// dominated by it).  One strategy per process: malloc_cost <GiB> <mode>, mode = big | pieces <n> | async | vmm <GiB per chunk>
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
    else if (!strcmp(mode, "vmm")) {
        // virtual memory management API: one address range, physical chunks of `piece` GiB mapped into it
        const double piece_gib = argc > 3 ? atof(argv[3]) : 1.0;
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        size_t piece = (size_t)(piece_gib * 1073741824.0); piece = (piece + gran - 1) / gran * gran;
        const size_t n = (bytes + piece - 1) / piece, total = n * piece;
        void* va = nullptr; CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (size_t i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], piece, &prop, 0)); CK(hipMemMap((char*)va + i * piece, piece, 0, h[i], 0)); }
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
        t1 = now(); printf("  alloc %zu x %.2f GiB (granularity %zu KiB) %.3f s", n, piece / 1073741824.0, gran >> 10, t1 - t0);
        t0 = now(); CK(hipMemset(va, 1, total)); CK(hipDeviceSynchronize()); t1 = now(); printf("  memset %.3f s", t1 - t0);
        size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); printf("  free before release %.1f GiB", fr / 1073741824.0);
        t0 = now();
        CK(hipMemUnmap(va, total)); for (size_t i = 0; i < n; ++i) CK(hipMemRelease(h[i])); CK(hipMemAddressFree(va, total));
        t1 = now(); CK(hipMemGetInfo(&fr, &tot)); printf("  release %.3f s, free after %.1f GiB", t1 - t0, fr / 1073741824.0);
        t0 = now(); void* p2; CK(hipMalloc(&p2, 1 << 30)); t1 = now(); printf("  next hipMalloc(1 GiB) %.3f s", t1 - t0);
    }
    printf("\n");
    return 0;
}
