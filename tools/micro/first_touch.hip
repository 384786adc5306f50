// first_touch.hip -- developer tool: what the FIRST use of device memory costs on a box, and what re-use costs.
// The amdgpu driver hands out cleared VRAM: blocks it has never given out (or has not wiped since their last owner)
// are cleared inside the allocation call.  first_touch <mode> <GiB per piece> <pieces> [<round 2: 1|0>]
//   mode = malloc | vmm; prints the time of every piece, then frees everything and (round 2) allocates again.
// Run it as the first GPU process of a fresh box, then again: the second process shows what a following process pays.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct piece { void* p = nullptr; hipMemGenericAllocationHandle_t h{}; size_t bytes = 0; bool vmm = false; };
static int get(piece& pc, size_t bytes, bool vmm) {
    pc.bytes = bytes; pc.vmm = vmm;
    if (!vmm) { CK(hipMalloc(&pc.p, bytes)); return 0; }
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    CK(hipMemAddressReserve(&pc.p, bytes, 0, nullptr, 0));
    CK(hipMemCreate(&pc.h, bytes, &prop, 0));
    CK(hipMemMap(pc.p, bytes, 0, pc.h, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(pc.p, bytes, &acc, 1));
    return 0;
}
static int put(piece& pc) {
    if (!pc.p) return 0;
    if (!pc.vmm) { CK(hipFree(pc.p)); }
    else { CK(hipMemUnmap(pc.p, pc.bytes)); CK(hipMemRelease(pc.h)); CK(hipMemAddressFree(pc.p, pc.bytes)); }
    pc.p = nullptr; return 0;
}
int main(int argc, char** argv) {
    const bool vmm = argc > 1 && !strcmp(argv[1], "vmm");
    const double gib = argc > 2 ? atof(argv[2]) : 8.0;
    const int n = argc > 3 ? atoi(argv[3]) : 8;
    const int round2 = argc > 4 ? atoi(argv[4]) : 1;
    const size_t bytes = (size_t)(gib * 1073741824.0) / (2u << 20) * (2u << 20);
    double t0 = now(); CK(hipFree(0)); printf("%s context %.3f s\n", vmm ? "vmm" : "malloc", now() - t0);
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); printf("free %.1f of %.1f GiB\n", fr / 1073741824.0, tot / 1073741824.0);
    std::vector<piece> v((size_t)n);
    for (int r = 0; r < 1 + round2; ++r) {
        double sum = 0;
        printf("round %d:", r + 1);
        for (int i = 0; i < n; ++i) {
            t0 = now(); if (get(v[(size_t)i], bytes, vmm)) return 1; const double dt = now() - t0; sum += dt;
            printf(" %.0f", dt * 1e3);
        }
        printf(" ms  | total %.3f s for %.0f GiB = %.1f ms per GiB\n", sum, gib * n, sum * 1e3 / (gib * n));
        // first write to the memory (a fill kernel over every piece)
        t0 = now();
        for (auto& pc : v) CK(hipMemsetAsync(pc.p, 1, pc.bytes, 0));
        CK(hipDeviceSynchronize());
        printf("   fill of all pieces %.3f s\n", now() - t0);
        t0 = now(); for (auto& pc : v) if (put(pc)) return 1; printf("   release %.3f s\n", now() - t0);
    }
    return 0;
}
