// Developer micro-benchmark: where a cold process spends its HIP start-up.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/init_cost tools/micro/init_cost.hip && /tmp/init_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop() {}
int main() {
    double t = now(), t1;
#define MARK(what) t1 = now(); printf("%-28s %8.1f ms\n", what, (t1 - t) * 1e3); t = t1;
    hipInit(0); MARK("hipInit");
    int n = 0; hipGetDeviceCount(&n); MARK("hipGetDeviceCount");
    hipSetDevice(0); MARK("hipSetDevice");
    hipFree(nullptr); MARK("hipFree(nullptr)");
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); MARK("hipStreamCreate");
    void* p = nullptr; hipMalloc(&p, 4096); MARK("hipMalloc 4 KiB");
    hipMemsetAsync(p, 0, 4096, s); hipStreamSynchronize(s); MARK("first memset + sync");
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s); hipStreamSynchronize(s); MARK("first kernel + sync");
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); MARK("second stream");
    hipMemsetAsync(p, 0, 4096, s2); hipStreamSynchronize(s2); MARK("memset on second stream");
    return 0;
}
