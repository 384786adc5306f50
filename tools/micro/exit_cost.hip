// Developer micro-benchmark: what a process leaves behind for the NEXT process's hipInit.
// exit_cost GB mode   -- allocates GB gigabytes (hipMalloc, touched), then mode 0: _exit at once; 1: hipFree, then _exit;
//                        2: hipFree + hipDeviceReset, then _exit.  Run tools/micro/init_cost right after it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int gb = argc > 1 ? atoi(argv[1]) : 0, mode = argc > 2 ? atoi(argv[2]) : 0;
    double t0 = now();
    (void)hipInit(0); (void)hipSetDevice(0);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t1 = now();
    std::vector<void*> blocks;
    for (int i = 0; i < gb; i += 4) { void* p = nullptr; if (hipMalloc(&p, 4ull << 30) != hipSuccess) break; (void)hipMemsetAsync(p, 1, 4ull << 30, s); blocks.push_back(p); }
    (void)hipStreamSynchronize(s);
    double t2 = now();
    if (mode >= 1) for (void* p : blocks) (void)hipFree(p);
    if (mode >= 2) { (void)hipStreamDestroy(s); (void)hipDeviceReset(); }
    double t3 = now();
    printf("A: init %.0f ms, alloc+touch %d GB %.0f ms, explicit release %.0f ms\n", (t1 - t0) * 1e3, gb, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    fflush(stdout);
    _exit(0);
}
