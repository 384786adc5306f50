// vg_core.cpp — error channel, device selection, the library stream and HIP-event profiling.
#include "vg_common.h"
#include <chrono>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <map>
#include <algorithm>
#include <thread>
#include <mutex>

static thread_local char g_err[1024] = "";

void vg_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* vg_last_error(void) { return g_err; }
extern "C" const char* vg_version(void) { return "vclust-mi355x 0.1.0 (gfx950, HIP)"; }
extern "C" void vg_free(void* p) { free(p); }

static int g_device = -1;
static hipStream_t g_stream = nullptr;
static bool g_pool_ready = false;          // release threshold of the current device's memory pool set (vg_dev_alloc)

extern "C" int vg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int vg_set_device(int device) {
    VG_API_BEGIN
    int n = vg_device_count();
    if (n <= 0) throw vg_error(VG_ENODEV, "no HIP device visible: libvclust_gpu has no CPU fallback");
    if (device < 0 || device >= n) throw vg_error(VG_EINVAL, "device index out of range");
    if (g_device >= 0 && device != g_device) {
        // cached blocks belong to the device they were allocated on: give them back before switching
        // (genome sets re-upload themselves on their next use, vg_genomes_to_device)
        vg_dev_trim();
        if (g_stream) { (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
        g_pool_ready = false;
    }
    VG_HIP(hipSetDevice(device));
    g_device = device;
    VG_API_END
}

// HIP's current device is a per-thread setting: every thread that comes through here (helper threads of the
// library included) is bound to the library's device
void vg_require_device() {
    static std::mutex mu;
    thread_local int bound = -1;
    if (g_device >= 0 && bound == g_device) return;
    std::lock_guard<std::mutex> lk(mu);
    if (g_device < 0) {
        int n = vg_device_count();
        if (n <= 0) throw vg_error(VG_ENODEV, "no HIP device visible: libvclust_gpu has no CPU fallback");
        VG_HIP(hipSetDevice(0));
        g_device = 0;
    } else VG_HIP(hipSetDevice(g_device));
    bound = g_device;
}

hipStream_t vg_stream() {
    vg_require_device();
    if (!g_stream) {
        static std::mutex mu; std::lock_guard<std::mutex> lk(mu);
        if (!g_stream) VG_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    }
    return g_stream;
}

void vg_host_mark(const char* what) {
    static const bool on = [] { const char* e = getenv("VG_HOST_TRACE"); return e && *e && *e != '0'; }();
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[vg host] %-28s +%.3f ms  @%.3f\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
            std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
    last = now;
}

int vg_host_threads() {
    static int n = [] { const char* e = getenv("VG_HOST_THREADS"); int v = e ? atoi(e) : (int)std::min(8u, std::thread::hardware_concurrency()); return std::max(1, std::min(v, 64)); }();
    return n;
}

// ---------------------------------------------------------------- caching device allocator
namespace {
struct block_info { size_t size; int device; };
std::mutex g_alloc_mu;
std::multimap<size_t, void*> g_free_blocks;          // size -> block (blocks of the current device only)
std::map<void*, block_info> g_block_size;            // every live or cached block and the device that owns it
size_t g_cached_bytes = 0, g_live_bytes = 0;
constexpr size_t ALLOC_GRAN = 1 << 12;
const bool g_alloc_trace = [] { const char* e = getenv("VG_ALLOC_TRACE"); return e && *e && *e != '0'; }();
}

// Blocks come from the device's stream-ordered memory pool (hipMallocAsync on the library stream, release threshold
// "never"): one hipMalloc of tens of GB costs ~30 ms per GiB on this platform (0.97 s for 32 GiB, measured with
// tools/micro/malloc_cost.hip), the pool hands out the same 32 GiB in 45 ms -- the CLI's wall time was mostly that.
// VG_ALLOC=malloc goes back to plain hipMalloc / hipFree.
static const bool g_pool_alloc = [] { const char* e = getenv("VG_ALLOC"); return !(e && !strcmp(e, "malloc")); }();
static hipError_t raw_alloc(void** p, size_t bytes) {
    if (g_pool_alloc) {
        hipStream_t s = vg_stream();
        if (!g_pool_ready) {
            hipMemPool_t pool; uint64_t thr = ~0ULL;
            if (hipDeviceGetDefaultMemPool(&pool, g_device) == hipSuccess) (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
            g_pool_ready = true;
        }
        hipError_t e = hipMallocAsync(p, bytes, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);     // usable from any stream (and by blocking copies) from here on
        return e;
    }
    return hipMalloc(p, bytes);
}
static void raw_free(void* p) {
    if (g_pool_alloc) { if (hipFreeAsync(p, vg_stream()) == hipSuccess) return; (void)hipGetLastError(); }
    (void)hipFree(p);
}

void* vg_dev_alloc(size_t bytes) {
    vg_require_device();
    size_t want = (bytes + ALLOC_GRAN - 1) / ALLOC_GRAN * ALLOC_GRAN;
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        auto it = g_free_blocks.lower_bound(want);
        if (it != g_free_blocks.end() && it->first <= want + want / 4 + (1 << 20)) {
            void* p = it->second; g_cached_bytes -= it->first; g_live_bytes += it->first; g_free_blocks.erase(it);
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = raw_alloc(&p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();                      // the failure is handled here: do not leave it as the sticky "last error"
        if (g_alloc_trace) fprintf(stderr, "[vg alloc] allocation of %.1f MB failed: trimming %.1f GB of cached blocks (live %.1f GB)\n", want / 1048576.0, g_cached_bytes / 1073741824.0, g_live_bytes / 1073741824.0);
        vg_dev_trim();                                // give cached blocks back and retry once
        e = raw_alloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); throw vg_error(VG_ENOMEM, std::string("device allocation: ") + hipGetErrorString(e)); }
    }
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_block_size[p] = { want, g_device };
    g_live_bytes += want;
    if (g_alloc_trace && want >= (64u << 20)) fprintf(stderr, "[vg alloc] new block %.1f MB (live %.1f GB, cached %.1f GB)\n", want / 1048576.0, g_live_bytes / 1073741824.0, g_cached_bytes / 1073741824.0);
    return p;
}

void vg_dev_free(void* p) {
    if (!p) return;
    std::unique_lock<std::mutex> lk(g_alloc_mu);
    auto it = g_block_size.find(p);
    if (it == g_block_size.end()) { lk.unlock(); raw_free(p); return; }
    g_live_bytes -= it->second.size;
    if (it->second.device != g_device) {
        // a block of another device (a genome set freed after vg_set_device): it must not be handed out here
        g_block_size.erase(it); lk.unlock(); (void)hipFree(p); return;       // (hipFree takes pool memory of any device)
    }
    g_free_blocks.emplace(it->second.size, p); g_cached_bytes += it->second.size;
}

void vg_dev_trim() {
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        for (auto& kv : g_free_blocks) { blocks.push_back(kv.second); g_block_size.erase(kv.second); }
        g_free_blocks.clear(); g_cached_bytes = 0;
    }
    if (!blocks.empty()) {
        (void)hipDeviceSynchronize();
        for (void* b : blocks) raw_free(b);
        if (g_pool_alloc) {
            // the pool keeps what it is given back: return it to the driver as well
            (void)hipStreamSynchronize(vg_stream());
            hipMemPool_t pool; if (hipDeviceGetDefaultMemPool(&pool, g_device) == hipSuccess) (void)hipMemPoolTrimTo(pool, 0);
        }
    }
}

extern "C" void vg_release_device_memory(void) { vg_dev_trim(); }
extern "C" int vg_copy(void* dst, const void* src, int64_t bytes, int to_host) {
    VG_API_BEGIN
    vg_require_device();
    if (bytes > 0) VG_HIP(hipMemcpy(dst, src, (size_t)bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice));
    VG_API_END
}

// ---------------------------------------------------------------- profiling
struct prof_entry { double ms = 0; int64_t launches = 0; double bytes = 0; int order = 0; };
struct pending_ev { std::string name; hipEvent_t e0, e1; double bytes; };
static bool g_prof = false;
static std::map<std::string, prof_entry> g_prof_tab;
static std::vector<pending_ev> g_pending;
static std::mutex g_prof_mu;

bool vg_profile_on() { return g_prof; }

vg_prof_scope::vg_prof_scope(const char* nm, double b, hipStream_t st) : name(nm), bytes(b), on(g_prof), stream(st) {
    if (!on) return;
    if (!stream) stream = vg_stream();
    hipStream_t s = stream;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(e0, s);
}
vg_prof_scope::~vg_prof_scope() {
    if (!on) return;
    (void)hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_pending.push_back({ name, e0, e1, bytes });
}

static void prof_drain() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& p : g_pending) {
        (void)hipEventSynchronize(p.e1);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            auto& e = g_prof_tab[p.name];
            if (e.launches == 0) e.order = (int)g_prof_tab.size();
            e.ms += ms; e.launches++; e.bytes += p.bytes;
        }
        (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1);
    }
    g_pending.clear();
}

extern "C" void vg_profile_enable(int on) { g_prof = on != 0; }
extern "C" void vg_profile_reset(void) { prof_drain(); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof_tab.clear(); }
extern "C" int vg_profile_get(vg_kernel_time* out, int cap) {
    prof_drain();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& kv : g_prof_tab) {
        if (n < cap) {
            memset(&out[n], 0, sizeof(out[n]));
            strncpy(out[n].name, kv.first.c_str(), sizeof(out[n].name) - 1);
            out[n].total_ms = kv.second.ms; out[n].launches = kv.second.launches; out[n].bytes = kv.second.bytes;
        }
        ++n;
    }
    return n;
}
