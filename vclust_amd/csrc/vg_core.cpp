// vg_core.cpp — error channel, device selection, the library stream and HIP-event profiling.
#include "vg_common.h"
#include <chrono>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <functional>
#include <algorithm>
#include <thread>
#include <mutex>
#include <atomic>

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
static hipStream_t g_side_stream = nullptr;     // a second queue for work that may run beside the library stream (buffer clears)

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
        vg_lz_drop_prepared(nullptr);
        // cached blocks belong to the device they were allocated on: give them back before switching
        // (genome sets re-upload themselves on their next use, vg_genomes_to_device)
        vg_dev_trim();
        if (g_stream) { (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
        if (g_side_stream) { (void)hipStreamDestroy(g_side_stream); g_side_stream = nullptr; }
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

// Copies between a caller's (pageable) buffer and the device.  hipMemcpyAsync on pageable memory registers the pages
// with the driver for the copy; when such a buffer is later unmapped (a freed result array, a freed task list) the
// driver's notifier evicts and restores the queues of the process, and the NEXT submission waits 12 .. 35 ms for that
// (seen as an erratic pause in front of the first kernel of vg_lz_align in a long-lived process; with the results of
// earlier calls kept alive the pause is gone).  So no caller's buffer is ever handed to the runtime: copies of 32 KiB
// and more go through the library's own pinned buffers (two of 8 MiB per direction, an event per buffer, the host
// memcpy of one chunk overlapping the transfer of the next); smaller ones use the runtime's staging path, and genome
// sets (GBs, uploaded once, freed at the end) keep their own upload.
namespace {
constexpr size_t RING = 8u << 20;
struct pin_ring {
    char* pin[2] = { nullptr, nullptr }; hipEvent_t ev[2] = { nullptr, nullptr }; bool busy[2] = { false, false }; int dev = -1;
    void ready() {
        int d = 0; VG_HIP(hipGetDevice(&d));
        if (d == dev) return;                                  // (events belong to a device: the library's device was changed)
        for (int k = 0; k < 2; ++k) {
            if (busy[k]) { (void)hipEventSynchronize(ev[k]); busy[k] = false; }
            if (ev[k]) { (void)hipEventDestroy(ev[k]); ev[k] = nullptr; }
            if (!pin[k]) VG_HIP(hipHostMalloc((void**)&pin[k], RING, hipHostMallocPortable));
            VG_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
        dev = d;
    }
    void wait(int k) { if (busy[k]) { VG_HIP(hipEventSynchronize(ev[k])); busy[k] = false; } }
};
std::mutex g_ring_mu; pin_ring g_up, g_down;
constexpr size_t STAGE_MIN = 32u << 10;
}

// asynchronous on s (the caller's buffer may be reused when the call returns)
void vg_upload_bytes(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    if (bytes < STAGE_MIN) { VG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s)); return; }
    std::lock_guard<std::mutex> lk(g_ring_mu);
    g_up.ready();
    int k = 0;
    for (size_t off = 0; off < bytes; off += RING, k ^= 1) {
        const size_t n = std::min(RING, bytes - off);
        g_up.wait(k);
        memcpy(g_up.pin[k], (const char*)src + off, n);
        VG_HIP(hipMemcpyAsync((char*)dst + off, g_up.pin[k], n, hipMemcpyHostToDevice, s));
        VG_HIP(hipEventRecord(g_up.ev[k], s)); g_up.busy[k] = true;
    }
}
// 32 KiB and more: complete when the call returns (everything queued on s before it has run); smaller: asynchronous
void vg_download_bytes(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    if (bytes < STAGE_MIN) { VG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s)); return; }
    std::lock_guard<std::mutex> lk(g_ring_mu);
    g_down.ready();
    const size_t nch = (bytes + RING - 1) / RING;
    for (size_t c = 0; c <= nch; ++c) {
        if (c < nch) {
            const int k = (int)(c & 1); const size_t off = c * RING, n = std::min(RING, bytes - off);
            VG_HIP(hipMemcpyAsync(g_down.pin[k], (const char*)src + off, n, hipMemcpyDeviceToHost, s));
            VG_HIP(hipEventRecord(g_down.ev[k], s)); g_down.busy[k] = true;
        }
        if (c >= 1) {
            const int k = (int)((c - 1) & 1); const size_t off = (c - 1) * RING, n = std::min(RING, bytes - off);
            g_down.wait(k);
            memcpy((char*)dst + off, g_down.pin[k], n);
        }
    }
}

namespace { std::mutex g_def_mu; std::vector<std::function<void()>> g_deferred; bool g_defer_on = false; }
void vg_defer_mode(bool on) { std::lock_guard<std::mutex> lk(g_def_mu); g_defer_on = on; }
static void run_detached(std::vector<std::function<void()>> fns) {
    if (fns.empty()) return;
    auto* box = new std::vector<std::function<void()>>(std::move(fns));
    try { std::thread([box] { for (auto& f : *box) { try { f(); } catch (...) {} } delete box; }).detach(); }
    catch (...) { for (auto& f : *box) { try { f(); } catch (...) {} } delete box; }
}
void vg_defer(std::function<void()> fn) {
    {
        std::lock_guard<std::mutex> lk(g_def_mu);
        if (g_defer_on) { g_deferred.push_back(std::move(fn)); return; }
    }
    std::vector<std::function<void()>> one; one.push_back(std::move(fn));
    run_detached(std::move(one));
}
void vg_deferred_start() {
    std::vector<std::function<void()>> fns;
    { std::lock_guard<std::mutex> lk(g_def_mu); fns.swap(g_deferred); }
    run_detached(std::move(fns));
}

hipStream_t vg_side_stream() {
    vg_require_device();
    if (!g_side_stream) {
        static std::mutex mu; std::lock_guard<std::mutex> lk(mu);
        if (!g_side_stream) VG_HIP(hipStreamCreateWithFlags(&g_side_stream, hipStreamNonBlocking));
    }
    return g_side_stream;
}

// time this process has spent inside the driver's allocation calls (hipMalloc / hipMemCreate + hipMemMap), all threads
static std::atomic<int64_t> g_alloc_wait_us{0};
double vg_alloc_wait_ms() { return (double)g_alloc_wait_us.load() / 1e3; }

// "[vg host] <phase> +<ms since the previous mark of this thread> ms  alloc <ms of that spent waiting for device
// allocations> ms  @<wall clock>": the allocation wait is the driver's (it clears memory as it hands it out), not the
// kernels' -- bench.py's cli_wall.breakdown_s books it on its own line
void vg_host_mark(const char* what) {
    static const bool on = [] { const char* e = getenv("VG_HOST_TRACE"); return e && *e && *e != '0'; }();
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    static std::atomic<int64_t> reported_us{0};          // every microsecond of allocation wait is printed by exactly one mark
    const auto now = std::chrono::steady_clock::now();
    const int64_t al = g_alloc_wait_us.load();
    const int64_t before = reported_us.exchange(al);
    fprintf(stderr, "[vg host] %-28s +%.3f ms  alloc %.3f ms  @%.3f\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
            (double)std::max<int64_t>(0, al - before) / 1e3, std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
    last = now;
}

// Developer switches (kernel variants, experiment knobs: DESIGN.md section 5) are honoured only when VG_DEV_SWITCHES=1 is
// set beside them: a stray VG_INDEX_PATH in a user's environment does not change which kernels run.  The knobs meant for
// users (VG_HOST_THREADS, VG_WORKSPACE_GB, VG_ONESHOT_INDEX_GB, VG_INDEX_BUDGET_GB, VG_LZ_WEAK_SEED, the traces) are read plainly.
const char* vg_dev_getenv(const char* name) {
    static const bool on = [] { const char* e = getenv("VG_DEV_SWITCHES"); return e && *e == '1'; }();
    return on ? getenv(name) : nullptr;
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
static size_t g_parked_bytes = 0;                    // blocks set aside by vg_dev_park_cache (under g_alloc_mu): the process's own memory, like live and cached ones
std::map<void*, block_info> g_block_size;            // every live or cached block and the device that owns it
size_t g_cached_bytes = 0, g_live_bytes = 0;
constexpr size_t ALLOC_GRAN = 1 << 12;
const bool g_alloc_trace = [] { const char* e = getenv("VG_ALLOC_TRACE"); return e && *e && *e != '0'; }();
}

// Device blocks come from plain hipMalloc.  What an allocation costs on this platform is not a property of the call but
// of the memory it lands on (tools/micro/first_touch.hip, profiles/r04_first_touch.txt): memory the driver holds clean
// is handed out in well under a millisecond per 8 GiB, memory it still has to wipe -- what the PREVIOUS process on the
// device released, or never-cleared memory of a fresh box -- costs 25-32 ms per GiB, paid as a stall of one allocation
// call (up to 6 s were seen when a 240 GiB process had just gone).  Nothing in the process can shorten that, so the
// cold one-shot calls of the CLI keep their footprint small instead (vg_one_shot: k-mer sub-shards under a workspace
// budget in the prefilter, small index batches in the align stage): the exposure is ~10 GB, not ~100 GB.
// The virtual memory management path (one reserved range, 2 GiB physical chunks: hipMemCreate / hipMemMap) remains behind
// VG_ALLOC=vmm for experiments: round 3 used it for the CLI's large blocks and took its speed on clean memory for a
// property of the API; two aborts inside a long test process were seen with it on for everything and never explained.
static int g_vmm_mode = [] { const char* e = vg_dev_getenv("VG_ALLOC"); return !e ? 0 : !strcmp(e, "vmm") ? 1 : !strcmp(e, "malloc") ? -1 : 0; }();
static bool g_vmm_alloc = g_vmm_mode > 0;
// one-shot mode: set by the whole-stage calls (vg_prefilter / vg_align) for their duration
static std::atomic<int> g_one_shot{0};
void vg_one_shot_begin() { ++g_one_shot; }
void vg_one_shot_end() { --g_one_shot; }
bool vg_one_shot() { return g_one_shot.load() > 0; }
namespace {
struct vmm_block { std::vector<hipMemGenericAllocationHandle_t> handles; size_t total; };
std::map<void*, vmm_block> g_vmm_blocks;
std::mutex g_vmm_mu;
constexpr size_t VMM_MIN = 1ull << 30, VMM_CHUNK = 2ull << 30;
}
static void vmm_release(void* va, vmm_block& b, size_t mapped_chunks) {
    if (mapped_chunks) (void)hipMemUnmap(va, std::min(b.total, mapped_chunks * VMM_CHUNK));
    for (auto& h : b.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(va, b.total);
}
static hipError_t vmm_alloc(void** p, size_t bytes) {
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = g_device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) return e;
    if (gran < (2u << 20)) gran = 2u << 20;
    vmm_block b; b.total = (bytes + gran - 1) / gran * gran;
    void* va = nullptr;
    e = hipMemAddressReserve(&va, b.total, 0, nullptr, 0);
    if (e != hipSuccess) return e;
    size_t mapped = 0;
    for (size_t off = 0; off < b.total; off += VMM_CHUNK) {
        const size_t n = std::min(VMM_CHUNK, b.total - off);
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, n, &prop, 0);
        if (e != hipSuccess) break;
        b.handles.push_back(h);
        e = hipMemMap((char*)va + off, n, 0, h, 0);
        if (e != hipSuccess) break;
        ++mapped;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, b.total, &acc, 1);
    }
    if (e != hipSuccess) { vmm_release(va, b, mapped); return e; }
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    g_vmm_blocks[va] = std::move(b);
    *p = va;
    return hipSuccess;
}
static hipError_t raw_alloc_untimed(void** p, size_t bytes, const char** path) {
    if (g_vmm_alloc && bytes >= VMM_MIN) {
        *path = "vmm";
        const hipError_t e = vmm_alloc(p, bytes);
        if (e == hipSuccess || e == hipErrorOutOfMemory) return e;
        (void)hipGetLastError();                      // the API is not usable here: plain allocation
    }
    *path = "hipMalloc";
    return hipMalloc(p, bytes);
}
static hipError_t raw_alloc(void** p, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    const char* path = "";
    const hipError_t e = raw_alloc_untimed(p, bytes, &path);
    const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    g_alloc_wait_us += us;
    if (g_alloc_trace && bytes >= (256u << 20))
        fprintf(stderr, "[vg alloc] %s of %.2f GiB took %.1f ms (%.1f ms per GiB)%s\n", path, bytes / 1073741824.0, us / 1e3,
                us / 1e3 / (bytes / 1073741824.0), e == hipSuccess ? "" : " -- FAILED");
    return e;
}
static void raw_free(void* p) {
    {
        std::unique_lock<std::mutex> lk(g_vmm_mu);
        auto it = g_vmm_blocks.find(p);
        if (it != g_vmm_blocks.end()) {
            vmm_block b = std::move(it->second); g_vmm_blocks.erase(it); lk.unlock();
            vmm_release(p, b, b.handles.size());
            return;
        }
    }
    (void)hipFree(p);
}

// allocations that are only an optimisation (a second set of scan buffers, pre-zeroed row pointers) must neither wait for
// another process's memory nor trim the cache: vg_dev_try_scope makes every allocation of its thread fail fast
static thread_local int t_try_alloc = 0;
vg_dev_try_scope::vg_dev_try_scope() { ++t_try_alloc; }
vg_dev_try_scope::~vg_dev_try_scope() { --t_try_alloc; }

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
    if (e != hipSuccess && t_try_alloc > 0) { (void)hipGetLastError(); throw vg_error(VG_ENOMEM, "device allocation (opportunistic): out of memory"); }
    if (e != hipSuccess) {
        (void)hipGetLastError();                      // the failure is handled here: do not leave it as the sticky "last error"
        if (g_alloc_trace) fprintf(stderr, "[vg alloc] allocation of %.1f MB failed: trimming %.1f GB of cached blocks (live %.1f GB)\n", want / 1048576.0, g_cached_bytes / 1073741824.0, g_live_bytes / 1073741824.0);
        vg_dev_trim();                                // give cached blocks back and retry
        e = raw_alloc(&p, want);
        // (another process may be on its way out -- the CLI returns before the driver has torn its context down -- and its
        // memory comes back within a fraction of a second: wait for it a little before giving up)
        // -- but only when the shortfall can BE another process's: what the device reports in use beyond this process's own
        // blocks must cover it; a set that does not fit beside the caller's own live blocks fails at once)
        for (int tries = 0; e != hipSuccess && tries < 40; ++tries) {
            (void)hipGetLastError();
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || want > tot) break;
            size_t mine = 0; { std::lock_guard<std::mutex> lk(g_alloc_mu); mine = g_live_bytes + g_cached_bytes + g_parked_bytes; }
            const size_t used = tot - fr, foreign = used > mine ? used - mine : 0;
            if (fr >= want || foreign + fr < want) break;              // (enough is free: fragmentation, waiting does not help; or nobody else holds enough)
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
            e = raw_alloc(&p, want);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); throw vg_error(VG_ENOMEM, std::string("device allocation: ") + hipGetErrorString(e)); }
    }
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_block_size[p] = { want, g_device };
    g_live_bytes += want;
    if (g_alloc_trace && want >= (64u << 20)) fprintf(stderr, "[vg alloc] new block %.1f MB at %p (live %.1f GB, cached %.1f GB)\n", want / 1048576.0, p, g_live_bytes / 1073741824.0, g_cached_bytes / 1073741824.0);
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
        g_block_size.erase(it); lk.unlock(); raw_free(p); return;
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
    }
}

// (also the index plan a vg_lz_prepare may have parked -- pools of up to 2 x 24 GiB with kernels queued on them: a caller
// who prepares and then skips the align hands the HBM back here; the plan is process-global, one per process)
// Placement trials (vg_prefilter.hip): the cached blocks -- one placement of a workspace -- are set aside so that the next
// pass allocates afresh beside them; afterwards either the new blocks are kept and the parked ones freed, or the new ones
// are freed and the parked ones come back.
static std::multimap<size_t, void*> g_parked;
void vg_dev_park_cache() {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    for (auto& kv : g_free_blocks) { g_parked.emplace(kv.first, kv.second); g_parked_bytes += kv.first; }
    g_free_blocks.clear(); g_cached_bytes = 0;
}
size_t vg_dev_cached_bytes() { std::lock_guard<std::mutex> lk(g_alloc_mu); return g_cached_bytes; }
void vg_dev_unpark(bool restore_parked) {
    if (restore_parked) vg_dev_trim();                   // the newer placement goes back to the driver
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        for (auto& kv : g_parked) {
            if (restore_parked) { g_free_blocks.emplace(kv.first, kv.second); g_cached_bytes += kv.first; }
            else { drop.push_back(kv.second); g_block_size.erase(kv.second); }
        }
        g_parked.clear(); g_parked_bytes = 0;
    }
    if (!drop.empty()) { (void)hipDeviceSynchronize(); for (void* b : drop) raw_free(b); }
}

extern "C" void vg_release_device_memory(void) {
    // (callable from any thread: the calling thread is put on the library's device first, so that the synchronisation in
    // front of the frees waits for THAT device's queues; no exception crosses the C boundary)
    try { vg_require_device(); vg_lz_drop_prepared(nullptr); vg_dev_unpark(false); vg_dev_trim(); }
    catch (...) { (void)hipGetLastError(); }
}

// allocator self-test: `cycles` times allocate blocks of the given sizes from the library's allocator (whichever path
// VG_ALLOC selects), write a pattern into the first and last MiB of each with a copy from the host, read it back,
// release the blocks and return the cache to the driver.  (tests/test_gpu_parity.py runs it under VG_ALLOC=vmm.)
extern "C" int vg_alloc_selftest(const int64_t* sizes, int n_sizes, int cycles) {
    VG_API_BEGIN
    if (!sizes || n_sizes <= 0 || cycles <= 0) throw vg_error(VG_EINVAL, "vg_alloc_selftest: bad argument");
    vg_require_device();
    hipStream_t s = vg_stream();
    constexpr size_t MB = 1u << 20;
    std::vector<uint32_t> pat(MB / 4), back(MB / 4);
    for (int c = 0; c < cycles; ++c) {
        std::vector<void*> blocks;
        struct guard { std::vector<void*>& b; ~guard() { for (void* p : b) vg_dev_free(p); vg_dev_trim(); } } g{ blocks };
        for (int i = 0; i < n_sizes; ++i) {
            if (sizes[i] < (int64_t)MB) throw vg_error(VG_EINVAL, "vg_alloc_selftest: blocks of at least 1 MiB");
            blocks.push_back(vg_dev_alloc((size_t)sizes[i]));
        }
        for (int i = 0; i < n_sizes; ++i) {
            for (size_t off : { (size_t)0, (size_t)sizes[i] - MB }) {
                for (size_t j = 0; j < pat.size(); ++j) pat[j] = (uint32_t)(j * 2654435761u + (uint32_t)c * 97u + (uint32_t)i * 7919u + (uint32_t)(off >> 20));
                vg_upload_bytes((char*)blocks[(size_t)i] + off, pat.data(), MB, s);
                vg_download_bytes(back.data(), (char*)blocks[(size_t)i] + off, MB, s);
                VG_HIP(hipStreamSynchronize(s));
                if (memcmp(pat.data(), back.data(), MB) != 0) throw vg_error(VG_EHIP, "vg_alloc_selftest: a block does not hold what was written to it");
            }
        }
    }
    VG_API_END
}
extern "C" int vg_copy(void* dst, const void* src, int64_t bytes, int to_host) {
    VG_API_BEGIN
    vg_require_device();
    if (bytes > 0) {
        hipStream_t s = vg_stream();
        if (to_host) vg_download_bytes(dst, src, (size_t)bytes, s); else vg_upload_bytes(dst, src, (size_t)bytes, s);
        VG_HIP(hipStreamSynchronize(s));
    }
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
