// vg_common.h — shared host-side declarations of libvclust_gpu (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <stdint.h>
#include <string>
#include <functional>
#include <vector>
#include <stdexcept>
#include <thread>
#include <exception>
#include "../../include/vclust_gpu.h"

// ---------------------------------------------------------------- errors
void vg_set_error(const char* fmt, ...);
struct vg_error : std::runtime_error {
    int code;
    vg_error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define VG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    throw vg_error(VG_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
// wraps a C-ABI body: exceptions -> error code + vg_last_error()
#define VG_API_BEGIN try {
#define VG_API_END } catch (const vg_error& e) { vg_set_error("%s", e.what()); return e.code; } \
    catch (const std::bad_alloc&) { vg_set_error("out of host memory"); return VG_ENOMEM; } \
    catch (const std::exception& e) { vg_set_error("%s", e.what()); return VG_EINVAL; } \
    return VG_OK;

void vg_require_device();        // throws VG_ENODEV when no HIP device is usable
hipStream_t vg_stream();         // the library's compute stream on the current device
hipStream_t vg_side_stream();                     // a second queue of the same device (created on first use)
void* vg_dev_alloc(size_t bytes); // caching device allocator (throws vg_error)
void  vg_dev_free(void* p);
// while one lives, a failing allocation of this thread throws VG_ENOMEM at once: no trim of the cache, no wait for another
// process's memory (for buffers that are an optimisation only)
struct vg_dev_try_scope { vg_dev_try_scope(); ~vg_dev_try_scope(); };
// copies between a caller's (pageable) buffer and the device on stream s, staged through the library's pinned buffers
// (vg_core.cpp); a download of 32 KiB or more has completed when the call returns
void  vg_upload_bytes(void* dst, const void* src, size_t bytes, hipStream_t s);
void  vg_download_bytes(void* dst, const void* src, size_t bytes, hipStream_t s);
// one empty launch out of the translation unit (loads its code object): for the warm-up thread of the whole-stage calls
void  vg_warm_prefilter(hipStream_t s);
void  vg_warm_align(hipStream_t s);
void  vg_dev_trim();             // return all cached blocks to the driver
// placement trials: set the cached blocks aside (the next allocations are fresh ones beside them); then either restore them
// (the newer cache goes back to the driver) or free them
void  vg_dev_park_cache();
void  vg_dev_unpark(bool restore_parked);
size_t vg_dev_cached_bytes();
// Clean-up work that takes the address-space lock for long (unmapping GBs of FASTA): a whole-stage call parks it
// (vg_defer_mode) until the main thread sits in a long wait for the GPU (vg_deferred_start); otherwise it runs at once
// on a helper thread.
void  vg_defer_mode(bool on);
void  vg_defer(std::function<void()> fn);
void  vg_deferred_start();
// a cold one-shot call (vg_prefilter / vg_align, the CLI): the library keeps its device footprint small for its duration
// (the first use of device memory is what such a process may have to wait for, vg_core.cpp)
void  vg_one_shot_begin(); void vg_one_shot_end(); bool vg_one_shot();
struct vg_one_shot_scope { vg_one_shot_scope() { vg_one_shot_begin(); } ~vg_one_shot_scope() { vg_one_shot_end(); } };

// ---------------------------------------------------------------- device buffers
template <class T> struct dbuf {
    T* p = nullptr; size_t n = 0; bool owned = true;
    dbuf() {}
    explicit dbuf(size_t count) { alloc(count); }
    dbuf(const dbuf&) = delete; dbuf& operator=(const dbuf&) = delete;
    dbuf(dbuf&& o) noexcept : p(o.p), n(o.n), owned(o.owned) { o.p = nullptr; o.n = 0; o.owned = true; }
    dbuf& operator=(dbuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; owned = o.owned; o.p = nullptr; o.n = 0; o.owned = true; } return *this; }
    ~dbuf() { release(); }
    // blocks come from a caching allocator (vg_core.cpp): every kernel and copy of the library is
    // issued on one in-order stream, so a released block can be handed out again without a
    // device synchronisation and the hot path never calls hipMalloc/hipFree.
    void alloc(size_t count) {
        release(); n = count;
        if (count) p = (T*)vg_dev_alloc(count * sizeof(T));
    }
    void release() { if (p && owned) vg_dev_free(p); p = nullptr; n = 0; owned = true; }
    // non-owning window into another buffer (the owner must outlive it)
    void view(T* ptr, size_t count) { release(); p = ptr; n = count; owned = false; }
    size_t bytes() const { return n * sizeof(T); }
    void zero(hipStream_t s) { if (n) VG_HIP(hipMemsetAsync(p, 0, bytes(), s)); }
    void upload(const T* h, size_t count, hipStream_t s) { vg_upload_bytes(p, h, count * sizeof(T), s); }
    void download(T* h, size_t count, hipStream_t s) const { vg_download_bytes(h, p, count * sizeof(T), s); }
};

// ---------------------------------------------------------------- profiling (HIP events on vg_stream)
struct vg_prof_scope {
    const char* name; double bytes; hipEvent_t e0 = nullptr, e1 = nullptr; bool on; hipStream_t stream;
    vg_prof_scope(const char* name, double algorithmic_bytes = 0, hipStream_t on_stream = nullptr);   // nullptr: the library stream
    ~vg_prof_scope();
};
bool vg_profile_on();

// ---------------------------------------------------------------- genome set
// Layout (host and HBM): bases 2-bit packed, 16 per uint32 word, little end first
// (base i of the padded stream sits in word i/16, bits 2*(i%16)..+1); N mask 1 bit per base,
// 32 per word.  Every genome starts at a multiple of VG_ALIGN bases of the padded stream;
// padding bases are A with mask bit 1, and every genome is followed by at least one (a k-mer window that
// leaves its genome therefore always contains a masked base).
constexpr int64_t VG_ALIGN = 64;        // minimum alignment; the set's block size is 1 << align_shift
// std::vector storage without the value-initialising fill (resize() leaves new elements untouched):
// the 2-bit arrays of a big set are zeroed / written by many threads, not by one
template <class T> struct no_init_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = no_init_alloc<U>; };
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};

struct vg_genomes {
    int n = 0;
    int align_shift = 6;                 // genomes start at multiples of (1 << align_shift) bases
    std::vector<uint32_t> blk2g;         // block (1 << align_shift bases) -> genome id
    std::vector<std::string> names;
    std::vector<int64_t> len;        // n
    std::vector<int32_t> n_parts;    // n
    std::vector<int64_t> base_off;   // n+1, padded base offsets
    std::vector<uint8_t> has_n;      // n
    std::vector<uint32_t, no_init_alloc<uint32_t>> packed;    // base_off[n]/16 words (+ slack)
    std::vector<uint32_t, no_init_alloc<uint32_t>> nmask;     // base_off[n]/32 words (+ slack)
    // device residency
    int device = -1;
    dbuf<uint32_t> d_packed, d_nmask;
    mutable dbuf<uint32_t> d_planes;     // the bases once more as bit planes ((lo, hi) word pair per 32 bases): what the LZ parse reads; made on first use (vg_align.hip)
    dbuf<int64_t> d_base_off, d_len;
    dbuf<uint8_t> d_has_n;
    dbuf<uint32_t> d_blk2g;
    int64_t padded_total() const { return base_off.empty() ? 0 : base_off.back(); }
    // L1 order (stable sort by length, descending), computed once: order[rank] = input id, rank[id]
    mutable std::vector<int32_t> len_order, len_rank;
};
void vg_length_order(const vg_genomes* g);        // fills g->len_order / g->len_rank on first use
int  vg_align_tasks_perm(const vg_genomes* g, const vg_pair_count* pairs, int64_t n_pairs, vg_task** tasks, int64_t* n_tasks, uint32_t* perm /* n_pairs entries, or null */);
const uint32_t* vg_genome_planes(const vg_genomes* g, hipStream_t s);   // g->d_planes, made from the resident 2-bit codes on first use (queued on s; vg_align.hip)
void vg_lz_drop_prepared(const vg_genomes* g);    // forget the index plan vg_lz_prepare left for g (nullptr: whatever it left)

// vg_genomes_load with the upload to the library's device overlapped with the packing (vg_genomes.cpp; whole-stage calls)
int vg_genomes_load_resident(const char* const* paths, int n_paths, int multisample, int n_threads, vg_genomes** out);
// RANGE shards held by several ranks (vg_prefilter.hip, k_slice_scan): every rank scans 1/world of the BASES and the kept
// masks + level-1 counts of each rank's digit range travel to it.  The exchange is an all-to-all of device memory:
// for every part, block d of `send` (bytes send_off[d] .. send_off[d + 1]) goes to rank d and block s of `recv` (bytes
// recv_off[s] .. recv_off[s + 1]) arrives from rank s.  `status` is the caller's status so far: the implementation
// agrees on it with all ranks BEFORE anything travels and throws on every rank when one of them failed.
struct vg_xpart { const void* send; const int64_t* send_off; void* recv; const int64_t* recv_off; };
struct vg_slice_exchange {
    int rank = 0, world = 1;
    bool emulate = false;       // developer / tests: no peers -- their slices are scanned by this process (one GPU stands in for `world`)
    std::function<void(int status, const vg_xpart* parts, int n_parts)> alltoallv;
    bool agreed = false;        // the agreement in front of the exchange has taken place (the failure handling of the caller pairs it otherwise)
};
// whether a shard pass of (g, k, fraction) over `world` ranks takes the sliced scan: a pure function of its arguments
void vg_set_spgemm_hook(std::function<void()> fn);      // developer experiment: run once right before the next SpGEMM launch
bool vg_slice_exchange_applies(const vg_genomes* g, int k, double fraction, int world);
int vg_kmer_shard_mode(const vg_genomes* g, double fraction, int n_shards);      // 1 = RANGE shards, 2 = HASH shards (what a rank of an n_shards-way call would do)
// one k-mer range shard of vg_kmer_shared with the (a, b, shared) records left in HBM (vg_prefilter.hip; used by vg_dist.hip)
void vg_kmer_shared_device(vg_genomes* g, int k, double fraction, int shard, int n_shards, uint32_t min_shared,
                           int64_t* set_sizes, dbuf<vg_pair_count>& pairs, int64_t* n_pairs, vg_slice_exchange* xs = nullptr,
                           int* mode_out = nullptr /* how the k-mers were cut: 1 = RANGE shards, 2 = HASH shards */);

// ---------------------------------------------------------------- host threads
// fn(lo, hi, t) over [0, n) cut into contiguous chunks, one per thread (the library's host loops over
// 10^5..10^6 pairs / tasks: a few threads are enough; an exception in a chunk is rethrown on the caller)
int vg_host_threads();
const char* vg_dev_getenv(const char* name);      // a developer switch: read only when VG_DEV_SWITCHES=1 is set
// developer aid: with VG_HOST_TRACE=1 prints the wall time since the previous mark (host-side phases of a call)
void vg_host_mark(const char* what);
double vg_alloc_wait_ms();            // wall time this process has spent inside the driver's device-allocation calls
template <class F> void vg_parallel_chunks(int64_t n, int n_thr, F fn);
// append one genome given codes (0..3, >3 = N); used by the FASTA reader and vg_genomes_from_codes
void vg_genomes_append(vg_genomes* g, const std::string& name, const uint8_t* codes, int64_t len, int n_parts);
void vg_genomes_finish(vg_genomes* g);
// block size for a set of n genomes with total_len bases: ~mean/16, power of two in [64, 4096]
int vg_choose_align_shift(int64_t total_len, int64_t n);

// ---------------------------------------------------------------- host-side helpers shared by writers
int  vg_fmt_num(double x, char* buf);                 // LZ-ANI number format
int  vg_fmt_len_ratio(int64_t a, int64_t b, char* buf);
double vg_ani_shorter(int64_t shared, int64_t na, int64_t nb, int k);

template <class F> void vg_parallel_chunks(int64_t n, int n_thr, F fn) {
    if (n_thr < 1) n_thr = 1;
    if (n < 4096 || n_thr == 1) { fn((int64_t)0, n, 0); return; }
    std::vector<std::thread> th; std::vector<std::exception_ptr> err((size_t)n_thr);
    for (int t = 0; t < n_thr; ++t) th.emplace_back([&, t]() {
        try { fn(n * t / n_thr, n * (t + 1) / n_thr, t); } catch (...) { err[(size_t)t] = std::current_exception(); } });
    for (auto& x : th) x.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
}
