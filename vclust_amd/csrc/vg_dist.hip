// vg_dist.hip — one process per GPU: the sharded forms of the two stages behind the C ABI.
//
// The reference has no distributed layer (single node, threads); north_star shards the path
// over the GPUs of one node with a gather of result rows over RCCL/xGMI.  The path shards without any
// collective inside the kernels:
//   prefilter  rank r handles the k-mers of range r of `world` (RANGE or HASH shards, vg_prefilter.hip) (vg_kmer_shared's
//              shard/n_shards): per-genome set sizes and per-pair shared counts of the shards ADD UP.  What travels
//              is bounded by the RESULT, not by the partial lists: a pair whose total reaches min_shared has at least
//              ceil(min_shared / world) shared k-mers on SOME rank, so (1) every rank nominates the pairs it holds
//              that many of (keys only), (2) the union of the nominations is formed on every rank, (3) every rank
//              reports its count for each pair of the union (0 if it has none) and the counts are summed.  Pairs that
//              share one or two k-mers per range -- the bulk of the partial lists on real data -- never leave their
//              GPU.  Everything stays in HBM: all-gathers are device to device, sort / unique / lookup are kernels.
//   align      the task list is cut into `world` contiguous reference-id ranges with about equal task counts --
//              a pure function of the list, so no communication -- each rank indexes 1/world of the references
//              and the 12-byte rows come back in one all-gather of known sizes (regions: one more, variable).
// The exchange itself is a vg_comm: either callbacks supplied by the host application (MPI, torch.distributed
// over gloo for CPU tests, ...) or the built-in RCCL communicator (vg_comm_rccl_create: ncclAllGather on the
// library's stream; librccl is loaded on first use so that hosts without it still load this library; the entry
// points are typed by <rccl/rccl.h>, so a signature drift is a compile error here, not a crash on the 8-GPU box).
// Failures are agreed on before every exchange (one status word per rank): every compute section between two
// exchanges runs under a guard whose status feeds the next agreement, so a rank that fails -- in its shard, in an
// allocation, in a sort -- makes every rank return the error instead of leaving the others inside a collective.
// VG_DIST_FORCE=1 (tests): a world of one still goes through the exchanges and merges (RCCL accepts nranks = 1).
#include "vg_common.h"
#include <rocprim/rocprim.hpp>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// Build hosts without the RCCL development headers: the few declarations the dlopen'ed entry points are typed by
// (librccl itself is only needed at run time, by vg_comm_rccl_create).  With the headers present they are the check.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
}
#endif
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <vector>
#include <thread>
#include <string>

// ------------------------------------------------------------------ communicator
struct vg_comm {
    int rank = 0, world = 1;
    bool force = false;                                          // VG_DIST_FORCE: exchanges also at world == 1
    vg_allgather_fn allgather = nullptr; void* ctx = nullptr;   // callback form
    // built-in RCCL form
    void* nccl_lib = nullptr; ncclComm_t nccl_comm = nullptr;
    decltype(&ncclAllGather) p_allgather = nullptr;
    decltype(&ncclCommDestroy) p_destroy = nullptr;
    // the all-to-all of the sliced k-mer scan (vg_slice_exchange): grouped point-to-point transfers, one per peer and
    // direction -- on xGMI every pair of GPUs has its own link, so the seven transfers of a rank run side by side
    decltype(&ncclSend) p_send = nullptr; decltype(&ncclRecv) p_recv = nullptr;
    decltype(&ncclGroupStart) p_group_start = nullptr; decltype(&ncclGroupEnd) p_group_end = nullptr;
    decltype(&ncclCommCount) p_count = nullptr;
    decltype(&ncclCommAbort) p_abort = nullptr;
    // HBM staging of host gathers over RCCL: reserved inside guarded sections, so that an allocation failure is
    // agreed on like any other instead of striking between an agreement and its exchange
    mutable dbuf<char> st_send, st_recv;
    bool exchanges() const { return world > 1 || (force && (allgather || p_allgather)); }
};

namespace {
void check(int rc) { if (rc != VG_OK) throw vg_error(rc, vg_last_error()); }
bool env_force() { const char* e = getenv("VG_DIST_FORCE"); return e && *e && *e != '0'; }

void reserve_staging(const vg_comm* c, int64_t bytes) {
    if (!c->p_allgather) return;
    const size_t need = (size_t)std::max<int64_t>(bytes, 64);
    if (c->st_send.n < need) c->st_send.alloc(need + need / 4);
    if (c->st_recv.n < need * c->world) c->st_recv.alloc((need + need / 4) * c->world);
}
// all ranks contribute `bytes` bytes of HOST memory; recv = world * bytes in rank order
void gather_host(const vg_comm* c, const void* send, void* recv, int64_t bytes) {
    if (!c->exchanges()) { memcpy(recv, send, (size_t)bytes); return; }
    if (c->allgather) {
        if (c->allgather(c->ctx, send, recv, bytes, 0) != 0) throw vg_error(VG_EIO, "vg_comm: allgather callback failed");
        return;
    }
    // RCCL moves device memory: stage through HBM
    hipStream_t s = vg_stream();
    reserve_staging(c, bytes);
    vg_prof_scope ps("exchange", (double)bytes * c->world);       // (HIP events on the library stream: bench.py prints it per rank)
    vg_upload_bytes(c->st_send.p, send, (size_t)bytes, s);
    if (c->p_allgather(c->st_send.p, c->st_recv.p, (size_t)bytes, ncclChar, c->nccl_comm, s) != ncclSuccess) throw vg_error(VG_EIO, "ncclAllGather failed");
    vg_download_bytes(recv, c->st_recv.p, (size_t)bytes * c->world, s);
    VG_HIP(hipStreamSynchronize(s));
}
// the same for DEVICE memory (buffers of the current device)
void gather_device(const vg_comm* c, const void* send, void* recv, int64_t bytes) {
    hipStream_t s = vg_stream();
    if (!c->exchanges()) { VG_HIP(hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, s)); return; }
    if (c->allgather) {
        VG_HIP(hipStreamSynchronize(s));                          // the callback runs on the application's own stream
        if (c->allgather(c->ctx, send, recv, bytes, 1) != 0) throw vg_error(VG_EIO, "vg_comm: allgather callback failed");
        return;
    }
    vg_prof_scope ps("exchange", (double)bytes * c->world);
    if (c->p_allgather(send, recv, (size_t)bytes, ncclChar, c->nccl_comm, s) != ncclSuccess) throw vg_error(VG_EIO, "ncclAllGather failed");
}
// every rank learns whether any rank failed: throws the first failure on all of them
void agree(const vg_comm* c, int my_rc, const char* what) {
    std::vector<int32_t> all((size_t)c->world, 0); int32_t mine = my_rc;
    gather_host(c, &mine, all.data(), sizeof(int32_t));
    // (the first failed rank names the error code; a rank that failed itself -- whether or not it is that first one -- says why)
    for (int r = 0; r < c->world; ++r) if (all[(size_t)r] != 0)
        throw vg_error(all[(size_t)r], std::string(what) + ": rank " + std::to_string(r) + " failed" + (my_rc != 0 ? std::string(": ") + vg_last_error() : std::string()));
}
// the same agreement carrying a checksum every rank must hold alike (e.g. of an input list all ranks are to pass identically)
void agree_same(const vg_comm* c, int my_rc, uint32_t checksum, const char* what, const char* mismatch) {
    std::vector<uint32_t> all((size_t)c->world * 2, 0); const uint32_t mine[2] = { (uint32_t)my_rc, checksum };
    gather_host(c, mine, all.data(), sizeof mine);
    for (int r = 0; r < c->world; ++r) if ((int32_t)all[(size_t)2 * r] != 0)
        throw vg_error((int32_t)all[(size_t)2 * r], std::string(what) + ": rank " + std::to_string(r) + " failed" + (r == c->rank ? std::string(": ") + vg_last_error() : std::string()));
    for (int r = 1; r < c->world; ++r) if (all[(size_t)2 * r + 1] != all[1])
        throw vg_error(VG_EINVAL, std::string(what) + ": rank " + std::to_string(r) + " and rank 0 " + mismatch);
}
// the agreement in front of an all-to-all: status and plan word as in agree_same, plus every rank's block sizes in both
// directions -- each rank holds the whole (sender, receiver) matrix afterwards and all reach the same verdict: a layout two
// ranks computed differently (word_of / st_of from different tile counts) stops every rank HERE instead of hanging or
// writing past a block inside grouped ncclSend / ncclRecv
void agree_exchange(const vg_comm* c, int my_rc, uint32_t plan, const vg_xpart* parts, int n_parts, const char* what, const char* mismatch) {
    // (the message has ONE size whatever the rank has to say -- a rank that failed before it knew its blocks, or that does
    // not slice at all, pairs the collective with n_parts = 0)
    constexpr int MAX_XPARTS = 4;
    if (n_parts > MAX_XPARTS) throw vg_error(VG_EINVAL, "internal error: more exchange parts than the agreement carries");
    const int W = c->world;
    const size_t nw = 3 + (size_t)MAX_XPARTS * 2 * (size_t)W;
    std::vector<int64_t> mine(nw, 0), all(nw * (size_t)W, 0);
    mine[0] = my_rc; mine[1] = plan; mine[2] = n_parts;
    for (int i = 0; i < n_parts; ++i) for (int r = 0; r < W; ++r) {
        mine[3 + ((size_t)i * 2 + 0) * W + r] = parts[i].send_off[r + 1] - parts[i].send_off[r];
        mine[3 + ((size_t)i * 2 + 1) * W + r] = parts[i].recv_off[r + 1] - parts[i].recv_off[r];
    }
    gather_host(c, mine.data(), all.data(), (int64_t)(nw * sizeof(int64_t)));
    for (int r = 0; r < W; ++r) if (all[(size_t)r * nw] != 0)
        throw vg_error((int)all[(size_t)r * nw], std::string(what) + ": rank " + std::to_string(r) + " failed" + (r == c->rank ? std::string(": ") + vg_last_error() : std::string()));
    for (int r = 1; r < W; ++r) if (all[(size_t)r * nw + 1] != all[1] || all[(size_t)r * nw + 2] != all[2])
        throw vg_error(VG_EINVAL, std::string(what) + ": rank " + std::to_string(r) + " and rank 0 " + mismatch);
    for (int i = 0; i < n_parts; ++i) for (int a = 0; a < W; ++a) for (int b = 0; b < W; ++b)
        if (all[(size_t)a * nw + 3 + ((size_t)i * 2 + 0) * W + b] != all[(size_t)b * nw + 3 + ((size_t)i * 2 + 1) * W + a])
            throw vg_error(VG_EINVAL, std::string(what) + ": rank " + std::to_string(a) + " sends a block of part " + std::to_string(i) + " to rank " + std::to_string(b) +
                           " whose size the receiver computed differently (the ranks disagree on the exchange layout)");
}
// a compute section between two exchanges: its failure becomes this rank's status word of the next agreement
template <class F> void guarded(const vg_comm* c, const char* what, F fn) {
    int rc = VG_OK;
    try { fn(); }
    catch (const vg_error& e) { vg_set_error("%s", e.what()); rc = e.code; }
    catch (const std::bad_alloc&) { vg_set_error("out of host memory"); rc = VG_ENOMEM; }
    catch (const std::exception& e) { vg_set_error("%s", e.what()); rc = VG_EINVAL; }
    agree(c, rc, what);
}

// all-to-all of DEVICE memory (vg_xpart: block d of send -> rank d, block s of recv <- rank s), after the agreement on
// `status`.  RCCL: one group of ncclSend / ncclRecv per peer; a callback communicator has only its all-gather, so every
// rank contributes its whole send buffer (padded to the largest) and picks its blocks -- world x the traffic, which is
// what the CPU / one-GPU tests of the protocol can afford and nothing else uses.
void alltoallv_device(const vg_comm* c, const vg_xpart* parts, int n_parts, bool self_through_rccl_always = false) {
    hipStream_t s = vg_stream();
    const int W = c->world, me = c->rank;
    if (c->p_allgather) {
        if (!c->p_send || !c->p_recv || !c->p_group_start || !c->p_group_end) throw vg_error(VG_EIO, "librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
        double bytes = 0;
        for (int i = 0; i < n_parts; ++i) bytes += (double)(parts[i].recv_off[W] - parts[i].recv_off[0]);
        vg_prof_scope ps("exchange", bytes);
        const bool self_through_rccl = self_through_rccl_always || (c->force && W == 1);      // (the self-test, VG_DIST_FORCE with one rank: the point-to-point entry points are exercised on a one-GPU box)
        if (c->p_group_start() != ncclSuccess) throw vg_error(VG_EIO, "ncclGroupStart failed");
        bool ok = true;
        for (int i = 0; i < n_parts && ok; ++i) for (int r = 0; r < W && ok; ++r) {
            const vg_xpart& x = parts[i];
            const int64_t sb = x.send_off[r + 1] - x.send_off[r], rb = x.recv_off[r + 1] - x.recv_off[r];
            if (r == me && !self_through_rccl) continue;
            if (sb > 0) ok = c->p_send((const char*)x.send + x.send_off[r], (size_t)sb, ncclChar, r, c->nccl_comm, s) == ncclSuccess;
            if (ok && rb > 0) ok = c->p_recv((char*)x.recv + x.recv_off[r], (size_t)rb, ncclChar, r, c->nccl_comm, s) == ncclSuccess;
        }
        const bool ended = c->p_group_end() == ncclSuccess;
        if (!ok || !ended) {
            // (the peers are inside their group call, behind an agreement that said everybody was fine: aborting the
            // communicator is what gets them out with an error instead of a hang; every later collective of it fails)
            if (c->p_abort && c->nccl_comm) (void)c->p_abort(c->nccl_comm);
            throw vg_error(VG_EIO, "ncclSend / ncclRecv failed after the agreement: communicator aborted");
        }
        if (!self_through_rccl) for (int i = 0; i < n_parts; ++i) {
            const vg_xpart& x = parts[i];
            const int64_t sb = x.send_off[me + 1] - x.send_off[me];
            if (sb != x.recv_off[me + 1] - x.recv_off[me]) throw vg_error(VG_EINVAL, "vg_comm: a rank's own block has two sizes");
            if (sb > 0) VG_HIP(hipMemcpyAsync((char*)x.recv + x.recv_off[me], (const char*)x.send + x.send_off[me], (size_t)sb, hipMemcpyDeviceToDevice, s));
        }
        return;
    }
    for (int i = 0; i < n_parts; ++i) {
        const vg_xpart& x = parts[i];
        // every rank's block layout, then every rank's send buffer
        std::vector<int64_t> all_off((size_t)(W + 1) * W);
        gather_host(c, x.send_off, all_off.data(), (int64_t)sizeof(int64_t) * (W + 1));
        int64_t pad = 64; for (int r = 0; r < W; ++r) pad = std::max(pad, all_off[(size_t)r * (W + 1) + W] - all_off[(size_t)r * (W + 1)]);
        dbuf<char> d_send((size_t)pad), d_all((size_t)pad * W);
        const int64_t mine = x.send_off[W] - x.send_off[0];
        if (mine > 0) VG_HIP(hipMemcpyAsync(d_send.p, (const char*)x.send + x.send_off[0], (size_t)mine, hipMemcpyDeviceToDevice, s));
        gather_device(c, d_send.p, d_all.p, pad);
        for (int r = 0; r < W; ++r) {
            const int64_t* off = &all_off[(size_t)r * (W + 1)];
            const int64_t b = off[me + 1] - off[me];
            if (b != x.recv_off[r + 1] - x.recv_off[r]) throw vg_error(VG_EINVAL, "vg_comm: sender and receiver disagree on a block size");
            if (b > 0) VG_HIP(hipMemcpyAsync((char*)x.recv + x.recv_off[r], d_all.p + (size_t)r * pad + (off[me] - off[0]), (size_t)b, hipMemcpyDeviceToDevice, s));
        }
        VG_HIP(hipStreamSynchronize(s));                              // the staging buffers go out of scope
    }
}

inline int dgrid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }
// output slot of a kept element behind a global cursor, ONE atomic per wave (a global atomic per element on one word
// costs ~10 ns each: 4.5 ms per 450 000 records); to be called by all active lanes of the wave
__device__ __forceinline__ unsigned long long wave_slot(unsigned long long* cursor, bool keep) {
    const unsigned long long b = __ballot(keep);
    if (!b) return 0;
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(b);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(cursor, (unsigned long long)__popcll(b));
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)base, leader), hi = (uint32_t)__shfl((int)(uint32_t)(base >> 32), leader);
    return (((unsigned long long)hi << 32) | lo) + (unsigned long long)__popcll(b & ((1ULL << lane) - 1ULL));
}
// keys of the pairs this rank nominates (count >= thr), compacted through a cursor (their order is irrelevant: they are sorted next)
__global__ void k_nominate(const vg_pair_count* __restrict__ rec, int64_t n, uint32_t thr, uint64_t* __restrict__ keys, unsigned long long* __restrict__ cursor) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const vg_pair_count r = rec[i];
        const bool keep = r.shared >= thr;
        const unsigned long long o = wave_slot(cursor, keep);
        if (keep) keys[o] = ((uint64_t)r.a << 32) | r.b;
    }
}
__global__ void k_local_keys(const vg_pair_count* __restrict__ rec, int64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const vg_pair_count r = rec[i]; keys[i] = ((uint64_t)r.a << 32) | r.b; vals[i] = r.shared;
    }
}
// gathered layout: world blocks of `pad` keys, the first cnt[r] of block r are real -> one dense array
__global__ void k_squeeze_keys(const uint64_t* __restrict__ all, const int64_t* __restrict__ prefix, int world, int64_t pad, uint64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)world * pad; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / pad); const int64_t j = i - (int64_t)r * pad;
        if (j < prefix[r + 1] - prefix[r]) out[prefix[r] + j] = all[i];
    }
}
// this rank's count of every pair of the union (binary search in its sorted keys)
__global__ void k_lookup_counts(const uint64_t* __restrict__ uni, int64_t nu, const uint64_t* __restrict__ lk, const uint32_t* __restrict__ lv, int64_t nl,
                                uint32_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = uni[i];
        int64_t lo = 0, hi = nl;
        while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (lk[m] < key) lo = m + 1; else hi = m; }
        out[i] = (lo < nl && lk[lo] == key) ? lv[lo] : 0u;
    }
}
__global__ void k_sum_counts(const uint64_t* __restrict__ uni, int64_t nu, const uint32_t* __restrict__ cnt /* world x nu */, int world, uint32_t min_shared,
                             vg_pair_count* __restrict__ out, unsigned long long* __restrict__ cursor) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t t = 0;
        for (int r = 0; r < world; ++r) t += cnt[(int64_t)r * nu + i];
        const bool keep = t >= min_shared;
        const unsigned long long o = wave_slot(cursor, keep);
        if (keep) { out[o].a = (uint32_t)(uni[i] >> 32); out[o].b = (uint32_t)uni[i]; out[o].shared = (uint32_t)t; }
    }
}
}  // namespace

extern "C" int vg_comm_create(int rank, int world, vg_allgather_fn allgather, void* ctx, vg_comm** out) {
    VG_API_BEGIN
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !allgather)) throw vg_error(VG_EINVAL, "vg_comm_create: bad arguments");
    vg_comm* c = new vg_comm; c->rank = rank; c->world = world; c->allgather = allgather; c->ctx = ctx; c->force = env_force();
    *out = c;
    VG_API_END
}

static void* load_rccl() {
    for (const char* nm : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) return h; }
    throw vg_error(VG_EIO, std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found"));
}

extern "C" int vg_rccl_unique_id(void* out, int64_t bytes) {
    VG_API_BEGIN
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the RCCL unique id around as 128 bytes");
    if (!out || bytes < (int64_t)sizeof(ncclUniqueId)) throw vg_error(VG_EINVAL, "vg_rccl_unique_id: needs a 128-byte buffer");
    void* h = load_rccl();
    auto get = (decltype(&ncclGetUniqueId))dlsym(h, "ncclGetUniqueId");
    if (!get) throw vg_error(VG_EIO, "librccl has no ncclGetUniqueId");
    ncclUniqueId id;
    if (get(&id) != ncclSuccess) throw vg_error(VG_EIO, "ncclGetUniqueId failed");
    memcpy(out, &id, sizeof id);
    VG_API_END
}

extern "C" int vg_comm_rccl_create(int rank, int world, const void* unique_id, int64_t id_bytes, vg_comm** out) {
    VG_API_BEGIN
    if (!out || !unique_id || id_bytes < (int64_t)sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) throw vg_error(VG_EINVAL, "vg_comm_rccl_create: bad arguments");
    vg_require_device();
    void* h = load_rccl();
    ncclUniqueId id; memcpy(&id, unique_id, sizeof id);
    auto init = (decltype(&ncclCommInitRank))dlsym(h, "ncclCommInitRank");
    auto ag = (decltype(&ncclAllGather))dlsym(h, "ncclAllGather");
    auto destroy = (decltype(&ncclCommDestroy))dlsym(h, "ncclCommDestroy");
    if (!init || !ag || !destroy) throw vg_error(VG_EIO, "librccl lacks ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    ncclComm_t comm = nullptr;
    if (init(&comm, world, id, rank) != ncclSuccess) throw vg_error(VG_EIO, "ncclCommInitRank failed");
    vg_comm* c = new vg_comm; c->rank = rank; c->world = world; c->nccl_lib = h; c->nccl_comm = comm; c->p_allgather = ag; c->p_destroy = destroy; c->force = env_force();
    c->p_send = (decltype(&ncclSend))dlsym(h, "ncclSend"); c->p_recv = (decltype(&ncclRecv))dlsym(h, "ncclRecv");
    c->p_group_start = (decltype(&ncclGroupStart))dlsym(h, "ncclGroupStart"); c->p_group_end = (decltype(&ncclGroupEnd))dlsym(h, "ncclGroupEnd");
    c->p_count = (decltype(&ncclCommCount))dlsym(h, "ncclCommCount");
    c->p_abort = (decltype(&ncclCommAbort))dlsym(h, "ncclCommAbort");
    reserve_staging(c, 1 << 16);                    // status words and counts never allocate
    *out = c;
    VG_API_END
}

extern "C" void vg_comm_free(vg_comm* c) {
    if (!c) return;
    if (c->nccl_comm && c->p_destroy) { (void)hipDeviceSynchronize(); (void)c->p_destroy(c->nccl_comm); }
    c->st_send.release(); c->st_recv.release();
    delete c;
}
// 1 = the built-in RCCL communicator, 0 = a callback communicator; the rank count RCCL itself reports for it (ncclCommCount;
// -1: not an RCCL communicator, or the query failed): what a scaling record needs to say which exchange it measured
extern "C" int vg_comm_kind(const vg_comm* c) { return c && c->p_allgather ? 1 : 0; }
extern "C" int vg_comm_rccl_ranks(const vg_comm* c) {
    int n = -1;
    if (!c || !c->nccl_comm || !c->p_count || c->p_count(c->nccl_comm, &n) != ncclSuccess) return -1;
    return n;
}
extern "C" int vg_comm_rank(const vg_comm* c) { return c ? c->rank : 0; }
extern "C" int vg_comm_world(const vg_comm* c) { return c ? c->world : 1; }

// exchange self-test (no kernels: usable without a GPU for the callback form with host memory): every rank
// contributes a rank-dependent pattern and checks what it receives
extern "C" int vg_comm_selftest(const vg_comm* c, int64_t bytes) {
    VG_API_BEGIN
    if (!c || bytes < 1) throw vg_error(VG_EINVAL, "vg_comm_selftest: bad arguments");
    std::vector<uint8_t> send((size_t)bytes), recv((size_t)bytes * c->world);
    for (int64_t i = 0; i < bytes; ++i) send[(size_t)i] = (uint8_t)(i * 131 + c->rank * 17 + 3);
    gather_host(c, send.data(), recv.data(), bytes);
    for (int r = 0; r < c->world; ++r) for (int64_t i = 0; i < bytes; ++i)
        if (recv[(size_t)(r * bytes + i)] != (uint8_t)(i * 131 + r * 17 + 3)) throw vg_error(VG_EIO, "vg_comm_selftest: wrong data from rank " + std::to_string(r));
    agree(c, 0, "vg_comm_selftest");
    if (c->p_allgather) {
        // the built-in communicator also carries the all-to-all of the sliced k-mer scan (grouped ncclSend / ncclRecv):
        // blocks of unequal sizes between every pair of ranks, a rank's own block through RCCL too, checked on arrival
        const int W = c->world;
        auto blk = [&](int from, int to) { return (int64_t)(1000 + 37 * from + 11 * to); };
        std::vector<int64_t> soff((size_t)W + 1, 0), roff((size_t)W + 1, 0);
        for (int r = 0; r < W; ++r) { soff[(size_t)r + 1] = soff[(size_t)r] + blk(c->rank, r); roff[(size_t)r + 1] = roff[(size_t)r] + blk(r, c->rank); }
        std::vector<uint8_t> hs((size_t)soff[(size_t)W]), hr((size_t)roff[(size_t)W]);
        for (int r = 0; r < W; ++r) for (int64_t i = 0; i < blk(c->rank, r); ++i) hs[(size_t)(soff[(size_t)r] + i)] = (uint8_t)(i * 7 + c->rank * 29 + r * 3 + 1);
        int rc = VG_OK;
        try {
            dbuf<uint8_t> ds(hs.size()), dr(hr.size());
            hipStream_t s = vg_stream();
            ds.upload(hs.data(), hs.size(), s);
            const vg_xpart part{ ds.p, soff.data(), dr.p, roff.data() };
            alltoallv_device(c, &part, 1, true);
            dr.download(hr.data(), hr.size(), s); VG_HIP(hipStreamSynchronize(s));
            for (int r = 0; r < W; ++r) for (int64_t i = 0; i < blk(r, c->rank); ++i)
                if (hr[(size_t)(roff[(size_t)r] + i)] != (uint8_t)(i * 7 + r * 29 + c->rank * 3 + 1)) throw vg_error(VG_EIO, "vg_comm_selftest: wrong all-to-all block from rank " + std::to_string(r));
        } catch (const vg_error& e) { vg_set_error("%s", e.what()); rc = e.code; }
        agree(c, rc, "vg_comm_selftest all-to-all");
    }
    VG_API_END
}

// owner rank of every task (reference-range partition): references are cut into `world` contiguous id ranges
// holding about the same number of tasks; a pure function of the task list
extern "C" int vg_align_owner(const vg_task* tasks, int64_t n_tasks, int n_genomes, int world, int32_t* owner) {
    VG_API_BEGIN
    if ((!tasks && n_tasks) || (!owner && n_tasks) || world < 1 || n_genomes < 0) throw vg_error(VG_EINVAL, "vg_align_owner: bad arguments");
    // (every rank of a sharded run does this for the whole list: a few threads, per-thread counters)
    const int n_thr = n_tasks >= (1 << 18) ? std::min(vg_host_threads(), 8) : 1;
    std::vector<std::vector<int64_t>> part((size_t)n_thr, std::vector<int64_t>((size_t)n_genomes + 1, 0));
    vg_parallel_chunks(n_tasks, n_thr, [&](int64_t lo, int64_t hi, int t) {
        auto& pr = part[(size_t)t];
        for (int64_t i = lo; i < hi; ++i) { if (tasks[i].r >= (uint32_t)n_genomes) throw vg_error(VG_EINVAL, "task id out of range"); pr[tasks[i].r]++; }
    });
    std::vector<int32_t> own_ref((size_t)n_genomes + 1, 0);
    int64_t before = 0;
    for (int r = 0; r < n_genomes; ++r) {
        own_ref[(size_t)r] = n_tasks ? (int32_t)std::min<int64_t>(world - 1, before * world / n_tasks) : 0;
        for (int t = 0; t < n_thr; ++t) before += part[(size_t)t][(size_t)r];
    }
    vg_parallel_chunks(n_tasks, n_thr, [&](int64_t lo, int64_t hi, int) { for (int64_t i = lo; i < hi; ++i) owner[i] = own_ref[tasks[i].r]; });
    VG_API_END
}

// ------------------------------------------------------------------ prefilter, sharded
extern "C" int vg_kmer_shared_sharded(vg_genomes* g, int k, double fraction, uint32_t min_shared, const vg_comm* c,
                                      int64_t* set_sizes, vg_pair_count** pairs, int64_t* n_pairs) {
    VG_API_BEGIN
    if (!g || !c || !set_sizes || !pairs || !n_pairs) throw vg_error(VG_EINVAL, "vg_kmer_shared_sharded: null argument");
    *pairs = nullptr; *n_pairs = 0;
    const int n = vg_genomes_count(g);
    if (!c->exchanges()) { check(vg_kmer_shared(g, k, fraction, 0, 1, min_shared, set_sizes, pairs, n_pairs)); return VG_OK; }
    const int W = c->world;
    if (min_shared < 1) min_shared = 1;
    hipStream_t s = vg_stream();
    // ---- this rank's shard of the k-mer range: partial sizes (host) and partial counts (left in HBM)
    std::vector<int64_t> part_sizes((size_t)std::max(n, 1) + 1, 0);           // (+ 1: how this rank cut the k-mers, compared below)
    int my_mode = 0;
    dbuf<vg_pair_count> d_loc; int64_t n_loc = 0;
    dbuf<uint64_t> d_nom, lk, lk2; dbuf<uint32_t> lv, lv2; dbuf<unsigned long long> d_cur(1);
    unsigned long long n_nom = 0;
    const uint32_t thr = (min_shared + (uint32_t)W - 1) / (uint32_t)W;        // pigeonhole: some rank holds >= ceil(T / W) of a pair that reaches T
    // RANGE shards: every rank scans 1/W of the bases, kept masks and level-1 counts travel (k_slice_scan); the agreement
    // in front of that exchange is paired below by a rank that never reaches it
    vg_slice_exchange xs; xs.rank = c->rank; xs.world = W;
    const bool sliced = W > 1 && vg_slice_exchange_applies(g, k, fraction, W);
    // EVERY rank of a world > 1 goes through ONE "prefilter scan" agreement in front of its shard's exchanges, sliced or
    // not, and the agreement carries how the rank is about to work: (sliced scan?, RANGE / HASH cut).  Both are decided from
    // process-local knobs (vg_set_subshards, VG_RANGE_SCAN, VG_INDEX_PATH); a rank that decided differently would otherwise
    // skip or add a collective and leave its peers inside grouped ncclSend / ncclRecv.  A mismatch throws on every rank.
    const uint32_t my_plan = W > 1 ? (uint32_t)(sliced ? 1 : 0) | ((uint32_t)vg_kmer_shard_mode(g, fraction, W) << 1) : 0u;
    const char* plan_mismatch = "plan the prefilter shard differently (sliced scan / RANGE against HASH shards: do the ranks differ in vg_set_subshards, VG_RANGE_SCAN or VG_INDEX_PATH?)";
    xs.alltoallv = [&](int status, const vg_xpart* parts, int n_parts) {
        xs.agreed = true;
        agree_exchange(c, status, my_plan, parts, n_parts, "prefilter scan", plan_mismatch);
        alltoallv_device(c, parts, n_parts);
    };
    guarded(c, "prefilter shard", [&] {
        // (a rank that does not slice pairs the agreement here, in front of its pass)
        if (W > 1 && !sliced) { xs.agreed = true; agree_exchange(c, VG_OK, my_plan, nullptr, 0, "prefilter scan", plan_mismatch); }
        try { vg_kmer_shared_device(g, k, fraction, c->rank, W, 1u, part_sizes.data(), d_loc, &n_loc, sliced ? &xs : nullptr, &my_mode); }
        catch (...) {
            if (W > 1 && !xs.agreed) { xs.agreed = true; try { agree_exchange(c, VG_EINVAL, my_plan, nullptr, 0, "prefilter scan", plan_mismatch); } catch (...) {} }
            throw;
        }
        if (W > 1 && !xs.agreed) { xs.agreed = true; agree_exchange(c, VG_OK, my_plan, nullptr, 0, "prefilter scan", plan_mismatch); }
        // nominations, and this rank's records sorted by key for the lookups of step 3
        d_nom.alloc((size_t)std::max<int64_t>(n_loc, 1)); d_cur.zero(s);
        lk.alloc((size_t)std::max<int64_t>(n_loc, 1)); lk2.alloc(lk.n); lv.alloc(lk.n); lv2.alloc(lk.n);
        if (n_loc) {
            hipLaunchKernelGGL(k_nominate, dim3(dgrid(n_loc)), dim3(256), 0, s, (const vg_pair_count*)d_loc.p, n_loc, thr, d_nom.p, d_cur.p);
            hipLaunchKernelGGL(k_local_keys, dim3(dgrid(n_loc)), dim3(256), 0, s, (const vg_pair_count*)d_loc.p, n_loc, lk.p, lv.p);
            size_t tb = 0;
            VG_HIP(rocprim::radix_sort_pairs(nullptr, tb, lk.p, lk2.p, lv.p, lv2.p, (size_t)n_loc, 0u, 64u, s));
            dbuf<char> tmp(tb);
            VG_HIP(rocprim::radix_sort_pairs((void*)tmp.p, tb, lk.p, lk2.p, lv.p, lv2.p, (size_t)n_loc, 0u, 64u, s));
        }
        d_cur.download(&n_nom, 1, s);
        VG_HIP(hipStreamSynchronize(s));
        d_loc.release();
        reserve_staging(c, (int64_t)sizeof(int64_t) * (std::max(n, 1) + 1));
    });
    // ---- set sizes add up (n words per rank); the extra word says how the rank cut its k-mers: RANGE and HASH shards do
    // not tile the key space together, so a run whose ranks disagree (a per-process knob differs) stops here, on every rank
    {
        const size_t nw = (size_t)std::max(n, 1) + 1;
        part_sizes[nw - 1] = my_mode;
        std::vector<int64_t> all_sizes(nw * W, 0);
        gather_host(c, part_sizes.data(), all_sizes.data(), (int64_t)sizeof(int64_t) * (int64_t)nw);
        for (int r = 0; r < W; ++r) if (all_sizes[(size_t)r * nw + nw - 1] != all_sizes[nw - 1])
            throw vg_error(VG_EINVAL, "vg_kmer_shared_sharded: rank " + std::to_string(r) + " cut its k-mers differently from rank 0 (RANGE against HASH shards: do the ranks differ in vg_set_subshards?)");
        for (int i = 0; i < n; ++i) { int64_t t = 0; for (int r = 0; r < W; ++r) t += all_sizes[(size_t)r * nw + i]; set_sizes[i] = t; }
    }
    // ---- 1: nominations travel (keys only), 2: their union
    std::vector<int64_t> cnt((size_t)W, 0), prefix((size_t)W + 1, 0);
    const int64_t my_nom = (int64_t)n_nom;
    gather_host(c, &my_nom, cnt.data(), sizeof(int64_t));
    int64_t pad = 1; for (int r = 0; r < W; ++r) { pad = std::max(pad, cnt[(size_t)r]); prefix[(size_t)r + 1] = prefix[(size_t)r] + cnt[(size_t)r]; }
    const int64_t total = prefix[(size_t)W];
    dbuf<uint64_t> d_send, d_all, d_uni; dbuf<uint32_t> d_cnt, d_cnt_all; unsigned long long nu = 0;
    guarded(c, "prefilter nominations", [&] {
        d_send.alloc((size_t)pad); d_all.alloc((size_t)pad * W); d_send.zero(s);
        if (my_nom) VG_HIP(hipMemcpyAsync(d_send.p, d_nom.p, sizeof(uint64_t) * (size_t)my_nom, hipMemcpyDeviceToDevice, s));
    });
    gather_device(c, d_send.p, d_all.p, pad * (int64_t)sizeof(uint64_t));
    guarded(c, "prefilter union", [&] {
        const size_t nt = (size_t)std::max<int64_t>(total, 1);
        dbuf<uint64_t> keys(nt), keys2(nt); dbuf<int64_t> d_prefix((size_t)W + 1); dbuf<unsigned long long> d_nu(1);
        d_prefix.upload(prefix.data(), prefix.size(), s);
        d_uni.alloc(nt);
        if (total) {
            hipLaunchKernelGGL(k_squeeze_keys, dim3(dgrid(pad * W)), dim3(256), 0, s, (const uint64_t*)d_all.p, (const int64_t*)d_prefix.p, W, pad, keys.p);
            size_t tb = 0, tb2 = 0;
            VG_HIP(rocprim::radix_sort_keys(nullptr, tb, keys.p, keys2.p, (size_t)total, 0u, 64u, s));
            VG_HIP(rocprim::unique(nullptr, tb2, keys2.p, d_uni.p, d_nu.p, (size_t)total, rocprim::equal_to<uint64_t>(), s));
            dbuf<char> tmp(std::max(tb, tb2));
            VG_HIP(rocprim::radix_sort_keys((void*)tmp.p, tb, keys.p, keys2.p, (size_t)total, 0u, 64u, s));
            VG_HIP(rocprim::unique((void*)tmp.p, tb2, keys2.p, d_uni.p, d_nu.p, (size_t)total, rocprim::equal_to<uint64_t>(), s));
            d_nu.download(&nu, 1, s); VG_HIP(hipStreamSynchronize(s));
        }
        // ---- 3: this rank's count of every pair of the union (the union is the same on every rank)
        d_cnt.alloc((size_t)std::max<unsigned long long>(nu, 1)); d_cnt_all.alloc(d_cnt.n * W);
        if (nu) hipLaunchKernelGGL(k_lookup_counts, dim3(dgrid((int64_t)nu)), dim3(256), 0, s, (const uint64_t*)d_uni.p, (int64_t)nu, (const uint64_t*)lk2.p,
                                   (const uint32_t*)lv2.p, n_loc, d_cnt.p);
    });
    if (nu) gather_device(c, d_cnt.p, d_cnt_all.p, (int64_t)nu * (int64_t)sizeof(uint32_t));
    vg_pair_count* out = nullptr; unsigned long long n_out = 0;
    guarded(c, "prefilter sums", [&] {
        dbuf<vg_pair_count> d_out((size_t)std::max<unsigned long long>(nu, 1));
        d_cur.zero(s);
        if (nu) hipLaunchKernelGGL(k_sum_counts, dim3(dgrid((int64_t)nu)), dim3(256), 0, s, (const uint64_t*)d_uni.p, (int64_t)nu, (const uint32_t*)d_cnt_all.p, W, min_shared, d_out.p, d_cur.p);
        d_cur.download(&n_out, 1, s); VG_HIP(hipStreamSynchronize(s));
        out = (vg_pair_count*)malloc(sizeof(vg_pair_count) * std::max<size_t>(1, (size_t)n_out));
        if (!out) throw vg_error(VG_ENOMEM, "out of host memory");
        if (n_out) { d_out.download(out, (size_t)n_out, s); VG_HIP(hipStreamSynchronize(s)); }
        // the cursor order depends on scheduling; every rank returns the same list
        std::sort(out, out + n_out, [](const vg_pair_count& x, const vg_pair_count& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
    });
    *pairs = out; *n_pairs = (int64_t)n_out;
    VG_API_END
}

// ------------------------------------------------------------------ align, sharded
extern "C" int vg_lz_align_sharded(vg_genomes* g, const vg_task* tasks, int64_t n_tasks, const vg_lz_params* p, const vg_comm* c,
                                   vg_pair_stat* stats, vg_region** regions, int64_t* n_regions) {
    VG_API_BEGIN
    if (!g || !c || (!tasks && n_tasks) || !p || (!stats && n_tasks)) throw vg_error(VG_EINVAL, "vg_lz_align_sharded: null argument");
    if (!c->exchanges()) { check(vg_lz_align(g, tasks, n_tasks, p, stats, regions, n_regions)); return VG_OK; }
    if (regions) { *regions = nullptr; if (n_regions) *n_regions = 0; }
    const int W = c->world;
    std::vector<int32_t> owner; std::vector<int64_t> mine, per_rank((size_t)W, 0);
    std::vector<vg_pair_stat> my_stats, send, all;
    vg_region* my_reg = nullptr; int64_t my_nreg = 0, pad = 1;
    struct guard { vg_region** q; ~guard() { if (*q) vg_free(*q); } } gr{ &my_reg };
    guarded(c, "align shard", [&] {
        owner.resize((size_t)std::max<int64_t>(n_tasks, 1));
        check(vg_align_owner(tasks, n_tasks, vg_genomes_count(g), W, owner.data()));
        for (int64_t t = 0; t < n_tasks; ++t) { per_rank[(size_t)owner[(size_t)t]]++; if (owner[(size_t)t] == c->rank) mine.push_back(t); }
        std::vector<vg_task> my_tasks(mine.size());
        for (size_t i = 0; i < mine.size(); ++i) my_tasks[i] = tasks[mine[i]];
        my_stats.resize(std::max<size_t>(1, mine.size()));
        check(vg_lz_align(g, my_tasks.data(), (int64_t)my_tasks.size(), p, my_stats.data(), regions ? &my_reg : nullptr, regions ? &my_nreg : nullptr));
        // rows: sizes are known to every rank (per_rank), one padded all-gather
        for (int r = 0; r < W; ++r) pad = std::max(pad, per_rank[(size_t)r]);
        send.assign((size_t)pad, vg_pair_stat{}); all.resize((size_t)pad * W);
        if (!mine.empty()) memcpy(send.data(), my_stats.data(), sizeof(vg_pair_stat) * mine.size());
        reserve_staging(c, pad * (int64_t)sizeof(vg_pair_stat));
    });
    gather_host(c, send.data(), all.data(), pad * (int64_t)sizeof(vg_pair_stat));
    std::vector<int64_t> cursor((size_t)W, 0);
    for (int64_t t = 0; t < n_tasks; ++t) { const int r = owner[(size_t)t]; stats[t] = all[(size_t)(r * pad + cursor[(size_t)r]++)]; }
    if (regions) {
        // regions: variable length, task ids translated to the global list
        for (int64_t i = 0; i < my_nreg; ++i) my_reg[i].task = (uint32_t)mine[my_reg[i].task];
        std::vector<int64_t> cnt((size_t)W, 0);
        gather_host(c, &my_nreg, cnt.data(), sizeof(int64_t));
        int64_t rpad = 1, tot = 0; for (int r = 0; r < W; ++r) { rpad = std::max(rpad, cnt[(size_t)r]); tot += cnt[(size_t)r]; }
        std::vector<vg_region> rs, ra; vg_region* o = nullptr;
        guarded(c, "align regions", [&] {
            rs.assign((size_t)rpad, vg_region{}); ra.resize((size_t)rpad * W);
            if (my_nreg) memcpy(rs.data(), my_reg, sizeof(vg_region) * (size_t)my_nreg);
            o = (vg_region*)malloc(sizeof(vg_region) * std::max<size_t>(1, (size_t)tot));
            if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
            reserve_staging(c, rpad * (int64_t)sizeof(vg_region));
        });
        gather_host(c, rs.data(), ra.data(), rpad * (int64_t)sizeof(vg_region));
        int64_t w = 0;
        for (int r = 0; r < W; ++r) { memcpy(o + w, ra.data() + (size_t)r * rpad, sizeof(vg_region) * (size_t)cnt[(size_t)r]); w += cnt[(size_t)r]; }
        *regions = o; if (n_regions) *n_regions = tot;
    }
    VG_API_END
}

// reference ranges of `world` ranks from the candidate pairs (every pair is one task with r = a and one with r = b: the
// rule of vg_align_owner without the task list), the position of both tasks of every pair in their owners' lists (tasks
// listed in pair order: r = a first), and rank `me`'s own list
namespace {
void pairs_share(int n, const vg_pair_count* cand, int64_t n_cand, int W, int me, std::vector<int32_t>& own_ref,
                 std::vector<uint32_t>* la, std::vector<uint32_t>* lb, std::vector<int64_t>& per_rank, std::vector<vg_task>& mine) {
    const int64_t n_tasks = 2 * n_cand;
    own_ref.assign((size_t)n + 1, 0); per_rank.assign((size_t)W, 0); mine.clear();
    // (every rank of a sharded call does this for the whole pair list between the stages: a few threads from 10^5 pairs on)
    const int T = n_cand >= (1 << 17) ? std::max(1, std::min(vg_host_threads(), 8)) : 1;
    std::vector<int64_t> per_ref((size_t)n + 1, 0);
    if (T == 1) {
        for (int64_t i = 0; i < n_cand; ++i) {
            if (cand[i].a >= (uint32_t)n || cand[i].b >= (uint32_t)n) throw vg_error(VG_EINVAL, "pair id out of range");
            per_ref[cand[i].a]++; per_ref[cand[i].b]++;
        }
    } else {
        vg_parallel_chunks(n_cand, T, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; ++i) {
                if (cand[i].a >= (uint32_t)n || cand[i].b >= (uint32_t)n) throw vg_error(VG_EINVAL, "pair id out of range");
                __atomic_fetch_add(&per_ref[cand[i].a], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&per_ref[cand[i].b], 1, __ATOMIC_RELAXED);
            }
        });
    }
    int64_t before = 0;
    for (int r = 0; r < n; ++r) { own_ref[(size_t)r] = n_tasks ? (int32_t)std::min<int64_t>(W - 1, before * W / n_tasks) : 0; before += per_ref[(size_t)r]; }
    if (T == 1) {
        for (int64_t i = 0; i < n_cand; ++i) {
            const uint32_t a = cand[i].a, b = cand[i].b;
            const int ra = own_ref[a], rb = own_ref[b];
            const int64_t ia = per_rank[(size_t)ra]++;
            if (la) (*la)[(size_t)i] = (uint32_t)ia;
            if (ra == me) mine.push_back({ b, a });
            const int64_t ib = per_rank[(size_t)rb]++;
            if (lb) (*lb)[(size_t)i] = (uint32_t)ib;
            if (rb == me) mine.push_back({ a, b });
        }
        return;
    }
    // positions in the owners' lists are running counts in pair order: per chunk the tasks of every owner are counted,
    // the chunks' counts become their starting positions, and every chunk then numbers its own pairs
    std::vector<std::vector<int64_t>> cnt((size_t)T, std::vector<int64_t>((size_t)W, 0));
    vg_parallel_chunks(n_cand, T, [&](int64_t lo, int64_t hi, int t) {
        auto& c = cnt[(size_t)t];
        for (int64_t i = lo; i < hi; ++i) { c[(size_t)own_ref[cand[i].a]]++; c[(size_t)own_ref[cand[i].b]]++; }
    });
    std::vector<int64_t> mine_at((size_t)T + 1, 0);
    for (int r = 0; r < W; ++r) for (int t = 0; t < T; ++t) { const int64_t c = cnt[(size_t)t][(size_t)r]; cnt[(size_t)t][(size_t)r] = per_rank[(size_t)r]; per_rank[(size_t)r] += c; if (r == me) mine_at[(size_t)t + 1] = c; }
    for (int t = 0; t < T; ++t) mine_at[(size_t)t + 1] += mine_at[(size_t)t];
    mine.resize((size_t)mine_at[(size_t)T]);
    vg_parallel_chunks(n_cand, T, [&](int64_t lo, int64_t hi, int t) {
        auto& c = cnt[(size_t)t]; int64_t at = mine_at[(size_t)t];
        for (int64_t i = lo; i < hi; ++i) {
            const uint32_t a = cand[i].a, b = cand[i].b;
            const int ra = own_ref[a], rb = own_ref[b];
            const int64_t ia = c[(size_t)ra]++;
            if (la) (*la)[(size_t)i] = (uint32_t)ia;
            if (ra == me) mine[(size_t)at++] = { b, a };
            const int64_t ib = c[(size_t)rb]++;
            if (lb) (*lb)[(size_t)i] = (uint32_t)ib;
            if (rb == me) mine[(size_t)at++] = { a, b };
        }
    });
}
}
// rank `rank`'s share of the align tasks of the candidate pairs under the reference-range partition (a pure function of
// the pairs: what vg_lz_align_pairs_sharded lists before it launches; exported for tools and tests)
extern "C" int vg_align_pairs_share(const vg_genomes* g, const vg_pair_count* cand, int64_t n_cand, int world, int rank,
                                    vg_task** tasks_out, int64_t* n_tasks_out) {
    VG_API_BEGIN
    if (!g || (!cand && n_cand) || !tasks_out || !n_tasks_out || world < 1 || rank < 0 || rank >= world) throw vg_error(VG_EINVAL, "vg_align_pairs_share: bad arguments");
    std::vector<int32_t> own_ref; std::vector<int64_t> per_rank; std::vector<vg_task> mine;
    pairs_share(vg_genomes_count(g), cand, n_cand, world, rank, own_ref, nullptr, nullptr, per_rank, mine);
    vg_task* o = (vg_task*)malloc(sizeof(vg_task) * std::max<size_t>(1, mine.size()));
    if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
    if (!mine.empty()) memcpy(o, mine.data(), sizeof(vg_task) * mine.size());
    *tasks_out = o; *n_tasks_out = (int64_t)mine.size();
    VG_API_END
}

// ------------------------------------------------------------------ align, sharded, from the candidate PAIRS
// vg_align_tasks + vg_lz_align_sharded in one call whose host work is off the critical path: a rank needs only ITS tasks to
// start its kernels -- the genomes' task counts follow from the pairs (every pair is one task with r = a and one with
// r = b), the reference ranges from the counts, and the rank's tasks are listed in pair order -- so the canonical task
// list of the whole set (3 ms per 450 000 pairs, which every rank of vg_lz_align_sharded's caller builds BEFORE anything
// is launched) is assembled on a helper thread while the kernels run, and is only needed to place the gathered rows.
// Every rank receives the canonical task list (vg_free) and all rows (vg_free).  No regions (use vg_lz_align_sharded).
extern "C" int vg_lz_align_pairs_sharded(vg_genomes* g, const vg_pair_count* cand, int64_t n_cand, const vg_lz_params* p, const vg_comm* c,
                                         vg_task** tasks_out, int64_t* n_tasks_out, vg_pair_stat** stats_out) {
    VG_API_BEGIN
    if (!g || !c || (!cand && n_cand) || !p || !tasks_out || !n_tasks_out || !stats_out) throw vg_error(VG_EINVAL, "vg_lz_align_pairs_sharded: null argument");
    *tasks_out = nullptr; *n_tasks_out = 0; *stats_out = nullptr;
    const int n = vg_genomes_count(g);
    const int64_t n_tasks = 2 * n_cand;
    struct free_guard2 { void* p = nullptr; ~free_guard2() { if (p) free(p); } } tasks_g, stats_g;
    if (!c->exchanges()) {
        check(vg_lz_prepare(g, cand, n_cand, p));
        int64_t nt = 0;
        check(vg_align_tasks(g, cand, n_cand, (vg_task**)&tasks_g.p, &nt));
        stats_g.p = malloc(sizeof(vg_pair_stat) * (size_t)std::max<int64_t>(1, nt));
        if (!stats_g.p) throw vg_error(VG_ENOMEM, "out of host memory");
        check(vg_lz_align(g, (const vg_task*)tasks_g.p, nt, p, (vg_pair_stat*)stats_g.p, nullptr, nullptr));
        *tasks_out = (vg_task*)tasks_g.p; *n_tasks_out = nt; *stats_out = (vg_pair_stat*)stats_g.p; tasks_g.p = nullptr; stats_g.p = nullptr;
        return VG_OK;
    }
    const int W = c->world;
    std::vector<int32_t> own_ref;
    std::vector<uint32_t> la((size_t)std::max<int64_t>(n_cand, 1)), lb((size_t)std::max<int64_t>(n_cand, 1)), perm((size_t)std::max<int64_t>(n_cand, 1));
    std::vector<int64_t> per_rank;
    std::vector<vg_pair_stat> send, all;
    int64_t pad = 1, nt = 0;
    vg_host_mark("align pairs: enter");
    // every rank must pass the SAME pairs in the SAME order (the positions of a rank's rows in the gathered blocks follow
    // from the order): an order-sensitive checksum of the list travels with the section's status word
    uint32_t cand_ck = 0;
    {
        const int T = n_cand >= (1 << 17) ? std::max(1, std::min(vg_host_threads(), 8)) : 1;
        std::vector<uint64_t> part((size_t)T, 0);
        vg_parallel_chunks(n_cand, T, [&](int64_t lo, int64_t hi, int t) {
            uint64_t h = 0;
            for (int64_t i = lo; i < hi; ++i) { uint64_t x = (((uint64_t)cand[i].a << 32) | cand[i].b) + 0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1); x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 29; h += x; }
            part[(size_t)t] = h;
        });
        uint64_t h = (uint64_t)n_cand; for (uint64_t v : part) h += v;
        cand_ck = (uint32_t)(h ^ (h >> 32));
    }
    int rc_shard = VG_OK;
    try { [&] {
        if (n_cand >= (1LL << 31)) throw vg_error(VG_EOVERFLOW, "more than 2^31 candidate pairs in one call");
        // the canonical list of the whole set: a helper thread, beside everything below
        int rc_list = VG_OK; std::string err_list;
        std::thread th([&] { rc_list = vg_align_tasks_perm(g, cand, n_cand, (vg_task**)&tasks_g.p, &nt, perm.data()); if (rc_list != VG_OK) err_list = vg_last_error(); });
        struct joiner { std::thread& t; ~joiner() { if (t.joinable()) t.join(); } } jn{ th };
        std::vector<vg_task> mine;
        pairs_share(n, cand, n_cand, W, c->rank, own_ref, &la, &lb, per_rank, mine);
        vg_host_mark("align pairs: own tasks listed");
        std::vector<vg_pair_stat> my_stats(std::max<size_t>(1, mine.size()));
        check(vg_lz_align(g, mine.data(), (int64_t)mine.size(), p, my_stats.data(), nullptr, nullptr));
        th.join();
        vg_host_mark("align pairs: canonical list joined");
        if (rc_list != VG_OK) throw vg_error(rc_list, err_list);
        for (int r = 0; r < W; ++r) pad = std::max(pad, per_rank[(size_t)r]);
        send.assign((size_t)pad, vg_pair_stat{}); all.resize((size_t)pad * W);
        if (!mine.empty()) memcpy(send.data(), my_stats.data(), sizeof(vg_pair_stat) * mine.size());
        stats_g.p = malloc(sizeof(vg_pair_stat) * (size_t)std::max<int64_t>(1, n_tasks));
        if (!stats_g.p) throw vg_error(VG_ENOMEM, "out of host memory");
        reserve_staging(c, pad * (int64_t)sizeof(vg_pair_stat));
    }(); }
    catch (const vg_error& e) { vg_set_error("%s", e.what()); rc_shard = e.code; }
    catch (const std::bad_alloc&) { vg_set_error("out of host memory"); rc_shard = VG_ENOMEM; }
    catch (const std::exception& e) { vg_set_error("%s", e.what()); rc_shard = VG_EINVAL; }
    agree_same(c, rc_shard, cand_ck, "align shard", "were given different candidate pair lists (the same pairs in the same order are required: vg_kmer_shared_sharded returns them sorted)");
    gather_host(c, send.data(), all.data(), pad * (int64_t)sizeof(vg_pair_stat));
    // rows into the canonical order: couple cidx came from pair perm[cidx]; its two tasks are that pair's (r = a) and (r = b) tasks
    {
        const vg_task* tk = (const vg_task*)tasks_g.p; vg_pair_stat* st = (vg_pair_stat*)stats_g.p;
        vg_parallel_chunks(n_cand, n_cand >= (1 << 18) ? std::min(vg_host_threads(), 8) : 1, [&](int64_t lo, int64_t hi, int) {
            for (int64_t cidx = lo; cidx < hi; ++cidx) {
                const uint32_t pi = perm[(size_t)cidx]; const uint32_t a = cand[pi].a, b = cand[pi].b;
                for (int d = 0; d < 2; ++d) {
                    const uint32_t r = tk[2 * cidx + d].r;
                    const bool is_a = r == a;
                    st[2 * cidx + d] = all[(size_t)own_ref[is_a ? a : b] * (size_t)pad + (is_a ? la[pi] : lb[pi])];
                }
            }
        });
    }
    vg_host_mark("align pairs: rows placed");
    *tasks_out = (vg_task*)tasks_g.p; *n_tasks_out = nt; *stats_out = (vg_pair_stat*)stats_g.p; tasks_g.p = nullptr; stats_g.p = nullptr;
    VG_API_END
}

// ------------------------------------------------------------------ whole stages, sharded (rank 0 writes the files)
namespace {
struct genomes_guard { vg_genomes* g = nullptr; ~genomes_guard() { if (g) vg_genomes_free(g); } };
struct free_guard { void* p = nullptr; ~free_guard() { if (p) vg_free(p); } };
}

extern "C" int vg_prefilter_sharded(const char* const* fasta_paths, int n_paths, const char* out_path, const vg_prefilter_params* p, const vg_comm* c) {
    VG_API_BEGIN
    if (!fasta_paths || n_paths <= 0 || !out_path || !p || !c) throw vg_error(VG_EINVAL, "vg_prefilter_sharded: null argument");
    if (p->k < 15 || p->k > 30) throw vg_error(VG_EINVAL, "k must be in 15..30");
    if (!(p->kmers_fraction > 0.0) || p->kmers_fraction > 1.0) throw vg_error(VG_EINVAL, "kmers_fraction must be in (0,1]");
    genomes_guard gg;
    // (every rank reads the whole input on the same host: the ranks share its cores instead of each taking `num_threads`)
    const int ingest_threads = std::max(1, std::min(p->num_threads, (int)std::max(1u, std::thread::hardware_concurrency() / (unsigned)std::max(1, c->world))));
    int rc = vg_genomes_load(fasta_paths, n_paths, p->is_multifasta, ingest_threads, &gg.g);
    agree(c, rc, "ingest");
    std::vector<int64_t> sizes((size_t)std::max(1, vg_genomes_count(gg.g)));
    free_guard pairs; int64_t np = 0;
    check(vg_kmer_shared_sharded(gg.g, p->k, p->kmers_fraction, (uint32_t)std::max(1, p->min_kmers), c, sizes.data(), (vg_pair_count**)&pairs.p, &np));
    rc = VG_OK;
    if (c->rank == 0) rc = vg_write_fltr(gg.g, p->k, p->kmers_fraction, p->min_kmers, p->min_ident, p->max_seqs, sizes.data(), (const vg_pair_count*)pairs.p, np, out_path);
    agree(c, rc, "fltr.txt writer");
    VG_API_END
}

extern "C" int vg_align_sharded(const char* const* fasta_paths, int n_paths, const char* out_path, const vg_align_params* p, const vg_comm* c) {
    VG_API_BEGIN
    if (!fasta_paths || n_paths <= 0 || !out_path || !p || !c) throw vg_error(VG_EINVAL, "vg_align_sharded: null argument");
    genomes_guard gg;
    const int ingest_threads = std::max(1, std::min(p->num_threads, (int)std::max(1u, std::thread::hardware_concurrency() / (unsigned)std::max(1, c->world))));
    int rc = vg_genomes_load(fasta_paths, n_paths, p->is_multifasta, ingest_threads, &gg.g);
    free_guard pairs, tasks, regions, rows; int64_t np = 0, nt = 0, nr = 0;
    const bool want_aln = p->out_aln_path != nullptr;
    if (rc == VG_OK) rc = vg_read_filter(gg.g, p->filter_path, p->filter_threshold, (vg_pair_count**)&pairs.p, &np);
    // without an alignment table the ranks start from the pairs (the canonical list is assembled beside the kernels)
    if (rc == VG_OK && want_aln) rc = vg_align_tasks(gg.g, (const vg_pair_count*)pairs.p, np, (vg_task**)&tasks.p, &nt);
    agree(c, rc, "ingest / filter");
    std::vector<vg_pair_stat> stats_v;
    const vg_pair_stat* stats = nullptr;
    if (want_aln) {
        stats_v.resize((size_t)std::max<int64_t>(1, nt));
        check(vg_lz_align_sharded(gg.g, (const vg_task*)tasks.p, nt, &p->lz, c, stats_v.data(), (vg_region**)&regions.p, &nr));
        stats = stats_v.data();
    } else {
        check(vg_lz_align_pairs_sharded(gg.g, (const vg_pair_count*)pairs.p, np, &p->lz, c, (vg_task**)&tasks.p, &nt, (vg_pair_stat**)&rows.p));
        stats = (const vg_pair_stat*)rows.p;
    }
    rc = VG_OK;
    if (c->rank == 0) rc = vg_write_ani(gg.g, (const vg_task*)tasks.p, stats, nt, (const vg_region*)regions.p, nr, out_path, p);
    agree(c, rc, "ani.tsv writer");
    VG_API_END
}
