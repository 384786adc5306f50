// vg_dist.hip — one process per GPU: the sharded forms of the two stages behind the C ABI.
//
// The reference has no distributed layer (single node, threads); north_star shards the path
// over the GPUs of one node with a gather of result rows over RCCL/xGMI.  The path shards without any
// collective inside the kernels:
//   prefilter  rank r handles the k-mers whose shard hash falls into range r of `world` (vg_kmer_shared's
//              shard/n_shards): per-genome set sizes and per-pair shared counts of the shards ADD UP.  The
//              partial (a, b, count) records -- set sizes ride along as diagonal records (g, g, size) -- travel in
//              ONE padded all-gather (plus a one-word count exchange) and are summed on the device (radix sort
//              + segmented reduction).  Thresholds can only be applied to the SUM, so they follow the merge.
//   align      the task list is cut into `world` contiguous reference-id ranges with about equal task counts --
//              a pure function of the list, so no communication -- each rank indexes 1/world of the references
//              and the 12-byte rows come back in one all-gather of known sizes (regions: one more, variable).
// The exchange itself is a vg_comm: either callbacks supplied by the host application (MPI, torch.distributed
// over gloo for CPU tests, ...) or the built-in RCCL communicator (vg_comm_rccl_create: ncclAllGather on the
// library's stream; librccl is loaded on first use so that hosts without it still load this library).
// Failures are agreed on before every exchange (one status word per rank), so a rank that fails makes every rank
// return the error instead of leaving the others inside a collective.
#include "vg_common.h"
#include <rocprim/rocprim.hpp>
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <vector>

// ------------------------------------------------------------------ communicator
struct vg_comm {
    int rank = 0, world = 1;
    vg_allgather_fn allgather = nullptr; void* ctx = nullptr;   // callback form
    // built-in RCCL form
    void* nccl_lib = nullptr; void* nccl_comm = nullptr;
    int (*p_allgather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*p_destroy)(void*) = nullptr;
};

namespace {
void check(int rc) { if (rc != VG_OK) throw vg_error(rc, vg_last_error()); }

// all ranks contribute `bytes` bytes of HOST memory; recv = world * bytes in rank order
void gather_host(const vg_comm* c, const void* send, void* recv, int64_t bytes) {
    if (c->world == 1) { memcpy(recv, send, (size_t)bytes); return; }
    if (c->allgather) {
        if (c->allgather(c->ctx, send, recv, bytes, 0) != 0) throw vg_error(VG_EIO, "vg_comm: allgather callback failed");
        return;
    }
    // RCCL moves device memory: stage through HBM
    hipStream_t s = vg_stream();
    dbuf<char> d_send((size_t)std::max<int64_t>(bytes, 1)), d_recv((size_t)std::max<int64_t>(bytes, 1) * c->world);
    d_send.upload((const char*)send, (size_t)bytes, s);
    if (c->p_allgather(d_send.p, d_recv.p, (size_t)bytes, /*ncclChar*/ 0, c->nccl_comm, s) != 0) throw vg_error(VG_EIO, "ncclAllGather failed");
    d_recv.download((char*)recv, (size_t)bytes * c->world, s);
    VG_HIP(hipStreamSynchronize(s));
}
// the same for DEVICE memory (buffers of the current device)
void gather_device(const vg_comm* c, const void* send, void* recv, int64_t bytes) {
    hipStream_t s = vg_stream();
    if (c->world == 1) { VG_HIP(hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, s)); return; }
    if (c->allgather) {
        VG_HIP(hipStreamSynchronize(s));                          // the callback runs on the application's own stream
        if (c->allgather(c->ctx, send, recv, bytes, 1) != 0) throw vg_error(VG_EIO, "vg_comm: allgather callback failed");
        return;
    }
    if (c->p_allgather(send, recv, (size_t)bytes, 0, c->nccl_comm, s) != 0) throw vg_error(VG_EIO, "ncclAllGather failed");
}
// every rank learns whether any rank failed: throws the first failure on all of them
void agree(const vg_comm* c, int my_rc, const char* what) {
    std::vector<int32_t> all((size_t)c->world, 0); int32_t mine = my_rc;
    gather_host(c, &mine, all.data(), sizeof(int32_t));
    for (int r = 0; r < c->world; ++r) if (all[(size_t)r] != 0)
        throw vg_error(all[(size_t)r], std::string(what) + ": rank " + std::to_string(r) + " failed" + (r == c->rank ? std::string(": ") + vg_last_error() : std::string()));
}

struct rec_t { uint64_t key; uint64_t val; };
__global__ void k_split_rec(const rec_t* __restrict__ rec, const int64_t* __restrict__ valid_prefix, int world, int64_t pad, int64_t n_out,
                            uint64_t* __restrict__ keys, uint64_t* __restrict__ vals) {
    // gathered layout: world blocks of `pad` records, the first cnt[r] of block r are real; valid_prefix[r] = real records before block r
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)world * pad; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / pad); const int64_t j = i - (int64_t)r * pad;
        const int64_t cnt = valid_prefix[r + 1] - valid_prefix[r];
        if (j < cnt) { const rec_t x = rec[i]; keys[valid_prefix[r] + j] = x.key; vals[valid_prefix[r] + j] = x.val; }
    }
}
}  // namespace

extern "C" int vg_comm_create(int rank, int world, vg_allgather_fn allgather, void* ctx, vg_comm** out) {
    VG_API_BEGIN
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !allgather)) throw vg_error(VG_EINVAL, "vg_comm_create: bad arguments");
    vg_comm* c = new vg_comm; c->rank = rank; c->world = world; c->allgather = allgather; c->ctx = ctx;
    *out = c;
    VG_API_END
}

static void* load_rccl() {
    for (const char* nm : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) return h; }
    throw vg_error(VG_EIO, std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found"));
}

extern "C" int vg_rccl_unique_id(void* out, int64_t bytes) {
    VG_API_BEGIN
    if (!out || bytes < 128) throw vg_error(VG_EINVAL, "vg_rccl_unique_id: needs a 128-byte buffer");
    void* h = load_rccl();
    auto get = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    if (!get) throw vg_error(VG_EIO, "librccl has no ncclGetUniqueId");
    if (get(out) != 0) throw vg_error(VG_EIO, "ncclGetUniqueId failed");
    VG_API_END
}

extern "C" int vg_comm_rccl_create(int rank, int world, const void* unique_id, int64_t id_bytes, vg_comm** out) {
    VG_API_BEGIN
    if (!out || !unique_id || id_bytes < 128 || world < 1 || rank < 0 || rank >= world) throw vg_error(VG_EINVAL, "vg_comm_rccl_create: bad arguments");
    vg_require_device();
    void* h = load_rccl();
    struct id128 { char b[128]; } id; memcpy(id.b, unique_id, 128);
    auto init = (int (*)(void**, int, id128, int))dlsym(h, "ncclCommInitRank");
    auto ag = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    auto destroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    if (!init || !ag || !destroy) throw vg_error(VG_EIO, "librccl lacks ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    void* comm = nullptr;
    if (init(&comm, world, id, rank) != 0) throw vg_error(VG_EIO, "ncclCommInitRank failed");
    vg_comm* c = new vg_comm; c->rank = rank; c->world = world; c->nccl_lib = h; c->nccl_comm = comm; c->p_allgather = ag; c->p_destroy = destroy;
    *out = c;
    VG_API_END
}

extern "C" void vg_comm_free(vg_comm* c) {
    if (!c) return;
    if (c->nccl_comm && c->p_destroy) { (void)hipDeviceSynchronize(); (void)c->p_destroy(c->nccl_comm); }
    delete c;
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
    VG_API_END
}

// owner rank of every task (reference-range partition): references are cut into `world` contiguous id ranges
// holding about the same number of tasks; a pure function of the task list
extern "C" int vg_align_owner(const vg_task* tasks, int64_t n_tasks, int n_genomes, int world, int32_t* owner) {
    VG_API_BEGIN
    if ((!tasks && n_tasks) || (!owner && n_tasks) || world < 1 || n_genomes < 0) throw vg_error(VG_EINVAL, "vg_align_owner: bad arguments");
    std::vector<int64_t> per_ref((size_t)n_genomes + 1, 0);
    for (int64_t t = 0; t < n_tasks; ++t) { if (tasks[t].r >= (uint32_t)n_genomes) throw vg_error(VG_EINVAL, "task id out of range"); per_ref[tasks[t].r]++; }
    std::vector<int32_t> own_ref((size_t)n_genomes + 1, 0);
    int64_t before = 0;
    for (int r = 0; r < n_genomes; ++r) {
        own_ref[(size_t)r] = n_tasks ? (int32_t)std::min<int64_t>(world - 1, before * world / n_tasks) : 0;
        before += per_ref[(size_t)r];
    }
    for (int64_t t = 0; t < n_tasks; ++t) owner[t] = own_ref[tasks[t].r];
    VG_API_END
}

// ------------------------------------------------------------------ prefilter, sharded
extern "C" int vg_kmer_shared_sharded(vg_genomes* g, int k, double fraction, uint32_t min_shared, const vg_comm* c,
                                      int64_t* set_sizes, vg_pair_count** pairs, int64_t* n_pairs) {
    VG_API_BEGIN
    if (!g || !c || !set_sizes || !pairs || !n_pairs) throw vg_error(VG_EINVAL, "vg_kmer_shared_sharded: null argument");
    *pairs = nullptr; *n_pairs = 0;
    const int n = vg_genomes_count(g);
    if (c->world == 1) { check(vg_kmer_shared(g, k, fraction, 0, 1, min_shared, set_sizes, pairs, n_pairs)); return VG_OK; }
    // this rank's shard of the k-mer range: partial sizes and partial counts (every pair with >= 1 shared k-mer here)
    std::vector<int64_t> part_sizes((size_t)std::max(n, 1), 0);
    vg_pair_count* loc = nullptr; int64_t n_loc = 0;
    int rc = vg_kmer_shared(g, k, fraction, c->rank, c->world, 1u, part_sizes.data(), &loc, &n_loc);
    struct guard { void* p; ~guard() { if (p) vg_free(p); } } gl{ loc };
    agree(c, rc, "prefilter shard");
    // records: (a << 32 | b, count) and the diagonal (g << 32 | g, size)
    const int64_t n_rec = n_loc + n;
    std::vector<rec_t> rec((size_t)std::max<int64_t>(n_rec, 1));
    for (int64_t i = 0; i < n_loc; ++i) rec[(size_t)i] = { ((uint64_t)loc[i].a << 32) | loc[i].b, loc[i].shared };
    for (int i = 0; i < n; ++i) rec[(size_t)(n_loc + i)] = { ((uint64_t)i << 32) | (uint64_t)i, (uint64_t)part_sizes[(size_t)i] };
    std::vector<int64_t> cnt((size_t)c->world, 0);
    gather_host(c, &n_rec, cnt.data(), sizeof(int64_t));
    int64_t pad = 1, total = 0; std::vector<int64_t> prefix((size_t)c->world + 1, 0);
    for (int r = 0; r < c->world; ++r) { pad = std::max(pad, cnt[(size_t)r]); prefix[(size_t)r + 1] = prefix[(size_t)r] + cnt[(size_t)r]; }
    total = prefix[(size_t)c->world];
    hipStream_t s = vg_stream();
    dbuf<rec_t> d_send((size_t)pad), d_all((size_t)pad * c->world);
    d_send.zero(s);
    if (n_rec) d_send.upload(rec.data(), (size_t)n_rec, s);
    gather_device(c, d_send.p, d_all.p, pad * (int64_t)sizeof(rec_t));
    // sum the partial records on the device: sort by key, reduce by key
    dbuf<int64_t> d_prefix((size_t)c->world + 1); d_prefix.upload(prefix.data(), prefix.size(), s);
    const size_t nt = (size_t)std::max<int64_t>(total, 1);
    dbuf<uint64_t> keys(nt), vals(nt), keys2(nt), vals2(nt), ukeys(nt), usums(nt); dbuf<unsigned long long> d_nu(1);
    hipLaunchKernelGGL(k_split_rec, dim3(1024), dim3(256), 0, s, d_all.p, (const int64_t*)d_prefix.p, c->world, pad, total, keys.p, vals.p);
    size_t tb = 0, tb2 = 0;
    VG_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys2.p, vals.p, vals2.p, (size_t)total, 0u, 64u, s));
    VG_HIP(rocprim::reduce_by_key(nullptr, tb2, keys2.p, vals2.p, (size_t)total, ukeys.p, usums.p, d_nu.p, rocprim::plus<uint64_t>(), rocprim::equal_to<uint64_t>(), s));
    dbuf<char> tmp(std::max(tb, tb2));
    VG_HIP(rocprim::radix_sort_pairs((void*)tmp.p, tb, keys.p, keys2.p, vals.p, vals2.p, (size_t)total, 0u, 64u, s));
    VG_HIP(rocprim::reduce_by_key((void*)tmp.p, tb2, keys2.p, vals2.p, (size_t)total, ukeys.p, usums.p, d_nu.p, rocprim::plus<uint64_t>(), rocprim::equal_to<uint64_t>(), s));
    unsigned long long nu = 0; d_nu.download(&nu, 1, s); VG_HIP(hipStreamSynchronize(s));
    std::vector<uint64_t> hk((size_t)nu), hv((size_t)nu);
    if (nu) { ukeys.download(hk.data(), (size_t)nu, s); usums.download(hv.data(), (size_t)nu, s); VG_HIP(hipStreamSynchronize(s)); }
    for (int i = 0; i < n; ++i) set_sizes[i] = 0;
    vg_pair_count* out = (vg_pair_count*)malloc(sizeof(vg_pair_count) * std::max<size_t>(1, (size_t)nu));
    if (!out) throw vg_error(VG_ENOMEM, "out of host memory");
    int64_t m = 0;
    for (size_t i = 0; i < (size_t)nu; ++i) {
        const uint32_t a = (uint32_t)(hk[i] >> 32), b = (uint32_t)hk[i];
        if (a == b) { if ((int)a < n) set_sizes[a] = (int64_t)hv[i]; }
        else if (hv[i] >= min_shared) { out[m].a = a; out[m].b = b; out[m].shared = (uint32_t)hv[i]; ++m; }
    }
    *pairs = out; *n_pairs = m;
    VG_API_END
}

// ------------------------------------------------------------------ align, sharded
extern "C" int vg_lz_align_sharded(vg_genomes* g, const vg_task* tasks, int64_t n_tasks, const vg_lz_params* p, const vg_comm* c,
                                   vg_pair_stat* stats, vg_region** regions, int64_t* n_regions) {
    VG_API_BEGIN
    if (!g || !c || (!tasks && n_tasks) || !p || (!stats && n_tasks)) throw vg_error(VG_EINVAL, "vg_lz_align_sharded: null argument");
    if (c->world == 1) { check(vg_lz_align(g, tasks, n_tasks, p, stats, regions, n_regions)); return VG_OK; }
    if (regions) { *regions = nullptr; if (n_regions) *n_regions = 0; }
    std::vector<int32_t> owner((size_t)std::max<int64_t>(n_tasks, 1));
    check(vg_align_owner(tasks, n_tasks, vg_genomes_count(g), c->world, owner.data()));
    std::vector<int64_t> mine; std::vector<int64_t> per_rank((size_t)c->world, 0);
    for (int64_t t = 0; t < n_tasks; ++t) { per_rank[(size_t)owner[(size_t)t]]++; if (owner[(size_t)t] == c->rank) mine.push_back(t); }
    std::vector<vg_task> my_tasks(mine.size());
    for (size_t i = 0; i < mine.size(); ++i) my_tasks[i] = tasks[mine[i]];
    std::vector<vg_pair_stat> my_stats(std::max<size_t>(1, mine.size()));
    vg_region* my_reg = nullptr; int64_t my_nreg = 0;
    int rc = vg_lz_align(g, my_tasks.data(), (int64_t)my_tasks.size(), p, my_stats.data(), regions ? &my_reg : nullptr, regions ? &my_nreg : nullptr);
    struct guard { void* q; ~guard() { if (q) vg_free(q); } } gr{ my_reg };
    agree(c, rc, "align shard");
    // rows: sizes are known to every rank (per_rank), one padded all-gather
    int64_t pad = 1; for (int r = 0; r < c->world; ++r) pad = std::max(pad, per_rank[(size_t)r]);
    std::vector<vg_pair_stat> send((size_t)pad), all((size_t)pad * c->world);
    memset(send.data(), 0, sizeof(vg_pair_stat) * (size_t)pad);
    if (!mine.empty()) memcpy(send.data(), my_stats.data(), sizeof(vg_pair_stat) * mine.size());
    gather_host(c, send.data(), all.data(), pad * (int64_t)sizeof(vg_pair_stat));
    std::vector<int64_t> cursor((size_t)c->world, 0);
    for (int64_t t = 0; t < n_tasks; ++t) { const int r = owner[(size_t)t]; stats[t] = all[(size_t)(r * pad + cursor[(size_t)r]++)]; }
    if (regions) {
        // regions: variable length, task ids translated to the global list
        for (int64_t i = 0; i < my_nreg; ++i) my_reg[i].task = (uint32_t)mine[my_reg[i].task];
        std::vector<int64_t> cnt((size_t)c->world, 0);
        gather_host(c, &my_nreg, cnt.data(), sizeof(int64_t));
        int64_t rpad = 1, tot = 0; for (int r = 0; r < c->world; ++r) { rpad = std::max(rpad, cnt[(size_t)r]); tot += cnt[(size_t)r]; }
        std::vector<vg_region> rs((size_t)rpad), ra((size_t)rpad * c->world);
        memset(rs.data(), 0, sizeof(vg_region) * (size_t)rpad);
        if (my_nreg) memcpy(rs.data(), my_reg, sizeof(vg_region) * (size_t)my_nreg);
        gather_host(c, rs.data(), ra.data(), rpad * (int64_t)sizeof(vg_region));
        vg_region* o = (vg_region*)malloc(sizeof(vg_region) * std::max<size_t>(1, (size_t)tot));
        if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
        int64_t w = 0;
        for (int r = 0; r < c->world; ++r) { memcpy(o + w, ra.data() + (size_t)r * rpad, sizeof(vg_region) * (size_t)cnt[(size_t)r]); w += cnt[(size_t)r]; }
        *regions = o; if (n_regions) *n_regions = tot;
    }
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
    int rc = vg_genomes_load(fasta_paths, n_paths, p->is_multifasta, p->num_threads, &gg.g);
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
    int rc = vg_genomes_load(fasta_paths, n_paths, p->is_multifasta, p->num_threads, &gg.g);
    free_guard pairs, tasks, regions; int64_t np = 0, nt = 0, nr = 0;
    if (rc == VG_OK) rc = vg_read_filter(gg.g, p->filter_path, p->filter_threshold, (vg_pair_count**)&pairs.p, &np);
    if (rc == VG_OK) rc = vg_align_tasks(gg.g, (const vg_pair_count*)pairs.p, np, (vg_task**)&tasks.p, &nt);
    agree(c, rc, "ingest / filter");
    std::vector<vg_pair_stat> stats((size_t)std::max<int64_t>(1, nt));
    const bool want_aln = p->out_aln_path != nullptr;
    check(vg_lz_align_sharded(gg.g, (const vg_task*)tasks.p, nt, &p->lz, c, stats.data(), want_aln ? (vg_region**)&regions.p : nullptr, &nr));
    rc = VG_OK;
    if (c->rank == 0) rc = vg_write_ani(gg.g, (const vg_task*)tasks.p, stats.data(), nt, (const vg_region*)regions.p, nr, out_path, p);
    agree(c, rc, "ani.tsv writer");
    VG_API_END
}
