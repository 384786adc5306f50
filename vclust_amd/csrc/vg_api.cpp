// vg_api.cpp — the two whole-stage entry points the reference's front-end needs:
//   vg_prefilter  replaces kmer-db build + all2all + distance (vclust.py:1433-1471)
//   vg_align      replaces lz-ani all2all                    (vclust.py:1497-1521)
// Both are compositions of the finer C-ABI calls (ingest -> HBM -> integer kernels -> writers).
#include "vg_common.h"
#include <stdlib.h>
#include <thread>
#include <vector>

namespace {
// vg_set_process_ends_after_call(1) (vclust.py, for its one-shot `prefilter` / `align` processes): the process ends right
// after the call, so the genome set -- 1.5 GB of host arrays to unmap, 1.4 GB of device blocks to hand back -- is left to
// the exit instead of being released first (0.1 s that the caller would wait for).  A setter, not an environment variable:
// an embedding process that merely inherits an environment keeps the library's normal ownership.
static bool g_leak_at_exit = false;
static bool leak_at_exit() { return g_leak_at_exit; }
struct genomes_guard { vg_genomes* g = nullptr; ~genomes_guard() { if (g && !leak_at_exit()) vg_genomes_free(g); } };
struct free_guard { void* p = nullptr; ~free_guard() { if (p) vg_free(p); } };
void check(int rc) { if (rc != VG_OK) throw vg_error(rc, vg_last_error()); }
// parked clean-up (vg_defer) is released when the stage's kernels are in flight, and in any case when the call ends
struct defer_scope { defer_scope() { vg_defer_mode(true); } ~defer_scope() { vg_defer_mode(false); vg_deferred_start(); } };
// The HIP context (120-170 ms on a cold process) is created on a helper thread while the FASTA is parsed; the
// caller joins before its first device call.  Failures are left to that call, which reports them.
// It also pays the other one-time costs of the first device operations there: the runtime's fill kernel (the first
// hipMemsetAsync of a process loads it: 50-90 ms were seen in front of the first kernel of vg_kmer_shared) and the code
// object of the stage's own kernels.
struct device_warmup {
    std::thread th;
    explicit device_warmup(bool align_stage) {
        try {
            th = std::thread([align_stage] {
                try {
                    vg_require_device(); hipStream_t s = vg_stream(); (void)hipFree(nullptr);
                    void* p = vg_dev_alloc(4096);
                    (void)hipMemsetAsync(p, 0, 4096, s);
                    if (align_stage) vg_warm_align(s); else { vg_warm_prefilter(s); (void)vg_side_stream(); }     // (a second queue costs 8-20 ms to create)
                    (void)hipStreamSynchronize(s);
                    vg_dev_free(p);
                    vg_host_mark("device warm");
                } catch (...) {}
            });
        } catch (...) {}
    }
    void join() { if (th.joinable()) th.join(); }
    ~device_warmup() { join(); }
};
}
namespace {
// vg_release_device_memory on a helper thread for the lifetime of the object (joined at its end, also when the writer throws)
struct background_release {
    std::thread th;
    // (best effort: the thread selects the library's device first -- a fresh thread starts on device 0 --, and nothing it
    // throws may leave it: a failure only means the blocks stay cached until the next release)
    static void release() noexcept { try { vg_require_device(); vg_release_device_memory(); } catch (...) { (void)hipGetLastError(); } }
    background_release() { try { th = std::thread([] { release(); }); } catch (...) { release(); } }
    ~background_release() { if (th.joinable()) th.join(); }
};
}
extern "C" void vg_set_process_ends_after_call(int on) { g_leak_at_exit = on != 0; }

extern "C" int vg_prefilter(const char* const* fasta_paths, int n_paths, const char* out_path,
                            const vg_prefilter_params* p) {
    VG_API_BEGIN
    if (!fasta_paths || n_paths <= 0 || !out_path || !p) throw vg_error(VG_EINVAL, "vg_prefilter: null argument");
    if (p->k < 15 || p->k > 30) throw vg_error(VG_EINVAL, "k must be in 15..30");
    if (!(p->kmers_fraction > 0.0) || p->kmers_fraction > 1.0) throw vg_error(VG_EINVAL, "kmers_fraction must be in (0,1]");
    vg_host_mark("vg_prefilter: enter");
    vg_one_shot_scope one_shot;
    defer_scope parked;
    device_warmup warm(false);
    genomes_guard gg;
    check(vg_genomes_load_resident(fasta_paths, n_paths, p->is_multifasta, p->num_threads, &gg.g));
    warm.join();
    vg_require_device();
    std::vector<int64_t> sizes((size_t)std::max(1, vg_genomes_count(gg.g)));
    free_guard pairs; int64_t np = 0;
    // on a single device the --min-kmers cut can be applied on the GPU already
    uint32_t min_emit = (uint32_t)std::max(1, p->min_kmers);
    check(vg_kmer_shared(gg.g, p->k, p->kmers_fraction, 0, 1, min_emit, sizes.data(),
                         (vg_pair_count**)&pairs.p, &np));
    // a one-shot process: the tens of GB of workspace go back to the driver NOW, so that the scrub of that memory
    // runs beside the writer and the start of the next process (`vclust.py align`) instead of in front of its
    // first allocation
    // (handing ~10 GB back is tens of milliseconds inside the driver: on a helper thread, beside the writer, which is host work only)
    {
        background_release rel;
        check(vg_write_fltr(gg.g, p->k, p->kmers_fraction, p->min_kmers, p->min_ident, p->max_seqs,
                            sizes.data(), (const vg_pair_count*)pairs.p, np, out_path));
    }
    vg_host_mark("fltr.txt written");
    VG_API_END
}

extern "C" int vg_align(const char* const* fasta_paths, int n_paths, const char* out_path, const vg_align_params* p) {
    VG_API_BEGIN
    if (!fasta_paths || n_paths <= 0 || !out_path || !p) throw vg_error(VG_EINVAL, "vg_align: null argument");
    vg_host_mark("vg_align: enter");
    vg_one_shot_scope one_shot;
    defer_scope parked;
    device_warmup warm(true);
    genomes_guard gg;
    check(vg_genomes_load_resident(fasta_paths, n_paths, p->is_multifasta, p->num_threads, &gg.g));
    warm.join();
    vg_require_device();
    free_guard pairs, tasks, regions; int64_t np = 0, nt = 0, nr = 0;
    check(vg_read_filter(gg.g, p->filter_path, p->filter_threshold, (vg_pair_count**)&pairs.p, &np));
    vg_host_mark("filter read");
    check(vg_lz_prepare(gg.g, (const vg_pair_count*)pairs.p, np, &p->lz));          // (the first index batch is built beside the task list)
    check(vg_align_tasks(gg.g, (const vg_pair_count*)pairs.p, np, (vg_task**)&tasks.p, &nt));
    std::vector<vg_pair_stat> stats((size_t)std::max<int64_t>(1, nt));
    const bool want_aln = p->out_aln_path != nullptr;
    check(vg_lz_align(gg.g, (const vg_task*)tasks.p, nt, &p->lz, stats.data(), want_aln ? (vg_region**)&regions.p : nullptr, &nr));
    {
        background_release rel;
        check(vg_write_ani(gg.g, (const vg_task*)tasks.p, stats.data(), nt, (const vg_region*)regions.p, nr, out_path, p));
    }
    vg_host_mark("ani.tsv written");
    VG_API_END
}
