/*
 * vclust_gpu.h — C ABI of libvclust_gpu.so, the MI355X-native replacement of the two native
 * tools on Vclust's prefilter -> align hot path.
 *
 * The reference has no library ABI for this path: vclust.py drives `bin/kmer-db` and
 * `bin/lz-ani` through subprocess.run() (vclust.py:762-807).  Every entry point below
 * names the process invocation it replaces:
 *
 *   vg_prefilter      <- kmer-db build (vclust.py:953-964) + all2all-sp/-parts
 *                        (vclust.py:1005-1017) + distance ani-shorter (vclust.py:1045-1055),
 *                        i.e. the whole body of handle_prefilter (vclust.py:1433-1471)
 *   vg_align          <- lz-ani all2all (vclust.py:1142-1181, run at vclust.py:1521)
 *   vg_version        <- `kmer-db -version` / `lz-ani --version` (vclust.py:1323-1331)
 *
 * The finer-grained functions (genome sets, integer kernels, writers) are what the two
 * calls above are made of; they are exported so that (a) the Python front-end can shard the
 * integer work over one process per GPU and gather rows with torch.distributed (RCCL), and
 * (b) the parity tests can compare integers with the CPU oracle.
 *
 * Conventions: every function returns 0 on success and a negative code on error;
 * vg_last_error() returns a thread-local, library-owned, NUL-terminated message.  Calls
 * block.  Paths are UTF-8, owned by the caller.  Arrays returned through `T** out` are
 * owned by the library and released with vg_free().  No C++ exceptions cross the ABI.
 * There is NO CPU fallback: without a HIP device every compute call fails with VG_ENODEV.
 * One process drives one GPU: the library keeps one device, one pair of queues and one workspace cache per process, and
 * its compute entry points are to be called by one thread at a time (they use many threads and the whole GPU themselves).
 */
#ifndef VCLUST_GPU_H
#define VCLUST_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    VG_OK = 0, VG_EINVAL = -1, VG_EIO = -2, VG_ENODEV = -3, VG_EHIP = -4, VG_ENOMEM = -5,
    VG_EOVERFLOW = -6
};

const char* vg_version(void);
const char* vg_last_error(void);
void        vg_free(void* p);
/* number of visible HIP devices (0 when there is none); selects the device used by this
 * process for all later calls (one process per GPU) */
int vg_device_count(void);
int vg_set_device(int device);
/* the library keeps released device blocks in a cache (no hipMalloc on the hot path); this returns them to
 * the driver, e.g. before another process needs the HBM -- together with the reference indexes a vg_lz_prepare
 * may have left behind (that plan is process-global: one per process, replaced by the next vg_lz_prepare / vg_lz_align) */
void vg_release_device_memory(void);
/* allocator self-test (no reference call site): `cycles` rounds of allocating blocks of the given byte sizes through the
 * library's device allocator, writing and reading back a pattern at both ends of each, releasing them and returning the
 * cache to the driver */
int vg_alloc_selftest(const int64_t* sizes, int n_sizes, int cycles);
/* blocking copy between host memory and memory of the current device (for vg_comm callbacks that stage through the host) */
int vg_copy(void* dst, const void* src, int64_t bytes, int to_host);

/* ------------------------------------------------------------------ genome sets ------- */
typedef struct vg_genomes vg_genomes;

/* FASTA / FASTA.gz ingest.  n_paths == 1 && multisample: one genome per record
 * (`-multisample-fasta`, vclust.py:962-963 / `--multisample-fasta true`, :1159-1160);
 * otherwise one genome per file, records joined by one N (vclust.py:692-700).
 * Genome name = first header token (multisample) or file name (directory mode). */
int vg_genomes_load(const char* const* paths, int n_paths, int multisample, int n_threads,
                    vg_genomes** out);
/* synthetic / in-memory input: codes 0..3 = ACGT, >3 = N; genome g = codes[offsets[g] ..
 * offsets[g+1]); names may be NULL ("g<idx>") */
int vg_genomes_from_codes(const uint8_t* codes, const int64_t* offsets, int n_genomes,
                          const char* const* names, vg_genomes** out);
void        vg_genomes_free(vg_genomes* g);
int         vg_genomes_count(const vg_genomes* g);
int64_t     vg_genomes_total_len(const vg_genomes* g);
int         vg_genomes_lengths(const vg_genomes* g, int64_t* out /* n */);
const char* vg_genomes_name(const vg_genomes* g, int idx);
/* the bases of genome idx as the set holds them on the host (codes 0..3 = ACGT, 4 = N), out = len[idx] bytes:
 * the inverse of the packing, for checks of the reader (no reference call site) */
int         vg_genomes_codes(const vg_genomes* g, int idx, uint8_t* out);
/* 2-bit packed bases + N mask -> HBM of the current device (idempotent); the kernels' own copy of the bases -- bit planes,
 * DESIGN.md section 3 -- is made from them on the device at the first prefilter / align call */
int vg_genomes_to_device(vg_genomes* g);

/* ------------------------------------------------------------------ prefilter --------- */
typedef struct { uint32_t a, b, shared; } vg_pair_count;   /* a > b (input order ids) */

/* Integer core of the prefilter on the GPU: per-genome distinct canonical k-mer counts and
 * the sparse all-vs-all shared-k-mer counts (K1+K2 of SURVEY §8a).
 *   fraction        --kmers-fraction (vclust.py:241-248); 1.0 = all k-mers
 *   shard/n_shards  k-mer range handled by this call -- a range of the scrambled key's top bits for sets below 2^32 bases
 *                   without a fraction, a hash of its low bits otherwise -- (multi-GPU: one shard per rank;
 *                   set sizes and shared counts of the shards ADD UP)
 *   min_shared      pairs with fewer shared k-mers are not emitted (use 1 when n_shards>1)
 * set_sizes: n entries.  pairs: unordered; released with vg_free(). */
int vg_kmer_shared(vg_genomes* g, int k, double fraction, int shard, int n_shards,
                   uint32_t min_shared, int64_t* set_sizes, vg_pair_count** pairs,
                   int64_t* n_pairs);
/* distinct canonical k-mers of one genome, ascending (parity tests) */
int vg_kmer_set(vg_genomes* g, int idx, int k, double fraction, uint64_t** out, int64_t* n_out);

/* K3+K4: ani-shorter transform, thresholds, fltr.txt (example/output/fltr.txt layout).
 * Sums duplicate (a,b) entries first, so concatenated per-shard outputs are accepted. */
int vg_write_fltr(const vg_genomes* g, int k, double fraction, int min_kmers, double min_ident,
                  int max_seqs, const int64_t* set_sizes, const vg_pair_count* pairs,
                  int64_t n_pairs, const char* out_path);

/* K3 without the file: the pairs that vg_write_fltr would print (shared >= min_kmers and
 * ani-shorter >= min_ident, same arithmetic), for a prefilter -> align hand-over in memory.
 * One entry per pair expected (vg_kmer_shared output, or merged shard outputs); *out is malloc'ed. */
int vg_filter_pairs(int k, int min_kmers, double min_ident, const int64_t* set_sizes, int64_t n_genomes,
                    const vg_pair_count* pairs, int64_t n_pairs, vg_pair_count** out, int64_t* n_out);

typedef struct {            /* mirrors the prefilter sub-parser, vclust.py:208-262 */
    int    k;               /* -k, 15..30 */
    int    min_kmers;       /* --min-kmers */
    double min_ident;       /* --min-ident */
    int    batch_size;      /* --batch-size: accepted, results do not depend on it */
    double kmers_fraction;  /* --kmers-fraction */
    int    max_seqs;        /* --max-seqs */
    int    num_threads;     /* host threads for ingest */
    int    verbosity;
    int    is_multifasta;   /* args.is_multifasta (vclust.py:689-693) */
} vg_prefilter_params;
int vg_prefilter(const char* const* fasta_paths, int n_paths, const char* out_path,
                 const vg_prefilter_params* p);
/* on != 0: the process ends right after its vg_prefilter / vg_align call, so the call leaves the genome set (GBs of host
 * and device memory) to the process exit instead of releasing it piece by piece first (vclust.py's one-shot processes;
 * no reference call site).  Off by default: an embedding process keeps the normal ownership. */
void vg_set_process_ends_after_call(int on);

/* ------------------------------------------------------------------ align ------------- */
typedef struct { int mal, msl, mrd, mqd, reg, aw, am, ar; } vg_lz_params;  /* vclust.py:363-418 */
typedef struct { uint32_t q, r; } vg_task;       /* ordered pair, ids in input order */
typedef struct { uint32_t n_match, aln_len, n_regions; } vg_pair_stat;     /* L5 integers */
typedef struct {            /* one local alignment, 0-based inclusive; r* in fwd|N|rc space */
    uint32_t task; int32_t qstart, qend, rstart, rend; int32_t n_match;
} vg_region;

/* The constants of the LZ restatement that a handful of events of the reference's 12-genome example decide (DESIGN.md
 * section 2, profiles/r04_lz_fit_leave_one_out.md) -- everything else of the parse is held by dozens to thousands of
 * golden regions.  Process-wide; NULL restores the fitted values.  No reference call site: lz-ani has no such options;
 * they exist so that a maintainer with the upstream binaries can re-decide them (tools/compare_with_upstream.py).
 *   weak_seed_ratio  3   R3: a seed shorter than 1/ratio of the literal run it bridges gives one symbol of the anchor
 *                        margin away (0 = never)                                                  [one event]
 *   anchor_margin   -1   R3: a far anchor replaces the seed when longer by MORE than this; -1 = msl - 1 (6 against 7: two pairs)
 *   seed_choice      3   R3: among seeds 3 = longest, then closest to the prediction; 1 = closest, then longest   [three regions]
 * (The fourth thin constant, the oracle's strand separator -- two pairs --, has no counterpart here: the kernels keep one
 * separator symbol and confine every match, extension, window and gap score to its strand by bounds.) */
typedef struct { int weak_seed_ratio, anchor_margin, seed_choice; } vg_lz_fit;
void vg_set_lz_fit(const vg_lz_fit* f);

/* LZ parse of every task on the GPU.  stats: n_tasks entries (caller-owned).
 * regions/n_regions may be NULL; otherwise all kept regions, vg_free(): rows and regions come from ONE parse (round 6); the
 * regions of a task are contiguous and in query order (tasks in the library's reference-grouped order: sort on `task` if
 * the caller's order is needed -- vg_write_ani does). */
int vg_lz_align(vg_genomes* g, const vg_task* tasks, int64_t n_tasks, const vg_lz_params* p,
                vg_pair_stat* stats, vg_region** regions, int64_t* n_regions);

/* Optional head start of vg_lz_align (no reference call site: lz-ani is one process): the indexes of the genomes named
 * by the candidate pairs are planned and the first batch is queued on the device at once, so that they are built while
 * the caller assembles the task list (vg_align_tasks).  vg_lz_align takes them over when its tasks name exactly these
 * references under the same parameters; otherwise they are dropped.  Results never depend on it. */
int vg_lz_prepare(vg_genomes* g, const vg_pair_count* pairs, int64_t n_pairs, const vg_lz_params* p);

/* L1: stable sort by length, descending.  order[rank] = input index. */
int vg_align_order(const vg_genomes* g, int32_t* order /* n */);
/* L2: candidate pairs (input-order ids, a > b) from a Kmer-db filter file with value >= thr,
 * or all pairs when path == NULL */
int vg_read_filter(const vg_genomes* g, const char* path, double thr,
                   vg_pair_count** pairs, int64_t* n_pairs);

/* L7: the canonical ordered-pair list of a candidate set: for ranks a < b the couple
 * (q = b, r = a), (q = a, r = b), couples ascending in (a, b).  ids are input-order ids. */
int vg_align_tasks(const vg_genomes* g, const vg_pair_count* pairs, int64_t n_pairs,
                   vg_task** tasks, int64_t* n_tasks);
/* HBM budget (bytes) for the per-reference indexes of one vg_lz_align batch (default 24 GiB; without a call a set whose
 * indexes all fit in twice the default is built as one batch) */
void vg_set_index_budget(int64_t bytes);
/* vg_kmer_shared cuts sets that exceed the 32-bit row numbering of one pass (2^32 padded bases)
 * into sub-shards of the k-mer range automatically -- the role of `--batch-size` in the
 * reference (vclust.py:229-239, 1403-1442); n > 0 forces that many sub-shards (tests), 0 = automatic. */
void vg_set_subshards(int n);
/* How one RANGE shard call of vg_kmer_shared (shard / n_shards, sets below 2^32 bases) obtains its kept k-mers:
 * 0 (default) = it scans every base of the set itself; 1 = the sliced scan of the multi-GPU path (rank `shard` scans 1/n_shards
 * of the bases and receives the kept masks and level-1 counts of its k-mer range from the others) with the peers' slices
 * computed by this process: one GPU stands in for the world (tools/strong_scaling_sim.py, tests).  Results are identical.
 * vg_kmer_shared_sharded always exchanges when the sliced scan applies; no reference call site (kmer-db is one process). */
void vg_set_range_scan(int mode);
/* Placement trials: the first dense whole-set pass of vg_kmer_shared in a long-lived process may
 * repeat itself on up to n freshly allocated workspaces and keep the fastest placement (same results; ~0.25 s per
 * placement at 100 k genomes, once).  n <= 1 = none, THE DEFAULT: an embedding application opts in (bench.py does, and
 * says so in its line).  Any failure inside a trial pass is swallowed: the call returns what its first pass computed.
 * No reference call site (kmer-db is one process per call). */
void vg_set_placement_trials(int n);

typedef struct {            /* mirrors the align sub-parser and cmd_lzani, vclust.py:290-421, 1142-1181 */
    vg_lz_params lz;
    double out_tani, out_gani, out_ani, out_qcov, out_rcov;   /* 0 = off */
    const char* filter_path; double filter_threshold;         /* NULL = all-vs-all */
    const char* out_aln_path;                                 /* NULL = none */
    const char* const* out_columns; int n_out_columns;        /* ALIGN_OUTFMT[fmt] */
    int num_threads; int verbosity; int is_multifasta;
} vg_align_params;
/* L6-L8: rows + ids file (+ alignment table when regions != NULL and out_aln_path set).
 * tasks/stats: 2 entries per unordered pair, (q=hi-rank, r=lo-rank) then the reverse. */
int vg_write_ani(const vg_genomes* g, const vg_task* tasks, const vg_pair_stat* stats,
                 int64_t n_tasks, const vg_region* regions, int64_t n_regions,
                 const char* out_path, const vg_align_params* p);
int vg_align(const char* const* fasta_paths, int n_paths, const char* out_path,
             const vg_align_params* p);

/* ------------------------------------------------------------------ one process per GPU - */
/* The reference is single-node / thread-parallel (no distributed layer); these entry points
 * shard the two stages over `world` processes, one GPU each (vg_set_device before the first call).  The only
 * communication is an all-gather of integer records, supplied as a vg_comm:
 *   vg_comm_create       the host application's own exchange (MPI_Allgather, torch.distributed, ...): the callback
 *                        gathers `bytes` bytes from every rank into recv (world * bytes, rank order); on_device != 0
 *                        means send / recv are device pointers of the current HIP device, else host pointers
 *   vg_comm_rccl_create  built-in: RCCL (ncclAllGather over xGMI) on the library's stream; rank 0 obtains the
 *                        128-byte id with vg_rccl_unique_id and hands it to the other ranks by any means
 * A rank that fails -- in its shard, in an allocation, in a merge -- makes every rank return the error: every compute
 * section between two exchanges feeds its status into the next agreement.  VG_DIST_FORCE=1 (tests) runs the exchanges
 * with a world of one. */
typedef struct vg_comm vg_comm;
typedef int (*vg_allgather_fn)(void* ctx, const void* send, void* recv, int64_t bytes, int on_device);
int  vg_comm_create(int rank, int world, vg_allgather_fn allgather, void* ctx, vg_comm** out);
int  vg_rccl_unique_id(void* out /* >= 128 bytes */, int64_t bytes);
int  vg_comm_rccl_create(int rank, int world, const void* unique_id, int64_t id_bytes, vg_comm** out);
void vg_comm_free(vg_comm* c);
/* 1 = built-in RCCL communicator, 0 = callback communicator; the number of ranks RCCL itself reports for the communicator
 * (ncclCommCount; -1 when it is not an RCCL communicator): a scaling record states which exchange it measured */
int  vg_comm_kind(const vg_comm* c);
int  vg_comm_rccl_ranks(const vg_comm* c);
int  vg_comm_rank(const vg_comm* c);
int  vg_comm_world(const vg_comm* c);
/* exchange self-test: every rank sends a pattern of `bytes` bytes and checks what it receives (no GPU needed
 * for a callback communicator over host memory) */
int  vg_comm_selftest(const vg_comm* c, int64_t bytes);
/* vg_kmer_shared over all ranks: rank r counts the k-mers of range r and keeps its partial (a, b, count)
 * list in HBM.  Sets below 2^32 bases (RANGE shards, two partition levels): every rank scans 1/world of the BASES and an
 * all-to-all (RCCL: grouped ncclSend / ncclRecv, one xGMI link per peer) hands each rank the kept-k-mer masks (one bit per
 * base) and level-1 counts of its own range -- no rank computes a k-mer it does not keep or forward.  Exchanged after that: the set sizes, the KEYS of the pairs a rank holds >= ceil(min_shared / world) of (a pair
 * that reaches min_shared in total has that many on some rank), and every rank's count for each pair of the union of
 * those keys; the counts are summed and the threshold is applied to the SUM.  Every rank receives the global result,
 * sorted by (a, b); device-to-device all-gathers only. */
int vg_kmer_shared_sharded(vg_genomes* g, int k, double fraction, uint32_t min_shared, const vg_comm* c,
                           int64_t* set_sizes, vg_pair_count** pairs, int64_t* n_pairs);
/* owner rank of every task: references cut into `world` contiguous id ranges with about equal task counts */
int vg_align_owner(const vg_task* tasks, int64_t n_tasks, int n_genomes, int world, int32_t* owner /* n_tasks */);
/* vg_lz_align over all ranks (reference-range partition): every rank receives all rows (and regions) */
int vg_lz_align_sharded(vg_genomes* g, const vg_task* tasks, int64_t n_tasks, const vg_lz_params* p, const vg_comm* c,
                        vg_pair_stat* stats, vg_region** regions, int64_t* n_regions);
/* rank `rank`'s share of the align tasks of the candidate pairs under the reference-range partition (a pure function of
 * the pairs; tasks in pair order, (q = b, r = a) before (q = a, r = b)); *tasks released with vg_free() */
int vg_align_pairs_share(const vg_genomes* g, const vg_pair_count* pairs, int64_t n_pairs, int world, int rank,
                         vg_task** tasks, int64_t* n_tasks);
/* (every rank passes the SAME pairs in the SAME order -- vg_kmer_shared_sharded returns them sorted; a checksum of the list
 * is compared across the ranks and a difference is an error on all of them) */
/* vg_align_tasks + vg_lz_align_sharded from the candidate pairs in one call: a rank derives its own tasks from the pairs
 * and starts its kernels at once, the canonical task list of the whole set is assembled beside them and only places the
 * gathered rows.  *tasks (canonical order, as vg_align_tasks) and *stats (one row per task) are released with vg_free(). */
int vg_lz_align_pairs_sharded(vg_genomes* g, const vg_pair_count* pairs, int64_t n_pairs, const vg_lz_params* p, const vg_comm* c,
                              vg_task** tasks, int64_t* n_tasks, vg_pair_stat** stats);
/* the two whole stages (vg_prefilter / vg_align) on `world` GPUs: every rank ingests the FASTA, rank 0 writes */
int vg_prefilter_sharded(const char* const* fasta_paths, int n_paths, const char* out_path,
                         const vg_prefilter_params* p, const vg_comm* c);
int vg_align_sharded(const char* const* fasta_paths, int n_paths, const char* out_path,
                     const vg_align_params* p, const vg_comm* c);

/* ------------------------------------------------------------------ synthetic input --- */
/* Workload generator of SURVEY.md 8(d) (bench / test input; no reference call site: the reference ships no
 * generator).  Same draws as vclust_amd/synth.py (splitmix64, counter based), multithreaded.  The family plan
 * -- family index, members, ancestor length per family -- comes from the caller.  *codes_out (bases 0..3 of
 * all genomes) and *offsets_out (n_genomes + 1) are released with vg_free(). */
int vg_synth_plan(const int64_t* fam_idx, const int32_t* members, const int32_t* lengths, int64_t n_fam,
                  uint64_t seed, double p_lo, double p_hi, int n_indels, int n_threads,
                  uint8_t** codes_out, int64_t** offsets_out, int64_t* n_genomes);

/* ------------------------------------------------------------------ measurement ------- */
/* per-kernel HIP-event timing on the library's stream (bench.py's roofline leg) */
void vg_profile_enable(int on);
void vg_profile_reset(void);
/* returns number of kernels recorded; fills up to cap entries */
typedef struct { char name[48]; double total_ms; int64_t launches; double bytes; } vg_kernel_time;
int  vg_profile_get(vg_kernel_time* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
