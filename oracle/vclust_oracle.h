/*
 * vclust_oracle.h — CPU restatement (TEST INFRASTRUCTURE ONLY) of the
 * prefilter -> align hot path of refresh-bio/vclust.
 *
 * This oracle is the checker for the HIP product path.  It must never be
 * imported, linked or executed by anything except tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke().
 *
 * The native sources of the reference path (3rd_party/kmer-db, 3rd_party/lz-ani;
 * reference .gitmodules:1-6) are ABSENT from /root/reference, and their pinned
 * versions are not recoverable.  The restatement therefore
 * follows the published algorithm of those tools as far as it can be
 * reconstructed, and is pinned on the reference's golden vectors
 * example/output/{fltr.txt,ani.tsv,ani.ids.tsv,ani.aln.tsv}; the achieved
 * agreement is reported by tests/test_oracle_golden.py and in DESIGN.md.
 */
#ifndef VCLUST_ORACLE_H
#define VCLUST_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- symbols ---------- */
enum { VO_A = 0, VO_C = 1, VO_G = 2, VO_T = 3, VO_NREF = 4, VO_NQRY = 5 };

/* ---------- genome set ---------- */
typedef struct {
    char*    name;      /* first header token */
    uint8_t* seq;       /* codes 0..3, 4 = non-ACGT */
    int64_t  len;
    int32_t  n_parts;   /* contigs joined into this genome (directory mode) */
} vo_genome;

typedef struct {
    vo_genome* g;
    int32_t    n;
    int32_t    cap;
} vo_genome_set;

/* Read FASTA / FASTA.gz.  multisample != 0: one genome per record (vclust.py:962-963,
 * 1159-1160); else one genome per file (records concatenated as parts). */
int  vo_read_fasta(const char* path, int multisample, vo_genome_set* out);
void vo_free_genomes(vo_genome_set* s);

/* ---------- prefilter (kmer-db build / all2all-sp / distance; vclust.py:953-1055) ---------- */
/* distinct canonical k-mers of one genome, sorted ascending; returns count */
int64_t vo_kmer_set(const uint8_t* seq, int64_t len, int k, uint64_t** out_sorted);
/* |A ∩ B| for sorted distinct arrays */
int64_t vo_shared(const uint64_t* a, int64_t na, const uint64_t* b, int64_t nb);
/* ani-shorter transform */
double  vo_ani_shorter(int64_t shared, int64_t na, int64_t nb, int k);
/* whole prefilter: writes fltr.txt. */
int vo_prefilter(const vo_genome_set* s, int k, int min_kmers, double min_ident,
                 int n_threads, const char* out_path);
uint64_t vo_mix64(uint64_t x);
int64_t vo_kmer_set_f(const uint8_t* seq, int64_t len, int k, double fraction, uint64_t** out_sorted);
typedef struct { uint32_t a, b, shared; } vo_pair_count;   /* a > b (row, column of the lower triangle) */
/* sparse all-pairs shared k-mer counts (only pairs with shared > 0) and per-genome set sizes */
int vo_shared_all(const vo_genome_set* s, int k, double fraction,
                  int64_t* set_sizes, vo_pair_count** out_pairs, int64_t* n_pairs);
/* the same counts with every stage on all OpenMP threads (bench.py's cpu_baseline; vo_shared_all is the checker);
 * stage_s[3] = seconds of {sets, index, pair count} */
void vo_set_threads(int n);      /* OpenMP team size of the calls that follow */
void vo_last_stage_cpu(double* out4);           /* CPU seconds (all threads) of {sets, index, pair count, lz} of the last *_mt call */
void vo_note_stage_cpu(int stage, double seconds);
int vo_shared_all_mt(const vo_genome_set* s, int k, double fraction, int64_t* set_sizes, vo_pair_count** out_pairs,
                     int64_t* n_pairs, double* stage_s, int* threads_used);
int vo_write_fltr(const vo_genome_set* s, int k, double fraction, int min_kmers, double min_ident,
                  int max_seqs, const int64_t* set_sizes, vo_pair_count* pairs, int64_t n_pairs,
                  const char* out_path);

/* ---------- LZ-ANI parse (lz-ani all2all; vclust.py:1142-1181) ---------- */
typedef struct {
    int mal, msl, mrd, mqd, reg, aw, am, ar;
} vo_lz_params;

/* Knobs for the rules the goldens had to decide.  The defaults (vo_lz_default_variant) are the
 * fitted rules R1-R8 documented in lz_oracle.c; the alternatives are kept so that the fit can
 * be re-run with oracle/lzfit when new evidence (e.g. the upstream binary) appears. */
typedef struct {
    int sep_len;                 /* N symbols between forward strand and reverse complement */
    int anchor_while_predicting; /* 0 never, 1 after a failed seed search, 2 before the seed search */
    int bwd_bound_kept;          /* left extension bounded by the last kept region (1) or by the literal run (0) */
    int bwd_exact_first;         /* maximal exact left extension before the approximate one */
    int seed_window;             /* 0: |p-pred|<=mrd  1: pred0-back<=p<=pred0+fwd  2: pred0-back<=p, p-pred<=fwd */
    int seed_back, seed_fwd;     /* window bounds (seed_fwd<0: mrd-1) */
    int seed_choice;             /* 0 longest,first  1 closest  2 first  3 longest,closest */
    int lit_reset_ge;            /* drop prediction at lit>=mqd (1) or lit>mqd (0) */
    int gap_mode;                /* 0 old diagonal, 1 new diagonal, 2 best split, 3 none */
    int fwd_after_close;
    int loop_le;                 /* main loop bound i+mal<=n (1) or i+mal<n (0) */
    int anchor_tie;              /* 0 smallest pos, 1 largest pos */
    int reg_on_span;             /* region kept on query span (1) or on matches (0) */
    int rend_mode;               /* 0 true end, 1 max with chained ends, 2 also max with the gap end on the old diagonal */
    int trace;
    int anchor_margin;           /* anchor_while_predicting 3: a far anchor beats a seed when longer by more than this */
    int weak_seed_ratio;         /* the margin drops by one when lit > ratio * seed length (0 = never) */
} vo_lz_variant;

typedef struct {
    int32_t qstart, qend;   /* 0-based inclusive, query coordinates */
    int32_t rstart, rend;   /* 0-based inclusive in the canonical space fwd | N | rc */
    int32_t n_match;
    int32_t n_mismatch;     /* = qend-qstart+1-n_match */
} vo_region;

typedef struct vo_ref_index vo_ref_index;

void vo_lz_default_variant(vo_lz_variant* v);
vo_ref_index* vo_lz_build_index(const uint8_t* ref, int64_t len,
                                const vo_lz_params* p, const vo_lz_variant* v);
void vo_lz_free_index(vo_ref_index* idx);
/* parse query against index; regions (already filtered by reg) written to *out (malloc'd). */
int  vo_lz_parse(const vo_ref_index* idx, const uint8_t* qry, int64_t qlen,
                 const vo_lz_params* p, const vo_lz_variant* v,
                 vo_region** out, int* n_out);
/* regions are reported in the canonical space fwd | N | rc; 1-based forward coordinates of a position
 * on a given strand (a region's strand is that of its rstart; rend may lie past the strand's end) */
int64_t vo_rr_to_fwd1(const vo_ref_index* idx, int64_t rr_pos);
int64_t vo_rr_to_fwd1s(const vo_ref_index* idx, int64_t rr_pos, int rev);
int     vo_rr_is_rev(const vo_ref_index* idx, int64_t rr_pos);

/* ---------- whole align stage (lz-ani all2all) ---------- */
typedef struct {
    vo_lz_params lz;
    double out_tani, out_gani, out_ani, out_qcov, out_rcov;  /* 0 = off (vclust.py:1170-1176) */
    const char* filter_path; double filter_threshold;        /* NULL = all-vs-all (vclust.py:1165-1166) */
    const char* out_aln_path;                                /* NULL = none (vclust.py:1167-1168) */
    const char* const* out_columns; int n_out_columns;       /* ALIGN_OUTFMT[fmt] (vclust.py:38-47) */
    int n_threads;
} vo_align_params;
/* per ordered pair integer result (L5) */
typedef struct { uint32_t q, r; uint32_t n_match, aln_len, n_regions; } vo_pair_stat;
/* sorts genomes by length (stable, descending), writes out_path, <out_path minus .tsv>.ids.tsv and
 * optionally the alignment table */
int vo_align(const vo_genome_set* s, const char* out_path, const vo_align_params* p);
/* integer statistics of one ordered pair with the default variant (for parity tests) */
int vo_lz_pair_stat(const uint8_t* qry, int64_t qlen, const uint8_t* ref, int64_t rlen,
                    const vo_lz_params* p, uint32_t* n_match, uint32_t* aln_len, uint32_t* n_regions);

/* the whole path in memory (prefilter counts -> thresholds -> both directions of every kept pair), ids in
 * input order, rows (q=a,r=b),(q=b,r=a) per kept pair a > b; *rows_out is malloc'd.  OpenMP over references. */
int vo_path_rows(const vo_genome_set* s, int k, int min_kmers, double min_ident, const vo_lz_params* lz,
                 vo_pair_stat** rows_out, int64_t* n_rows);
int vo_path_rows_mt(const vo_genome_set* s, int k, int min_kmers, double min_ident, const vo_lz_params* lz,
                    vo_pair_stat** rows_out, int64_t* n_rows, double* stage_s /* 4: sets, index, pair count, LZ */, int* threads_used);

/* ---------- formatting ---------- */
/* writes the LZ-ANI style number into buf (>= 32 bytes), returns length */
int vo_fmt_num(double x, char* buf);
int vo_fmt_len_ratio(int64_t a, int64_t b, char* buf);

#ifdef __cplusplus
}
#endif
#endif
