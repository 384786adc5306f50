/*
 * prefilter_oracle.c — CPU restatement of the Kmer-db prefilter (TEST INFRASTRUCTURE ONLY).
 *
 * Reference call sites: kmer-db build (vclust.py:953-964), all2all-sp/-parts with
 * `-sparse -min num-kmers:N -min ani-shorter:I [-sample-rows ani-shorter:M]`
 * (vclust.py:1005-1017), distance ani-shorter (vclust.py:1045-1055).  Native source
 * (3rd_party/kmer-db, .gitmodules:1-3) is absent; the arithmetic below reproduces
 * example/output/fltr.txt 13/13 to the last digit:
 *   K1  per genome: set of distinct canonical k-mers (min of k-mer and reverse complement,
 *       A<C<G<T), k-mers containing a non-ACGT symbol skipped;
 *   K2  shared(a,b) = |K_a ∩ K_b|, kept when shared >= min_kmers;
 *   K3  j = shared / min(|K_a|,|K_b|);  ani_shorter = 1 + ln(2j/(1+j))/k; kept when >= min_ident;
 *   K4  fltr.txt layout.
 * Unpinned by any reference fixture (our definition, identical in the HIP path):
 *   --kmers-fraction f<1 keeps canonical k-mer x iff mix64(x) < f*2^64;
 *   --max-seqs M keeps, per output row, the M entries of highest ani (ties: lower column).
 */
#include "vclust_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

uint64_t vo_mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

static void radix_sort_u64(uint64_t* a, uint64_t* tmp, int64_t n, int bits) {
    for (int sh = 0; sh < bits; sh += 8) {
        int64_t cnt[257]; memset(cnt, 0, sizeof cnt);
        for (int64_t i = 0; i < n; ++i) cnt[((a[i] >> sh) & 255) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) tmp[cnt[(a[i] >> sh) & 255]++] = a[i];
        uint64_t* t = a; a = tmp; tmp = t;
    }
    /* number of passes may be odd: caller passes bits as a multiple of 16 */
}

int64_t vo_kmer_set_f(const uint8_t* seq, int64_t len, int k, double fraction, uint64_t** out_sorted) {
    uint64_t mask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    uint64_t thr = fraction >= 1.0 ? UINT64_MAX : (uint64_t)ldexp(fraction, 64);
    uint64_t* a = (uint64_t*)malloc(sizeof(uint64_t) * (len > 0 ? len : 1));
    int64_t n = 0; uint64_t fw = 0, rc = 0; int valid = 0;
    for (int64_t i = 0; i < len; ++i) {
        uint8_t c = seq[i];
        if (c > 3) { valid = 0; fw = rc = 0; continue; }
        fw = ((fw << 2) | c) & mask;
        rc = (rc >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
        if (++valid >= k) {
            uint64_t cano = fw < rc ? fw : rc;
            if (fraction >= 1.0 || vo_mix64(cano) < thr) a[n++] = cano;
        }
    }
    uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * (n > 0 ? n : 1));
    int bits = 2 * k; bits = (bits + 15) / 16 * 16;
    radix_sort_u64(a, tmp, n, bits);
    free(tmp);
    int64_t u = 0;
    for (int64_t i = 0; i < n; ++i) if (i == 0 || a[i] != a[i - 1]) a[u++] = a[i];
    *out_sorted = a;
    return u;
}

int64_t vo_kmer_set(const uint8_t* seq, int64_t len, int k, uint64_t** out_sorted) {
    return vo_kmer_set_f(seq, len, k, 1.0, out_sorted);
}

int64_t vo_shared(const uint64_t* a, int64_t na, const uint64_t* b, int64_t nb) {
    int64_t i = 0, j = 0, s = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) ++i; else if (a[i] > b[j]) ++j; else { ++s; ++i; ++j; }
    }
    return s;
}

double vo_ani_shorter(int64_t shared, int64_t na, int64_t nb, int k) {
    int64_t mn = na < nb ? na : nb;
    if (mn <= 0 || shared <= 0) return 0.0;
    double j = (double)shared / (double)mn;
    return 1.0 + log(2.0 * j / (1.0 + j)) / (double)k;
}

/* ---- sparse all-pairs shared counts through an inverted index ---- */
typedef struct { uint64_t kmer; uint32_t g; } kg_t;

static void sort_kg(kg_t* a, kg_t* tmp, int64_t n, int bits) {
    for (int sh = 0; sh < bits; sh += 8) {
        int64_t cnt[257]; memset(cnt, 0, sizeof cnt);
        for (int64_t i = 0; i < n; ++i) cnt[((a[i].kmer >> sh) & 255) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) tmp[cnt[(a[i].kmer >> sh) & 255]++] = a[i];
        kg_t* t = a; a = tmp; tmp = t;
    }
}

typedef struct { uint64_t key; uint32_t cnt; } pc_t;

int vo_shared_all(const vo_genome_set* s, int k, double fraction,
                  int64_t* set_sizes, vo_pair_count** out_pairs, int64_t* n_pairs) {
    int n = s->n;
    int64_t total = 0;
    uint64_t** sets = (uint64_t**)calloc(n, sizeof(uint64_t*));
    #pragma omp parallel for schedule(dynamic) reduction(+:total)
    for (int g = 0; g < n; ++g) {
        set_sizes[g] = vo_kmer_set_f(s->g[g].seq, s->g[g].len, k, fraction, &sets[g]);
        total += set_sizes[g];
    }
    kg_t* a = (kg_t*)malloc(sizeof(kg_t) * (total > 0 ? total : 1));
    kg_t* tmp = (kg_t*)malloc(sizeof(kg_t) * (total > 0 ? total : 1));
    int64_t o = 0;
    for (int g = 0; g < n; ++g) {            /* genome order => stable sort keeps genomes ascending in a run */
        for (int64_t i = 0; i < set_sizes[g]; ++i) { a[o].kmer = sets[g][i]; a[o].g = (uint32_t)g; ++o; }
        free(sets[g]);
    }
    free(sets);
    int bits = (2 * k + 15) / 16 * 16;
    sort_kg(a, tmp, total, bits);
    free(tmp);
    /* pair counts in an open-addressing table */
    uint64_t cap = 1 << 16; pc_t* tab = (pc_t*)calloc(cap, sizeof(pc_t)); uint64_t used = 0;
    for (int64_t i = 0; i < total;) {
        int64_t j = i + 1;
        while (j < total && a[j].kmer == a[i].kmer) ++j;
        for (int64_t x = i; x < j; ++x)
            for (int64_t y = x + 1; y < j; ++y) {
                uint64_t key = ((uint64_t)a[y].g << 32) | a[x].g;      /* (row = larger id, col = smaller id) */
                if ((used + 1) * 2 > cap) {
                    uint64_t nc = cap * 2; pc_t* nt = (pc_t*)calloc(nc, sizeof(pc_t));
                    for (uint64_t t = 0; t < cap; ++t) if (tab[t].cnt) {
                        uint64_t h = vo_mix64(tab[t].key) & (nc - 1);
                        while (nt[h].cnt) h = (h + 1) & (nc - 1);
                        nt[h] = tab[t];
                    }
                    free(tab); tab = nt; cap = nc;
                }
                uint64_t h = vo_mix64(key) & (cap - 1);
                while (tab[h].cnt && tab[h].key != key) h = (h + 1) & (cap - 1);
                if (!tab[h].cnt) { tab[h].key = key; ++used; }
                tab[h].cnt++;
            }
        i = j;
    }
    free(a);
    vo_pair_count* pr = (vo_pair_count*)malloc(sizeof(vo_pair_count) * (used > 0 ? used : 1));
    int64_t m = 0;
    for (uint64_t t = 0; t < cap; ++t) if (tab[t].cnt) {
        pr[m].a = (uint32_t)(tab[t].key >> 32); pr[m].b = (uint32_t)tab[t].key; pr[m].shared = tab[t].cnt; ++m;
    }
    free(tab);
    *out_pairs = pr; *n_pairs = m;
    return 0;
}

/* ---- the same counts on all host threads: what bench.py times as cpu_baseline (kmer-db runs `-t T` end to end,
 * vclust.py:953-1055).  vo_shared_all above stays the checker (tests/test_oracle_golden.py holds the two equal):
 *   sets       per genome, OpenMP over genomes (as above);
 *   index      (k-mer, genome) records scattered into NP partitions by a hash of the k-mer -- per-thread counts, prefix
 *              sums, per-thread write cursors: no atomics --, every partition sorted by k-mer on its own thread;
 *   pair count runs of equal k-mers -> increments into per-THREAD open-addressing tables; the tables' (pair, count)
 *              entries are then dealt by a hash of the pair into one bucket per thread and every bucket is summed on
 *              its own thread (sort + reduce).
 * stage_s (may be NULL) receives the wall seconds of {sets, index, pair count}; *threads_used the OpenMP team size. ---- */
#include <omp.h>
#include <malloc.h>
#include <time.h>
static double now_s(void) { return omp_get_wtime(); }
static double cpu_s(void) { struct timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
void vo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
/* CPU seconds (all threads of the process) of the four stages of the last vo_shared_all_mt / vo_path_rows_mt call: CPU / wall
 * of a stage = the threads that were actually busy in it (bench.py prints it beside the wall seconds) */
static double g_stage_cpu[4];
void vo_last_stage_cpu(double* out4) { for (int i = 0; i < 4; ++i) out4[i] = g_stage_cpu[i]; }
void vo_note_stage_cpu(int stage, double seconds) { if (stage >= 0 && stage < 4) g_stage_cpu[stage] = seconds; }
static void* big_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }     /* (2 MiB pages by madvise were measured: their first touch is 4 x SLOWER in this VM) */
static int cmp_kg(const void* x, const void* y) {
    const kg_t* a = (const kg_t*)x; const kg_t* b = (const kg_t*)y;
    return a->kmer < b->kmer ? -1 : (a->kmer > b->kmer);
}
static int cmp_pc(const void* x, const void* y) {
    const pc_t* a = (const pc_t*)x; const pc_t* b = (const pc_t*)y;
    return a->key < b->key ? -1 : (a->key > b->key);
}
typedef struct { pc_t* tab; uint64_t cap, used; } ptab_t;
static void ptab_add(ptab_t* T, uint64_t key, uint32_t add) {
    if ((T->used + 1) * 2 > T->cap) {
        uint64_t nc = T->cap * 2; pc_t* nt = (pc_t*)calloc(nc, sizeof(pc_t));
        for (uint64_t t = 0; t < T->cap; ++t) if (T->tab[t].cnt) {
            uint64_t h = vo_mix64(T->tab[t].key) & (nc - 1);
            while (nt[h].cnt) h = (h + 1) & (nc - 1);
            nt[h] = T->tab[t];
        }
        free(T->tab); T->tab = nt; T->cap = nc;
    }
    uint64_t h = vo_mix64(key) & (T->cap - 1);
    while (T->tab[h].cnt && T->tab[h].key != key) h = (h + 1) & (T->cap - 1);
    if (!T->tab[h].cnt) { T->tab[h].key = key; ++T->used; }
    T->tab[h].cnt += add;
}

int vo_shared_all_mt(const vo_genome_set* s, int k, double fraction, int64_t* set_sizes, vo_pair_count** out_pairs,
                     int64_t* n_pairs, double* stage_s, int* threads_used) {
    const int n = s->n;
    int T = 1;
    #pragma omp parallel
    {
        #pragma omp single
        T = omp_get_num_threads();
    }
    if (threads_used) *threads_used = T;
    /* a k-mer set is 320 kB: above malloc's default mmap threshold every set would be an mmap + munmap of its own, 100 000 of
     * them from 256 threads through one address-space lock; from the per-thread arenas they are pointer bumps */
    static int tuned = 0;
    if (!tuned) { tuned = 1; mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30);      /* (32 MiB is the largest threshold glibc accepts) */ }
    double t0 = now_s(), c0 = cpu_s();
    uint64_t** sets = (uint64_t**)calloc(n > 0 ? n : 1, sizeof(uint64_t*));
    #pragma omp parallel for schedule(dynamic)
    for (int g = 0; g < n; ++g) set_sizes[g] = vo_kmer_set_f(s->g[g].seq, s->g[g].len, k, fraction, &sets[g]);
    double t1 = now_s(), c1 = cpu_s();
    /* index: partition by hash, then sort every partition */
    int PB = 6; while ((1 << PB) < 16 * T && PB < 14) ++PB;
    const int NP = 1 << PB;
    int64_t* cnt = (int64_t*)calloc((size_t)T * NP + 1, sizeof(int64_t));      /* [thread][partition] */
    #pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        int64_t* c = cnt + (size_t)t * NP;
        #pragma omp for schedule(static)
        for (int g = 0; g < n; ++g) for (int64_t i = 0; i < set_sizes[g]; ++i) c[vo_mix64(sets[g][i]) >> (64 - PB)]++;
    }
    int64_t* pstart = (int64_t*)calloc((size_t)NP + 1, sizeof(int64_t));
    int64_t total = 0;
    for (int p = 0; p < NP; ++p) {           /* partition-major, thread-minor: a partition is one contiguous stretch */
        pstart[p] = total;
        for (int t = 0; t < T; ++t) { int64_t c = cnt[(size_t)t * NP + p]; cnt[(size_t)t * NP + p] = total; total += c; }
    }
    pstart[NP] = total;
    /* the (k-mer, genome) records are GBs: the block is kept between calls (returning 13 GB to the kernel and faulting them in
     * again costs the port a second per call, serially, inside the timed stages -- the GPU side caches its blocks too) */
    static kg_t* a_keep = NULL; static size_t a_cap = 0;
    if ((size_t)total > a_cap) { free(a_keep); a_cap = (size_t)total + (size_t)total / 8 + 1; a_keep = (kg_t*)big_alloc(sizeof(kg_t) * a_cap); }
    kg_t* a = a_keep;
    #pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        int64_t* c = cnt + (size_t)t * NP;
        #pragma omp for schedule(static)                                        /* the same genomes as in the counting loop */
        for (int g = 0; g < n; ++g) {
            for (int64_t i = 0; i < set_sizes[g]; ++i) { const uint64_t x = sets[g][i]; kg_t* o = &a[c[vo_mix64(x) >> (64 - PB)]++]; o->kmer = x; o->g = (uint32_t)g; }
            free(sets[g]);
        }
    }
    free(sets); free(cnt);
    {
        /* every partition sorted by k-mer on its own thread (LSD radix over the key bytes, a scratch buffer per thread) */
        int64_t pmax = 0; for (int p = 0; p < NP; ++p) if (pstart[p + 1] - pstart[p] > pmax) pmax = pstart[p + 1] - pstart[p];
        const int bits = (2 * k + 7) / 8 * 8, passes = bits / 8;
        #pragma omp parallel num_threads(T)
        {
            kg_t* tmp = (kg_t*)big_alloc(sizeof(kg_t) * (size_t)(pmax > 0 ? pmax : 1));
            #pragma omp for schedule(dynamic, 1)
            for (int p = 0; p < NP; ++p) {
                const int64_t m = pstart[p + 1] - pstart[p];
                sort_kg(a + pstart[p], tmp, m, bits);
                if (passes & 1) memcpy(a + pstart[p], tmp, sizeof(kg_t) * (size_t)m);      /* an odd number of passes ends in the scratch buffer */
            }
            free(tmp);
        }
    }
    double t2 = now_s(), c2 = cpu_s();
    /* pair counts: per-thread tables over the partitions */
    ptab_t* tabs = (ptab_t*)calloc((size_t)T, sizeof(ptab_t));
    for (int t = 0; t < T; ++t) { tabs[t].cap = 1 << 12; tabs[t].tab = (pc_t*)calloc(tabs[t].cap, sizeof(pc_t)); }
    #pragma omp parallel num_threads(T)
    {
        ptab_t* tb = &tabs[omp_get_thread_num()];
        #pragma omp for schedule(dynamic, 1)
        for (int p = 0; p < NP; ++p) {
            const int64_t e = pstart[p + 1];
            for (int64_t i = pstart[p]; i < e;) {
                int64_t j = i + 1;
                while (j < e && a[j].kmer == a[i].kmer) ++j;
                for (int64_t x = i; x < j; ++x) for (int64_t y = x + 1; y < j; ++y) {
                    const uint32_t ga = a[x].g, gb = a[y].g;                 /* distinct: a set holds a k-mer once */
                    ptab_add(tb, ga > gb ? ((uint64_t)ga << 32) | gb : ((uint64_t)gb << 32) | ga, 1u);
                }
                i = j;
            }
        }
    }
    free(pstart);
    /* merge: entries dealt by a hash of the pair into T buckets, every bucket summed on its own thread */
    int64_t* bc = (int64_t*)calloc((size_t)T * T + 1, sizeof(int64_t));           /* [source thread][bucket] */
    #pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; ++t) for (uint64_t h = 0; h < tabs[t].cap; ++h) if (tabs[t].tab[h].cnt) bc[(size_t)t * T + (vo_mix64(tabs[t].tab[h].key ^ 0x9e3779b97f4a7c15ULL) % (uint64_t)T)]++;
    int64_t* bstart = (int64_t*)calloc((size_t)T + 1, sizeof(int64_t));
    int64_t ne = 0;
    for (int b = 0; b < T; ++b) { bstart[b] = ne; for (int t = 0; t < T; ++t) { int64_t c = bc[(size_t)t * T + b]; bc[(size_t)t * T + b] = ne; ne += c; } }
    bstart[T] = ne;
    pc_t* ent = (pc_t*)malloc(sizeof(pc_t) * (ne > 0 ? ne : 1));
    #pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; ++t) {
        for (uint64_t h = 0; h < tabs[t].cap; ++h) if (tabs[t].tab[h].cnt) ent[bc[(size_t)t * T + (vo_mix64(tabs[t].tab[h].key ^ 0x9e3779b97f4a7c15ULL) % (uint64_t)T)]++] = tabs[t].tab[h];
        free(tabs[t].tab);
    }
    free(tabs); free(bc);
    int64_t* bout = (int64_t*)calloc((size_t)T + 1, sizeof(int64_t));             /* distinct pairs of every bucket */
    #pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int b = 0; b < T; ++b) {
        pc_t* e = ent + bstart[b]; const int64_t m = bstart[b + 1] - bstart[b];
        qsort(e, (size_t)m, sizeof(pc_t), cmp_pc);
        int64_t u = 0;
        for (int64_t i = 0; i < m; ++i) { if (u > 0 && e[u - 1].key == e[i].key) e[u - 1].cnt += e[i].cnt; else e[u++] = e[i]; }
        bout[b] = u;
    }
    int64_t np = 0; for (int b = 0; b < T; ++b) np += bout[b];
    vo_pair_count* pr = (vo_pair_count*)malloc(sizeof(vo_pair_count) * (np > 0 ? np : 1));
    int64_t m = 0;
    for (int b = 0; b < T; ++b) for (int64_t i = 0; i < bout[b]; ++i) {
        const pc_t* e = &ent[bstart[b] + i];
        pr[m].a = (uint32_t)(e->key >> 32); pr[m].b = (uint32_t)e->key; pr[m].shared = e->cnt; ++m;
    }
    free(ent); free(bstart); free(bout);
    double t3 = now_s(), c3 = cpu_s();
    g_stage_cpu[0] = c1 - c0; g_stage_cpu[1] = c2 - c1; g_stage_cpu[2] = c3 - c2;
    if (stage_s) { stage_s[0] = t1 - t0; stage_s[1] = t2 - t1; stage_s[2] = t3 - t2; }
    *out_pairs = pr; *n_pairs = np;
    return 0;
}

static int cmp_pair(const void* x, const void* y) {
    const vo_pair_count* a = (const vo_pair_count*)x; const vo_pair_count* b = (const vo_pair_count*)y;
    if (a->a != b->a) return a->a < b->a ? -1 : 1;
    return a->b < b->b ? -1 : (a->b > b->b);
}

typedef struct { uint32_t col; double ani; } ent_t;
static int cmp_ent_ani(const void* x, const void* y) {
    const ent_t* a = (const ent_t*)x; const ent_t* b = (const ent_t*)y;
    if (a->ani != b->ani) return a->ani > b->ani ? -1 : 1;
    return a->col < b->col ? -1 : (a->col > b->col);
}
static int cmp_ent_col(const void* x, const void* y) {
    const ent_t* a = (const ent_t*)x; const ent_t* b = (const ent_t*)y;
    return a->col < b->col ? -1 : (a->col > b->col);
}

int vo_write_fltr(const vo_genome_set* s, int k, double fraction, int min_kmers, double min_ident,
                  int max_seqs, const int64_t* set_sizes, vo_pair_count* pairs, int64_t n_pairs,
                  const char* out_path) {
    FILE* f = fopen(out_path, "w");
    if (!f) return -1;
    qsort(pairs, n_pairs, sizeof(vo_pair_count), cmp_pair);
    fprintf(f, "kmer-length: %d fraction: %g ,", k, fraction);
    for (int g = 0; g < s->n; ++g) fprintf(f, "%s,", s->g[g].name);
    fprintf(f, "\n");
    int64_t p = 0;
    ent_t* row = (ent_t*)malloc(sizeof(ent_t) * (s->n > 0 ? s->n : 1));
    for (int g = 0; g < s->n; ++g) {
        fprintf(f, "%s,", s->g[g].name);
        int m = 0;
        while (p < n_pairs && pairs[p].a == (uint32_t)g) {
            if ((int64_t)pairs[p].shared >= min_kmers) {
                double ani = vo_ani_shorter(pairs[p].shared, set_sizes[g], set_sizes[pairs[p].b], k);
                if (ani >= min_ident) { row[m].col = pairs[p].b; row[m].ani = ani; ++m; }
            }
            ++p;
        }
        if (max_seqs > 0 && m > max_seqs) {
            qsort(row, m, sizeof(ent_t), cmp_ent_ani); m = max_seqs;
            qsort(row, m, sizeof(ent_t), cmp_ent_col);
        }
        for (int e = 0; e < m; ++e) fprintf(f, "%u:%.6f,", row[e].col + 1, row[e].ani);
        fprintf(f, "\n");
    }
    free(row);
    fclose(f);
    return 0;
}

int vo_prefilter(const vo_genome_set* s, int k, int min_kmers, double min_ident,
                 int n_threads, const char* out_path) {
    (void)n_threads;
    int64_t* sizes = (int64_t*)calloc(s->n > 0 ? s->n : 1, sizeof(int64_t));
    vo_pair_count* pairs; int64_t np;
    vo_shared_all(s, k, 1.0, sizes, &pairs, &np);
    int rc = vo_write_fltr(s, k, 1.0, min_kmers, min_ident, 0, sizes, pairs, np, out_path);
    free(pairs); free(sizes);
    return rc;
}
