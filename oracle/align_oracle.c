/*
 * align_oracle.c — CPU restatement of `lz-ani all2all` around the pairwise parse
 * (TEST INFRASTRUCTURE ONLY).  Call site: vclust.py:1142-1181.  Stages:
 *   L1  stable sort of genomes by length, descending; ids = rank; `<out - .tsv>.ids.tsv`
 *       with header id/seq_len/no_parts (example/output/ani.ids.tsv);
 *   L2  optional Kmer-db filter (fltr.txt layout, example/output/fltr.txt), pairs with
 *       value >= threshold are aligned in both directions;
 *   L5  per ordered pair: M = sum nt_match, A = sum alnlen, n = #regions;
 *   L6  ani = M/A, gani = M/Lq, qcov = A/Lq, rcov = A_rev/Lr, tani = (M+M_rev)/(Lq+Lr),
 *       len_ratio = min(L)/max(L);
 *   L7  ani.tsv rows: for a<b (ids): (q=b,r=a) then (q=a,r=b)  (example/output/ani.tsv);
 *   L8  alignment table (example/output/ani.aln.tsv), per pair sorted by alnlen desc, qstart asc.
 */
#include "vclust_oracle.h"
#include <time.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int idx; int64_t len; } ord_t;
static int cmp_ord(const void* x, const void* y) {
    const ord_t* a = (const ord_t*)x; const ord_t* b = (const ord_t*)y;
    if (a->len != b->len) return a->len > b->len ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx);
}

int vo_lz_pair_stat(const uint8_t* qry, int64_t qlen, const uint8_t* ref, int64_t rlen,
                    const vo_lz_params* p, uint32_t* n_match, uint32_t* aln_len, uint32_t* n_regions) {
    vo_lz_variant v; vo_lz_default_variant(&v);
    vo_ref_index* ix = vo_lz_build_index(ref, rlen, p, &v);
    vo_region* regs; int n;
    vo_lz_parse(ix, qry, qlen, p, &v, &regs, &n);
    uint32_t m = 0, a = 0;
    for (int k = 0; k < n; ++k) { m += regs[k].n_match; a += regs[k].qend - regs[k].qstart + 1; }
    free(regs); vo_lz_free_index(ix);
    *n_match = m; *aln_len = a; *n_regions = (uint32_t)n;
    return 0;
}

/* ---- filter file (SURVEY §8a K4 layout) ---- */
typedef struct { uint32_t a, b; } upair;   /* sorted ids, a < b */
static int cmp_upair(const void* x, const void* y) {
    const upair* p = (const upair*)x; const upair* q = (const upair*)y;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    return p->b < q->b ? -1 : (p->b > q->b);
}

static char* read_line(FILE* f) {
    size_t cap = 1 << 16, n = 0; char* buf = (char*)malloc(cap); int c;
    while ((c = fgetc(f)) != EOF) {
        if (n + 2 > cap) { cap *= 2; buf = (char*)realloc(buf, cap); }
        if (c == '\n') { buf[n] = 0; return buf; }
        buf[n++] = (char)c;
    }
    if (n == 0) { free(buf); return NULL; }
    buf[n] = 0; return buf;
}

typedef struct { const char* name; int id; } nm_t;
static int cmp_nm(const void* x, const void* y) { return strcmp(((const nm_t*)x)->name, ((const nm_t*)y)->name); }

static int load_filter(const char* path, double thr, const vo_genome_set* s, const int* rank_of_input,
                       upair** out, int64_t* n_out) {
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    nm_t* nm = (nm_t*)malloc(sizeof(nm_t) * s->n);
    for (int i = 0; i < s->n; ++i) { nm[i].name = s->g[i].name; nm[i].id = rank_of_input[i]; }
    qsort(nm, s->n, sizeof(nm_t), cmp_nm);
    char* hdr = read_line(f);
    if (!hdr) { fclose(f); free(nm); return -2; }
    /* header: "kmer-length: K fraction: F ,name1,name2,...," */
    int ncol = 0, cap = 1024; int* col_id = (int*)malloc(sizeof(int) * cap);
    char* p = strchr(hdr, ',');
    while (p && *(p + 1)) {
        char* e = strchr(p + 1, ','); if (!e) break;
        *e = 0;
        nm_t key = { p + 1, 0 };
        nm_t* hit = (nm_t*)bsearch(&key, nm, s->n, sizeof(nm_t), cmp_nm);
        if (ncol == cap) { cap *= 2; col_id = (int*)realloc(col_id, sizeof(int) * cap); }
        col_id[ncol++] = hit ? hit->id : -1;
        p = e;
    }
    free(hdr);
    int64_t n = 0, pc = 1024; upair* pr = (upair*)malloc(sizeof(upair) * pc);
    char* line;
    while ((line = read_line(f)) != NULL) {
        char* e = strchr(line, ',');
        if (!e) { free(line); continue; }
        *e = 0;
        nm_t key = { line, 0 };
        nm_t* hit = (nm_t*)bsearch(&key, nm, s->n, sizeof(nm_t), cmp_nm);
        int row = hit ? hit->id : -1;
        char* q = e + 1;
        while (*q) {
            char* colon = strchr(q, ':'); if (!colon) break;
            char* comma = strchr(colon, ','); if (comma) *comma = 0;
            int ci = atoi(q) - 1; double val = atof(colon + 1);
            if (row >= 0 && ci >= 0 && ci < ncol && col_id[ci] >= 0 && val >= thr && col_id[ci] != row) {
                if (n == pc) { pc *= 2; pr = (upair*)realloc(pr, sizeof(upair) * pc); }
                uint32_t a = (uint32_t)row, b = (uint32_t)col_id[ci];
                pr[n].a = a < b ? a : b; pr[n].b = a < b ? b : a; ++n;
            }
            if (!comma) break;
            q = comma + 1;
        }
        free(line);
    }
    fclose(f); free(nm); free(col_id);
    qsort(pr, n, sizeof(upair), cmp_upair);
    int64_t u = 0;
    for (int64_t i = 0; i < n; ++i) if (i == 0 || cmp_upair(&pr[i], &pr[i - 1])) pr[u++] = pr[i];
    *out = pr; *n_out = u;
    return 0;
}

typedef struct { vo_region* r; int n; } reglist;
static int cmp_region_out(const void* x, const void* y) {
    const vo_region* a = (const vo_region*)x; const vo_region* b = (const vo_region*)y;
    int la = a->qend - a->qstart, lb = b->qend - b->qstart;
    if (la != lb) return la > lb ? -1 : 1;
    return a->qstart < b->qstart ? -1 : (a->qstart > b->qstart);
}

int vo_align(const vo_genome_set* s, const char* out_path, const vo_align_params* p) {
    int n = s->n;
    ord_t* ord = (ord_t*)malloc(sizeof(ord_t) * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { ord[i].idx = i; ord[i].len = s->g[i].len; }
    qsort(ord, n, sizeof(ord_t), cmp_ord);
    int* rank = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) rank[ord[i].idx] = i;
#define G(id) (s->g[ord[id].idx])

    /* ids file */
    {
        size_t L = strlen(out_path);
        char* ids = (char*)malloc(L + 16);
        strcpy(ids, out_path);
        if (L > 4 && !strcmp(out_path + L - 4, ".tsv")) ids[L - 4] = 0;
        strcat(ids, ".ids.tsv");
        FILE* f = fopen(ids, "w"); free(ids);
        if (!f) return -1;
        fprintf(f, "id\tseq_len\tno_parts\n");
        for (int i = 0; i < n; ++i) fprintf(f, "%s\t%lld\t%d\n", G(i).name, (long long)G(i).len, G(i).n_parts);
        fclose(f);
    }

    /* candidate pairs */
    upair* pairs = NULL; int64_t np = 0;
    if (p->filter_path) {
        if (load_filter(p->filter_path, p->filter_threshold, s, rank, &pairs, &np)) return -2;
    } else {
        np = (int64_t)n * (n - 1) / 2;
        pairs = (upair*)malloc(sizeof(upair) * (np > 0 ? np : 1));
        int64_t o = 0;
        for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) { pairs[o].a = a; pairs[o].b = b; ++o; }
    }

    /* ordered-pair tasks grouped by reference */
    int64_t nt = 2 * np;
    vo_pair_stat* st = (vo_pair_stat*)calloc(nt > 0 ? nt : 1, sizeof(vo_pair_stat));
    reglist* rl = p->out_aln_path ? (reglist*)calloc(nt > 0 ? nt : 1, sizeof(reglist)) : NULL;
    for (int64_t i = 0; i < np; ++i) {
        st[2 * i].q = pairs[i].b; st[2 * i].r = pairs[i].a;          /* row (q=b, r=a) */
        st[2 * i + 1].q = pairs[i].a; st[2 * i + 1].r = pairs[i].b;  /* row (q=a, r=b) */
    }
    /* bucket tasks by reference so that every index is built once */
    int64_t* roff = (int64_t*)calloc(n + 1, sizeof(int64_t));
    for (int64_t t = 0; t < nt; ++t) roff[st[t].r + 1]++;
    for (int r = 0; r < n; ++r) roff[r + 1] += roff[r];
    int64_t* rtask = (int64_t*)malloc(sizeof(int64_t) * (nt > 0 ? nt : 1));
    int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
    memcpy(cur, roff, sizeof(int64_t) * (n + 1));
    for (int64_t t = 0; t < nt; ++t) rtask[cur[st[t].r]++] = t;
    free(cur);
    vo_lz_variant var; vo_lz_default_variant(&var);
    vo_ref_index** idx_of = p->out_aln_path ? (vo_ref_index**)calloc(n > 0 ? n : 1, sizeof(void*)) : NULL;

    #pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < n; ++r) {
        if (roff[r] == roff[r + 1]) continue;
        vo_ref_index* ix = vo_lz_build_index(G(r).seq, G(r).len, &p->lz, &var);
        for (int64_t k = roff[r]; k < roff[r + 1]; ++k) {
            int64_t t = rtask[k];
            vo_region* regs; int nr;
            vo_lz_parse(ix, G(st[t].q).seq, G(st[t].q).len, &p->lz, &var, &regs, &nr);
            uint32_t m = 0, a = 0;
            for (int x = 0; x < nr; ++x) { m += regs[x].n_match; a += regs[x].qend - regs[x].qstart + 1; }
            st[t].n_match = m; st[t].aln_len = a; st[t].n_regions = (uint32_t)nr;
            if (rl) { rl[t].r = regs; rl[t].n = nr; } else free(regs);
        }
        if (idx_of) idx_of[r] = ix; else vo_lz_free_index(ix);
    }

    /* ani.tsv */
    FILE* f = fopen(out_path, "w");
    if (!f) return -1;
    for (int c = 0; c < p->n_out_columns; ++c) fprintf(f, "%s%s", c ? "\t" : "", p->out_columns[c]);
    fprintf(f, "\n");
    for (int64_t t = 0; t < nt; ++t) {
        const vo_pair_stat* x = &st[t]; const vo_pair_stat* rev = &st[t ^ 1];
        int64_t lq = G(x->q).len, lr = G(x->r).len;
        double ani = x->aln_len ? (double)x->n_match / (double)x->aln_len : 0.0;
        double gani = lq ? (double)x->n_match / (double)lq : 0.0;
        double qcov = lq ? (double)x->aln_len / (double)lq : 0.0;
        double rcov = lr ? (double)rev->aln_len / (double)lr : 0.0;
        double tani = (lq + lr) ? (double)(x->n_match + rev->n_match) / (double)(lq + lr) : 0.0;
        if (p->out_tani > 0 && tani < p->out_tani) continue;
        if (p->out_gani > 0 && gani < p->out_gani) continue;
        if (p->out_ani > 0 && ani < p->out_ani) continue;
        if (p->out_qcov > 0 && qcov < p->out_qcov) continue;
        if (p->out_rcov > 0 && rcov < p->out_rcov) continue;
        char buf[64];
        for (int c = 0; c < p->n_out_columns; ++c) {
            const char* col = p->out_columns[c];
            if (c) fputc('\t', f);
            if (!strcmp(col, "qidx")) fprintf(f, "%u", x->q);
            else if (!strcmp(col, "ridx")) fprintf(f, "%u", x->r);
            else if (!strcmp(col, "query")) fputs(G(x->q).name, f);
            else if (!strcmp(col, "reference")) fputs(G(x->r).name, f);
            else if (!strcmp(col, "tani")) { vo_fmt_num(tani, buf); fputs(buf, f); }
            else if (!strcmp(col, "gani")) { vo_fmt_num(gani, buf); fputs(buf, f); }
            else if (!strcmp(col, "ani")) { vo_fmt_num(ani, buf); fputs(buf, f); }
            else if (!strcmp(col, "qcov")) { vo_fmt_num(qcov, buf); fputs(buf, f); }
            else if (!strcmp(col, "rcov")) { vo_fmt_num(rcov, buf); fputs(buf, f); }
            else if (!strcmp(col, "num_alns")) fprintf(f, "%u", x->n_regions);
            else if (!strcmp(col, "len_ratio")) { vo_fmt_len_ratio(lq, lr, buf); fputs(buf, f); }
            else if (!strcmp(col, "qlen")) fprintf(f, "%lld", (long long)lq);
            else if (!strcmp(col, "rlen")) fprintf(f, "%lld", (long long)lr);
            else if (!strcmp(col, "nt_match")) fprintf(f, "%u", x->n_match);
            else if (!strcmp(col, "nt_mismatch")) fprintf(f, "%u", x->aln_len - x->n_match);
        }
        fputc('\n', f);
    }
    fclose(f);

    if (p->out_aln_path) {
        FILE* fa = fopen(p->out_aln_path, "w");
        if (!fa) return -1;
        fprintf(fa, "query\treference\tpident\talnlen\tqstart\tqend\trstart\trend\tnt_match\tnt_mismatch\n");
        for (int64_t t = 0; t < nt; ++t) {
            if (rl[t].n > 0) qsort(rl[t].r, rl[t].n, sizeof(vo_region), cmp_region_out);
            vo_ref_index* ix = idx_of[st[t].r];
            for (int k = 0; k < rl[t].n; ++k) {
                vo_region* g = &rl[t].r[k];
                int alnlen = g->qend - g->qstart + 1; char buf[64];
                vo_fmt_num(100.0 * g->n_match / alnlen, buf);
                fprintf(fa, "%s\t%s\t%s\t%d\t%d\t%d\t%lld\t%lld\t%d\t%d\n", G(st[t].q).name, G(st[t].r).name,
                        buf, alnlen, g->qstart + 1, g->qend + 1,
                        (long long)vo_rr_to_fwd1(ix, g->rstart), (long long)vo_rr_to_fwd1s(ix, g->rend, vo_rr_is_rev(ix, g->rstart)),
                        g->n_match, g->n_mismatch);
            }
            free(rl[t].r);
        }
        fclose(fa);
        for (int r = 0; r < n; ++r) vo_lz_free_index(idx_of[r]);
        free(idx_of); free(rl);
    }
#undef G
    free(st); free(roff); free(rtask); free(pairs); free(ord); free(rank);
    return 0;
}


/* ---- the whole path in memory: prefilter counts -> thresholds -> LZ parse of both directions of every kept
 * pair (genomes in memory -> integer rows in memory; no files).  Same scope as the HIP path's device-resident
 * step, so it is what bench.py times as cpu_baseline and what the full-set parity tests compare with.
 * Ids are input-order genome indices; rows come in pair order, (q=a,r=b) then (q=b,r=a) with a > b. ---- */
static int path_rows_impl(const vo_genome_set* s, int k, int min_kmers, double min_ident, const vo_lz_params* lz,
                          vo_pair_stat** rows_out, int64_t* n_rows, int mt, double* stage_s, int* threads_used);
int vo_path_rows(const vo_genome_set* s, int k, int min_kmers, double min_ident, const vo_lz_params* lz,
                 vo_pair_stat** rows_out, int64_t* n_rows) {
    return path_rows_impl(s, k, min_kmers, min_ident, lz, rows_out, n_rows, 0, NULL, NULL);
}
/* the same with every stage on all host threads (vo_shared_all_mt; the LZ loop is OpenMP over references in both):
 * bench.py's cpu_baseline.  stage_s[4] = wall seconds of {sets, index, pair count, LZ}; rows in an order of its own
 * (the pair order of the merge): callers compare as sets. */
int vo_path_rows_mt(const vo_genome_set* s, int k, int min_kmers, double min_ident, const vo_lz_params* lz,
                    vo_pair_stat** rows_out, int64_t* n_rows, double* stage_s, int* threads_used) {
    return path_rows_impl(s, k, min_kmers, min_ident, lz, rows_out, n_rows, 1, stage_s, threads_used);
}
static int path_rows_impl(const vo_genome_set* s, int k, int min_kmers, double min_ident, const vo_lz_params* lz,
                          vo_pair_stat** rows_out, int64_t* n_rows, int mt, double* stage_s, int* threads_used) {
    int n = s->n;
    int64_t* sizes = (int64_t*)calloc(n > 0 ? n : 1, sizeof(int64_t));
    vo_pair_count* pairs; int64_t np;
    if (mt) vo_shared_all_mt(s, k, 1.0, sizes, &pairs, &np, stage_s, threads_used);
    else vo_shared_all(s, k, 1.0, sizes, &pairs, &np);
    const double t_lz0 = omp_get_wtime();
    struct timespec c_lz0; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &c_lz0);
    int64_t kept = 0;
    for (int64_t i = 0; i < np; ++i) {
        if ((int64_t)pairs[i].shared < min_kmers) continue;
        if (vo_ani_shorter(pairs[i].shared, sizes[pairs[i].a], sizes[pairs[i].b], k) < min_ident) continue;
        pairs[kept++] = pairs[i];
    }
    int64_t nt = 2 * kept;
    vo_pair_stat* st = (vo_pair_stat*)calloc(nt > 0 ? nt : 1, sizeof(vo_pair_stat));
    for (int64_t i = 0; i < kept; ++i) {
        st[2 * i].q = pairs[i].a; st[2 * i].r = pairs[i].b;
        st[2 * i + 1].q = pairs[i].b; st[2 * i + 1].r = pairs[i].a;
    }
    free(pairs); free(sizes);
    int64_t* roff = (int64_t*)calloc(n + 1, sizeof(int64_t));
    for (int64_t t = 0; t < nt; ++t) roff[st[t].r + 1]++;
    for (int r = 0; r < n; ++r) roff[r + 1] += roff[r];
    int64_t* rtask = (int64_t*)malloc(sizeof(int64_t) * (nt > 0 ? nt : 1));
    int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
    memcpy(cur, roff, sizeof(int64_t) * (n + 1));
    for (int64_t t = 0; t < nt; ++t) rtask[cur[st[t].r]++] = t;
    free(cur);
    vo_lz_variant var; vo_lz_default_variant(&var);
    #pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < n; ++r) {
        if (roff[r] == roff[r + 1]) continue;
        vo_ref_index* ix = vo_lz_build_index(s->g[r].seq, s->g[r].len, lz, &var);
        for (int64_t x = roff[r]; x < roff[r + 1]; ++x) {
            int64_t t = rtask[x];
            vo_region* regs; int nr;
            vo_lz_parse(ix, s->g[st[t].q].seq, s->g[st[t].q].len, lz, &var, &regs, &nr);
            uint32_t m = 0, a = 0;
            for (int y = 0; y < nr; ++y) { m += regs[y].n_match; a += regs[y].qend - regs[y].qstart + 1; }
            st[t].n_match = m; st[t].aln_len = a; st[t].n_regions = (uint32_t)nr;
            free(regs);
        }
        vo_lz_free_index(ix);
    }
    free(roff); free(rtask);
    if (mt && stage_s) stage_s[3] = omp_get_wtime() - t_lz0;
    { struct timespec c1; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &c1); vo_note_stage_cpu(3, (double)(c1.tv_sec - c_lz0.tv_sec) + 1e-9 * (double)(c1.tv_nsec - c_lz0.tv_nsec)); }
    *rows_out = st; *n_rows = nt;
    return 0;
}
