/*
 * oracle_cli.c — command-line face of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Mirrors the two native tool invocations of the reference front-end so that whole-file
 * outputs (fltr.txt, ani.tsv, ani.ids.tsv, ani.aln.tsv) can be byte-compared with the HIP
 * product path and timed as bench.py's cpu_baseline:
 *   oracle_cli prefilter -o fltr.txt [-k 25] [--min-kmers 20] [--min-ident 0.7]
 *                        [--kmers-fraction 1] [--max-seqs 0] [-t N] INPUT...
 *   oracle_cli align -o ani.tsv [--filter fltr.txt THR] [--out-aln F] [--outfmt standard|lite|complete]
 *                    [--mal ..] ... [--out-ani v] ... [-t N] INPUT...
 * One INPUT = multi-FASTA (one genome per record); several INPUTs = one genome per file
 * (vclust.py:687-702).
 */
#include "vclust_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const char* FIELDS[] = { "qidx", "ridx", "query", "reference", "tani", "gani", "ani", "qcov",
                                "rcov", "num_alns", "len_ratio", "qlen", "rlen", "nt_match", "nt_mismatch" };

static int read_inputs(int n, char** paths, vo_genome_set* gs) {
    int multi = (n == 1);
    for (int i = 0; i < n; ++i)
        if (vo_read_fasta(paths[i], multi, gs)) { fprintf(stderr, "oracle: cannot read %s\n", paths[i]); return -1; }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: oracle_cli prefilter|align ...\n"); return 0; }
    const char* cmd = argv[1];
    const char* out = NULL; int threads = 0;
    int k = 25, min_kmers = 20, max_seqs = 0; double min_ident = 0.7, fraction = 1.0;
    vo_align_params ap; memset(&ap, 0, sizeof ap);
    ap.lz = (vo_lz_params){ 11, 7, 40, 40, 35, 15, 7, 3 };
    const char* outfmt = "standard";
    char** inputs = (char**)malloc(sizeof(char*) * argc); int ni = 0;
    for (int a = 2; a < argc; ++a) {
        const char* s = argv[a];
#define NEXT (a + 1 < argc ? argv[++a] : "")
        if (!strcmp(s, "-o")) out = NEXT;
        else if (!strcmp(s, "-t")) threads = atoi(NEXT);
        else if (!strcmp(s, "-k")) k = atoi(NEXT);
        else if (!strcmp(s, "--min-kmers")) min_kmers = atoi(NEXT);
        else if (!strcmp(s, "--min-ident")) min_ident = atof(NEXT);
        else if (!strcmp(s, "--kmers-fraction")) fraction = atof(NEXT);
        else if (!strcmp(s, "--max-seqs")) max_seqs = atoi(NEXT);
        else if (!strcmp(s, "--filter")) { ap.filter_path = NEXT; ap.filter_threshold = atof(NEXT); }
        else if (!strcmp(s, "--out-aln")) ap.out_aln_path = NEXT;
        else if (!strcmp(s, "--outfmt")) outfmt = NEXT;
        else if (!strcmp(s, "--mal")) ap.lz.mal = atoi(NEXT);
        else if (!strcmp(s, "--msl")) ap.lz.msl = atoi(NEXT);
        else if (!strcmp(s, "--mrd")) ap.lz.mrd = atoi(NEXT);
        else if (!strcmp(s, "--mqd")) ap.lz.mqd = atoi(NEXT);
        else if (!strcmp(s, "--reg")) ap.lz.reg = atoi(NEXT);
        else if (!strcmp(s, "--aw")) ap.lz.aw = atoi(NEXT);
        else if (!strcmp(s, "--am")) ap.lz.am = atoi(NEXT);
        else if (!strcmp(s, "--ar")) ap.lz.ar = atoi(NEXT);
        else if (!strcmp(s, "--out-tani")) ap.out_tani = atof(NEXT);
        else if (!strcmp(s, "--out-gani")) ap.out_gani = atof(NEXT);
        else if (!strcmp(s, "--out-ani")) ap.out_ani = atof(NEXT);
        else if (!strcmp(s, "--out-qcov")) ap.out_qcov = atof(NEXT);
        else if (!strcmp(s, "--out-rcov")) ap.out_rcov = atof(NEXT);
        else inputs[ni++] = argv[a];
#undef NEXT
    }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    if (!out || ni == 0) { fprintf(stderr, "oracle_cli: need -o and inputs\n"); return 2; }
    vo_genome_set gs = {0};
    if (read_inputs(ni, inputs, &gs)) return 1;
    int rc = 0;
    if (!strcmp(cmd, "prefilter")) {
        int64_t* sizes = (int64_t*)calloc(gs.n > 0 ? gs.n : 1, sizeof(int64_t));
        vo_pair_count* pairs; int64_t np;
        vo_shared_all(&gs, k, fraction, sizes, &pairs, &np);
        rc = vo_write_fltr(&gs, k, fraction, min_kmers, min_ident, max_seqs, sizes, pairs, np, out);
        free(pairs); free(sizes);
    } else if (!strcmp(cmd, "align")) {
        int nc = 11; const char* cols[15];
        if (!strcmp(outfmt, "lite")) { nc = 0; cols[nc++] = FIELDS[0]; cols[nc++] = FIELDS[1]; for (int c = 4; c < 11; ++c) cols[nc++] = FIELDS[c]; }
        else { nc = !strcmp(outfmt, "complete") ? 15 : 11; for (int c = 0; c < nc; ++c) cols[c] = FIELDS[c]; }
        ap.out_columns = cols; ap.n_out_columns = nc; ap.n_threads = threads;
        rc = vo_align(&gs, out, &ap);
    } else { fprintf(stderr, "oracle_cli: unknown command %s\n", cmd); rc = 2; }
    vo_free_genomes(&gs);
    free(inputs);
    return rc ? 1 : 0;
}
