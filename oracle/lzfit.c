/*
 * lzfit.c — development harness: run the oracle LZ parse for every ordered pair of a
 * multi-FASTA file and print the regions in the reference's --out-alignment layout
 * (example/output/ani.aln.tsv) so they can be scored against the golden regions.
 * usage: lzfit <multifasta> [key=value ...]   (variant knobs of vo_lz_variant)
 */
#include "vclust_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: lzfit fasta [knob=value...]\n"); return 2; }
    vo_genome_set gs = {0};
    if (vo_read_fasta(argv[1], 1, &gs)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    vo_lz_params p = {11, 7, 40, 40, 35, 15, 7, 3};
    vo_lz_variant v; vo_lz_default_variant(&v);
    const char* only_q = NULL; const char* only_r = NULL;
    for (int a = 2; a < argc; ++a) {
        char* eq = strchr(argv[a], '='); if (!eq) continue;
        *eq = 0; const char* k = argv[a]; int val = atoi(eq + 1);
#define K(name) if (!strcmp(k, #name)) v.name = val;
        K(sep_len) K(anchor_while_predicting) K(bwd_bound_kept) K(bwd_exact_first) K(seed_window)
        K(seed_back) K(seed_fwd) K(seed_choice) K(lit_reset_ge) K(gap_mode) K(fwd_after_close)
        K(loop_le) K(anchor_tie) K(reg_on_span) K(rend_mode) K(trace) K(anchor_margin) K(weak_seed_ratio)
#undef K
        if (!strcmp(k, "q")) only_q = eq + 1;
        if (!strcmp(k, "r")) only_r = eq + 1;
        if (!strcmp(k, "mal")) p.mal = val;
        if (!strcmp(k, "msl")) p.msl = val;
        if (!strcmp(k, "mrd")) p.mrd = val;
        if (!strcmp(k, "mqd")) p.mqd = val;
        if (!strcmp(k, "reg")) p.reg = val;
        if (!strcmp(k, "aw")) p.aw = val;
        if (!strcmp(k, "am")) p.am = val;
        if (!strcmp(k, "ar")) p.ar = val;
    }
    printf("query\treference\tpident\talnlen\tqstart\tqend\trstart\trend\tnt_match\tnt_mismatch\n");
    for (int r = 0; r < gs.n; ++r) {
        if (only_r && strcmp(only_r, gs.g[r].name)) continue;
        vo_ref_index* ix = vo_lz_build_index(gs.g[r].seq, gs.g[r].len, &p, &v);
        for (int q = 0; q < gs.n; ++q) {
            if (q == r) continue;
            if (only_q && strcmp(only_q, gs.g[q].name)) continue;
            vo_region* regs; int n;
            if (v.trace) fprintf(stderr, "PAIR %s %s\n", gs.g[q].name, gs.g[r].name);
            vo_lz_parse(ix, gs.g[q].seq, gs.g[q].len, &p, &v, &regs, &n);
            for (int k = 0; k < n; ++k) {
                vo_region* g = &regs[k];
                int alnlen = g->qend - g->qstart + 1;
                char buf[64]; vo_fmt_num(100.0 * g->n_match / alnlen, buf);
                printf("%s\t%s\t%s\t%d\t%d\t%d\t%lld\t%lld\t%d\t%d\n", gs.g[q].name, gs.g[r].name, buf, alnlen,
                       g->qstart + 1, g->qend + 1,
                       (long long)vo_rr_to_fwd1(ix, g->rstart), (long long)vo_rr_to_fwd1s(ix, g->rend, vo_rr_is_rev(ix, g->rstart)),
                       g->n_match, g->n_mismatch);
            }
            free(regs);
        }
        vo_lz_free_index(ix);
    }
    vo_free_genomes(&gs);
    return 0;
}
