/*
 * fasta_oracle.c — FASTA / FASTA.gz reader of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Input conventions follow the reference front-end: a single (multi-)FASTA file is read
 * with one genome per record, a directory with one genome per file
 * (vclust.py:687-702, 962-963, 1159-1160).  Genome name = first header token.
 */
#include "vclust_oracle.h"
#include <zlib.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

static uint8_t code_of(int ch) {
    switch (ch) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

static vo_genome* add_genome(vo_genome_set* s, const char* name) {
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 16;
        s->g = (vo_genome*)realloc(s->g, sizeof(vo_genome) * s->cap);
    }
    vo_genome* g = &s->g[s->n++];
    memset(g, 0, sizeof(*g));
    g->name = strdup(name);
    return g;
}

int vo_read_fasta(const char* path, int multisample, vo_genome_set* out) {
    gzFile f = gzopen(path, "rb");
    if (!f) return -1;
    gzbuffer(f, 1 << 20);
    vo_genome* cur = NULL; int64_t cap = 0;
    enum { BUF = 1 << 16 };
    char* buf = (char*)malloc(BUF);
    int in_header = 0; char hdr[4096]; int hl = 0;
    int n;
    while ((n = gzread(f, buf, BUF)) > 0) {
        for (int i = 0; i < n; ++i) {
            char ch = buf[i];
            if (in_header) {
                if (ch == '\n') {
                    hdr[hl] = 0; in_header = 0;
                    char* e = hdr; while (*e && *e != ' ' && *e != '\t' && *e != '\r') ++e; *e = 0;
                    if (multisample || cur == NULL) {
                        if (multisample) cur = add_genome(out, hdr);
                        else {
                            /* directory mode: genome named after the file (basename) */
                            const char* b = strrchr(path, '/'); b = b ? b + 1 : path;
                            cur = add_genome(out, b);
                        }
                        cap = 0;
                    } else {
                        /* further record of the same genome: parts are separated by one N */
                        if (cur->len + 1 > cap) { cap = cap ? cap * 2 : 1 << 16; cur->seq = (uint8_t*)realloc(cur->seq, cap); }
                        cur->seq[cur->len++] = 4;
                    }
                    cur->n_parts++;
                } else if (hl < (int)sizeof(hdr) - 1) hdr[hl++] = ch;
                continue;
            }
            if (ch == '>') { in_header = 1; hl = 0; continue; }
            if (ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t') continue;
            if (!cur) continue;
            if (cur->len + 1 > cap) { cap = cap ? cap * 2 : 1 << 16; cur->seq = (uint8_t*)realloc(cur->seq, cap); }
            cur->seq[cur->len++] = code_of((unsigned char)ch);
        }
    }
    free(buf);
    gzclose(f);
    return 0;
}

void vo_free_genomes(vo_genome_set* s) {
    for (int i = 0; i < s->n; ++i) { free(s->g[i].name); free(s->g[i].seq); }
    free(s->g); s->g = NULL; s->n = s->cap = 0;
}
