/*
 * fmt_oracle.c — number formatting of the LZ-ANI tables (TEST INFRASTRUCTURE ONLY).
 * Rule reverse-engineered from example/output/ani.tsv and ani.aln.tsv:
 *   - if some "%.{p}g", p in 1..6, reproduces the double exactly, print the shortest one
 *     ("1", "0.5625", "92", and 100 -> "1e+02");
 *   - otherwise 6 significant digits, fixed notation, trailing zeros kept, exact decimal
 *     ties rounded half-up ("0.5703125" -> "0.570313").
 */
#include "vclust_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int vo_fmt_num(double x, char* buf) {
    for (int p = 1; p <= 6; ++p) {
        char t[64];
        snprintf(t, sizeof t, "%.*g", p, x);
        if (strtod(t, NULL) == x) { strcpy(buf, t); return (int)strlen(buf); }
    }
    /* exact decimal expansion, then manual half-up rounding at 6 significant digits */
    char t[512];
    int neg = x < 0; if (neg) x = -x;
    snprintf(t, sizeof t, "%.120f", x);
    char* dot = strchr(t, '.');
    int int_len = (int)(dot - t);
    /* digits without the dot */
    char d[512]; int nd = 0;
    for (char* c = t; *c; ++c) if (*c != '.') d[nd++] = *c;
    d[nd] = 0;
    int first = 0; while (first < nd && d[first] == '0') ++first;   /* first significant digit */
    int keep_end = first + 6;                                        /* exclusive index */
    if (keep_end > nd) keep_end = nd;
    int round_up = (keep_end < nd && d[keep_end] >= '5');
    d[keep_end] = 0;
    if (round_up) {
        int j = keep_end - 1;
        while (j >= 0) { if (d[j] == '9') { d[j] = '0'; --j; } else { d[j]++; break; } }
        if (j < 0) { memmove(d + 1, d, keep_end + 1); d[0] = '1'; ++int_len; ++keep_end; }
    }
    /* rebuild: integer part d[0..int_len), fraction d[int_len..keep_end) */
    int o = 0;
    if (neg) buf[o++] = '-';
    if (keep_end < int_len) {          /* >= 1e6: pad integer digits with zeros */
        for (int j = 0; j < keep_end; ++j) buf[o++] = d[j];
        for (int j = keep_end; j < int_len; ++j) buf[o++] = '0';
        buf[o] = 0; return o;
    }
    for (int j = 0; j < int_len; ++j) buf[o++] = d[j];
    if (keep_end > int_len) {
        buf[o++] = '.';
        for (int j = int_len; j < keep_end; ++j) buf[o++] = d[j];
    }
    buf[o] = 0;
    return o;
}

int vo_fmt_len_ratio(int64_t a, int64_t b, char* buf) {
    if (a == b) { strcpy(buf, "1"); return 1; }
    int64_t lo = a < b ? a : b, hi = a < b ? b : a;
    return snprintf(buf, 32, "%.4f", (double)lo / (double)hi);
}
