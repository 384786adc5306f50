// vg_io.cpp — host side of the two stages: everything that turns the GPU's integers into the
// reference's files.  Floating point appears only here (ani-shorter uses log; ANI ratios are
// fp64 quotients of integers), so file identity with the CPU oracle reduces to integer identity.
//   fltr.txt            layout of example/output/fltr.txt        
//   ani.tsv, ids.tsv    layout of example/output/ani{,.ids}.tsv  
//   ani.aln.tsv         layout of example/output/ani.aln.tsv     
#include "vg_common.h"
#include <cmath>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <unordered_map>
#include <mutex>
#include <numeric>
#include <atomic>
#include <thread>

// ---------------------------------------------------------------- numbers
double vg_ani_shorter(int64_t shared, int64_t na, int64_t nb, int k) {
    int64_t mn = std::min(na, nb);
    if (mn <= 0 || shared <= 0) return 0.0;
    double j = (double)shared / (double)mn;
    return 1.0 + log(2.0 * j / (1.0 + j)) / (double)k;
}

// LZ-ANI prints a double with the shortest "%.{1..6}g" that reproduces it exactly, else with six
// significant digits in fixed notation (zeros kept), exact decimal ties rounded away from zero.
static int fmt_num_exact(double x, char* buf) {
    for (int prec = 1; prec <= 6; ++prec) {
        char t[64];
        snprintf(t, sizeof t, "%.*g", prec, x);
        if (strtod(t, nullptr) == x) { strcpy(buf, t); return (int)strlen(buf); }
    }
    bool neg = x < 0; double ax = neg ? -x : x;
    char full[400];
    snprintf(full, sizeof full, "%.150f", ax);                 // exact binary expansion
    std::string digits; int int_len = 0; bool seen_dot = false;
    for (char* c = full; *c; ++c) { if (*c == '.') { seen_dot = true; continue; } digits.push_back(*c); if (!seen_dot) ++int_len; }
    size_t first = 0; while (first < digits.size() && digits[first] == '0') ++first;
    size_t keep = std::min(digits.size(), first + 6);
    bool up = keep < digits.size() && digits[keep] >= '5';
    digits.resize(keep);
    if (up) {
        int j = (int)keep - 1;
        while (j >= 0 && digits[j] == '9') { digits[j] = '0'; --j; }
        if (j >= 0) digits[j]++; else { digits.insert(digits.begin(), '1'); ++int_len; }
    }
    std::string out = neg ? "-" : "";
    if ((int)digits.size() <= int_len) { out += digits; out.append(int_len - digits.size(), '0'); }
    else { out += digits.substr(0, int_len); out.push_back('.'); out += digits.substr(int_len); }
    strcpy(buf, out.c_str());
    return (int)out.size();
}

// Fast path for the overwhelmingly common case (a generic quotient that has no <= 6-digit decimal
// representation and is not near a rounding tie): six significant digits by one multiplication.
// Anything within 1e-6 of a short decimal or of a tie goes through the exact routine above.
int vg_fmt_num(double x, char* buf) {
    static const double P10[] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18 };
    if (!(x > 1e-12 && x < 1e6)) return fmt_num_exact(x, buf);
    int e = 0;                                                  // x in [10^e, 10^(e+1))
    if (x >= 1.0) { while (e < 6 && x >= P10[e + 1]) ++e; }
    else { e = -1; while (e > -13 && x < 1.0 / P10[-e]) --e; }
    const int k = 5 - e;                                        // y = x * 10^k in [1e5, 1e6)
    const double y = (k >= 0) ? x * P10[k] : x / P10[-k];
    const double fl = floor(y); const double fr = y - fl;
    if (!(y >= 1e5 && y < 999999.0)) return fmt_num_exact(x, buf);
    if (fr < 1e-6 || fr > 1.0 - 1e-6 || (fr > 0.5 - 1e-6 && fr < 0.5 + 1e-6)) return fmt_num_exact(x, buf);
    long d = (long)fl + (fr > 0.5 ? 1 : 0);                     // 6 significant digits
    char dig[8]; for (int i = 5; i >= 0; --i) { dig[i] = (char)('0' + d % 10); d /= 10; }
    int o = 0;
    if (e >= 0) {                                               // e+1 integer digits (e <= 5)
        for (int i = 0; i <= e; ++i) buf[o++] = dig[i];
        if (e < 5) { buf[o++] = '.'; for (int i = e + 1; i < 6; ++i) buf[o++] = dig[i]; }
    } else {
        buf[o++] = '0'; buf[o++] = '.';
        for (int i = 0; i < -e - 1; ++i) buf[o++] = '0';
        for (int i = 0; i < 6; ++i) buf[o++] = dig[i];
    }
    buf[o] = 0;
    return o;
}

int vg_fmt_len_ratio(int64_t a, int64_t b, char* buf) {
    if (a == b) { strcpy(buf, "1"); return 1; }
    return snprintf(buf, 32, "%.4f", (double)std::min(a, b) / (double)std::max(a, b));
}

// ---------------------------------------------------------------- fltr.txt
extern "C" int vg_filter_pairs(int k, int min_kmers, double min_ident, const int64_t* set_sizes, int64_t n_genomes,
                               const vg_pair_count* pairs, int64_t n_pairs, vg_pair_count** out, int64_t* n_out) {
    VG_API_BEGIN
    if (!set_sizes || (!pairs && n_pairs) || !out || !n_out) throw vg_error(VG_EINVAL, "vg_filter_pairs: null argument");
    // ani >= min_ident  <=>  j = shared / min(|A|, |B|) >= J*, J* = E / (2 - E), E = exp((min_ident - 1) k): one
    // division per pair instead of a logarithm.  A pair within 1e-9 (relative) of the threshold is decided by the
    // formula itself, so the kept set is exactly the formula's.
    vg_host_mark("filter_pairs: enter");
    const double E = std::exp((min_ident - 1.0) * (double)k);
    const double Jstar = E < 2.0 ? E / (2.0 - E) : INFINITY;
    vg_pair_count* o = (vg_pair_count*)malloc(sizeof(vg_pair_count) * std::max<size_t>(1, (size_t)n_pairs));
    if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
    size_t total = 0;
    auto keeps = [&](const vg_pair_count& p) {
        if ((int64_t)p.a >= n_genomes || (int64_t)p.b >= n_genomes) throw vg_error(VG_EINVAL, "pair id out of range");
        if ((int64_t)p.shared < min_kmers) return false;
        const int64_t mn = std::min(set_sizes[p.a], set_sizes[p.b]);
        if (mn <= 0 || p.shared == 0) return 0.0 >= min_ident;
        const double j = (double)p.shared / (double)mn;
        if (std::isfinite(Jstar) && std::fabs(j - Jstar) > 1e-9 * Jstar) return j > Jstar;
        return vg_ani_shorter(p.shared, set_sizes[p.a], set_sizes[p.b], k) >= min_ident;
    };
    try {
        // (10^5 .. 10^6 pairs -- every rank of a sharded call does this between the stages --: chunks over a few threads, kept pairs counted, then written in order)
        const int T = n_pairs >= (1 << 17) ? std::max(1, std::min(vg_host_threads(), 8)) : 1;
        if (T == 1) { for (int64_t i = 0; i < n_pairs; ++i) if (keeps(pairs[i])) o[total++] = pairs[i]; }
        else {
            std::vector<int64_t> kept((size_t)T + 1, 0);
            std::vector<uint8_t> flag((size_t)n_pairs);
            vg_parallel_chunks(n_pairs, T, [&](int64_t a, int64_t b, int t) { int64_t c = 0; for (int64_t i = a; i < b; ++i) { flag[(size_t)i] = keeps(pairs[i]); c += flag[(size_t)i]; } kept[(size_t)t + 1] = c; });
            for (int t = 0; t < T; ++t) kept[(size_t)t + 1] += kept[(size_t)t];
            vg_parallel_chunks(n_pairs, T, [&](int64_t a, int64_t b, int t) { int64_t at = kept[(size_t)t]; for (int64_t i = a; i < b; ++i) if (flag[(size_t)i]) o[at++] = pairs[i]; });
            total = (size_t)kept[(size_t)T];
        }
    } catch (...) { free(o); throw; }
    vg_host_mark("filter_pairs: done");
    struct { size_t n; size_t size() const { return n; } } keep{ total };
    *out = o; *n_out = (int64_t)keep.size();
    VG_API_END
}

extern "C" int vg_write_fltr(const vg_genomes* g, int k, double fraction, int min_kmers, double min_ident,
                             int max_seqs, const int64_t* set_sizes, const vg_pair_count* pairs,
                             int64_t n_pairs, const char* out_path) {
    VG_API_BEGIN
    if (!g || !set_sizes || (!pairs && n_pairs) || !out_path) throw vg_error(VG_EINVAL, "vg_write_fltr: null argument");
    // rows by a counting sort on the row genome (the pairs arrive in no order), then every row sorted by column
    const size_t ng = (size_t)std::max(g->n, 0);
    std::vector<size_t> first(ng + 2, 0);
    for (int64_t i = 0; i < n_pairs; ++i) { const uint32_t a = std::max(pairs[i].a, pairs[i].b); if (a < ng) first[a + 1]++; }
    for (size_t a = 0; a < ng; ++a) first[a + 1] += first[a];
    struct cell { uint32_t b, shared; };
    std::vector<cell> cells(first[ng]);
    {
        std::vector<size_t> cur(first.begin(), first.begin() + (std::ptrdiff_t)ng);
        for (int64_t i = 0; i < n_pairs; ++i) {
            const uint32_t a = std::max(pairs[i].a, pairs[i].b), bb = std::min(pairs[i].a, pairs[i].b);
            if (a < ng) cells[cur[a]++] = { bb, pairs[i].shared };
        }
    }
    FILE* f = fopen(out_path, "w");
    if (!f) throw vg_error(VG_EIO, std::string("cannot write ") + out_path);
    fprintf(f, "kmer-length: %d fraction: %g ,", k, fraction);
    for (int i = 0; i < g->n; ++i) fprintf(f, "%s,", g->names[i].c_str());
    fputc('\n', f);
    // rows are formatted by several threads, each its own contiguous range, and written in order
    const int n_thr = std::max(1, std::min(vg_host_threads(), 16));
    std::vector<std::string> text((size_t)n_thr);
    vg_parallel_chunks((int64_t)ng, n_thr, [&](int64_t lo, int64_t hi, int t) {
        std::string& o = text[(size_t)t];
        struct ent { uint32_t col; double ani; };
        std::vector<ent> row; char buf[64];
        for (int64_t a = lo; a < hi; ++a) {
            o += g->names[(size_t)a]; o += ',';
            cell* c0 = cells.data() + first[(size_t)a]; cell* c1 = cells.data() + first[(size_t)a + 1];
            std::sort(c0, c1, [](const cell& x, const cell& y) { return x.b < y.b; });
            row.clear();
            for (cell* c = c0; c < c1;) {
                uint64_t sh = 0; const uint32_t bcol = c->b;
                for (; c < c1 && c->b == bcol; ++c) sh += c->shared;           // per-shard partial counts add up
                if ((int64_t)sh < min_kmers || bcol >= (uint32_t)g->n || bcol == (uint32_t)a) continue;
                const double ani = vg_ani_shorter((int64_t)sh, set_sizes[a], set_sizes[bcol], k);
                if (ani >= min_ident) row.push_back({ bcol, ani });
            }
            if (max_seqs > 0 && (int)row.size() > max_seqs) {
                std::sort(row.begin(), row.end(), [](const ent& x, const ent& y) { return x.ani != y.ani ? x.ani > y.ani : x.col < y.col; });
                row.resize((size_t)max_seqs);
                std::sort(row.begin(), row.end(), [](const ent& x, const ent& y) { return x.col < y.col; });
            }
            for (auto& e : row) { const int n = snprintf(buf, sizeof buf, "%u:%.6f,", e.col + 1, e.ani); o.append(buf, (size_t)n); }
            o += '\n';
        }
    });
    for (auto& o : text) if (!o.empty() && fwrite(o.data(), 1, o.size(), f) != o.size()) { fclose(f); throw vg_error(VG_EIO, std::string("write error on ") + out_path); }
    if (fclose(f)) throw vg_error(VG_EIO, std::string("write error on ") + out_path);
    VG_API_END
}

// ---------------------------------------------------------------- align: order, filter, tasks
void vg_length_order(const vg_genomes* g) {
    if ((int)g->len_order.size() == g->n && (int)g->len_rank.size() == g->n) return;
    std::vector<int32_t> order((size_t)g->n), rank((size_t)g->n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return g->len[x] > g->len[y]; });
    for (int i = 0; i < g->n; ++i) rank[(size_t)order[(size_t)i]] = i;
    g->len_order.swap(order); g->len_rank.swap(rank);
}

extern "C" int vg_align_order(const vg_genomes* g, int32_t* order) {
    VG_API_BEGIN
    if (!g || !order) throw vg_error(VG_EINVAL, "vg_align_order: null argument");
    vg_length_order(g);
    if (g->n) memcpy(order, g->len_order.data(), sizeof(int32_t) * (size_t)g->n);
    VG_API_END
}

extern "C" int vg_read_filter(const vg_genomes* g, const char* path, double thr, vg_pair_count** pairs, int64_t* n_pairs) {
    VG_API_BEGIN
    if (!g || !pairs || !n_pairs) throw vg_error(VG_EINVAL, "vg_read_filter: null argument");
    std::vector<vg_pair_count> v;
    if (!path) {
        for (uint32_t a = 1; a < (uint32_t)g->n; ++a) for (uint32_t b = 0; b < a; ++b) v.push_back({ a, b, 0 });
    } else {
        FILE* f = fopen(path, "r");
        if (!f) throw vg_error(VG_EIO, std::string("cannot open filter ") + path);
        // names -> genome ids.  A filter written from the same FASTA lists the genomes in the set's own order: a name is
        // first compared with the name the order predicts (one string compare); the hash map over all names is only
        // built when a name turns up somewhere else.
        std::unordered_map<std::string, uint32_t> by_name; std::once_flag by_name_once;
        auto lookup = [&](const char* b, size_t len, int64_t guess) -> int64_t {
            if (guess >= 0 && guess < (int64_t)g->n) { const std::string& nm = g->names[(size_t)guess]; if (nm.size() == len && memcmp(nm.data(), b, len) == 0) return guess; }
            std::call_once(by_name_once, [&] { by_name.reserve((size_t)g->n * 2); for (int i = 0; i < g->n; ++i) by_name.emplace(g->names[i], (uint32_t)i); });
            auto it = by_name.find(std::string(b, len));
            return it == by_name.end() ? -1 : (int64_t)it->second;
        };
        std::string data;
        { fseek(f, 0, SEEK_END); const long long sz = ftell(f); fseek(f, 0, SEEK_SET);
          if (sz > 0) { data.resize((size_t)sz); const size_t r = fread(&data[0], 1, (size_t)sz, f); data.resize(r); } }
        { char buf[1 << 16]; size_t r; while ((r = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, r); }    // (a file that grew, or no size: the rest)
        fclose(f);
        // header: column names -> genome ids
        std::vector<int64_t> col_id;
        size_t body = data.find('\n'); if (body == std::string::npos) body = data.size();
        {
            size_t he = body; if (he > 0 && data[he - 1] == '\r') --he;
            size_t c = data.find(',');
            while (c != std::string::npos && c + 1 < he) {
                size_t n2 = data.find(',', c + 1); if (n2 == std::string::npos || n2 > he) break;
                col_id.push_back(lookup(data.data() + c + 1, n2 - c - 1, (int64_t)col_id.size()));
                c = n2;
            }
        }
        // rows: parsed by several threads, each a range of whole lines, in place (no per-field strings)
        const char* base = data.data(); const size_t n_data = data.size();
        const size_t b0 = std::min(body + 1, n_data);
        const int n_thr = std::max(1, std::min(vg_host_threads(), 16));
        std::vector<std::vector<vg_pair_count>> part((size_t)n_thr);
        vg_parallel_chunks((int64_t)(n_data - b0), n_thr, [&](int64_t lo, int64_t hi, int t) {
            size_t p0 = b0 + (size_t)lo, p1 = b0 + (size_t)hi;
            if (lo > 0) { const char* nl = (const char*)memchr(base + p0 - 1, '\n', n_data - (p0 - 1)); p0 = nl ? (size_t)(nl - base) + 1 : n_data; }   // first whole line
            if ((size_t)hi < n_data - b0) { const char* nl = (const char*)memchr(base + p1 - 1, '\n', n_data - (p1 - 1)); p1 = nl ? (size_t)(nl - base) + 1 : n_data; }
            std::vector<vg_pair_count>& out = part[(size_t)t];
            // the row in front of this one, + 1.  The first row of a chunk has no row in front of it: the first chunk starts
            // at genome 0, the others look around the genome their byte offset suggests (rows of similar length: a few
            // thousand name compares at most) before the hash map over all names is built for them
            int64_t guess = lo == 0 ? 0 : -1;
            bool first_row = lo > 0;
            while (p0 < p1) {
                const char* nl = (const char*)memchr(base + p0, '\n', n_data - p0);
                size_t le = nl ? (size_t)(nl - base) : n_data; const size_t next = le + 1;
                if (le > p0 && base[le - 1] == '\r') --le;
                const char* q = base + p0; const char* const e = base + le;
                p0 = next;
                const char* c = (const char*)memchr(q, ',', (size_t)(e - q)); if (!c) continue;
                if (first_row) {
                    first_row = false;
                    const size_t len = (size_t)(c - q);
                    const int64_t est = (int64_t)((double)g->n * (double)(p0 - b0) / (double)std::max<size_t>(1, n_data - b0));
                    for (int64_t d = 0; d <= 4096 && guess < 0; ++d)
                        for (int64_t cand : { est + d, est - d }) {
                            if (cand < 0 || cand >= (int64_t)g->n) continue;
                            const std::string& nm = g->names[(size_t)cand];
                            if (nm.size() == len && memcmp(nm.data(), q, len) == 0) { guess = cand; break; }
                        }
                }
                const int64_t row = lookup(q, (size_t)(c - q), guess);
                if (row >= 0) guess = row + 1;
                q = c + 1;
                while (q < e) {
                    const char* n2 = (const char*)memchr(q, ',', (size_t)(e - q)); const char* fe = n2 ? n2 : e;
                    const char* colon = (const char*)memchr(q, ':', (size_t)(fe - q));
                    if (colon) {
                        long ci = 0; const char* d = q; bool ok = d < colon;
                        for (; d < colon; ++d) { if (*d < '0' || *d > '9') { ok = false; break; } ci = ci * 10 + (*d - '0'); }
                        if (!ok) ci = atol(std::string(q, colon).c_str());
                        ci -= 1;
                        // value: digits '.' digits with <= 15 digits is exact as mantissa / 10^k; anything else goes to strtod
                        double val; { uint64_t m = 0; int nd = 0, frac = -1; bool plain = colon + 1 < fe;
                            for (const char* x = colon + 1; x < fe && plain; ++x) {
                                if (*x >= '0' && *x <= '9') { m = m * 10 + (uint64_t)(*x - '0'); ++nd; if (frac >= 0) ++frac; }
                                else if (*x == '.' && frac < 0) frac = 0; else plain = false;
                            }
                            if (plain && nd <= 15) { static const double p10[16] = { 1, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15 };
                                val = (double)m / p10[frac < 0 ? 0 : frac]; }
                            else val = atof(std::string(colon + 1, fe).c_str()); }
                        if (row >= 0 && ci >= 0 && ci < (long)col_id.size() && col_id[(size_t)ci] >= 0 && col_id[(size_t)ci] != row && val >= thr) {
                            const uint32_t x = (uint32_t)row, y = (uint32_t)col_id[(size_t)ci];
                            out.push_back({ std::max(x, y), std::min(x, y), 0 });
                        }
                    }
                    if (!n2) break;
                    q = n2 + 1;
                }
            }
        });
        for (auto& pv : part) v.insert(v.end(), pv.begin(), pv.end());
        // ascending (a, b): two stable counting passes over the genome ids (b, then a) instead of a comparison sort
        if ((int64_t)v.size() > 4 * (int64_t)g->n) {
            std::vector<vg_pair_count> tmp(v.size()); std::vector<int64_t> at((size_t)g->n + 1);
            for (int pass = 0; pass < 2; ++pass) {
                std::fill(at.begin(), at.end(), 0);
                for (const auto& e : v) at[(size_t)(pass ? e.a : e.b) + 1]++;
                for (int i = 0; i < g->n; ++i) at[(size_t)i + 1] += at[(size_t)i];
                for (const auto& e : v) tmp[(size_t)at[(size_t)(pass ? e.a : e.b)]++] = e;
                v.swap(tmp);
            }
        } else
        std::sort(v.begin(), v.end(), [](const vg_pair_count& x, const vg_pair_count& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
        v.erase(std::unique(v.begin(), v.end(), [](const vg_pair_count& x, const vg_pair_count& y) { return x.a == y.a && x.b == y.b; }), v.end());
    }
    vg_pair_count* o = (vg_pair_count*)malloc(sizeof(vg_pair_count) * std::max<size_t>(1, v.size()));
    if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
    if (!v.empty()) memcpy(o, v.data(), sizeof(vg_pair_count) * v.size());
    *pairs = o; *n_pairs = (int64_t)v.size();
    VG_API_END
}

extern "C" int vg_align_tasks(const vg_genomes* g, const vg_pair_count* pairs, int64_t n_pairs,
                              vg_task** tasks, int64_t* n_tasks) {
    return vg_align_tasks_perm(g, pairs, n_pairs, tasks, n_tasks, nullptr);
}
// the same, also telling which pair every canonical couple came from (perm[c] = index into `pairs` of the couple whose
// rows are tasks 2c and 2c + 1): the sharded align stage places gathered rows through it
int vg_align_tasks_perm(const vg_genomes* g, const vg_pair_count* pairs, int64_t n_pairs,
                        vg_task** tasks, int64_t* n_tasks, uint32_t* perm) {
    VG_API_BEGIN
    if (!g || (!pairs && n_pairs) || !tasks || !n_tasks) throw vg_error(VG_EINVAL, "vg_align_tasks: null argument");
    if (perm && n_pairs >= (1LL << 32)) throw vg_error(VG_EOVERFLOW, "more than 2^32 pairs");
    vg_host_mark("align_tasks: enter");
    vg_length_order(g);
    vg_host_mark("align_tasks: length order");
    const std::vector<int32_t>& order = g->len_order; const std::vector<int32_t>& rank = g->len_rank;
    // couples sorted by (lo, hi) of the length ranks: two stable counting passes, on hi and then on lo (LSD order).
    // One thread: 10^5..10^6 couples are a few milliseconds of cache-resident work, less than starting helpers costs.
    struct rp { int32_t lo, hi; uint32_t idx; };
    if (n_pairs >= (1 << 21) && vg_host_threads() > 1) {
        // millions of couples (contigs-1M: 3.5 M, 75 ms on one thread): range partition on lo over the threads (counts,
        // offsets, scatter), every range sorted on its own, the task couples written in parallel
        const int T = std::max(2, std::min(vg_host_threads(), 16));
        const int64_t ng = std::max<int64_t>(1, g->n);
        std::vector<rp, no_init_alloc<rp>> v, tmp; v.resize((size_t)n_pairs); tmp.resize((size_t)n_pairs);      // (84 MB at 3.5 M couples: not cleared by one thread first)
        std::vector<std::vector<int64_t>> cnt((size_t)T, std::vector<int64_t>((size_t)T + 1, 0));
        std::atomic<bool> bad(false);
        auto bucket_of = [&](int32_t lo) { return (int)((int64_t)lo * T / ng); };
        vg_parallel_chunks(n_pairs, T, [&](int64_t a, int64_t b, int t) {
            for (int64_t i = a; i < b; ++i) {
                if (pairs[i].a >= (uint32_t)g->n || pairs[i].b >= (uint32_t)g->n) { bad = true; return; }
                const int32_t x = rank[pairs[i].a], y = rank[pairs[i].b];
                v[(size_t)i] = { std::min(x, y), std::max(x, y), (uint32_t)i };
                cnt[(size_t)t][(size_t)bucket_of(v[(size_t)i].lo)]++;
            }
        });
        if (bad.load()) throw vg_error(VG_EINVAL, "pair id out of range");
        std::vector<int64_t> b_first((size_t)T + 1, 0);
        { int64_t run = 0; for (int bk = 0; bk < T; ++bk) { b_first[(size_t)bk] = run; for (int t = 0; t < T; ++t) { const int64_t c = cnt[(size_t)t][(size_t)bk]; cnt[(size_t)t][(size_t)bk] = run; run += c; } } b_first[(size_t)T] = run; }
        vg_parallel_chunks(n_pairs, T, [&](int64_t a, int64_t b, int t) {
            for (int64_t i = a; i < b; ++i) tmp[(size_t)cnt[(size_t)t][(size_t)bucket_of(v[(size_t)i].lo)]++] = v[(size_t)i];
        });
        vg_task* o = (vg_task*)malloc(sizeof(vg_task) * std::max<size_t>(1, 2 * tmp.size()));
        if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
        auto finish = [&](int64_t a, int64_t b) {
            for (int64_t bk = a; bk < b; ++bk) {
                std::sort(tmp.begin() + b_first[(size_t)bk], tmp.begin() + b_first[(size_t)bk + 1], [](const rp& x, const rp& y) { return x.lo != y.lo ? x.lo < y.lo : (x.hi != y.hi ? x.hi < y.hi : x.idx < y.idx); });
                for (int64_t i = b_first[(size_t)bk]; i < b_first[(size_t)bk + 1]; ++i) {
                    o[2 * i] = { (uint32_t)order[(size_t)tmp[(size_t)i].hi], (uint32_t)order[(size_t)tmp[(size_t)i].lo] };
                    o[2 * i + 1] = { (uint32_t)order[(size_t)tmp[(size_t)i].lo], (uint32_t)order[(size_t)tmp[(size_t)i].hi] };
                    if (perm) perm[i] = tmp[(size_t)i].idx;
                }
            }
        };
        { std::vector<std::thread> th; for (int t = 1; t < T; ++t) th.emplace_back(finish, (int64_t)t, (int64_t)t + 1); finish(0, 1); for (auto& x : th) x.join(); }
        vg_host_mark("align_tasks: done");
        *tasks = o; *n_tasks = (int64_t)(2 * tmp.size());
        return VG_OK;
    }
    std::vector<rp> v((size_t)n_pairs), tmp((size_t)n_pairs);
    std::vector<int64_t> at_lo((size_t)g->n + 1, 0), at_hi((size_t)g->n + 1, 0);
    for (int64_t i = 0; i < n_pairs; ++i) {
        if (pairs[i].a >= (uint32_t)g->n || pairs[i].b >= (uint32_t)g->n) throw vg_error(VG_EINVAL, "pair id out of range");
        const int32_t x = rank[pairs[i].a], y = rank[pairs[i].b];
        v[(size_t)i] = { std::min(x, y), std::max(x, y), (uint32_t)i };
        at_lo[(size_t)v[(size_t)i].lo + 1]++; at_hi[(size_t)v[(size_t)i].hi + 1]++;
    }
    for (int key = 0; key < g->n; ++key) { at_lo[(size_t)key + 1] += at_lo[(size_t)key]; at_hi[(size_t)key + 1] += at_hi[(size_t)key]; }
    for (int64_t i = 0; i < n_pairs; ++i) tmp[(size_t)at_hi[(size_t)v[(size_t)i].hi]++] = v[(size_t)i];
    for (int64_t i = 0; i < n_pairs; ++i) v[(size_t)at_lo[(size_t)tmp[(size_t)i].lo]++] = tmp[(size_t)i];
    vg_task* o = (vg_task*)malloc(sizeof(vg_task) * std::max<size_t>(1, 2 * v.size()));
    if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
    for (size_t i = 0; i < v.size(); ++i) {
        o[2 * i] = { (uint32_t)order[(size_t)v[i].hi], (uint32_t)order[(size_t)v[i].lo] };       // row (q = b, r = a)
        o[2 * i + 1] = { (uint32_t)order[(size_t)v[i].lo], (uint32_t)order[(size_t)v[i].hi] };   // row (q = a, r = b)
        if (perm) perm[i] = v[i].idx;
    }
    vg_host_mark("align_tasks: done");
    *tasks = o; *n_tasks = (int64_t)(2 * v.size());
    VG_API_END
}

// ---------------------------------------------------------------- ani.tsv / ids.tsv / aln.tsv
extern "C" int vg_write_ani(const vg_genomes* g, const vg_task* tasks, const vg_pair_stat* stats, int64_t n_tasks,
                            const vg_region* regions, int64_t n_regions, const char* out_path, const vg_align_params* p) {
    VG_API_BEGIN
    if (!g || (!tasks && n_tasks) || (!stats && n_tasks) || !out_path || !p) throw vg_error(VG_EINVAL, "vg_write_ani: null argument");
    if (n_tasks & 1) throw vg_error(VG_EINVAL, "vg_write_ani: tasks must come in (q,r),(r,q) couples");
    std::vector<int32_t> order(g->n), rank(g->n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return g->len[x] > g->len[y]; });
    for (int i = 0; i < g->n; ++i) rank[order[i]] = i;
    {
        std::string ids(out_path);
        if (ids.size() > 4 && ids.compare(ids.size() - 4, 4, ".tsv") == 0) ids.resize(ids.size() - 4);
        ids += ".ids.tsv";
        FILE* f = fopen(ids.c_str(), "w");
        if (!f) throw vg_error(VG_EIO, "cannot write " + ids);
        fprintf(f, "id\tseq_len\tno_parts\n");
        for (int i = 0; i < g->n; ++i) fprintf(f, "%s\t%lld\t%d\n", g->names[order[i]].c_str(), (long long)g->len[order[i]], g->n_parts[order[i]]);
        if (fclose(f)) throw vg_error(VG_EIO, "write error on " + ids);
    }
    FILE* f = fopen(out_path, "w");
    if (!f) throw vg_error(VG_EIO, std::string("cannot write ") + out_path);
    for (int c = 0; c < p->n_out_columns; ++c) fprintf(f, "%s%s", c ? "\t" : "", p->out_columns[c]);
    fputc('\n', f);
    // rows are formatted in parallel into per-chunk buffers and written in order
    for (int c = 0; c < p->n_out_columns; ++c) {
        static const char* known[] = { "qidx", "ridx", "query", "reference", "tani", "gani", "ani", "qcov", "rcov", "num_alns",
                                       "len_ratio", "qlen", "rlen", "nt_match", "nt_mismatch" };
        bool ok = false; for (const char* kn : known) ok |= !strcmp(kn, p->out_columns[c]);
        if (!ok) { fclose(f); throw vg_error(VG_EINVAL, std::string("unknown output column ") + p->out_columns[c]); }
    }
    for (int64_t t = 0; t < n_tasks; ++t)
        if (tasks[t ^ 1].q != tasks[t].r || tasks[t ^ 1].r != tasks[t].q) { fclose(f); throw vg_error(VG_EINVAL, "vg_write_ani: task couple mismatch"); }
    const int nthreads = std::max(1, std::min(p->num_threads > 0 ? p->num_threads : 8, 64));
    const int64_t chunk = 1 << 14;
    const int64_t n_chunks = (n_tasks + chunk - 1) / chunk;
    std::vector<std::string> out_chunks((size_t)n_chunks);
    auto format_chunk = [&](int64_t ci) {
        std::string& o = out_chunks[(size_t)ci];
        o.reserve((size_t)chunk * 96);
        char buf[64];
        for (int64_t t = ci * chunk; t < std::min(n_tasks, (ci + 1) * chunk); ++t) {
            const vg_task& tk = tasks[t]; const vg_pair_stat& x = stats[t]; const vg_pair_stat& rev = stats[t ^ 1];
            int64_t lq = g->len[tk.q], lr = g->len[tk.r];
            double ani = x.aln_len ? (double)x.n_match / (double)x.aln_len : 0.0;
            double gani = lq ? (double)x.n_match / (double)lq : 0.0;
            double qcov = lq ? (double)x.aln_len / (double)lq : 0.0;
            double rcov = lr ? (double)rev.aln_len / (double)lr : 0.0;
            double tani = (lq + lr) ? (double)((uint64_t)x.n_match + rev.n_match) / (double)(lq + lr) : 0.0;
            if (p->out_tani > 0 && tani < p->out_tani) continue;
            if (p->out_gani > 0 && gani < p->out_gani) continue;
            if (p->out_ani > 0 && ani < p->out_ani) continue;
            if (p->out_qcov > 0 && qcov < p->out_qcov) continue;
            if (p->out_rcov > 0 && rcov < p->out_rcov) continue;
            for (int c = 0; c < p->n_out_columns; ++c) {
                const char* col = p->out_columns[c];
                if (c) o.push_back('\t');
                if (!strcmp(col, "qidx")) o += std::to_string(rank[tk.q]);
                else if (!strcmp(col, "ridx")) o += std::to_string(rank[tk.r]);
                else if (!strcmp(col, "query")) o += g->names[tk.q];
                else if (!strcmp(col, "reference")) o += g->names[tk.r];
                else if (!strcmp(col, "tani")) { vg_fmt_num(tani, buf); o += buf; }
                else if (!strcmp(col, "gani")) { vg_fmt_num(gani, buf); o += buf; }
                else if (!strcmp(col, "ani")) { vg_fmt_num(ani, buf); o += buf; }
                else if (!strcmp(col, "qcov")) { vg_fmt_num(qcov, buf); o += buf; }
                else if (!strcmp(col, "rcov")) { vg_fmt_num(rcov, buf); o += buf; }
                else if (!strcmp(col, "num_alns")) o += std::to_string(x.n_regions);
                else if (!strcmp(col, "len_ratio")) { vg_fmt_len_ratio(lq, lr, buf); o += buf; }
                else if (!strcmp(col, "qlen")) o += std::to_string(lq);
                else if (!strcmp(col, "rlen")) o += std::to_string(lr);
                else if (!strcmp(col, "nt_match")) o += std::to_string(x.n_match);
                else if (!strcmp(col, "nt_mismatch")) o += std::to_string(x.aln_len - x.n_match);
            }
            o.push_back('\n');
        }
    };
    {
        std::atomic<int64_t> next(0);
        auto work = [&]() { for (;;) { int64_t ci = next.fetch_add(1); if (ci >= n_chunks) break; format_chunk(ci); } };
        std::vector<std::thread> th;
        for (int t = 1; t < std::min<int64_t>(nthreads, n_chunks); ++t) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
    }
    for (auto& o : out_chunks) if (!o.empty() && fwrite(o.data(), 1, o.size(), f) != o.size()) { fclose(f); throw vg_error(VG_EIO, std::string("write error on ") + out_path); }
    if (fclose(f)) throw vg_error(VG_EIO, std::string("write error on ") + out_path);

    if (p->out_aln_path && regions) {
        std::vector<vg_region> v(regions, regions + n_regions);
        std::sort(v.begin(), v.end(), [](const vg_region& x, const vg_region& y) {
            if (x.task != y.task) return x.task < y.task;
            int lx = x.qend - x.qstart, ly = y.qend - y.qstart;
            if (lx != ly) return lx > ly;
            return x.qstart < y.qstart;
        });
        FILE* fa = fopen(p->out_aln_path, "w");
        if (!fa) throw vg_error(VG_EIO, std::string("cannot write ") + p->out_aln_path);
        fprintf(fa, "query\treference\tpident\talnlen\tqstart\tqend\trstart\trend\tnt_match\tnt_mismatch\n");
        for (auto& r : v) {
            if ((int64_t)r.task >= n_tasks) continue;
            const vg_task& tk = tasks[r.task];
            int64_t L = g->len[tk.r];
            // fwd | N | rc space -> 1-based forward; the strand of a region is that of its rstart (rend is a
            // virtual end and may lie past the end of the strand)
            const bool rev = r.rstart > L;
            auto fwd1 = [&](int64_t rr) { return rev ? L - (rr - (L + 1)) : rr + 1; };
            int alnlen = r.qend - r.qstart + 1; char buf[64];
            vg_fmt_num(100.0 * r.n_match / alnlen, buf);
            fprintf(fa, "%s\t%s\t%s\t%d\t%d\t%d\t%lld\t%lld\t%d\t%d\n", g->names[tk.q].c_str(), g->names[tk.r].c_str(), buf, alnlen,
                    r.qstart + 1, r.qend + 1, (long long)fwd1(r.rstart), (long long)fwd1(r.rend), r.n_match, alnlen - r.n_match);
        }
        if (fclose(fa)) throw vg_error(VG_EIO, std::string("write error on ") + p->out_aln_path);
    }
    VG_API_END
}
