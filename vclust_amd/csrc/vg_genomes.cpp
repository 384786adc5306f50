// vg_genomes.cpp — FASTA / FASTA.gz ingest, 2-bit packing, HBM residency.
// Input conventions follow the reference front-end (vclust.py:687-702, 962-963, 1159-1160):
// a single (multi-)FASTA file = one genome per record, a directory = one genome per file.
#include "vg_common.h"
#include <zlib.h>
#include <string.h>
#include <stdlib.h>
#include <thread>
#include <atomic>
#include <algorithm>

static inline uint8_t code_of(unsigned char ch) {
    switch (ch) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

void vg_genomes_append(vg_genomes* g, const std::string& name, const uint8_t* codes, int64_t len, int n_parts) {
    if (g->base_off.empty()) g->base_off.push_back(0);
    int64_t off = g->base_off.back();
    const int64_t al = 1LL << g->align_shift;
    int64_t padded = (len + al - 1) / al * al;
    if (padded == 0) padded = al;
    g->packed.resize((off + padded) / 16, 0u);
    g->nmask.resize((off + padded) / 32, 0u);
    uint32_t* pk = g->packed.data() + off / 16;
    uint32_t* mk = g->nmask.data() + off / 32;
    bool any_n = false;
    for (int64_t i = 0; i < len; ++i) {
        uint8_t c = codes[i];
        if (c > 3) { mk[i >> 5] |= 1u << (i & 31); any_n = true; }
        else pk[i >> 4] |= (uint32_t)c << (2 * (i & 15));
    }
    for (int64_t i = len; i < padded; ++i) mk[i >> 5] |= 1u << (i & 31);
    g->names.push_back(name);
    g->len.push_back(len);
    g->n_parts.push_back(n_parts);
    g->has_n.push_back(any_n ? 1 : 0);
    g->base_off.push_back(off + padded);
    for (int64_t b = off >> g->align_shift; b < (off + padded) >> g->align_shift; ++b) g->blk2g.push_back((uint32_t)g->n);
    g->n++;
}

int vg_choose_align_shift(int64_t total_len, int64_t n) {
    int64_t mean = n > 0 ? total_len / n : 0;
    int sh = 6;
    while (sh < 12 && (1LL << (sh + 1)) <= mean / 16) ++sh;
    return sh;
}

void vg_genomes_finish(vg_genomes* g) {
    if (g->base_off.empty()) g->base_off.push_back(0);
    // slack so that kernels may read a few words past the last genome
    g->packed.resize(g->padded_total() / 16 + 16, 0u);
    g->nmask.resize(g->padded_total() / 32 + 16, 0xffffffffu);
    g->blk2g.push_back(g->n > 0 ? (uint32_t)(g->n - 1) : 0u);
}

namespace {
struct rec { std::string name; std::vector<uint8_t> codes; int n_parts = 0; };

// parse one file into records (multisample) or one record (one genome per file)
void read_fasta(const std::string& path, bool multisample, std::vector<rec>& out) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw vg_error(VG_EIO, "cannot open " + path);
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(1 << 20);
    bool in_header = false; std::string hdr;
    rec* cur = nullptr;
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) {
        for (int i = 0; i < n; ++i) {
            char ch = buf[i];
            if (in_header) {
                if (ch == '\n') {
                    in_header = false;
                    size_t e = 0; while (e < hdr.size() && hdr[e] != ' ' && hdr[e] != '\t' && hdr[e] != '\r') ++e;
                    hdr.resize(e);
                    if (multisample || !cur) {
                        out.emplace_back(); cur = &out.back();
                        if (multisample) cur->name = hdr;
                        else { size_t s = path.find_last_of('/'); cur->name = s == std::string::npos ? path : path.substr(s + 1); }
                    } else cur->codes.push_back(4);      // contigs of one genome are separated by one N
                    cur->n_parts++;
                } else hdr.push_back(ch);
                continue;
            }
            if (ch == '>') { in_header = true; hdr.clear(); continue; }
            if (ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t') continue;
            if (cur) cur->codes.push_back(code_of((unsigned char)ch));
        }
    }
    int zerr = 0; (void)gzerror(f, &zerr);
    gzclose(f);
    if (n < 0 || (zerr != Z_OK && zerr != Z_STREAM_END)) throw vg_error(VG_EIO, "read error in " + path);
}
}  // namespace

extern "C" int vg_genomes_load(const char* const* paths, int n_paths, int multisample, int n_threads,
                               vg_genomes** out) {
    VG_API_BEGIN
    if (!paths || n_paths <= 0 || !out) throw vg_error(VG_EINVAL, "vg_genomes_load: bad arguments");
    std::vector<std::vector<rec>> per_file(n_paths);
    bool multi = multisample && n_paths == 1;
    int nt = std::max(1, std::min(n_threads > 0 ? n_threads : 1, n_paths));
    std::atomic<int> next(0); std::string first_err; std::atomic<bool> failed(false);
    auto work = [&]() {
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= n_paths || failed.load()) break;
            try { read_fasta(paths[i], multi, per_file[i]); }
            catch (const std::exception& e) { if (!failed.exchange(true)) first_err = e.what(); }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (failed.load()) throw vg_error(VG_EIO, first_err);
    vg_genomes* g = new vg_genomes();
    {
        int64_t tot = 0, cnt = 0;
        for (auto& v : per_file) for (auto& r : v) { tot += (int64_t)r.codes.size(); ++cnt; }
        g->align_shift = vg_choose_align_shift(tot, cnt);
    }
    for (auto& v : per_file)
        for (auto& r : v) vg_genomes_append(g, r.name, r.codes.data(), (int64_t)r.codes.size(), r.n_parts);
    vg_genomes_finish(g);
    *out = g;
    VG_API_END
}

extern "C" int vg_genomes_from_codes(const uint8_t* codes, const int64_t* offsets, int n_genomes,
                                     const char* const* names, vg_genomes** out) {
    VG_API_BEGIN
    if (!codes || !offsets || n_genomes < 0 || !out) throw vg_error(VG_EINVAL, "vg_genomes_from_codes: bad arguments");
    vg_genomes* g = new vg_genomes();
    g->align_shift = vg_choose_align_shift(n_genomes > 0 ? offsets[n_genomes] - offsets[0] : 0, n_genomes);
    for (int i = 0; i < n_genomes; ++i) {
        std::string nm = names && names[i] ? names[i] : ("g" + std::to_string(i));
        vg_genomes_append(g, nm, codes + offsets[i], offsets[i + 1] - offsets[i], 1);
    }
    vg_genomes_finish(g);
    *out = g;
    VG_API_END
}

extern "C" void vg_genomes_free(vg_genomes* g) { delete g; }
extern "C" int vg_genomes_count(const vg_genomes* g) { return g ? g->n : 0; }
extern "C" int64_t vg_genomes_total_len(const vg_genomes* g) {
    int64_t t = 0; if (g) for (auto l : g->len) t += l; return t;
}
extern "C" int vg_genomes_lengths(const vg_genomes* g, int64_t* out) {
    if (!g || !out) return VG_EINVAL;
    for (int i = 0; i < g->n; ++i) out[i] = g->len[i];
    return VG_OK;
}
extern "C" const char* vg_genomes_name(const vg_genomes* g, int idx) {
    if (!g || idx < 0 || idx >= g->n) return "";
    return g->names[idx].c_str();
}

extern "C" int vg_genomes_to_device(vg_genomes* g) {
    VG_API_BEGIN
    if (!g) throw vg_error(VG_EINVAL, "null genome set");
    vg_require_device();
    int dev = 0; VG_HIP(hipGetDevice(&dev));
    if (g->device == dev) return VG_OK;
    hipStream_t s = vg_stream();
    g->d_packed.alloc(g->packed.size()); g->d_packed.upload(g->packed.data(), g->packed.size(), s);
    g->d_nmask.alloc(g->nmask.size());   g->d_nmask.upload(g->nmask.data(), g->nmask.size(), s);
    g->d_base_off.alloc(g->base_off.size()); g->d_base_off.upload(g->base_off.data(), g->base_off.size(), s);
    g->d_len.alloc(std::max<size_t>(1, g->len.size())); if (g->n) g->d_len.upload(g->len.data(), g->len.size(), s);
    g->d_has_n.alloc(std::max<size_t>(1, g->has_n.size())); if (g->n) g->d_has_n.upload(g->has_n.data(), g->has_n.size(), s);
    g->d_blk2g.alloc(g->blk2g.size()); g->d_blk2g.upload(g->blk2g.data(), g->blk2g.size(), s);
    VG_HIP(hipStreamSynchronize(s));
    g->device = dev;
    VG_API_END
}
