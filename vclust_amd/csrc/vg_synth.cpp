// vg_synth.cpp — multithreaded generator of the synthetic workloads of SURVEY.md §8(d) (bench / test input,
// not part of the prefilter -> align path).  Bit-identical to vclust_amd/synth.py, which is the definition:
// every draw is splitmix64 as a counter-based generator, draw(stream, i) = mix64(stream + (i + 1) * GOLDEN).
// The family plan (family index, members, ancestor length) comes from the Python side, so no libm call
// is involved here and the two implementations agree to the last base (tests/test_synth.py pins sha256).
#include "vg_common.h"
#include <thread>
#include <atomic>
#include <vector>
#include <cstring>

#pragma clang fp contract(off)

namespace {
constexpr uint64_t G = 0x9E3779B97F4A7C15ULL, M1 = 0xBF58476D1CE4E5B9ULL, M2 = 0x94D049BB133111EBULL;
inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * M1; z = (z ^ (z >> 27)) * M2; return z ^ (z >> 31); }
inline uint64_t stream_id(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    uint64_t z = mix64(seed + G);
    z = mix64(z ^ (a * M1));
    z = mix64(z ^ (b * M2 + 1));
    return mix64(z ^ (c * G + 2));
}
inline uint64_t draw(uint64_t stream, uint64_t i) { return mix64(stream + (i + 1) * G); }
inline double unit(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
inline int64_t below(uint64_t x, int64_t n) { return (int64_t)(((x >> 32) * (uint64_t)n) >> 32); }

void mutate(uint64_t seed, uint64_t fam, uint64_t mem, const std::vector<uint8_t>& anc, double p_lo, double p_hi, int n_indels,
            std::vector<uint8_t>& g) {
    g = anc;
    const uint64_t sc = stream_id(seed, fam, mem, 1);
    auto ctl = [&](int i) { return draw(sc, (uint64_t)i); };
    const double p = p_lo + (p_hi - p_lo) * unit(ctl(0));
    const uint64_t sd = stream_id(seed, fam, mem, 2);
    for (size_t i = 0; i < g.size(); ++i) {
        const uint64_t d = draw(sd, i);
        if (unit(d) < p) g[i] = (uint8_t)((g[i] + (uint8_t)((d & 0xffff) % 3) + 1) & 3);
    }
    for (int t = 0; t < n_indels; ++t) {
        const uint64_t c0 = ctl(8 + 4 * t), c1 = ctl(9 + 4 * t), c2 = ctl(10 + 4 * t);
        const int64_t pos = below(c0, (int64_t)g.size()); const int64_t ln = 1 + below(c1, 50);
        if (c2 >> 63) {
            const int64_t e = std::min<int64_t>(pos + ln, (int64_t)g.size());
            g.erase(g.begin() + pos, g.begin() + e);
        } else {
            std::vector<uint8_t> ins((size_t)ln);
            const uint64_t si = stream_id(seed, fam, mem, 3 + (uint64_t)t);
            for (int64_t j = 0; j < ln; ++j) ins[(size_t)j] = (uint8_t)(draw(si, (uint64_t)j) >> 62);
            g.insert(g.begin() + pos, ins.begin(), ins.end());
        }
    }
    if (unit(ctl(1)) < 0.2 && g.size() > 6000) {
        const int64_t ln = 1000 + below(ctl(2), 4001); const int64_t pos = below(ctl(3), (int64_t)g.size() - ln);
        for (int64_t a = pos, b = pos + ln - 1; a <= b; ++a, --b) {
            const uint8_t x = (uint8_t)(3 - g[(size_t)a]), y = (uint8_t)(3 - g[(size_t)b]);
            g[(size_t)a] = y; g[(size_t)b] = x;
        }
    }
    if (unit(ctl(4)) < 0.2 && g.size() > 6000) {
        const int64_t ln = 1000 + below(ctl(5), 4001); const int64_t pos = below(ctl(6), (int64_t)g.size() - ln);
        std::vector<uint8_t> seg(g.begin() + pos, g.begin() + pos + ln);
        g.erase(g.begin() + pos, g.begin() + pos + ln);
        const int64_t dst = below(ctl(7), (int64_t)g.size());
        g.insert(g.begin() + dst, seg.begin(), seg.end());
    }
}
}  // namespace

// fam_idx / members / lengths: the family plan (n_fam entries).  Writes the base codes (0..3) of all genomes,
// family after family, member after member, into *codes_out (vg_free) and their n_genomes + 1 offsets into
// *offsets_out (vg_free).
extern "C" int vg_synth_plan(const int64_t* fam_idx, const int32_t* members, const int32_t* lengths, int64_t n_fam,
                             uint64_t seed, double p_lo, double p_hi, int n_indels, int n_threads,
                             uint8_t** codes_out, int64_t** offsets_out, int64_t* n_genomes) {
    VG_API_BEGIN
    if (!fam_idx || !members || !lengths || !codes_out || !offsets_out || !n_genomes) throw vg_error(VG_EINVAL, "vg_synth_plan: null argument");
    std::vector<int64_t> first((size_t)n_fam + 1, 0);
    for (int64_t f = 0; f < n_fam; ++f) first[(size_t)f + 1] = first[(size_t)f] + members[f];
    const int64_t ng = first[(size_t)n_fam];
    std::vector<std::vector<uint8_t>> seqs((size_t)ng);
    std::atomic<int64_t> next{0};
    const int nt = std::max(1, std::min<int>(n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency(), 256));
    auto work = [&]() {
        std::vector<uint8_t> anc;
        for (;;) {
            const int64_t f = next.fetch_add(1);
            if (f >= n_fam) break;
            anc.resize((size_t)lengths[f]);
            const uint64_t sa = stream_id(seed, (uint64_t)fam_idx[f], 0, 0);
            for (size_t i = 0; i < anc.size(); ++i) anc[i] = (uint8_t)(draw(sa, i) >> 62);
            for (int m = 0; m < members[f]; ++m)
                mutate(seed, (uint64_t)fam_idx[f], (uint64_t)m + 1, anc, p_lo, p_hi, n_indels, seqs[(size_t)(first[(size_t)f] + m)]);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work);
    for (auto& x : th) x.join();
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)ng + 1));
    if (!off) throw vg_error(VG_ENOMEM, "out of host memory");
    off[0] = 0;
    for (int64_t i = 0; i < ng; ++i) off[i + 1] = off[i] + (int64_t)seqs[(size_t)i].size();
    uint8_t* codes = (uint8_t*)malloc((size_t)std::max<int64_t>(off[ng], 1));
    if (!codes) { free(off); throw vg_error(VG_ENOMEM, "out of host memory"); }
    std::atomic<int64_t> nx2{0};
    auto copy = [&]() {
        for (;;) {
            const int64_t i = nx2.fetch_add(64);
            if (i >= ng) break;
            for (int64_t j = i; j < std::min(ng, i + 64); ++j)
                if (!seqs[(size_t)j].empty()) memcpy(codes + off[j], seqs[(size_t)j].data(), seqs[(size_t)j].size());
        }
    };
    th.clear();
    for (int t = 0; t < nt; ++t) th.emplace_back(copy);
    for (auto& x : th) x.join();
    *codes_out = codes; *offsets_out = off; *n_genomes = ng;
    VG_API_END
}
