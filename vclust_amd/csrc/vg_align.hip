// vg_align.hip — LZ-ANI pairwise parse on gfx950.  Replaces `lz-ani all2all`
// (vclust.py:1142-1181); restates rules R1-R9 of oracle/lz_oracle.c and is
// parity-checked against it integer for integer (and, through the golden example, against the
// reference's own ani.tsv / ani.aln.tsv).
//
// Design: one 64-lane wavefront owns one ordered pair (query -> reference).  The parse is a
// sequential left-to-right scan with data-dependent jumps, so the wave speculates: lane l
// probes query position i+l (one walk over the bucket of its msl-mer serves the seed lookup near
// the prediction and the anchor lookup, and the lane makes the R2/R3 choice between them); a
// ballot + find-first picks the first hit, which is exactly the sequential result.  Exact
// extension, the (aw, am, ar) approximate extension and the gap score (best placement of one
// indel, R7) are bit-parallel: every lane compares 32 bases (their two bit planes: three logic operations give the
// 32-bit mismatch mask), window rule and split points are evaluated on match / mismatch bit masks with ballots.
// Only integers leave.
//
// Reference side: per reference genome, RR = forward | N | reverse complement is materialised
// as bit planes (+ N mask) with ONE direct-address index built on the device: 4^msl buckets of
// entries pos | tag, tag = the bases behind the msl-mer (an anchor candidate is an entry whose tag
// equals the query's).  The strands are separate worlds: every extension, window and gap score is
// clipped to the strand of its match (the oracle's long separator).  Integer, latency-bound work:
// no MFMA, the levers are occupancy and L2 locality of the reference (tasks are grouped by
// reference and dealt to workgroups XCD-aware).
#include "vg_common.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <numeric>
#include <cstring>
#include <memory>
#include <mutex>
#include <optional>

namespace {

constexpr int RR_PAD = 128;     // mask=1 padding bases behind RR

struct ref_desc {
    int64_t rr_w;      // word offset of packed RR
    int64_t mask_w;    // word offset of RR mask
    int64_t stab;      // offset into the bucket table pool (4^msl entries: END of each bucket)
    int64_t sent;      // offset into the entry pool (one entry per RR position whose msl-mer is valid)
    int32_t L;         // forward length
    int32_t n_rr;      // 2L + 1
    int32_t genome;    // genome id
    int32_t has_n;     // reference genome contains N
    int32_t pos_bits;  // index entry = pos | tag << pos_bits
    int32_t tag_bits;  // tag = the first tag_bits / 2 bases behind the msl-mer (<= mal - msl bases, <= 14 bits, <= 32 - pos_bits)
};

// ablate: developer timing experiments only; pw_*: probe widths; weak_ratio, margin, seed_choice: the constants of the
// restatement that a handful of events of the reference's example hold (vg_lz_fit, include/vclust_gpu.h): R3's weak-seed
// ratio (3; 0 = the rule is off), the symbols a far anchor must be longer than the seed by, minus one (msl - 1), and the
// tie-break of seeds (3 = longest, then closest to the prediction; 1 = closest, then longest)
struct lz_dev_params { int mal, msl, mrd, mqd, reg, aw, am, ar; int ablate; int pw_after, pw_miss; int weak_ratio; int margin; int seed_choice; };

// ------------------------------------------------------------------ bit helpers
// Two layouts of a sequence.  (1) The genome set's 2-bit codes, 16 bases per word (vg_common.h): what the prefilter and
// the index builds read.  (2) BIT PLANES, what the parse reads -- the query's genome (vg_genomes::d_planes, made once
// per resident set by k_genome_planes) and the reference's RR: bases 32 w .. 32 w + 31 are the word pair (lo, hi) =
// (bit 0 of every code, bit 1 of every code) at words 2 w, 2 w + 1.  Two sequences differ at a base iff
// (lo ^ lo') | (hi ^ hi') has its bit set: the mismatch mask of 32 bases is ONE 32-bit word straight from three
// logic operations, and everything the parse does with it -- window counts, runs of matches, first / last set bit,
// lane-to-lane carries -- is 32-bit arithmetic (the 2-bit layout gave a 64-bit mask with every second bit unused:
// twice the VALU instructions for each of those steps, round 5).

// 32 mask bits starting at base position p
__device__ __forceinline__ uint32_t loadm32(const uint32_t* __restrict__ mk, int64_t p) {
    int64_t w = p >> 5; int sh = (int)(p & 31);
    uint64_t m = (uint64_t)mk[w] | ((uint64_t)mk[w + 1] << 32);
    return (uint32_t)(m >> sh);
}
// even-bit mask (one base per 2 bits) -> one bit per base
__device__ __forceinline__ uint32_t squeeze16(uint32_t x) {       // the 16 even bits of x -> its low 16 bits
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0x0000FFFFu;
}
__device__ __forceinline__ uint32_t squeeze(uint64_t x) {         // per 32-bit half: no 64-bit shifts
    return squeeze16((uint32_t)x) | (squeeze16((uint32_t)(x >> 32)) << 16);
}
// 32 two-bit codes -> their bit planes
struct planes32 { uint32_t lo, hi; };
__device__ __forceinline__ planes32 planes_of(uint64_t codes) { return { squeeze(codes), squeeze(codes >> 1) }; }

// 32 bases of a bit-plane array starting at base position p (p >= 0): one 16-byte load (8-byte aligned; every plane
// array has a word pair of slack), one funnel shift per plane
__device__ __forceinline__ planes32 loadp(const uint32_t* __restrict__ pl, int64_t p) {
    const uint32_t off = ((uint32_t)p >> 5) << 3; const uint32_t sh = (uint32_t)p & 31u;
    uint4 v; __builtin_memcpy(&v, (const char*)pl + off, 16);
    asm volatile("" :: "v"(v.w));
    return { __builtin_amdgcn_alignbit(v.z, v.x, sh), __builtin_amdgcn_alignbit(v.w, v.y, sh) };
}
__device__ __forceinline__ uint32_t diff32(const planes32 a, const planes32 b) { return (a.lo ^ b.lo) | (a.hi ^ b.hi); }
// mask of the base slots j in [0, 32) with lo <= j < hi
__device__ __forceinline__ uint32_t slots(int lo, int hi) {
    if (lo < 0) lo = 0; if (hi > 32) hi = 32;
    if (hi <= lo) return 0u;
    const uint32_t a = (hi == 32) ? ~0u : ((1u << hi) - 1u);
    return a & ~((1u << lo) - 1u);                       // (lo < hi <= 32)
}

struct pair_ctx {
    const uint32_t* qpl; const uint32_t* qmk; int qlen; int q_has_n;       // query: bit planes of its genome, N mask
    const uint32_t* rpl; const uint32_t* rmk; int n_rr; int L; int r_has_n;    // reference: bit planes of RR, mask
};

// mismatch mask of the 32 positions q[qp+j] vs rr[rp+j] (bit j); positions outside the query or
// outside the reference window [rlo, rhi), and N positions, are mismatches.  qp/rp may be negative or
// run past the end.  The window is one strand of RR (forward [0, L) or reverse complement
// [L+1, n_rr)): matches, extensions and gap scores never cross from one strand into the other.
__device__ __forceinline__ uint32_t mism32(const pair_ctx& c, int qp, int rp, int rlo, int rhi) {
    const int lo = max(-qp, rlo - rp); const int hi = min(c.qlen - qp, rhi - rp);
    if (lo <= 0 && hi >= 32) {
        // fast path: the whole chunk lies inside both sequences
        uint32_t mm = diff32(loadp(c.qpl, qp), loadp(c.rpl, rp));
        if (c.q_has_n) mm |= loadm32(c.qmk, qp);
        if (c.r_has_n) mm |= loadm32(c.rmk, rp);
        return mm;
    }
    const uint32_t ok = slots(lo, hi);
    if (ok == 0) return ~0u;
    const int qs = qp < 0 ? 0 : qp, rs = rp < 0 ? 0 : rp;     // clamp loads; shifted back below (a shift of 32 or more has ok == 0)
    planes32 xq = loadp(c.qpl, qs), xr = loadp(c.rpl, rs);
    if (qp < 0) { xq.lo <<= -qp; xq.hi <<= -qp; }
    if (rp < 0) { xr.lo <<= -rp; xr.hi <<= -rp; }
    uint32_t mm = diff32(xq, xr);
    if (c.q_has_n) { uint32_t sp = loadm32(c.qmk, qs); if (qp < 0) sp <<= -qp; mm |= sp; }
    if (c.r_has_n) { uint32_t sp = loadm32(c.rmk, rs); if (rp < 0) sp <<= -rp; mm |= sp; }
    return mm | ~ok;
}
// the strand of RR that holds position rp
__device__ __forceinline__ int strand_lo(const pair_ctx& c, int rp) { return rp > c.L ? c.L + 1 : 0; }
__device__ __forceinline__ int strand_hi(const pair_ctx& c, int rp) { return rp > c.L ? c.n_rr : c.L; }

// exact match length from (qp, rp), lane-local, at most cap bases examined (multiple of 32)
__device__ __forceinline__ int match_len_lane(const pair_ctx& c, int qp, int rp, int cap) {
    int l = 0;
    while (l < cap) {
        const uint32_t mm = mism32(c, qp + l, rp + l, strand_lo(c, rp), strand_hi(c, rp));
        if (mm) return l + __builtin_ctz(mm);
        l += 32;
    }
    return l;
}

// exact match length (<= 32) of the query chunk already held in registers (xq, qbad = mask of
// its unusable slots: beyond the query end or N) against rr[rp ..]: only the reference side is loaded
__device__ __forceinline__ int match_len32_q(const pair_ctx& c, const planes32 xq, uint32_t qbad, int rp) {
    uint32_t mm = diff32(xq, loadp(c.rpl, rp)) | qbad;
    if (c.r_has_n) mm |= loadm32(c.rmk, rp);
    const int hi = c.n_rr - rp; if (hi < 32) mm |= ~slots(0, hi);
    const int sj = c.L - rp; if ((unsigned)sj < 32u) mm |= 1u << sj;
    return mm ? __builtin_ctz(mm) : 32;
}

// ---- cross-lane helpers that stay off the LDS crossbar (ds_bpermute costs ~100 cycles of
// latency each, and a wave that owns a pair runs these in a dependent chain) ----------------
// value of a wave-uniform lane: v_readlane
__device__ __forceinline__ uint32_t lane32(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane); }
// lane l receives the value of lane l-1 (lane 0 receives 0): DPP wave_shr:1
__device__ __forceinline__ uint32_t lane_prev32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
// sum over the wave of a per-lane value in [0, 63]: six ballots + scalar popcounts
__device__ __forceinline__ int wave_sum(int v) {
    int t = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) t += __popcll(__ballot((v >> b) & 1)) << b;
    return t;
}

// Exact + approximate extension in one pass (R2/R3 match length + R4, or R5 to the left).
// dir = +1: positions qp+e (e = 0 is the first symbol of the match); dir = -1: positions qp-1-e.
// At most `bound` positions.  Returns the accepted length; *n_match = matching symbols inside it.
//
// Sequentially the reference takes the maximal exact run and then walks on while the last aw
// symbols hold <= am mismatches, cutting back to the end of the last run of >= ar matches.  The
// symbol that ends the exact run is a mismatch, so window counts and match runs never straddle
// the two phases, and the whole thing is the window automaton run over the mismatch mask from
// e = 0, with the exact run as a lower bound of the result.  Bit-parallel: 32 bases per lane
// (one word of mismatch bits), 2 048 per round; only mismatch positions can start a violation.
// mismatch mask of round 0 of extend(), split out so that a caller can issue the loads of several
// extensions back to back (one memory round trip instead of one per extension)
__device__ __forceinline__ uint32_t extend_mask0(const pair_ctx& c, int qp, int rp, int dir, int bound, int lane, int rlo, int rhi) {
    uint32_t mm = ~0u;
    const int e0 = 32 * lane;
    if (e0 < bound) {
        if (dir > 0) mm = mism32(c, qp + e0, rp + e0, rlo, rhi);
        else mm = __brev(mism32(c, qp - e0 - 32, rp - e0 - 32, rlo, rhi));        // slot j <-> position e0 + j
        const int rem = bound - e0; if (rem < 32) mm |= ~slots(0, rem);
    }
    return mm;
}

__device__ __forceinline__ int extend(const pair_ctx& c, const lz_dev_params& P, int qp, int rp, int dir, int bound,
                                      int lane, int* n_match, uint32_t mm_first, int rlo, int rhi) {
    int accepted = 0, matches_total = 0;
    int first_mm = -1;            // position of the first mismatch (end of the exact run)
    uint32_t carry_mm = 0;        // mismatch bits of the previous 32 positions (0 before e = 0)
    uint32_t carry_ok = 0;        // match bits of the previous 32 positions (none before e = 0)
    int base = 0;                 // first position of this round
    int cum_before = 0;           // matches in [0, base)
    const uint32_t awmask = (P.aw >= 32) ? ~0u : ((1u << P.aw) - 1u);
    if (bound <= 0) { *n_match = 0; return 0; }
    for (;;) {
        uint32_t mm = mm_first;
        if (base > 0) {
            mm = ~0u;
            const int e0 = base + 32 * lane;
            if (e0 < bound) {
                if (dir > 0) mm = mism32(c, qp + e0, rp + e0, rlo, rhi);
                else mm = __brev(mism32(c, qp - e0 - 32, rp - e0 - 32, rlo, rhi));    // slot j <-> position e0 + j
                const int rem = bound - e0; if (rem < 32) mm |= ~slots(0, rem);
            }
        }
        const unsigned long long anyb = __ballot(mm != 0);
        if (first_mm < 0 && anyb) {
            const int fl = __builtin_ctzll(anyb);
            first_mm = base + 32 * fl + __builtin_ctz(lane32(mm, fl));
        }
        uint32_t prev_mm = lane_prev32(mm); const uint32_t okb = ~mm; uint32_t prev_ok = lane_prev32(okb);
        if (lane == 0) { prev_mm = carry_mm; prev_ok = carry_ok; }
        // first violation in this lane: never if the chunk plus the aw-1 symbols before it hold <= am
        int viol = 32;
        {
            const uint32_t tail = (P.aw > 1) ? (prev_mm >> (32 - (P.aw - 1))) : 0u;
            if ((int)(__popc(mm) + __popc(tail)) > P.am) {
                uint32_t bits = mm;
                // a window holds the mismatches of the tail plus those up to the tested one: the first
                // am - |tail| mismatches of this chunk cannot push any window over am
                for (int skip = P.am - (int)__popc(tail); skip > 0; --skip) bits &= bits - 1;
                while (bits) {
                    const int j = __builtin_ctz(bits);
                    const uint32_t hi = (j == 31) ? mm : (mm & ((2u << j) - 1u));
                    int cnt;
                    if (j + 1 >= P.aw) cnt = __popc(hi & (awmask << (j + 1 - P.aw)));
                    else cnt = __popc(hi) + __popc(prev_mm >> (32 - (P.aw - 1 - j)));
                    if (cnt > P.am) { viol = j; break; }
                    bits &= bits - 1;
                }
            }
        }
        const unsigned long long vb = __ballot(viol < 32);
        if (anyb) {
            // positions ending a run of >= ar matches, strictly before the violation
            uint32_t run = okb;
            for (int t = 1; t < P.ar; ++t) run &= __builtin_amdgcn_alignbit(okb, prev_ok, (uint32_t)(32 - t));     // ({okb, prev_ok} >> (32 - t)): bit j = ok[j - t]
            const int fv = vb ? __builtin_ctzll(vb) : 64;
            const int vj = (int)lane32((uint32_t)viol, fv & 63);
            uint32_t cand = run;
            if (lane > fv) cand = 0;
            else if (lane == fv) cand &= (vj == 0) ? 0u : ((1u << vj) - 1u);
            const unsigned long long cb = __ballot(cand != 0);
            if (cb) {
                const int hl = 63 - __builtin_clzll(cb);
                const uint32_t ch = lane32(cand, hl); const uint32_t mh = lane32(mm, hl);
                const int hj = 31 - __builtin_clz(ch);
                accepted = base + 32 * hl + hj + 1;
                int tot = wave_sum(lane < hl ? 32 - __popc(mm) : 0);
                const uint32_t upto = (hj == 31) ? ~0u : ((2u << hj) - 1u);
                tot += (hj + 1) - __popc(mh & upto);
                matches_total = cum_before + tot;
            }
        } else {
            // 2 048 matches in a row
            accepted = base + 2048; matches_total = cum_before + 2048;
        }
        if (vb) break;
        if (base + 2048 >= bound) break;
        cum_before += anyb ? wave_sum(32 - __popc(mm)) : 2048;
        carry_mm = lane32(mm, 63); carry_ok = lane32(okb, 63);
        base += 2048;
    }
    if (first_mm < 0) first_mm = bound;                    // the whole range matched
    if (first_mm > bound) first_mm = bound;
    if (accepted < first_mm) { accepted = first_mm; matches_total = first_mm; }   // exact run is always taken
    if (accepted > bound) { accepted = bound; }
    *n_match = matches_total;
    return accepted;
}

__device__ __forceinline__ uint64_t low_bits64(int n) { return n >= 64 ? ~0ULL : (n <= 0 ? 0ULL : ((1ULL << n) - 1)); }

// Score of the g literals q[i-g .. i) in front of a chained match (ev_pos, len) whose predecessor ended at
// reference position pred0 (oracle/lz_oracle.c, gap rule): the literals are laid against the reference
// stretch [pred0, E), E = ev_pos + len, with ONE indel placed where it keeps most matches -- a prefix of
// `a` literals on the old diagonal (literal k <-> pred0 + k), a suffix flush with E (literal k <-> E - g + k),
// and, when the run is longer than the stretch, skip = g - (E - pred0) literals in between that match
// nothing.  Ties take the longest prefix.  Returns the matches; *pm / *sm = those of the prefix / suffix.
// Bit-parallel: lane l compares 32 literals on both diagonals; 64 split points are ranked per round by
// the 64 lanes from two 64-bit match masks (wave-uniform reads, no LDS).
__device__ __forceinline__ int gap_score(const pair_ctx& c, int i, int g, int pred0, int E, int lane, int rlo, int rhi,
                                         int* pm_out, int* sm_out) {
    const int reflen = E - pred0;
    const int skip = (reflen >= 0 && g > reflen) ? g - reflen : 0;
    const int q0 = i - g;
    // match masks: bit k of window w = literal 64 w + k matches
    uint32_t ok_o = 0, ok_n = 0;
    {
        const int e0 = 32 * lane;
        if (e0 < g) {
            const uint32_t in = (g - e0 >= 32) ? 0xffffffffu : ((1u << (g - e0)) - 1u);
            ok_o = ~mism32(c, q0 + e0, pred0 + e0, rlo, rhi) & in;
            ok_n = ~mism32(c, q0 + e0, E - g + e0, rlo, rhi) & in;
        }
    }
    int tot_n = 0;
    for (int w2 = 0; 32 * w2 < g; ++w2) tot_n += __popc(lane32(ok_n, w2));
    int best_v = -1, best_pm = 0, best_sm = 0;
    const int nbits = 32 - __builtin_clz((unsigned)g | 1u);
    int po = 0;                                   // old-diagonal matches in windows before w
    const int n_split = g - skip;                 // split points a = 0 .. n_split
    for (int w = 0; 64 * w <= n_split; ++w) {
        const uint64_t O = (uint64_t)lane32(ok_o, 2 * w) | ((uint64_t)lane32(ok_o, (2 * w + 1) & 63) << 32);
        const int a = 64 * w + lane;
        const int s0 = 64 * w + skip;             // suffix start of lane 0 of this round
        const int ws = s0 >> 6;
        const uint64_t N0 = (uint64_t)lane32(ok_n, (2 * ws) & 63) | ((uint64_t)lane32(ok_n, (2 * ws + 1) & 63) << 32);
        const uint64_t N1 = (uint64_t)lane32(ok_n, (2 * ws + 2) & 63) | ((uint64_t)lane32(ok_n, (2 * ws + 3) & 63) << 32);
        int pn0 = 0;                              // new-diagonal matches in windows before ws
        for (int w2 = 0; w2 < 2 * ws && 32 * w2 < g; ++w2) pn0 += __popc(lane32(ok_n, w2));
        const int pn1 = pn0 + __popcll(N0);
        const int sp = s0 + lane;                 // suffix start of this lane
        const bool second = (sp >> 6) > ws;
        const int before = second ? pn1 + __popcll(N1 & low_bits64(sp & 63)) : pn0 + __popcll(N0 & low_bits64(sp & 63));
        const int pre = po + __popcll(O & low_bits64(lane));
        const int suf = tot_n - before;
        // wave maximum of pre + suf over the valid split points, ties -> largest a: ballots only
        const bool valid = a <= n_split;
        const int v = pre + suf;
        unsigned long long act = __ballot(valid);
        if (act) {
            for (int b = nbits - 1; b >= 0; --b) {
                const unsigned long long m = __ballot(valid && ((v >> b) & 1)) & act;
                if (m) act = m;
            }
            const int wl = 63 - __builtin_clzll(act);
            const int vmax = (int)lane32((uint32_t)v, wl);
            if (vmax >= best_v) { best_v = vmax; best_pm = (int)lane32((uint32_t)pre, wl); best_sm = (int)lane32((uint32_t)suf, wl); }
        }
        po += __popcll(O);
    }
    *pm_out = best_pm; *sm_out = best_sm;
    return best_pm + best_sm;
}

// ------------------------------------------------------------------ index construction
// 32 symbols of RR = forward | N | reverse complement, starting at RR position 32*ch, as bit planes, straight
// from the genome's plane words (no per-base loop): the reverse-complement part is a
// bit reversal + inversion of the two planes of 32 forward bases.
__device__ __forceinline__ void rr_chunk_planes(const uint32_t* __restrict__ gpl, const uint32_t* __restrict__ gmk, int L, int64_t ch,
                                                uint32_t* lo_out, uint32_t* hi_out, uint32_t* mask_out) {
    const int64_t p0 = ch * 32;
    uint32_t lo = 0, hi = 0, mask = 0xffffffffu;
    const int64_t nf = (int64_t)L - p0;
    if (nf > 0) {
        const uint32_t xl = gpl[2 * ch], xh = gpl[2 * ch + 1], m = gmk[ch];
        if (nf >= 32) { lo = xl; hi = xh; mask = m; }
        else { const uint32_t lowm = (1u << nf) - 1u; lo = xl & lowm; hi = xh & lowm; mask = (m & lowm) | ~lowm; }
    }
    const int64_t jlo = (L + 1 - p0) > 0 ? (L + 1 - p0) : 0;
    const int64_t jhi = (2 * (int64_t)L - p0) < 31 ? (2 * (int64_t)L - p0) : 31;
    if (jlo <= jhi) {
        const int64_t fstart = 2 * (int64_t)L - p0 - 31;          // (>= -31 here)
        planes32 x; uint32_t m;
        if (fstart >= 0) { x = loadp(gpl, fstart); m = loadm32(gmk, fstart); }
        else { x = loadp(gpl, 0); x.lo <<= -fstart; x.hi <<= -fstart; m = loadm32(gmk, 0) << (-fstart); }
        // reverse complement of 32 bases: the planes bit-reversed and inverted
        const uint32_t rl = ~__brev(x.lo), rh = ~__brev(x.hi), mr = __brev(m);
        const uint32_t hi1 = (jhi == 31) ? 0xffffffffu : ((1u << (jhi + 1)) - 1);
        const uint32_t lo1 = (jlo == 0) ? 0u : ((1u << jlo) - 1);
        const uint32_t sl1 = hi1 & ~lo1;
        lo |= rl & sl1; hi |= rh & sl1;
        mask = (mask & ~sl1) | (mr & sl1);
    }
    *lo_out = lo & ~mask; *hi_out = hi & ~mask; *mask_out = mask;
}

// Tag of the index entry of RR position p (x = the bases from p on, first base in the low bits): the bases
// that follow its msl-mer.  An anchor lookup keeps the entries whose tag equals the query's: they agree with
// the query on msl + tag_bits / 2 bases without a look at the sequence (the exact length decides the rest).
// Both are read off the bit planes (xl, xh = the planes of the bases from p on): bucket = the msl low bits of each plane
// side by side, tag = the next ceil(tag_bits / 2) bits of the low plane and floor(tag_bits / 2) of the high one -- a
// bijection of the msl-mer and an injection of the (at most mal - msl) bases behind it, which is all build and probe need.
__device__ __forceinline__ uint32_t bucket_of(uint32_t xl, uint32_t xh, int msl) { const uint32_t m = (1u << msl) - 1u; return (xl & m) | ((xh & m) << msl); }
__device__ __forceinline__ uint32_t tag_of(uint32_t xl, uint32_t xh, int msl, int tag_bits) {
    const int tl = (tag_bits + 1) >> 1, th = tag_bits >> 1;
    return ((xl >> msl) & ((1u << tl) - 1u)) | (((xh >> msl) & ((1u << th) - 1u)) << tl);
}

// ---- path A (references up to 2^21 RR symbols, msl <= 7): one 1024-thread workgroup builds RR
// and both bucket indexes of a reference by counting sort in LDS.  Entries leave the LDS through a
// staging window (buckets are filled range by range), so every global store of the index is a
// coalesced copy: no 4-byte scatter to HBM (the first version wrote 10x the index size).
constexpr int LDS_TAB = 16384;      // bucket table: anchors use B <= 14 bits, seeds 4^msl <= 16384
constexpr int LDS_STAGE = 22016;    // staged entries per window (tab + stage + scan scratch fill the 160 KiB of a CU)
constexpr int TOP_BITS = 9;         // big references: entries are first dealt into 2^TOP_BITS bins
constexpr int BIG_RR = 131072;      // RR symbols from which the linear (binned) build is used
// workgroup barrier that waits for this wave's LDS traffic only: global stores stay in flight
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive scan over the 64 lanes of a wave with DPP adds (row shifts, then the two row broadcasts): six
// dependent VALU instructions instead of six LDS-crossbar round trips
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
    return v;
}

// exclusive scan of tab[0, n) by 1024 threads (16 waves, each a contiguous slice, 64 consecutive entries per
// trip: no bank conflicts); returns the total.  wtot: 16 words of scratch.
__device__ __forceinline__ uint32_t lds_scan_exclusive_waves(uint32_t* tab, int n, uint32_t* wtot) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_wave = (((n + 15) >> 4) + 63) & ~63;
    uint32_t carry = 0;
    for (int r = 0; r < per_wave; r += 64) {
        const int i = w * per_wave + r + lane;
        const uint32_t v = i < n ? tab[i] : 0u;
        const uint32_t inc = wave_scan_inclusive(v);
        if (i < n) tab[i] = carry + inc - v;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (lane == 0) wtot[w] = carry;
    lds_sync();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const uint32_t t = wtot[k]; if (k < w) off += t; total += t; }
    for (int r = 0; r < per_wave; r += 64) { const int i = w * per_wave + r + lane; if (i < n) tab[i] += off; }
    lds_sync();
    return total;
}

__global__ void __launch_bounds__(1024)
k_build_index_lds(const ref_desc* __restrict__ refs, const int* __restrict__ slot_list, int n_list,
                  const uint32_t* __restrict__ gplanes, const uint32_t* __restrict__ nmask, const int64_t* __restrict__ base_off,
                  uint32_t* __restrict__ rr_pool, uint32_t* __restrict__ mask_pool, int mal, int msl,
                  uint32_t* __restrict__ stab_pool, uint32_t* __restrict__ sent_pool,
                  uint32_t* __restrict__ scratch_pool, int64_t scratch_stride) {
    __shared__ uint32_t tab[LDS_TAB];
    __shared__ uint32_t stage[LDS_STAGE];
    __shared__ uint32_t part[1024];
    __shared__ uint32_t s_bstart[(1 << TOP_BITS) + 1];
    __shared__ int s_whi;
    // per workgroup: (bucket | tag << 18) per RR position, then the (bucket|tag, position) list by top-level bin
    uint32_t* scratch = scratch_pool + (int64_t)blockIdx.x * scratch_stride * 3;
    uint2* binned = reinterpret_cast<uint2*>(scratch + scratch_stride);
    (void)mal;
    for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
        const ref_desc rd = refs[slot_list[li]];
        const int64_t g0 = base_off[rd.genome];
        const uint32_t* gpk = gplanes + (g0 >> 4); const uint32_t* gmk = nmask + (g0 >> 5);
        uint32_t* pk = rr_pool + rd.rr_w; uint32_t* mk = mask_pool + rd.mask_w;
        const int64_t chunks = ((int64_t)rd.n_rr + RR_PAD + 31) / 32 + 2;
        for (int64_t ch = threadIdx.x; ch < chunks; ch += blockDim.x) {
            uint32_t lo, hi, m;
            rr_chunk_planes(gpk, gmk, rd.L, ch, &lo, &hi, &m);
            pk[2 * ch] = lo; pk[2 * ch + 1] = hi; mk[ch] = m;
        }
        __threadfence_block();
        __syncthreads();
        {   // one index: msl-mer buckets, entries tagged with the bases that follow
            const int nbits = 2 * msl;
            const int w = msl;
            uint32_t* gtab = stab_pool + rd.stab;
            uint32_t* gent = sent_pool + rd.sent;
            const bool big = rd.n_rr >= BIG_RR || nbits > 14;
            // pass 0: bucket (and tag) of every position -> scratch; sizes of the buckets (plain path) or
            // of the 512 top-level bins (big path).  4 consecutive positions out of one 128-bit window.
            const int tsh = nbits > TOP_BITS ? nbits - TOP_BITS : 0;             // bucket -> top-level bin
            const int n_cnt = big ? (1 << (nbits - tsh)) : (1 << nbits);
            for (int i = threadIdx.x; i < n_cnt; i += blockDim.x) tab[i] = 0;
            lds_sync();
            const int n4 = (rd.n_rr + 3) & ~3;
            // (the msl + tag bases of a position are <= 32 bits: one funnel shift over two words serves each of four
            // consecutive positions; the loads of four trips are issued together)
            for (int pb = 4 * threadIdx.x; pb < n4; pb += 16 * blockDim.x) {
                uint4 w4[4]; uint32_t m0[4], m1[4];              // (lo, hi) of the chunk of p0 and of the next one
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p0 = pb + u * 4 * (int)blockDim.x;
                    const bool in = p0 < n4;
                    w4[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (in) __builtin_memcpy(&w4[u], pk + 2 * (p0 >> 5), 16);
                    m0[u] = in ? mk[p0 >> 5] : ~0u; m1[u] = in ? mk[(p0 >> 5) + 1] : ~0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p0 = pb + u * 4 * (int)blockDim.x;
                    if (p0 >= n4) continue;
                    const uint32_t sh = (uint32_t)(p0 & 31);
                    // (one funnel shift per plane and mask serves the four positions: bits 0 .. 3 + 7 + 7 from p0 on)
                    const uint32_t XL = __builtin_amdgcn_alignbit(w4[u].z, w4[u].x, sh), XH = __builtin_amdgcn_alignbit(w4[u].w, w4[u].y, sh);
                    const uint32_t XM = __builtin_amdgcn_alignbit(m1[u], m0[u], sh);
                    uint32_t out[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p = p0 + j;
                        uint32_t bt = 0xffffffffu;
                        if (p + w <= rd.n_rr && ((XM >> j) & ((1u << w) - 1u)) == 0) {
                            const uint32_t xl = XL >> j, xh = XH >> j;
                            bt = bucket_of(xl, xh, msl) | (tag_of(xl, xh, msl, rd.tag_bits) << 18);
                            atomicAdd(&tab[(bt & 0x3ffffu) >> (big ? tsh : 0)], 1u);
                        }
                        out[j] = bt;
                    }
                    *reinterpret_cast<uint4*>(&scratch[p0]) = make_uint4(out[0], out[1], out[2], out[3]);
                }
            }
            __threadfence_block();
            __syncthreads();
            if (!big) {
                const uint32_t total = lds_scan_exclusive_waves(tab, n_cnt, part);     // tab[b] = first slot of bucket b
                // fill, one window of whole buckets holding <= LDS_STAGE entries at a time.  Buckets at or
                // beyond the window start are still untouched, so end(b) = tab[b + 1] (or the total).
                // Every window re-reads the scratch: fine for a handful of windows (< BIG_RR symbols).
                const int nb = n_cnt;
                int w_lo = 0;
                while (w_lo < nb) {
                    const uint32_t base = tab[w_lo];
                    if (threadIdx.x == 0) s_whi = w_lo + 1;
                    lds_sync();
                    int best = w_lo + 1;
                    for (int b2 = w_lo + threadIdx.x; b2 < nb; b2 += blockDim.x) {
                        const uint32_t e2 = (b2 + 1 < nb) ? tab[b2 + 1] : total;
                        if (e2 - base <= (uint32_t)LDS_STAGE) best = b2 + 1; else break;     // ends ascend
                    }
                    atomicMax(&s_whi, best);
                    lds_sync();
                    const int w_hi = s_whi;
                    const uint32_t e_lo = (w_lo + 1 < nb) ? tab[w_lo + 1] : total;
                    const bool direct = (e_lo - base > (uint32_t)LDS_STAGE);   // one bucket larger than the stage
                    lds_sync();
                    // four independent 16-byte loads per thread and trip: the loop is latency bound
                    for (int pb = 4 * threadIdx.x; pb < n4; pb += 16 * blockDim.x) {
                        uint4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int p0 = pb + u * 4 * (int)blockDim.x;
                            v[u] = (p0 < n4) ? *reinterpret_cast<const uint4*>(&scratch[p0]) : make_uint4(~0u, ~0u, ~0u, ~0u);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int p0 = pb + u * 4 * (int)blockDim.x;
                            const uint32_t bts[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t bt = bts[j];
                                const int b2 = (int)(bt & 0x3ffffu);
                                if (bt == 0xffffffffu || b2 < w_lo || b2 >= w_hi) continue;
                                const uint32_t ent = (uint32_t)(p0 + j) | ((bt >> 18) << rd.pos_bits);
                                const uint32_t slot = atomicAdd(&tab[b2], 1u);
                                if (direct) gent[slot] = ent; else stage[slot - base] = ent;
                            }
                        }
                    }
                    lds_sync();
                    if (!direct) {
                        const uint32_t cnt = tab[w_hi - 1] - base;   // cursor of the last bucket == its end
                        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) gent[base + i] = stage[i];
                    }
                    lds_sync();
                    w_lo = w_hi;
                }
                for (int i = threadIdx.x; i < nb; i += blockDim.x) gtab[i] = tab[i];   // END of every bucket
                lds_sync();
                continue;
            }
            // ---- big references: a scratch re-read per window would be quadratic, so the entries are first
            // dealt into 512 top-level bins (contiguous bucket ranges) in a second scratch; a window is then
            // a contiguous slice of that list and every pass is linear in the reference.
            const int n_bins = n_cnt;
            const uint32_t total = lds_scan_exclusive_waves(tab, n_bins, part);
            for (int i = threadIdx.x; i <= n_bins; i += blockDim.x) { s_bstart[i] = i < n_bins ? tab[i] : total; }
            lds_sync();
            for (int pb = 4 * threadIdx.x; pb < n4; pb += 16 * blockDim.x) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p0 = pb + u * 4 * (int)blockDim.x;
                    v[u] = (p0 < n4) ? *reinterpret_cast<const uint4*>(&scratch[p0]) : make_uint4(~0u, ~0u, ~0u, ~0u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p0 = pb + u * 4 * (int)blockDim.x;
                    const uint32_t bts[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (bts[j] == 0xffffffffu) continue;
                        const uint32_t slot = atomicAdd(&tab[(bts[j] & 0x3ffffu) >> tsh], 1u);
                        binned[slot] = make_uint2(bts[j], (uint32_t)(p0 + j));
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            const int per_bin = 1 << tsh;                          // buckets per top-level bin
            int w_lo = 0;
            while (w_lo < n_bins) {
                // window = as many whole bins as fit the stage (entries) and the table (buckets)
                if (threadIdx.x == 0) {
                    int hi = w_lo + 1;
                    while (hi < n_bins && s_bstart[hi + 1] - s_bstart[w_lo] <= (uint32_t)LDS_STAGE && (hi + 1 - w_lo) * per_bin <= LDS_TAB) ++hi;
                    s_whi = hi;
                }
                lds_sync();
                const int w_hi = s_whi;
                const uint32_t base = s_bstart[w_lo], cnt = s_bstart[w_hi] - base;
                const int nbw = (w_hi - w_lo) * per_bin; const uint32_t b_lo = (uint32_t)w_lo << tsh;
                const bool direct = cnt > (uint32_t)LDS_STAGE;     // a single bin larger than the stage
                for (int i = threadIdx.x; i < nbw; i += blockDim.x) tab[i] = 0;
                lds_sync();
                // (four independent loads per thread and trip in both passes over the window: one workgroup per CU,
                // nothing else hides the latency of a load per trip)
                for (uint32_t i0 = threadIdx.x; i0 < cnt; i0 += 4 * blockDim.x) {
                    uint32_t bx[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + u * blockDim.x; bx[u] = i < cnt ? binned[base + i].x : 0xffffffffu; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (i0 + u * blockDim.x < cnt) atomicAdd(&tab[(bx[u] & 0x3ffffu) - b_lo], 1u);
                }
                lds_sync();
                lds_scan_exclusive_waves(tab, nbw, part);
                for (uint32_t i0 = threadIdx.x; i0 < cnt; i0 += 4 * blockDim.x) {
                    uint2 e4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + u * blockDim.x; e4[u] = i < cnt ? binned[base + i] : make_uint2(0u, 0u); }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (i0 + u * blockDim.x >= cnt) continue;
                        const uint2 e = e4[u];
                        const uint32_t ent = e.y | ((e.x >> 18) << rd.pos_bits);
                        const uint32_t slot = atomicAdd(&tab[(e.x & 0x3ffffu) - b_lo], 1u);
                        if (direct) gent[base + slot] = ent; else stage[slot] = ent;
                    }
                }
                lds_sync();
                if (!direct) for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) gent[base + i] = stage[i];
                for (int i = threadIdx.x; i < nbw; i += blockDim.x) gtab[b_lo + i] = base + tab[i];       // END of every bucket
                lds_sync();
                w_lo = w_hi;
            }
        }
    }
}

// ---- path A0 (references up to REG_MAX_RR symbols -- genomes below 49 kb --, msl <= 7): the same counting sort
// with every position's (bucket, tag) and then (slot, tag) held in REGISTERS (96 per thread: three quarters of
// the CU's register file) and the genome / RR held in the LDS, so that after the one coalesced read of the
// genome nothing is read from global memory again.  Slots are handed out once (LDS atomics); once the bucket
// table has been written out the whole LDS becomes the staging buffer, and the entries leave through two or
// three fixed slot-range windows (no atomics, no window search, no scratch).  Barriers order the LDS only
// (lds_sync), so the stores of one window -- and of one reference -- drain while the next is being prepared.
constexpr int REG_IT = 24;                              // trips of 4 positions x 1024 threads
constexpr int REG_MAX_RR = 4 * 1024 * REG_IT - 256;     // 98 048 RR symbols
constexpr int REG_RR_WORDS = REG_MAX_RR / 16 + 32;
constexpr int REG_MK_WORDS = REG_MAX_RR / 32 + 16;
constexpr int REG_GEN_WORDS = 12288;                    // genome words + mask + scan scratch (before the windows)
constexpr int REG_LDS_WORDS = LDS_TAB + REG_GEN_WORDS + REG_RR_WORDS + REG_MK_WORDS;
constexpr int REG_STAGE = REG_LDS_WORDS / 1024 * 1024;  // entries per staging window: the whole LDS block
constexpr int REG_STAGE0 = (REG_LDS_WORDS - LDS_TAB) / 1024 * 1024;   // slots staged while the table is still live
constexpr int REG_SLOT_BITS = 17;                       // slot < n_rr < 2^17; tag (<= 14 bits) above it

// RR, bucket table and entries of ONE reference by the 1 024 threads of a workgroup (lds: REG_LDS_WORDS words).
// IT: trips of 4 positions x 1 024 threads the registers hold -- 24 (references up to REG_MAX_RR symbols), 20 (the 40 kb
// genomes of the benchmark sets: 16 registers and a sixth of every pass less), 16, 12, 8 or 4 (contigs of a few kb).
template <int IT>
__device__ __forceinline__ void build_ref_reg(uint32_t* const lds, const ref_desc rd,
                  const uint32_t* __restrict__ gplanes, const uint32_t* __restrict__ nmask, const int64_t* __restrict__ base_off,
                  uint32_t* __restrict__ rr_pool, uint32_t* __restrict__ mask_pool, int msl,
                  uint32_t* __restrict__ stab_pool, uint32_t* __restrict__ sent_pool) {
    uint32_t* const tab = lds;                                   // bucket table
    uint32_t* const gw = lds + LDS_TAB;                          // genome words, mask, scan scratch
    uint32_t* const gm = gw + 3328; uint32_t* const wtot = gw + 8192;
    uint32_t* const s_rr = gw + REG_GEN_WORDS;                   // RR = fwd | N | rc of the reference
    uint32_t* const s_mk = s_rr + REG_RR_WORDS;
    uint32_t* const stage = lds;                                 // the windows use all of it
    const int nb = 1 << (2 * msl);
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                            // per-position constants are recomputed per reference, not kept (and spilled) across the loop
        const int64_t g0 = base_off[rd.genome];
        const uint32_t* gpk = gplanes + (g0 >> 4); const uint32_t* gmk = nmask + (g0 >> 5);
        uint32_t* pk = rr_pool + rd.rr_w; uint32_t* mk = mask_pool + rd.mask_w;
        uint32_t* gtab = stab_pool + rd.stab; uint32_t* gent = sent_pool + rd.sent;
        const int n_gw = (rd.L >> 4) + 4, n_gm = (rd.L >> 5) + 3;
        for (int i = tid; i < n_gw; i += 1024) gw[i] = gpk[i];
        for (int i = tid; i < n_gm; i += 1024) gm[i] = gmk[i];
        for (int i = tid; i < nb; i += 1024) tab[i] = 0;
        lds_sync();
        const int chunks = (rd.n_rr + RR_PAD + 31) / 32 + 2;
        for (int ch = tid; ch < chunks; ch += 1024) {
            uint32_t lo, hi, m;
            rr_chunk_planes(gw, gm, rd.L, ch, &lo, &hi, &m);
            pk[2 * ch] = lo; pk[2 * ch + 1] = hi; mk[ch] = m;
            s_rr[2 * ch] = lo; s_rr[2 * ch + 1] = hi; s_mk[ch] = m;
        }
        lds_sync();
        // pass 0: bucket | tag << 18 of every position (4 consecutive positions out of one 128-bit window), bucket sizes
        uint32_t reg[IT][4];
        const int n4 = (rd.n_rr + 3) & ~3;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int p0 = 4 * (tid + 1024 * it);
            reg[it][0] = reg[it][1] = reg[it][2] = reg[it][3] = 0xffffffffu;
            if (p0 < n4) {
                // the msl + tag bases of a position are <= 32 bits: one funnel shift over two words serves each of the
                // four positions (p0 is a multiple of 4: their bit offsets 2 * (p0 & 15) + 2 j stay below 32)
                const int wi = 2 * (p0 >> 5); const uint32_t sh = (uint32_t)(p0 & 31);
                // (msl + tag bases of the four positions = bits 0 .. 3 + 7 + 7 of the planes from p0 on: one funnel shift per
                // plane and one for the mask serve all four)
                const uint32_t XL = __builtin_amdgcn_alignbit(s_rr[wi + 2], s_rr[wi], sh), XH = __builtin_amdgcn_alignbit(s_rr[wi + 3], s_rr[wi + 1], sh);
                const uint32_t XM = __builtin_amdgcn_alignbit(s_mk[(p0 >> 5) + 1], s_mk[p0 >> 5], sh);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = p0 + j;
                    if (p + msl <= rd.n_rr && ((XM >> j) & ((1u << msl) - 1u)) == 0) {
                        const uint32_t xl = XL >> j, xh = XH >> j;
                        const uint32_t bt = bucket_of(xl, xh, msl) | (tag_of(xl, xh, msl, rd.tag_bits) << 18);
                        atomicAdd(&tab[bt & 0x3ffffu], 1u);
                        reg[it][j] = bt;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_sync();
        const uint32_t total = lds_scan_exclusive_waves(tab, nb, wtot);     // tab[b] = first slot of bucket b
        // every entry takes its slot (an invalid position keeps 0xffffffff: its slot field lies beyond every
        // window); afterwards tab[b] = END of bucket b.  The entries of the first REG_STAGE0 slots are staged
        // right away, in the LDS that the table does not occupy (one window sweep less).
        uint32_t* const stage0 = lds + LDS_TAB;
        {
            uint32_t t4 = 4u * (uint32_t)tid;
            asm volatile("" : "+v"(t4));
#pragma unroll
            for (int it = 0; it < IT; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t bt = reg[it][j];
                    if (bt != 0xffffffffu) {
                        const uint32_t slot = atomicAdd(&tab[bt & 0x3ffffu], 1u);
                        reg[it][j] = slot | ((bt >> 18) << REG_SLOT_BITS);
                        if (slot < (uint32_t)REG_STAGE0) stage0[slot] = (t4 + (uint32_t)(4096 * it + j)) | ((bt >> 18) << rd.pos_bits);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_sync();
        // (copies out of the LDS: four words per lane and store)
        auto copy_out = [&](uint32_t* dst, const uint32_t* src, uint32_t n) {
            for (uint32_t i = 4u * (uint32_t)tid; i < n; i += 4096u) {
                if (i + 4 <= n) { const uint4 v = make_uint4(src[i], src[i + 1], src[i + 2], src[i + 3]); __builtin_memcpy(dst + i, &v, 16); }
                else for (uint32_t j = i; j < n; ++j) dst[j] = src[j];
            }
        };
        copy_out(gtab, tab, (uint32_t)nb);
        copy_out(gent, stage0, min((uint32_t)REG_STAGE0, total));
        lds_sync();
        for (uint32_t base = REG_STAGE0; base < total; base += REG_STAGE) {
            uint32_t t4 = 4u * (uint32_t)tid;
            asm volatile("" : "+v"(t4));                         // nothing below is hoisted out of the window loop (it would triple the live registers)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t v = reg[it][j];
                    asm volatile("" : "+v"(v));
                    const uint32_t d = (v & ((1u << REG_SLOT_BITS) - 1u)) - base;
                    if (d < (uint32_t)REG_STAGE)
                        stage[d] = (t4 + (uint32_t)(4096 * it + j)) | ((v >> REG_SLOT_BITS) << rd.pos_bits);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_sync();
            copy_out(gent + base, stage, min((uint32_t)REG_STAGE, total - base));
            lds_sync();
        }
    }
}
template <int IT>
__global__ void __launch_bounds__(1024)
k_build_index_reg(const ref_desc* __restrict__ refs, const int* __restrict__ slot_list, int n_list,
                  const uint32_t* __restrict__ gplanes, const uint32_t* __restrict__ nmask, const int64_t* __restrict__ base_off,
                  uint32_t* __restrict__ rr_pool, uint32_t* __restrict__ mask_pool, int msl,
                  uint32_t* __restrict__ stab_pool, uint32_t* __restrict__ sent_pool) {
    __shared__ uint32_t lds[REG_LDS_WORDS];
    for (int li = blockIdx.x; li < n_list; li += gridDim.x)
        build_ref_reg<IT>(lds, refs[slot_list[li]], gplanes, nmask, base_off, rr_pool, mask_pool, msl, stab_pool, sent_pool);
}

// ---- path A1 (references of REG_MAX_RR .. MID_MAX_RR symbols -- genomes of 49 .. 262 kb --, msl <= 7): the positions do
// not fit the registers, and k_build_index_lds parks every position's (bucket, tag) in a global scratch that each
// staging window reads again (30 bytes of traffic per RR symbol).  Here NOTHING is parked: RR is written once (2 bits
// per symbol: it stays in the L2) and every pass recomputes (bucket, tag) from it.  Pass 0 counts the buckets (table in
// the LDS); the entries then leave bucket range by bucket range -- a window is the longest run of whole buckets whose
// entries fit the staging part of the LDS: the table entry of a bucket of the window is its cursor (LDS atomics), the
// staged entries go out as one coalesced copy.  5 bytes per RR symbol reach the HBM.
constexpr int MID_MAX_RR = 1 << 19;
constexpr int MID_STAGE = 23552;                        // staged entries per window (92 KiB beside the 64 KiB table)
__global__ void __launch_bounds__(1024)
k_build_index_mid(const ref_desc* __restrict__ refs, const int* __restrict__ slot_list, int n_list,
                  const uint32_t* __restrict__ gplanes, const uint32_t* __restrict__ nmask, const int64_t* __restrict__ base_off,
                  uint32_t* __restrict__ rr_pool, uint32_t* __restrict__ mask_pool, int msl,
                  uint32_t* __restrict__ stab_pool, uint32_t* __restrict__ sent_pool) {
    __shared__ uint32_t tab[LDS_TAB];
    __shared__ uint32_t stage[MID_STAGE];
    __shared__ uint32_t wtot[16];
    const int nb = 1 << (2 * msl);
    const int tid = threadIdx.x;
    for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
        const ref_desc rd = refs[slot_list[li]];
        const int64_t g0 = base_off[rd.genome];
        const uint32_t* gpk = gplanes + (g0 >> 4); const uint32_t* gmk = nmask + (g0 >> 5);
        uint32_t* pk = rr_pool + rd.rr_w; uint32_t* mk = mask_pool + rd.mask_w;
        uint32_t* gtab = stab_pool + rd.stab; uint32_t* gent = sent_pool + rd.sent;
        const int chunks = (rd.n_rr + RR_PAD + 31) / 32 + 2;
        for (int ch = tid; ch < chunks; ch += 1024) {
            uint32_t lo, hi, m;
            rr_chunk_planes(gpk, gmk, rd.L, ch, &lo, &hi, &m);
            pk[2 * ch] = lo; pk[2 * ch + 1] = hi; mk[ch] = m;
        }
        for (int i = tid; i < nb; i += 1024) tab[i] = 0;
        __threadfence_block();
        __syncthreads();                                          // RR is read back below (this workgroup's own stores)
        const int n16 = (rd.n_rr + 15) & ~15;
        // a thread takes 16 positions per trip: one RR word, its successor and the N mask of the stretch.  The words of
        // the NEXT trip are asked for before this trip's are used (the passes are latency-bound: 16 waves per CU).
        // (w.x, w.y = the planes (lo, hi) from p0 on: p0 & 31 is 0 or 16, and the sixteen positions need the bits up to
        // 15 + msl + tag <= 15 + 7 + 7 of each plane -- one word each, so position j's fields are plain bit fields at j)
        auto load16 = [&](int p0, uint4& w, uint32_t& ok) {
            __builtin_memcpy(&w, pk + 2 * (p0 >> 5), 16);
            if (p0 & 16) { w.x = __builtin_amdgcn_alignbit(w.z, w.x, 16u); w.y = __builtin_amdgcn_alignbit(w.w, w.y, 16u); }
            const uint32_t m = __builtin_amdgcn_alignbit(mk[(p0 >> 5) + 1], mk[p0 >> 5], (uint32_t)(p0 & 31));
            uint32_t bad = m;                                     // bit j: an N among the symbols j .. j + msl - 1
            for (int q = 1; q < msl; ++q) bad |= m >> q;
            ok = ~bad & 0xffffu;
            const int last = rd.n_rr - msl - p0;                  // the last j at which a whole msl-mer starts
            if (last < 15) ok &= last < 0 ? 0u : (2u << last) - 1u;
        };
        // pass 0: bucket sizes
        {
            int p0 = 16 * tid; uint4 w = make_uint4(0u, 0u, 0u, 0u); uint32_t ok = 0;
            if (p0 < n16) load16(p0, w, ok);
            while (p0 < n16) {
                const int pn = p0 + 16384; uint4 a = make_uint4(0u, 0u, 0u, 0u); uint32_t aok = 0;
                if (pn < n16) load16(pn, a, aok);
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if ((ok >> j) & 1u) atomicAdd(&tab[bucket_of(w.x >> j, w.y >> j, msl)], 1u);
                p0 = pn; w = a; ok = aok;
            }
        }
        lds_sync();
        const uint32_t total = lds_scan_exclusive_waves(tab, nb, wtot);     // tab[b] = first slot of bucket b
        // windows of whole buckets; tab[b] of a bucket not yet taken is still its START
        for (int b_lo = 0; b_lo < nb;) {
            const uint32_t base = tab[b_lo];
            // largest b_hi in (b_lo, nb] with (start of b_hi, or the total) - base <= MID_STAGE
            int lo = b_lo, hi = nb;                               // invariant: the buckets [b_lo, lo) fit
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; const uint32_t e = mid < nb ? tab[mid] : total; if (e - base <= (uint32_t)MID_STAGE) lo = mid; else hi = mid - 1; }
            const bool direct = lo == b_lo;                       // ONE bucket beyond the staging window: its entries go straight out
            // A window that starts and ends where the HIGH plane's field changes (bucket = lo field | hi field << msl) is
            // tested on that field alone -- three instructions per position and pass instead of five.  From an aligned
            // start the window is cut back to the last such boundary it reaches; a group of 2^msl buckets too large
            // for one window is taken bucket by bucket up to its end, which is a boundary again.
            const int gmask = (1 << msl) - 1;
            int b_hi = direct ? b_lo + 1 : lo;
            if (!direct) {
                if ((b_lo & gmask) == 0) { if ((b_hi & ~gmask) > b_lo) b_hi &= ~gmask; }
                else b_hi = min(b_hi, (b_lo | gmask) + 1);
            }
            const bool by_hi = ((b_lo | b_hi) & gmask) == 0;
            const uint32_t h_lo = (uint32_t)(b_lo >> msl), h_w = (uint32_t)((b_hi - b_lo) >> msl);
            const uint32_t w_end = b_hi < nb ? tab[b_hi] : total;
            const uint32_t width = (uint32_t)(b_hi - b_lo);
            lds_sync();                                           // (everybody has read the table before its cursors move)
            if (w_end > base) {
                int p0 = 16 * tid; uint4 w = make_uint4(0u, 0u, 0u, 0u); uint32_t ok = 0;
                if (p0 < n16) load16(p0, w, ok);
                while (p0 < n16) {
                    const int pn = p0 + 16384; uint4 a = make_uint4(0u, 0u, 0u, 0u); uint32_t aok = 0;
                    if (pn < n16) load16(pn, a, aok);
                    uint32_t hits = 0;
                    if (by_hi) {
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            if (((w.y >> j) & (uint32_t)gmask) - h_lo < h_w) hits |= 1u << j;
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            if (bucket_of(w.x >> j, w.y >> j, msl) - (uint32_t)b_lo < width) hits |= 1u << j;
                    }
                    hits &= ok;
                    while (hits) {                                // two hits per trip: both cursors are asked for before either entry is stored
                        const int j1 = __ffs(hits) - 1; hits &= hits - 1;
                        const int j2 = hits ? __ffs(hits) - 1 : -1; hits &= hits - 1;     // (0 & anything stays 0)
                        const uint32_t xl1 = w.x >> j1, xh1 = w.y >> j1;
                        const uint32_t s1 = atomicAdd(&tab[bucket_of(xl1, xh1, msl)], 1u);
                        uint32_t xl2 = 0, xh2 = 0, s2 = 0;
                        if (j2 >= 0) {
                            xl2 = w.x >> j2; xh2 = w.y >> j2;
                            s2 = atomicAdd(&tab[bucket_of(xl2, xh2, msl)], 1u);
                        }
                        const uint32_t e1 = (uint32_t)(p0 + j1) | (tag_of(xl1, xh1, msl, rd.tag_bits) << rd.pos_bits);
                        if (direct) gent[s1] = e1; else stage[s1 - base] = e1;
                        if (j2 >= 0) {
                            const uint32_t e2 = (uint32_t)(p0 + j2) | (tag_of(xl2, xh2, msl, rd.tag_bits) << rd.pos_bits);
                            if (direct) gent[s2] = e2; else stage[s2 - base] = e2;
                        }
                    }
                    p0 = pn; w = a; ok = aok;
                }
                lds_sync();
                if (!direct) {
                    const uint32_t n = w_end - base;
                    for (uint32_t i = 4u * (uint32_t)tid; i < n; i += 4096u) {
                        if (i + 4 <= n && ((base + i) & 3u) == 0) { const uint4 v = make_uint4(stage[i], stage[i + 1], stage[i + 2], stage[i + 3]); __builtin_memcpy(gent + base + i, &v, 16); }
                        else for (uint32_t j = i; j < min(n, i + 4); ++j) gent[base + j] = stage[j];
                    }
                    lds_sync();
                }
            }
            b_lo = b_hi;
        }
        // every cursor now stands at the END of its bucket: the table the parse reads
        for (uint32_t i = 4u * (uint32_t)tid; i < (uint32_t)nb; i += 4096u) { const uint4 v = make_uint4(tab[i], tab[i + 1], tab[i + 2], tab[i + 3]); __builtin_memcpy(gtab + i, &v, 16); }
        lds_sync();
    }
}

// ---- path B (large references / long seeds): global-memory counting sort
__global__ void __launch_bounds__(256)
k_build_rr(const ref_desc* __restrict__ refs, const int* __restrict__ slot_list, int n_list,
           const int64_t* __restrict__ chunk_off /* n_list+1, in 32-base chunks */,
           const uint32_t* __restrict__ gplanes, const uint32_t* __restrict__ nmask, const int64_t* __restrict__ base_off,
           uint32_t* __restrict__ rr_pool, uint32_t* __restrict__ mask_pool) {
    const int64_t total = chunk_off[n_list];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_list - 1;
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (chunk_off[mid] <= t) lo = mid; else hi = mid - 1; }
        const ref_desc rd = refs[slot_list[lo]];
        const int64_t ch = t - chunk_off[lo];
        const int64_t g0 = base_off[rd.genome];
        uint32_t pl_lo, pl_hi, m;
        rr_chunk_planes(gplanes + (g0 >> 4), nmask + (g0 >> 5), rd.L, ch, &pl_lo, &pl_hi, &m);
        rr_pool[rd.rr_w + 2 * ch] = pl_lo; rr_pool[rd.rr_w + 2 * ch + 1] = pl_hi;
        mask_pool[rd.mask_w + ch] = m;
    }
}

// count (fill == 0) or place (fill == 1) the anchor / seed entries of every RR position
__global__ void __launch_bounds__(256)
k_index_pass(const ref_desc* __restrict__ refs, const int* __restrict__ slot_list, int n_list, const int64_t* __restrict__ chunk_off,
             const uint32_t* __restrict__ rr_pool, const uint32_t* __restrict__ mask_pool, int mal, int msl, int fill,
             uint32_t* __restrict__ stab_pool,
             uint32_t* __restrict__ sent_pool) {
    const int64_t total = chunk_off[n_list] * 32;
    (void)mal;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t chunk = t >> 5;
        int lo = 0, hi = n_list - 1;
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (chunk_off[mid] <= chunk) lo = mid; else hi = mid - 1; }
        const ref_desc rd = refs[slot_list[lo]];
        const int64_t p = t - chunk_off[lo] * 32;
        if (p >= rd.n_rr) continue;
        const uint32_t* pk = rr_pool + rd.rr_w; const uint32_t* mk = mask_pool + rd.mask_w;
        const planes32 x = loadp(pk, p);
        uint64_t m = (uint64_t)mk[p >> 5] | ((uint64_t)mk[(p >> 5) + 1] << 32);
        m >>= (p & 31);
        if (p + msl <= rd.n_rr && (m & ((1ULL << msl) - 1)) == 0) {
            uint32_t b = bucket_of(x.lo, x.hi, msl);
            uint32_t slot = atomicAdd(&stab_pool[rd.stab + b], 1u);
            if (fill) sent_pool[rd.sent + slot] = (uint32_t)p | (tag_of(x.lo, x.hi, msl, rd.tag_bits) << rd.pos_bits);
        }
    }
}

// exclusive scan of each bucket table, one workgroup per (reference, table)
__global__ void __launch_bounds__(256)
k_scan_tables(const ref_desc* __restrict__ refs, const int* __restrict__ slot_list, int msl,
              uint32_t* __restrict__ stab_pool) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    const ref_desc rd = refs[slot_list[blockIdx.x]];
    uint32_t* tab = stab_pool + rd.stab;                    // one table per reference: the msl-mer buckets
    const int64_t n = 1LL << (2 * msl);
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 256 * 8) {
        uint32_t v[8]; uint32_t s = 0;
        int64_t i0 = base + (int64_t)threadIdx.x * 8;
        for (int j = 0; j < 8; ++j) { v[j] = (i0 + j < n) ? tab[i0 + j] : 0; s += v[j]; }
        part[threadIdx.x] = s;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            uint32_t add = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        uint32_t excl = part[threadIdx.x] - s + carry;
        for (int j = 0; j < 8; ++j) { if (i0 + j < n) tab[i0 + j] = excl; excl += v[j]; }
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
}

// ------------------------------------------------------------------ the parse
struct task_dev { uint32_t q, r_slot, out_idx, pad; };

// One wave parses one (query, reference) task (S == 1), or the S waves of a workgroup share it
// (S > 1, "segments"): wave w starts a speculative parse at query position w * seg_len with an empty
// state while wave w - 1 is still busy with its own segment.  Right after an event (a match placed
// at query i, reference j) the scan state is a function of (i, j) alone, so every wave logs its
// events together with its running sums; a wave that runs past the end of its segment stops as soon
// as one of its own events equals a logged event of the wave that owns that stretch: from there on
// the two parses are the same parse.  The chain of such hand-overs is followed from wave 0 and the
// partial sums are added up; the result is identical to the single-wave parse, only the critical
// path is ~S times shorter.
constexpr int SEG_LOG_CAP = 250;          // (4 x 250 records + the hand-over words: 20 KiB of LDS per workgroup, eight workgroups per CU)
constexpr int PW_AFTER_EVENT = 32;
struct seg_rec { int i_ev, ev_pos; uint32_t VM, VA, VN; };     // VN: bit 31 = open region already spans >= reg
// --out-aln in ONE parse: a wave writes its kept regions into chunks of RCHUNK records it takes from a global cursor
// (one atomic per chunk); a record carries the task's place in the sorted list and the region's number inside the task,
// so a placing pass (k_regions_place) moves it to its final slot once the rows -- hence every task's region count -- are
// known.  The unused tail of a wave's last chunk is marked.  A cursor beyond the arena's capacity = nothing was written
// there: the host repeats the batch with an arena of exactly the size the cursor reports (the count of chunks a parse
// takes is a function of its rows).
constexpr int FUSED_SLOTS = 16384;         // EXPERIMENT (VG_LZ_INDEX=fused): 4^msl slots of eight words per reference (msl = 7)
constexpr int RCHUNK = 8;
struct region_rec { uint32_t task; int32_t qstart, qend, rstart, rend, n_match; uint32_t k, t; };     // 32 bytes; task = ~0u: unused slot
static_assert(sizeof(region_rec) == 32, "arena record");

#define PARSE_ARGS \
    const task_dev* __restrict__ tasks, int64_t n_tasks, const ref_desc* __restrict__ refs, \
    const uint32_t* __restrict__ planes, const uint32_t* __restrict__ nmask, const int64_t* __restrict__ base_off, \
    const int64_t* __restrict__ glen, const uint8_t* __restrict__ g_has_n, \
    const uint32_t* __restrict__ rr_pool, const uint32_t* __restrict__ mask_pool, \
    const uint32_t* __restrict__ stab_pool, const uint32_t* __restrict__ sent_pool, \
    lz_dev_params P, vg_pair_stat* __restrict__ stats, \
    region_rec* __restrict__ arena, unsigned long long* __restrict__ arena_cursor, unsigned long long arena_cap, \
    const uint32_t* __restrict__ fslots
#define PARSE_ARG_NAMES tasks, n_tasks, refs, planes, nmask, base_off, glen, g_has_n, rr_pool, mask_pool, \
    stab_pool, sent_pool, P, stats, arena, arena_cursor, arena_cap, fslots

// FAST: the default LZ-ANI parameters and a set without N as compile-time constants (shift counts, loop bounds
// and the mask paths fold away); the host launches it when both hold.
template <int S, bool DEV, bool FAST = false, bool REG = false, bool FUSED = false>
__device__ __forceinline__ void lz_parse_body(PARSE_ARGS) {
    static_assert(!REG || S == 1, "regions are emitted by the one-wave parse (query order)");
    if (FAST) { P.mal = 11; P.msl = 7; P.mrd = 40; P.mqd = 40; P.reg = 35; P.aw = 15; P.am = 7; P.ar = 3; P.ablate = 0; P.weak_ratio = 3; P.margin = 6; P.seed_choice = 3; }
    const int ABL = DEV ? P.ablate : 0;           // developer timing knobs: compiled out of the production kernels
    __shared__ seg_rec s_log[S > 1 ? S * SEG_LOG_CAP : 1];
    __shared__ int s_cnt[4], s_sync_v[4], s_sync_idx[4];
    __shared__ uint32_t s_end[4][3];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform: task data lives in SGPRs
    // XCD-aware dealing: consecutive task groups of one reference stay on one XCD (block b runs on XCD b % 8)
    const int64_t per_xcd = gridDim.x / 8;           // grid is a multiple of 8 workgroups
    const int64_t vblk = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int64_t t = (S == 1) ? vblk * 4 + w : vblk;
    if (t >= n_tasks) return;                        // S > 1: the whole workgroup leaves together
    const task_dev tk = tasks[t];
    ref_desc rd = refs[tk.r_slot];
    if (FAST) { rd.tag_bits = 8; rd.has_n = 0; }
    pair_ctx c;
    const int64_t qb = base_off[tk.q];
    c.qpl = planes + (qb >> 4); c.qmk = nmask + (qb >> 5); c.qlen = (int)glen[tk.q]; c.q_has_n = FAST ? 0 : g_has_n[tk.q];
    c.rpl = rr_pool + rd.rr_w; c.rmk = mask_pool + rd.mask_w; c.n_rr = rd.n_rr; c.L = rd.L; c.r_has_n = rd.has_n;
    const uint32_t* stab = stab_pool + rd.stab; const uint32_t* sent = sent_pool + rd.sent;

    const long long t_start = (ABL & (32 | 1024)) ? (long long)wall_clock64() : 0;
    const int lim = c.qlen - P.mal;
    // segments: only worth it for queries of a few thousand bases
    const int seg_len = (S > 1 && lim >= S * 2048) ? ((((lim + S - 1) / S) + 63) & ~63) : (lim > 0 ? lim : 1);
    const int seg_start = (S > 1) ? min(w * seg_len, lim > 0 ? lim : 0) : 0;
    int phase_end = (S > 1 && w < S - 1) ? min((w + 1) * seg_len, lim) : lim;
    int i = seg_start, lit = 0, pred = 0; bool alive = false;
    int n_events = 0, n_iter = 0, n_ab = 0, n_sb = 0;
    int pw = 64;                 // lanes (query positions) probed per trip
    bool synced = false; int sync_v = -1, sync_idx = 0, log_n = 0, look_v = -1, look_cur = 0;
    const bool prof = (ABL & 128) != 0; const int psel = (ABL >> 8) & 7;
    long long pc[6] = {0, 0, 0, 0, 0, 0}; long long tp = prof ? (long long)clock64() : 0;
#define PROF_MARK(k) do { if (DEV && prof) { long long tn_ = (long long)clock64(); pc[k] += tn_ - tp; tp = tn_; } } while (0)
    bool in_region = false; int r_qstart = 0, r_rstart = 0, r_qend = 0, r_rend = -1, r_match = 0, vend = 0;
    int kept_end = seg_start;
    uint32_t M = 0, A = 0, NR = 0;
    unsigned long long chunk_base = 0;

    auto close_region = [&]() {
        if (in_region) {
            int span = r_qend - r_qstart + 1;
            if (span >= P.reg) {
                M += (uint32_t)r_match; A += (uint32_t)span; NR += 1; kept_end = r_qend + 1;
                if (REG) {
                    const uint32_t kk = NR - 1;
                    if ((kk & (RCHUNK - 1)) == 0) {
                        // a new chunk: one atomic on the cursor by the wave's first lane, broadcast through SGPRs
                        unsigned long long b = 0;
                        if (lane == 0) b = atomicAdd(arena_cursor, (unsigned long long)RCHUNK);
                        chunk_base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                                     (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
                    }
                    const unsigned long long at = chunk_base + (kk & (RCHUNK - 1));
                    if (lane == 0 && at < arena_cap) {
                        region_rec rg; rg.task = tk.out_idx; rg.qstart = r_qstart; rg.qend = r_qend;
                        rg.rstart = r_rstart; rg.rend = r_rend; rg.n_match = r_match; rg.k = kk; rg.t = (uint32_t)t;
                        arena[at] = rg;
                    }
                }
            }
            in_region = false;
        }
    };

    for (int phase = 0; phase < (S > 1 ? 2 : 1); ++phase) {
    while (i < phase_end) {
        // ---- speculative probe of positions i .. i+63
        const int qi = i + lane;
        int best_len = 0, best_pos = 0; bool hit_close = false;
        if (qi < lim && lane < pw) {
            const bool alive_l = alive && (lit + lane <= P.mqd);
            const int pred_l = pred + lane;
            const planes32 xq = loadp(c.qpl, qi);
            bool q_ok_a = true, q_ok_s = (qi + P.msl <= c.qlen);
            uint32_t qbad = (c.qlen - qi < 32) ? ~slots(0, c.qlen - qi) : 0u;     // slots past the query end
            if (c.q_has_n) {
                const uint32_t m = loadm32(c.qmk, qi);
                q_ok_a = (m & ((1u << P.mal) - 1u)) == 0;                          // (mal <= 31)
                q_ok_s = q_ok_s && (m & ((1u << P.msl) - 1u)) == 0;
                qbad |= m;
            }
            // R2 (anchor = longest exact match >= mal over all occurrences of the mal-mer) and R3 (seed >= msl
            // near the prediction) read ONE bucket: the entries of the query's msl-mer, four per trip.  An
            // entry is an anchor candidate when its tag (the bases behind the msl-mer) equals the query's, a
            // seed candidate when it lies in the prediction's window; a candidate is verified once.
            const bool do_a = q_ok_a && !(ABL & 16);
            const bool do_s = alive_l && q_ok_s && !(ABL & 1);
            uint32_t s_u = 0, s_e = 0;
            const uint32_t posmask = (rd.pos_bits >= 32) ? 0xffffffffu : ((1u << rd.pos_bits) - 1u);
            const uint32_t qtag = tag_of(xq.lo, xq.hi, P.msl, rd.tag_bits);
            const bool probing = (do_a || do_s) && q_ok_s;
            const uint32_t b = bucket_of(xq.lo, xq.hi, P.msl);
            uint4 f0 = make_uint4(~0u, ~0u, ~0u, ~0u), f1 = f0;
            if (FUSED) {
                // EXPERIMENT (VG_LZ_INDEX=fused): the bucket's first entries sit in a fixed 32-byte slot -- no bounds word in
                // front of them, ONE line per probing lane; a word with bit 31 set ends the list (all ones) or, as the slot's
                // last word, points at the bucket's remaining entries in the ordinary entry array
                if (probing) { const uint32_t* sl = fslots + ((size_t)tk.r_slot * FUSED_SLOTS + b) * 8; __builtin_memcpy(&f0, sl, 16); __builtin_memcpy(&f1, sl + 4, 16); }
            } else if (probing) {
                // bucket bounds = two neighbouring table words: one 8-byte load
                uint2 bb; __builtin_memcpy(&bb, stab + (b ? b - 1 : 0u), 8);
                s_u = b ? bb.x : 0u; s_e = b ? bb.y : bb.x;
            }
            const int pred0 = pred - lit;                        // reference end of the previous match
            const bool pred_rc = pred0 > c.L;                    // strand of the prediction
            // R3 window: not before the end of the previous match, less than mrd ahead of the advancing
            // prediction, and on the prediction's strand
            const int win_lo = pred0;
            const int win_hi = pred_rc ? pred_l + P.mrd - 1 : min(pred_l + P.mrd - 1, c.L - 1);
            int sbest_len = 0, sbest_pos = 0, sbest_ad = 0, ncap_a = 0, ncap_s = 0;
            const uint32_t win_span = (uint32_t)(win_hi - win_lo);                  // window test: one unsigned compare
            const bool win_any = win_hi >= win_lo;
            // four entries of the bucket (vm: which of them exist): anchor candidates by tag, seed candidates by window, each verified once
            auto scan4 = [&](const uint4 v, const unsigned vm) {
                const uint32_t e4[4] = { v.x, v.y, v.z, v.w };
                unsigned am = 0, sm = 0;
                // (all four entries are tested without a branch; those past the bucket end are masked out afterwards)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t ps = e4[j] & posmask;
                    am |= (uint32_t)(rd.tag_bits == 0 || (e4[j] >> rd.pos_bits) == qtag) << j;
                    sm |= (uint32_t)(ps - (uint32_t)win_lo <= win_span) << j;
                }
                am &= do_a ? vm : 0u; sm &= (do_s && win_any) ? vm : 0u;
                unsigned cm = am | sm;
                while (cm) {
                    const int j = __builtin_ctz(cm); cm &= cm - 1;
                    const uint32_t ej = j == 0 ? e4[0] : j == 1 ? e4[1] : j == 2 ? e4[2] : e4[3];
                    const int rp = (int)(ej & posmask);
                    const int l0 = match_len32_q(c, xq, qbad, rp);
                    if ((am >> j) & 1u) {
                        int l = l0;
                        if (l >= P.mal) {
                            if (l >= 32) {
                                // rare: several long candidates need their exact lengths to be ranked
                                if (ncap_a++ > 0 || best_len >= 32) {
                                    l = match_len_lane(c, qi, rp, 1 << 30);
                                    if (best_len == 32) best_len = match_len_lane(c, qi, best_pos, 1 << 30);
                                }
                            }
                            if (l > best_len || (l == best_len && rp < best_pos)) { best_len = l; best_pos = rp; }      // longest; ties -> smallest position
                        }
                    }
                    if ((sm >> j) & 1u) {
                        int l = l0;
                        if (l >= P.msl) {
                            if (l >= 32) {
                                if (ncap_s++ > 0 || sbest_len >= 32) {
                                    l = match_len_lane(c, qi, rp, 1 << 30);
                                    if (sbest_len == 32) sbest_len = match_len_lane(c, qi, sbest_pos, 1 << 30);
                                }
                            }
                            // longest; ties -> closest to the prediction, then smallest position (seed_choice 1: closest first, then longest)
                            const int ad = abs(rp - pred_l);
                            const bool better = P.seed_choice == 1 ? (sbest_len == 0 || ad < sbest_ad || (ad == sbest_ad && (l > sbest_len || (l == sbest_len && rp < sbest_pos))))
                                                                   : (l > sbest_len || (l == sbest_len && (ad < sbest_ad || (ad == sbest_ad && rp < sbest_pos))));
                            if (better) { sbest_len = l; sbest_pos = rp; sbest_ad = ad; }
                        }
                    }
                }
            };
            if (FUSED) {
                if (DEV) { ++n_ab; }
                // (a slot's words with bit 31 set are not entries: padding or the pointer)
                const unsigned vm0 = (unsigned)(!(f0.x >> 31)) | ((unsigned)(!(f0.y >> 31)) << 1) | ((unsigned)(!(f0.z >> 31)) << 2) | ((unsigned)(!(f0.w >> 31)) << 3);
                const unsigned vm1 = (unsigned)(!(f1.x >> 31)) | ((unsigned)(!(f1.y >> 31)) << 1) | ((unsigned)(!(f1.z >> 31)) << 2) | ((unsigned)(!(f1.w >> 31)) << 3);
                if (vm0) scan4(f0, vm0);
                if (vm1) scan4(f1, vm1);
                if ((f1.w >> 31) && f1.w != ~0u) {
                    // the rest of a bucket of more than eight entries: count << 23 | first entry; count 255 = ask the bounds table
                    s_u = f1.w & 0x7fffffu; const uint32_t cnt = (f1.w >> 23) & 0xffu;
                    if (cnt == 255u) { uint2 bb; __builtin_memcpy(&bb, stab + (b ? b - 1 : 0u), 8); s_e = b ? bb.y : bb.x; }
                    else s_e = s_u + cnt;
                }
                const unsigned long long hit0 = __ballot(best_len > 0 || sbest_len > 0);
                if (hit0 && lane > __builtin_ctzll(hit0)) s_u = s_e;
            }
            while (s_u < s_e) {
                if (DEV) { ++n_ab; }
                // four consecutive entries = one 16-byte load (the pool carries four entries of slack; entries past
                // the bucket end are ignored below)
                uint4 v; __builtin_memcpy(&v, sent + s_u, 16);
                const uint32_t left = s_e - s_u;
                scan4(v, left >= 4u ? 0xfu : ((1u << left) - 1u));
                s_u += 4;
                // positions behind the first one that already has a match cannot become the event: stop their walks
                const unsigned long long hit = __ballot(best_len > 0 || sbest_len > 0);
                if (hit && lane > __builtin_ctzll(hit)) s_u = s_e;
            }
            // R2/R3 choice: without a prediction the anchor; with one the seed, unless an anchor is longer
            // than the seed by at least msl (a far, long match beats a short close one)
            if (best_len > 0 && sbest_len > 0 && best_pos != sbest_pos && best_len >= 32 && sbest_len + P.margin + 1 > 32) {
                if (best_len == 32) best_len = match_len_lane(c, qi, best_pos, 1 << 30);
                if (sbest_len == 32) sbest_len = match_len_lane(c, qi, sbest_pos, 1 << 30);
            }
            // (one symbol less when the seed is weak: shorter than a third of the literal run it would bridge.  This constant
            // rests on ONE event of the reference's example -- profiles/r04_lz_fit_leave_one_out.md --, so it is a parameter:
            // VG_LZ_WEAK_SEED=0 switches the rule off, =n sets the ratio, in the product and, through the variant, in the oracle)
            if (best_len > 0 && (sbest_len == 0 || best_len > sbest_len + P.margin - ((P.weak_ratio > 0 && lit + lane > P.weak_ratio * sbest_len) ? 1 : 0))) {
                const int d = best_pos - pred_l;
                hit_close = alive_l && ((best_pos > c.L) == pred_rc) && d >= -P.mrd && d <= P.mrd;
            } else if (sbest_len > 0) { best_len = sbest_len; best_pos = sbest_pos; hit_close = true; }
        }
        const unsigned long long hb = __ballot(best_len > 0);
        if (DEV) ++n_iter;
        PROF_MARK(0);
        if (!hb) {
            // pw literals (or the tail); widen the next probe: a stretch without matches is scanned 64 at a time
            const int n = min(pw, lim - i);
            i += n; lit += n; if (alive) { pred += n; if (lit > P.mqd) alive = false; }
            pw = pw < P.pw_miss ? P.pw_miss : 64;
            continue;
        }
        const int f = __builtin_ctzll(hb);
        pw = P.pw_after;            // the next match usually starts within a few positions of the end of this one
        // Pairs that need many events are the tail of the launch: a wave raises its own issue priority
        // as its event count grows, so the heavy pairs overtake the light ones sharing their SIMD.
        if (!(ABL & 64)) {
            ++n_events;
            if (n_events == 24) __builtin_amdgcn_s_setprio(1);
            else if (n_events == 64) __builtin_amdgcn_s_setprio(2);
            else if (n_events == 160) __builtin_amdgcn_s_setprio(3);
        }
        const int ev_pos = (int)lane32((uint32_t)best_pos, f);
        const bool ev_close = lane32((uint32_t)hit_close, f) != 0;
        // literals in front of the event
        i += f; lit += f; if (alive) { pred += f; if (lit > P.mqd) alive = false; }
        const int gap_end_ref = pred - 1;
        const int ev_i = i;
        const int ev_len = (int)lane32((uint32_t)best_len, f);        // exact length, capped at 32 unless it had to be ranked
        // everything below stays on the strand of the match
        const int rlo = strand_lo(c, ev_pos), rhi = strand_hi(c, ev_pos);
        // first-round loads of the left extension and of the right extension are independent: issue them
        // together, then run the window logic
        // (closing the open region may move kept_end: use the value it will have)
        const int kept_after = (in_region && r_qend - r_qstart + 1 >= P.reg) ? r_qend + 1 : kept_end;
        const int bwd_bound = ev_close ? 0 : i - kept_after;
        const uint32_t mm_b = (!ev_close && !(ABL & 2)) ? extend_mask0(c, i, ev_pos, -1, bwd_bound, lane, rlo, rhi) : ~0u;
        const uint32_t mm_f = extend_mask0(c, i, ev_pos, +1, 1 << 30, lane, rlo, rhi);
        PROF_MARK(1);
        if (!ev_close) {
            // R5: new region, extended to the left (exact, then approximate), not into the last kept region
            close_region();
            int bm = 0;
            const int b = (ABL & 2) ? 0 : extend(c, P, i, ev_pos, -1, bwd_bound, lane, &bm, mm_b, rlo, rhi);
            r_qstart = i - b; r_rstart = ev_pos - b; r_match = bm; r_rend = -1;
            in_region = true;
        }
        PROF_MARK(2);
        int fe, fm = 0;
        {   // the match itself and R4, one pass
            fe = (ABL & 4) ? P.mal : extend(c, P, i, ev_pos, +1, 1 << 30, lane, &fm, mm_f, rlo, rhi);
            r_match += fm;
        }
        if (ev_close) {
            // R7: the literal run in front of a chained match, laid against [pred0, end of the exact match)
            int pm = 0, sm = 0;
            if (lit > 0 && !(ABL & 8)) {
                // exact length of the match = position of the first mismatch of the forward pass
                int xl = ev_len;
                if (xl >= 32) {
                    const unsigned long long mb = __ballot(mm_f != 0);
                    xl = mb ? 32 * __builtin_ctzll(mb) + __builtin_ctz(lane32(mm_f, __builtin_ctzll(mb))) : 2048;
                    if (xl >= 2048) xl = match_len_lane(c, i, ev_pos, 1 << 30);
                }
                r_match += gap_score(c, i, lit, pred - lit, ev_pos + xl, lane, rlo, rhi, &pm, &sm);
            }
            // reference end of the region: a symbol that matches nothing moves it by one, one matched on the
            // old diagonal pulls it up to its own position, one matched on the new diagonal leaves it
            const int tru = gap_end_ref + 1, vir = vend + lit - pm;
            vend = max(tru, vir) - sm + (fe - fm);
            r_rend = max(r_rend, max(ev_pos + fe - 1, vend - 1));
            r_rstart = min(r_rstart, ev_pos);
        }
        i += fe; pred = ev_pos + fe; lit = 0; alive = true;
        if (!ev_close) { r_rend = pred - 1; vend = pred; }
        PROF_MARK(3);
        r_qend = i - 1;
        if (S > 1) {
            // After an event at (ev_i, ev_pos) the scan state is (i, pred, lit = 0, alive): a function of the
            // event alone.  The open regions of two parses that meet in the same event may have started
            // at different places, but once both already span >= reg they are kept by both, end at the same
            // place and set the same kept_end; their sums then differ by a constant, which the
            // "virtual" sums (open region counted up to here) carry across the hand-over.
            const bool span_ok = i - r_qstart >= P.reg;
            if (phase == 1) {
                const int v = min(S - 1, ev_i / seg_len);
                if (v != look_v) { look_v = v; look_cur = 0; }
                const seg_rec* lg = s_log + v * SEG_LOG_CAP; const int nv = s_cnt[v];
                while (look_cur < nv && lg[look_cur].i_ev < ev_i) ++look_cur;
                if (span_ok && look_cur < nv && lg[look_cur].i_ev == ev_i && lg[look_cur].ev_pos == ev_pos && (lg[look_cur].VN >> 31)) {
                    synced = true; sync_v = v; sync_idx = look_cur;
                    M += (uint32_t)r_match; A -= (uint32_t)r_qstart;      // virtual sums; the region itself is the owner's now
                    break;
                }
            } else if (log_n < SEG_LOG_CAP) {
                if (lane == 0) {
                    seg_rec rc; rc.i_ev = ev_i; rc.ev_pos = ev_pos;
                    rc.VM = M + (uint32_t)r_match; rc.VA = A - (uint32_t)r_qstart; rc.VN = NR | (span_ok ? 0x80000000u : 0u);
                    s_log[w * SEG_LOG_CAP + log_n] = rc;
                }
                ++log_n;
            }
        }
    }
    if (S > 1 && phase == 0) {
        if (lane == 0) s_cnt[w] = log_n;
        __syncthreads();                             // every segment's log is complete
        phase_end = lim;
    }
    if (synced) break;
    }
    if (!synced) close_region();
    if (S > 1) {
        if (lane == 0) { s_end[w][0] = M; s_end[w][1] = A; s_end[w][2] = NR; s_sync_v[w] = sync_v; s_sync_idx[w] = sync_idx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            // follow the hand-overs from wave 0: its sums up to the hand-over, then the next wave's from there
            uint32_t tm = 0, ta = 0, tn = 0, bm = 0, ba = 0, bn = 0; int cur = 0;
            for (;;) {
                tm += s_end[cur][0] - bm; ta += s_end[cur][1] - ba; tn += s_end[cur][2] - bn;
                const int v = s_sync_v[cur];
                if (v < 0) break;
                const seg_rec rc = s_log[v * SEG_LOG_CAP + s_sync_idx[cur]];
                bm = rc.VM; ba = rc.VA; bn = rc.VN & 0x7fffffffu; cur = v;
            }
            vg_pair_stat st; st.n_match = tm; st.aln_len = ta; st.n_regions = tn;
            stats[tk.out_idx] = st;
        }
        return;
    }
    if (prof) { M = (uint32_t)(pc[psel] >> 4); A = (uint32_t)n_events; }
    if (ABL & 1024) {
        // wave-level trip counts of the bucket loops = max over lanes
        int mab = n_ab, msb = n_sb;
        for (int o = 32; o; o >>= 1) { mab = max(mab, __shfl_xor(mab, o)); msb = max(msb, __shfl_xor(msb, o)); }
        M = (uint32_t)n_iter; A = (uint32_t)n_events; if (ABL & 2048) { M = (uint32_t)mab; A = (uint32_t)msb; }
    }
    if (ABL & (32 | 1024)) NR = (uint32_t)((long long)wall_clock64() - t_start);        // developer timing: 100 MHz ticks
    if (REG && (NR & (RCHUNK - 1)) != 0) {
        // the unused tail of the last chunk: marked, one slot per lane
        const uint32_t used = NR & (RCHUNK - 1);
        const unsigned long long at = chunk_base + used + (uint32_t)lane;
        if (used + (uint32_t)lane < (uint32_t)RCHUNK && at < arena_cap) arena[at].task = ~0u;
    }
    if (lane == 0) { vg_pair_stat st; st.n_match = M; st.aln_len = A; st.n_regions = NR; stats[tk.out_idx] = st; }
}

// The parse is bound by dependent memory round trips, so resident waves are throughput: the
// register budget is capped for the occupancy named in each kernel (waves per SIMD).
#define PARSE_KERNEL(NAME, S, DEV, WAVES, FAST, REG, ...) \
    __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) NAME(PARSE_ARGS) { \
        lz_parse_body<S, DEV, FAST, REG, ##__VA_ARGS__>(PARSE_ARG_NAMES); }
PARSE_KERNEL(k_lz_parse, 1, false, 8, false, false)
PARSE_KERNEL(k_lz_parse_fast, 1, false, 8, true, false)
PARSE_KERNEL(k_lz_parse_seg, 4, false, 8, false, false)
PARSE_KERNEL(k_lz_parse_seg_fast, 4, false, 8, true, false)
PARSE_KERNEL(k_lz_parse_regions, 1, false, 8, false, true)            // --out-aln: rows AND regions from one parse
PARSE_KERNEL(k_lz_parse_fast_regions, 1, false, 8, true, true)
PARSE_KERNEL(k_lz_parse_fast_fused, 1, false, 8, true, false, true)     // EXPERIMENT: probes read fused bucket slots (VG_LZ_INDEX=fused)
// EXPERIMENT: the fused slots of a batch's references made FROM the ordinary index (a conversion pass, so that the probe
// side can be measured before a build kernel writes this layout itself): one workgroup per reference, a thread per bucket
__global__ void __launch_bounds__(256)
k_index_fuse(const ref_desc* __restrict__ refs, int first_ref, int n_refs, const uint32_t* __restrict__ stab_pool, const uint32_t* __restrict__ sent_pool,
             uint32_t* __restrict__ fslots) {
    for (int r = blockIdx.x; r < n_refs; r += gridDim.x) {
        const ref_desc rd = refs[first_ref + r];
        const uint32_t* stab = stab_pool + rd.stab; const uint32_t* sent = sent_pool + rd.sent;
        uint32_t* out = fslots + (size_t)(first_ref + r) * FUSED_SLOTS * 8;
        for (int b = threadIdx.x; b < FUSED_SLOTS; b += blockDim.x) {
            const uint32_t lo = b ? stab[b - 1] : 0u, hi = stab[b], n = hi - lo;
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (uint32_t)j < n ? sent[lo + j] : ~0u;
            if (n > 8) w[7] = 0x80000000u | (((n - 7) < 255u ? (n - 7) : 255u) << 23) | (lo + 7);
            uint4 a = make_uint4(w[0], w[1], w[2], w[3]), c2 = make_uint4(w[4], w[5], w[6], w[7]);
            __builtin_memcpy(out + (size_t)b * 8, &a, 16); __builtin_memcpy(out + (size_t)b * 8 + 4, &c2, 16);
        }
    }
}
#ifdef VG_DEV_KERNELS      // developer build only (VG_DEV=1 python -m vclust_amd.build --force): timing knobs and counters
PARSE_KERNEL(k_lz_parse_dev, 1, true, 3, false, false)
#endif

// --out-aln, behind the parse: region counts of a batch's tasks in sorted-list order (the scan of them = every task's first
// slot), and the move of every arena record to slot first[t] + k
__global__ void __launch_bounds__(256)
k_region_counts(const task_dev* __restrict__ tasks, int64_t n_tasks, const vg_pair_stat* __restrict__ stats, unsigned long long* __restrict__ cnt) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n_tasks; t += (int64_t)gridDim.x * blockDim.x)
        cnt[t] = t < n_tasks ? (unsigned long long)stats[tasks[t].out_idx].n_regions : 0ULL;
}
__global__ void __launch_bounds__(256)
k_regions_place(const region_rec* __restrict__ arena, unsigned long long n_slots, const unsigned long long* __restrict__ first,
                unsigned long long n_out, vg_region* __restrict__ out, unsigned int* __restrict__ bad) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (unsigned long long)gridDim.x * blockDim.x) {
        const region_rec r = arena[i];
        if (r.task == ~0u) continue;
        const unsigned long long at = first[r.t] + r.k;
        if (at >= first[r.t + 1] || at >= n_out) { atomicAdd(bad, 1u); continue; }      // (rows and records of one parse cannot disagree: reported, never written)
        vg_region o; o.task = r.task; o.qstart = r.qstart; o.qend = r.qend; o.rstart = r.rstart; o.rend = r.rend; o.n_match = r.n_match;
        out[at] = o;
    }
}

inline int grid_for(int64_t n, int block = 256, int max_blocks = 256 * 16) {
    int64_t b = (n + block - 1) / block; if (b < 1) b = 1;
    return (int)std::min<int64_t>(b, max_blocks);
}

}  // namespace

// The constants of the LZ restatement that <= 3 events of the reference's example decide (vg_lz_fit): process-wide, set
// through vg_set_lz_fit; their initial values may come from developer switches (honoured only beside VG_DEV_SWITCHES=1):
// VG_LZ_WEAK_SEED, VG_LZ_ANCHOR_MARGIN, VG_LZ_SEED_CHOICE -- what the CLI-level tests use, read by the oracle too.
static std::mutex g_fit_mu;
static bool g_fit_set = false;
static vg_lz_fit g_fit{ 3, -1, 3 };
static vg_lz_fit lz_fit_now() {
    std::lock_guard<std::mutex> lk(g_fit_mu);
    if (!g_fit_set) {
        g_fit_set = true;
        if (const char* e = vg_dev_getenv("VG_LZ_WEAK_SEED")) if (*e) g_fit.weak_seed_ratio = atoi(e);
        if (const char* e = vg_dev_getenv("VG_LZ_ANCHOR_MARGIN")) if (*e) g_fit.anchor_margin = atoi(e);
        if (const char* e = vg_dev_getenv("VG_LZ_SEED_CHOICE")) if (*e) g_fit.seed_choice = atoi(e);
        if (g_fit.weak_seed_ratio != 3 || g_fit.anchor_margin >= 0 || g_fit.seed_choice != 3)
            fprintf(stderr, "libvclust_gpu: LZ fit constants changed by developer switches (weak seed ratio %d, anchor margin %d, seed choice %d): results differ from the reference's\n",
                    g_fit.weak_seed_ratio, g_fit.anchor_margin, g_fit.seed_choice);
    }
    return g_fit;
}
extern "C" void vg_set_lz_fit(const vg_lz_fit* f) {
    std::lock_guard<std::mutex> lk(g_fit_mu);
    g_fit_set = true;
    g_fit = f ? *f : vg_lz_fit{ 3, -1, 3 };
}
static bool g_index_budget_set = false;          // VG_INDEX_BUDGET_GB / vg_set_index_budget: the caller's figure is taken as it is
static int64_t g_index_budget_bytes = [] { const char* e = getenv("VG_INDEX_BUDGET_GB"); const double v = e ? atof(e) : 0.0; g_index_budget_set = v >= 0.0625; return v >= 0.0625 ? (int64_t)(v * 1073741824.0) : (24LL << 30); }();
// ---- task grouping on the device: the caller's (q, r) list is counted per reference, stably sorted on r
// (rocPRIM radix sort of the 17..32-bit reference ids with the list position as value) and turned into the
// device task records -- the host only sees the per-reference counts it plans the batches from.
namespace {
__global__ void __launch_bounds__(256)
k_task_count(const vg_task* __restrict__ tasks, int64_t n_tasks, int n_genomes, const int64_t* __restrict__ len,
             uint32_t* __restrict__ ref_cnt, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
             unsigned long long* __restrict__ sums /* [0] sum len[q], [1] sum len[r], [2] max len[q], [3] bad ids */) {
    unsigned long long sq = 0, sr = 0, mq = 0, bad = 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tasks; t += (int64_t)gridDim.x * blockDim.x) {
        const vg_task tk = tasks[t];
        keys[t] = tk.r; vals[t] = (uint32_t)t;
        if (tk.q >= (uint32_t)n_genomes || tk.r >= (uint32_t)n_genomes) { bad = 1; continue; }
        atomicAdd(&ref_cnt[tk.r], 1u);
        const unsigned long long ql = (unsigned long long)len[tk.q];
        sq += ql; sr += (unsigned long long)len[tk.r]; mq = ql > mq ? ql : mq;
    }
    for (int o = 32; o > 0; o >>= 1) {
        sq += __shfl_xor(sq, o); sr += __shfl_xor(sr, o); bad |= __shfl_xor(bad, o);
        const unsigned long long m2 = __shfl_xor(mq, o); mq = m2 > mq ? m2 : mq;
    }
    if ((threadIdx.x & 63) == 0) {
        if (sq) atomicAdd(&sums[0], sq);
        if (sr) atomicAdd(&sums[1], sr);
        if (mq) atomicMax(&sums[2], mq);
        if (bad) atomicOr(&sums[3], 1ULL);
    }
}
__global__ void __launch_bounds__(256)
k_task_records(const vg_task* __restrict__ tasks, const uint32_t* __restrict__ sorted_idx, int64_t n_tasks,
               const uint32_t* __restrict__ ord /* genome -> ordinal among the references that have tasks */, task_dev* __restrict__ td) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_tasks; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t t = sorted_idx[i];
        const vg_task tk = tasks[t];
        td[i] = { tk.q, ord[tk.r], t, 0u };
    }
}
}  // namespace

// ---- reference plan of a vg_lz_align call: which references are indexed in which batch, their descriptors, the lists
// of the build kernels and one set of index pools sized for the largest batch.  It depends on the REFERENCES of the task
// list only (not on the tasks), so vg_lz_prepare can make it -- and build the first batch's indexes -- from the candidate
// pairs alone, while the caller still assembles the canonical task list on the host.
namespace {
struct lz_batch {
    int64_t pos = 0, end = 0;            // sorted task range
    int first_ref = 0, n_refs = 0;       // reference ordinals [first_ref, first_ref + n_refs)
    std::vector<int64_t> chunk_off{ 0 };
    std::vector<int> reg_list, mid_list, small_list, large_list; std::vector<int64_t> large_chunks{ 0 };
    int reg_class_end[6] = {0, 0, 0, 0, 0, 0};      // reg_list is ordered by register class (24, 20, 16, 12, 8, 4 trips): end of each class
    int64_t rr_words = 0, mask_words = 0, stab_tot = 0, sent_n = 0, scratch_words = 0, stride = 0;
    int nblk_build = 0;
    double bytes_alg = 0; int64_t q_max = 0, q_sum = 0;          // SURVEY 8(d) bytes of the batch; longest / total query
};
struct lz_slot { dbuf<uint32_t> rr_pool, mask_pool, stab_pool, sent_pool, scratch; dbuf<int> d_reg, d_mid, d_small, d_large; dbuf<int64_t> d_lchunk; };
struct lz_plan {
    std::vector<uint32_t> ref_ids;            // references that have tasks, ascending
    std::vector<lz_batch> batches;
    std::vector<ref_desc> all_refs;           // indexed by the reference ordinal the task records carry
    dbuf<ref_desc> d_refs;
    lz_slot slot;
    int64_t budget = 0; int mal = 0, msl = 0; const vg_genomes* g = nullptr; int n_genomes = 0;
    bool batch0_built = false;
    hipEvent_t built_ev = nullptr;            // batch 0 was queued on another queue than the library's: vg_lz_align waits for it
    ~lz_plan() { if (built_ev) { (void)hipEventSynchronize(built_ev); (void)hipEventDestroy(built_ev); } }
};
int64_t lz_batch_budget(const vg_genomes* g, const vg_lz_params* p, const std::vector<uint32_t>& ref_ids);
void lz_plan_references(const vg_genomes* g, const vg_lz_params* p, lz_plan& P);
void lz_plan_alloc(lz_plan& P);
void lz_build_batch(const vg_genomes* g, const vg_lz_params* p, lz_plan& P, size_t bi, hipStream_t sb);
std::mutex g_prep_mu;
std::unique_ptr<lz_plan> g_prepared;          // left by vg_lz_prepare for the next vg_lz_align
// The HOST side of the last plan (batches, descriptors, build lists: 1-2 ms of bookkeeping per 100 000 references), kept
// without its device pools: a long-lived process that aligns against the same references again -- the steps of a
// resident service, bench.py -- takes it back instead of cutting the batches anew.  It depends on (set, parameters, budget,
// reference ids) only; it is dropped with the set (vg_genomes_free), by vg_release_device_memory, and never kept by the
// cold one-shot calls.  Results do not depend on it.
std::unique_ptr<lz_plan> g_plan_cache;
// (under g_prep_mu) the cached plan when it is the plan of exactly these references, with fresh pools; else nullptr
std::unique_ptr<lz_plan> lz_take_cached_plan(const vg_genomes* g, const vg_lz_params* p, const std::vector<uint32_t>& ref_ids) {
    if (!g_plan_cache) return nullptr;
    lz_plan& Q = *g_plan_cache;
    if (Q.g != g || Q.mal != p->mal || Q.msl != p->msl || Q.n_genomes != g->n || Q.ref_ids != ref_ids || Q.budget != lz_batch_budget(g, p, ref_ids)) { g_plan_cache.reset(); return nullptr; }
    return std::move(g_plan_cache);
}
void lz_keep_plan_host_side(std::unique_ptr<lz_plan>& plan) {
    if (!plan || vg_one_shot()) return;
    plan->slot = lz_slot(); plan->d_refs.release();            // (the pools go back to the allocator: nothing of the device is pinned)
    plan->batch0_built = false;
    if (plan->built_ev) { (void)hipEventSynchronize(plan->built_ev); (void)hipEventDestroy(plan->built_ev); plan->built_ev = nullptr; }
    std::lock_guard<std::mutex> lk(g_prep_mu);
    g_plan_cache = std::move(plan);
}
}
void vg_lz_drop_prepared(const vg_genomes* g) {
    std::lock_guard<std::mutex> lk(g_prep_mu);
    // (a build deferred to the next SpGEMM -- developer experiment -- holds a raw pointer to the plan: it goes with it)
    if (g_prepared && (!g || g_prepared->g == g)) { vg_set_spgemm_hook(nullptr); (void)hipStreamSynchronize(vg_stream()); g_prepared.reset(); }
    if (g_plan_cache && (!g || g_plan_cache->g == g)) g_plan_cache.reset();
}

// The genome set as bit planes (what the parse reads its queries from): made once per resident set, on the library's
// stream, from the 2-bit codes (a streaming pass: 0.5 ms per Gbp).
namespace {
__global__ void __launch_bounds__(256)
k_genome_planes(const uint32_t* __restrict__ packed, int64_t n_pairs, uint32_t* __restrict__ planes) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_pairs; w += (int64_t)gridDim.x * blockDim.x) {
        uint2 v; __builtin_memcpy(&v, packed + 2 * w, 8);
        const planes32 pl = planes_of((uint64_t)v.x | ((uint64_t)v.y << 32));
        const uint2 o = make_uint2(pl.lo, pl.hi); __builtin_memcpy(planes + 2 * w, &o, 8);
    }
}
}  // namespace
static std::mutex g_planes_mu;
const uint32_t* vg_genome_planes(const vg_genomes* g, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_planes_mu);
    // The planes are always MADE on the library's stream; a caller on another queue (the developer experiments that build
    // indexes on a queue of their own) waits for that stream's work up to here, whether this call launched the kernel or an
    // earlier one did -- a pointer returned to a second queue never names planes that are still being written.
    hipStream_t lib = vg_stream();
    if (g->d_planes.n != g->d_packed.n || !g->d_planes.p) {
        const size_t words = g->d_packed.n & ~(size_t)1;
        g->d_planes.alloc(g->d_packed.n);
        hipLaunchKernelGGL(k_genome_planes, dim3(grid_for((int64_t)(words / 2))), dim3(256), 0, lib, (const uint32_t*)g->d_packed.p, (int64_t)(words / 2), g->d_planes.p);
    }
    if (s != lib) {
        hipEvent_t e = nullptr;
        VG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        VG_HIP(hipEventRecord(e, lib)); VG_HIP(hipStreamWaitEvent(s, e, 0));
        (void)hipEventDestroy(e);
    }
    return g->d_planes.p;
}

static int64_t g_segment_task_limit = 32768;
// VG_LZ_BUILD=lds: the scratch-based LDS build also for short references (tests compare the two)
static const bool g_no_reg_build = [] { const char* e = vg_dev_getenv("VG_LZ_BUILD"); return e && strcmp(e, "lds") == 0; }();

extern "C" int vg_lz_align(vg_genomes* g, const vg_task* tasks, int64_t n_tasks, const vg_lz_params* p,
                           vg_pair_stat* stats, vg_region** regions, int64_t* n_regions) {
    VG_API_BEGIN
    if (!g || (!tasks && n_tasks) || !p || (!stats && n_tasks)) throw vg_error(VG_EINVAL, "vg_lz_align: null argument");
    if (p->mal < 8 || p->mal > 31 || p->msl < 4 || p->msl > 12 || p->msl > p->mal) throw vg_error(VG_EINVAL, "mal must be 8..31, msl 4..12 and <= mal");
    if (p->aw < 1 || p->aw > 32 || p->ar < 1 || p->ar > 16 || p->am < 0 || p->mrd < 0 || p->mqd < 0 || p->reg < 0)
        throw vg_error(VG_EINVAL, "aw must be 1..32, ar 1..16");
    if (p->mqd > 2000) throw vg_error(VG_EINVAL, "mqd must be <= 2000 (one wave scores a literal run of at most 2 048 symbols)");
    vg_require_device();
    int rc = vg_genomes_to_device(g); if (rc) return rc;
    hipStream_t s = vg_stream();
    vg_host_mark("vg_lz_align: enter");
    const uint32_t* d_planes = vg_genome_planes(g, s);

    if (regions) { *regions = nullptr; if (n_regions) *n_regions = 0; }
    if (n_tasks == 0) return VG_OK;
    for (int i = 0; i < g->n; ++i) if (g->len[i] > (1 << 29)) throw vg_error(VG_EOVERFLOW, "genome longer than 2^29 bases");
    if (n_tasks >= (1LL << 32)) throw vg_error(VG_EOVERFLOW, "more than 2^32 - 1 ordered pairs in one call: split the task list");

    // group tasks by reference ON THE DEVICE (k_task_count, a stable radix sort on the reference id, k_task_records):
    // device task records (query, ordinal of the reference among the references that have tasks, position in
    // the caller's list); the host gets the per-reference counts back and plans the batches from them.
    static const bool two_pass_regions = [] { const char* e = vg_dev_getenv("VG_LZ_REGIONS"); return e && !strcmp(e, "two-pass"); }();      // developer switch: the checker of the one-parse --out-aln
    std::vector<task_dev> td;                                 // host copy, fetched for the two-pass --out-aln only
    std::vector<int64_t> ref_first((size_t)g->n + 1, 0);     // first sorted task of reference r
    std::vector<uint32_t> ref_ids;                            // references that have tasks, ascending
    int64_t q_max = 0; double q_sum = 0, bytes_alg_all = 0;
    dbuf<task_dev> d_tasks((size_t)n_tasks);
    {
        dbuf<vg_task> d_raw((size_t)n_tasks); d_raw.upload(tasks, (size_t)n_tasks, s);
        dbuf<uint32_t> d_cnt((size_t)g->n), d_keys((size_t)n_tasks), d_vals((size_t)n_tasks), d_keys2((size_t)n_tasks), d_vals2((size_t)n_tasks);
        dbuf<unsigned long long> d_sums(4);
        d_cnt.zero(s); d_sums.zero(s);
        hipLaunchKernelGGL(k_task_count, dim3(grid_for(n_tasks, 256, 512)), dim3(256), 0, s,   // (few workgroups: every wave ends in three atomics on the same words)
                           (const vg_task*)d_raw.p, n_tasks, g->n, g->d_len.p, d_cnt.p,
                           d_keys.p, d_vals.p, d_sums.p);
        std::vector<uint32_t> cnt((size_t)g->n); unsigned long long sums[4];
        d_cnt.download(cnt.data(), cnt.size(), s); d_sums.download(sums, 4, s);
        int id_bits = 1; while ((1LL << id_bits) < g->n) ++id_bits;
        size_t tmp_bytes = 0;
        VG_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys.p, d_keys2.p, d_vals.p, d_vals2.p, (size_t)n_tasks, 0u, (unsigned)id_bits, s));
        dbuf<char> tmp(tmp_bytes);
        VG_HIP(rocprim::radix_sort_pairs((void*)tmp.p, tmp_bytes, d_keys.p, d_keys2.p, d_vals.p, d_vals2.p, (size_t)n_tasks, 0u, (unsigned)id_bits, s));
        VG_HIP(hipStreamSynchronize(s));
        if (sums[3]) throw vg_error(VG_EINVAL, "task id out of range");
        std::vector<uint32_t> ord((size_t)g->n, 0);
        for (int i = 0; i < g->n; ++i) {
            if (cnt[(size_t)i]) { ord[(size_t)i] = (uint32_t)ref_ids.size(); ref_ids.push_back((uint32_t)i); }
            ref_first[(size_t)i + 1] = ref_first[(size_t)i] + cnt[(size_t)i];
        }
        q_max = (int64_t)sums[2]; q_sum = (double)sums[0];
        bytes_alg_all = ((double)sums[0] + (double)sums[1]) / 4.0 + 20.0 * (double)n_tasks;
        dbuf<uint32_t> d_ord((size_t)g->n); d_ord.upload(ord.data(), ord.size(), s);
        hipLaunchKernelGGL(k_task_records, dim3(grid_for(n_tasks)), dim3(256), 0, s, (const vg_task*)d_raw.p, (const uint32_t*)d_vals2.p, n_tasks,
                           (const uint32_t*)d_ord.p, d_tasks.p);
        if (regions && two_pass_regions) { td.resize((size_t)n_tasks); d_tasks.download(td.data(), td.size(), s); }
        VG_HIP(hipStreamSynchronize(s));                      // the scratch buffers go out of scope
    }
    vg_host_mark("lz: tasks grouped");
#ifdef VG_DEV_KERNELS
    const char* abl = vg_dev_getenv("VG_LZ_ABLATE");
#else
    const char* abl = nullptr;                                // (the product library has no timing knobs)
#endif
    // probe widths (speculation only: results do not depend on them): positions probed right after an event, and after a
    // first miss, before the scan goes to 64 per trip
    static const int pw_after = [] { const char* e = vg_dev_getenv("VG_LZ_PW"); const int v = e ? atoi(e) : PW_AFTER_EVENT; return std::max(1, std::min(v, 64)); }();
    static const int pw_miss = [] { const char* e = vg_dev_getenv("VG_LZ_PW2"); const int v = e ? atoi(e) : 64; return std::max(1, std::min(v, 64)); }();
    // R3's weak-seed ratio (a single-event fit): 3 unless VG_LZ_WEAK_SEED says otherwise (0 = off)
    const vg_lz_fit fit = lz_fit_now();
    const int weak_ratio = std::max(0, std::min(fit.weak_seed_ratio, 1000));
    const int margin = fit.anchor_margin >= 0 ? std::min(fit.anchor_margin, 1000) : p->msl - 1;
    const int seed_choice = fit.seed_choice == 1 ? 1 : 3;
    const lz_dev_params P{ p->mal, p->msl, p->mrd, p->mqd, p->reg, p->aw, p->am, p->ar, abl ? atoi(abl) : 0, pw_after, pw_miss, weak_ratio, margin, seed_choice };
    // default parameters on a set without N: the kernel with those as compile-time constants (VG_LZ_KERNEL=general: never)
    static const bool no_fast = [] { const char* e = vg_dev_getenv("VG_LZ_KERNEL"); return e && !strcmp(e, "general"); }();
    bool fast_params = !no_fast && !abl && weak_ratio == 3 && margin == 6 && seed_choice == 3 && p->mal == 11 && p->msl == 7 && p->mrd == 40 && p->mqd == 40 && p->reg == 35 && p->aw == 15 && p->am == 7 && p->ar == 3;
    for (int i = 0; fast_params && i < g->n; ++i) if (g->has_n[(size_t)i] || g->len[(size_t)i] >= (1 << 22)) fast_params = false;   // (tag: 8 bits beside <= 24 position bits)

    dbuf<vg_pair_stat> d_stats((size_t)n_tasks);
    dbuf<uint32_t> fused_pool;                        // EXPERIMENT VG_LZ_INDEX=fused: 32-byte bucket slots of all references of the call
    const bool want_regions = regions != nullptr;
    std::vector<vg_region> h_regions;                 // all kept regions, batch after batch
    std::vector<vg_pair_stat> h_stats;                // host copy of the rows (sizes the region buffer)
    std::vector<vg_pair_stat> first_pass;             // --out-aln: the rows of the counting pass, batch after batch

    // ---- plan: references are taken in id order, batch after batch, each batch's indexes under the budget.
    // (Building batch b + 1 on a second stream while batch b is parsed was measured: the 160 KiB-LDS build
    // workgroups and the parse waves only split the CUs between them, 321 vs 319 ms at 100 k genomes -- so
    // the batches run back to back on the library stream, as few and as large as the budget allows.)
    // A plan left by vg_lz_prepare for exactly these references (same set, same parameters, same budget) is taken over
    // with its first batch already built (or being built: same stream).
    std::unique_ptr<lz_plan> plan;
    {
        std::lock_guard<std::mutex> lk(g_prep_mu);
        if (g_prepared) {
            vg_set_spgemm_hook(nullptr);                          // (a deferred build that has not run yet never will: the plan changes hands or goes)
            lz_plan& Q = *g_prepared;
            if (Q.g == g && Q.mal == p->mal && Q.msl == p->msl && Q.ref_ids == ref_ids && Q.budget == lz_batch_budget(g, p, ref_ids)) plan = std::move(g_prepared);
            else g_prepared.reset();                             // (its pools go back to the allocator: one stream, in order)
        }
    }
    if (!plan) {
        { std::lock_guard<std::mutex> lk(g_prep_mu); plan = lz_take_cached_plan(g, p, ref_ids); }
        if (plan) lz_plan_alloc(*plan);
        else {
            plan.reset(new lz_plan);
            plan->g = g; plan->ref_ids = ref_ids;
            lz_plan_references(g, p, *plan);
        }
    }
    if (plan->built_ev) VG_HIP(hipStreamWaitEvent(s, plan->built_ev, 0));
    std::vector<lz_batch>& batches = plan->batches;
    dbuf<ref_desc>& d_refs = plan->d_refs;
    lz_slot& slot = plan->slot;
    // the task ranges of the batches (sorted task list: tasks of a reference are contiguous)
    for (auto& B : batches) {
        B.pos = ref_first[ref_ids[(size_t)B.first_ref]];
        const size_t ri_end = (size_t)B.first_ref + (size_t)B.n_refs;
        B.end = ri_end < ref_ids.size() ? ref_first[ref_ids[ri_end]] : n_tasks;
        B.bytes_alg = bytes_alg_all * (double)(B.end - B.pos) / (double)n_tasks;       // the call's SURVEY 8(d) bytes, by task share
        B.q_max = q_max; B.q_sum = (int64_t)(q_sum * (double)(B.end - B.pos) / (double)n_tasks);
    }
    vg_host_mark("lz: batches planned");
    hipStream_t sb = s;
    static const char* seg_env = vg_dev_getenv("VG_LZ_SEGMENTS");
    for (size_t bi = 0; bi < batches.size(); ++bi) {
        lz_batch& B = batches[bi];
        lz_slot& L = slot;
        if (!(bi == 0 && plan->batch0_built)) lz_build_batch(g, p, *plan, bi, sb);
        {
            const int64_t nt = B.end - B.pos;
            std::optional<vg_prof_scope> ps; ps.emplace("lz_parse", B.bytes_alg);
            // Four waves per pair (segments) shorten the critical path: worth it when the launch would
            // otherwise last as long as its slowest pair -- few tasks, or queries several times longer than
            // the average one (mixed contig sets).  With many uniform tasks one wave per pair keeps every
            // SIMD busy without the duplicated stretches.
            const bool uneven = B.q_max * nt > 3 * B.q_sum;
            const bool segments = !want_regions && (seg_env ? atoi(seg_env) > 1 : ((n_tasks <= g_segment_task_limit && nt <= g_segment_task_limit) || uneven));
            if (want_regions && !two_pass_regions) {
                // --out-aln: rows and regions from ONE parse (one wave per pair: regions leave in query order).  The arena is
                // sized from the batch (a chunk per task + a region per 256 query symbols: four times what diverged phage
                // families produce); should a batch need more, the cursor says exactly how much and the batch is repeated.
                const int64_t nblk = ((nt + 3) / 4 + 7) / 8 * 8;
                unsigned long long cap = (unsigned long long)nt * RCHUNK + (unsigned long long)(B.q_sum / 256) + 1024;
                static const long long cap_env = [] { const char* e = vg_dev_getenv("VG_LZ_ARENA"); return e ? atoll(e) : 0LL; }();      // developer switch (tests): a first arena of that many records
                if (cap_env > 0) cap = (unsigned long long)cap_env;
                {   // (a generous first arena must not be what makes a large call fail: at most an eighth of the free device memory;
                    // a batch that needs more says so through the cursor and is repeated with exactly what it needs)
                    size_t fr = 0, tot = 0;
                    if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) {
                        const unsigned long long most = std::max<unsigned long long>((unsigned long long)nt * RCHUNK / 4 + 1024, (unsigned long long)(fr / 8) / sizeof(region_rec));
                        if (cap > most) cap = most;
                    } else (void)hipGetLastError();
                }
                dbuf<unsigned long long> d_cur(1), d_first((size_t)nt + 1);
                dbuf<unsigned int> d_bad(1);
                for (int attempt = 0;; ++attempt) {
                    dbuf<region_rec> d_arena((size_t)cap);
                    d_cur.zero(s); d_bad.zero(s);
                    if (!ps) ps.emplace("lz_parse", B.bytes_alg);
                    if (fast_params) hipLaunchKernelGGL(k_lz_parse_fast_regions, dim3((unsigned)nblk), dim3(256), 0, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                       g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                       L.sent_pool.p, P, d_stats.p, d_arena.p, d_cur.p, cap, (const uint32_t*)nullptr);
                    else hipLaunchKernelGGL(k_lz_parse_regions, dim3((unsigned)nblk), dim3(256), 0, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                       g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                       L.sent_pool.p, P, d_stats.p, d_arena.p, d_cur.p, cap, (const uint32_t*)nullptr);
                    ps.reset();                                   // (the scope times the parse; what follows is the placing pass)
                    std::optional<vg_prof_scope> ps2; ps2.emplace("lz_regions_place", 0);      // (its kernels, not the downloads)
                    hipLaunchKernelGGL(k_region_counts, dim3(grid_for(nt + 1)), dim3(256), 0, s, (const task_dev*)(d_tasks.p + B.pos), nt, (const vg_pair_stat*)d_stats.p, d_first.p);
                    size_t tb = 0;
                    VG_HIP(rocprim::exclusive_scan(nullptr, tb, d_first.p, d_first.p, 0ULL, (size_t)nt + 1, rocprim::plus<unsigned long long>(), s));
                    dbuf<char> tmp(tb);
                    VG_HIP(rocprim::exclusive_scan((void*)tmp.p, tb, d_first.p, d_first.p, 0ULL, (size_t)nt + 1, rocprim::plus<unsigned long long>(), s));
                    ps2.reset();
                    unsigned long long used = 0, nr = 0;
                    d_cur.download(&used, 1, s);
                    VG_HIP(hipMemcpyAsync(&nr, d_first.p + nt, sizeof nr, hipMemcpyDeviceToHost, s));
                    VG_HIP(hipStreamSynchronize(s));
                    if (used > cap) {
                        if (attempt) throw vg_error(VG_EHIP, "internal error: the region arena of the LZ parse overflowed twice");
                        cap = used; continue;                     // (records beyond the arena were not written: once more, with room for all)
                    }
                    if (nr) {
                        dbuf<vg_region> d_regions((size_t)nr);
                        ps2.emplace("lz_regions_place", (double)used * 32.0 + (double)nr * 24.0);
                        hipLaunchKernelGGL(k_regions_place, dim3(grid_for((int64_t)used)), dim3(256), 0, s, (const region_rec*)d_arena.p, used,
                                           (const unsigned long long*)d_first.p, nr, d_regions.p, d_bad.p);
                        ps2.reset();
                        unsigned int bad = 0; d_bad.download(&bad, 1, s);
                        const size_t at = h_regions.size();
                        h_regions.resize(at + (size_t)nr);
                        d_regions.download(h_regions.data() + at, (size_t)nr, s);
                        VG_HIP(hipStreamSynchronize(s));
                        if (bad) throw vg_error(VG_EHIP, "internal error: region records of the LZ parse disagree with its rows");
                    }
                    break;
                }
                continue;
            }
            if (segments && P.ablate == 0 && nt < (1LL << 31)) {
                const int64_t nblk = (nt + 7) / 8 * 8;
                if (fast_params) hipLaunchKernelGGL(k_lz_parse_seg_fast, dim3((unsigned)nblk), dim3(256), 0, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                   g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                   L.sent_pool.p, P, d_stats.p, (region_rec*)nullptr, (unsigned long long*)nullptr, 0ULL, (const uint32_t*)nullptr);
                else hipLaunchKernelGGL(k_lz_parse_seg, dim3((unsigned)nblk), dim3(256), 0, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                   g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                   L.sent_pool.p, P, d_stats.p, (region_rec*)nullptr, (unsigned long long*)nullptr, 0ULL, (const uint32_t*)nullptr);
            } else {
                const int64_t nblk = ((nt + 3) / 4 + 7) / 8 * 8;
#ifdef VG_DEV_KERNELS
                if (P.ablate) {
                    hipLaunchKernelGGL(k_lz_parse_dev, dim3((unsigned)nblk), dim3(256), 0, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                   g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                   L.sent_pool.p, P, d_stats.p, (region_rec*)nullptr, (unsigned long long*)nullptr, 0ULL, (const uint32_t*)nullptr);
                } else
#endif
                static const bool fused_index = [] { const char* e = vg_dev_getenv("VG_LZ_INDEX"); return e && !strcmp(e, "fused"); }();      // EXPERIMENT
                if (fast_params && fused_index && p->msl == 7) {
                    // EXPERIMENT (profiles/r06_parse_fused_slots.md): the probes read fixed 32-byte bucket slots made from the
                    // ordinary index by a conversion pass of its own scope -- the probe side of a layout measured before any
                    // build kernel writes it
                    ps.reset();
                    if (!fused_pool.p) fused_pool.alloc((size_t)plan->all_refs.size() * FUSED_SLOTS * 8 + 8);
                    {
                        vg_prof_scope pf("lz_index_fuse", (double)B.n_refs * FUSED_SLOTS * 32.0);
                        hipLaunchKernelGGL(k_index_fuse, dim3((unsigned)std::min(B.n_refs, 256 * 16)), dim3(256), 0, s, (const ref_desc*)d_refs.p, B.first_ref, B.n_refs,
                                           (const uint32_t*)L.stab_pool.p, (const uint32_t*)L.sent_pool.p, fused_pool.p);
                    }
                    ps.emplace("lz_parse", B.bytes_alg);
                    hipLaunchKernelGGL(k_lz_parse_fast_fused, dim3((unsigned)nblk), dim3(256), 0, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                   g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                   L.sent_pool.p, P, d_stats.p, (region_rec*)nullptr, (unsigned long long*)nullptr, 0ULL, (const uint32_t*)fused_pool.p);
                } else
                if (fast_params) {
                    // (developer experiment: VG_LZ_OCC_KB reserves that much unused LDS per workgroup, i.e. caps the resident waves)
                    static const size_t occ_lds = [] { const char* e = vg_dev_getenv("VG_LZ_OCC_KB"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();
                    hipLaunchKernelGGL(k_lz_parse_fast, dim3((unsigned)nblk), dim3(256), occ_lds, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                   g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                   L.sent_pool.p, P, d_stats.p, (region_rec*)nullptr, (unsigned long long*)nullptr, 0ULL, (const uint32_t*)nullptr);
                } else {
                    hipLaunchKernelGGL(k_lz_parse, dim3((unsigned)nblk), dim3(256), 0, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                   g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                   L.sent_pool.p, P, d_stats.p, (region_rec*)nullptr, (unsigned long long*)nullptr, 0ULL, (const uint32_t*)nullptr);
                }
            }
            if (want_regions) {
                // VG_LZ_REGIONS=two-pass (developer switch, the checker of the one-parse path): the rows just computed give
                // every task's region count, the parse runs a SECOND time into an arena that is known to fit, and the same
                // placing pass orders the records
                h_stats.resize((size_t)n_tasks);
                d_stats.download(h_stats.data(), (size_t)n_tasks, s);
                VG_HIP(hipStreamSynchronize(s));
                std::vector<unsigned long long> off((size_t)nt + 1, 0);
                if (first_pass.empty()) first_pass.resize((size_t)n_tasks);
                unsigned long long chunks = 0;
                for (int64_t t = 0; t < nt; ++t) {
                    const uint32_t oi = td[(size_t)(B.pos + t)].out_idx;
                    first_pass[oi] = h_stats[oi];
                    off[(size_t)t + 1] = off[(size_t)t] + h_stats[oi].n_regions;
                    chunks += (h_stats[oi].n_regions + RCHUNK - 1) / RCHUNK;
                }
                const unsigned long long nr = off[(size_t)nt];
                if (nr) {
                    const unsigned long long cap = chunks * RCHUNK;
                    dbuf<unsigned long long> d_off((size_t)nt + 1), d_cur(1); d_off.upload(off.data(), off.size(), s);
                    dbuf<unsigned int> d_bad(1); d_cur.zero(s); d_bad.zero(s);
                    dbuf<region_rec> d_arena((size_t)cap);
                    dbuf<vg_region> d_regions((size_t)nr);
                    const int64_t nblk = ((nt + 3) / 4 + 7) / 8 * 8;
                    hipLaunchKernelGGL(k_lz_parse_regions, dim3((unsigned)nblk), dim3(256), 0, s, d_tasks.p + B.pos, nt, d_refs.p, d_planes, g->d_nmask.p,
                                   g->d_base_off.p, g->d_len.p, g->d_has_n.p, L.rr_pool.p, L.mask_pool.p, L.stab_pool.p,
                                   L.sent_pool.p, P, d_stats.p, d_arena.p, d_cur.p, cap, (const uint32_t*)nullptr);
                    hipLaunchKernelGGL(k_regions_place, dim3(grid_for((int64_t)cap)), dim3(256), 0, s, (const region_rec*)d_arena.p, cap,
                                       (const unsigned long long*)d_off.p, nr, d_regions.p, d_bad.p);
                    unsigned int bad = 0; d_bad.download(&bad, 1, s);
                    const size_t at = h_regions.size();
                    h_regions.resize(at + (size_t)nr);
                    d_regions.download(h_regions.data() + at, (size_t)nr, s);
                    VG_HIP(hipStreamSynchronize(s));
                    if (bad) throw vg_error(VG_EHIP, "internal error: the region pass and the counting pass of the LZ parse disagree");
                }
            }
        }
    }
    vg_host_mark("lz: launched");
    vg_deferred_start();                                      // (the host now waits for the kernels: parked clean-up runs beside them)
    VG_HIP(hipStreamSynchronize(s));
    VG_HIP(hipGetLastError());
    vg_host_mark("lz: kernels done");
    d_stats.download(stats, (size_t)n_tasks, s);
    VG_HIP(hipStreamSynchronize(s));
    vg_host_mark("lz: rows downloaded");
    if (want_regions) {
        // the region pass (general kernel) rewrote the rows: they must be the rows the slots were sized from
        for (int64_t t = 0; t < n_tasks && !first_pass.empty(); ++t)
            if (stats[t].n_regions != first_pass[(size_t)t].n_regions || stats[t].n_match != first_pass[(size_t)t].n_match || stats[t].aln_len != first_pass[(size_t)t].aln_len)
                throw vg_error(VG_EHIP, "internal error: the region pass and the counting pass of the LZ parse disagree on task " + std::to_string(t));
        vg_region* o = (vg_region*)malloc(sizeof(vg_region) * std::max<size_t>(1, h_regions.size()));
        if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
        if (!h_regions.empty()) memcpy(o, h_regions.data(), sizeof(vg_region) * h_regions.size());
        *regions = o; if (n_regions) *n_regions = (int64_t)h_regions.size();
    }
    lz_keep_plan_host_side(plan);
    VG_API_END
}


// ------------------------------------------------------------------ reference plan (see lz_plan)
namespace {
int64_t lz_ref_need(const vg_genomes* g, const vg_lz_params* p, uint32_t r, int64_t* chunks_out) {
    const int64_t stab_n = 1LL << (2 * p->msl);
    const int64_t n_rr = 2 * g->len[r] + 1;
    const int64_t chunks = (n_rr + RR_PAD + 31) / 32 + 2;
    if (chunks_out) *chunks_out = chunks;
    return chunks * 12 + (stab_n + n_rr) * 4;
}
// default budget 24 GiB; a set whose indexes all fit in twice that is built in ONE batch (every batch boundary is a
// tail of the parse launch with idle CUs: 100 k genomes, 40 GB of indexes, 43.5 vs 44.3 ms of parse)
int64_t lz_batch_budget(const vg_genomes* g, const vg_lz_params* p, const std::vector<uint32_t>& ref_ids) {
    int64_t batch_budget = g_index_budget_bytes;
    if (!g_index_budget_set && vg_one_shot()) {
        // a cold one-shot call (the CLI): one small set of index pools reused by every batch -- 6 GiB of device memory to
        // touch for the first time instead of 40 (vg_core.cpp), for a few batch boundaries (1-3 ms each)
        static const double gb = [] { const char* e = getenv("VG_ONESHOT_INDEX_GB"); const double v = e ? atof(e) : 0.0; return v >= 0.0625 ? v : 6.0; }();
        batch_budget = (int64_t)(gb * 1073741824.0);
    } else if (!g_index_budget_set) {
        int64_t all = 0;
        for (size_t ri = 0; ri < ref_ids.size() && all <= 2 * batch_budget; ++ri) all += lz_ref_need(g, p, ref_ids[ri], nullptr);
        if (all <= 2 * batch_budget) batch_budget = std::max(batch_budget, all);
    }
    return batch_budget;
}
// batches, descriptors, build lists, pools and the descriptor upload for P.ref_ids
void lz_plan_references(const vg_genomes* g, const vg_lz_params* p, lz_plan& P) {
    hipStream_t s = vg_stream();
    const std::vector<uint32_t>& ref_ids = P.ref_ids;
    const int64_t stab_n = 1LL << (2 * p->msl);
    P.mal = p->mal; P.msl = p->msl; P.g = g; P.n_genomes = g->n;
    const int64_t batch_budget = P.budget = lz_batch_budget(g, p, ref_ids);
    std::vector<lz_batch>& batches = P.batches;
    std::vector<ref_desc>& all_refs = P.all_refs;
    all_refs.assign(ref_ids.size(), ref_desc{});
    for (size_t ri = 0; ri < ref_ids.size();) {
        batches.emplace_back();
        lz_batch& B = batches.back();
        B.first_ref = (int)ri;
        int64_t bytes = 0;
        B.chunk_off.reserve(ref_ids.size() - ri + 1); B.reg_list.reserve(ref_ids.size() - ri);
        const int tag_want = 2 * (p->mal - p->msl);
        while (ri < ref_ids.size()) {
            const uint32_t r = ref_ids[ri];
            const int64_t L = g->len[r]; const int64_t n_rr = 2 * L + 1;
            int64_t chunks = 0; const int64_t need = lz_ref_need(g, p, r, &chunks);
            if (B.n_refs > 0 && bytes + need > batch_budget) break;
            ref_desc& rd = all_refs[ri];                         // (zeroed by the assign above)
            rd.rr_w = B.rr_words; rd.mask_w = B.mask_words; rd.stab = B.stab_tot; rd.sent = B.sent_n;
            rd.L = (int32_t)L; rd.n_rr = (int32_t)n_rr; rd.genome = (int32_t)r; rd.has_n = g->has_n[r];
            rd.pos_bits = n_rr <= 2 ? 1 : 64 - __builtin_clzll((unsigned long long)(n_rr - 1));       // the smallest pb >= 1 with 2^pb >= n_rr
            rd.tag_bits = std::max(0, std::min({ tag_want, 14, 32 - rd.pos_bits }));
            B.rr_words += chunks * 2; B.mask_words += chunks; B.stab_tot += stab_n; B.sent_n += n_rr;
            B.chunk_off.push_back(B.chunk_off.back() + chunks);
            bytes += need; ++B.n_refs; ++ri;
        }
        const int n_refs = B.n_refs;
        // split the batch: LDS counting sort for ordinary references, global path for the rest
        for (int i = 0; i < n_refs; ++i) {
            const int gi = B.first_ref + i;                      // ordinal = index into all_refs / the device array
            const bool small = all_refs[(size_t)gi].n_rr <= (1 << 21) && p->msl <= 7;
            if (small && all_refs[(size_t)gi].n_rr <= REG_MAX_RR && !g_no_reg_build) B.reg_list.push_back(gi);
            else if (small && all_refs[(size_t)gi].n_rr <= MID_MAX_RR && !g_no_reg_build) B.mid_list.push_back(gi);
            else if (small) B.small_list.push_back(gi);
            else { B.large_list.push_back(gi); B.large_chunks.push_back(B.large_chunks.back() + (B.chunk_off[(size_t)i + 1] - B.chunk_off[(size_t)i])); }
        }
        // longest references first: the persistent workgroups take them round-robin, so their loads even out
        // (nothing to even out when the lengths are within 25 % of each other)
        auto longest_first = [&](std::vector<int>& list) {
            int32_t mn = INT32_MAX, mx = 0;
            for (int i : list) { mn = std::min(mn, all_refs[(size_t)i].n_rr); mx = std::max(mx, all_refs[(size_t)i].n_rr); }
            if (!list.empty() && (int64_t)mx * 4 > (int64_t)mn * 5 && (int64_t)(mx - mn) <= 8 * (int64_t)list.size() + 65536) {
                // a stable counting sort on the length, descending (10^6 references: a comparison sort costs 40 ms)
                std::vector<int64_t> at((size_t)(mx - mn) + 2, 0);
                for (int i : list) at[(size_t)(mx - all_refs[(size_t)i].n_rr) + 1]++;
                for (size_t b = 1; b < at.size(); ++b) at[b] += at[b - 1];
                std::vector<int> sorted(list.size());
                for (int i : list) sorted[(size_t)at[(size_t)(mx - all_refs[(size_t)i].n_rr)]++] = i;
                list.swap(sorted);
            } else if (!list.empty() && (int64_t)mx * 4 > (int64_t)mn * 5) {
                std::vector<uint64_t> keyed(list.size());
                for (size_t i = 0; i < keyed.size(); ++i) keyed[i] = ((uint64_t)(0x7fffffffu - (uint32_t)all_refs[(size_t)list[i]].n_rr) << 32) | (uint32_t)list[i];
                std::sort(keyed.begin(), keyed.end());           // length descending, ordinal ascending (= the stable order)
                for (size_t i = 0; i < keyed.size(); ++i) list[i] = (int)(uint32_t)keyed[i];
            }
        };
        longest_first(B.reg_list); longest_first(B.mid_list); longest_first(B.small_list);
        {   // the register build comes in six sizes (24 .. 4 trips of 4 096 positions): a reference takes the smallest that
            // holds it, so a 4 kb contig does not pay the register passes of a 49 kb genome; classes are contiguous in the list
            std::vector<int> by_class[6];
            for (int i : B.reg_list) {
                const int trips = (all_refs[(size_t)i].n_rr + 256 + 4095) / 4096;        // (REG_MAX_RR leaves 256 symbols of slack)
                const int cls = trips > 20 ? 0 : trips > 16 ? 1 : trips > 12 ? 2 : trips > 8 ? 3 : trips > 4 ? 4 : 5;
                by_class[cls].push_back(i);
            }
            B.reg_list.clear();
            for (int c = 0; c < 6; ++c) { B.reg_list.insert(B.reg_list.end(), by_class[c].begin(), by_class[c].end()); B.reg_class_end[c] = (int)B.reg_list.size(); }
        }
        if (!B.small_list.empty()) {
            int64_t max_rr = 0; for (int i : B.small_list) max_rr = std::max<int64_t>(max_rr, all_refs[(size_t)i].n_rr);
            B.nblk_build = (int)std::min<size_t>(B.small_list.size(), 512);
            B.stride = (max_rr + 63) / 64 * 64;
            B.scratch_words = (int64_t)B.nblk_build * B.stride * 3;
        }
    }
    vg_host_mark("lz plan: batches cut");
    lz_plan_alloc(P);
}
// the device side of a plan: the reference descriptors of the whole call go up once; one set of index buffers, sized for
// the largest batch, is reused by every batch.  (Also what a plan taken from the host-side cache still needs.)
void lz_plan_alloc(lz_plan& P) {
    hipStream_t s = vg_stream();
    std::vector<lz_batch>& batches = P.batches;
    std::vector<ref_desc>& all_refs = P.all_refs;
    P.d_refs.alloc(std::max<size_t>(1, all_refs.size()));
    if (!all_refs.empty()) P.d_refs.upload(all_refs.data(), all_refs.size(), s);
    vg_host_mark("lz plan: descriptors queued");
    size_t m_rr = 0, m_mask = 0, m_stab = 1, m_sent = 0, m_scr = 1, m_reg = 1, m_mid = 1, m_small = 1, m_large = 1, m_lch = 1;
    for (auto& B : batches) {
        m_rr = std::max(m_rr, (size_t)B.rr_words); m_mask = std::max(m_mask, (size_t)B.mask_words);
        m_stab = std::max(m_stab, (size_t)B.stab_tot); m_sent = std::max(m_sent, (size_t)B.sent_n); m_scr = std::max(m_scr, (size_t)B.scratch_words);
        m_reg = std::max(m_reg, B.reg_list.size()); m_mid = std::max(m_mid, B.mid_list.size()); m_small = std::max(m_small, B.small_list.size()); m_large = std::max(m_large, B.large_list.size());
        m_lch = std::max(m_lch, B.large_chunks.size());
    }
    lz_slot& L = P.slot;
    L.rr_pool.alloc(m_rr + 8); L.mask_pool.alloc(m_mask + 8); L.stab_pool.alloc(m_stab); L.sent_pool.alloc(m_sent + 4);
    L.scratch.alloc(m_scr); L.d_reg.alloc(m_reg); L.d_mid.alloc(m_mid); L.d_small.alloc(m_small); L.d_large.alloc(m_large); L.d_lchunk.alloc(m_lch);
}
// RR, bucket tables and entries of the references of batch bi into the plan's pools (asynchronous on sb)
void lz_build_batch(const vg_genomes* g, const vg_lz_params* p, lz_plan& P, size_t bi, hipStream_t sb) {
    lz_batch& B = P.batches[bi];
    lz_slot& L = P.slot;
    dbuf<ref_desc>& d_refs = P.d_refs;
    if (!B.reg_list.empty()) L.d_reg.upload(B.reg_list.data(), B.reg_list.size(), sb);
    if (!B.mid_list.empty()) L.d_mid.upload(B.mid_list.data(), B.mid_list.size(), sb);
    if (!B.small_list.empty()) L.d_small.upload(B.small_list.data(), B.small_list.size(), sb);
    if (!B.large_list.empty()) { L.d_large.upload(B.large_list.data(), B.large_list.size(), sb); L.d_lchunk.upload(B.large_chunks.data(), B.large_chunks.size(), sb); }
    const int64_t total_chunks = B.chunk_off.back();
    const uint32_t* gpl = vg_genome_planes(g, sb);            // the builds read the genomes as bit planes
    vg_prof_scope ps("lz_build_index", (double)total_chunks * 32 * (0.375 + 0.375 + 4), sb);
    if (!B.large_list.empty()) VG_HIP(hipMemsetAsync(L.stab_pool.p, 0, (size_t)B.stab_tot * sizeof(uint32_t), sb));
    if (!B.reg_list.empty()) {
        auto launch = [&](auto kern, int first, int end) {
            if (end > first) hipLaunchKernelGGL(kern, dim3((unsigned)std::min(end - first, 512)), dim3(1024), 0, sb, d_refs.p, L.d_reg.p + first,
                                                end - first, gpl, g->d_nmask.p, g->d_base_off.p, L.rr_pool.p, L.mask_pool.p, p->msl, L.stab_pool.p, L.sent_pool.p);
        };
        const int* ce = B.reg_class_end;
        launch(k_build_index_reg<24>, 0, ce[0]); launch(k_build_index_reg<20>, ce[0], ce[1]); launch(k_build_index_reg<16>, ce[1], ce[2]);
        launch(k_build_index_reg<12>, ce[2], ce[3]); launch(k_build_index_reg<8>, ce[3], ce[4]); launch(k_build_index_reg<4>, ce[4], ce[5]);
    }
    if (!B.mid_list.empty()) {
        hipLaunchKernelGGL(k_build_index_mid, dim3((unsigned)std::min<size_t>(B.mid_list.size(), 512)), dim3(1024), 0, sb, d_refs.p, L.d_mid.p,
                           (int)B.mid_list.size(), gpl, g->d_nmask.p, g->d_base_off.p, L.rr_pool.p, L.mask_pool.p, p->msl,
                           L.stab_pool.p, L.sent_pool.p);
    }
    if (!B.small_list.empty()) {
        hipLaunchKernelGGL(k_build_index_lds, dim3(B.nblk_build), dim3(1024), 0, sb, d_refs.p, L.d_small.p, (int)B.small_list.size(),
                           gpl, g->d_nmask.p, g->d_base_off.p, L.rr_pool.p, L.mask_pool.p, p->mal, p->msl,
                           L.stab_pool.p, L.sent_pool.p, L.scratch.p, B.stride);
    }
    if (!B.large_list.empty()) {
        const int nl = (int)B.large_list.size(); const int64_t lc = B.large_chunks.back();
        hipLaunchKernelGGL(k_build_rr, dim3(grid_for(lc)), dim3(256), 0, sb, d_refs.p, L.d_large.p, nl, L.d_lchunk.p, gpl,
                           g->d_nmask.p, g->d_base_off.p, L.rr_pool.p, L.mask_pool.p);
        hipLaunchKernelGGL(k_index_pass, dim3(grid_for(lc * 32)), dim3(256), 0, sb, d_refs.p, L.d_large.p, nl, L.d_lchunk.p, L.rr_pool.p,
                           L.mask_pool.p, p->mal, p->msl, 0, L.stab_pool.p, L.sent_pool.p);
        hipLaunchKernelGGL(k_scan_tables, dim3(nl), dim3(256), 0, sb, d_refs.p, L.d_large.p, p->msl, L.stab_pool.p);
        hipLaunchKernelGGL(k_index_pass, dim3(grid_for(lc * 32)), dim3(256), 0, sb, d_refs.p, L.d_large.p, nl, L.d_lchunk.p, L.rr_pool.p,
                           L.mask_pool.p, p->mal, p->msl, 1, L.stab_pool.p, L.sent_pool.p);
    }
}
}  // namespace

// Optional head start of vg_lz_align: the reference indexes of the genomes named by the candidate pairs (every genome of
// a pair is a reference of one of its two tasks) are planned and the first batch is queued on the device NOW, so that
// the build runs while the caller assembles the canonical task list (vg_align_tasks) and vg_lz_align groups it: host
// work that otherwise leaves the device idle.  vg_lz_align takes the plan over when its task list names exactly these
// references under the same parameters; otherwise the plan is dropped and nothing is lost but the build.
extern "C" int vg_lz_prepare(vg_genomes* g, const vg_pair_count* pairs, int64_t n_pairs, const vg_lz_params* p) {
    VG_API_BEGIN
    if (!g || (!pairs && n_pairs) || !p) throw vg_error(VG_EINVAL, "vg_lz_prepare: null argument");
    if (p->mal < 8 || p->mal > 31 || p->msl < 4 || p->msl > 12 || p->msl > p->mal) throw vg_error(VG_EINVAL, "mal must be 8..31, msl 4..12 and <= mal");
    vg_require_device();
    int rc = vg_genomes_to_device(g); if (rc) return rc;
    vg_lz_drop_prepared(nullptr);
    if (n_pairs == 0) return VG_OK;
    for (int i = 0; i < g->n; ++i) if (g->len[i] > (1 << 29)) throw vg_error(VG_EOVERFLOW, "genome longer than 2^29 bases");
    std::vector<uint8_t> is_ref((size_t)g->n, 0);
    for (int64_t i = 0; i < n_pairs; ++i) {
        if (pairs[i].a >= (uint32_t)g->n || pairs[i].b >= (uint32_t)g->n) throw vg_error(VG_EINVAL, "pair id out of range");
        is_ref[pairs[i].a] = 1; is_ref[pairs[i].b] = 1;
    }
    std::vector<uint32_t> ref_ids;
    for (int i = 0; i < g->n; ++i) if (is_ref[(size_t)i]) ref_ids.push_back((uint32_t)i);
    vg_host_mark("lz prepare: references listed");
    std::unique_ptr<lz_plan> plan;
    { std::lock_guard<std::mutex> lk(g_prep_mu); plan = lz_take_cached_plan(g, p, ref_ids); }
    if (plan) { lz_plan_alloc(*plan); vg_host_mark("lz prepare: plan of the last call taken over"); }
    else {
        plan.reset(new lz_plan);
        plan->ref_ids = std::move(ref_ids);
        plan->g = g;
        lz_plan_references(g, p, *plan);
        vg_host_mark("lz prepare: planned");
    }
    // (developer experiment VG_LZ_PREPARE_QUEUE=own: the build runs on a queue of its own, beside whatever the caller
    // launches next on the library's queue -- a prefilter pass, if the references are known before it)
    static const bool own_queue = [] { const char* e = vg_dev_getenv("VG_LZ_PREPARE_QUEUE"); return e && !strcmp(e, "own"); }();
    static const bool at_spgemm = [] { const char* e = vg_dev_getenv("VG_LZ_PREPARE_QUEUE"); return e && !strcmp(e, "spgemm"); }();
    if (at_spgemm) {
        // the build is queued (own queue) when the next prefilter pass reaches its SpGEMM: random reads beside LDS-staged writes
        lz_plan* raw = plan.get(); const vg_lz_params pp = *p;
        VG_HIP(hipEventCreateWithFlags(&plan->built_ev, hipEventDisableTiming));
        VG_HIP(hipEventRecord(plan->built_ev, vg_stream()));                   // (a pass that never comes: the event is complete anyway)
        raw->batch0_built = false;
        vg_set_spgemm_hook([g, pp, raw] {
            static hipStream_t q = nullptr;
            if (!q && hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) return;
            hipEvent_t e0 = nullptr; if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess) return;
            (void)hipEventRecord(e0, vg_stream()); (void)hipStreamWaitEvent(q, e0, 0); (void)hipEventDestroy(e0);
            lz_build_batch(g, &pp, *raw, 0, q);
            (void)hipEventRecord(raw->built_ev, q);
            raw->batch0_built = true;
        });
        vg_host_mark("lz: first batch of indexes deferred to the SpGEMM");
        std::lock_guard<std::mutex> lk(g_prep_mu);
        g_prepared = std::move(plan);
        return VG_OK;
    }
    if (own_queue) {
        static hipStream_t q = nullptr;
        if (!q) VG_HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        hipEvent_t e0 = nullptr; VG_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
        VG_HIP(hipEventRecord(e0, vg_stream())); VG_HIP(hipStreamWaitEvent(q, e0, 0)); (void)hipEventDestroy(e0);      // (the pools may have work of the library queue pending)
        lz_build_batch(g, p, *plan, 0, q);
        VG_HIP(hipEventCreateWithFlags(&plan->built_ev, hipEventDisableTiming));
        VG_HIP(hipEventRecord(plan->built_ev, q));
    } else
    lz_build_batch(g, p, *plan, 0, vg_stream());
    plan->batch0_built = true;
    vg_host_mark("lz: first batch of indexes queued");
    std::lock_guard<std::mutex> lk(g_prep_mu);
    g_prepared = std::move(plan);
    VG_API_END
}

extern "C" void vg_set_index_budget(int64_t bytes) { if (bytes > (64 << 20)) { g_index_budget_bytes = bytes; g_index_budget_set = true; } }

// (see vg_warm_prefilter)
namespace { __global__ void k_warm_align() {} }
void vg_warm_align(hipStream_t s) { hipLaunchKernelGGL(k_warm_align, dim3(1), dim3(64), 0, s); }
