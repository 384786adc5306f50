// vg_prefilter.hip — Kmer-db prefilter on gfx950: canonical k-mer extraction from the genomes' bit planes, the
// inverted index by an own two-level MSD partition + LDS bucket sort (no general radix sort on the hot path), and
// the sparse genome x genome shared-k-mer matrix as a row-wise SpGEMM (A * A^T) with LDS hash accumulators.
// Replaces `kmer-db build` + `all2all-sp` (vclust.py:953-1017); restates SURVEY §8a K1/K2, parity-checked against
// oracle/prefilter_oracle.c.
//
// All kernels are HBM / latency bound integer kernels (no MFMA).  Per padded base position the pipeline reads 3 bits
// of sequence (twice: histogram and scatter), writes and reads one 8-byte level-1 record and one 8-byte level-2
// record, and writes one u32 of genome list (CSC side) and one u32 row pointer (CSR side: 0, or 1 + the start of
// the k-mer's run in the genome list); the random traffic is the row-pointer scatter of the bucket kernel and the
// reads of the short genome lists of shared k-mers in the SpGEMM.
#include "vg_common.h"
#include <functional>
#include <map>
#include <mutex>
#include <optional>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

constexpr uint64_t SENT = ~0ULL;
// k-mers are stored scrambled by a bijection of the 2k-bit canonical value that spreads any composition bias
// over the HIGH key bits (the partition digits): one Feistel round on the two k-bit halves,
//   (H, L) -> (H ^ top k bits of (L * SCRAMBLE32 mod 2^32), L)
// -- one 32-bit multiply instead of the 64-bit one of key = canonical * odd mod 4^k (four quarter-rate
// instructions per k-mer, three passes over every k-mer of the set).
constexpr uint32_t SCRAMBLE32 = 0x9E3779B1u;
__host__ __device__ __forceinline__ uint64_t scramble_key(uint64_t cano, int k) {
    const uint32_t L = (uint32_t)cano & ((1u << k) - 1u);
    const uint32_t H = (uint32_t)(cano >> k);
    return ((uint64_t)(H ^ ((L * SCRAMBLE32) >> (32 - k))) << k) | L;
}
__host__ __device__ __forceinline__ uint64_t unscramble_key(uint64_t key, int k) { return scramble_key(key, k); }   // an involution
constexpr uint32_t DUP_BIT = 0x80000000u;
// Row pointers: one u32 per base position (dense) or kept k-mer (compact): 0 = nothing to do, else 1 + the start
// of the k-mer's run in gen[].  The run is ascending in genome id and contains the position's own genome, so a
// walk from the start needs no length: it ends at the first genome >= a.

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

// reverse the order of the 32 two-bit groups of x
__device__ __forceinline__ uint64_t rev2(uint64_t x) {
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}

// ------------------------------------------------------------------ K1: canonical k-mers
struct kmer_args {
    const uint32_t* planes; const uint32_t* nmask; const uint32_t* blk2g; const int64_t* base_off; const int64_t* len;
    int64_t P; int k; int use_frac; uint64_t frac_thr; uint32_t shard, n_shards; int blk_shift;
    uint32_t dig_lo, dig_n;          // RANGE shards: keep the keys whose top DIG_BITS bits d satisfy d - dig_lo < dig_n (all: 0, 1 << DIG_BITS)
    uint32_t dig_max;                // RANGE shards: the widest digit range of any shard of this cut (buffers of all passes are sized alike)
};
// A k-mer range shard is one of two things (shard_mode, one rule for every rank and pass of a set):
//  RANGE  whole set below 2^32 padded bases, no --kmers-fraction: shard s of S owns the keys whose top 11 bits -- the
//         level-1 digit of the bucket pipeline -- fall into [s * 2048 / S, (s + 1) * 2048 / S).  The pass keeps the DENSE
//         kernels (k-mers from the packed bases, 8-byte records, a bucket's segment of a tile as long as in a whole
//         pass: 16 records, whatever S is) and only a 1/S slice of every table and record buffer exists.
//  HASH   otherwise (sets beyond 2^32 bases, fractions): a second multiplicative hash of the key's low half picks the
//         shard; the kept k-mers are materialised first (compact source).  Sub-sampling every bucket thins the buckets,
//         which is what a set of 25 G bases needs: its final buckets would hold 6 000 entries otherwise.
constexpr int DIG_BITS = 11;

// scrambled canonical k-mer starting at padded base position p, or SENT (window crosses the genome
// end, contains N, or the k-mer is not kept by --kmers-fraction / belongs to another shard).  Every genome
// is followed by at least one masked padding base, so "crosses the end" IS "contains a masked base": the
// mask is the only validity test (no length / offset lookups behind the genome id).
//
// The bases are read as BIT PLANES (vg_genomes::d_planes: the word pair (lo, hi) of every 32 bases): the k-mer at a
// position is the k-bit windows (fl, fh) of the two planes, its reverse complement the bit-reversed, inverted windows,
// and the two k-bit halves the key is made of ARE the planes: canonical = the smaller of (fh : fl) and (rh : rl) as
// 2k-bit numbers -- a bijection of the sequence, chosen the same way from either strand, which is all the grouping of
// equal k-mers needs (round 5: the 2-bit codes cost a per-dword reversal of 2-bit groups, 64-bit shifts and a 64-bit
// compare per k-mer, three times per position of the set).  Only --kmers-fraction sees the k-mer as the NUMBER the
// oracle hashes (2-bit codes, first base most significant): that path re-assembles it (cano_codes).
__device__ __forceinline__ uint64_t spread32(uint32_t v) {           // bit i -> bit 2 i
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    return (x | (x << 1)) & 0x5555555555555555ULL;
}
// the canonical k-mer as the oracle's number: 2-bit codes, first base most significant, the smaller strand
__host__ __device__ __forceinline__ uint64_t cano_codes_of(uint64_t x /* 2-bit codes, first base in the low bits */, int k) {
    uint64_t fwd = 0, rc = 0;
    for (int i = 0; i < k; ++i) { const uint64_t c = (x >> (2 * i)) & 3ULL; fwd = (fwd << 2) | c; rc |= (3ULL - c) << (2 * i); }
    return fwd < rc ? fwd : rc;
}
__device__ __forceinline__ uint64_t cano_codes(uint32_t fl, uint32_t fh, int k) {
    const int k2 = 2 * k;
    const uint64_t x = spread32(fl) | (spread32(fh) << 1);                  // first base in the low bits
    const uint64_t km = (1ULL << k2) - 1ULL;
    const uint64_t fwd = rev2(x) >> (64 - k2);                               // first base most significant
    const uint64_t rc = ~x & km;
    return fwd < rc ? fwd : rc;
}
// fl, fh = the planes of the k-mer in their k low bits (first base = bit 0; the bits above are ignored) -> scrambled
// canonical key or SENT
__device__ __forceinline__ uint64_t canon_key(const kmer_args& A, uint32_t fl, uint32_t fh) {
    const int k = A.k;                                                       // 8 .. 31
    const uint32_t nk = (1u << k) - 1u;
    fl &= nk; fh &= nk;
    const uint32_t rl = __brev(~fl) >> (32 - k), rh = __brev(~fh) >> (32 - k);      // (the ones above bit k - 1 of ~f leave at the low end)
    const bool f = fh < rh || (fh == rh && fl < rl);
    const uint32_t H = f ? fh : rh, L = f ? fl : rl;
    if (A.use_frac && !(mix64(cano_codes(fl, fh, k)) < A.frac_thr)) return SENT;
    const uint64_t key = ((uint64_t)(H ^ ((L * SCRAMBLE32) >> (32 - k))) << k) | L;      // = scramble_key((H << k) | L): bit 2k stays 0; SENT has it set
    if (A.dig_n < (1u << DIG_BITS)) {
        if ((uint32_t)(key >> (2 * A.k - DIG_BITS)) - A.dig_lo >= A.dig_n) return SENT;     // RANGE shard
    } else if (A.n_shards > 1) {
        // HASH shard = a cheap second multiplicative hash of the key's LOW half (one 32-bit multiply): it
        // must not be a function of the high bits, which the partition relies on being uniform
        const uint32_t h2 = (uint32_t)key * 0x85ebca6bu;
        if ((uint32_t)(((uint64_t)h2 * A.n_shards) >> 32) != A.shard) return SENT;
    }
    return key;
}
// the k-mer at base position p of a plane array (global or LDS): the word pairs of its 32-base chunk and of the next one
__device__ __forceinline__ uint64_t canon_key_at(const kmer_args& A, const uint32_t* pl, uint32_t pair /* p >> 5 */, uint32_t sh /* p & 31 */) {
    const uint32_t* w = pl + 2 * (size_t)pair;
    return canon_key(A, __builtin_amdgcn_alignbit(w[2], w[0], sh), __builtin_amdgcn_alignbit(w[3], w[1], sh));
}
// the four k-mers starting at padded positions p0 .. p0+3 (p0 a multiple of 4) from (l0, h0, l1, h1) = the planes of the
// 32-base chunk that holds p0 and of the next one, and m0, m1 = the N-mask words of the same chunks
__device__ __forceinline__ void kmers4_words(const kmer_args& A, uint32_t p0_low, uint32_t l0, uint32_t h0, uint32_t l1, uint32_t h1,
                                             uint32_t m0, uint32_t m1, uint64_t out[4]) {
    const uint32_t sh = p0_low & 31u;                                     // <= 28
    const uint32_t nk = (1u << A.k) - 1u;
    if (A.k <= 29) {
        // the four windows are bits j .. j + k - 1 <= 31 of the 32 bases from p0 on: one funnel shift per plane and one
        // for the mask serve all four, and a wave whose positions are all valid (every wave but the few that touch a
        // genome end or an N) runs the four k-mers as straight-line code
        const uint32_t XL = __builtin_amdgcn_alignbit(l1, l0, sh), XH = __builtin_amdgcn_alignbit(h1, h0, sh);
        const uint32_t XM = __builtin_amdgcn_alignbit(m1, m0, sh);
        if (__all((XM & ((8u << A.k) - 1u)) == 0)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = canon_key(A, XL >> j, XH >> j);
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = (((XM >> j) & nk) == 0) ? canon_key(A, XL >> j, XH >> j) : SENT;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint64_t key = SENT;
        if ((__builtin_amdgcn_alignbit(m1, m0, sh + j) & nk) == 0)
            key = canon_key(A, __builtin_amdgcn_alignbit(l1, l0, sh + j), __builtin_amdgcn_alignbit(h1, h0, sh + j));
        out[j] = key;
    }
}
__device__ __forceinline__ uint64_t kmer_at(const kmer_args& A, int64_t p, uint32_t* genome) {
    *genome = A.blk2g[p >> A.blk_shift];
    const int64_t mw = p >> 5; const uint32_t msh = (uint32_t)(p & 31);
    const uint64_t m = ((uint64_t)A.nmask[mw] | ((uint64_t)A.nmask[mw + 1] << 32)) >> msh;
    if ((m & ((1ULL << A.k) - 1)) != 0) return SENT;
    uint4 w; __builtin_memcpy(&w, A.planes + 2 * (p >> 5), 16);
    return canon_key(A, __builtin_amdgcn_alignbit(w.z, w.x, msh), __builtin_amdgcn_alignbit(w.w, w.y, msh));
}

// the four k-mers starting at padded positions p0 .. p0+3 (p0 a multiple of 4): one genome lookup,
// the planes of two chunks and one 64-bit N window serve all four
__device__ __forceinline__ void kmers4(const kmer_args& A, int64_t p0, uint64_t out[4], uint32_t* genome) {
    *genome = A.blk2g[p0 >> A.blk_shift];
    const int64_t mw = p0 >> 5;
    uint4 w; __builtin_memcpy(&w, A.planes + 2 * mw, 16);
    kmers4_words(A, (uint32_t)(p0 & 31), w.x, w.y, w.z, w.w, A.nmask[mw], A.nmask[mw + 1], out);
}

// Dense form (all k-mers kept: one shard, fraction 1): four consecutive padded base positions per
// thread (one sequence window for the four), 32 contiguous bytes of keys per thread.
__global__ void __launch_bounds__(256)
k_kmer_extract(kmer_args A, uint64_t* __restrict__ keys, unsigned long long* __restrict__ n_valid,
               int* __restrict__ kept_per_genome) {
    const int lane = threadIdx.x & 63;
    unsigned long long local_valid = 0;
    for (int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; p0 < A.P + 252; p0 += (int64_t)gridDim.x * blockDim.x * 4) {
        uint64_t kk[4] = {SENT, SENT, SENT, SENT}; uint32_t g = 0;
        const bool in = p0 < A.P;                                   // P is a multiple of 64 (so of 4)
        if (in) {
            kmers4(A, p0, kk, &g);
            uint4* o = reinterpret_cast<uint4*>(keys + p0);
            o[0] = make_uint4((uint32_t)kk[0], (uint32_t)(kk[0] >> 32), (uint32_t)kk[1], (uint32_t)(kk[1] >> 32));
            o[1] = make_uint4((uint32_t)kk[2], (uint32_t)(kk[2] >> 32), (uint32_t)kk[3], (uint32_t)(kk[3] >> 32));
        }
        const int mine = (kk[0] != SENT) + (kk[1] != SENT) + (kk[2] != SENT) + (kk[3] != SENT);
        local_valid += mine;
        const uint32_t g0 = __shfl(g, 0);
        if (__all(g == g0 || !in)) {
            int tot = mine;
            for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
            if (lane == 0 && tot) atomicAdd(&kept_per_genome[g0], tot);
        } else if (mine) atomicAdd(&kept_per_genome[g], mine);
    }
    for (int o = 32; o > 0; o >>= 1) local_valid += __shfl_down(local_valid, o);
    if (lane == 0 && local_valid) atomicAdd(n_valid, local_valid);
}

// Compact form (k-mer range shards, --kmers-fraction): only kept k-mers are written, in position
// order.  Pass 1 records per 64-position wave the ballot of kept lanes and its popcount; after an
// exclusive scan of the popcounts pass 2 recomputes the k-mers (cheaper than a dense key array) and
// writes (key, c) at c = wave_base + rank: the compact index c is the k-mer's row number, so positions
// never need more than the 32 bits of c (P itself may exceed 2^32).
// bit i of x (i < 16) -> bit 4 i
__device__ __forceinline__ unsigned long long spread16x4(unsigned long long x) {
    x = (x | (x << 24)) & 0x000000ff000000ffULL;
    x = (x | (x << 12)) & 0x000f000f000f000fULL;
    x = (x | (x << 6)) & 0x0303030303030303ULL;
    x = (x | (x << 3)) & 0x1111111111111111ULL;
    return x;
}

// four positions per thread: a wave covers 256 consecutive positions = four 64-position masks
// KC > 0: k = KC and no --kmers-fraction as compile-time constants (shift counts and masks of the k-mer arithmetic fold)
template <int KC>
__global__ void __launch_bounds__(256)
k_kmer_count(kmer_args A, unsigned long long* __restrict__ wave_mask, uint32_t* __restrict__ wave_cnt,
             int* __restrict__ kept_per_genome, uint64_t* __restrict__ stage, int stage_cap, unsigned int* __restrict__ stage_over) {
    if (KC > 0) { A.k = KC; A.use_frac = 0; }
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ULL << lane) - 1ULL;
    for (int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; p0 < A.P + 252; p0 += (int64_t)gridDim.x * blockDim.x * 4) {
        // P is a multiple of 64, not of 256: the last wave may be partly outside
        uint64_t kk[4] = {SENT, SENT, SENT, SENT}; uint32_t g = 0;
        if (p0 < A.P) kmers4(A, p0, kk, &g);
        const unsigned long long b0 = __ballot(kk[0] != SENT), b1 = __ballot(kk[1] != SENT);
        const unsigned long long b2 = __ballot(kk[2] != SENT), b3 = __ballot(kk[3] != SENT);
        const int64_t wbase = (p0 - 4 * lane) >> 6;                    // first of the wave's four masks
        if (lane < 4 && ((wbase + lane) << 6) < A.P) {
            const int q = 16 * lane;
            const unsigned long long mk = spread16x4((b0 >> q) & 0xffffULL) | (spread16x4((b1 >> q) & 0xffffULL) << 1)
                                        | (spread16x4((b2 >> q) & 0xffffULL) << 2) | (spread16x4((b3 >> q) & 0xffffULL) << 3);
            wave_mask[wbase + lane] = mk; wave_cnt[wbase + lane] = (uint32_t)__popcll(mk);
        }
        const int mine = (kk[0] != SENT) + (kk[1] != SENT) + (kk[2] != SENT) + (kk[3] != SENT);
        if (stage && mine) {
            // the kept k-mers of the wave's 256 positions, compacted in position order into the chunk's slots:
            // the emit pass becomes a copy (k_kmer_gather) instead of a second k-mer computation
            int r = __popcll(b0 & below) + __popcll(b1 & below) + __popcll(b2 & below) + __popcll(b3 & below);
            uint64_t* dst = stage + (size_t)(wbase >> 2) * stage_cap;
            bool over = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (kk[j] != SENT) { if (r < stage_cap) dst[r] = kk[j]; else over = true; ++r; }
            if (over) atomicOr(stage_over, 1u);
        }
        const uint32_t g0 = __shfl(g, 0);
        if (__all(g == g0 || p0 >= A.P)) {
            const int tot = __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
            if (lane == 0 && tot) atomicAdd(&kept_per_genome[g0], tot);
        } else if (mine) atomicAdd(&kept_per_genome[g], mine);
    }
}

// staged k-mers of every 256-position chunk -> their final places (c = row number = sort payload)
__global__ void __launch_bounds__(256)
k_kmer_gather(const uint64_t* __restrict__ stage, int stage_cap, const uint32_t* __restrict__ wave_base, int64_t W,
              uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int64_t n_chunks = (W + 3) >> 2;
    for (int64_t ch = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; ch < n_chunks; ch += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const uint32_t base = wave_base[ch << 2];
        const uint32_t cnt = wave_base[min<int64_t>((ch + 1) << 2, W)] - base;
        const uint64_t* src = stage + (size_t)ch * stage_cap;
        for (uint32_t t = lane; t < cnt; t += 64) { keys[base + t] = src[t]; if (pos) pos[base + t] = base + t; }
    }
}

__global__ void __launch_bounds__(256)
k_kmer_emit(kmer_args A, const unsigned long long* __restrict__ wave_mask, const uint32_t* __restrict__ wave_base,
            uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < A.P; p += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long m = wave_mask[p >> 6];
        if (!((m >> (p & 63)) & 1ULL)) continue;
        uint32_t g;
        const uint64_t key = kmer_at(A, p, &g);
        const uint32_t c = wave_base[p >> 6] + (uint32_t)__popcll(m & ((1ULL << (p & 63)) - 1ULL));
        keys[c] = key; if (pos) pos[c] = c;                      // payload = compact index (row number)
    }
}

// Sparse form of the emit pass (few k-mers kept: many shards / small fractions): a wave takes
// eight 64-position masks, lists their set bits and gives every lane one KEPT position, so the
// k-mer is recomputed only where it is written.
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int r) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const int c = __popcll(m & ((1ULL << w) - 1ULL));
        if (r >= c) { r -= c; m >>= w; pos += w; }
    }
    return pos;
}

__global__ void __launch_bounds__(256)
k_kmer_emit_sparse(kmer_args A, const unsigned long long* __restrict__ wave_mask, const uint32_t* __restrict__ wave_base,
                   uint64_t* __restrict__ keys, uint32_t* __restrict__ pos, int nw /* 64-position words per wave and trip, 1 .. 8: about 56 kept positions */) {
    const int lane = threadIdx.x & 63;
    const int64_t W = A.P >> 6;
    const int64_t n_chunks = (W + nw - 1) / nw;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // (the masks and row bases of the NEXT chunk travel while this one is worked on: a trip is a chain of dependent loads)
    unsigned long long mine = 0; uint32_t base = 0;
    if (wave0 < n_chunks && lane < nw && wave0 * nw + lane < W) { mine = wave_mask[wave0 * nw + lane]; base = wave_base[wave0 * nw + lane]; }
    for (int64_t ch = wave0; ch < n_chunks; ch += n_waves) {
        const int64_t w0 = ch * nw;
        const unsigned long long cur = mine; const uint32_t cur_base = base;
        { const int64_t nx = (ch + n_waves) * nw + lane; mine = 0; base = 0; if (ch + n_waves < n_chunks && lane < nw && nx < W) { mine = wave_mask[nx]; base = wave_base[nx]; } }
        unsigned long long m[8]; int pre[9]; uint32_t bs[8]; pre[0] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bs[j] = (uint32_t)__builtin_amdgcn_readlane((int)cur_base, j);
            m[j] = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cur, j)
                 | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cur >> 32), j) << 32);
            pre[j + 1] = pre[j] + __popcll(m[j]);
        }
        for (int t = lane; t < pre[8]; t += 64) {
            int j = 0;
#pragma unroll
            for (int q = 1; q < 8; ++q) j += t >= pre[q];
            unsigned long long mj = m[0]; int pj = pre[0]; uint32_t bj = bs[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) if (j == q) { mj = m[q]; pj = pre[q]; bj = bs[q]; }
            const int r = t - pj;
            const int64_t p = ((w0 + j) << 6) + nth_set_bit(mj, r);
            // the kept position's k-mer: one 16-byte load of bases (the window spans at most three words and a bit), one
            // 8-byte load of the mask; the position was kept by this pass's mask, so the shard tests of canon_key pass
            uint4 wv; __builtin_memcpy(&wv, A.planes + 2 * (p >> 5), 16);
            const uint32_t sh = (uint32_t)(p & 31);
            const uint64_t key = canon_key(A, __builtin_amdgcn_alignbit(wv.z, wv.x, sh), __builtin_amdgcn_alignbit(wv.w, wv.y, sh));
            const uint32_t c = bj + (uint32_t)r;
            keys[c] = key; if (pos) pos[c] = c;
        }
    }
}

// HASH sub-shards of one call (sets beyond 2^32 bases: 10^6 contigs run as seven): every pass used to scan ALL bases with
// k_kmer_count to keep 1/sub of the k-mers -- sub x the k-mer arithmetic of the set.  k_multi_mask scans once and leaves the
// kept mask of EVERY sub-shard (one bit per base and sub-shard: a valid k-mer belongs to exactly one); a pass then lists
// its kept positions from its mask and computes only their k-mers (k_kmer_emit_sparse).  Masks of sub-shard t at
// masks + t * mask_stride.  A wave's 256 positions: one 32-position word per sub-shard and eighth, in the wave's own LDS.
constexpr int MM_MAX_SUB = 32;
template <int KC>
__global__ void __launch_bounds__(256)
k_multi_mask(kmer_args A, uint32_t first_shard, uint32_t n_sub, uint32_t n_total, unsigned long long* __restrict__ masks, int64_t mask_stride) {
    if (KC > 0) { A.k = KC; A.use_frac = 0; }
    A.n_shards = 1; A.shard = 0; A.dig_lo = 0; A.dig_n = 1u << DIG_BITS;      // every valid k-mer is computed; its sub-shard is found below
    extern __shared__ uint32_t s_mm[];                                        // [4 waves][n_sub][8]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* sm = s_mm + (size_t)wv * n_sub * 8;
    for (int i = lane; i < (int)n_sub * 8; i += 64) sm[i] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; p0 < A.P + 252; p0 += (int64_t)gridDim.x * blockDim.x * 4) {
        uint64_t kk[4] = {SENT, SENT, SENT, SENT}; uint32_t g = 0;
        if (p0 < A.P) kmers4(A, p0, kk, &g);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (kk[j] != SENT) {
            const uint32_t h2 = (uint32_t)kk[j] * 0x85ebca6bu;                // the HASH shard of canon_key
            const uint32_t t = (uint32_t)(((uint64_t)h2 * n_total) >> 32) - first_shard;
            if (t < n_sub) atomicOr(&sm[t * 8u + (uint32_t)(lane >> 3)], 1u << (4 * (lane & 7) + j));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t wbase = (p0 - 4 * lane) >> 6;                           // first of the wave's four 64-position words
        for (int i = lane; i < (int)n_sub * 8; i += 64) {
            const uint32_t v = sm[i]; sm[i] = 0u;
            const int t = i >> 3, wd = i & 7;
            if (((wbase + (wd >> 1)) << 6) < A.P) reinterpret_cast<uint32_t*>(masks + (int64_t)t * mask_stride)[wbase * 2 + wd] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
__global__ void k_kept_from_goff(const uint32_t* __restrict__ goff, int n, int* __restrict__ kept) {
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) kept[g] = (int)(goff[g + 1] - goff[g]);
}

// compact space: genome of compact index c.  cblk[c >> CBLK_SHIFT] is the genome holding the first
// index of that block; a genome's range is [wave_base[base_off[g]/64], wave_base[base_off[g+1]/64]).
constexpr int CBLK_SHIFT = 10;
struct compact_map { const uint32_t* goff; const uint32_t* cblk; };     // goff[g] = first compact index of genome g (n + 1 entries)
__device__ __forceinline__ uint32_t genome_of_compact(const compact_map& M, uint32_t c) {
    uint32_t g = M.cblk[c >> CBLK_SHIFT];
    while (c >= M.goff[g + 1]) ++g;
    return g;
}
__global__ void k_cblk(const uint32_t* __restrict__ wave_base, const int64_t* __restrict__ base_off, int n, uint32_t* __restrict__ cblk,
                       uint32_t* __restrict__ goff) {
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
        const uint32_t c0 = wave_base[base_off[g] >> 6], c1 = wave_base[base_off[g + 1] >> 6];
        goff[g] = c0; if (g == n - 1) goff[n] = c1;
        for (uint64_t b = ((uint64_t)c0 + (1u << CBLK_SHIFT) - 1) >> CBLK_SHIFT; (b << CBLK_SHIFT) < c1; ++b) cblk[b] = (uint32_t)g;
    }
}

__global__ void k_iota(uint32_t* v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}

// ------------------------------------------------------------------ partial sort
// The device radix sort only orders the top `sort_bits` key bits (3-4 passes instead of 7).
// Entries with equal prefix form small contiguous groups (the scramble makes prefixes uniform:
// about n / 2^sort_bits entries each, plus the repeats of a k-mer); k_group_runs below finishes
// the order inside the groups on the fly.
constexpr int GS_TILE = 1024;       // list entries owned by a workgroup per trip
constexpr int GS_HALO = 256;        // staged beyond the tile so that groups starting inside it are complete

// ------------------------------------------------------------------ K2a: runs of the inverted index
// One pass over the sorted (k-mer, position) list: every entry finds the start of its run of equal k-mers
// by galloping over its neighbours (runs are short; the neighbours are in cache), flags duplicates (same
// k-mer, same genome), records its genome (the CSC side of the SpGEMM) and, for k-mers shared by >= 2
// entries, scatters one row pointer (1 + run start) to its base position (the CSR side).  Singletons write
// nothing.
__device__ __forceinline__ int64_t run_lower(const uint64_t* __restrict__ keys, int64_t i, uint64_t key) {
    int64_t lo = i;                                   // invariant: keys[lo] == key
    int64_t step = 1;
    while (lo - step >= 0 && keys[lo - step] == key) { lo -= step; step <<= 1; }
    // first index in (lo - step, lo] holding key
    int64_t a = lo - step < -1 ? -1 : lo - step;      // keys[a] != key (or a == -1)
    while (lo - a > 1) { int64_t m = (a + lo) >> 1; if (keys[m] == key) lo = m; else a = m; }
    return lo;
}

__global__ void __launch_bounds__(256)
k_runs(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ blk2g, int blk_shift,
       int64_t n, uint32_t* __restrict__ gen, uint32_t* __restrict__ rowinfo,
       compact_map M, int* __restrict__ dup_per_genome) {
    // four consecutive entries per thread: their keys, positions and genomes are fetched with
    // independent loads first (the kernel is bound by load latency, not by bandwidth)
    const int64_t n4 = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = q << 2;
        uint64_t kk[6]; uint32_t pp[5]; uint32_t gg[5];
#pragma unroll
        for (int j = 0; j < 6; ++j) { const int64_t i = i0 - 1 + j; kk[j] = (i >= 0 && i < n) ? keys[i] : SENT; }
#pragma unroll
        for (int j = 0; j < 5; ++j) { const int64_t i = i0 - 1 + j; pp[j] = (i >= 0 && i < n) ? pos[i] : 0u; }
#pragma unroll
        for (int j = 0; j < 5; ++j) gg[j] = M.cblk ? genome_of_compact(M, pp[j]) : blk2g[pp[j] >> blk_shift];
#pragma unroll
        for (int j = 1; j < 5; ++j) {
            const int64_t i = i0 - 1 + j;
            if (i >= n) continue;
            const uint64_t key = kk[j]; const uint32_t p = pp[j]; const uint32_t g = gg[j];
            const bool has_prev = kk[j - 1] == key;             // SENT never equals a real key
            const bool has_next = kk[j + 1] == key;
            const bool dup = has_prev && gg[j - 1] == g;
            gen[i] = g | (dup ? DUP_BIT : 0u);
            if (dup) { atomicAdd(&dup_per_genome[g], 1); continue; }
            if (!has_prev && !has_next) continue;               // singleton k-mer: no partner possible
            if (!has_prev) continue;                            // the run's smallest genome: no partner b < a
            rowinfo[p] = (uint32_t)run_lower(keys, i, key) + 1u;        // p = base position (dense) or compact index
        }
    }
}

// ------------------------------------------------------------------ K1b + K2a fused
// The common case: every equal-prefix group fits a staged tile.  Each entry ranks itself inside
// its group by the full key (stable) and, from the same two scans, knows the run of its own k-mer:
// where it starts in the fully sorted list, how long it is, and whether the entry in front of it
// in that order belongs to the same genome (duplicate).  It writes its genome to its final place
// and scatters the row pointer; the sorted keys themselves are never written.  A group that
// does not fit the staged window (a k-mer that occurs hundreds of times: low-complexity sequence,
// conserved genes) is queued for k_long_groups.
__global__ void __launch_bounds__(256)
k_group_runs(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ blk2g, int blk_shift,
             int64_t n, int low_bit, uint32_t* __restrict__ gen, uint32_t* __restrict__ rowinfo,
             compact_map M, int* __restrict__ dup_per_genome, int64_t* __restrict__ long_list, unsigned int* __restrict__ n_long,
             unsigned int long_cap) {
    __shared__ uint64_t sk[GS_TILE + GS_HALO + 1];
    __shared__ uint32_t sp[GS_TILE + GS_HALO + 1];
    __shared__ unsigned long long sh[(GS_TILE + GS_HALO + 1) / 64 + 2];      // bit j: entry j starts an equal-prefix group
    constexpr int SLOTS = (GS_TILE + GS_HALO + 1 + 255) / 256 * 256;
    const int64_t n_tiles = (n + GS_TILE - 1) / GS_TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t t0 = tile * GS_TILE;
        const int m = (int)min<int64_t>((int64_t)GS_TILE + GS_HALO + 1, n - t0 + 1);
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += blockDim.x) {
            const int64_t g = t0 - 1 + j;
            const uint32_t p = g >= 0 ? pos[g] : 0u;
            sk[j] = g >= 0 ? keys[g] : ~0ULL;
            sp[j] = p;
        }
        __syncthreads();
        // group heads as a bit map (slot 0 and everything from slot m on count as heads), so that an entry
        // finds the bounds of its group with two bit scans instead of walking to them
        for (int j = threadIdx.x; j < SLOTS; j += blockDim.x) {
            const bool head = j == 0 || j >= m || (sk[j] >> low_bit) != (sk[j - 1] >> low_bit);
            const unsigned long long hb = __ballot(head);
            if ((threadIdx.x & 63) == 0 && (j >> 6) < (int)(sizeof(sh) / sizeof(sh[0]))) sh[j >> 6] = hb;
        }
        if (threadIdx.x == 0) sh[sizeof(sh) / sizeof(sh[0]) - 1] = ~0ULL;
        __syncthreads();
        const int own = (int)min<int64_t>(GS_TILE, n - t0);
        for (int j = 1 + threadIdx.x; j < m; j += blockDim.x) {
            int w = j >> 6;
            unsigned long long x = sh[w] & (~0ULL >> (63 - (j & 63)));
            while (x == 0) x = sh[--w];
            const int gs = (w << 6) + 63 - __builtin_clzll(x);
            if (gs < 1 || gs > own) continue;                            // the group starts in another tile
            w = j >> 6;
            x = (j & 63) == 63 ? 0ULL : sh[w] & (~0ULL << ((j & 63) + 1));
            while (x == 0) x = sh[++w];
            const int ge = min(m, (w << 6) + __builtin_ctzll(x));
            if (ge == m && t0 - 1 + m < n) {                              // runs past the halo: one entry queues the group
                if (j == gs) { const unsigned int o = atomicAdd(n_long, 1u); if (o < long_cap) long_list[o] = t0 - 1 + gs; }
                continue;
            }
            // rank inside the group by the full key (stable), and the run of the entry's own k-mer
            const uint64_t key = sk[j];
            int lt = 0, eq_before = 0, eq_after = 0, prev_eq = -1;
            for (int t = gs; t < ge; ++t) {
                const uint64_t kt = sk[t];
                lt += kt < key;
                if (kt == key) { if (t < j) { ++eq_before; prev_eq = t; } else if (t > j) ++eq_after; }
            }
            const uint32_t rl = (uint32_t)(eq_before + eq_after + 1);
            if (rl < 2) continue;                                        // singleton k-mer: no partner, and nobody reads its gen[] slot
            const int64_t rs = t0 - 1 + gs + lt;                         // first entry of this k-mer's run, sorted order
            // genomes are looked up here, not staged: 12 bytes of LDS per entry keep five workgroups on a CU
            const uint32_t pj = sp[j];
            const uint32_t g = M.cblk ? genome_of_compact(M, pj) : blk2g[pj >> blk_shift];
            bool dup = false;
            if (prev_eq >= 0) { const uint32_t pq = sp[prev_eq]; dup = (M.cblk ? genome_of_compact(M, pq) : blk2g[pq >> blk_shift]) == g; }
            gen[rs + eq_before] = g | (dup ? DUP_BIT : 0u);
            if (dup) { atomicAdd(&dup_per_genome[g], 1); continue; }
            if (eq_before == 0) continue;                                // the run's smallest genome: no partner b < a
            const uint32_t p = sp[j];
            rowinfo[p] = (uint32_t)rs + 1u;                             // p = base position (dense) or compact index
        }
    }
}

// A queued group: one workgroup walks it.  If it is a single k-mer (the usual case: one long run),
// start and length of the run are the group's, the order is already final, and every entry gets
// its genome / duplicate flag / row pointer in parallel.  Several k-mers sharing the prefix of
// a long group need a real sort: *need_full_sort sends the call to the general path.
__global__ void __launch_bounds__(256)
k_long_groups(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ blk2g, int blk_shift,
              int64_t n, int low_bit, const int64_t* __restrict__ long_list, uint32_t* __restrict__ gen, uint32_t* __restrict__ rowinfo,
              compact_map M, int* __restrict__ dup_per_genome, unsigned int* __restrict__ need_full_sort) {
    __shared__ int64_t s_end;
    const int64_t gs = long_list[blockIdx.x];
    const uint64_t key0 = keys[gs]; const uint64_t pre = key0 >> low_bit;
    if (threadIdx.x == 0) s_end = n;
    __syncthreads();
    bool same = true;
    for (int64_t base = gs; base < n; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        bool inside = false;
        if (i < n) {
            const uint64_t kx = keys[i];
            inside = (kx >> low_bit) == pre;
            if (inside) same = same && kx == key0; else atomicMin((unsigned long long*)&s_end, (unsigned long long)i);
        }
        if (!__syncthreads_and(inside)) break;                     // the group ends inside this chunk (or at n)
    }
    if (!__syncthreads_and(same)) { if (threadIdx.x == 0) atomicOr(need_full_sort, 1u); return; }
    const int64_t ge = s_end;
    for (int64_t i = gs + threadIdx.x; i < ge; i += blockDim.x) {
        const uint32_t p = pos[i];
        const uint32_t g = M.cblk ? genome_of_compact(M, p) : blk2g[p >> blk_shift];
        bool dup = false;
        if (i > gs) { const uint32_t pp = pos[i - 1]; dup = (M.cblk ? genome_of_compact(M, pp) : blk2g[pp >> blk_shift]) == g; }
        gen[i] = g | (dup ? DUP_BIT : 0u);
        if (dup || i == gs) { if (dup) atomicAdd(&dup_per_genome[g], 1); continue; }   // (the run's smallest genome has no partner b < a)
        rowinfo[p] = (uint32_t)gs + 1u;
    }
}

// ------------------------------------------------------------------ K2b: row-wise SpGEMM
// One workgroup per genome a: for every distinct k-mer of a, walk the (ascending) genome list
// of that k-mer and count partners b < a in an LDS hash table; emit (a, b, shared).
// LDS hash slots per workgroup: 2^11 (16 KiB: five workgroups per CU hide the latency of the list
// gathers) for the first try, 2^13 (64 KiB) for the rows whose partners did not fit, dense after that
constexpr uint32_t HT_EMPTY = 0xffffffffu;
constexpr int LONG_RUN = 48;           // runs longer than this are walked by the whole workgroup
constexpr int LQ_CAP = 512;


template <int HT_BITS>
__device__ __forceinline__ bool ht_add(uint32_t* hk, uint32_t* hc, uint32_t b, uint32_t* n_used) {
    constexpr int HT_SIZE = 1 << HT_BITS;
    uint32_t h = (b * 2654435761u) >> (32 - HT_BITS);
    for (int probe = 0; probe < HT_SIZE; ++probe) {
        uint32_t cur = hk[h];
        if (cur == b) { atomicAdd(&hc[h], 1u); return true; }
        if (cur == HT_EMPTY) {
            uint32_t old = atomicCAS(&hk[h], HT_EMPTY, b);
            if (old == HT_EMPTY) { atomicAdd(n_used, 1u); atomicAdd(&hc[h], 1u); return true; }
            if (old == b) { atomicAdd(&hc[h], 1u); return true; }
        }
        h = (h + 1) & (HT_SIZE - 1);
    }
    return false;
}

// COMPACT (one-wave workgroups, blockDim = 64): the non-zero row pointers of a trip are first packed into an LDS queue and
// the list walks then run over the queue with every lane busy.  Only ~1 row pointer in 5 is non-zero (a k-mer with a
// partner in a smaller genome); walked where they lie, the sixteen walk bodies of a trip execute for a dozen lanes each.
// For short rows -- shards and sub-shards of the k-mer range, sets of 10^6 contigs -- that issue time, not the list
// gathers, is what the kernel costs.
template <int HT_BITS, bool COMPACT = false>
__global__ void __launch_bounds__(256)
k_spgemm(const uint32_t* __restrict__ rowinfo, const uint32_t* __restrict__ gen, uint64_t n_gen,
         const int64_t* __restrict__ base_off, const int64_t* __restrict__ len, const uint32_t* __restrict__ wave_base,
         int n_genomes, uint32_t min_emit, const uint32_t* __restrict__ row_list, int n_rows,
         vg_pair_count* __restrict__ out, unsigned long long* __restrict__ out_cursor,
         unsigned long long out_cap, uint32_t* __restrict__ overflow_rows, uint32_t* __restrict__ n_overflow) {
    constexpr int HT_SIZE = 1 << HT_BITS;
    constexpr int ROWS_PER_TRIP = 16;             // independent row-pointer loads per thread and trip
    __shared__ uint32_t hk[HT_SIZE];
    __shared__ uint32_t hc[HT_SIZE];
    constexpr int LQC = COMPACT ? 64 : LQ_CAP;         // long-run queue (one wave drains it often: a short one keeps the LDS small)
    __shared__ uint32_t lq[LQC];
    __shared__ uint32_t rq[COMPACT ? 64 * ROWS_PER_TRIP : 1];
    __shared__ uint32_t s_used, s_lq, s_fail, s_first;
    // XCD-aware dealing (workgroup b runs on XCD b % 8): consecutive rows -- neighbouring genomes, which share
    // their k-mers' genome lists when they are related -- go to ONE XCD, so the lists are re-read from its L2
    // A workgroup takes the rows k = its slot, slot + slots, ... of its XCD's share.  (Every launch gives a workgroup ONE
    // row today: a few thousand persistent one-wave workgroups were measured and were no faster -- 2.7 against 1.9 ms for a
    // shard of eight at 100 k genomes, the same at 10^6 contigs -- so the dispatch of 10^6 workgroups is not what costs.)
    const int per_xcd = (n_rows + 7) / 8;                      // the grid is a multiple of 8 workgroups
    for (int k = (int)(blockIdx.x / 8); k < per_xcd; k += (int)(gridDim.x / 8)) {
    const int row = (int)(blockIdx.x % 8) * per_xcd + k;
    if (row >= n_rows) break;
    const uint32_t a = row_list ? row_list[row] : (uint32_t)row;
    __syncthreads();                                           // (the previous row's table has been read out)
    for (int i = threadIdx.x; i < HT_SIZE; i += blockDim.x) { hk[i] = HT_EMPTY; hc[i] = 0; }
    if (threadIdx.x == 0) { s_used = 0; s_lq = 0; s_fail = 0; }
    __syncthreads();
    // row a = its base positions (dense) or its kept k-mers (compact index space)
    const int64_t p0 = wave_base ? (int64_t)wave_base[base_off[a] >> 6] : base_off[a];
    const int64_t L = wave_base ? (int64_t)wave_base[base_off[a + 1] >> 6] - p0 : len[a];
    // one walk: the genome list of the run, four entries per memory round trip; ascending, and genome a itself is
    // in it: the first genome >= a ends the walk (entries past the run are never reached)
    auto walk = [&](uint32_t rs) {
        bool done = false;
        uint32_t e = 0;
        for (; e < (uint32_t)LONG_RUN && !done; e += 4) {
            uint32_t g4[4];
            __builtin_memcpy(g4, gen + (size_t)rs + e, 16);            // one 16-byte load (the list carries 4 slack entries)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (done) continue;
                const uint32_t g = g4[j];
                if (g & DUP_BIT) continue;
                if (g >= a) { done = true; continue; }
                if (!ht_add<HT_BITS>(hk, hc, g, &s_used)) s_fail = 1;
            }
        }
        if (!done) {
            // a long run: the rest is walked by the whole workgroup (queue full: this thread goes on alone)
            const uint32_t slot = atomicAdd(&s_lq, 1u);
            if (slot < LQC) lq[slot] = rs + e;
            else for (uint64_t x = (uint64_t)rs + e; x < n_gen; ++x) {
                const uint32_t g = gen[x];
                if (g & DUP_BIT) continue;
                if (g >= a) break;
                if (!ht_add<HT_BITS>(hk, hc, g, &s_used)) s_fail = 1;
            }
        }
    };
    for (int64_t base = 0; base < L; base += (int64_t)blockDim.x * ROWS_PER_TRIP) {
        uint32_t rr4[ROWS_PER_TRIP];
#pragma unroll
        for (int u = 0; u < ROWS_PER_TRIP; ++u) {
            const int64_t i = base + (int64_t)u * blockDim.x + threadIdx.x;
            rr4[u] = (i < L) ? rowinfo[p0 + i] : 0u;
        }
        if (COMPACT) {
            const int lane = threadIdx.x & 63;
            uint32_t nq = 0;
#pragma unroll
            for (int u = 0; u < ROWS_PER_TRIP; ++u) {
                const bool nz = rr4[u] != 0u;
                const unsigned long long b = __ballot(nz);
                if (nz) rq[nq + (uint32_t)__popcll(b & ((1ULL << lane) - 1ULL))] = rr4[u];
                nq += (uint32_t)__popcll(b);
            }
            __syncthreads();
            for (uint32_t qi = (uint32_t)lane; qi < nq; qi += 64u) walk(rq[qi] - 1u);
        } else {
#pragma unroll
        for (int u = 0; u < ROWS_PER_TRIP; ++u) {
            if (!rr4[u]) continue;
            walk(rr4[u] - 1u);
        }
        }
        // drain the long-run queue cooperatively when it fills up
        __syncthreads();
        if (s_lq >= LQC / 2 || base + (int64_t)blockDim.x * ROWS_PER_TRIP >= L) {
            const uint32_t nq = s_lq < LQC ? s_lq : LQC;
            for (uint32_t qi = 0; qi < nq; ++qi) {
                for (uint64_t c0 = lq[qi]; ; c0 += blockDim.x) {
                    // one chunk of the list: everything in front of the first genome >= a counts
                    if (threadIdx.x == 0) s_first = 0xffffffffu;
                    __syncthreads();
                    const uint64_t x = c0 + threadIdx.x;
                    const uint32_t g = x < n_gen ? gen[x] : 0u;
                    const bool stop = x >= n_gen || (!(g & DUP_BIT) && g >= a);
                    if (stop) atomicMin(&s_first, (uint32_t)threadIdx.x);
                    __syncthreads();
                    const uint32_t f = s_first;
                    if (threadIdx.x < f && !(g & DUP_BIT)) { if (!ht_add<HT_BITS>(hk, hc, g, &s_used)) s_fail = 1; }
                    __syncthreads();
                    if (f != 0xffffffffu) break;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) s_lq = 0;
            __syncthreads();
        }
    }
    __syncthreads();
    if (s_fail || s_used > HT_SIZE * 7 / 8) {
        // too many partners for the LDS table: hand the row to the dense fallback
        if (threadIdx.x == 0) { uint32_t o = atomicAdd(n_overflow, 1u); overflow_rows[o] = a; }
        continue;
    }
    // output: the workgroup takes ONE range of the global cursor for all its pairs (threads place themselves inside it
    // through an LDS counter).  One global atomic per PAIR on the one cursor word was the floor of this kernel: ~10 ns
    // each, 4.5 ms for the 450 000 pairs of 100 k genomes whatever the shard held, most of the time at 10^6 contigs.
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < HT_SIZE; i += blockDim.x) mine += (hk[i] != HT_EMPTY && hc[i] >= min_emit) ? 1u : 0u;
    if (threadIdx.x == 0) s_lq = 0;
    __syncthreads();
    const uint32_t at = mine ? atomicAdd(&s_lq, mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t tot = s_lq; s_first = 0; if (tot) { const unsigned long long o = atomicAdd(out_cursor, (unsigned long long)tot); lq[0] = (uint32_t)o; lq[1] = (uint32_t)(o >> 32); } }
    __syncthreads();
    if (mine) {
        unsigned long long o = ((unsigned long long)lq[1] << 32 | lq[0]) + at;
        for (int i = threadIdx.x; i < HT_SIZE; i += blockDim.x) {
            const uint32_t b = hk[i];
            if (b != HT_EMPTY && hc[i] >= min_emit) { if (o < out_cap) { out[o].a = a; out[o].b = b; out[o].shared = hc[i]; } ++o; }
        }
    }
    }
}

// dense fallback for rows whose partner set does not fit the LDS table: one workgroup per
// overflowing row, counters in a private global array of n_genomes entries
__global__ void __launch_bounds__(256)
k_spgemm_dense(const uint32_t* __restrict__ rowinfo, const uint32_t* __restrict__ gen, uint64_t n_gen,
               const int64_t* __restrict__ base_off, const int64_t* __restrict__ len, const uint32_t* __restrict__ wave_base,
               int n_genomes, uint32_t min_emit, const uint32_t* __restrict__ rows, uint32_t* __restrict__ dense /* gridDim.x * n_genomes, zeroed */,
               vg_pair_count* __restrict__ out, unsigned long long* __restrict__ out_cursor, unsigned long long out_cap) {
    const uint32_t a = rows[blockIdx.x];
    uint32_t* cnt = dense + (size_t)blockIdx.x * n_genomes;
    const int64_t p0 = wave_base ? (int64_t)wave_base[base_off[a] >> 6] : base_off[a];
    const int64_t L = wave_base ? (int64_t)wave_base[base_off[a + 1] >> 6] - p0 : len[a];
    for (int64_t i = threadIdx.x; i < L; i += blockDim.x) {
        const uint32_t r = rowinfo[p0 + i];
        if (!r) continue;
        for (uint64_t x = (uint64_t)r - 1; x < n_gen; ++x) {
            const uint32_t g = gen[x];
            if (g & DUP_BIT) continue;
            if (g >= a) break;
            atomicAdd(&cnt[g], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < a; b += blockDim.x) {
        uint32_t c = cnt[b];
        if (c >= min_emit && c > 0) {
            unsigned long long o = atomicAdd(out_cursor, 1ULL);
            if (o < out_cap) { out[o].a = a; out[o].b = b; out[o].shared = c; }
        }
    }
}

// ------------------------------------------------------------------ K1b + K2a, bucket pipeline
// The inverted index without a general radix sort (this replaces rocprim::radix_sort_pairs + k_group_runs on
// the hot path).  Keys are scrambled, so their top bits are uniform: one or two MSD partition levels
// (2^11 ways, then up to 2^11) cut the k-mers into buckets of ~1 000 that are sorted inside the LDS, where the
// runs of equal k-mers are found and (genome list, row pointers) written exactly as k_group_runs does.
//   level 1   k_part_count<SRC>      per super-tile histogram of the top B1 key bits -> one row of T1[st][b]
//             k_scan_columns_a/b/c   T1 -> write offsets in place (column-major record order), bucket starts
//             k_part_scatter_dense   dense source: tiles of 32 768 positions, the k-mers computed again from
//                                    the packed bases held in the LDS, sorted there as a PERMUTATION only and
//                                    computed a third time at output; one contiguous run per (tile, bucket);
//                                    8-byte records when 2k - 11 + 25 <= 64 (lvl2_tab), else 12-byte (w0, w1, pay)
//             k_part_scatter<SRC,1>  other sources (kept k-mers of a fraction / shard; small inputs): tiles of
//                                    8 192 records staged in the LDS
//   level 2   units = the records of ONE level-1 bucket from u_st super-tiles (~65 536), bounds read off T1
//             k_part_count2          histogram of the next B2 bits -> one row of T2[b1][unit][b2]
//             k_scan_units           per level-1 bucket: T2 -> write offsets in place + the final bucket starts
//             k_part_scatter2_narrow tiles of 16 384 records narrowed to 8 bytes (<= 32 key bits left), bin-major
//                                    output; k_part_scatter<SRC_PLANES,2> for wider keys
//   buckets   k_bucket_runs          LDS counting sort on the next 9 (11) bits, in-bin ranking, runs, duplicates,
//                                    gen[] and row pointers; 1 536-entry buckets, 6 144 on a second attempt
// No global atomics, no host round trip between the levels, deterministic output.  A bucket that does not fit the
// LDS (a k-mer present many hundreds of times) or an input too small sends the call to the general path.
constexpr int PT_THREADS = 1024;
constexpr int PT_TILE = 8192;            // elements staged per trip: 96 KiB of LDS + bins
constexpr int PT_PER = PT_TILE / PT_THREADS;
constexpr int PT_MAXBINS = 4096;         // 2^11 ways at level 1; two level-1 buckets x 2^11 ways at level 2
constexpr int BK_CAP = 1536;             // largest bucket the LDS sort takes (22 KiB of LDS: seven workgroups per CU)
constexpr int BK_THREADS = 256;

struct part_src {                        // where the elements of a partition level come from
    kmer_args A;                         // level 1, dense: padded base positions (k-mers computed on the fly)
    const uint64_t* keys; const uint32_t* pos;      // level 1, compact: kept k-mers and their row numbers (pos == nullptr: the row number of an element is its index)
    const uint32_t* rec;                 // level 2: the level-1 records (w0, w1, pay), 12 bytes each
    int64_t n;                           // number of source slots (positions or elements)
    int k2;                              // key bits = 2k
    // RANGE shard of the dense source: level-1 bins are counted from the shard's first bucket, and the payload of a
    // record is the k-mer's ROW NUMBER = its rank among the kept k-mers in position order (wbase[p / 64] + the kept
    // positions below p in its 64-position word of wmask), so the row pointers of a pass are as few as its k-mers
    int bin_lo;
    const unsigned long long* wmask; const uint32_t* wbase;
};
// row number of padded position p (a kept one) in a RANGE pass
__device__ __forceinline__ uint32_t row_of(const unsigned long long* __restrict__ wmask, const uint32_t* __restrict__ wbase, int64_t p) {
    return wbase[p >> 6] + (uint32_t)__popcll(wmask[p >> 6] & ((1ULL << (p & 63)) - 1ULL));
}
enum { SRC_DENSE = 0, SRC_ARRAYS = 1, SRC_PLANES = 2 };

__device__ __forceinline__ void key_words(uint64_t key, int k2, uint32_t* w0, uint32_t* w1) {
    // (w0 : w1) = the key, top aligned in 64 bits
    const uint64_t t = key << (64 - k2);
    *w0 = (uint32_t)(t >> 32); *w1 = (uint32_t)t;
}

// workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the wave's global
// stores (s_waitcnt vmcnt(0)), which serialises "store the tile" with "prepare the next one"; the kernels below
// order nothing but LDS contents between their phases, so their stores stay in flight across the barrier.
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive prefix sum over the 1 024 threads of a workgroup (scratch: 16 words of LDS)
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_wave[wv] = x;
    lds_sync();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const uint32_t s = s_wave[i]; if (i < wv) base += s; tot += s; }
    lds_sync();
    if (total) *total = tot;
    return base + x - v;
}

// the PT_PER elements of thread t in a tile starting at source slot t0: slot = t0 + j * 1024 + t (planes,
// arrays) or the four consecutive positions 4 * (...) (dense: one sequence window serves four k-mers)
template <int SRC>
__device__ __forceinline__ void load_tile(const part_src& S, int64_t t0, int64_t t_end, uint32_t w0[PT_PER], uint32_t w1[PT_PER],
                                          uint32_t pay[PT_PER], bool ok[PT_PER], uint32_t* genome4 /* dense: genome of each group of 4 */) {
    if (SRC == SRC_DENSE) {
#pragma unroll
        for (int q = 0; q < PT_PER / 4; ++q) {
            const int64_t p0 = t0 + ((int64_t)q * PT_THREADS + threadIdx.x) * 4;
            uint64_t kk[4] = {SENT, SENT, SENT, SENT}; uint32_t g = 0;
            if (p0 < t_end) kmers4(S.A, p0, kk, &g);
            genome4[q] = g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ok[4 * q + j] = kk[j] != SENT;
                key_words(kk[j], S.k2, &w0[4 * q + j], &w1[4 * q + j]);
                pay[4 * q + j] = (uint32_t)(p0 + j);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PT_PER; ++j) {
            const int64_t i = t0 + (int64_t)j * PT_THREADS + threadIdx.x;
            ok[j] = i < t_end;
            w0[j] = 0; w1[j] = 0; pay[j] = 0;
            if (ok[j]) {
                if (SRC == SRC_ARRAYS) { key_words(S.keys[i], S.k2, &w0[j], &w1[j]); pay[j] = S.pos ? S.pos[i] : (uint32_t)i; }
                else { const uint32_t* r = S.rec + 3 * i; w0[j] = r[0]; w1[j] = r[1]; pay[j] = r[2]; }
            }
        }
    }
}

// The same tile in two steps, for the scatter kernel: fetch_tile only issues the loads (the raw words of the
// NEXT tile travel while the current one is sorted in the LDS), decode_tile turns them into (w0, w1, pay).
constexpr int PT_RAW = 3 * PT_PER;
template <int SRC>
__device__ __forceinline__ void fetch_tile(const part_src& S, int64_t t0, int64_t t_end, uint32_t raw[PT_RAW]) {
    if (SRC == SRC_DENSE) {
#pragma unroll
        for (int q = 0; q < PT_PER / 4; ++q) {
            const int64_t p0 = t0 + ((int64_t)q * PT_THREADS + threadIdx.x) * 4;
            if (p0 < t_end) {
                __builtin_memcpy(&raw[6 * q], S.A.planes + 2 * (p0 >> 5), 16);
                raw[6 * q + 4] = S.A.nmask[p0 >> 5]; raw[6 * q + 5] = S.A.nmask[(p0 >> 5) + 1];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PT_PER; ++j) {
            const int64_t i = t0 + (int64_t)j * PT_THREADS + threadIdx.x;
            if (i < t_end) {
                if (SRC == SRC_ARRAYS) { __builtin_memcpy(&raw[3 * j], S.keys + i, 8); raw[3 * j + 2] = S.pos ? S.pos[i] : (uint32_t)i; }
                else __builtin_memcpy(&raw[3 * j], S.rec + 3 * i, 12);
            }
        }
    }
}
template <int SRC>
__device__ __forceinline__ void decode_tile(const part_src& S, int64_t t0, int64_t t_end, const uint32_t raw[PT_RAW], uint32_t w0[PT_PER],
                                            uint32_t w1[PT_PER], uint32_t pay[PT_PER], bool ok[PT_PER]) {
    if (SRC == SRC_DENSE) {
#pragma unroll
        for (int q = 0; q < PT_PER / 4; ++q) {
            const int64_t p0 = t0 + ((int64_t)q * PT_THREADS + threadIdx.x) * 4;
            uint64_t kk[4] = {SENT, SENT, SENT, SENT};
            if (p0 < t_end) {
                kmers4_words(S.A, (uint32_t)(p0 & 31), raw[6 * q], raw[6 * q + 1], raw[6 * q + 2], raw[6 * q + 3], raw[6 * q + 4], raw[6 * q + 5], kk);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ok[4 * q + j] = kk[j] != SENT;
                key_words(kk[j], S.k2, &w0[4 * q + j], &w1[4 * q + j]);
                pay[4 * q + j] = (uint32_t)(p0 + j);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PT_PER; ++j) {
            const int64_t i = t0 + (int64_t)j * PT_THREADS + threadIdx.x;
            ok[j] = i < t_end;
            w0[j] = 0; w1[j] = 0; pay[j] = 0;
            if (ok[j]) {
                if (SRC == SRC_ARRAYS) { key_words((uint64_t)raw[3 * j] | ((uint64_t)raw[3 * j + 1] << 32), S.k2, &w0[j], &w1[j]); pay[j] = raw[3 * j + 2]; }
                else { w0[j] = raw[3 * j]; w1[j] = raw[3 * j + 1]; pay[j] = raw[3 * j + 2]; }
            }
        }
    }
}

// level-1 histogram: one super-tile (st_tiles tiles) per trip; T[st * 2^B1 + b] = elements of bucket b.
// The dense source also counts the kept k-mers per genome (set sizes) on the way.  The raw words of the next
// tile are requested before the current one is counted (one workgroup per CU: nothing else hides the latency).
// KC > 0: k = KC, all k-mers kept, one shard -- as compile-time constants (the default k = 25 of a whole set)
template <int SRC, int KC = 0, bool RANGE = false>
__global__ void __launch_bounds__(PT_THREADS)
k_part_count(part_src S, int B1, int nb /* level-1 buckets of this pass */, int st_tiles, int64_t n_st, uint32_t* __restrict__ T, int* __restrict__ kept_per_genome,
             unsigned long long* __restrict__ wave_mask, uint32_t* __restrict__ wave_cnt /* RANGE shards of the dense source: kept mask and count of every 64 positions */) {
    if (KC > 0) { S.A.k = KC; S.k2 = 2 * KC; S.A.use_frac = 0; S.A.n_shards = 1; }
    if (!RANGE) { S.A.dig_lo = 0; S.A.dig_n = 1u << DIG_BITS; S.bin_lo = 0; wave_mask = nullptr; }      // (the whole pass: the shard tests and the masks fold away)
    __shared__ uint32_t hist[PT_MAXBINS];
    const int lane = threadIdx.x & 63;
    for (int64_t st = blockIdx.x; st < n_st; st += gridDim.x) {
        for (int b = threadIdx.x; b < nb; b += PT_THREADS) hist[b] = 0;
        __syncthreads();
        const int64_t s0 = st * st_tiles * PT_TILE, s1 = min(S.n, s0 + (int64_t)st_tiles * PT_TILE);
        uint32_t raw[PT_RAW], gq[PT_PER / 4];
#pragma unroll
        for (int i = 0; i < PT_RAW; ++i) raw[i] = 0;
        auto fetch_genomes = [&](int64_t t0) {
#pragma unroll
            for (int q = 0; q < PT_PER / 4; ++q) {
                const int64_t p0 = t0 + ((int64_t)q * PT_THREADS + threadIdx.x) * 4;
                gq[q] = (SRC == SRC_DENSE && kept_per_genome && p0 < s1) ? S.A.blk2g[p0 >> S.A.blk_shift] : 0u;
            }
        };
        fetch_tile<SRC>(S, s0, s1, raw); fetch_genomes(s0);
        for (int64_t t0 = s0; t0 < s1; t0 += PT_TILE) {
            uint32_t w0[PT_PER], w1[PT_PER], pay[PT_PER], g4[PT_PER / 4]; bool ok[PT_PER];
            decode_tile<SRC>(S, t0, s1, raw, w0, w1, pay, ok);
#pragma unroll
            for (int q = 0; q < PT_PER / 4; ++q) g4[q] = gq[q];
            if (t0 + PT_TILE < s1) { fetch_tile<SRC>(S, t0 + PT_TILE, s1, raw); fetch_genomes(t0 + PT_TILE); }
#pragma unroll
            for (int j = 0; j < PT_PER; ++j) if (ok[j]) atomicAdd(&hist[B1 ? (w0[j] >> (32 - B1)) - (uint32_t)S.bin_lo : 0u], 1u);
            if (SRC == SRC_DENSE && wave_mask) {
                // the wave's 256 consecutive positions of every group q = four 64-position words: a lane's four flags are a
                // nibble, eight lanes OR their nibbles into a 32-bit half-word (three cross-lane steps), two halves make a word
#pragma unroll
                for (int q = 0; q < PT_PER / 4; ++q) {
                    uint32_t v = ((uint32_t)ok[4 * q] | ((uint32_t)ok[4 * q + 1] << 1) | ((uint32_t)ok[4 * q + 2] << 2) | ((uint32_t)ok[4 * q + 3] << 3)) << (4 * (lane & 7));
                    v |= (uint32_t)__shfl_xor((int)v, 1); v |= (uint32_t)__shfl_xor((int)v, 2); v |= (uint32_t)__shfl_xor((int)v, 4);
                    const uint32_t hi = (uint32_t)__shfl_down((int)v, 8);
                    const int64_t wfirst = (t0 + ((int64_t)q * PT_THREADS + (threadIdx.x & ~63u)) * 4) >> 6;
                    if ((lane & 15) == 0 && ((wfirst + (lane >> 4)) << 6) < S.n) {
                        wave_mask[wfirst + (lane >> 4)] = (unsigned long long)v | ((unsigned long long)hi << 32);
                        wave_cnt[wfirst + (lane >> 4)] = (uint32_t)(__popc(v) + __popc(hi));
                    }
                }
            }
            if (SRC == SRC_DENSE && kept_per_genome) {
#pragma unroll
                for (int q = 0; q < PT_PER / 4; ++q) {
                    const uint32_t g = g4[q]; const uint32_t g0 = __shfl(g, 0);
                    if (__all(g == g0)) {
                        // the wave's 256 positions lie in one genome: four ballots count its kept k-mers
                        const int tot = __popcll(__ballot(ok[4 * q])) + __popcll(__ballot(ok[4 * q + 1])) + __popcll(__ballot(ok[4 * q + 2])) + __popcll(__ballot(ok[4 * q + 3]));
                        if (lane == 0 && tot) atomicAdd(&kept_per_genome[g0], tot);
                    } else {
                        const int mine = (int)ok[4 * q] + (int)ok[4 * q + 1] + (int)ok[4 * q + 2] + (int)ok[4 * q + 3];
                        if (mine) atomicAdd(&kept_per_genome[g], mine);
                    }
                }
            }
        }
        __syncthreads();
        for (int b = threadIdx.x; b < nb; b += PT_THREADS) T[st * nb + b] = hist[b];      // [super-tile][bucket]: one contiguous row
        __syncthreads();
    }
}

// ---- RANGE shards of SEVERAL RANKS: the scan of the bases is cut by POSITION, the kept masks travel.
// k_part_count<.., RANGE> computes the k-mer of every base of the set on every rank to keep 1/world of them (8.3 of the
// 27.6 ms of a rank's prefilter at eight ranks over 100 k genomes).  Here rank r scans only the super-tiles
// [st_lo, st_hi) -- 1/world of the bases -- and sorts what it finds by the rank that OWNS the k-mer's level-1 digit:
//   * one kept bit per position and owner: mask block d = the 64-position words of this rank's slice with the bits of
//     the positions whose k-mer belongs to rank d (every valid position has exactly one owner);
//   * the level-1 counts [super-tile of the slice][digit], cut into the owners' digit ranges: table block d.
// An all-to-all hands every rank the blocks of its own digit range from all slices (vg_slice_exchange): concatenated in
// slice order they ARE the wave_mask and the level-1 table k_part_count<.., RANGE> would have produced.  Per rank
// 1/world of the k-mer arithmetic, and P / 8 bytes of masks + the table received whatever the world is.
// Needs the level-1 digit to be the shard digit (two partition levels: B1 = DIG_BITS).
constexpr int SX_MAX_WORLD = 32;
struct slice_plan {
    int world;
    int64_t st_lo, st_hi;                   // this rank's super-tiles
    int64_t word_lo, words;                 // its 64-position words: first, count (block d of the masks starts at d * words)
    unsigned long long* mask;               // send masks
    uint32_t* T;                            // send tables: block d at T + t_off[d], [st - st_lo][dig_lo[d + 1] - dig_lo[d]]
    int64_t t_off[SX_MAX_WORLD];
    uint32_t dig_lo[SX_MAX_WORLD + 1];      // first digit of every rank's range, and 1 << DIG_BITS
};
// owner of level-1 digit b: the largest d with floor(d * 2^11 / world) <= b
__device__ __forceinline__ uint32_t digit_owner(uint32_t b, uint32_t world) { return ((b + 1u) * world - 1u) >> DIG_BITS; }

template <int KC>
__global__ void __launch_bounds__(PT_THREADS)
k_slice_scan(part_src S, slice_plan X, int st_tiles, int* __restrict__ kept_per_genome) {
    if (KC > 0) { S.A.k = KC; S.k2 = 2 * KC; S.A.use_frac = 0; }
    S.A.n_shards = 1; S.A.dig_lo = 0; S.A.dig_n = 1u << DIG_BITS;      // every k-mer of the slice is computed; its owner is found below
    __shared__ uint32_t hist[1 << DIG_BITS];
    // per wave and group q of the tile: one 32-position word per owner and eighth of the wave's 256 positions
    extern __shared__ uint32_t s_m[];                                   // [16 waves][PT_PER / 4][world][8]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t world = (uint32_t)X.world;
    const int64_t W_total = S.n >> 6;
    for (int i = threadIdx.x; i < 16 * (PT_PER / 4) * (int)world * 8; i += PT_THREADS) s_m[i] = 0;
    for (int64_t st = X.st_lo + blockIdx.x; st < X.st_hi; st += gridDim.x) {
        for (int b = threadIdx.x; b < (1 << DIG_BITS); b += PT_THREADS) hist[b] = 0;
        __syncthreads();
        const int64_t s0 = st * st_tiles * PT_TILE, s1 = min(S.n, s0 + (int64_t)st_tiles * PT_TILE);
        uint32_t raw[PT_RAW], gq[PT_PER / 4];
#pragma unroll
        for (int i = 0; i < PT_RAW; ++i) raw[i] = 0;
        auto fetch_genomes = [&](int64_t t0) {
#pragma unroll
            for (int q = 0; q < PT_PER / 4; ++q) {
                const int64_t p0 = t0 + ((int64_t)q * PT_THREADS + threadIdx.x) * 4;
                gq[q] = p0 < s1 ? S.A.blk2g[p0 >> S.A.blk_shift] : 0u;
            }
        };
        fetch_tile<SRC_DENSE>(S, s0, s1, raw); fetch_genomes(s0);
        for (int64_t t0 = s0; t0 < s1; t0 += PT_TILE) {
            uint32_t w0[PT_PER], w1[PT_PER], pay[PT_PER], g4[PT_PER / 4]; bool ok[PT_PER];
            decode_tile<SRC_DENSE>(S, t0, s1, raw, w0, w1, pay, ok);
#pragma unroll
            for (int q = 0; q < PT_PER / 4; ++q) g4[q] = gq[q];
            if (t0 + PT_TILE < s1) { fetch_tile<SRC_DENSE>(S, t0 + PT_TILE, s1, raw); fetch_genomes(t0 + PT_TILE); }
#pragma unroll
            for (int q = 0; q < PT_PER / 4; ++q) {
                uint32_t* sm = s_m + (size_t)((wv * (PT_PER / 4) + q) * (int)world) * 8;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (ok[4 * q + j]) {
                    const uint32_t b = w0[4 * q + j] >> (32 - DIG_BITS);
                    atomicAdd(&hist[b], 1u);
                    atomicOr(&sm[digit_owner(b, world) * 8u + (uint32_t)(lane >> 3)], 1u << (4 * (lane & 7) + j));
                }
            }
            // the wave's own words leave (and are cleared) without a workgroup barrier: LDS operations of one wave complete in order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < PT_PER / 4; ++q) {
                uint32_t* sm = s_m + (size_t)((wv * (PT_PER / 4) + q) * (int)world) * 8;
                const int64_t wfirst = (t0 + ((int64_t)q * PT_THREADS + (threadIdx.x & ~63u)) * 4) >> 6;      // first of the wave's four 64-position words
                for (int i = lane; i < (int)world * 8; i += 64) {
                    const uint32_t v = sm[i]; sm[i] = 0u;
                    const int d = i >> 3, wd = i & 7;
                    if (wfirst + (wd >> 1) < W_total)
                        reinterpret_cast<uint32_t*>(X.mask)[((int64_t)d * X.words + (wfirst - X.word_lo)) * 2 + wd] = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (kept_per_genome) {
#pragma unroll
                for (int q = 0; q < PT_PER / 4; ++q) {
                    const uint32_t g = g4[q]; const uint32_t g0 = __shfl(g, 0);
                    if (__all(g == g0)) {
                        const int tot = __popcll(__ballot(ok[4 * q])) + __popcll(__ballot(ok[4 * q + 1])) + __popcll(__ballot(ok[4 * q + 2])) + __popcll(__ballot(ok[4 * q + 3]));
                        if (lane == 0 && tot) atomicAdd(&kept_per_genome[g0], tot);
                    } else {
                        const int mine = (int)ok[4 * q] + (int)ok[4 * q + 1] + (int)ok[4 * q + 2] + (int)ok[4 * q + 3];
                        if (mine) atomicAdd(&kept_per_genome[g], mine);
                    }
                }
            }
        }
        __syncthreads();
        for (int b = threadIdx.x; b < (1 << DIG_BITS); b += PT_THREADS) {
            const uint32_t d = digit_owner((uint32_t)b, world);
            const uint32_t nbd = X.dig_lo[d + 1] - X.dig_lo[d];
            X.T[X.t_off[d] + (st - X.st_lo) * nbd + ((uint32_t)b - X.dig_lo[d])] = hist[b];
        }
        __syncthreads();
    }
}
// kept k-mers of every 64-position word of a mask (what k_part_count<.., RANGE> writes beside its masks)
__global__ void k_mask_popc(const unsigned long long* __restrict__ m, int64_t n, uint32_t* __restrict__ cnt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) cnt[i] = (uint32_t)__popcll(m[i]);
}

// Level 2 works on UNITS: the records of one level-1 bucket that came from u_st consecutive super-tiles (about
// 65 536 of them).  A unit's range is read straight off the scanned level-1 table, so the host builds no chunk
// lists and level 2 follows level 1 without a round trip.  Table entry of (b1, unit U, b2):
// (b1 * n_u + U) * 2^B2 + b2 -- a unit's counters are one contiguous row; k_scan_units makes them write offsets.
//
// SHORT records (dense source, 2k - B1 + 25 <= 64): level 1 writes 8 bytes instead of 12,
//   rec = (key bits below the level-1 digit) << 25 | (position mod 2^25);
// the missing high position bits are those of the 2^25-position group the record came from, and a unit spans at
// most four groups whose boundaries are again entries of the level-1 table.
constexpr int SR_POS_BITS = 25;
struct lvl2_tab {
    const uint32_t* T1s; int64_t n_st;      // level-1 write offsets [super-tile][bucket] (k_scan_columns)
    const uint32_t* off1; int nb1;          // start of every level-1 bucket (+ the total)
    int u_st, n_u;                          // super-tiles per unit, units per level-1 bucket
    int g_st;                               // SHORT: super-tiles per position group (0: a unit lies inside one group)
    int st_shift;                           // SHORT: log2(positions per super-tile)
    int kr;                                 // SHORT: key bits kept in a record = 2k - B1
    const uint32_t* wbase; int64_t W;       // SHORT, RANGE shard: rows before every 64-position word (payloads are row numbers), words
};
__device__ __forceinline__ void lvl2_unit(const lvl2_tab& L, int64_t u, uint32_t* b1, uint32_t* U, int64_t* r0, int64_t* r1) {
    *b1 = (uint32_t)(u / L.n_u); *U = (uint32_t)(u % L.n_u);
    const int64_t st0 = (int64_t)*U * L.u_st, st1 = st0 + L.u_st;
    *r0 = L.T1s[st0 * L.nb1 + *b1];
    *r1 = st1 < L.n_st ? L.T1s[st1 * L.nb1 + *b1] : L.off1[*b1 + 1];
}
// SHORT: position base of the unit's first group and the record indices at which its 2nd..4th group start
__device__ __forceinline__ void lvl2_groups(const lvl2_tab& L, uint32_t b1, uint32_t U, int64_t r1, uint32_t* base, uint32_t gb[3]) {
    const int64_t st0 = (int64_t)U * L.u_st;
    *base = (uint32_t)(((uint64_t)st0 << L.st_shift) >> SR_POS_BITS << SR_POS_BITS);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int64_t stg = st0 + (int64_t)(j + 1) * L.g_st;
        gb[j] = (L.g_st > 0 && stg < st0 + L.u_st && stg < L.n_st) ? L.T1s[stg * L.nb1 + b1] : (uint32_t)r1;
    }
}

// level-2 histogram of the next B2 key bits, one unit per trip
template <bool SHORT>
__global__ void __launch_bounds__(PT_THREADS)
k_part_count2(const uint32_t* __restrict__ rec, int B1, int B2, int64_t n_units, lvl2_tab L, uint32_t* __restrict__ T) {
    __shared__ uint32_t hist[PT_MAXBINS];
    const int nb2 = 1 << B2;
    for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        for (int b = threadIdx.x; b < nb2; b += PT_THREADS) hist[b] = 0;
        __syncthreads();
        uint32_t b1, U; int64_t s0, s1;
        lvl2_unit(L, u, &b1, &U, &s0, &s1);
        if (SHORT) {
            // 8-byte records, two per 16-byte load (from an even record index), four loads per thread and trip
            const int sh = SR_POS_BITS + L.kr - B2;
            for (int64_t i0 = (s0 & ~(int64_t)1) + 2 * (int64_t)threadIdx.x; i0 < s1; i0 += 8 * PT_THREADS) {
                uint4 r[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t i = i0 + (int64_t)v * 2 * PT_THREADS;
                    r[v] = make_uint4(0u, 0u, 0u, 0u);
                    if (i + 1 < s1 && i >= s0) __builtin_memcpy(&r[v], rec + 2 * i, 16);
                    else { if (i >= s0 && i < s1) __builtin_memcpy(&r[v].x, rec + 2 * i, 8); if (i + 1 >= s0 && i + 1 < s1) __builtin_memcpy(&r[v].z, rec + 2 * (i + 1), 8); }
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t i = i0 + (int64_t)v * 2 * PT_THREADS;
                    if (i >= s0 && i < s1) atomicAdd(&hist[(uint32_t)((((uint64_t)r[v].y << 32) | r[v].x) >> sh) & (uint32_t)(nb2 - 1)], 1u);
                    if (i + 1 >= s0 && i + 1 < s1) atomicAdd(&hist[(uint32_t)((((uint64_t)r[v].w << 32) | r[v].z) >> sh) & (uint32_t)(nb2 - 1)], 1u);
                }
            }
        } else
        // eight independent loads per thread and trip (one workgroup per CU: the loop is latency bound otherwise)
        for (int64_t i0 = s0 + threadIdx.x; i0 < s1; i0 += 8 * PT_THREADS) {
            uint32_t d[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int64_t i = i0 + (int64_t)v * PT_THREADS;
                d[v] = i < s1 ? rec[3 * i] >> (32 - B1 - B2) : 0u;
            }
#pragma unroll
            for (int v = 0; v < 8; ++v) if (i0 + (int64_t)v * PT_THREADS < s1) atomicAdd(&hist[d[v] & (uint32_t)(nb2 - 1)], 1u);
        }
        __syncthreads();
        const uint64_t t0 = ((uint64_t)b1 * L.n_u + U) * nb2;                     // [bucket][unit][b2]: one contiguous row
        for (int d = threadIdx.x; d < nb2; d += PT_THREADS) T[t0 + d] = hist[d];
        __syncthreads();
    }
}

// scatter of one level: tiles are sorted by bin in the LDS and leave as contiguous segments.
// LEVEL 1: bins = level-1 buckets, write offsets Ts[b * n_st + st].  LEVEL 2: one unit of one level-1 bucket, bins = b2.
template <int SRC, int LEVEL>
__global__ void __launch_bounds__(PT_THREADS)
k_part_scatter(part_src S, int B1, int B2, int unit_tiles, int64_t n_units, const uint32_t* __restrict__ Ts, lvl2_tab L,
               uint32_t* __restrict__ o_rec, int narrow_shift, int nbins1 /* LEVEL 1: buckets of this pass */) {
    __shared__ uint32_t s_w0[PT_TILE], s_w1[PT_TILE], s_pay[PT_TILE];
    __shared__ uint32_t thist[PT_MAXBINS], tstart[PT_MAXBINS], cursor[PT_MAXBINS];
    __shared__ uint32_t s_wave[16];
    const int nbins = LEVEL == 1 ? nbins1 : (1 << B2);
    for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        int64_t s0 = u * unit_tiles * PT_TILE, s1 = min(S.n, s0 + (int64_t)unit_tiles * PT_TILE);
        uint32_t b1 = 0, cl = 0;
        if (LEVEL == 2) lvl2_unit(L, u, &b1, &cl, &s0, &s1);
        lds_sync();
        for (int b = threadIdx.x; b < nbins; b += PT_THREADS) {
            const uint32_t off = LEVEL == 1 ? Ts[u * nbins + b] : Ts[((uint64_t)b1 * L.n_u + cl) * nbins + b];
            cursor[b] = off;
        }
        // The raw words of tile t + 1 are requested while tile t is sorted; they (and, the memory pipeline being
        // in order, the stores of tile t - 1) must have landed before tile t's stores are issued -- so the stores
        // of one tile drain while the next is ranked and staged, instead of being waited for at its first load.
        uint32_t raw[PT_RAW];
#pragma unroll
        for (int i = 0; i < PT_RAW; ++i) raw[i] = 0;
        fetch_tile<SRC>(S, s0, s1, raw);
        for (int64_t t0 = s0; t0 < s1; t0 += PT_TILE) {
            for (int b = threadIdx.x; b < nbins; b += PT_THREADS) thist[b] = 0;
            lds_sync();
            uint32_t w0[PT_PER], w1[PT_PER], pay[PT_PER], bin[PT_PER], rk[PT_PER]; bool ok[PT_PER];
            decode_tile<SRC>(S, t0, s1, raw, w0, w1, pay, ok);
            if (SRC == SRC_DENSE && S.wbase) {
                // RANGE shard: the payload is the row number (the four positions of a group lie in one 64-position word)
#pragma unroll
                for (int q = 0; q < PT_PER / 4; ++q) {
                    const int64_t p0 = t0 + ((int64_t)q * PT_THREADS + threadIdx.x) * 4;
                    if (p0 < s1) {
                        const unsigned long long m = S.wmask[p0 >> 6]; const uint32_t rb = S.wbase[p0 >> 6];
#pragma unroll
                        for (int j = 0; j < 4; ++j) pay[4 * q + j] = rb + (uint32_t)__popcll(m & ((1ULL << ((p0 & 63) + j)) - 1ULL));
                    }
                }
            }
            if (t0 + PT_TILE < s1) fetch_tile<SRC>(S, t0 + PT_TILE, s1, raw);
#pragma unroll
            for (int j = 0; j < PT_PER; ++j) {
                bin[j] = 0; rk[j] = 0;
                if (ok[j]) {
                    bin[j] = LEVEL == 1 ? (B1 ? (w0[j] >> (32 - B1)) - (uint32_t)S.bin_lo : 0u) : ((w0[j] >> (32 - B1 - B2)) & (uint32_t)(nbins - 1));
                    rk[j] = atomicAdd(&thist[bin[j]], 1u);
                }
            }
            lds_sync();
            // exclusive scan of the tile histogram (nbins <= 4 096: four bins per thread)
            {
                constexpr int BPT = PT_MAXBINS / PT_THREADS;
                uint32_t c[BPT], tot = 0;
#pragma unroll
                for (int u = 0; u < BPT; ++u) { const int b = BPT * (int)threadIdx.x + u; c[u] = b < nbins ? thist[b] : 0u; tot += c[u]; }
                uint32_t run = block_scan_1024(tot, s_wave, nullptr);
#pragma unroll
                for (int u = 0; u < BPT; ++u) { const int b = BPT * (int)threadIdx.x + u; if (b < nbins) tstart[b] = run; run += c[u]; }
            }
            lds_sync();
#pragma unroll
            for (int j = 0; j < PT_PER; ++j) if (ok[j]) {
                const uint32_t slot = tstart[bin[j]] + rk[j];
                s_w0[slot] = w0[j]; s_w1[slot] = w1[j]; s_pay[slot] = pay[j];
            }
            lds_sync();
            const uint32_t n_tile = tstart[nbins - 1] + thist[nbins - 1];
#pragma unroll
            for (int i = 0; i < PT_RAW; ++i) asm volatile("" : "+v"(raw[i]));      // the prefetched words have arrived
            // one record per thread and trip, stored with ONE 12-byte (8-byte) instruction: consecutive threads
            // hold consecutive slots, so a wave writes each bin's segment of the tile as one contiguous run
            for (uint32_t slot = threadIdx.x; slot < n_tile; slot += PT_THREADS) {
                const uint32_t w = s_w0[slot];
                const uint32_t b = LEVEL == 1 ? (B1 ? (w >> (32 - B1)) - (uint32_t)S.bin_lo : 0u) : ((w >> (32 - B1 - B2)) & (uint32_t)(nbins - 1));
                const uint64_t dst = (uint64_t)cursor[b] + (slot - tstart[b]);
                if (narrow_shift >= 0) {
                    // the bucket fixes the top narrow_shift key bits and at most 32 remain: one word carries them
                    const uint2 v = make_uint2((uint32_t)(((((uint64_t)w << 32) | s_w1[slot]) << narrow_shift) >> 32), s_pay[slot]);
                    __builtin_memcpy(o_rec + 2 * dst, &v, 8);
                } else {
                    const uint32_t v[3] = { w, s_w1[slot], s_pay[slot] };
                    __builtin_memcpy(o_rec + 3 * dst, v, 12);
                }
            }
            lds_sync();
            for (int b = threadIdx.x; b < nbins; b += PT_THREADS) cursor[b] += thist[b];
            lds_sync();
        }
    }
}

// Level-1 scatter of the dense source with tiles of 32 768 positions.  The records are not staged: the LDS holds
// the tile's packed bases (12 KiB) and, after the counting sort, only the PERMUTATION (a 16-bit local position
// per output slot); the output loop computes each record's k-mer again from the bases in the LDS.  Four times
// the tile of k_part_scatter in less LDS, so a bucket's segment of a tile is four times as long (partial-line
// writes are what bounds this pass).
constexpr int RT_TILE = 32768;
constexpr int RT_PER = RT_TILE / PT_THREADS;
__device__ __forceinline__ uint64_t kmer_key_lds(const kmer_args& A, const uint32_t* s_pk, uint32_t lp) {
    // (the position is a kept one: no validity, fraction or shard test -- the count pass made them)
    const uint32_t k = (uint32_t)A.k, nk = (1u << k) - 1u;
    const uint32_t* w = s_pk + 2 * (lp >> 5); const uint32_t sh = lp & 31u;
    const uint32_t fl = __builtin_amdgcn_alignbit(w[2], w[0], sh) & nk, fh = __builtin_amdgcn_alignbit(w[3], w[1], sh) & nk;
    const uint32_t rl = __brev(~fl) >> (32 - k), rh = __brev(~fh) >> (32 - k);
    const bool f = fh < rh || (fh == rh && fl < rl);
    const uint32_t H = f ? fh : rh, L = f ? fl : rl;
    return ((uint64_t)(H ^ ((L * SCRAMBLE32) >> (32 - k))) << k) | L;
}
template <int KC>
__global__ void __launch_bounds__(PT_THREADS)
k_part_scatter_dense(part_src S, int B1, int nbins /* level-1 buckets of this pass: 2^B1, or the RANGE shard's share */, int unit_tiles, int64_t n_units,
                     const uint32_t* __restrict__ Ts, uint32_t* __restrict__ o_rec, int short_kr /* > 0: SHORT 8-byte records keeping this many key bits */) {
    if (KC > 0) { S.A.k = KC; S.k2 = 2 * KC; S.A.use_frac = 0; S.A.n_shards = 1; S.A.dig_lo = 0; S.A.dig_n = 1u << DIG_BITS; S.bin_lo = 0; }
    __shared__ uint32_t s_pk[RT_TILE / 16 + 8], s_mk[RT_TILE / 32 + 4];
    __shared__ uint16_t s_perm[RT_TILE];
    __shared__ uint32_t thist[PT_MAXBINS], tstart[PT_MAXBINS], cursor[PT_MAXBINS];
    __shared__ uint32_t s_wave[16];
    __shared__ unsigned long long s_wm[RT_TILE / 64]; __shared__ uint32_t s_wb[RT_TILE / 64];      // RANGE shard: kept masks and row bases of the tile
    const int64_t n_pk = (S.A.P >> 4) + 16, n_mk = (S.A.P >> 5) + 16;      // words the arrays hold (vg_genomes_finish: 16 of slack)
    for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int64_t s0 = u * unit_tiles * RT_TILE, s1 = min(S.n, s0 + (int64_t)unit_tiles * RT_TILE);
        lds_sync();
        for (int b = threadIdx.x; b < nbins; b += PT_THREADS) cursor[b] = Ts[u * nbins + b];
        // the tile's packed bases and mask: three + two words per thread, requested one tile ahead (the barriers of
        // this kernel wait for LDS traffic only, so the loads travel while the current tile is sorted)
        constexpr int NPK = (RT_TILE / 16 + 4 + PT_THREADS - 1) / PT_THREADS, NMK = (RT_TILE / 32 + 2 + PT_THREADS - 1) / PT_THREADS;
        uint32_t pf_pk[NPK], pf_mk[NMK];
        auto fetch_bases = [&](int64_t t0) {
#pragma unroll
            for (int v = 0; v < NPK; ++v) { const int i = v * PT_THREADS + (int)threadIdx.x; const int64_t w = (t0 >> 4) + i; pf_pk[v] = (i < RT_TILE / 16 + 4 && w < n_pk) ? S.A.planes[w] : 0u; }
#pragma unroll
            for (int v = 0; v < NMK; ++v) { const int i = v * PT_THREADS + (int)threadIdx.x; const int64_t w = (t0 >> 5) + i; pf_mk[v] = (i < RT_TILE / 32 + 2 && w < n_mk) ? S.A.nmask[w] : 0xffffffffu; }
        };
        fetch_bases(s0);
        for (int64_t t0 = s0; t0 < s1; t0 += RT_TILE) {
#pragma unroll
            for (int v = 0; v < NPK; ++v) { const int i = v * PT_THREADS + (int)threadIdx.x; if (i < RT_TILE / 16 + 4) s_pk[i] = pf_pk[v]; }
#pragma unroll
            for (int v = 0; v < NMK; ++v) { const int i = v * PT_THREADS + (int)threadIdx.x; if (i < RT_TILE / 32 + 2) s_mk[i] = pf_mk[v]; }
            for (int b = threadIdx.x; b < nbins; b += PT_THREADS) thist[b] = 0;
            uint32_t grp_row0 = 0;                                // RANGE: row number of the first kept k-mer of the tile's 2^25-position group
            if (S.wbase) {
                const int64_t W = S.n >> 6;
                if (threadIdx.x < RT_TILE / 64) {
                    const int64_t w = (t0 >> 6) + threadIdx.x;
                    s_wm[threadIdx.x] = w < W ? S.wmask[w] : 0ULL; s_wb[threadIdx.x] = S.wbase[w < W ? w : W];
                }
                grp_row0 = S.wbase[(t0 >> SR_POS_BITS) << (SR_POS_BITS - 6)];
            }
            lds_sync();
            if (t0 + RT_TILE < s1) fetch_bases(t0 + RT_TILE);
            uint32_t br[RT_PER];                                  // bin | rank in bin << 12, or all ones
#pragma unroll
            for (int q = 0; q < RT_PER / 4; ++q) {
                const uint32_t lp0 = ((uint32_t)q * PT_THREADS + threadIdx.x) * 4;
                uint64_t kk[4] = {SENT, SENT, SENT, SENT};
                if (t0 + lp0 < s1) {
                    const uint32_t wi = 2u * (lp0 >> 5);
                    kmers4_words(S.A, lp0 & 31u, s_pk[wi], s_pk[wi + 1], s_pk[wi + 2], s_pk[wi + 3], s_mk[lp0 >> 5], s_mk[(lp0 >> 5) + 1], kk);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    br[4 * q + j] = 0xffffffffu;
                    if (kk[j] != SENT) {
                        const uint32_t bin = B1 ? (uint32_t)(kk[j] >> (S.k2 - B1)) - (uint32_t)S.bin_lo : 0u;
                        br[4 * q + j] = bin | (atomicAdd(&thist[bin], 1u) << 12);
                    }
                }
            }
            lds_sync();
            {
                constexpr int BPT = PT_MAXBINS / PT_THREADS;
                uint32_t c[BPT], tot = 0;
#pragma unroll
                for (int v = 0; v < BPT; ++v) { const int b = BPT * (int)threadIdx.x + v; c[v] = b < nbins ? thist[b] : 0u; tot += c[v]; }
                uint32_t run = block_scan_1024(tot, s_wave, nullptr);
#pragma unroll
                for (int v = 0; v < BPT; ++v) { const int b = BPT * (int)threadIdx.x + v; if (b < nbins) tstart[b] = run; run += c[v]; }
            }
            lds_sync();
#pragma unroll
            for (int q = 0; q < RT_PER / 4; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t v = br[4 * q + j];
                    if (v != 0xffffffffu) s_perm[tstart[v & 0xfffu] + (v >> 12)] = (uint16_t)(((uint32_t)q * PT_THREADS + threadIdx.x) * 4 + j);
                }
            }
            lds_sync();
            const uint32_t n_tile = tstart[nbins - 1] + thist[nbins - 1];
            for (uint32_t slot = threadIdx.x; slot < n_tile; slot += PT_THREADS) {
                const uint32_t lp = s_perm[slot];
                const uint64_t key = kmer_key_lds(S.A, s_pk, lp);
                uint32_t v[3];
                key_words(key, S.k2, &v[0], &v[1]);
                v[2] = (uint32_t)(t0 + lp);
                uint32_t low = v[2];                              // SHORT: the payload inside its 2^25-position group
                if (S.wbase) {
                    // RANGE shard: row number instead of position; inside the group it is counted from the group's first row
                    v[2] = s_wb[lp >> 6] + (uint32_t)__popcll(s_wm[lp >> 6] & ((1ULL << (lp & 63u)) - 1ULL));
                    low = v[2] - grp_row0;
                }
                const uint32_t b = B1 ? (v[0] >> (32 - B1)) - (uint32_t)S.bin_lo : 0u;
                const uint64_t dst = (uint64_t)cursor[b] + (slot - tstart[b]);
                if (short_kr > 0) {
                    const uint64_t r = ((key & ((1ULL << short_kr) - 1)) << SR_POS_BITS) | (uint64_t)(low & ((1u << SR_POS_BITS) - 1u));
                    __builtin_memcpy(o_rec + 2 * dst, &r, 8);
                } else __builtin_memcpy(o_rec + 3 * dst, v, 12);
            }
            lds_sync();
            for (int b = threadIdx.x; b < nbins; b += PT_THREADS) cursor[b] += thist[b];
            lds_sync();
        }
    }
}

// Level-1 scatter of a RANGE shard (dense source, 32 768-position tiles): only 1/n_shards of the positions are kept,
// and the count pass has left their mask.  The kept positions of the tile are first LISTED in position order (s_list:
// one cheap loop over the set bits of the thread's half-word), then every step works on list entries, evenly spread
// over the threads: the k-mer arithmetic is done for the kept positions only (k_part_scatter_dense computes all
// 32 768 k-mers of a tile to throw 7/8 of them away at eight shards), and an entry's row number is the row of the
// tile's first kept k-mer + its index in the list.  A tile (or half-tile: a tile whose kept k-mers exceed the list is
// taken as two halves, each of 16 384 positions) is sorted as a permutation of list indices, the record is computed
// again at output, exactly as in k_part_scatter_dense.
constexpr int RG_LIST = 16384;
constexpr int RG_PER = RG_LIST / PT_THREADS;
constexpr int RG_MAXBINS = 2048;
// TT: positions per tile -- 32 768, or 65 536 when a shard keeps so little of a tile (a sixth or less: six ranks or passes
// and more) that two tiles' kept k-mers fit the list: half the per-tile barriers and scans, segments twice as long
template <int KC, int TT>
__global__ void __launch_bounds__(PT_THREADS)
k_part_scatter_range(part_src S, int B1, int nbins /* <= 2048 */, int unit_tiles, int64_t n_units, const uint32_t* __restrict__ Ts,
                     uint32_t* __restrict__ o_rec, int short_kr /* > 0: SHORT 8-byte records keeping this many key bits */) {
    if (KC > 0) { S.A.k = KC; S.k2 = 2 * KC; S.A.use_frac = 0; }
    __shared__ uint32_t s_pk[TT / 16 + 8];
    __shared__ uint16_t s_list[RG_LIST], s_perm[RG_LIST];
    __shared__ uint32_t thist[RG_MAXBINS], tstart[RG_MAXBINS], cursor[RG_MAXBINS];
    __shared__ uint32_t s_wave[16];
    __shared__ unsigned long long s_wm[TT / 64]; __shared__ uint32_t s_wb[TT / 64 + 1];
    const int64_t n_pk = (S.A.P >> 4) + 16;
    const int64_t W = S.n >> 6;
    for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int64_t s0 = u * unit_tiles * TT, s1 = min(S.n, s0 + (int64_t)unit_tiles * TT);
        lds_sync();
        for (int b = threadIdx.x; b < nbins; b += PT_THREADS) cursor[b] = Ts[u * nbins + b];
        constexpr int NPK = (TT / 16 + 4 + PT_THREADS - 1) / PT_THREADS;
        uint32_t pf_pk[NPK];
        auto fetch_bases = [&](int64_t t0) {
#pragma unroll
            for (int v = 0; v < NPK; ++v) { const int i = v * PT_THREADS + (int)threadIdx.x; const int64_t w = (t0 >> 4) + i; pf_pk[v] = (i < TT / 16 + 4 && w < n_pk) ? S.A.planes[w] : 0u; }
        };
        fetch_bases(s0);
        for (int64_t t0 = s0; t0 < s1; t0 += TT) {
#pragma unroll
            for (int v = 0; v < NPK; ++v) { const int i = v * PT_THREADS + (int)threadIdx.x; if (i < TT / 16 + 4) s_pk[i] = pf_pk[v]; }
            for (int i = (int)threadIdx.x; i <= TT / 64; i += PT_THREADS) {
                const int64_t w = (t0 >> 6) + i;
                if (i < TT / 64) s_wm[i] = w < W ? S.wmask[w] : 0ULL;
                s_wb[i] = S.wbase[w < W ? w : W];
            }
            const uint32_t grp_row0 = S.wbase[(t0 >> SR_POS_BITS) << (SR_POS_BITS - 6)];      // first row of the tile's 2^25-position group
            lds_sync();
            if (t0 + TT < s1) fetch_bases(t0 + TT);
            // words per sub-tile: the whole tile if its kept k-mers fit the list, else halves, else quarters (16 384 positions always fit)
            int wn = TT / 64;
            if (s_wb[TT / 64] - s_wb[0] > (uint32_t)RG_LIST) {
                wn = TT / 128;
                if (TT > 32768 && (s_wb[TT / 128] - s_wb[0] > (uint32_t)RG_LIST || s_wb[TT / 64] - s_wb[TT / 128] > (uint32_t)RG_LIST)) wn = TT / 256;
            }
            for (int w0 = 0; w0 < TT / 64; w0 += wn) {
                const uint32_t row0 = s_wb[w0], n_sub = s_wb[w0 + wn] - row0;
                for (int b = threadIdx.x; b < nbins; b += PT_THREADS) thist[b] = 0;
                // list the kept positions: thread t takes the 32-position half-word t of the sub-tile
                for (int hw = (int)threadIdx.x; hw < 2 * wn; hw += PT_THREADS) {
                    const int w = w0 + (hw >> 1);
                    const unsigned long long m64 = s_wm[w];
                    uint32_t m = (hw & 1) ? (uint32_t)(m64 >> 32) : (uint32_t)m64;
                    uint32_t at = s_wb[w] - row0 + ((hw & 1) ? (uint32_t)__popc((uint32_t)m64) : 0u);
                    const uint32_t lp_base = (uint32_t)w * 64u + (uint32_t)(hw & 1) * 32u;
                    while (m) { s_list[at++] = (uint16_t)(lp_base + (uint32_t)__builtin_ctz(m)); m &= m - 1u; }
                }
                lds_sync();
                uint32_t br[RG_PER];                              // bin | rank in bin << 12
#pragma unroll
                for (int q = 0; q < RG_PER; ++q) {
                    const uint32_t i = (uint32_t)q * PT_THREADS + threadIdx.x;
                    br[q] = 0xffffffffu;
                    if (i < n_sub) {
                        const uint64_t key = kmer_key_lds(S.A, s_pk, s_list[i]);
                        const uint32_t bin = B1 ? (uint32_t)(key >> (S.k2 - B1)) - (uint32_t)S.bin_lo : 0u;
                        br[q] = bin | (atomicAdd(&thist[bin], 1u) << 12);
                    }
                }
                lds_sync();
                {
                    constexpr int BPT = RG_MAXBINS / PT_THREADS;
                    uint32_t c[BPT], tot = 0;
#pragma unroll
                    for (int v = 0; v < BPT; ++v) { const int b = BPT * (int)threadIdx.x + v; c[v] = b < nbins ? thist[b] : 0u; tot += c[v]; }
                    uint32_t run = block_scan_1024(tot, s_wave, nullptr);
#pragma unroll
                    for (int v = 0; v < BPT; ++v) { const int b = BPT * (int)threadIdx.x + v; if (b < nbins) tstart[b] = run; run += c[v]; }
                }
                lds_sync();
#pragma unroll
                for (int q = 0; q < RG_PER; ++q) {
                    const uint32_t v = br[q];
                    if (v != 0xffffffffu) s_perm[tstart[v & 0xfffu] + (v >> 12)] = (uint16_t)((uint32_t)q * PT_THREADS + threadIdx.x);
                }
                lds_sync();
                for (uint32_t slot = threadIdx.x; slot < n_sub; slot += PT_THREADS) {
                    const uint32_t i = s_perm[slot];
                    const uint64_t key = kmer_key_lds(S.A, s_pk, s_list[i]);
                    uint32_t v[3];
                    key_words(key, S.k2, &v[0], &v[1]);
                    v[2] = row0 + i;                              // the row number
                    const uint32_t b = B1 ? (v[0] >> (32 - B1)) - (uint32_t)S.bin_lo : 0u;
                    const uint64_t dst = (uint64_t)cursor[b] + (slot - tstart[b]);
                    if (short_kr > 0) {
                        const uint64_t r = ((key & ((1ULL << short_kr) - 1)) << SR_POS_BITS) | (uint64_t)((v[2] - grp_row0) & ((1u << SR_POS_BITS) - 1u));
                        __builtin_memcpy(o_rec + 2 * dst, &r, 8);
                    } else __builtin_memcpy(o_rec + 3 * dst, v, 12);
                }
                lds_sync();
                for (int b = threadIdx.x; b < nbins; b += PT_THREADS) cursor[b] += thist[b];
                lds_sync();
            }
        }
    }
}

// Level-2 scatter for 8-byte output records (the bucket fixes the top narrow_shift key bits, <= 32 remain):
// tiles of 16 384 records, staged in the LDS already narrowed.  The staged record no longer holds its bin, so
// the output runs bin-major: eight lanes take one bin's slots (a bin's segment of the tile: 8 records = 64 bytes
// on average), eight bins per wave and trip.
constexpr int NT_TILE = 16384;
constexpr int NT_PER = NT_TILE / PT_THREADS;
constexpr int NT_MAXBINS = 2048;
template <bool SHORT>
__global__ void __launch_bounds__(PT_THREADS)
k_part_scatter2_narrow(const uint32_t* __restrict__ rec, int B1, int B2, int64_t n_units, const uint32_t* __restrict__ Ts, lvl2_tab L,
                       uint32_t* __restrict__ o_rec, int narrow_shift) {
    __shared__ uint2 s_rec[NT_TILE];
    __shared__ uint32_t thist[NT_MAXBINS], tstart[NT_MAXBINS], cursor[NT_MAXBINS];
    __shared__ uint32_t s_wave[16];
    const int nbins = 1 << B2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        uint32_t b1, U; int64_t s0, s1;
        lvl2_unit(L, u, &b1, &U, &s0, &s1);
        uint32_t gbase = 0, gb[3] = {0, 0, 0};
        uint32_t grow0 = 0, grow1 = 0, grow2 = 0, grow3 = 0;      // RANGE shard: first row of each of the unit's (up to four) position groups
        if (SHORT) {
            lvl2_groups(L, b1, U, s1, &gbase, gb);
            if (L.wbase) {
                const int64_t w0 = (int64_t)(gbase >> 6), gw = 1LL << (SR_POS_BITS - 6);
                grow0 = L.wbase[min(w0, L.W)]; grow1 = L.wbase[min(w0 + gw, L.W)]; grow2 = L.wbase[min(w0 + 2 * gw, L.W)]; grow3 = L.wbase[min(w0 + 3 * gw, L.W)];
            }
        }
        lds_sync();
        for (int b = threadIdx.x; b < nbins; b += PT_THREADS) cursor[b] = Ts[((uint64_t)b1 * L.n_u + U) * nbins + b];
        for (int64_t t0 = s0; t0 < s1; t0 += NT_TILE) {
            for (int b = threadIdx.x; b < nbins; b += PT_THREADS) thist[b] = 0;
            lds_sync();
            uint32_t key[NT_PER], pay[NT_PER], br[NT_PER];      // br = bin | rank in bin << 11, or all ones
            if (SHORT) {
                // two 8-byte records per 16-byte load: element 2 p and 2 p + 1 of the thread are the records
                // t0 + 2 (p * 1024 + t) and the next one
                uint4 rr[NT_PER / 2];
#pragma unroll
                for (int p = 0; p < NT_PER / 2; ++p) {
                    const int64_t i = t0 + 2 * ((int64_t)p * PT_THREADS + threadIdx.x);
                    rr[p] = make_uint4(0u, 0u, 0u, 0u);
                    if (i + 1 < s1) __builtin_memcpy(&rr[p], rec + 2 * i, 16);
                    else if (i < s1) __builtin_memcpy(&rr[p].x, rec + 2 * i, 8);
                }
#pragma unroll
                for (int p = 0; p < NT_PER / 2; ++p) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = 2 * p + h;
                        const int64_t i = t0 + 2 * ((int64_t)p * PT_THREADS + threadIdx.x) + h;
                        br[j] = 0xffffffffu; key[j] = 0; pay[j] = 0;
                        if (i < s1) {
                            const uint64_t r = h ? (((uint64_t)rr[p].w << 32) | rr[p].z) : (((uint64_t)rr[p].y << 32) | rr[p].x);
                            const uint32_t ui = (uint32_t)i;
                            const uint32_t gsel = (uint32_t)(ui >= gb[0]) + (uint32_t)(ui >= gb[1]) + (uint32_t)(ui >= gb[2]);
                            key[j] = (uint32_t)(r >> SR_POS_BITS) << (32 - (L.kr - B2));
                            const uint32_t low = (uint32_t)r & ((1u << SR_POS_BITS) - 1u);
                            pay[j] = L.wbase ? (gsel == 0 ? grow0 : gsel == 1 ? grow1 : gsel == 2 ? grow2 : grow3) + low : gbase + (gsel << SR_POS_BITS) + low;
                            br[j] = (uint32_t)(r >> (SR_POS_BITS + L.kr - B2)) & (uint32_t)(nbins - 1);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NT_PER; ++j) {
                    const int64_t i = t0 + (int64_t)j * PT_THREADS + threadIdx.x;
                    br[j] = 0xffffffffu; key[j] = 0; pay[j] = 0;
                    if (i < s1) {
                        uint32_t r[3]; __builtin_memcpy(r, rec + 3 * i, 12);
                        key[j] = (uint32_t)(((((uint64_t)r[0] << 32) | r[1]) << narrow_shift) >> 32); pay[j] = r[2];
                        br[j] = (r[0] >> (32 - B1 - B2)) & (uint32_t)(nbins - 1);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NT_PER; ++j) if (br[j] != 0xffffffffu) br[j] |= atomicAdd(&thist[br[j]], 1u) << 11;
            lds_sync();
            {
                constexpr int BPT = NT_MAXBINS / PT_THREADS;
                uint32_t cn[BPT], tot = 0;
#pragma unroll
                for (int v = 0; v < BPT; ++v) { const int b = BPT * (int)threadIdx.x + v; cn[v] = b < nbins ? thist[b] : 0u; tot += cn[v]; }
                uint32_t run = block_scan_1024(tot, s_wave, nullptr);
#pragma unroll
                for (int v = 0; v < BPT; ++v) { const int b = BPT * (int)threadIdx.x + v; if (b < nbins) tstart[b] = run; run += cn[v]; }
            }
            lds_sync();
#pragma unroll
            for (int j = 0; j < NT_PER; ++j) if (br[j] != 0xffffffffu) s_rec[tstart[br[j] & 0x7ffu] + (br[j] >> 11)] = make_uint2(key[j], pay[j]);
            lds_sync();
            // bin-major output: wave w owns the bins [w * nbins / 16, (w + 1) * nbins / 16)
            const int per_wave = (nbins + 15) >> 4;
            for (int bb = wv * per_wave; bb < min(nbins, (wv + 1) * per_wave); bb += 8) {
                const int b = bb + (lane >> 3);
                const bool live = b < min(nbins, (wv + 1) * per_wave);
                const uint32_t st = live ? tstart[b] : 0u, cnt = live ? thist[b] : 0u;
                const uint64_t cur = live ? cursor[b] : 0ull;
                for (uint32_t i = (uint32_t)(lane & 7); __any(i < cnt); i += 8) {
                    if (i < cnt) { const uint2 v = s_rec[st + i]; __builtin_memcpy(o_rec + 2 * (cur + i), &v, 8); }
                }
            }
            lds_sync();
            for (int b = threadIdx.x; b < nbins; b += PT_THREADS) cursor[b] += thist[b];
            lds_sync();
        }
    }
}

// ---- the tables' prefix sums.  A table is [row][column] = [super-tile][bucket] (level 1) or, per level-1 bucket,
// [unit][b2] (level 2); the records are laid out column-major (bucket after bucket, inside a bucket row after
// row), so the write offset of (row r, column c) is
//     start(c) + sum of T[r'][c] over r' < r,      start(c) = sum of the totals of the columns < c.
// Rows are contiguous, so all passes read and write whole lines (a one-dimensional scan needs the transposed
// layout: every histogram row a 2^B-way strided write, every cursor load a strided read).
// Level 1 (tens of thousands of rows): column sums per slab of rows, one workgroup scans the slab sums and the
// column totals, the slabs are then rewritten in place.
constexpr int SC_SLAB = 128;          // rows per slab
__global__ void __launch_bounds__(PT_THREADS)
k_scan_columns_a(const uint32_t* __restrict__ T, int64_t n_rows, int n_cols, uint32_t* __restrict__ S) {
    const int64_t r0 = (int64_t)blockIdx.x * SC_SLAB, r1 = min(n_rows, r0 + SC_SLAB);
    for (int c = threadIdx.x; c < n_cols; c += PT_THREADS) {
        uint32_t sum = 0;
        for (int64_t r = r0; r < r1; ++r) sum += T[r * n_cols + c];
        S[(int64_t)blockIdx.x * n_cols + c] = sum;
    }
}
__global__ void __launch_bounds__(PT_THREADS)
k_scan_columns_b(uint32_t* __restrict__ S, int n_slabs, int n_cols /* <= 4 096 */, uint32_t* __restrict__ start /* n_cols + 1 */) {
    __shared__ uint32_t s_wave[16];
    constexpr int CPT = PT_MAXBINS / PT_THREADS;
    uint32_t tot[CPT], sum = 0;
#pragma unroll
    for (int v = 0; v < CPT; ++v) {
        const int c = CPT * (int)threadIdx.x + v;
        tot[v] = 0;
        if (c < n_cols) for (int sl = 0; sl < n_slabs; ++sl) tot[v] += S[(int64_t)sl * n_cols + c];
        sum += tot[v];
    }
    uint32_t total = 0;
    uint32_t run = block_scan_1024(sum, s_wave, &total);
#pragma unroll
    for (int v = 0; v < CPT; ++v) {
        const int c = CPT * (int)threadIdx.x + v;
        if (c >= n_cols) continue;
        start[c] = run;
        uint32_t acc = run;
        for (int sl = 0; sl < n_slabs; ++sl) { const uint32_t t = S[(int64_t)sl * n_cols + c]; S[(int64_t)sl * n_cols + c] = acc; acc += t; }
        run += tot[v];
    }
    if (threadIdx.x == 0) start[n_cols] = total;
}
__global__ void __launch_bounds__(PT_THREADS)
k_scan_columns_c(uint32_t* __restrict__ T, int64_t n_rows, int n_cols, const uint32_t* __restrict__ S) {
    const int64_t r0 = (int64_t)blockIdx.x * SC_SLAB, r1 = min(n_rows, r0 + SC_SLAB);
    for (int c = threadIdx.x; c < n_cols; c += PT_THREADS) {
        uint32_t acc = S[(int64_t)blockIdx.x * n_cols + c];
        for (int64_t r = r0; r < r1; ++r) { const uint32_t t = T[r * n_cols + c]; T[r * n_cols + c] = acc; acc += t; }
    }
}
// Level 2: one workgroup per level-1 bucket scans its [unit][b2] table in place (a few dozen rows); the offsets
// start at the bucket's own start, and the starts of its final buckets go to boff.
__global__ void __launch_bounds__(PT_THREADS)
k_scan_units(uint32_t* __restrict__ T, int n_u, int nb2 /* <= 4 096 */, const uint32_t* __restrict__ off1, int nb1, uint32_t* __restrict__ boff) {
    __shared__ uint32_t s_wave[16];
    constexpr int CPT = PT_MAXBINS / PT_THREADS;
    const int b1 = blockIdx.x;
    uint32_t* tab = T + (size_t)b1 * n_u * nb2;
    uint32_t tot[CPT], sum = 0;
#pragma unroll
    for (int v = 0; v < CPT; ++v) {
        const int c = CPT * (int)threadIdx.x + v;
        tot[v] = 0;
        if (c < nb2) for (int u = 0; u < n_u; ++u) tot[v] += tab[(size_t)u * nb2 + c];
        sum += tot[v];
    }
    uint32_t run = off1[b1] + block_scan_1024(sum, s_wave, nullptr);
#pragma unroll
    for (int v = 0; v < CPT; ++v) {
        const int c = CPT * (int)threadIdx.x + v;
        if (c >= nb2) continue;
        boff[(size_t)b1 * nb2 + c] = run;
        uint32_t acc = run;
        for (int u = 0; u < n_u; ++u) { const uint32_t t = tab[(size_t)u * nb2 + c]; tab[(size_t)u * nb2 + c] = acc; acc += t; }
        run += tot[v];
    }
    if (b1 == nb1 - 1 && threadIdx.x == 0) boff[(size_t)nb1 * nb2] = off1[nb1];
}

// One workgroup per bucket.  The bucket's keys agree on their top `pbits` bits and are uniform below them, so a
// counting sort on the NEXT 10 bits (LDS histogram, scan, scatter from registers) leaves sub-bins that hold one
// or two runs of equal k-mers, contiguous in the LDS; every entry then ranks itself inside its sub-bin by
// comparing with the few members.  From the same loop it learns its run: start in the fully sorted list,
// length, its place in position order, and the entry in front of it (duplicate test).  Output exactly as
// k_group_runs: gen[] at the final place, one row pointer scattered; the sorted keys are never written.
constexpr int BK_PER = BK_CAP / BK_THREADS;
constexpr int BK_MAXBIN = 768;           // a sub-bin beyond this (one k-mer occurring hundreds of times) takes the general path
// NARROW: 8-byte records (key bits below the bucket's in one word, position): an entry is ONE u64 `key << 32 | pos`
// in the LDS, so the ranking loop reads a single word per sub-bin member and every relation it needs (smaller
// key, equal key, equal key at a smaller position, the predecessor in the run) is a compare of that word.
// THREADS x BK_PER = the largest bucket taken; SUBBITS: digit of the LDS counting sort (sub-bins of ~2 entries)
template <bool NARROW, int THREADS, int SUBBITS>
__global__ void __launch_bounds__(THREADS)
k_bucket_runs(const uint32_t* __restrict__ rec, int stride /* 3: (w0, w1, pay); 2: (key bits below the bucket's, pay) */,
              const uint32_t* __restrict__ boff, int64_t n_buckets, int pbits, const uint32_t* __restrict__ blk2g, int blk_shift,
              uint32_t* __restrict__ gen, uint32_t* __restrict__ rowinfo, compact_map M, int* __restrict__ dup_per_genome,
              const uint32_t* __restrict__ bucket_list /* null: all buckets; else the n_buckets listed ones */,
              uint32_t* __restrict__ over_list, unsigned int* __restrict__ n_over) {
    constexpr int BK_SUB = 1 << SUBBITS, CAP = THREADS * BK_PER;
    static_assert(CAP <= 8192 && BK_MAXBIN <= 1024, "slot | rank << 13 of the run members");
    __shared__ uint32_t s_big;
    __shared__ uint64_t sk[CAP];                  // NARROW: key << 32 | pos; else (w0 << 32) | w1 -- in sub-bin order
    __shared__ uint32_t sp[NARROW ? 1 : CAP];
    __shared__ uint32_t sgen[CAP];                  // gen[] of the bucket in final order: leaves as one contiguous copy
    __shared__ uint32_t cnt[BK_SUB + 1], start[BK_SUB + 1];
    __shared__ uint32_t s_wave[THREADS / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t it = blockIdx.x; it < n_buckets; it += gridDim.x) {
        const int64_t bk = bucket_list ? (int64_t)bucket_list[it] : it;
        const uint32_t b0 = boff[bk], b1 = boff[bk + 1];
        const uint32_t n_u = b1 - b0;
        if (n_u <= 1) continue;                                // empty, or one singleton k-mer
        // a bucket this variant does not take is queued (for the larger variant, then for k_bucket_big): nothing
        // of it has been written when it is handed on
        if (n_u > (uint32_t)CAP) { if (threadIdx.x == 0) over_list[atomicAdd(n_over, 1u)] = (uint32_t)bk; continue; }
        const int n = (int)n_u;
        lds_sync();
        for (int b = threadIdx.x; b <= BK_SUB; b += THREADS) cnt[b] = 0;
        if (threadIdx.x == 0) s_big = 0;
        lds_sync();
        uint64_t key[BK_PER]; uint32_t pj[BK_PER]; uint32_t sb[BK_PER], ar[BK_PER];
        if (NARROW) {
            // two 8-byte records per 16-byte load: entries 2 p and 2 p + 1 of the thread are the records 2 (p * THREADS + t), + 1
            static_assert(BK_PER % 2 == 0, "pairs of records per thread");
#pragma unroll
            for (int p = 0; p < BK_PER / 2; ++p) {
                const int j = 2 * (p * THREADS + (int)threadIdx.x);
                uint4 rr = make_uint4(0u, 0u, 0u, 0u);
                if (j + 1 < n) __builtin_memcpy(&rr, rec + 2 * (uint64_t)(b0 + j), 16);
                else if (j < n) __builtin_memcpy(&rr.x, rec + 2 * (uint64_t)(b0 + j), 8);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * p + h;
                    const uint32_t a = h ? rr.z : rr.x, pay = h ? rr.w : rr.y;
                    sb[q] = 0; ar[q] = 0; key[q] = 0; pj[q] = 0;
                    if (j + h < n) { pj[q] = pay; key[q] = ((uint64_t)a << 32) | pay; sb[q] = a >> (32 - SUBBITS); }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < BK_PER; ++q) {
                const int j = q * THREADS + threadIdx.x;
                sb[q] = 0; ar[q] = 0; key[q] = 0; pj[q] = 0;
                if (j < n) {
                    const uint32_t* r = rec + (uint64_t)(b0 + j) * stride;
                    const uint32_t a = r[0], c = stride == 3 ? r[1] : 0u;
                    pj[q] = r[stride - 1];
                    key[q] = ((uint64_t)a << 32) | c;
                    sb[q] = (uint32_t)((key[q] << pbits) >> (64 - SUBBITS));
                }
            }
        }
        // (entry q of the thread: record q * THREADS + t, or with pairs 2 ((q / 2) * THREADS + t) + q % 2)
        auto rec_of = [&](int q) -> int { return NARROW ? 2 * ((q >> 1) * THREADS + (int)threadIdx.x) + (q & 1) : q * THREADS + (int)threadIdx.x; };
#pragma unroll
        for (int q = 0; q < BK_PER; ++q) if (rec_of(q) < n) ar[q] = atomicAdd(&cnt[sb[q]], 1u);
        lds_sync();
        {   // exclusive scan of the sub-bin counters: BK_SUB / THREADS per thread
            constexpr int CPT = BK_SUB / THREADS;
            uint32_t c4[CPT], tot = 0;
#pragma unroll
            for (int u = 0; u < CPT; ++u) { c4[u] = cnt[CPT * threadIdx.x + u]; tot += c4[u]; if (c4[u] > (uint32_t)BK_MAXBIN) s_big = 1; }
            uint32_t x = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
            if (lane == 63) s_wave[wv] = x;
            lds_sync();
            uint32_t base = 0;
            for (int i = 0; i < wv; ++i) base += s_wave[i];
            uint32_t run = base + x - tot;
#pragma unroll
            for (int u = 0; u < CPT; ++u) { start[CPT * threadIdx.x + u] = run; run += c4[u]; }
            if (threadIdx.x == THREADS - 1) start[BK_SUB] = run;
        }
        lds_sync();
        // a sub-bin beyond BK_MAXBIN (one k-mer occurring hundreds of times): the in-bin ranking is quadratic
        if (s_big) { if (threadIdx.x == 0) over_list[atomicAdd(n_over, 1u)] = (uint32_t)bk; continue; }
#pragma unroll
        for (int q = 0; q < BK_PER; ++q) if (rec_of(q) < n) {
            const uint32_t slot = start[sb[q]] + ar[q];
            sk[slot] = key[q]; if (!NARROW) sp[slot] = pj[q];
        }
        lds_sync();
        // Every entry of a run of >= 2 learns its place in the sorted list (start of the run + rank by position) and puts
        // its genome there.  Whether it repeats a k-mer of its OWN genome is read off that list afterwards: the entry in
        // front of it in the run is the slot in front of it (the loop carries three counters, no predecessor, and the
        // predecessor's genome is an LDS word instead of a second look-up in global memory).
        uint32_t at[BK_PER], gq[BK_PER];                       // slot in the bucket | rank in the run << 13 (CAP <= 8192, rank < BK_MAXBIN), genome; ~0: nothing to do
#pragma unroll
        for (int q = 0; q < BK_PER; ++q) {
            at[q] = 0xffffffffu; gq[q] = 0;
            if (rec_of(q) >= n) continue;
            const uint32_t s0 = start[sb[q]], s1 = start[sb[q] + 1];
            if (s1 - s0 < 2) continue;                           // alone in its sub-bin: a singleton k-mer
            const uint64_t kq = key[q]; const uint32_t pq = pj[q];
            uint32_t lt = 0, eq = 0, before = 0;
            if (NARROW) {
                const uint32_t kh = (uint32_t)(kq >> 32);
#pragma unroll 4
                for (uint32_t t = s0; t < s1; ++t) {
                    const uint64_t vt = sk[t];
                    const uint32_t th = (uint32_t)(vt >> 32), pt = (uint32_t)vt;
                    lt += th < kh;
                    const bool same = th == kh;
                    eq += same;
                    before += same && pt < pq;                   // same key: the order of the entries is that of the positions
                }
            } else {
                for (uint32_t t = s0; t < s1; ++t) {
                    const uint64_t kt = sk[t];
                    lt += kt < kq;
                    if (kt == kq) { ++eq; before += sp[t] < pq; }
                }
            }
            if (eq < 2) continue;                                // singleton k-mer: no partner, nobody reads its gen[] slot
            const uint32_t g = M.cblk ? genome_of_compact(M, pq) : blk2g[pq >> blk_shift];
            sgen[s0 + lt + before] = g;
            if (before) { at[q] = (s0 + lt + before) | (before << 13); gq[q] = g; }      // (the run's smallest position: no partner b < a, not a repeat)
        }
        lds_sync();
#pragma unroll
        for (int q = 0; q < BK_PER; ++q) {
            if (at[q] == 0xffffffffu) continue;
            const uint32_t slot = at[q] & 0x1fffu, g = gq[q];
            if ((sgen[slot - 1] & ~DUP_BIT) == g) { sgen[slot] = g | DUP_BIT; atomicAdd(&dup_per_genome[g], 1); continue; }
            rowinfo[pj[q]] = b0 + (slot - (at[q] >> 13)) + 1u;   // 1 + the first entry of this k-mer's run in the sorted list
        }
        lds_sync();
        // (the slots of singleton k-mers carry whatever the LDS held: nobody reads them)
        for (uint32_t i = 4u * threadIdx.x; i < (uint32_t)n; i += 4u * THREADS) {
            if (i + 4 <= (uint32_t)n) { const uint4 v = make_uint4(sgen[i], sgen[i + 1], sgen[i + 2], sgen[i + 3]); __builtin_memcpy(gen + b0 + i, &v, 16); }
            else for (uint32_t j = i; j < (uint32_t)n; ++j) gen[b0 + j] = sgen[j];
        }
    }
}

// ---- buckets beyond the LDS variants (a k-mer present in thousands of genomes, poly-A runs, conserved genes):
// one 1 024-thread workgroup sorts the bucket in global scratch -- a stable LSD radix sort on (key, position), eight
// bits per pass, passes whose digit is constant skipped -- and then writes runs, duplicates, gen[] and row pointers
// from the sorted order.  Duplicates (the same k-mer again in the same genome) are squeezed OUT of the genome list:
// the non-duplicates of a run are written first, so a walk over a poly-A run of 100 000 entries in 2 000 genomes
// touches 2 000 entries, not 100 000 (the tail of the run's slots is never read: walks end at their own genome).
// Scratch per entry: two (u64 key, u32 position) copies; the dead copy of the last pass holds the two scan arrays.
constexpr int BB_THREADS = 1024;
constexpr int BB_DIGITS = 12;            // 4 bytes of position, then 8 bytes of key
__device__ __forceinline__ uint32_t bb_scan_incl_add(uint32_t v, uint32_t* s_w, uint32_t* carry) {
    // inclusive prefix sum over the workgroup, plus *carry (sum of the chunks before); returns with *carry advanced
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wv] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BB_THREADS / 64; ++i) { const uint32_t t = s_w[i]; if (i < wv) base += t; tot += t; }
    const uint32_t r = *carry + base + x;
    __syncthreads();
    if (threadIdx.x == 0) *carry += tot;
    __syncthreads();
    return r;
}
__device__ __forceinline__ uint32_t bb_scan_incl_max(uint32_t v, uint32_t* s_w, uint32_t* carry) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x = max(x, y); }
    if (lane == 63) s_w[wv] = x;
    __syncthreads();
    uint32_t base = *carry, tot = *carry;
#pragma unroll
    for (int i = 0; i < BB_THREADS / 64; ++i) { const uint32_t t = s_w[i]; if (i < wv) base = max(base, t); tot = max(tot, t); }
    const uint32_t r = max(base, x);
    __syncthreads();
    if (threadIdx.x == 0) *carry = tot;
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(BB_THREADS)
k_bucket_big(const uint32_t* __restrict__ rec, int stride, const uint32_t* __restrict__ boff, const uint32_t* __restrict__ list,
             const uint64_t* __restrict__ soff /* scratch offset of every listed bucket, in entries */, int pbits,
             uint64_t* __restrict__ kA, uint32_t* __restrict__ pA, uint64_t* __restrict__ kB, uint32_t* __restrict__ pB,
             const uint32_t* __restrict__ blk2g, int blk_shift, uint32_t* __restrict__ gen, uint32_t* __restrict__ rowinfo,
             compact_map M, int* __restrict__ dup_per_genome) {
    __shared__ uint32_t hist[BB_DIGITS][256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t wcnt[BB_THREADS / 64][256];
    __shared__ uint32_t s_w[BB_THREADS / 64];
    __shared__ uint32_t s_carry_a, s_carry_b;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t bk = list[blockIdx.x];
    const uint32_t b0 = boff[bk];
    const uint32_t n = boff[bk + 1] - b0;
    uint64_t* k0 = kA + soff[blockIdx.x]; uint32_t* p0 = pA + soff[blockIdx.x];
    uint64_t* k1 = kB + soff[blockIdx.x]; uint32_t* p1 = pB + soff[blockIdx.x];
    for (int i = threadIdx.x; i < BB_DIGITS * 256; i += BB_THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();
    // copy in + all twelve digit histograms (a histogram does not depend on the order of the entries)
    for (uint32_t i = threadIdx.x; i < n; i += BB_THREADS) {
        const uint32_t* r = rec + (uint64_t)(b0 + i) * stride;
        uint64_t key; uint32_t pos;
        if (stride == 2) { key = r[0]; pos = r[1]; }
        else { key = ((((uint64_t)r[0] << 32) | r[1]) << pbits) >> pbits; pos = r[2]; }
        k0[i] = key; p0[i] = pos;
#pragma unroll
        for (int d = 0; d < 4; ++d) atomicAdd(&hist[d][(pos >> (8 * d)) & 255u], 1u);
#pragma unroll
        for (int d = 0; d < 8; ++d) atomicAdd(&hist[4 + d][(uint32_t)(key >> (8 * d)) & 255u], 1u);
    }
    __syncthreads();
    for (int d = 0; d < BB_DIGITS; ++d) {
        // a digit that is the same in every entry needs no pass
        bool trivial = false;
        if (threadIdx.x < 256 && hist[d][threadIdx.x] == n) trivial = true;
        if (__syncthreads_or(trivial)) continue;
        if (threadIdx.x < 64) {          // exclusive scan of the 256 counters by one wave, four per lane
            uint32_t c[4], tot = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { c[u] = hist[d][4 * lane + u]; tot += c[u]; }
            uint32_t x = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
            uint32_t run = x - tot;
#pragma unroll
            for (int u = 0; u < 4; ++u) { gbase[4 * lane + u] = run; run += c[u]; }
        }
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += BB_THREADS) {
            const uint32_t i = c0 + threadIdx.x;
            const bool in = i < n;
            uint64_t key = 0; uint32_t pos = 0;
            if (in) { key = k0[i]; pos = p0[i]; }
            const uint32_t dg = in ? (d < 4 ? (pos >> (8 * d)) & 255u : (uint32_t)(key >> (8 * (d - 4))) & 255u) : 256u;
            // lanes of the wave holding the same digit (stable: rank = equal lanes below)
            unsigned long long same = __ballot(in);
#pragma unroll
            for (int b = 0; b < 8; ++b) { const unsigned long long bal = __ballot((dg >> b) & 1u); same &= ((dg >> b) & 1u) ? bal : ~bal; }
            const uint32_t rank = (uint32_t)__popcll(same & ((1ULL << lane) - 1ULL));
            for (int j = threadIdx.x; j < (BB_THREADS / 64) * 256; j += BB_THREADS) (&wcnt[0][0])[j] = 0;
            __syncthreads();
            if (in && rank == 0) wcnt[wv][dg] = (uint32_t)__popcll(same);
            __syncthreads();
            uint32_t before = 0;
            if (in) for (int w2 = 0; w2 < wv; ++w2) before += wcnt[w2][dg];
            const uint32_t dst = in ? gbase[dg] + before + rank : 0u;
            __syncthreads();
            if (threadIdx.x < 256) { uint32_t t = 0; for (int w2 = 0; w2 < BB_THREADS / 64; ++w2) t += wcnt[w2][threadIdx.x]; gbase[threadIdx.x] += t; }
            if (in) { k1[dst] = key; p1[dst] = pos; }
            __syncthreads();
        }
        { uint64_t* tk = k0; k0 = k1; k1 = tk; uint32_t* tp = p0; p0 = p1; p1 = tp; }
        __threadfence_block();
        __syncthreads();
    }
    // (k0, p0) = the bucket in (key, position) order.  Pass 1: head flags, duplicates; inclusive scans of
    // "not a duplicate" (S) and of the run start (RS) go to the dead copy
    uint32_t* S = reinterpret_cast<uint32_t*>(k1);        // 2 u32 per entry
    if (threadIdx.x == 0) { s_carry_a = 0; s_carry_b = 0; }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n; c0 += BB_THREADS) {
        const uint32_t i = c0 + threadIdx.x;
        uint32_t nd = 0, hd = 0;
        if (i < n) {
            const uint64_t key = k0[i]; const uint32_t pos = p0[i];
            const bool head = i == 0 || k0[i - 1] != key;
            bool dup = false;
            if (!head) {
                const uint32_t g = M.cblk ? genome_of_compact(M, pos) : blk2g[pos >> blk_shift];
                const uint32_t pp = p0[i - 1];
                dup = (M.cblk ? genome_of_compact(M, pp) : blk2g[pp >> blk_shift]) == g;
            }
            nd = dup ? 0u : 1u; hd = head ? i : 0u;
        }
        const uint32_t s_incl = bb_scan_incl_add(nd, s_w, &s_carry_a);
        const uint32_t rs = bb_scan_incl_max(hd, s_w, &s_carry_b);
        if (i < n) { S[2 * (uint64_t)i] = s_incl | (nd ? 0u : 0x80000000u); S[2 * (uint64_t)i + 1] = rs; }
    }
    __threadfence_block();
    __syncthreads();
    // Pass 2: outputs
    for (uint32_t i = threadIdx.x; i < n; i += BB_THREADS) {
        const uint32_t sv = S[2 * (uint64_t)i], rs = S[2 * (uint64_t)i + 1];
        const uint32_t pos = p0[i];
        const uint32_t g = M.cblk ? genome_of_compact(M, pos) : blk2g[pos >> blk_shift];
        if (sv & 0x80000000u) { atomicAdd(&dup_per_genome[g], 1); continue; }
        const uint32_t base = rs ? (S[2 * (uint64_t)(rs - 1)] & 0x7fffffffu) : 0u;
        const uint32_t rank = (sv & 0x7fffffffu) - 1u - base;       // among the run's non-duplicates
        // (a singleton k-mer writes its own slot: nobody reads it)
        gen[(uint64_t)b0 + rs + rank] = g;
        if (rank > 0) rowinfo[pos] = b0 + rs + 1u;
    }
}

__global__ void k_bucket_sizes(const uint32_t* __restrict__ boff, const uint32_t* __restrict__ list, int64_t n, uint32_t* __restrict__ sizes) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) sizes[i] = boff[list[i] + 1] - boff[list[i]];
}

inline int grid_for(int64_t n, int block = 256, int max_blocks = 256 * 16) {
    int64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    return (int)std::min<int64_t>(b, max_blocks);
}

struct max_op { __device__ __host__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

}  // namespace

static int g_force_subshards = 0;
// k-mers one pass is cut for: its row numbers are 32 bits (4.29e9) and sub-sharding starts at SUB_PASS_START expected
// k-mers.  The expectation is an upper bound (every padded position counted) and HASH shards are even to a few 10^-5; a
// pass that overflows all the same throws VG_EOVERFLOW and the loop is retried one cut finer.  (4.2e9 per pass would
// make 10^6 contigs six passes instead of seven; its record buffers then peak near the device's memory -- ~250 GB are
// in use at seven -- so the cut stays at 3.6e9.)
constexpr double SUB_PASS_KMERS = 3.6e9, SUB_PASS_START = 3.9e9;

// RANGE or HASH shards (see kmer_args): a property of the set and the fraction, the same on every rank and in every pass
static bool range_shards(const vg_genomes* g, double fraction, int n_shards) {
    return n_shards > 1 && n_shards <= 256 && !(fraction < 1.0) && g->padded_total() < (1LL << 32);
}
static kmer_args make_kmer_args(const vg_genomes* g, int k, double fraction, int shard, int n_shards) {
    const int use_frac = fraction < 1.0;
    kmer_args A{ vg_genome_planes(g, vg_stream()), g->d_nmask.p, g->d_blk2g.p, g->d_base_off.p, g->d_len.p, g->padded_total(), k, use_frac,
                 use_frac ? (uint64_t)std::ldexp(fraction, 64) : ~0ULL, (uint32_t)shard, (uint32_t)n_shards, g->align_shift, 0u, 1u << DIG_BITS, 1u << DIG_BITS };
    if (range_shards(g, fraction, n_shards)) {
        A.dig_max = (uint32_t)(((1u << DIG_BITS) + (uint32_t)n_shards - 1u) / (uint32_t)n_shards);
        const uint32_t lo = (uint32_t)(((uint64_t)shard << DIG_BITS) / (uint64_t)n_shards), hi = (uint32_t)(((uint64_t)(shard + 1) << DIG_BITS) / (uint64_t)n_shards);
        A.shard = 0; A.n_shards = 1; A.dig_lo = lo; A.dig_n = hi - lo;
    }
    return A;
}

// shared pipeline: extract -> sort.  Returns sorted keys/pos of the n_valid real k-mers and the
// per-genome number of k-mers kept by extraction.  compact = true: only kept k-mers were written
// and the sort payload is the compact index itself (wave_base / cblk map it back to its genome).
struct sorted_index {
    dbuf<uint64_t> keys; dbuf<uint32_t> pos; dbuf<int> kept; int64_t n_valid = 0;
    bool compact = false; dbuf<unsigned long long> wave_mask; dbuf<uint32_t> wave_base; dbuf<uint32_t> cblk, goff;
    int low_bit = 0;            // keys are ordered on bits >= low_bit only (finish = false)
    dbuf<uint64_t> spare64; dbuf<uint32_t> spare32;     // buffers the sort no longer needs (reused by the caller)
};

// general path: order the list on ALL key bits (the prefix order is thrown away: a stable LSD sort
// from bit 0 keeps equal k-mers in position order) -- O(n) whatever the data looks like
static void finish_sort(sorted_index& si, int k) {
    if (si.low_bit > 0 && si.n_valid > 0) {
        hipStream_t s = vg_stream();
        const size_t n = (size_t)si.n_valid;
        dbuf<uint64_t> k2(n); dbuf<uint32_t> p2(n);
        size_t tmp_bytes = 0;
        VG_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, si.keys.p, k2.p, si.pos.p, p2.p, n, 0u, (unsigned)(2 * k), s));
        dbuf<char> tmp(tmp_bytes);
        vg_prof_scope ps("radix_sort_full", (double)n * 12.0 * 2.0 * ((2 * k + 7) / 8));
        VG_HIP(rocprim::radix_sort_pairs((void*)tmp.p, tmp_bytes, si.keys.p, k2.p, si.pos.p, p2.p, n, 0u, (unsigned)(2 * k), s));
        VG_HIP(hipStreamSynchronize(s));
        si.keys = std::move(k2); si.pos = std::move(p2);
    }
    si.low_bit = 0;
}

// The k-mer scan of a sub-shard (k_kmer_count: arithmetic-bound, it reads the genomes and writes only its own buffers)
// does not depend on anything the previous sub-shard computes: the sub-shard loop launches the scan of sub-shard t + 1
// on the second queue when sub-shard t has finished its own scan, beside t's partition / bucket / SpGEMM kernels
// (memory- and LDS-bound).  Safe with the single-queue caching allocator: the buffers are taken while the library
// queue is idle (run_extract_sort has just synchronised), so no block they receive has work pending on it, and they
// are handed on to run_extract_sort(t + 1), which waits for the scan's event before anything reads them.
struct precount {
    dbuf<int> kept; dbuf<unsigned long long> wave_mask; dbuf<uint32_t> wave_cnt; dbuf<uint64_t> stage; dbuf<unsigned int> d_over;
    const vg_genomes* g = nullptr; int k = 0, shard = -1, n_shards = 0, stage_cap = 0; hipEvent_t done = nullptr;
    void drop() {
        if (done) { (void)hipEventSynchronize(done); (void)hipEventDestroy(done); done = nullptr; }     // nothing is freed under a running scan
        kept.release(); wave_mask.release(); wave_cnt.release(); stage.release(); d_over.release(); shard = -1; g = nullptr;
    }
    ~precount() { drop(); }
};
static precount g_precount;
// the index stage's milliseconds (level-1 count ... bucket kernels) of the last pass (placement trials of vg_kmer_shared; measured only while they run)
static bool g_time_bucket_pass = false;
static float g_last_bucket_ms = 0.f;
struct pass_timer {
    float* out; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr;
    pass_timer(float* o, hipStream_t st) : out(o), s(st) { if (out && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, s); else out = nullptr; }
    ~pass_timer() { if (!out) return; (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1); float ms = 0; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *out = ms; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
};
// the kept mask of the pass that is about to run, left by the sub-shard loop's one scan of all sub-shards (k_multi_mask)
struct pass_mask { const vg_genomes* g = nullptr; int k = 0, shard = -1, n_shards = 0; const unsigned long long* mask = nullptr; };
static pass_mask g_pass_mask;
static int compact_stage_cap(double keep) { return (int)std::min<double>(256.0, std::ceil(1.5 * 256.0 * keep) + 24.0); }
// to be called while the library queue is idle
static void launch_precount(vg_genomes* g, int k, int shard, int n_shards) {
    precount& pc = g_precount;
    pc.drop();
    const int64_t P = g->padded_total(), W = P / 64, n_chunks = (W + 3) / 4;
    pc.stage_cap = compact_stage_cap(1.0 / n_shards);
    pc.kept.alloc((size_t)std::max(1, g->n)); pc.wave_mask.alloc((size_t)W + 1); pc.wave_cnt.alloc((size_t)W + 1);
    pc.stage.alloc((size_t)n_chunks * pc.stage_cap); pc.d_over.alloc(1);
    hipStream_t side = vg_side_stream();
    pc.kept.zero(side); pc.d_over.zero(side);
    VG_HIP(hipMemsetAsync(pc.wave_cnt.p + W, 0, sizeof(uint32_t), side));
    const kmer_args A = make_kmer_args(g, k, 1.0, shard, n_shards);
    if (k == 25) hipLaunchKernelGGL(k_kmer_count<25>, dim3(grid_for((P + 255) / 4)), dim3(256), 0, side, A, pc.wave_mask.p, pc.wave_cnt.p, pc.kept.p, pc.stage.p, pc.stage_cap, pc.d_over.p);
    else hipLaunchKernelGGL(k_kmer_count<0>, dim3(grid_for((P + 255) / 4)), dim3(256), 0, side, A, pc.wave_mask.p, pc.wave_cnt.p, pc.kept.p, pc.stage.p, pc.stage_cap, pc.d_over.p);
    VG_HIP(hipEventCreateWithFlags(&pc.done, hipEventDisableTiming));
    VG_HIP(hipEventRecord(pc.done, side));
    pc.g = g; pc.k = k; pc.shard = shard; pc.n_shards = n_shards;
}

static void run_extract_sort(vg_genomes* g, int k, double fraction, int shard, int n_shards, sorted_index& out, bool finish = true, bool do_sort = true) {
    hipStream_t s = vg_stream();
    const int64_t P = g->padded_total();
    // the scan of this very sub-shard may already be running (or done) on the second queue
    const bool pre = g_precount.g == g && g_precount.k == k && g_precount.shard == shard && g_precount.n_shards == n_shards && !(fraction < 1.0) && n_shards > 1;
    const unsigned long long* premask = (g_pass_mask.g == g && g_pass_mask.k == k && g_pass_mask.shard == shard && g_pass_mask.n_shards == n_shards && !pre) ? g_pass_mask.mask : nullptr;
    if (pre) { VG_HIP(hipStreamWaitEvent(s, g_precount.done, 0)); out.kept = std::move(g_precount.kept); }
    else { out.kept.alloc((size_t)std::max(1, g->n)); out.kept.zero(s); }
    const int use_frac = fraction < 1.0;
    const kmer_args A = make_kmer_args(g, k, fraction, shard, n_shards);
    out.compact = use_frac || n_shards > 1;
    if (!out.compact && P >= (1LL << 32)) throw vg_error(VG_EOVERFLOW, "dense k-mer pass needs < 2^32 padded bases");
    if (P >= (1LL << 37)) throw vg_error(VG_EOVERFLOW, "genome set exceeds 2^37 padded bases");
    dbuf<uint64_t> keys_a, keys_b; dbuf<uint32_t> pos_a, pos_b;
    int64_t n_sort = P; unsigned long long nv = 0;
    if (!out.compact) {
        keys_a.alloc((size_t)P); keys_b.alloc((size_t)P); pos_a.alloc((size_t)P + 4); pos_b.alloc((size_t)P);
        dbuf<unsigned long long> d_nvalid(1); d_nvalid.zero(s);
        {
            vg_prof_scope ps("kmer_extract", (double)P * (3.0 / 8.0 + 8.0));
            hipLaunchKernelGGL(k_kmer_extract, dim3(grid_for((P + 255) / 4)), dim3(256), 0, s, A, keys_a.p, d_nvalid.p, out.kept.p);
        }
        hipLaunchKernelGGL(k_iota, dim3(grid_for(P)), dim3(256), 0, s, pos_a.p, P);
        d_nvalid.download(&nv, 1, s);
    } else {
        const int64_t W = P / 64;
        // slots per 256-position chunk for the kept k-mers (1.5 x the expectation + slack; a chunk that
        // overflows sends the call to the recomputing emit pass)
        const double keep = (use_frac ? fraction : 1.0) / n_shards;
        const int stage_cap = compact_stage_cap(keep);
        const int64_t n_chunks = (W + 3) / 4;
        dbuf<uint32_t> wave_cnt; dbuf<uint64_t> stage; dbuf<unsigned int> d_over;
        out.wave_base.alloc((size_t)W + 1);
        if (pre) {
            out.wave_mask = std::move(g_precount.wave_mask); wave_cnt = std::move(g_precount.wave_cnt);
            stage = std::move(g_precount.stage); d_over = std::move(g_precount.d_over);
            (void)hipEventDestroy(g_precount.done); g_precount.done = nullptr; g_precount.shard = -1; g_precount.g = nullptr;
        } else if (premask) {
            // the pass's kept mask exists already (one scan served every sub-shard): count its words; the k-mers are computed
            // for the kept positions only by the emit pass below
            out.wave_mask.view(const_cast<unsigned long long*>(premask), (size_t)W + 1); wave_cnt.alloc((size_t)W + 1);
            VG_HIP(hipMemsetAsync(wave_cnt.p + W, 0, sizeof(uint32_t), s));
            if (W > 0) hipLaunchKernelGGL(k_mask_popc, dim3(grid_for(W)), dim3(256), 0, s, premask, W, wave_cnt.p);
        } else {
            out.wave_mask.alloc((size_t)W + 1); wave_cnt.alloc((size_t)W + 1);
            VG_HIP(hipMemsetAsync(wave_cnt.p + W, 0, sizeof(uint32_t), s));
            stage.alloc((size_t)n_chunks * stage_cap);
            d_over.alloc(1); d_over.zero(s);
        }
        if (!pre && !premask) {
            vg_prof_scope ps("kmer_count", (double)P * (3.0 / 8.0 + 12.0 / 64.0));
            if (A.k == 25 && !A.use_frac)
                hipLaunchKernelGGL(k_kmer_count<25>, dim3(grid_for((P + 255) / 4)), dim3(256), 0, s, A, out.wave_mask.p, wave_cnt.p, out.kept.p,
                                   stage.p, stage_cap, d_over.p);
            else
                hipLaunchKernelGGL(k_kmer_count<0>, dim3(grid_for((P + 255) / 4)), dim3(256), 0, s, A, out.wave_mask.p, wave_cnt.p, out.kept.p,
                                   stage.p, stage_cap, d_over.p);
        }
        size_t tb = 0;
        VG_HIP(rocprim::exclusive_scan(nullptr, tb, wave_cnt.p, out.wave_base.p, 0u, (size_t)W + 1, rocprim::plus<uint32_t>(), s));
        dbuf<char> tmp(tb);
        VG_HIP(rocprim::exclusive_scan((void*)tmp.p, tb, wave_cnt.p, out.wave_base.p, 0u, (size_t)W + 1, rocprim::plus<uint32_t>(), s));
        uint32_t total = 0; unsigned int over = premask ? 1u : 0u;          // (a given mask: no staged k-mers, the emit pass computes them)
        int64_t total64 = 0;
        if (premask) {
            // the words' counts are 32 bits each, their sum may not be: add them up in 64 bits before trusting the scan
            dbuf<unsigned long long> d_tot(1); size_t tb2 = 0;
            VG_HIP(rocprim::reduce(nullptr, tb2, wave_cnt.p, d_tot.p, 0ULL, (size_t)W + 1, rocprim::plus<unsigned long long>(), s));
            dbuf<char> tmp2(tb2);
            VG_HIP(rocprim::reduce((void*)tmp2.p, tb2, wave_cnt.p, d_tot.p, 0ULL, (size_t)W + 1, rocprim::plus<unsigned long long>(), s));
            unsigned long long t64 = 0; d_tot.download(&t64, 1, s);
            VG_HIP(hipMemcpyAsync(&total, out.wave_base.p + W, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            VG_HIP(hipStreamSynchronize(s));
            total64 = (int64_t)t64;
        } else {
            std::vector<int> kept_h((size_t)std::max(1, g->n));
            VG_HIP(hipMemcpyAsync(&total, out.wave_base.p + W, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            d_over.download(&over, 1, s);
            out.kept.download(kept_h.data(), (size_t)g->n, s);
            VG_HIP(hipStreamSynchronize(s));
            for (int i = 0; i < g->n; ++i) total64 += kept_h[i];
        }
        if (total64 >= (1LL << 32) - 1) throw vg_error(VG_EOVERFLOW, "more than 2^32 k-mers in one shard: use more shards");
        nv = total; n_sort = (int64_t)total;
        const size_t na = (size_t)std::max<int64_t>(n_sort, 1);
        keys_a.alloc(na); if (do_sort) pos_a.alloc(na + 4);        // (unsorted, a k-mer's row number is its index: the bucket pipeline needs no array for it)
        if (do_sort) { keys_b.alloc(na); pos_b.alloc(na); }         // the sort's outputs (the bucket pipeline sorts nothing)
        vg_host_mark("extract: counted");
        if (n_sort > 0 && !over) {
            vg_prof_scope ps("kmer_emit", (double)n_sort * (8.0 + 12.0));
            hipLaunchKernelGGL(k_kmer_gather, dim3(grid_for(n_chunks * 64)), dim3(256), 0, s, stage.p, stage_cap, out.wave_base.p, W, keys_a.p, pos_a.p);
        } else if (n_sort > 0) {
            vg_prof_scope ps("kmer_emit_recompute", (double)P * 3.0 / 8.0 + (double)n_sort * 12.0);
            // (a wave lists the kept positions of nw words and gives every lane one: about 56 of them, so that a second trip
            // of a few lanes is rare -- eight words at a seventh kept are 73 positions, two trips for every chunk)
            const int nw = (int)std::max<int64_t>(1, std::min<int64_t>(8, 56 * P / (64 * std::max<int64_t>(n_sort, 1))));
            if (n_sort * 4 <= P) hipLaunchKernelGGL(k_kmer_emit_sparse, dim3(grid_for(P / 8)), dim3(256), 0, s, A, out.wave_mask.p, out.wave_base.p, keys_a.p, pos_a.p, nw);
            else hipLaunchKernelGGL(k_kmer_emit, dim3(grid_for(P)), dim3(256), 0, s, A, out.wave_mask.p, out.wave_base.p, keys_a.p, pos_a.p);
        }
        out.cblk.alloc((size_t)(n_sort >> CBLK_SHIFT) + 2); out.cblk.zero(s);
        out.goff.alloc((size_t)g->n + 1);
        hipLaunchKernelGGL(k_cblk, dim3(grid_for(g->n)), dim3(256), 0, s, out.wave_base.p, g->d_base_off.p, g->n, out.cblk.p, out.goff.p);
        // (a given mask: the kept k-mers of a genome are its rows)
        if (premask && g->n > 0) hipLaunchKernelGGL(k_kept_from_goff, dim3(grid_for(g->n)), dim3(256), 0, s, (const uint32_t*)out.goff.p, g->n, out.kept.p);
    }
    // Stable LSD radix sort on the top key bits only (bit 2k is the sentinel flag, so sentinels end
    // up behind every real k-mer); k_group_runs then orders the small equal-prefix groups.
    const unsigned int end_bit = (unsigned)(2 * k + 1);
    // a few entries per equal-prefix group on average: the run kernel finishes the order in the LDS,
    // at a cost that grows with the group size (measured break-even: ~2.5 entries per prefix)
    unsigned int sort_bits = n_sort <= (5LL << 23) ? 24u : (n_sort <= (5LL << 31) ? 32u : 40u);
    if (sort_bits > end_bit) sort_bits = end_bit;
    const unsigned int begin_bit = end_bit - sort_bits;
    if (!do_sort) {
        // the bucket pipeline takes the kept k-mers as they are (compact mode only: every entry is a real k-mer)
        VG_HIP(hipStreamSynchronize(s));
        vg_host_mark("extract: emitted");
        out.keys = std::move(keys_a); out.pos = std::move(pos_a); out.n_valid = (int64_t)nv; out.low_bit = (int)end_bit;
        return;
    }
    if (n_sort > 0) {
        size_t tmp_bytes = 0;
        VG_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a.p, keys_b.p, pos_a.p, pos_b.p, (size_t)n_sort, begin_bit, end_bit, s));
        dbuf<char> tmp(tmp_bytes);
        int passes = (int)(sort_bits + 7) / 8;
        vg_prof_scope ps("radix_sort_pairs", (double)n_sort * 12.0 * 2.0 * passes);
        VG_HIP(rocprim::radix_sort_pairs((void*)tmp.p, tmp_bytes, keys_a.p, keys_b.p, pos_a.p, pos_b.p, (size_t)n_sort, begin_bit, end_bit, s));
    }
    VG_HIP(hipStreamSynchronize(s));
    out.keys = std::move(keys_b); out.pos = std::move(pos_b); out.n_valid = (int64_t)nv; out.low_bit = (int)begin_bit;
    out.spare64 = std::move(keys_a); out.spare32 = std::move(pos_a);      // the sort's input: free again, same sizes as rowinfo / gen
    if (finish) finish_sort(out, k);
}


// partition digits of the bucket pipeline for n_expect elements: false = the pipeline declines (buckets beyond the LDS sorts)
static bool bucket_digits(int64_t n_expect, int* total_bits_out, int* B1_out, int* B2_out) {
    static const int tb_env = [] { const char* e = vg_dev_getenv("VG_TOTAL_BITS"); return e ? atoi(e) : 0; }();     // developer experiments
    int total_bits = 0; while ((n_expect >> total_bits) > 1024 && total_bits < 22) ++total_bits;
    if (tb_env > 0 && tb_env < total_bits) total_bits = tb_env;
    if ((n_expect >> total_bits) > 4096) return false;
    static const char* b2_env = vg_dev_getenv("VG_B2");          // developer experiments: level-2 bits
    // level 1 takes 11 bits whenever there are two levels: its segment length does not depend on the digit (tiles of
    // 32 768), level 2's grows as its digit shrinks, and 2k - 11 key bits fit the short records up to k = 25
    const int B2 = total_bits > 11 ? (b2_env ? atoi(b2_env) : total_bits - 11) : 0;
    *total_bits_out = total_bits; *B2_out = B2; *B1_out = total_bits - B2;
    return true;
}
static int g_index_path = -1;      // -1 = not read yet; 0 = radix (rocPRIM) path forced; 1 = buckets
static bool index_path_buckets() {
    if (g_index_path < 0) { const char* e = vg_dev_getenv("VG_INDEX_PATH"); g_index_path = (e && !strcmp(e, "radix")) ? 0 : 1; }
    return g_index_path != 0;
}
// 0 = every rank of a RANGE cut scans all bases (k_part_count<.., RANGE>); 1 = sliced scan, the peers emulated by this
// process (vg_set_range_scan: tools/strong_scaling_sim.py, tests); the sharded entry points always exchange when it applies
static int g_range_scan_mode = 0;
bool vg_slice_exchange_applies(const vg_genomes* g, int k, double fraction, int world) {
    (void)k;
    static const bool replicated = [] { const char* e = vg_dev_getenv("VG_RANGE_SCAN"); return e && !strcmp(e, "replicated"); }();      // developer A/B
    if (replicated || world < 2 || world > SX_MAX_WORLD || g_force_subshards > 1 || !range_shards(g, fraction, world)) return false;
    const int64_t P = g->padded_total();
    if (!index_path_buckets() || P < (1 << 16) || P >= (1LL << 32)) return false;
    int total_bits = 0, B1 = 0, B2 = 0;
    return bucket_digits(P, &total_bits, &B1, &B2) && total_bits > 11 && B1 == DIG_BITS;      // the level-1 digit is the shard digit
}

// The count pass of the RANGE shard of rank xs->rank by sliced scan + exchange (k_slice_scan): fills T1s ([super-tile][own
// digit]), wave_mask and wave_cnt exactly as k_part_count<.., RANGE> does; d_kept receives the VALID k-mers of the rank's
// slice of the bases per genome (all digits): the ranks' (kept - duplicates) still add up to the set sizes.
static void sliced_count(int k, const part_src& S, vg_slice_exchange* xs, int st_tiles, int64_t n_st, int nb1, uint32_t* T1s,
                         unsigned long long* wave_mask, uint32_t* wave_cnt, int* d_kept, int n_genomes, hipStream_t s) {
    const int W = xs->world, me = xs->rank;
    const int64_t W_total = S.n >> 6, wps = (int64_t)st_tiles * PT_TILE / 64;                  // 64-position words; per super-tile
    auto st_of = [&](int r) { return n_st * r / W; };
    auto word_of = [&](int r) { return std::min<int64_t>(W_total, st_of(r) * wps); };
    uint32_t dig_lo[SX_MAX_WORLD + 1];
    for (int r = 0; r <= W; ++r) dig_lo[r] = (uint32_t)(((uint64_t)r << DIG_BITS) / (uint64_t)W);
    if ((int)(dig_lo[me + 1] - dig_lo[me]) != nb1) throw vg_error(VG_EINVAL, "internal error: sliced scan and shard digits disagree");
    struct sent { dbuf<unsigned long long> mask; dbuf<uint32_t> T; std::vector<int64_t> m_off, t_off; };
    auto scan = [&](int r, sent& o, int* kept) {
        slice_plan X; memset(&X, 0, sizeof X);
        X.world = W; X.st_lo = st_of(r); X.st_hi = st_of(r + 1); X.word_lo = word_of(r); X.words = word_of(r + 1) - word_of(r);
        const int64_t rows = X.st_hi - X.st_lo;
        o.m_off.assign((size_t)W + 1, 0); o.t_off.assign((size_t)W + 1, 0);
        for (int d = 0; d <= W; ++d) { o.m_off[(size_t)d] = (int64_t)d * X.words * 8; o.t_off[(size_t)d] = rows * (int64_t)dig_lo[d] * 4; }
        o.mask.alloc((size_t)std::max<int64_t>(1, (int64_t)W * X.words)); o.T.alloc((size_t)std::max<int64_t>(1, rows << DIG_BITS));
        for (int d = 0; d < W; ++d) X.t_off[d] = o.t_off[(size_t)d] / 4;
        for (int d = 0; d <= W; ++d) X.dig_lo[d] = dig_lo[d];
        X.mask = o.mask.p; X.T = o.T.p;
        if (rows <= 0) return;
        const size_t lds = (size_t)16 * (PT_PER / 4) * W * 8 * sizeof(uint32_t);
        const int grid = (int)std::min<int64_t>(rows, 512);
        if (k == 25 && !S.A.use_frac) hipLaunchKernelGGL(k_slice_scan<25>, dim3(grid), dim3(PT_THREADS), lds, s, S, X, st_tiles, kept);
        else hipLaunchKernelGGL(k_slice_scan<0>, dim3(grid), dim3(PT_THREADS), lds, s, S, X, st_tiles, kept);
    };
    sent mine; int status = VG_OK; std::string err;
    try { vg_prof_scope ps("kmer_slice_scan", (double)S.n / W * (3.0 / 8.0 + 1.0 / 8.0)); scan(me, mine, d_kept); }
    catch (const vg_error& e) { status = e.code; err = e.what(); }
    if (xs->emulate) {
        if (status != VG_OK) throw vg_error(status, err);
        // one GPU stands in for the world: the peers' slices are scanned here, one after the other, and the blocks of this
        // rank's digit range are copied out of their send buffers (what the all-to-all delivers)
        dbuf<int> scratch_kept((size_t)std::max(1, n_genomes));
        for (int r = 0; r < W; ++r) {
            sent other; const sent* src = &mine;
            if (r != me) { vg_prof_scope ps("emulated_peer_scan", 0); scratch_kept.zero(s); scan(r, other, scratch_kept.p); src = &other; }
            const int64_t mb = src->m_off[(size_t)me + 1] - src->m_off[(size_t)me], tb = src->t_off[(size_t)me + 1] - src->t_off[(size_t)me];
            if (mb) VG_HIP(hipMemcpyAsync(wave_mask + word_of(r), (const char*)src->mask.p + src->m_off[(size_t)me], (size_t)mb, hipMemcpyDeviceToDevice, s));
            if (tb) VG_HIP(hipMemcpyAsync(T1s + st_of(r) * nb1, (const char*)src->T.p + src->t_off[(size_t)me], (size_t)tb, hipMemcpyDeviceToDevice, s));
            if (r != me) VG_HIP(hipStreamSynchronize(s));        // (the peer's buffers go out of scope)
        }
    } else {
        std::vector<int64_t> m_roff((size_t)W + 1), t_roff((size_t)W + 1);
        for (int r = 0; r <= W; ++r) { m_roff[(size_t)r] = word_of(r) * 8; t_roff[(size_t)r] = st_of(r) * (int64_t)nb1 * 4; }
        if (mine.m_off.empty()) { mine.m_off.assign((size_t)W + 1, 0); mine.t_off.assign((size_t)W + 1, 0); }      // (the scan failed before its plan: nothing travels anyway)
        if (status != VG_OK) vg_set_error("%s", err.c_str());
        const vg_xpart parts[2] = { { mine.mask.p, mine.m_off.data(), wave_mask, m_roff.data() }, { mine.T.p, mine.t_off.data(), T1s, t_roff.data() } };
        xs->alltoallv(status, parts, 2);
    }
    if (W_total > 0) hipLaunchKernelGGL(k_mask_popc, dim3(grid_for(W_total)), dim3(256), 0, s, (const unsigned long long*)wave_mask, W_total, wave_cnt);
    VG_HIP(hipStreamSynchronize(s));                                 // the send buffers go out of scope
}

// ---- the bucket pipeline (see the kernels above).  Source: the packed bases themselves (dense) or the kept
// k-mers of a shard / fraction (keys, row numbers).  Fills gen[] and rowinfo[] like k_group_runs; false = the
// input does not suit it (tiny, skewed, a bucket beyond the LDS): the caller takes the general path.
// set by the sub-shard loop (kmer_shared_subshards): the k-mer scan of the NEXT sub-shard, started on the second queue at
// a point where the library queue is idle
static std::function<void()> g_after_extract;
static std::function<void()> g_spgemm_hook;
void vg_set_spgemm_hook(std::function<void()> fn) { g_spgemm_hook = std::move(fn); }
static void run_scan_hook() { if (g_after_extract) { auto hook = std::move(g_after_extract); g_after_extract = nullptr; hook(); } }
// A RANGE shard of the dense source (A.dig_n < 2^11): `ri` receives the kept masks, the row bases and the row -> genome
// map of the pass, n_rows_info becomes the number of kept k-mers, and the row pointers are indexed by row number.
static bool build_index_buckets(vg_genomes* g, int k, bool dense, const kmer_args& A, const uint64_t* keys, const uint32_t* pos, int64_t n_src,
                                const compact_map& cmap_in, int* d_kept, dbuf<uint32_t>& gen, dbuf<uint32_t>& rowinfo, dbuf<uint32_t>& arena,
                                int64_t& n_rows_info, int* d_dups, int64_t* n_valid_out, sorted_index* ri = nullptr, vg_slice_exchange* xs = nullptr) {
    compact_map cmap = cmap_in;
    const bool range = dense && A.dig_n < (1u << DIG_BITS);
    if (range && !ri) throw vg_error(VG_EINVAL, "internal error: a range shard needs its row map");
    hipStream_t s = vg_stream();
    vg_host_mark("buckets: enter");
    pass_timer index_timer(g_time_bucket_pass ? &g_last_bucket_ms : nullptr, s);       // (placement trials: count ... bucket kernels of this pass)
    if (!index_path_buckets() || n_src < (1 << 16) || n_src >= (1LL << 32)) return false;
    // elements the partition will hold (a RANGE shard holds whole buckets of the set's own partition: the digits follow from n_src)
    const int64_t n_expect = dense && !range ? n_src / std::max<uint32_t>(1u, A.n_shards) : n_src;
    int total_bits = 0, B1 = 0, B2 = 0;
    if (!bucket_digits(n_expect, &total_bits, &B1, &B2)) return false;
    const bool big_buckets = (n_expect >> total_bits) > 1024;
    const int levels = total_bits > 11 ? 2 : 1;
    if (range && B1 > DIG_BITS) return false;
    // level-1 buckets of this pass: all 2^B1, or those the RANGE shard's digits fall into
    const int bin_lo = range ? (int)(A.dig_lo >> (DIG_BITS - B1)) : 0;
    const int nb1g = 1 << B1, nb1 = range ? (int)(((A.dig_lo + A.dig_n - 1) >> (DIG_BITS - B1)) - (uint32_t)bin_lo + 1) : nb1g, nb2 = 1 << B2;
    const bool narrow = levels == 2 && 2 * k - total_bits <= 32;      // level-2 output: one key word instead of two
    part_src S; memset(&S, 0, sizeof S);
    S.A = A; S.keys = keys; S.pos = pos; S.n = n_src; S.k2 = 2 * k; S.bin_lo = bin_lo;
    int st_tiles = (int)std::max<int64_t>(1, std::min<int64_t>(dense ? 16 : 8, n_src / ((int64_t)PT_TILE * 2048)));
    if (dense && st_tiles >= 4) st_tiles &= ~3;             // whole 32 768-position tiles for k_part_scatter_dense
    const int64_t n_st = (n_src + (int64_t)st_tiles * PT_TILE - 1) / ((int64_t)st_tiles * PT_TILE);
    const size_t t1n = (size_t)nb1 * (size_t)n_st;
    dbuf<uint32_t> T1s(t1n + 1);                              // counts [super-tile][bucket], scanned in place
    const int n_slabs = (int)((n_st + SC_SLAB - 1) / SC_SLAB);
    dbuf<uint32_t> slab((size_t)n_slabs * nb1), d_off1((size_t)nb1 + 1);
    dbuf<uint32_t> a_rec, b_rec;
    uint32_t n1 = 0; size_t n_cap = 0;                        // records of this pass; capacity its record buffers are sized for
    // The row pointers start as zeros (only the later members of a run get one): 15 GB at 100 k genomes, 2.7 ms of
    // fill.  With room to spare (they otherwise move into the level-1 record buffer once level 2 has read it) they get
    // their own block, cleared on the side queue beside the k-mer kernels of level 1, which are bound by arithmetic.
    hipEvent_t ev_rows_zero = nullptr;
    // (whatever way the function is left: everything that touches the block later is queued on s behind the clear)
    struct ev_guard { hipEvent_t& e; hipStream_t st; ~ev_guard() { if (e) { (void)hipStreamWaitEvent(st, e, 0); (void)hipEventDestroy(e); } } } ev_g{ ev_rows_zero, s };
    // (dense single-pass sets only, whose whole workspace is a fraction of the HBM: with sub-shards of 10^6 contigs the
    // extra 14 GB block pushed the caching allocator into trims and fresh hipMallocs -- 8.2 s per pass instead of 2.7)
    static const bool no_prezero = [] { const char* e = vg_dev_getenv("VG_ROWS_PREZERO"); return e && *e == '0'; }();      // developer A/B
    if (levels == 2 && dense && !no_prezero && !range) {
        // (the device's total memory: asked once -- hipMemGetInfo is a driver round trip on every pass otherwise)
        static const size_t tot = [] { size_t fr0 = 0, t0 = 0; return hipMemGetInfo(&fr0, &t0) == hipSuccess ? t0 : (size_t)0; }();
        if (tot && (size_t)n_rows_info * 4 * 8 <= tot / 2) try {
            { vg_dev_try_scope opportunistic; rowinfo.alloc((size_t)n_rows_info); }
            hipStream_t side = vg_side_stream();
            hipEvent_t ev_s = nullptr;
            VG_HIP(hipEventCreateWithFlags(&ev_s, hipEventDisableTiming));
            VG_HIP(hipEventRecord(ev_s, s));                   // (the block may have just been handed back by work still queued on s)
            VG_HIP(hipStreamWaitEvent(side, ev_s, 0));
            (void)hipEventDestroy(ev_s);
            VG_HIP(hipMemsetAsync(rowinfo.p, 0, (size_t)n_rows_info * sizeof(uint32_t), side));
            VG_HIP(hipEventCreateWithFlags(&ev_rows_zero, hipEventDisableTiming));
            VG_HIP(hipEventRecord(ev_rows_zero, side));
        } catch (...) {
            // (an optimisation only: the row pointers then live in the level-1 record buffer and are cleared in line)
            (void)hipGetLastError(); (void)hipDeviceSynchronize();
            if (ev_rows_zero) { (void)hipEventDestroy(ev_rows_zero); ev_rows_zero = nullptr; }
            rowinfo.release();
        } else (void)hipGetLastError();
    }
    // level-2 units and (dense source, k <= 25 at 2^11 buckets) short level-1 records, see lvl2_tab
    lvl2_tab L2; memset(&L2, 0, sizeof L2);
    const int64_t st_pos = (int64_t)st_tiles * PT_TILE;
    L2.T1s = T1s.p; L2.n_st = n_st; L2.off1 = d_off1.p; L2.nb1 = nb1;
    L2.u_st = (int)std::max<int64_t>(1, std::min<int64_t>(n_st, (65536LL * nb1g) / st_pos));
    L2.n_u = (int)((n_st + L2.u_st - 1) / L2.u_st);
    L2.kr = 2 * k - B1;
    { int sh = 0; while ((1LL << sh) < st_pos) ++sh; L2.st_shift = sh; }
    static const bool long_rec = [] { const char* e = vg_dev_getenv("VG_LEVEL1_RECORDS"); return e && !strcmp(e, "long"); }();
    static const bool old_scatter = [] { const char* e = vg_dev_getenv("VG_DENSE_SCATTER"); return e && !strcmp(e, "staged"); }();
    const bool tile32k = dense && st_tiles % 4 == 0 && B1 <= 12 && !old_scatter;          // k_part_scatter_dense applies
    bool short_rec = levels == 2 && tile32k && narrow && B2 <= 11 && L2.kr + SR_POS_BITS <= 64 && L2.kr - B2 >= 1 &&
                     (st_pos & (st_pos - 1)) == 0 && st_pos <= (1LL << SR_POS_BITS) && !long_rec;
    if (short_rec) {
        const int64_t g_st = (1LL << SR_POS_BITS) / st_pos;                       // super-tiles per position group
        if (L2.u_st <= g_st) L2.g_st = (g_st % L2.u_st == 0) ? 0 : -1;            // a unit inside one group
        else L2.g_st = (L2.u_st % g_st == 0 && L2.u_st / g_st <= 4) ? (int)g_st : -1;
        if (L2.g_st < 0) { short_rec = false; L2.g_st = 0; }
    }
    {
        // (a sliced scan and its exchange have scopes of their own: kmer_slice_scan, exchange)
        const double part_bytes = (double)n_src * (dense ? 2 * 3.0 / 8.0 : 8.0 + 12.0) + (double)n_src * (short_rec ? 8.0 : 12.0);
        std::optional<vg_prof_scope> ps;
        if (!(range && xs)) ps.emplace("kmer_partition", part_bytes);
        const int grid_c = (int)std::min<int64_t>(n_st, 512);
        const bool k25 = dense && k == 25 && !A.use_frac && A.n_shards == 1;          // the default: kernels with k as a constant
        unsigned long long* no_mask = nullptr; uint32_t* no_cnt = nullptr;
        dbuf<uint32_t> wave_cnt;
        const int64_t W = n_src / 64;
        if (range) {
            ri->compact = true;
            ri->wave_mask.alloc((size_t)W + 1); ri->wave_base.alloc((size_t)W + 1); wave_cnt.alloc((size_t)W + 1);
            VG_HIP(hipMemsetAsync(wave_cnt.p + W, 0, sizeof(uint32_t), s));
        }
        if (range && xs) {
            if (levels != 2 || B1 != DIG_BITS) throw vg_error(VG_EINVAL, "internal error: sliced scan without the shard digit as level-1 digit");
            sliced_count(k, S, xs, st_tiles, n_st, nb1, T1s.p, ri->wave_mask.p, wave_cnt.p, d_kept, g->n, s);
            ps.emplace("kmer_partition", part_bytes / xs->world);
        }
        else if (k25 && range) hipLaunchKernelGGL((k_part_count<SRC_DENSE, 25, true>), dim3(grid_c), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles, n_st, T1s.p, d_kept, ri->wave_mask.p, wave_cnt.p);
        else if (k25) hipLaunchKernelGGL((k_part_count<SRC_DENSE, 25>), dim3(grid_c), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles, n_st, T1s.p, d_kept, no_mask, no_cnt);
        else if (dense && range) hipLaunchKernelGGL((k_part_count<SRC_DENSE, 0, true>), dim3(grid_c), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles, n_st, T1s.p, d_kept, ri->wave_mask.p, wave_cnt.p);
        else if (dense) hipLaunchKernelGGL(k_part_count<SRC_DENSE>, dim3(grid_c), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles, n_st, T1s.p, d_kept, no_mask, no_cnt);
        else hipLaunchKernelGGL(k_part_count<SRC_ARRAYS>, dim3(grid_c), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles, n_st, T1s.p, (int*)nullptr, no_mask, no_cnt);
        dbuf<char> scan_tmp;
        if (range) {
            // rows before every 64-position word (the payloads of the level-1 records and the SpGEMM's row ranges)
            size_t tb = 0;
            VG_HIP(rocprim::exclusive_scan(nullptr, tb, wave_cnt.p, ri->wave_base.p, 0u, (size_t)W + 1, rocprim::plus<uint32_t>(), s));
            scan_tmp.alloc(tb);
            VG_HIP(rocprim::exclusive_scan((void*)scan_tmp.p, tb, wave_cnt.p, ri->wave_base.p, 0u, (size_t)W + 1, rocprim::plus<uint32_t>(), s));
            S.wmask = ri->wave_mask.p; S.wbase = ri->wave_base.p; L2.wbase = ri->wave_base.p; L2.W = W;
        }
        // counts -> write offsets, in place (see k_scan_columns_*)
        hipLaunchKernelGGL(k_scan_columns_a, dim3(n_slabs), dim3(PT_THREADS), 0, s, (const uint32_t*)T1s.p, n_st, nb1, slab.p);
        hipLaunchKernelGGL(k_scan_columns_b, dim3(1), dim3(PT_THREADS), 0, s, slab.p, n_slabs, nb1, d_off1.p);
        hipLaunchKernelGGL(k_scan_columns_c, dim3(n_slabs), dim3(PT_THREADS), 0, s, T1s.p, n_st, nb1, (const uint32_t*)slab.p);
        vg_host_mark("buckets: count+scan queued");
        VG_HIP(hipMemcpyAsync(&n1, d_off1.p + nb1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        VG_HIP(hipStreamSynchronize(s));
        vg_host_mark("buckets: count+scan done");
        *n_valid_out = (int64_t)n1; n_cap = (size_t)n1;
        if (range) {
            n_rows_info = std::max<int64_t>((int64_t)n1, 1);
            ri->n_valid = (int64_t)n1;
            ri->cblk.alloc((size_t)(n1 >> CBLK_SHIFT) + 2); ri->cblk.zero(s);
            ri->goff.alloc((size_t)g->n + 1);
            hipLaunchKernelGGL(k_cblk, dim3(grid_for(g->n)), dim3(256), 0, s, (const uint32_t*)ri->wave_base.p, g->d_base_off.p, g->n, ri->cblk.p, ri->goff.p);
            cmap = compact_map{ ri->goff.p, ri->cblk.p };
        }
        if (n1 == 0) {
            // no valid k-mer at all (every record shorter than k, or all N): empty outputs the SpGEMM can read
            if (!ev_rows_zero) { rowinfo.alloc((size_t)n_rows_info); rowinfo.zero(s); }
            gen.alloc(4); gen.zero(s);
            return true;
        }
        // two levels: the level-1 records are dead once level 2 has scattered them, and the genome list + row
        // descriptors are born after that: they take over the same block (48 GB less to allocate at 100 k genomes)
        // (the passes of a RANGE sub-shard loop ask for the same sizes, so that each finds the previous pass's blocks in the
        // allocator's cache: sized for the widest digit range of the loop plus a margin, not for this pass's exact count)
        if (range) n_cap = std::max<size_t>((size_t)n1, (size_t)((double)n_src * A.dig_max / (double)(1u << DIG_BITS) * 1.01) + 65536);
        const size_t rows_cap = range ? n_cap : (size_t)n_rows_info;
        a_rec.alloc(levels == 2 ? std::max((short_rec ? 2 : 3) * n_cap + 8, rows_cap + n_cap + 16) : 3 * n_cap + 8);
        const int grid_s = (int)std::min<int64_t>(n_st, 256);
        static const bool range_dense = [] { const char* e = vg_dev_getenv("VG_RANGE_SCATTER"); return e && !strcmp(e, "dense"); }();      // developer A/B
        static const bool range_32k = [] { const char* e = vg_dev_getenv("VG_RANGE_TILE"); return e && !strcmp(e, "32k"); }();      // developer A/B
        // (two tiles at a time when the shard keeps a sixth of the positions or less: their kept k-mers fit one list)
        const bool tile64k = st_tiles % 8 == 0 && !range_32k && (uint64_t)A.dig_n * 6 <= (1u << DIG_BITS);
        if (tile32k && range && nb1 <= RG_MAXBINS && !range_dense) {
            const int tsh = short_rec ? L2.kr : 0;
            if (tile64k) {
                if (k == 25 && !A.use_frac) hipLaunchKernelGGL((k_part_scatter_range<25, 65536>), dim3(grid_s), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles / 8, n_st, (const uint32_t*)T1s.p, a_rec.p, tsh);
                else hipLaunchKernelGGL((k_part_scatter_range<0, 65536>), dim3(grid_s), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles / 8, n_st, (const uint32_t*)T1s.p, a_rec.p, tsh);
            } else {
                if (k == 25 && !A.use_frac) hipLaunchKernelGGL((k_part_scatter_range<25, 32768>), dim3(grid_s), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles / 4, n_st, (const uint32_t*)T1s.p, a_rec.p, tsh);
                else hipLaunchKernelGGL((k_part_scatter_range<0, 32768>), dim3(grid_s), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles / 4, n_st, (const uint32_t*)T1s.p, a_rec.p, tsh);
            }
        }
        else if (tile32k)
            if (k25 && !range) hipLaunchKernelGGL(k_part_scatter_dense<25>, dim3(grid_s), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles / 4, n_st, (const uint32_t*)T1s.p, a_rec.p,
                                        short_rec ? L2.kr : 0);
            else hipLaunchKernelGGL(k_part_scatter_dense<0>, dim3(grid_s), dim3(PT_THREADS), 0, s, S, B1, nb1, st_tiles / 4, n_st, (const uint32_t*)T1s.p, a_rec.p,
                                    short_rec ? L2.kr : 0);
        else if (dense) hipLaunchKernelGGL((k_part_scatter<SRC_DENSE, 1>), dim3(grid_s), dim3(PT_THREADS), 0, s, S, B1, 0, st_tiles, n_st, (const uint32_t*)T1s.p,
                                      lvl2_tab{}, a_rec.p, -1, nb1);
        else hipLaunchKernelGGL((k_part_scatter<SRC_ARRAYS, 1>), dim3(grid_s), dim3(PT_THREADS), 0, s, S, B1, 0, st_tiles, n_st, (const uint32_t*)T1s.p,
                                lvl2_tab{}, a_rec.p, -1, nb1);
    }
    const int64_t nbk = levels == 1 ? nb1 : (int64_t)nb1 * nb2;
    dbuf<uint32_t> boff((size_t)nbk + 1);
    const uint32_t* f_rec = a_rec.p; int f_stride = 3;
    if (levels == 1) {
        VG_HIP(hipMemcpyAsync(boff.p, d_off1.p, ((size_t)nb1 + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));     // the level-1 buckets are the final ones
    } else {
        // level 2 on units of u_st super-tiles of one level-1 bucket (see lvl2_tab): everything it needs is in the
        // scanned level-1 table, so it is launched right behind level 1
        const int64_t n_units = (int64_t)nb1 * L2.n_u;
        const size_t t2n = (size_t)nb1 * (size_t)nb2 * (size_t)L2.n_u;
        dbuf<uint32_t> T2s(t2n);                              // counts [bucket][unit][b2], scanned in place
        {
            vg_prof_scope ps("kmer_partition2", (double)n1 * (4.0 + (short_rec ? 8.0 : 12.0) + (narrow ? 8.0 : 12.0)));
            const int grid_c2 = (int)std::min<int64_t>(n_units, 512);
            if (short_rec) hipLaunchKernelGGL(k_part_count2<true>, dim3(grid_c2), dim3(PT_THREADS), 0, s, (const uint32_t*)a_rec.p, B1, B2, n_units, L2, T2s.p);
            else hipLaunchKernelGGL(k_part_count2<false>, dim3(grid_c2), dim3(PT_THREADS), 0, s, (const uint32_t*)a_rec.p, B1, B2, n_units, L2, T2s.p);
            hipLaunchKernelGGL(k_scan_units, dim3(nb1), dim3(PT_THREADS), 0, s, T2s.p, L2.n_u, nb2, (const uint32_t*)d_off1.p, nb1, boff.p);
            b_rec.alloc((narrow ? 2 : 3) * n_cap + 8);
            static const bool staged2 = [] { const char* e = vg_dev_getenv("VG_LEVEL2_SCATTER"); return e && !strcmp(e, "staged"); }();
            const int grid_s2 = (int)std::min<int64_t>(n_units, 256);
            if (short_rec)
                hipLaunchKernelGGL(k_part_scatter2_narrow<true>, dim3(grid_s2), dim3(PT_THREADS), 0, s, (const uint32_t*)a_rec.p, B1, B2, n_units,
                                   (const uint32_t*)T2s.p, L2, b_rec.p, total_bits);
            else if (narrow && B2 <= 11 && !staged2)
                hipLaunchKernelGGL(k_part_scatter2_narrow<false>, dim3(grid_s2), dim3(PT_THREADS), 0, s, (const uint32_t*)a_rec.p, B1, B2, n_units,
                                   (const uint32_t*)T2s.p, L2, b_rec.p, total_bits);
            else {
                part_src S2; memset(&S2, 0, sizeof S2);
                S2.rec = a_rec.p; S2.n = (int64_t)n1; S2.k2 = 2 * k;
                hipLaunchKernelGGL((k_part_scatter<SRC_PLANES, 2>), dim3(grid_s2), dim3(PT_THREADS), 0, s, S2, B1, B2, 0, n_units,
                                   (const uint32_t*)T2s.p, L2, b_rec.p, narrow ? total_bits : -1, 0);
            }
        }
        VG_HIP(hipStreamSynchronize(s));                       // the tables and level-1 records go out of scope below
        vg_host_mark("buckets: level 2 done");
        // (the library queue is idle here too.  Starting the scan of the next HASH sub-shard HERE, beside the bucket kernel,
        // instead of in front of the SpGEMM was measured at 10^6 contigs once the SpGEMM had dropped to 16 ms and no longer
        // covered the 52 ms scan: 2 502 against 2 458 ms per step -- the bucket kernel takes 86 ms instead of 55 beside it;
        // the scan's arithmetic is additive wherever it runs.  VG_SUBSHARD_SCAN=early selects it.)
        static const bool early_scan = [] { const char* e = vg_dev_getenv("VG_SUBSHARD_SCAN"); return e && !strcmp(e, "early"); }();
        if (early_scan) run_scan_hook();
        vg_deferred_start();                                  // (the bucket kernel and the SpGEMM are the long waits of the call)
        f_rec = b_rec.p; f_stride = narrow ? 2 : 3;
        arena = std::move(a_rec);
        if (!ev_rows_zero) rowinfo.view(arena.p, (size_t)n_rows_info);
        gen.view(arena.p + (((size_t)n_rows_info + 3) & ~(size_t)3), (size_t)n1 + 4);
    }
    if (gen.n < (size_t)n1 + 4) gen.alloc((size_t)n1 + 4);
    if (rowinfo.n < (size_t)n_rows_info) rowinfo.alloc((size_t)n_rows_info);      // (one level: nothing to take over)
    // ordinary buckets (mean <= 1 024): 256 threads, 9-bit sub-bins.  A bucket beyond the 1 536 entries that variant
    // takes (k-mers shared by dozens of genomes) is queued for the 1 024-thread variant (6 144 entries, 11-bit
    // sub-bins); what that one cannot take either (a k-mer occurring many hundreds of times) goes to k_bucket_big.
    // Every bucket is finished by exactly one of the three: nothing is redone, nothing leaves the own pipeline.
    dbuf<unsigned int> d_nover(2); d_nover.zero(s);
    dbuf<uint32_t> over1((size_t)nbk), over2;
    if (ev_rows_zero) VG_HIP(hipStreamWaitEvent(s, ev_rows_zero, 0));
    else VG_HIP(hipMemsetAsync(rowinfo.p, 0, (size_t)n_rows_info * sizeof(uint32_t), s));
    const int pb = narrow ? 0 : total_bits;
    unsigned int n_over[2] = {0, 0};
#define VG_BUCKET_LAUNCH(NARROW_, THREADS_, SUBBITS_, GRID_, COUNT_, LIST_, OVER_, NOVER_) \
    hipLaunchKernelGGL((k_bucket_runs<NARROW_, THREADS_, SUBBITS_>), dim3(GRID_), dim3(THREADS_), 0, s, f_rec, f_stride, (const uint32_t*)boff.p, (int64_t)(COUNT_), \
                       pb, (const uint32_t*)g->d_blk2g.p, g->align_shift, gen.p, rowinfo.p, cmap, d_dups, (const uint32_t*)(LIST_), (OVER_), (NOVER_))
    {
        vg_prof_scope ps("bucket_sort_runs", (double)n1 * ((narrow ? 8 : 12) + 4 + 8));
        const int grid_b = (int)std::min<int64_t>(nbk, 256 * 16);

        if (big_buckets) { if (narrow) VG_BUCKET_LAUNCH(true, 1024, 11, grid_b, nbk, nullptr, over1.p, d_nover.p); else VG_BUCKET_LAUNCH(false, 1024, 11, grid_b, nbk, nullptr, over1.p, d_nover.p); }
        else { if (narrow) VG_BUCKET_LAUNCH(true, BK_THREADS, 9, grid_b, nbk, nullptr, over1.p, d_nover.p); else VG_BUCKET_LAUNCH(false, BK_THREADS, 9, grid_b, nbk, nullptr, over1.p, d_nover.p); }
        d_nover.download(n_over, 2, s);
        VG_HIP(hipStreamSynchronize(s));
    }
    const uint32_t* big_list = over1.p; unsigned int n_big = n_over[0];
    if (n_big > 0 && !big_buckets) {
        over2.alloc((size_t)n_big);
        vg_prof_scope ps("bucket_sort_runs_wide", 0);
        const int grid_b = (int)std::min<int64_t>(n_big, 256 * 16);
        if (narrow) VG_BUCKET_LAUNCH(true, 1024, 11, grid_b, n_big, over1.p, over2.p, d_nover.p + 1); else VG_BUCKET_LAUNCH(false, 1024, 11, grid_b, n_big, over1.p, over2.p, d_nover.p + 1);
        d_nover.download(n_over, 2, s);
        VG_HIP(hipStreamSynchronize(s));
        big_list = over2.p; n_big = n_over[1];
    }
#undef VG_BUCKET_LAUNCH
    if (n_big > 0) {
        // sizes of the queued buckets -> scratch offsets (the host adds them up)
        dbuf<uint32_t> d_sz(n_big); std::vector<uint32_t> sz(n_big);
        hipLaunchKernelGGL(k_bucket_sizes, dim3(grid_for(n_big)), dim3(256), 0, s, (const uint32_t*)boff.p, big_list, (int64_t)n_big, d_sz.p);
        d_sz.download(sz.data(), n_big, s);
        VG_HIP(hipStreamSynchronize(s));
        std::vector<uint64_t> so(n_big); uint64_t tot = 0;
        for (unsigned int i = 0; i < n_big; ++i) { so[i] = tot; tot += ((uint64_t)sz[i] + 3) & ~3ULL; }
        dbuf<uint64_t> d_so(n_big);
        d_so.upload(so.data(), n_big, s);
        const uint32_t* d_lst_p = big_list;
        dbuf<uint64_t> kA((size_t)tot + 4), kB((size_t)tot + 4); dbuf<uint32_t> pA((size_t)tot + 4), pB((size_t)tot + 4);
        {
            vg_prof_scope ps("bucket_big", (double)tot * 12.0 * 2.0);
            hipLaunchKernelGGL(k_bucket_big, dim3(n_big), dim3(BB_THREADS), 0, s, f_rec, f_stride, (const uint32_t*)boff.p, d_lst_p,
                               (const uint64_t*)d_so.p, pb, kA.p, pA.p, kB.p, pB.p, (const uint32_t*)g->d_blk2g.p, g->align_shift, gen.p, rowinfo.p, cmap, d_dups);
        }
        VG_HIP(hipStreamSynchronize(s));
    }
    return true;
}

// one pass over the k-mers of one shard: per-genome set sizes and (a, b, shared) of every pair
// dev_out != nullptr: the pairs stay in HBM (*dev_out, *dev_n of them) and host_pairs is left empty
static void kmer_shared_pass(vg_genomes* g, int k, double fraction, int shard, int n_shards, uint32_t min_shared,
                             int64_t* set_sizes, std::vector<vg_pair_count>& host_pairs,
                             dbuf<vg_pair_count>* dev_out = nullptr, unsigned long long* dev_n = nullptr, vg_slice_exchange* xs = nullptr) {
    hipStream_t s = vg_stream();
    const int n = g->n;
    sorted_index si;
    const int64_t P = g->padded_total();
    // RANGE shards (sets below 2^32 padded bases, no fraction) keep the dense source: the pass scans the bases, keeps the
    // k-mers of its level-1 buckets and numbers them as rows; everything behind level 1 is a 1/n_shards slice of the whole
    // pass, row pointers included.  HASH shards (larger sets, fractions) materialise their k-mers first (compact source).
    const bool range = range_shards(g, fraction, n_shards);
    const bool dense_src = !(fraction < 1.0) && (n_shards == 1 || range);
    int64_t nv = 0, n_rows_info = 0;
    dbuf<uint32_t> arena;                            // owner of rowinfo / gen when they are windows of one block
    dbuf<uint32_t> rowinfo; dbuf<uint32_t> gen;
    dbuf<int> d_dups((size_t)n); d_dups.zero(s);
    std::vector<int> kept((size_t)n), dups((size_t)n);
    // ---- the bucket pipeline first (own MSD partition + LDS sort); the general radix path when it declines
    bool bucket_ok = false;
    {
        dbuf<int> kept_b((size_t)n); kept_b.zero(s);
        kmer_args A = make_kmer_args(g, k, 1.0, dense_src ? shard : 0, dense_src ? n_shards : 1);      // (the compact source's keys are filtered already)
        if (dense_src) {
            if (P < (1LL << 32)) {
                n_rows_info = P;
                const compact_map none{ nullptr, nullptr };
                bucket_ok = build_index_buckets(g, k, true, A, nullptr, nullptr, P, none, kept_b.p, gen, rowinfo, arena, n_rows_info, d_dups.p, &nv, range ? &si : nullptr, range ? xs : nullptr);
                if (bucket_ok) kept_b.download(kept.data(), (size_t)n, s);
            }
        } else {
            run_extract_sort(g, k, fraction, shard, n_shards, si, false, /*do_sort=*/false);
            nv = si.n_valid; n_rows_info = std::max<int64_t>(nv, 1);
            const compact_map cm{ si.goff.p, si.cblk.p };
            int64_t nv2 = 0;
            bucket_ok = build_index_buckets(g, k, false, A, si.keys.p, si.pos.p, nv, cm, nullptr, gen, rowinfo, arena, n_rows_info, d_dups.p, &nv2);
            if (bucket_ok) si.kept.download(kept.data(), (size_t)n, s);
        }
        if (bucket_ok) { d_dups.download(dups.data(), (size_t)n, s); VG_HIP(hipStreamSynchronize(s)); }
        vg_host_mark("index built");
    }
    if (!bucket_ok) {
    rowinfo.release(); gen.release(); arena.release();
    d_dups.zero(s);
    si = sorted_index();
    run_extract_sort(g, k, fraction, shard, n_shards, si, false);
    nv = si.n_valid;
    n_rows_info = si.compact ? std::max<int64_t>(nv, 1) : P;      // row pointers: per kept k-mer or per base
    // row pointers and genome list live in the sort's input buffers (16 + 16 GB less at 100 k genomes)
    if (2 * si.spare64.n >= (size_t)n_rows_info) rowinfo.view(reinterpret_cast<uint32_t*>(si.spare64.p), (size_t)n_rows_info);
    else rowinfo.alloc((size_t)n_rows_info);
    VG_HIP(hipMemsetAsync(rowinfo.p, 0, (size_t)n_rows_info * sizeof(uint32_t), s));
    const compact_map cmap{ si.compact ? si.goff.p : nullptr, si.compact ? si.cblk.p : nullptr };
    gen = si.spare32.n >= (size_t)std::max<int64_t>(nv, 1) + 4 ? std::move(si.spare32) : dbuf<uint32_t>((size_t)std::max<int64_t>(nv, 1) + 4);
    constexpr unsigned int LONG_CAP = 1u << 16;
    dbuf<int64_t> long_list(LONG_CAP); dbuf<unsigned int> d_nlong(1), d_full(1); d_nlong.zero(s); d_full.zero(s);
    if (nv > 0) {
        vg_prof_scope ps("index_runs", (double)nv * (8 + 4 + 4 + 8));
        hipLaunchKernelGGL(k_group_runs, dim3(grid_for(nv, GS_TILE)), dim3(256), 0, s, si.keys.p, si.pos.p, g->d_blk2g.p, g->align_shift, nv,
                           si.low_bit, gen.p, rowinfo.p, cmap, d_dups.p, long_list.p, d_nlong.p, LONG_CAP);
    }
    unsigned int n_long = 0, need_full = 0;
    d_nlong.download(&n_long, 1, s); si.kept.download(kept.data(), (size_t)n, s); d_dups.download(dups.data(), (size_t)n, s);
    VG_HIP(hipStreamSynchronize(s));
    if (n_long > 0 && n_long <= LONG_CAP) {
        {
            vg_prof_scope ps("index_long_runs", 0);
            hipLaunchKernelGGL(k_long_groups, dim3(n_long), dim3(256), 0, s, si.keys.p, si.pos.p, g->d_blk2g.p, g->align_shift, nv, si.low_bit,
                               long_list.p, gen.p, rowinfo.p, cmap, d_dups.p, d_full.p);
        }
        d_full.download(&need_full, 1, s); d_dups.download(dups.data(), (size_t)n, s);
        VG_HIP(hipStreamSynchronize(s));
    }
    if (need_full || n_long > LONG_CAP) {
        // several frequent k-mers share a prefix group (or too many long groups): sort on all bits, general run pass
        rowinfo.zero(s); d_dups.zero(s);
        finish_sort(si, k);
        {
            vg_prof_scope ps("index_runs_general", (double)nv * (8 + 4 + 4 + 8));
            hipLaunchKernelGGL(k_runs, dim3(grid_for((nv + 3) / 4)), dim3(256), 0, s, si.keys.p, si.pos.p, g->d_blk2g.p, g->align_shift, nv,
                               gen.p, rowinfo.p, cmap, d_dups.p);
        }
        d_dups.download(dups.data(), (size_t)n, s);
        VG_HIP(hipStreamSynchronize(s));
    }
    }
    const bool compact_rows = bucket_ok ? (!dense_src || range) : si.compact;
    const uint32_t* wbase = compact_rows ? si.wave_base.p : nullptr;
    for (int i = 0; i < n; ++i) set_sizes[i] = (int64_t)kept[i] - dups[i];
    si.keys.release();
    // (the library queue is idle here -- the index stage ended with a synchronisation --: the sub-shard loop starts the
    // k-mer scan of the NEXT sub-shard on the second queue, beside this sub-shard's SpGEMM: arithmetic beside random
    // reads.  Started earlier, beside the partition kernels, the scan only took their CUs: 2.70 against 2.74 s.)
    run_scan_hook();
    if (g_spgemm_hook) { auto hook = std::move(g_spgemm_hook); g_spgemm_hook = nullptr; hook(); }      // (developer experiment: work queued beside the SpGEMM)
    // SpGEMM with a growing output buffer
    dbuf<unsigned long long> d_cursor(1);
    dbuf<uint32_t> d_over((size_t)n), d_nover(1);
    unsigned long long cap = std::max<unsigned long long>(1u << 20, (unsigned long long)n * 16);
    // Rows of a few thousand k-mers (10^6 contigs cut into sub-shards: 3 600 per row and pass) are one trip of a
    // 256-thread workgroup: a chain of dependent round trips with eight workgroups per CU to cover it.  Those rows go to
    // ONE-WAVE workgroups with a 512-slot table (26 per CU); a row that outgrows the table joins the overflow list.
    constexpr int64_t SMALL_ROW = 8192;
    dbuf<uint32_t> d_small, d_large; int n_small = 0, n_large = 0;
    if (n >= (1 << 16)) {
        std::vector<uint32_t> small_rows, large_rows;
        // rows of a genome in this pass: its kept k-mers (a sliced scan counted the k-mers of its slice of the BASES instead:
        // the rows are then read off the row map)
        std::vector<uint32_t> goff_h;
        if (xs && range && bucket_ok) { goff_h.resize((size_t)n + 1); si.goff.download(goff_h.data(), (size_t)n + 1, s); VG_HIP(hipStreamSynchronize(s)); }
        auto rows_of = [&](int i) { return !goff_h.empty() ? (int64_t)(goff_h[(size_t)i + 1] - goff_h[(size_t)i]) : compact_rows ? (int64_t)kept[(size_t)i] : g->len[(size_t)i]; };
        for (int i = 0; i < n; ++i) (rows_of(i) <= SMALL_ROW ? small_rows : large_rows).push_back((uint32_t)i);
        if (small_rows.size() * 2 >= (size_t)n) {
            n_small = (int)small_rows.size(); n_large = (int)large_rows.size();
            d_small.alloc(small_rows.size()); d_small.upload(small_rows.data(), small_rows.size(), s);
            if (n_large) { d_large.alloc(large_rows.size()); d_large.upload(large_rows.data(), large_rows.size(), s); }
        }
    }
    for (;;) {
        dbuf<vg_pair_count> d_out((size_t)cap);
        d_cursor.zero(s); d_nover.zero(s);
        {
            vg_prof_scope ps("spgemm_rows", (double)n_rows_info * 4.0);
            if (n_small) {
                if (n_large) hipLaunchKernelGGL(k_spgemm<11>, dim3((n_large + 7) / 8 * 8), dim3(256), 0, s, rowinfo.p, gen.p, (uint64_t)gen.n, g->d_base_off.p, g->d_len.p, wbase, n,
                                                min_shared, (const uint32_t*)d_large.p, n_large, d_out.p, d_cursor.p, cap, d_over.p, d_nover.p);
                hipLaunchKernelGGL((k_spgemm<9, true>), dim3((n_small + 7) / 8 * 8), dim3(64), 0, s, rowinfo.p, gen.p, (uint64_t)gen.n, g->d_base_off.p, g->d_len.p, wbase, n,
                                   min_shared, (const uint32_t*)d_small.p, n_small, d_out.p, d_cursor.p, cap, d_over.p, d_nover.p);
            } else
            hipLaunchKernelGGL(k_spgemm<11>, dim3((n + 7) / 8 * 8), dim3(256), 0, s, rowinfo.p, gen.p, (uint64_t)gen.n, g->d_base_off.p, g->d_len.p, wbase, n,
                               min_shared, (const uint32_t*)nullptr, n, d_out.p, d_cursor.p, cap, d_over.p, d_nover.p);
        }
        // one round trip in the common case: overflow count, pair count and the first pairs together
        constexpr size_t EAGER = 1 << 16;
        uint32_t nover = 0; unsigned long long produced = 0;
        host_pairs.resize(dev_out ? 0 : std::min<size_t>(EAGER, (size_t)cap));
        d_nover.download(&nover, 1, s); d_cursor.download(&produced, 1, s);
        if (!dev_out) d_out.download(host_pairs.data(), host_pairs.size(), s);
        VG_HIP(hipStreamSynchronize(s));
        vg_host_mark("spgemm done");
        if (dev_out && nover == 0 && produced <= cap) { *dev_out = std::move(d_out); *dev_n = produced; break; }
        if (!dev_out && nover == 0 && produced <= host_pairs.size()) { host_pairs.resize((size_t)produced); break; }
        if (nover > 0) {
            // second try with the 64 KiB table for the rows that overflowed the small one
            dbuf<uint32_t> d_rows2(nover), d_over2(nover);
            VG_HIP(hipMemcpyAsync(d_rows2.p, d_over.p, sizeof(uint32_t) * nover, hipMemcpyDeviceToDevice, s));
            d_nover.zero(s);
            {
                vg_prof_scope ps("spgemm_rows_wide", 0);
                hipLaunchKernelGGL(k_spgemm<13>, dim3((nover + 7) / 8 * 8), dim3(256), 0, s, rowinfo.p, gen.p, (uint64_t)gen.n, g->d_base_off.p, g->d_len.p, wbase, n,
                                   min_shared, (const uint32_t*)d_rows2.p, (int)nover, d_out.p, d_cursor.p, cap, d_over2.p, d_nover.p);
            }
            VG_HIP(hipMemcpyAsync(d_over.p, d_over2.p, sizeof(uint32_t) * nover, hipMemcpyDeviceToDevice, s));
            d_nover.download(&nover, 1, s);
            VG_HIP(hipStreamSynchronize(s));
        }
        if (nover > 0) {
            // dense fallback, a few rows at a time
            std::vector<uint32_t> rows(nover); d_over.download(rows.data(), nover, s); VG_HIP(hipStreamSynchronize(s));
            const uint32_t batch = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(nover, (int64_t)(1u << 28) / std::max(n, 1)));
            dbuf<uint32_t> dense((size_t)batch * n), d_rows(batch);
            for (uint32_t o = 0; o < nover; o += batch) {
                uint32_t nb = std::min(batch, nover - o);
                dense.zero(s);
                d_rows.upload(rows.data() + o, nb, s);
                vg_prof_scope ps("spgemm_dense_rows", 0);
                hipLaunchKernelGGL(k_spgemm_dense, dim3(nb), dim3(256), 0, s, rowinfo.p, gen.p, (uint64_t)gen.n, g->d_base_off.p,
                                   g->d_len.p, wbase, n, min_shared, d_rows.p, dense.p, d_out.p, d_cursor.p, cap);
                VG_HIP(hipStreamSynchronize(s));
            }
        }
        d_cursor.download(&produced, 1, s);
        VG_HIP(hipStreamSynchronize(s));
        if (produced <= cap) {
            if (dev_out) { *dev_out = std::move(d_out); *dev_n = produced; break; }
            host_pairs.resize((size_t)produced);
            if (produced) d_out.download(host_pairs.data(), (size_t)produced, s);
            VG_HIP(hipStreamSynchronize(s));
            break;
        }
        cap = produced + produced / 8 + 1024;     // rerun with a buffer that fits
    }
}

// sum of partial pair records (the sub-shards of one call): device radix sort on (a << 32 | b) + reduce by key
namespace {
__global__ void k_pair_keys(const vg_pair_count* __restrict__ rec, int64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const vg_pair_count r = rec[i]; keys[i] = ((uint64_t)r.a << 32) | r.b; vals[i] = r.shared;
    }
}
}
namespace {
__global__ void k_pair_keep(const uint32_t* __restrict__ sums, int64_t n, uint32_t min_shared, uint32_t* __restrict__ keep) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) keep[i] = sums[i] >= min_shared ? 1u : 0u;
}
__global__ void k_pair_place(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ sums, const uint32_t* __restrict__ keep,
                             const uint32_t* __restrict__ at, int64_t n, vg_pair_count* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!keep[i]) continue;
        vg_pair_count r; r.a = (uint32_t)(keys[i] >> 32); r.b = (uint32_t)keys[i]; r.shared = sums[i];
        out[at[i]] = r;
    }
}
}
// The partial lists of the sub-shards (device memory: they never travel) -> the list of pairs whose SUM reaches
// min_shared, ascending on (a, b), in HBM: keys gathered into one array, sorted, reduced by key, and the kept sums
// placed through a prefix sum of their flags (so the order is that of the keys: no host sort of millions of records).
static void sum_partial_pairs(const std::vector<dbuf<vg_pair_count>>& parts, const std::vector<unsigned long long>& counts, uint32_t min_shared,
                              dbuf<vg_pair_count>& out, unsigned long long* n_out) {
    *n_out = 0;
    size_t n = 0; for (unsigned long long c : counts) n += (size_t)c;
    if (!n) return;
    if (n >= (1ull << 32)) throw vg_error(VG_EOVERFLOW, "more than 2^32 partial pair records in one call");
    hipStream_t s = vg_stream();
    dbuf<uint64_t> k1(n), k2(n), uk(n); dbuf<uint32_t> v1(n), v2(n), us(n); dbuf<unsigned long long> d_nu(1);
    size_t off = 0;
    for (size_t t = 0; t < parts.size(); ++t) {
        if (!counts[t]) continue;
        hipLaunchKernelGGL(k_pair_keys, dim3(grid_for((int64_t)counts[t])), dim3(256), 0, s, (const vg_pair_count*)parts[t].p, (int64_t)counts[t], k1.p + off, v1.p + off);
        off += (size_t)counts[t];
    }
    size_t tb = 0, tb2 = 0, tb3 = 0;
    VG_HIP(rocprim::radix_sort_pairs(nullptr, tb, k1.p, k2.p, v1.p, v2.p, n, 0u, 64u, s));
    VG_HIP(rocprim::reduce_by_key(nullptr, tb2, k2.p, v2.p, n, uk.p, us.p, d_nu.p, rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s));
    VG_HIP(rocprim::exclusive_scan(nullptr, tb3, v1.p, v2.p, 0u, n, rocprim::plus<uint32_t>(), s));
    dbuf<char> tmp(std::max(tb, std::max(tb2, tb3)));
    VG_HIP(rocprim::radix_sort_pairs((void*)tmp.p, tb, k1.p, k2.p, v1.p, v2.p, n, 0u, 64u, s));
    VG_HIP(rocprim::reduce_by_key((void*)tmp.p, tb2, k2.p, v2.p, n, uk.p, us.p, d_nu.p, rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s));
    unsigned long long nu = 0; d_nu.download(&nu, 1, s); VG_HIP(hipStreamSynchronize(s));
    if (!nu) return;
    // keep flags in v1, their exclusive prefix sum in v2 (both free after the sort)
    hipLaunchKernelGGL(k_pair_keep, dim3(grid_for((int64_t)nu)), dim3(256), 0, s, (const uint32_t*)us.p, (int64_t)nu, min_shared, v1.p);
    VG_HIP(rocprim::exclusive_scan((void*)tmp.p, tb3, v1.p, v2.p, 0u, (size_t)nu, rocprim::plus<uint32_t>(), s));
    uint32_t last_at = 0, last_keep = 0;
    VG_HIP(hipMemcpyAsync(&last_at, v2.p + (nu - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    VG_HIP(hipMemcpyAsync(&last_keep, v1.p + (nu - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    VG_HIP(hipStreamSynchronize(s));
    const unsigned long long no = (unsigned long long)last_at + last_keep;
    out.alloc((size_t)std::max<unsigned long long>(no, 1));
    if (no) hipLaunchKernelGGL(k_pair_place, dim3(grid_for((int64_t)nu)), dim3(256), 0, s, (const uint64_t*)uk.p, (const uint32_t*)us.p, (const uint32_t*)v1.p,
                               (const uint32_t*)v2.p, (int64_t)nu, out.p);
    VG_HIP(hipStreamSynchronize(s));                           // the scratch buffers go out of scope
    *n_out = no;
}
// the sub-shard loop of one call: every pass leaves its partial list in HBM
static void kmer_shared_subshards(vg_genomes* g, int k, double fraction, int shard, int n_shards, int sub, uint32_t min_shared,
                                  int64_t* set_sizes, dbuf<vg_pair_count>& out, unsigned long long* n_out) {
    const int n = g->n;
    std::vector<int64_t> part((size_t)n);
    for (int i = 0; i < n; ++i) set_sizes[i] = 0;
    std::vector<dbuf<vg_pair_count>> parts((size_t)sub); std::vector<unsigned long long> counts((size_t)sub, 0ULL);
    std::vector<vg_pair_count> none;
    static const bool no_overlap = [] { const char* e = vg_dev_getenv("VG_SUBSHARD_OVERLAP"); return e && *e == '0'; }();      // developer A/B
    struct hook_guard { ~hook_guard() { g_after_extract = nullptr; g_precount.drop(); g_pass_mask = pass_mask(); } } hg;
    // HASH sub-shards: ONE scan of the bases leaves the kept masks of all passes (k_multi_mask); a pass then computes the
    // k-mers of its kept positions only.  (sub x P / 8 bytes of masks -- 22 GB at 10^6 contigs -- replace two sets of
    // staging buffers of P / 256 x ~80 x 8 bytes each; VG_SUBSHARD_SCAN=each is the scan per pass of round 4)
    static const bool scan_each = [] { const char* e = vg_dev_getenv("VG_SUBSHARD_SCAN"); return e && !strcmp(e, "each"); }();
    dbuf<unsigned long long> all_masks;
    const int64_t Wm = g->padded_total() / 64 + 1;
    const bool multi = !scan_each && sub >= 2 && sub <= MM_MAX_SUB && !range_shards(g, fraction, n_shards * sub);
    if (multi) {
        int rc = vg_genomes_to_device(g); if (rc) throw vg_error(rc, vg_last_error());
        hipStream_t s = vg_stream();
        all_masks.alloc((size_t)sub * (size_t)Wm);
        const kmer_args A = make_kmer_args(g, k, fraction, 0, 1);
        const int64_t P = g->padded_total();
        const size_t lds = (size_t)4 * sub * 8 * sizeof(uint32_t);
        vg_prof_scope ps("kmer_multi_mask", (double)P * (3.0 / 8.0 + sub / 8.0));
        if (k == 25 && !A.use_frac) hipLaunchKernelGGL(k_multi_mask<25>, dim3(grid_for((P + 255) / 4)), dim3(256), lds, s, A, (uint32_t)(shard * sub), (uint32_t)sub, (uint32_t)(n_shards * sub), all_masks.p, Wm);
        else hipLaunchKernelGGL(k_multi_mask<0>, dim3(grid_for((P + 255) / 4)), dim3(256), lds, s, A, (uint32_t)(shard * sub), (uint32_t)sub, (uint32_t)(n_shards * sub), all_masks.p, Wm);
    }
    for (int t = 0; t < sub; ++t) {
        g_after_extract = nullptr;
        g_pass_mask = pass_mask();
        if (multi) { g_pass_mask.g = g; g_pass_mask.k = k; g_pass_mask.shard = shard * sub + t; g_pass_mask.n_shards = n_shards * sub; g_pass_mask.mask = all_masks.p + (size_t)t * (size_t)Wm; }
        if (!multi && t + 1 < sub && !(fraction < 1.0) && !no_overlap && !range_shards(g, fraction, n_shards * sub))      // (HASH shards: the compact source scans first)
            g_after_extract = [=] {
                // (an optimisation only: without room for the second set of scan buffers the next sub-shard scans in line)
                try { vg_dev_try_scope opportunistic; launch_precount(g, k, shard * sub + t + 1, n_shards * sub); } catch (...) { (void)hipGetLastError(); g_precount.drop(); }
            };
        kmer_shared_pass(g, k, fraction, shard * sub + t, n_shards * sub, 1u, part.data(), none, &parts[(size_t)t], &counts[(size_t)t]);
        for (int i = 0; i < n; ++i) set_sizes[i] += part[i];
    }
    vg_host_mark("sub-shards done");
    sum_partial_pairs(parts, counts, min_shared, out, n_out);
}

// Placement trials (below): how many placements of the workspace the first dense pass of a long-lived process may try.
// 1 = none, the default: an embedder opts in (vg_set_placement_trials); the developer switch VG_PLACEMENT_TRIALS overrides.
static std::mutex g_trials_mu;
static int g_placement_trials = 1;
static std::map<std::pair<int64_t, int>, int> g_trials_done;       // (padded bases, k) -> tried (under g_trials_mu)
static int placement_trials_now() {
    static const int env = [] { const char* e = vg_dev_getenv("VG_PLACEMENT_TRIALS"); return e && *e ? atoi(e) : 0; }();
    std::lock_guard<std::mutex> lk(g_trials_mu);
    return env > 0 ? env : g_placement_trials;
}
extern "C" void vg_set_placement_trials(int n) {
    std::lock_guard<std::mutex> lk(g_trials_mu);
    g_placement_trials = n < 1 ? 1 : (n > 8 ? 8 : n);
    g_trials_done.clear();                                          // (a caller who asks again gets trials again)
}
extern "C" int vg_kmer_shared(vg_genomes* g, int k, double fraction, int shard, int n_shards, uint32_t min_shared,
                              int64_t* set_sizes, vg_pair_count** pairs, int64_t* n_pairs) {
    VG_API_BEGIN
    if (!g || !set_sizes || !pairs || !n_pairs) throw vg_error(VG_EINVAL, "vg_kmer_shared: null argument");
    if (k < 8 || k > 31) throw vg_error(VG_EINVAL, "k out of range (8..31)");
    if (n_shards < 1 || shard < 0 || shard >= n_shards) throw vg_error(VG_EINVAL, "bad shard");
    if (!(fraction > 0.0) || fraction > 1.0) throw vg_error(VG_EINVAL, "fraction must be in (0,1]");
    vg_host_mark("vg_kmer_shared: enter");
    vg_require_device();
    int rc = vg_genomes_to_device(g); if (rc) return rc;
    *pairs = nullptr; *n_pairs = 0;
    const int n = g->n;
    if (n == 0) return VG_OK;
    // Sets beyond the 32-bit row numbering of one pass (2^32 padded bases dense, ~2^31 kept k-mers per
    // shard) are cut into sub-shards of this shard's k-mer range; partial counts of a pair add up.
    const int64_t P = g->padded_total();
    int64_t real_bases = 0; for (int i = 0; i < g->n; ++i) real_bases += g->len[(size_t)i];
    const double expect = (double)real_bases * fraction / n_shards;        // k-mers kept by this shard, at most (padding positions hold none)
    const bool dense = fraction >= 1.0 && n_shards == 1;
    int sub = 1;
    static const int env_sub = [] { const char* e = vg_dev_getenv("VG_SUBSHARDS"); return e ? atoi(e) : 0; }();      // developer experiments
    if (g_force_subshards > 0) sub = g_force_subshards;
    else if (env_sub > 0) sub = env_sub;
    else if (dense ? P >= (1LL << 32) : expect >= SUB_PASS_START) sub = std::max(2, (int)std::ceil(expect / SUB_PASS_KMERS));      // row numbers of one pass are 32 bits
    else if (dense && vg_one_shot()) {
        // a cold one-shot call (the CLI): RANGE sub-shards under a workspace budget -- 16 bytes of records per kept k-mer
        // and pass.  Eight passes over 100 k genomes cost eight scans of the bases more than one pass (tens of ms) and
        // touch 8 GB of device memory instead of 66 GB: what a cold process may wait for is the first use of memory
        // (25-32 ms per GiB when the driver still has to wipe it, vg_core.cpp).  VG_WORKSPACE_GB sets the budget.
        static const double ws_gb = [] { const char* e = getenv("VG_WORKSPACE_GB"); const double v = e ? atof(e) : 0.0; return v > 0.01 ? v : 8.0; }();
        sub = (int)std::min(64.0, std::ceil(16.0 * (double)P / (ws_gb * 1073741824.0)));
    }
    if (sub < 1) sub = 1;
    std::vector<vg_pair_count> acc;
    if (sub == 1) {
        vg_slice_exchange xs; xs.rank = shard; xs.world = n_shards; xs.emulate = true;
        const bool sliced = g_range_scan_mode == 1 && vg_slice_exchange_applies(g, k, fraction, n_shards);
        kmer_shared_pass(g, k, fraction, shard, n_shards, min_shared, set_sizes, acc, nullptr, nullptr, sliced ? &xs : nullptr);
        // Placement trials.  Where the driver puts the workspace decides between two states of the scattering kernels (the
        // bucket kernel 39 against 44 ms at 100 k genomes: same virtual addresses, same requests, same UTCL1 misses, 1.3-2 x
        // the translation-in-flight and DRAM-credit stalls -- profiles/r05_placement_states.md), and it stays for the
        // life of the blocks.  The FIRST whole pass of a long-lived process over a large set therefore tries up to three
        // placements -- the cached blocks of the previous one set aside, the pass repeated on fresh ones -- and keeps the
        // workspace whose index stage (level-1 count ... bucket kernel) was fastest; the results of the passes are the same, the caller gets the first's.
        // OPT-IN (vg_set_placement_trials; bench.py asks for 4 and says so in its line): an embedder's first call does no hidden
        // extra passes.  The caller's result is already in `acc`: whatever happens in a trial pass -- out of memory beside the
        // parked blocks, a HIP error -- is swallowed here, the parked workspace comes back and the call returns what it computed.
        const int max_trials = placement_trials_now();
        size_t fr = 0, tot = 0;
        bool first_time = false;
        if (dense && !vg_one_shot() && max_trials > 1 && P >= (1LL << 30)) {
            std::lock_guard<std::mutex> lk(g_trials_mu);
            first_time = !g_trials_done[{ P, k }]++;
        }
        if (first_time && hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 2 * vg_dev_cached_bytes() + (8ULL << 30)) {
            struct timing_on { timing_on() { g_time_bucket_pass = true; } ~timing_on() { g_time_bucket_pass = false; } } on;
            std::vector<int64_t> sz2((size_t)n); std::vector<vg_pair_count> acc2;
            float best = 0.f;
            try {
                kmer_shared_pass(g, k, fraction, shard, n_shards, min_shared, sz2.data(), acc2);       // (the first pass paid for the allocations: time this placement on a second one)
                best = g_last_bucket_ms;
            } catch (...) { (void)hipGetLastError(); best = 0.f; }
            for (int trial = 1; trial < max_trials && best > 0.f; ++trial) {
                const double w0 = vg_alloc_wait_ms();
                vg_dev_park_cache();
                bool keep_new = false;
                try {
                    kmer_shared_pass(g, k, fraction, shard, n_shards, min_shared, sz2.data(), acc2);
                    kmer_shared_pass(g, k, fraction, shard, n_shards, min_shared, sz2.data(), acc2);
                    keep_new = g_last_bucket_ms > 0.f && g_last_bucket_ms < 0.97f * best;
                    if (getenv("VG_ALLOC_TRACE")) fprintf(stderr, "[vg placement] trial %d: index stage %.2f ms against %.2f ms -> %s\n", trial, g_last_bucket_ms, best, keep_new ? "kept" : "dropped");
                    if (keep_new) best = g_last_bucket_ms;
                } catch (...) { (void)hipGetLastError(); keep_new = false; }
                vg_dev_unpark(!keep_new);
                if (vg_alloc_wait_ms() - w0 > 500.0) break;          // (memory the driver still has to wipe: a trial costs seconds here)
            }
        }
    } else {
        // partial (a, b, count) records of every sub-shard stay in HBM and are summed ONCE there: sort on (a, b), reduce
        // by key, threshold on the sum
        dbuf<vg_pair_count> d_sum; unsigned long long n_sum = 0;
        // (HASH sub-shards are sized on an expectation: a set whose repeated k-mers crowd one sub-shard beyond the 32-bit row
        // numbers of a pass is cut finer instead of failing)
        for (int attempt = 0;; ++attempt) {
            try { kmer_shared_subshards(g, k, fraction, shard, n_shards, sub, min_shared, set_sizes, d_sum, &n_sum); break; }
            catch (const vg_error& e) { if (e.code != VG_EOVERFLOW || attempt >= 3 || g_force_subshards > 0) throw; ++sub; d_sum.release(); }
        }
        acc.resize((size_t)n_sum);
        if (n_sum) { d_sum.download(acc.data(), (size_t)n_sum, vg_stream()); VG_HIP(hipStreamSynchronize(vg_stream())); }
    }
    vg_host_mark("pass done");
    vg_pair_count* outp = (vg_pair_count*)malloc(sizeof(vg_pair_count) * std::max<size_t>(1, acc.size()));
    if (!outp) throw vg_error(VG_ENOMEM, "out of host memory");
    if (!acc.empty()) memcpy(outp, acc.data(), sizeof(vg_pair_count) * acc.size());
    *pairs = outp; *n_pairs = (int64_t)acc.size();
    vg_host_mark("vg_kmer_shared: return");
    VG_API_END
}

// how a rank of an n_shards-way call cuts the k-mers (1 = RANGE, 2 = HASH): the two do not tile the key space together, so
// the ranks of a sharded call compare notes BEFORE anything is exchanged (a per-process knob -- vg_set_subshards,
// VG_RANGE_SCAN, VG_INDEX_PATH -- can differ between processes).  A pure function of the set and the process's knobs.
int vg_kmer_shard_mode(const vg_genomes* g, double fraction, int n_shards) {
    const int64_t P = g->padded_total();
    int64_t real_bases = 0; for (int i = 0; i < g->n; ++i) real_bases += g->len[(size_t)i];
    const double expect = (double)real_bases * fraction / n_shards;
    const bool dense = fraction >= 1.0 && n_shards == 1;
    const bool one_pass = g_force_subshards <= 1 && !(dense ? P >= (1LL << 32) : expect >= SUB_PASS_START);
    const int sub_planned = one_pass ? 1 : std::max(2, g_force_subshards > 1 ? g_force_subshards : (int)std::ceil(expect / SUB_PASS_KMERS));
    return range_shards(g, fraction, n_shards * sub_planned) ? 1 : 2;
}
// internal (vg_dist.hip): one shard's pairs left in HBM (sub-shards included)
void vg_kmer_shared_device(vg_genomes* g, int k, double fraction, int shard, int n_shards, uint32_t min_shared,
                           int64_t* set_sizes, dbuf<vg_pair_count>& pairs, int64_t* n_pairs, vg_slice_exchange* xs, int* mode_out) {
    if (mode_out) *mode_out = 0;
    vg_require_device();
    int rc = vg_genomes_to_device(g); if (rc) throw vg_error(rc, vg_last_error());
    *n_pairs = 0;
    if (g->n == 0) return;
    const int64_t P = g->padded_total();
    int64_t real_bases = 0; for (int i = 0; i < g->n; ++i) real_bases += g->len[(size_t)i];
    const double expect = (double)real_bases * fraction / n_shards;
    const bool dense = fraction >= 1.0 && n_shards == 1;
    const bool one_pass = g_force_subshards <= 1 && !(dense ? P >= (1LL << 32) : expect >= SUB_PASS_START);
    if (mode_out) *mode_out = vg_kmer_shard_mode(g, fraction, n_shards);
    hipStream_t s = vg_stream();
    if (one_pass) {
        std::vector<vg_pair_count> none; unsigned long long n = 0;
        kmer_shared_pass(g, k, fraction, shard, n_shards, min_shared, set_sizes, none, &pairs, &n, xs);
        *n_pairs = (int64_t)n;
        return;
    }
    int sub = g_force_subshards > 1 ? g_force_subshards : (int)std::ceil(expect / SUB_PASS_KMERS);
    if (sub < 2) sub = 2;
    unsigned long long n = 0;
    for (int attempt = 0;; ++attempt) {
        try { kmer_shared_subshards(g, k, fraction, shard, n_shards, sub, min_shared, set_sizes, pairs, &n); break; }
        catch (const vg_error& e) { if (e.code != VG_EOVERFLOW || attempt >= 3 || g_force_subshards > 1) throw; ++sub; pairs.release(); }
    }
    if (!pairs.p) pairs.alloc(1);
    *n_pairs = (int64_t)n;
}

// developer/test knob: force the sub-shard loop on small inputs (0 = automatic)
static_assert(sizeof(vg_pair_count) == 12, "pair record layout");
extern "C" void vg_set_subshards(int n) { g_force_subshards = n; }
extern "C" void vg_set_range_scan(int mode) { g_range_scan_mode = mode == 1 ? 1 : 0; }

extern "C" int vg_kmer_set(vg_genomes* g, int idx, int k, double fraction, uint64_t** out, int64_t* n_out) {
    VG_API_BEGIN
    if (!g || !out || !n_out || idx < 0 || idx >= g->n) throw vg_error(VG_EINVAL, "vg_kmer_set: bad argument");
    if (k < 8 || k > 31) throw vg_error(VG_EINVAL, "k out of range (8..31)");
    vg_require_device();
    int rc = vg_genomes_to_device(g); if (rc) return rc;
    hipStream_t s = vg_stream();
    sorted_index si;
    run_extract_sort(g, k, fraction, 0, 1, si);
    // host-side filter of one genome's keys out of the sorted index (test-only entry point)
    std::vector<uint64_t> keys((size_t)si.n_valid); std::vector<uint32_t> pos((size_t)si.n_valid);
    if (si.n_valid) { si.keys.download(keys.data(), keys.size(), s); si.pos.download(pos.data(), pos.size(), s); }
    VG_HIP(hipStreamSynchronize(s));
    std::vector<uint64_t> mine;
    int64_t lo = g->base_off[idx], hi = g->base_off[idx + 1];
    if (si.compact) {                                          // payload = compact index: the genome's row range
        uint32_t c[2];
        VG_HIP(hipMemcpyAsync(&c[0], si.wave_base.p + (lo >> 6), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        VG_HIP(hipMemcpyAsync(&c[1], si.wave_base.p + (hi >> 6), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        VG_HIP(hipStreamSynchronize(s));
        lo = c[0]; hi = c[1];
    }
    for (size_t i = 0; i < keys.size(); ++i)
        if ((int64_t)pos[i] >= lo && (int64_t)pos[i] < hi) {
            // the key's two halves are the bit planes of the canonical strand: back to the number the caller knows (2-bit
            // codes, first base most significant, the smaller strand)
            const uint64_t hl = unscramble_key(keys[i], k), H = hl >> k, L = hl & ((1ULL << k) - 1ULL);
            uint64_t x = 0;
            for (int b = 0; b < k; ++b) x |= ((((H >> b) & 1ULL) << 1) | ((L >> b) & 1ULL)) << (2 * b);
            mine.push_back(cano_codes_of(x, k));
        }
    std::sort(mine.begin(), mine.end());
    mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
    uint64_t* o = (uint64_t*)malloc(sizeof(uint64_t) * std::max<size_t>(1, mine.size()));
    if (!o) throw vg_error(VG_ENOMEM, "out of host memory");
    if (!mine.empty()) memcpy(o, mine.data(), sizeof(uint64_t) * mine.size());
    *out = o; *n_out = (int64_t)mine.size();
    VG_API_END
}

// One empty launch: the first launch out of a translation unit loads its code object onto the device (tens of ms for
// this one).  The whole-stage calls do it on their warm-up thread while the FASTA is parsed (vg_api.cpp).
namespace { __global__ void k_warm_prefilter() {} }
void vg_warm_prefilter(hipStream_t s) { hipLaunchKernelGGL(k_warm_prefilter, dim3(1), dim3(64), 0, s); }
