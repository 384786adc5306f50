// vg_genomes.cpp — FASTA / FASTA.gz ingest, 2-bit packing, HBM residency.
// Input conventions follow the reference front-end (vclust.py:687-702, 962-963, 1159-1160):
// a single (multi-)FASTA file = one genome per record, a directory = one genome per file.
#include "vg_common.h"
#include <sys/mman.h>
#include <zlib.h>
#include <string.h>
#include <stdlib.h>
#include <thread>
#include <atomic>
#include <algorithm>
#include <memory>
#include <mutex>
#include <condition_variable>

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
    const int64_t padded = (len / al + 1) * al;      // at least one masked base behind every genome: no k-mer window reaches the next one
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

bool vg_fast_gunzip(const unsigned char* in, size_t n, int n_threads, char** out_p, size_t* out_n);      // vg_inflate.cpp

namespace {
// ---- whole-file buffers ---------------------------------------------------------------------
struct filebuf {            // plain files are mapped (no copy), gzip files are inflated into `own`
    std::vector<char> own; char* own_raw = nullptr; const char* ptr = nullptr; size_t len = 0; void* map = nullptr; size_t map_len = 0;
    std::mutex hole_mu; std::vector<std::pair<size_t, size_t>> holes;      // page ranges of the mapping already given back
    filebuf() {}
    filebuf(const filebuf&) = delete; filebuf& operator=(const filebuf&) = delete;
    // Unmapping 4 GB of FASTA in one call holds the address-space lock for 55-80 ms: wherever that call was put -- a
    // helper thread right after the ingest, or parked until the kernels run -- allocations, mappings and page faults of
    // the main thread queued up behind it (120 ms in front of the first small hipMalloc of vg_kmer_shared, 50 ms inside
    // the filter reader, 80 ms inside the bucket pipeline).  So the text is given back WHILE it is packed: the thread
    // that completes a stretch of genomes unmaps the whole pages of that stretch (drop; ~130 MB, 1-2 ms, beside 63
    // threads that do not fault), and what is left at the end (page edges, files of a directory) goes in slices.
    void drop(const char* lo, const char* hi) {
        if (!map || hi <= lo) return;
        const size_t pg = 4096;
        size_t a = ((size_t)(lo - (const char*)map) + pg - 1) / pg * pg, b = (size_t)(hi - (const char*)map) / pg * pg;
        if (b > map_len) b = map_len / pg * pg;
        if (b <= a) return;
        { std::lock_guard<std::mutex> lk(hole_mu); holes.emplace_back(a, b); }
        munmap((char*)map + a, b - a);
    }
    ~filebuf() {
        free(own_raw);
        if (!map) return;
        std::sort(holes.begin(), holes.end());
        const size_t slice = 32u << 20;
        auto give = [&](size_t a, size_t b) { for (size_t off = a; off < b; off += slice) munmap((char*)map + off, std::min(slice, b - off)); };
        size_t at = 0;
        for (auto& h : holes) { if (h.first > at) give(at, h.first); at = std::max(at, h.second); }
        if (map_len > at) give(at, map_len);
    }
    const char* data() const { return ptr; }
    size_t size() const { return len; }
};

// BGZF (bgzip, htslib): a series of gzip members of <= 64 KiB each, every member carrying its own compressed size in
// a 'BC' extra subfield -- the members can be found without inflating anything, so they are inflated in parallel.
// Returns false (nothing touched) when the file is not BGZF from its first to its last byte.
struct bgzf_block { size_t cdata, clen, out_off; uint32_t isize, crc; };
bool bgzf_blocks(const unsigned char* p, size_t n, std::vector<bgzf_block>& blocks, size_t* total) {
    size_t at = 0, out = 0;
    while (at < n) {
        if (n - at < 28 || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8 || !(p[at + 3] & 4)) return false;
        const size_t xlen = (size_t)p[at + 10] | ((size_t)p[at + 11] << 8);
        if (at + 12 + xlen > n) return false;
        size_t bsize = 0;
        for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {
            const size_t slen = (size_t)p[x + 2] | ((size_t)p[x + 3] << 8);
            if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= at + 12 + xlen) bsize = ((size_t)p[x + 4] | ((size_t)p[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || at + bsize > n || (p[at + 3] & ~4)) return false;     // (no name / comment / hcrc fields in BGZF)
        const unsigned char* tail = p + at + bsize - 8;
        bgzf_block b; b.cdata = at + 12 + xlen; b.clen = bsize - 12 - xlen - 8; b.out_off = out;
        b.crc = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
        b.isize = (uint32_t)tail[4] | ((uint32_t)tail[5] << 8) | ((uint32_t)tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (b.isize > 65536u) return false;
        out += b.isize; at += bsize;
        blocks.push_back(b);
    }
    *total = out;
    return !blocks.empty();
}
template <class F> void parallel_for(int64_t n, int n_threads, F fn) {
    std::atomic<int64_t> next(0);
    auto work = [&]() { for (;;) { int64_t i = next.fetch_add(1); if (i >= n) break; fn(i); } };
    std::vector<std::thread> th;
    for (int t = 1; t < std::min<int64_t>(n_threads, n); ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
}

bool slurp_bgzf(const std::string& path, FILE* f, filebuf& fb, int n_threads) {
    fseek(f, 0, SEEK_END); const long long sz = ftell(f);
    if (sz < 28) return false;
    void* m = mmap(nullptr, (size_t)sz, PROT_READ, MAP_PRIVATE, fileno(f), 0);
    if (m == MAP_FAILED) return false;
    struct unmap { void* m; size_t n; ~unmap() { munmap(m, n); } } um{ m, (size_t)sz };
    const unsigned char* p = (const unsigned char*)m;
    std::vector<bgzf_block> blocks; size_t total = 0;
    if (!bgzf_blocks(p, (size_t)sz, blocks, &total)) return false;
    vg_host_mark("ingest: bgzf members found");
    (void)madvise(m, (size_t)sz, MADV_WILLNEED);
    // (not a std::vector: its resize would clear 800 MB on one thread before the workers overwrite them)
    fb.own_raw = (char*)malloc(std::max<size_t>(total, 1));
    if (!fb.own_raw) throw vg_error(VG_ENOMEM, "out of host memory");
    char* const text = fb.own_raw;
    std::atomic<bool> bad(false);
    // runs of 64 blocks (<= 4 MiB of text) per work item
    const int64_t n_items = ((int64_t)blocks.size() + 63) / 64;
    parallel_for(n_items, n_threads, [&](int64_t it) {
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return; }
        for (size_t b = (size_t)it * 64; b < std::min(blocks.size(), (size_t)(it + 1) * 64) && !bad.load(std::memory_order_relaxed); ++b) {
            const bgzf_block& k = blocks[b];
            // (zlib: for 64 KiB members the own decoder's table builds and scratch copy cost more than its loop saves --
            // 3.3 s against 2.4 s for 814 MB on one thread)
            if (b != (size_t)it * 64 && inflateReset(&zs) != Z_OK) { bad = true; break; }
            zs.next_in = (Bytef*)(p + k.cdata); zs.avail_in = (uInt)k.clen;
            zs.next_out = (Bytef*)text + k.out_off; zs.avail_out = k.isize;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0 || zs.avail_in != 0 ||
                (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)text + k.out_off, k.isize) != k.crc) { bad = true; break; }
        }
        inflateEnd(&zs);
    });
    vg_host_mark("ingest: bgzf inflated");
    if (bad.load()) throw vg_error(VG_EIO, "read error in " + path + " (corrupt BGZF block)");
    fb.ptr = text; fb.len = total;
    return true;
}

void slurp(const std::string& path, filebuf& fb, int n_threads) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw vg_error(VG_EIO, "cannot open " + path);
    unsigned char magic[2] = { 0, 0 };
    size_t got = fread(magic, 1, 2, f);
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (!gz) {
        fseek(f, 0, SEEK_END); const long long sz = ftell(f);
        if (sz > 0) {
            void* m = mmap(nullptr, (size_t)sz, PROT_READ, MAP_PRIVATE, fileno(f), 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)sz, MADV_WILLNEED);
                fb.map = m; fb.map_len = (size_t)sz; fb.ptr = (const char*)m; fb.len = (size_t)sz;
                fclose(f);
                return;
            }
        }
        fseek(f, 0, SEEK_SET);                           // not mappable (pipe, special file): read it
        fb.own.resize((size_t)std::max<long long>(sz, 0));
        size_t off = 0;
        while (off < fb.own.size()) { size_t r = fread(fb.own.data() + off, 1, fb.own.size() - off, f); if (!r) break; off += r; }
        fclose(f);
        if (off != fb.own.size()) throw vg_error(VG_EIO, "read error in " + path);
        fb.ptr = fb.own.data(); fb.len = fb.own.size();
        return;
    }
    // bgzip output inflates block-parallel; any other gzip stream (one member, or members of unknown size) serially:
    // first by the library's own decoder (vg_inflate.cpp: ~1.3x zlib's inflate per thread, CRC-32 on worker threads,
    // every member verified), and by zlib if that one declines the file (VG_GZ=zlib: always zlib)
    bool done = false;
    try { done = slurp_bgzf(path, f, fb, n_threads); } catch (...) { fclose(f); throw; }
    static const bool zlib_only = [] { const char* e = vg_dev_getenv("VG_GZ"); return e && !strcmp(e, "zlib"); }();
    if (!done && !zlib_only) {
        fseek(f, 0, SEEK_END); const long long sz = ftell(f);
        void* m = sz > 0 ? mmap(nullptr, (size_t)sz, PROT_READ, MAP_PRIVATE, fileno(f), 0) : MAP_FAILED;
        if (m != MAP_FAILED) {
            (void)madvise(m, (size_t)sz, MADV_SEQUENTIAL);
            char* o = nullptr; size_t on = 0;
            if (vg_fast_gunzip((const unsigned char*)m, (size_t)sz, n_threads, &o, &on)) { fb.own_raw = o; fb.ptr = o; fb.len = on; done = true; vg_host_mark("ingest: gzip inflated by the own decoder"); }
            munmap(m, (size_t)sz);
        }
    }
    fclose(f);
    if (done) return;
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) throw vg_error(VG_EIO, "cannot open " + path);
    gzbuffer(g, 1 << 20);
    size_t off = 0; fb.own.resize(1 << 22);
    for (;;) {
        if (fb.own.size() - off < (1 << 20)) fb.own.resize(fb.own.size() * 2);
        int n = gzread(g, fb.own.data() + off, (unsigned)std::min<size_t>(fb.own.size() - off, 1u << 30));
        if (n < 0) { gzclose(g); throw vg_error(VG_EIO, "read error in " + path); }
        if (n == 0) break;
        off += (size_t)n;
    }
    int zerr = 0; (void)gzerror(g, &zerr);
    gzclose(g);
    if (zerr != Z_OK && zerr != Z_STREAM_END) throw vg_error(VG_EIO, "read error in " + path);
    fb.own.resize(off);
    fb.ptr = fb.own.data(); fb.len = off;
}

// one FASTA record inside a buffer: [hdr, hdr_end) header line without '>', [seq, end) sequence lines
struct record { const char* hdr; const char* hdr_end; const char* seq; const char* end; int64_t len; };

struct code_lut { uint8_t t[256]; code_lut() { for (int i = 0; i < 256; ++i) t[i] = code_of((unsigned char)i); } };
const code_lut LUT;
inline bool is_ws(char ch) { return ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t'; }

void find_records(const filebuf& fb, std::vector<record>& out, int n_threads) {
    const char* p = fb.data(); const char* e = p + fb.size();
    // record starts: '>' at the beginning of a line, found chunk-parallel with memchr
    const int T = std::max(1, n_threads);
    std::vector<std::vector<const char*>> starts(T);
    auto scan = [&](int t) {
        const char* lo = p + fb.size() * t / T; const char* hi = p + fb.size() * (t + 1) / T;
        // piece by piece: the pages of a piece are mapped in one call (instead of one minor fault per 64 KB) and scanned
        // while they are hot.  Small pieces on purpose: a populate call holds the address-space lock shared, and the
        // HIP context that is being created on the warm-up thread needs it exclusively for every mapping it makes --
        // with one call per 64 MB stretch the context took 350 ms instead of 150.
        const size_t piece = 2u << 20;
        for (const char* c0 = lo; c0 < hi; c0 += piece) {
            const char* c1 = std::min(hi, c0 + piece);
#ifdef MADV_POPULATE_READ
            if (fb.map) {
                const uintptr_t a = (uintptr_t)c0 & ~(uintptr_t)4095, b = ((uintptr_t)c1 + 4095) & ~(uintptr_t)4095;
                (void)madvise((void*)a, (size_t)(b - a), MADV_POPULATE_READ);
            }
#endif
            for (const char* q = c0; q < c1;) {
                const char* g = (const char*)memchr(q, '>', (size_t)(c1 - q));
                if (!g) break;
                if (g == p || g[-1] == '\n') starts[t].push_back(g);
                q = g + 1;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(scan, t);
    scan(0);
    for (auto& x : th) x.join();
    std::vector<const char*> all;
    for (auto& v : starts) all.insert(all.end(), v.begin(), v.end());
    out.resize(all.size());
    // (the header line of every record is touched here: 10^5 cache misses when done by one thread)
    parallel_for(((int64_t)all.size() + 1023) / 1024, T, [&](int64_t c) {
        for (size_t i = (size_t)c * 1024; i < std::min(all.size(), (size_t)(c + 1) * 1024); ++i) {
            record& r = out[i];
            r.hdr = all[i] + 1;
            const char* rec_end = i + 1 < all.size() ? all[i + 1] : e;
            const char* nl = (const char*)memchr(r.hdr, '\n', (size_t)(rec_end - r.hdr));
            r.hdr_end = nl ? nl : rec_end; r.seq = nl ? nl + 1 : rec_end; r.end = rec_end; r.len = 0;
        }
    });
}

// pack the bases of one record into the set's arrays starting at padded base position `at` (2-bit codes + N mask);
// returns true if an N was seen.  Callers own disjoint word ranges and the arrays are zero.  Line by line (memchr
// for the line end), sixteen symbols per trip: sixteen table look-ups OR-ed into one 32-bit group, no per-symbol
// branch, appended to the output through a 64-bit shift register (the output need not be word aligned).  A group
// with anything unusual in it (N, white space inside a line) goes symbol by symbol.
struct pack_lut { uint8_t t[256]; pack_lut() { for (int i = 0; i < 256; ++i) { const char ch = (char)i; t[i] = is_ws(ch) ? 5 : code_of((unsigned char)i); } } };
const pack_lut PLUT;
bool pack_record(const record& r, vg_genomes* g, int64_t at, int64_t* n_symbols) {
    uint32_t* pk = g->packed.data(); uint32_t* mk = g->nmask.data();
    bool any_n = false; int64_t i = at;                       // i = position of the next symbol
    uint64_t acc = 0; int nb = 2 * (int)(at & 15);            // pending output bits of word (i - pending symbols) >> 4
    int64_t wo = at >> 4;                                     // word the low bits of acc belong to
    auto flush = [&]() { while (nb >= 32) { pk[wo++] |= (uint32_t)acc; acc >>= 32; nb -= 32; } };
    const char* q = r.seq; const char* const end = r.end;
    while (q < end) {
        const char* nl = (const char*)memchr(q, '\n', (size_t)(end - q));
        const char* const le = nl ? nl : end;
        while (q < le) {
            const int m = (int)std::min<int64_t>(16, le - q);
            uint32_t w = 0, bad = 0;
            for (int j = 0; j < m; ++j) { const uint32_t c = PLUT.t[(unsigned char)q[j]]; w |= (c & 3u) << (2 * j); bad |= c; }
            if (bad < 4) { acc |= (uint64_t)w << nb; nb += 2 * m; i += m; q += m; flush(); continue; }
            for (int j = 0; j < m; ++j) {
                const uint32_t c = PLUT.t[(unsigned char)q[j]];
                if (c == 5) continue;
                if (c == 4) { mk[i >> 5] |= 1u << (i & 31); any_n = true; } else acc |= (uint64_t)c << nb;
                nb += 2; ++i; flush();
            }
            q += m;
        }
        q = nl ? nl + 1 : end;
    }
    if (nb > 0) pk[wo] |= (uint32_t)acc;
    *n_symbols = i - at;
    return any_n;
}

std::string first_token(const char* b, const char* e) {
    const char* q = b; while (q < e && *q != ' ' && *q != '\t' && *q != '\r') ++q;
    return std::string(b, q);
}
}  // namespace

// The N mask of a genome without N is its padding and nothing else: it is written on the device from the genome's
// length instead of travelling (a third of the upload of a set).  One thread per mask word; words of genomes that do
// hold an N are left alone (they are uploaded).
__global__ void __launch_bounds__(256)
k_mask_of_lengths(uint32_t* __restrict__ nmask, int64_t n_words, const uint32_t* __restrict__ blk2g, int align_shift,
                  const int64_t* __restrict__ base_off, const int64_t* __restrict__ len, const uint8_t* __restrict__ has_n) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p0 = w * 32;
        const uint32_t gi = blk2g[p0 >> align_shift];
        if (has_n[gi]) continue;
        const int64_t end = base_off[gi] + len[gi];              // first padding position of the genome
        nmask[w] = end >= p0 + 32 ? 0u : end <= p0 ? 0xffffffffu : (0xffffffffu << (uint32_t)(end - p0));
    }
}

// to_device: the packed arrays travel to the HBM of the library's device WHILE the genomes are packed (a helper
// thread uploads every stretch of genomes as soon as its last genome is done), and the set comes back resident
static void genomes_load_impl(const char* const* paths, int n_paths, int multisample, int n_threads, bool to_device, vg_genomes** out) {
    if (!paths || n_paths <= 0 || !out) throw vg_error(VG_EINVAL, "vg_genomes_load: bad arguments");
    const int T = std::max(1, n_threads);
    const bool multi = multisample && n_paths == 1;
    vg_host_mark("ingest: enter");
    // 1. whole files into memory (gz inflated), files in parallel.  The mappings and the record lists are given to a
    // helper thread at the end: unmapping 4 GB of FASTA costs ~80 ms the caller need not wait for.
    struct input_state { std::vector<filebuf> bufs; std::vector<std::vector<record>> recs; };
    auto* in_state = new input_state();
    struct in_guard { input_state* p; ~in_guard() { if (p) { input_state* q = p; try { vg_defer([q] { delete q; }); } catch (...) { delete q; } } } } in_g{ in_state };
    in_state->bufs = std::vector<filebuf>((size_t)n_paths);
    std::vector<filebuf>& bufs = in_state->bufs;
    {
        std::string first_err; std::atomic<bool> failed(false);
        parallel_for(n_paths, T, [&](int64_t i) {
            if (failed.load()) return;
            try { slurp(paths[i], bufs[(size_t)i], std::max(1, T / (int)std::min<int64_t>(n_paths, T))); }
            catch (const std::exception& e) { if (!failed.exchange(true)) first_err = e.what(); }
        });
        if (failed.load()) throw vg_error(VG_EIO, first_err);
    }
    vg_host_mark("ingest: files mapped");
    // 2. records and their lengths
    in_state->recs = std::vector<std::vector<record>>((size_t)n_paths);
    std::vector<std::vector<record>>& recs = in_state->recs;
    for (int i = 0; i < n_paths; ++i) find_records(bufs[(size_t)i], recs[(size_t)i], multi ? T : 1);
    struct gdesc { int file; size_t r0, r1; int64_t len; };
    std::vector<gdesc> gd;
    for (int i = 0; i < n_paths; ++i) {
        auto& v = recs[(size_t)i];
        if (multi) for (size_t r = 0; r < v.size(); ++r) gd.push_back({ i, r, r + 1, 0 });
        else if (!v.empty()) gd.push_back({ i, 0, v.size(), 0 });
    }
    // No counting pass over the text: the layout is made from an UPPER BOUND of every genome's length -- the bytes of its
    // sequence lines, line ends included (1/60 more than the truth for 60-column FASTA) --, the true length is what the
    // packer has written when it reaches the end of the record, and the difference is padding (masked like the padding
    // that aligns the next genome anyway).  One pass over 4 GB of text less per process.
    int64_t total = 0;
    for (auto& d : gd) {
        for (size_t r = d.r0; r < d.r1; ++r) d.len += (int64_t)(recs[(size_t)d.file][r].end - recs[(size_t)d.file][r].seq);
        d.len += (int64_t)(d.r1 - d.r0) - 1;                 // records of one genome are joined by one N
        total += d.len;
    }
    vg_host_mark("ingest: records found");
    // 3. layout, then parallel packing straight into the 2-bit arrays
    vg_genomes* g = new vg_genomes();
    std::unique_ptr<vg_genomes> guard(g);
    g->align_shift = vg_choose_align_shift(total, (int64_t)gd.size());
    const int64_t al = 1LL << g->align_shift;
    g->base_off.push_back(0);
    for (auto& d : gd) {
        const int64_t padded = (d.len / al + 1) * al;          // >= 1 masked base behind the genome (see vg_genomes_append)
        for (int64_t b = g->base_off.back() >> g->align_shift; b < (g->base_off.back() + padded) >> g->align_shift; ++b) g->blk2g.push_back((uint32_t)g->n);
        g->base_off.push_back(g->base_off.back() + padded);
        g->len.push_back(0); g->n_parts.push_back((int32_t)(d.r1 - d.r0)); g->has_n.push_back(0);      // (the length: set by the packer)
        const record& r0 = recs[(size_t)d.file][d.r0];
        if (multi) g->names.push_back(first_token(r0.hdr, r0.hdr_end));
        else { std::string pth = paths[d.file]; size_t sl = pth.find_last_of('/'); g->names.push_back(sl == std::string::npos ? pth : pth.substr(sl + 1)); }
        g->n++;
    }
    vg_host_mark("ingest: layout");
    // first touch in parallel: the arrays are written by the packing threads right below
    g->packed.reserve((size_t)(g->padded_total() / 16) + 16); g->nmask.reserve((size_t)(g->padded_total() / 32) + 16);   // + the slack vg_genomes_finish adds
    g->packed.resize((size_t)(g->padded_total() / 16)); g->nmask.resize((size_t)(g->padded_total() / 32));
    {
        const int64_t chunk = 1 << 20, np = ((int64_t)g->packed.size() + chunk - 1) / chunk, nm = ((int64_t)g->nmask.size() + chunk - 1) / chunk;
        parallel_for(np + nm, T, [&](int64_t c) {
            std::vector<uint32_t, no_init_alloc<uint32_t>>& v = c < np ? g->packed : g->nmask;
            const int64_t lo = (c < np ? c : c - np) * chunk, hi = std::min<int64_t>(lo + chunk, (int64_t)v.size());
#ifdef MADV_POPULATE_WRITE
            {   // fresh anonymous memory: populate the chunk's whole pages in one call (they come zeroed) instead of faulting page by page
                const uintptr_t a = ((uintptr_t)(v.data() + lo) + 4095) & ~(uintptr_t)4095, b = (uintptr_t)(v.data() + hi) & ~(uintptr_t)4095;
                if (b > a) (void)madvise((void*)a, (size_t)(b - a), MADV_POPULATE_WRITE);
            }
#endif
            memset(v.data() + lo, 0, (size_t)(hi - lo) * sizeof(uint32_t));
        });
    }
    vg_host_mark("ingest: arrays cleared");
    // stretches of genomes of ~32 MB of packed bases: the unit of the overlapped upload
    const int64_t n_g = (int64_t)gd.size();
    std::vector<int64_t> st_first;                          // first genome of every stretch (+ n_g)
    {
        const int64_t want = 128LL << 20;                   // bases per stretch (= 32 MB packed + 16 MB mask)
        int64_t last = 0; st_first.push_back(0);
        for (int64_t gi = 0; gi < n_g; ++gi) if (g->base_off[(size_t)gi + 1] - g->base_off[(size_t)last] >= want && gi + 1 < n_g) { st_first.push_back(gi + 1); last = gi + 1; }
        st_first.push_back(n_g);
    }
    const int64_t n_st = (int64_t)st_first.size() - 1;
    std::vector<int> st_of((size_t)n_g);
    for (int64_t c = 0; c < n_st; ++c) for (int64_t gi = st_first[(size_t)c]; gi < st_first[(size_t)c + 1]; ++gi) st_of[(size_t)gi] = (int)c;
    std::unique_ptr<std::atomic<int64_t>[]> st_left(new std::atomic<int64_t>[(size_t)std::max<int64_t>(n_st, 1)]);
    for (int64_t c = 0; c < n_st; ++c) st_left[(size_t)c].store(st_first[(size_t)c + 1] - st_first[(size_t)c]);
    std::unique_ptr<std::atomic<int>[]> st_has_n(new std::atomic<int>[(size_t)std::max<int64_t>(n_st, 1)]);      // a genome of the stretch holds an N: its mask travels
    for (int64_t c = 0; c < n_st; ++c) st_has_n[(size_t)c].store(0);
    std::mutex up_mu; std::condition_variable up_cv; std::vector<int64_t> up_queue; bool up_done = false; std::string up_err;
    std::thread uploader; int dev = 0;
    const size_t pk_words = (size_t)(g->padded_total() / 16) + 16, mk_words = (size_t)(g->padded_total() / 32) + 16;   // with the slack vg_genomes_finish adds
    if (to_device && n_g > 0) {
        vg_require_device();
        VG_HIP(hipGetDevice(&dev));
        g->d_packed.alloc(pk_words); g->d_nmask.alloc(mk_words);
        uploader = std::thread([&]() {
            try {
                vg_require_device();
                hipStream_t st; VG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                for (;;) {
                    int64_t c;
                    { std::unique_lock<std::mutex> lk(up_mu); up_cv.wait(lk, [&] { return !up_queue.empty() || up_done; }); if (up_queue.empty()) break; c = up_queue.back(); up_queue.pop_back(); }
                    const int64_t b0 = g->base_off[(size_t)st_first[(size_t)c]], b1 = g->base_off[(size_t)st_first[(size_t)c + 1]];      // multiples of 64 bases
                    VG_HIP(hipMemcpyAsync(g->d_packed.p + b0 / 16, g->packed.data() + b0 / 16, (size_t)(b1 - b0) / 16 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    if (st_has_n[(size_t)c].load()) VG_HIP(hipMemcpyAsync(g->d_nmask.p + b0 / 32, g->nmask.data() + b0 / 32, (size_t)(b1 - b0) / 32 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    VG_HIP(hipStreamSynchronize(st));
                }
                (void)hipStreamDestroy(st);
            } catch (const std::exception& e) { std::lock_guard<std::mutex> lk(up_mu); up_err = e.what(); }
        });
    }
    struct join_guard { std::thread& t; std::mutex& m; std::condition_variable& cv; bool& done;
                        ~join_guard() { { std::lock_guard<std::mutex> lk(m); done = true; } cv.notify_all(); if (t.joinable()) t.join(); } } jg{ uploader, up_mu, up_cv, up_done };
    parallel_for(n_g, T, [&](int64_t gi) {
        const gdesc& d = gd[(size_t)gi];
        int64_t at = g->base_off[(size_t)gi]; bool any_n = false;
        for (size_t r = d.r0; r < d.r1; ++r) {
            if (r > d.r0) { g->nmask[(size_t)(at >> 5)] |= 1u << (at & 31); ++at; any_n = true; }   // joining N
            int64_t n_sym = 0;
            any_n |= pack_record(recs[(size_t)d.file][r], g, at, &n_sym);
            at += n_sym;
        }
        g->len[(size_t)gi] = at - g->base_off[(size_t)gi];
        {   // padding up to the next genome: single bits up to a word boundary, then whole words
            int64_t i = at; const int64_t e = g->base_off[(size_t)gi + 1];
            for (; i < e && (i & 31); ++i) g->nmask[(size_t)(i >> 5)] |= 1u << (i & 31);
            for (; i + 32 <= e; i += 32) g->nmask[(size_t)(i >> 5)] = 0xffffffffu;
            for (; i < e; ++i) g->nmask[(size_t)(i >> 5)] |= 1u << (i & 31);
        }
        g->has_n[(size_t)gi] = any_n ? 1 : 0;
        if (any_n) st_has_n[(size_t)st_of[(size_t)gi]].store(1);
        if (st_left[(size_t)st_of[(size_t)gi]].fetch_sub(1) == 1) {
            // the last genome of its stretch: the stretch can travel, and nobody reads its text again
            const int64_t c = st_of[(size_t)gi];
            if (uploader.joinable()) {
                { std::lock_guard<std::mutex> lk(up_mu); up_queue.push_back(c); }
                up_cv.notify_one();
            }
            if (multi) {
                const gdesc& d0 = gd[(size_t)st_first[(size_t)c]]; const gdesc& d1 = gd[(size_t)st_first[(size_t)c + 1] - 1];
                bufs[(size_t)d0.file].drop(recs[(size_t)d0.file][d0.r0].hdr - 1, recs[(size_t)d1.file][d1.r1 - 1].end);
            }
        }
    });
    vg_genomes_finish(g);
    vg_host_mark("ingest: packed");
    if (uploader.joinable()) {
        { std::lock_guard<std::mutex> lk(up_mu); up_done = true; }
        up_cv.notify_all(); uploader.join();
        if (!up_err.empty()) throw vg_error(VG_EHIP, "upload of the genome arrays failed: " + up_err);
        // the slack words behind the last genome and the small arrays
        hipStream_t s = vg_stream();
        const size_t pk_tail = (size_t)(g->padded_total() / 16), mk_tail = (size_t)(g->padded_total() / 32);
        VG_HIP(hipMemcpyAsync(g->d_packed.p + pk_tail, g->packed.data() + pk_tail, (g->packed.size() - pk_tail) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        VG_HIP(hipMemcpyAsync(g->d_nmask.p + mk_tail, g->nmask.data() + mk_tail, (g->nmask.size() - mk_tail) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        g->d_base_off.alloc(g->base_off.size()); g->d_base_off.upload(g->base_off.data(), g->base_off.size(), s);
        g->d_len.alloc(std::max<size_t>(1, g->len.size())); g->d_len.upload(g->len.data(), g->len.size(), s);
        g->d_has_n.alloc(std::max<size_t>(1, g->has_n.size())); g->d_has_n.upload(g->has_n.data(), g->has_n.size(), s);
        g->d_blk2g.alloc(g->blk2g.size()); g->d_blk2g.upload(g->blk2g.data(), g->blk2g.size(), s);
        // the masks that did not travel (the uploader's stream has been drained: its copies are in place)
        hipLaunchKernelGGL(k_mask_of_lengths, dim3(4096), dim3(256), 0, s, g->d_nmask.p, (int64_t)mk_tail, (const uint32_t*)g->d_blk2g.p, g->align_shift,
                           (const int64_t*)g->d_base_off.p, (const int64_t*)g->d_len.p, (const uint8_t*)g->d_has_n.p);
        VG_HIP(hipStreamSynchronize(s));
        g->device = dev;
        vg_host_mark("ingest: resident");
    }
    *out = guard.release();
}

extern "C" int vg_genomes_load(const char* const* paths, int n_paths, int multisample, int n_threads,
                               vg_genomes** out) {
    VG_API_BEGIN
    genomes_load_impl(paths, n_paths, multisample, n_threads, false, out);
    VG_API_END
}
// internal (vg_api.cpp): ingest with the upload overlapped
int vg_genomes_load_resident(const char* const* paths, int n_paths, int multisample, int n_threads, vg_genomes** out) {
    VG_API_BEGIN
    genomes_load_impl(paths, n_paths, multisample, n_threads, true, out);
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

extern "C" void vg_genomes_free(vg_genomes* g) { if (g) vg_lz_drop_prepared(g); delete g; }
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

extern "C" int vg_genomes_codes(const vg_genomes* g, int idx, uint8_t* out) {
    if (!g || !out || idx < 0 || idx >= g->n) return VG_EINVAL;
    const int64_t p0 = g->base_off[(size_t)idx];
    for (int64_t j = 0; j < g->len[(size_t)idx]; ++j) {
        const int64_t p = p0 + j;
        const bool n = (g->nmask[(size_t)(p >> 5)] >> (p & 31)) & 1u;
        out[j] = n ? 4 : (uint8_t)((g->packed[(size_t)(p >> 4)] >> (2 * (p & 15))) & 3u);
    }
    return VG_OK;
}

extern "C" int vg_genomes_to_device(vg_genomes* g) {
    VG_API_BEGIN
    if (!g) throw vg_error(VG_EINVAL, "null genome set");
    vg_require_device();
    int dev = 0; VG_HIP(hipGetDevice(&dev));
    if (g->device == dev) return VG_OK;
    if (g->device >= 0) {
        // resident on another device: the allocator knows every block's device and returns these to THAT device's
        // driver instead of caching them here (vg_dev_free)
        g->d_packed.release(); g->d_planes.release(); g->d_nmask.release(); g->d_base_off.release(); g->d_len.release(); g->d_has_n.release(); g->d_blk2g.release();
        g->device = -1;
    }
    hipStream_t s = vg_stream();
    g->d_packed.alloc(g->packed.size()); g->d_packed.upload(g->packed.data(), g->packed.size(), s);
    g->d_nmask.alloc(g->nmask.size());   g->d_nmask.upload(g->nmask.data(), g->nmask.size(), s);
    g->d_base_off.alloc(g->base_off.size()); g->d_base_off.upload(g->base_off.data(), g->base_off.size(), s);
    g->d_len.alloc(std::max<size_t>(1, g->len.size())); if (g->n) g->d_len.upload(g->len.data(), g->len.size(), s);
    g->d_has_n.alloc(std::max<size_t>(1, g->has_n.size())); if (g->n) g->d_has_n.upload(g->has_n.data(), g->has_n.size(), s);
    g->d_blk2g.alloc(g->blk2g.size()); g->d_blk2g.upload(g->blk2g.data(), g->blk2g.size(), s);
    VG_HIP(hipStreamSynchronize(s));
    g->device = dev;
    vg_host_mark("genomes uploaded");
    VG_API_END
}
