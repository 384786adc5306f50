"""Synthetic genome sets of SURVEY.md section 8(d) / BASELINE.json configs[1..4].

Every random draw comes from splitmix64 used as a counter-based generator,

    draw(stream, i) = mix64(stream + (i + 1) * 0x9E3779B97F4A7C15)      (mod 2^64)
    mix64(z): z ^= z >> 30; z *= 0xBF58476D1CE4E5B9; z ^= z >> 27; z *= 0x94D049BB133111EB; z ^= z >> 31

so the sets can be regenerated bit for bit from any language; `stream` is mix64 of (seed, family, member,
purpose).  sha256 digests of the small sets are pinned in tests/golden/synth_sha256.json.

Mutation model: a family has a uniform-random ACGT ancestor; every member is the ancestor with a
per-member substitution rate p ~ U(p_lo, p_hi), `n_indels` indels of length U(1, 50), with probability 0.2
one inversion (reverse complement in place) of 1-5 kb and with probability 0.2 one translocation of 1-5 kb.

Named workloads (`make_workload`):
  phage-1k    100 families x 10 x 40 kb, seed 1                         BASELINE configs[1]
  imgvr-10k   10 000 contigs, log-uniform 5-200 kb, families geometric(0.2) <= 20, seed 2   configs[2]
  phage-100k  10 000 families x 10 x 40 kb, seed 3                      configs[3]
  contigs-1M  1 000 000 contigs, log-uniform 2-100 kb, seed 4           configs[4] (scaled with n=)
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream(seed, a, b=0, c=0):
    """Stream id of (seed, family a, member b, purpose c)."""
    with np.errstate(over='ignore'):
        z = mix64(np.uint64(seed) + _G)
        z = mix64(z ^ (np.uint64(a) * _M1))
        z = mix64(z ^ (np.uint64(b) * _M2 + np.uint64(1)))
        return mix64(z ^ (np.uint64(c) * _G + np.uint64(2)))


def draws(stream, n, start=0):
    """n consecutive 64-bit draws of a stream, from draw index `start`."""
    with np.errstate(over='ignore'):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        return mix64(np.uint64(stream) + i * _G)


def _unit(x):
    return (np.asarray(x, dtype=np.uint64) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _below(x, n):
    """Integer in [0, n) from a 64-bit draw (top bits, multiply-shift)."""
    return int((int(x) >> 32) * int(n) >> 32)


def _bases(stream, n):
    return (draws(stream, n) >> np.uint64(62)).astype(np.uint8)


def _mutate(seed, fam, mem, anc, p_lo, p_hi, n_indels):
    g = anc.copy()
    ctl = draws(_stream(seed, fam, mem, 1), 8 + 4 * n_indels)
    p = p_lo + (p_hi - p_lo) * float(_unit(ctl[0]))
    d = draws(_stream(seed, fam, mem, 2), len(g))
    mask = _unit(d) < p
    g[mask] = (g[mask] + ((d[mask] & np.uint64(0xffff)) % np.uint64(3)).astype(np.uint8) + np.uint8(1)) & np.uint8(3)
    for t in range(n_indels):
        c = ctl[8 + 4 * t: 12 + 4 * t]
        pos = _below(c[0], len(g)); ln = 1 + _below(c[1], 50)
        if int(c[2]) >> 63:
            g = np.concatenate([g[:pos], g[pos + ln:]])
        else:
            g = np.concatenate([g[:pos], _bases(_stream(seed, fam, mem, 3 + t), ln), g[pos:]])
    if float(_unit(ctl[1])) < 0.2 and len(g) > 6000:
        ln = 1000 + _below(ctl[2], 4001); pos = _below(ctl[3], len(g) - ln)
        g[pos:pos + ln] = (3 - g[pos:pos + ln])[::-1]
    if float(_unit(ctl[4])) < 0.2 and len(g) > 6000:
        ln = 1000 + _below(ctl[5], 4001); pos = _below(ctl[6], len(g) - ln)
        seg = g[pos:pos + ln].copy()
        rest = np.concatenate([g[:pos], g[pos + ln:]])
        dst = _below(ctl[7], len(rest))
        g = np.concatenate([rest[:dst], seg, rest[dst:]])
    return g


def _map_threads(fn, items):
    """numpy releases the GIL inside the array kernels: families are generated on all host cores."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    n_thr = min(len(items), max(1, min(os.cpu_count() or 1, 64)))
    if n_thr <= 1:
        return [fn(x) for x in items]
    with ThreadPoolExecutor(n_thr) as ex:
        return list(ex.map(fn, items, chunksize=max(1, len(items) // (8 * n_thr))))


def _native(plan, seed, p_lo, p_hi, n_indels):
    """The same sets from the multithreaded C++ generator of libvclust_gpu.so (vg_synth_plan); None when the
    library is not built.  plan: list of (family index, members, ancestor length)."""
    import ctypes as C
    try:
        from . import _lib
        lib = _lib.load()
    except Exception:
        return None
    fam = np.ascontiguousarray([p[0] for p in plan], dtype=np.int64)
    mem = np.ascontiguousarray([p[1] for p in plan], dtype=np.int32)
    ln = np.ascontiguousarray([p[2] for p in plan], dtype=np.int32)
    codes_p, off_p, ng = C.c_void_p(), C.c_void_p(), C.c_int64()
    # (0 = all host threads; the ranks of a multi-GPU run generate the same set side by side and share the host's cores)
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    n_thr = 0 if world <= 1 else max(1, (os.cpu_count() or 1) // world)
    _lib.check(lib.vg_synth_plan(fam.ctypes.data, mem.ctypes.data, ln.ctypes.data, len(plan), int(seed), float(p_lo), float(p_hi),
                                 int(n_indels), n_thr, C.byref(codes_p), C.byref(off_p), C.byref(ng)))
    n = ng.value
    offsets = np.ctypeslib.as_array(C.cast(off_p, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
    total = int(offsets[-1])
    codes = np.ctypeslib.as_array(C.cast(codes_p, C.POINTER(C.c_uint8)), shape=(max(total, 1),))[:total].copy()
    lib.vg_free(codes_p); lib.vg_free(off_p)
    return codes, offsets


def _finish(seqs):
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
    return codes, offsets


def make_families(n_families, members, length=40000, seed=1, p_lo=0.005, p_hi=0.12, n_indels=5,
                  length_range=None, first_family=0, native=True):
    """-> (codes uint8[total], offsets int64[n+1], names list).

    length_range=(lo, hi): ancestor lengths log-uniform in [lo, hi] instead of `length`.
    first_family: generate families first_family .. first_family + n_families - 1 of the set (a family depends
    on (seed, family index) only, so ranks can build disjoint slices of one set)."""
    def one_family(f):
        if length_range:
            u = float(_unit(draws(_stream(seed, f, 0, 100), 1)[0]))
            ln = int(np.exp(np.log(length_range[0]) + u * (np.log(length_range[1]) - np.log(length_range[0]))))
        else:
            ln = length
        anc = _bases(_stream(seed, f, 0, 0), ln)
        return [_mutate(seed, f, m + 1, anc, p_lo, p_hi, n_indels) for m in range(members)]

    names = [f'fam{f:05d}_m{m:02d}' for f in range(first_family, first_family + n_families) for m in range(members)]
    if native and not length_range:
        out = _native([(f, members, length) for f in range(first_family, first_family + n_families)], seed, p_lo, p_hi, n_indels)
        if out is not None:
            return out[0], out[1], names
    fams = _map_threads(one_family, range(first_family, first_family + n_families))
    seqs = [g for fam in fams for g in fam]
    codes, offsets = _finish(seqs)
    return codes, offsets, names


def make_contigs(n_contigs, len_lo=5000, len_hi=200000, max_family=20, seed=2, p_lo=0.005, p_hi=0.12, n_indels=5, native=True):
    """IMG/VR-like set (BASELINE configs[2] / [4]): ancestor lengths log-uniform in [len_lo, len_hi],
    family sizes geometric(0.2) capped at max_family, same mutation model as make_families.
    -> (codes, offsets, names, family id per contig)."""
    plan, total, fam = [], 0, 0
    while total < n_contigs:
        c = draws(_stream(seed, fam, 0, 100), 2)
        u = min(max(float(_unit(c[0])), 1e-300), 1.0 - 1e-16)
        members = min(int(np.ceil(np.log1p(-u) / np.log(0.8))), max_family, n_contigs - total)   # geometric(0.2)
        members = max(members, 1)
        ln = int(np.exp(np.log(len_lo) + float(_unit(c[1])) * (np.log(len_hi) - np.log(len_lo))))
        plan.append((fam, members, ln)); total += members; fam += 1

    def one_family(item):
        f, members, ln = item
        anc = _bases(_stream(seed, f, 0, 0), ln)
        return [_mutate(seed, f, m + 1, anc, p_lo, p_hi, n_indels) for m in range(members)]

    names = [f'ctg{f:06d}_{m:02d}' for f, members, _ in plan for m in range(members)]
    fam_of = np.array([f for f, members, _ in plan for _ in range(members)], dtype=np.int64)
    out = _native(plan, seed, p_lo, p_hi, n_indels) if native else None
    if out is not None:
        return out[0], out[1], names, fam_of
    fams = _map_threads(one_family, plan)
    seqs = [g for fm in fams for g in fm]
    codes, offsets = _finish(seqs)
    return codes, offsets, names, fam_of


WORKLOADS = {
    'phage-1k': dict(kind='families', n_families=100, members=10, length=40000, seed=1),
    'imgvr-10k': dict(kind='contigs', n=10000, len_lo=5000, len_hi=200000, seed=2),
    'phage-100k': dict(kind='families', n_families=10000, members=10, length=40000, seed=3),
    'contigs-1M': dict(kind='contigs', n=1000000, len_lo=2000, len_hi=100000, seed=4),
}


def make_workload(name, n=None):
    """Named workload of SURVEY 8(d); n scales it down (families for the phage sets, contigs otherwise).
    -> (codes, offsets, names, description)."""
    w = dict(WORKLOADS[name])
    if w['kind'] == 'families':
        nf = n if n is not None else w['n_families']
        codes, offsets, names = make_families(nf, w['members'], length=w['length'], seed=w['seed'])
        desc = f"{name}: {nf} families x {w['members']} members x {w['length']} bp, seed {w['seed']}"
    else:
        nc = n if n is not None else w['n']
        codes, offsets, names, _ = make_contigs(nc, len_lo=w['len_lo'], len_hi=w['len_hi'], seed=w['seed'])
        desc = f"{name}: {nc} contigs log-uniform {w['len_lo']}-{w['len_hi']} bp, families geometric(0.2) <= 20, seed {w['seed']}"
    return codes, offsets, names, desc


def family_pairs(n_families, members):
    """All within-family pairs as a structured (a > b) array: what the prefilter is expected
    to pass on random-ancestor data."""
    out = []
    for f in range(n_families):
        base = f * members
        for i in range(members):
            for j in range(i):
                out.append((base + i, base + j, 0))
    return np.array(out, dtype=[('a', '<u4'), ('b', '<u4'), ('shared', '<u4')])


def sha256(codes, offsets):
    """Digest of a set: offsets (int64 LE) followed by the base codes."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(offsets, dtype='<i8').tobytes())
    h.update(np.ascontiguousarray(codes, dtype=np.uint8).tobytes())
    return h.hexdigest()


def write_fasta(path, codes, offsets, names, width=60):
    lut = np.frombuffer(b'ACGTN', dtype=np.uint8)
    with open(path, 'wb') as fh:
        for i, nm in enumerate(names):
            s = lut[np.minimum(codes[offsets[i]:offsets[i + 1]], 4)]
            n = len(s)
            full = n // width
            body = np.empty(n + full + (1 if n % width else 0), dtype=np.uint8)
            if full:
                blk = body[:full * (width + 1)].reshape(full, width + 1)
                blk[:, :width] = s[:full * width].reshape(full, width)
                blk[:, width] = 10
            if n % width:
                body[full * (width + 1):-1] = s[full * width:]
                body[-1] = 10
            fh.write(b'>' + nm.encode() + b'\n')
            fh.write(body.tobytes())
