"""Synthetic phage-like genome sets (SURVEY §8d / BASELINE.md config table).

families x members: each family has a uniform-random ACGT ancestor; every member is the
ancestor with a per-member substitution rate p ~ U(p_lo, p_hi), a few indels of length
U(1, 50), with probability 0.2 one inversion of 1-5 kb and with probability 0.2 one
translocation.  Deterministic for a given seed (numpy PCG64).
"""
import numpy as np


def _mutate(rng, anc, p_lo, p_hi, n_indels):
    g = anc.copy()
    p = rng.uniform(p_lo, p_hi)
    mask = rng.random(len(g)) < p
    g[mask] = (g[mask] + rng.integers(1, 4, size=int(mask.sum()), dtype=np.uint8)) & 3
    for _ in range(n_indels):
        pos = int(rng.integers(0, len(g)))
        ln = int(rng.integers(1, 51))
        if rng.random() < 0.5:
            g = np.concatenate([g[:pos], g[pos + ln:]])
        else:
            g = np.concatenate([g[:pos], rng.integers(0, 4, size=ln, dtype=np.uint8), g[pos:]])
    if rng.random() < 0.2 and len(g) > 6000:
        ln = int(rng.integers(1000, 5001))
        pos = int(rng.integers(0, len(g) - ln))
        g[pos:pos + ln] = (3 - g[pos:pos + ln])[::-1]
    if rng.random() < 0.2 and len(g) > 6000:
        ln = int(rng.integers(1000, 5001))
        pos = int(rng.integers(0, len(g) - ln))
        seg = g[pos:pos + ln].copy()
        rest = np.concatenate([g[:pos], g[pos + ln:]])
        dst = int(rng.integers(0, len(rest)))
        g = np.concatenate([rest[:dst], seg, rest[dst:]])
    return g


def make_families(n_families, members, length=40000, seed=1, p_lo=0.005, p_hi=0.12, n_indels=5,
                  length_range=None):
    """-> (codes uint8[total], offsets int64[n+1], names list).

    length_range=(lo, hi): ancestor lengths log-uniform in [lo, hi] instead of `length`."""
    rng = np.random.default_rng(seed)
    seqs, names = [], []
    for f in range(n_families):
        if length_range:
            ln = int(np.exp(rng.uniform(np.log(length_range[0]), np.log(length_range[1]))))
        else:
            ln = length
        anc = rng.integers(0, 4, size=ln, dtype=np.uint8)
        for m in range(members):
            seqs.append(_mutate(rng, anc, p_lo, p_hi, n_indels))
            names.append(f'fam{f:05d}_m{m:02d}')
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
    return codes, offsets, names


def make_contigs(n_contigs, len_lo=5000, len_hi=200000, max_family=20, seed=2, p_lo=0.005, p_hi=0.12, n_indels=5):
    """IMG/VR-like set (BASELINE configs[2] / [4]): ancestor lengths log-uniform in [len_lo, len_hi],
    family sizes geometric(0.2) capped at max_family, same mutation model as make_families.
    -> (codes, offsets, names, family id per contig)."""
    rng = np.random.default_rng(seed)
    seqs, names, fam_of = [], [], []
    fam = 0
    while len(seqs) < n_contigs:
        members = min(int(rng.geometric(0.2)), max_family, n_contigs - len(seqs))
        ln = int(np.exp(rng.uniform(np.log(len_lo), np.log(len_hi))))
        anc = rng.integers(0, 4, size=ln, dtype=np.uint8)
        for m in range(members):
            seqs.append(_mutate(rng, anc, p_lo, p_hi, n_indels))
            names.append(f'ctg{fam:06d}_{m:02d}'); fam_of.append(fam)
        fam += 1
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    return np.concatenate(seqs), offsets, names, np.array(fam_of, dtype=np.int64)


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


def write_fasta(path, codes, offsets, names, width=60):
    lut = np.frombuffer(b'ACGTN', dtype=np.uint8)
    with open(path, 'wb') as fh:
        for i, nm in enumerate(names):
            s = lut[np.minimum(codes[offsets[i]:offsets[i + 1]], 4)].tobytes()
            fh.write(b'>' + nm.encode() + b'\n')
            for o in range(0, len(s), width):
                fh.write(s[o:o + width] + b'\n')
