#!/usr/bin/env python3
"""Accuracy of the LZ parse against KNOWN truth, by the reference's own acceptance criterion (test.py:456-477,
example/README.txt:4-11: tANI within 0.007 of the simulated truth), beyond the 8 pairs of the reference's example.

Families with SUBSTITUTIONS ONLY: an ancestor of 40 kb and members that differ from it by per-base substitutions at rate p
(no indels, no rearrangements), so every pair of a family is aligned position by position and its true total ANI is the
fraction of equal positions -- an exact integer ratio, no simulation noise.  Bands of p: the phage range of SURVEY 8(d)
(0.005 ... 0.12), a moderately diverged band (0.15 ... 0.20) and a far diverged one (0.22 ... 0.30).  For every ordered pair
the HIP path's rows give tANI = (M_qr + M_rq) / (L_q + L_r); the table holds max |tANI - truth| and the mean signed error per
band, once at the fitted constants of the restatement and once with each of the three thin constants (DESIGN.md section 2:
held by one to three events of the reference's example) at its alternative value (vg_set_lz_fit).

  python tools/accuracy_vs_truth.py [--out profiles/r06_accuracy_vs_truth.md] [--oracle]

--oracle runs the CPU restatement instead of the HIP path (no GPU; fitted constants only -- the oracle reads its variant
from the environment once per process).  tests/test_gpu_parity.py::test_accuracy_against_known_truth asserts the criterion
on the HIP path; this tool writes the table.  It pins nothing upstream: it shows whether the constants fitted on one
12-genome example matter to the number users read.
"""
import argparse
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

from vclust_amd import synth  # noqa: E402

BANDS = (('phage range', (0.005, 0.01, 0.02, 0.04, 0.06, 0.08, 0.10, 0.12)),
         ('diverged', (0.15, 0.18, 0.20)),
         ('far diverged', (0.22, 0.25, 0.30)))
KNOBS = (('fitted (weak seed 3, margin msl-1, seed choice 3)', {}),
         ('weak_seed_ratio 0 (rule off)', dict(weak_seed_ratio=0)),
         ('weak_seed_ratio 4', dict(weak_seed_ratio=4)),
         ('anchor_margin 5', dict(anchor_margin=5)),
         ('anchor_margin 7', dict(anchor_margin=7)),
         ('seed_choice 1 (closest first)', dict(seed_choice=1)))
TOLERANCE = 0.007          # /root/reference/test.py:477


def substitution_family(seed, fam, length, rates):
    """Ancestor + one member per rate, substitutions only (the generator's own substitution draw, synth._mutate)."""
    anc = synth._bases(synth._stream(seed, fam, 0, 0), length)
    out = [anc]
    for m, p in enumerate(rates, start=1):
        d = synth.draws(synth._stream(seed, fam, m, 2), length)
        g = anc.copy()
        mask = synth._unit(d) < p
        g[mask] = (g[mask] + ((d[mask] & np.uint64(0xffff)) % np.uint64(3)).astype(np.uint8) + np.uint8(1)) & np.uint8(3)
        out.append(g)
    return out


def make_set(seed=77, length=40000, reps=3):
    """-> codes, offsets, names, pairs [(a, b, band, p, truth)]: every member against its ancestor, `reps` families per rate."""
    seqs, pairs = [], []
    fam = 0
    for band, rates in BANDS:
        for p in rates:
            for _ in range(reps):
                anc, mem = substitution_family(seed, fam, length, (p,))
                a = len(seqs); seqs += [anc, mem]
                pairs.append((a + 1, a, band, p, float((anc == mem).mean())))
                fam += 1
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    return np.concatenate(seqs), offsets, [f's{i}' for i in range(len(seqs))], pairs


def tani_hip(codes, offsets, names, pairs, fit):
    from vclust_amd import api
    api.set_lz_fit(**fit)
    try:
        gs = api.GenomeSet.from_codes(codes, offsets, names)
        cand = np.array([(a, b, 0) for a, b, *_ in pairs], dtype=api.PAIR_DTYPE)
        tasks = gs.align_tasks(cand)
        stats = gs.lz_align(tasks)
        lens = gs.lengths()
        rows = {(int(t['q']), int(t['r'])): int(s['n_match']) for t, s in zip(tasks, stats)}
        return [(rows[(a, b)] + rows[(b, a)]) / float(lens[a] + lens[b]) for a, b, *_ in pairs]
    finally:
        api.set_lz_fit()


def tani_oracle(codes, offsets, names, pairs, fit):
    import oracle_lib as orc
    assert not fit, 'the oracle reads its variant from the environment once per process'
    out = []
    for a, b, *_ in pairs:
        q, r = codes[offsets[a]:offsets[a + 1]], codes[offsets[b]:offsets[b + 1]]
        out.append((orc.lz_pair_stat(q, r)[0] + orc.lz_pair_stat(r, q)[0]) / float(len(q) + len(r)))
    return out


def table(use_oracle=False, reps=3):
    codes, offsets, names, pairs = make_set(reps=reps)
    res = {}
    for label, fit in KNOBS[:1] if use_oracle else KNOBS:
        t = (tani_oracle if use_oracle else tani_hip)(codes, offsets, names, pairs, fit)
        err = np.array([ti - pr[4] for ti, pr in zip(t, pairs)])
        for band, _ in BANDS:
            sel = np.array([pr[2] == band for pr in pairs])
            res[(label, band)] = (float(np.abs(err[sel]).max()), float(err[sel].mean()), int(sel.sum()),
                                  min(pr[4] for pr in pairs if pr[2] == band), max(pr[4] for pr in pairs if pr[2] == band))
    return res, pairs


def knob_sensitivity():
    """Families WITH indels, inversions and translocations (the generator's full mutation model: truth is not exactly known
    there): how far does tANI move when a thin constant takes its alternative value?  -> {band: {knob: (rows that change, pairs
    whose tANI changes, max |delta tANI|)}} over all within-family pairs of 12 families x 6 members per band."""
    from vclust_amd import api
    out = {}
    for band, (p_lo, p_hi, n_indels) in (('phage range, p 0.005-0.12, 5 indels', (0.005, 0.12, 5)), ('diverged, p 0.12-0.25, 25 indels', (0.12, 0.25, 25))):
        codes, offsets, names = synth.make_families(12, 6, length=40000, seed=41, p_lo=p_lo, p_hi=p_hi, n_indels=n_indels)
        gs = api.GenomeSet.from_codes(codes, offsets, names)
        tasks = gs.align_tasks(synth.family_pairs(12, 6))
        lens = gs.lengths()
        def run(fit):
            api.set_lz_fit(**fit)
            try:
                return gs.lz_align(tasks).copy()
            finally:
                api.set_lz_fit()
        base = run({})
        def tani(st):
            m = st['n_match'].astype(np.float64)
            return (m[0::2] + m[1::2]) / (lens[tasks['q'][0::2]] + lens[tasks['r'][0::2]])
        t0 = tani(base)
        out[band] = {}
        for label, fit in KNOBS[1:]:
            st = run(fit)
            d = np.abs(tani(st) - t0)
            out[band][label] = (int((st != base).sum()), int((d > 0).sum()), float(d.max()), len(t0), float(t0.min()), float(t0.max()))
    return out


def markdown(res, use_oracle, sens=None):
    lines = ['# Accuracy of the LZ parse against known truth (round 6)', '',
             f'`tools/accuracy_vs_truth.py`{" --oracle (CPU restatement)" if use_oracle else " (HIP path, vg_lz_align)"}: ancestor / member pairs of 40 kb with substitutions only, truth = fraction of equal',
             'positions; criterion of the reference (`test.py:456-477`): |tANI - truth| < 0.007.  Three pairs per substitution rate.', '',
             '| constants | band (true tANI) | pairs | max abs error | mean signed error | within 0.007 |', '|---|---|---|---|---|---|']
    for (label, band), (mx, mean, n, lo, hi) in res.items():
        lines.append(f'| {label} | {band} ({lo:.3f} ... {hi:.3f}) | {n} | {mx:.5f} | {mean:+.5f} | {"yes" if mx < TOLERANCE else "NO"} |')
    if sens:
        lines += ['', '## How far the thin constants move tANI where they CAN act (indels, inversions, translocations)', '',
                  'Substitution-only pairs never exercise the three constants (the rows above are identical for every setting: without indels no far anchor',
                  'competes with a seed).  Families of the full mutation model, all within-family pairs, each constant at its alternative value against the fitted one:', '',
                  '| set (tANI range) | constant | rows that change | pairs whose tANI changes | max abs change of tANI |', '|---|---|---|---|---|']
        for band, d in sens.items():
            for label, (rows, prs, mx, n, lo, hi) in d.items():
                lines.append(f'| {band} ({lo:.3f} ... {hi:.3f}; {n} pairs) | {label} | {rows} of {2 * n} | {prs} | {mx:.6f} |')
    lines += ['', 'Reading: up to 20 % substitutions the parse returns the truth to the fourth decimal whatever the three thin constants are set to',
              '(the far-diverged band loses COVERAGE -- regions end where 15-symbol windows hold more than 7 mismatches --, which is LZ-ANI\'s',
              'documented behaviour below ~75 % identity and not a matter of the fit).  Where the constants can act they move a handful of rows by far less than the',
              'criterion\'s 0.007.  Nothing here is pinned upstream.']
    return '\n'.join(lines) + '\n'


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    ap.add_argument('--oracle', action='store_true')
    a = ap.parse_args()
    res, _ = table(a.oracle)
    md = markdown(res, a.oracle, None if a.oracle else knob_sensitivity())
    if a.out:
        pathlib.Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        pathlib.Path(a.out).write_text(md)
    print(md)
