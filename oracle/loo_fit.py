#!/usr/bin/env python3
"""Leave-one-genome-out report of the LZ rule fit (TEST INFRASTRUCTURE ONLY; needs oracle/_build/lzfit).

The constants of R2-R9 (the knobs of vo_lz_variant) were chosen on the 132 ordered pairs of the reference's one example.
For every genome G of the 12: the pairs that do not involve G (110 of 132) choose every knob again -- one knob at a
time against the final values of the others, best = most golden regions reproduced minus surplus regions, ties keep the
final value -- and the pairs that DO involve G (22) are then scored with the knobs so chosen.  A rule whose constant is
held by the evidence of many pairs survives every fold; one that rests on a single event is lost in the folds that hold
that event out.  Also printed per knob: how many pairs and regions tell its final value from the best alternative.

  python oracle/loo_fit.py tests/golden/example/multifasta.fna tests/golden/example/output/ani.aln.tsv
"""
import subprocess
import sys
import pathlib
from collections import defaultdict

ROOT = pathlib.Path(__file__).resolve().parent
LZFIT = ROOT / '_build' / 'lzfit'

# final value first, then the alternatives the fit looked at (vclust_oracle.h: vo_lz_variant)
KNOBS = {
    'anchor_while_predicting': [3, 0, 1, 2],
    'anchor_margin': [-1, 4, 5, 7, 8],          # -1 = msl - 1 = 6
    'weak_seed_ratio': [3, 0, 2, 4],
    'bwd_bound_kept': [1, 0],
    'bwd_exact_first': [1, 0],
    'seed_window': [2, 0, 1],
    'seed_choice': [3, 0, 1, 2],
    'lit_reset_ge': [0, 1],
    'gap_mode': [4, 0, 1, 2, 3],
    'fwd_after_close': [1, 0],
    'loop_le': [0, 1],
    'anchor_tie': [0, 1],
    'rend_mode': [6, 0, 1, 2, 3, 4, 5],
    'sep_len': [0, 1],                           # 0 = mrd + mqd + 1
}
RULE = {'anchor_while_predicting': 'R2/R3', 'anchor_margin': 'R3', 'weak_seed_ratio': 'R3', 'bwd_bound_kept': 'R5', 'bwd_exact_first': 'R5',
        'seed_window': 'R3', 'seed_choice': 'R3', 'lit_reset_ge': 'R6', 'gap_mode': 'R7', 'fwd_after_close': 'R4', 'loop_le': 'R2',
        'anchor_tie': 'R2', 'rend_mode': 'R9', 'sep_len': 'R1'}


def load(text):
    d = defaultdict(set)
    for line in text.splitlines()[1:]:
        c = line.split('\t')
        d[(c[0], c[1])].add(tuple(c[2:10]))
    return d


def run(fasta, cfg):
    args = [str(LZFIT), fasta] + [f'{k}={v}' for k, v in cfg.items()]
    return load(subprocess.run(args, check=True, capture_output=True, text=True).stdout)


def per_pair(gold, mine):
    """pair -> (golden regions reproduced, surplus regions)"""
    return {p: (len(gold[p] & mine.get(p, set())), len(mine.get(p, set()) - gold[p])) for p in gold}


def main():
    fasta, gold_path = sys.argv[1], sys.argv[2]
    gold = load(open(gold_path).read())
    genomes = sorted({p[0] for p in gold} | {p[1] for p in gold})
    final = {k: v[0] for k, v in KNOBS.items()}
    base = per_pair(gold, run(fasta, {}))
    assert all(base[p] == (len(gold[p]), 0) for p in gold), 'the final knobs no longer reproduce the goldens'
    single = {}                      # (knob, value) -> per-pair scores with that one knob changed
    for k, vals in KNOBS.items():
        for v in vals[1:]:
            single[(k, v)] = per_pair(gold, run(fasta, {k: v}))
    print('## what holds every constant (final value against each alternative, all 132 pairs)\n')
    print('| rule | knob | final | alternative | pairs that tell them apart | golden regions lost | surplus regions |')
    print('|---|---|---|---|---|---|---|')
    for (k, v), sc in single.items():
        diff = [p for p in gold if sc[p] != base[p]]
        lost = sum(base[p][0] - sc[p][0] for p in gold)
        print(f'| {RULE[k]} | {k} | {final[k]} | {v} | {len(diff)} | {lost} | {sum(sc[p][1] for p in gold)} |')
    print('\n## leave one genome out\n')
    print('| genome held out | golden regions in its 22 pairs | reproduced by the knobs the other 110 pairs choose | knobs chosen differently |')
    print('|---|---|---|---|')
    tot_g = tot_ok = 0
    for g in genomes:
        train = [p for p in gold if g not in p]
        test = [p for p in gold if g in p]
        chosen = {}
        for k, vals in KNOBS.items():
            best_v, best_s = vals[0], sum(base[p][0] - base[p][1] for p in train)
            for v in vals[1:]:
                s = sum(single[(k, v)][p][0] - single[(k, v)][p][1] for p in train)
                if s > best_s:
                    best_v, best_s = v, s
            if best_v != vals[0]:
                chosen[k] = best_v
        # a knob whose alternatives tie with the final value on the training pairs is NOT held by them either
        free = []
        for k, vals in KNOBS.items():
            s0 = sum(base[p][0] - base[p][1] for p in train)
            ties = [v for v in vals[1:] if sum(single[(k, v)][p][0] - single[(k, v)][p][1] for p in train) == s0]
            if ties:
                free.append(f'{k} (ties with {",".join(map(str, ties))})')
        sc = per_pair(gold, run(fasta, chosen)) if chosen else base
        n_g = sum(len(gold[p]) for p in test)
        n_ok = sum(sc[p][0] for p in test)
        # worst case over the ties: the tied alternative that loses most on the held-out pairs
        worst, worst_surplus = n_ok, sum(sc[p][1] for p in test)
        for k, vals in KNOBS.items():
            s0 = sum(base[p][0] - base[p][1] for p in train)
            for v in vals[1:]:
                if sum(single[(k, v)][p][0] - single[(k, v)][p][1] for p in train) == s0:
                    worst = min(worst, sum(single[(k, v)][p][0] for p in test))
                    worst_surplus = max(worst_surplus, sum(single[(k, v)][p][1] for p in test))
        tot_g += n_g; tot_ok += n_ok
        note = ', '.join(f'{k}={v}' for k, v in chosen.items()) or 'none'
        if free:
            note += '; undetermined without this genome: ' + '; '.join(free) + f' -> {worst} of {n_g} (+ {worst_surplus} surplus) with the worst tied choice'
        print(f'| {g} | {n_g} | {n_ok} | {note} |')
    print(f'| all folds | {tot_g} | {tot_ok} | |')


if __name__ == '__main__':
    main()
