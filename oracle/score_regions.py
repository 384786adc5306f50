#!/usr/bin/env python3
"""Score a region table (lzfit output) against the golden example/output/ani.aln.tsv.

Test infrastructure only.  Prints: exact regions / golden regions, per-field agreement,
and the number of ordered pairs whose (sum nt_match, sum alnlen, n_regions) agree.
"""
import sys
from collections import defaultdict


def load(path):
    d = defaultdict(list)
    with open(path) as fh:
        next(fh)
        for line in fh:
            c = line.rstrip('\n').split('\t')
            q, r = c[0], c[1]
            d[(q, r)].append(tuple(int(x) for x in c[3:10]) + (c[2],))
    return d


def main():
    gold = load(sys.argv[1])
    mine = load(sys.argv[2])
    verbose = len(sys.argv) > 3
    n_gold = sum(len(v) for v in gold.values())
    n_mine = sum(len(v) for v in mine.values())
    exact = 0
    q_ok = 0        # qstart,qend equal
    qr_ok = 0       # + rstart,rend
    qstart_ok = 0
    pairs_ok = 0
    pid_ok = 0
    bad_pairs = []
    for key in sorted(set(gold) | set(mine)):
        g = gold.get(key, [])
        m = mine.get(key, [])
        gs = set(g)
        ms = set(m)
        exact += len(gs & ms)
        gq = {(x[1], x[2]): x for x in g}
        mq = {(x[1], x[2]): x for x in m}
        for k in gq:
            if k in mq:
                q_ok += 1
                if gq[k][3:5] == mq[k][3:5]:
                    qr_ok += 1
                if gq[k][:7] == mq[k][:7] and gq[k][7] == mq[k][7]:
                    pid_ok += 1
        gstart = {x[1] for x in g}
        qstart_ok += sum(1 for x in m if x[1] in gstart)
        sg = (sum(x[5] for x in g), sum(x[0] for x in g), len(g))
        sm = (sum(x[5] for x in m), sum(x[0] for x in m), len(m))
        if sg == sm:
            pairs_ok += 1
        else:
            bad_pairs.append((key, sg, sm))
    print(f'golden regions {n_gold}  mine {n_mine}')
    print(f'exact (all 7 ints + pident) {exact}/{n_gold} = {exact / n_gold:.4f}')
    print(f'qstart matches {qstart_ok}  (qstart,qend) {q_ok}  +(rstart,rend) {qr_ok}')
    print(f'pair integer sums equal {pairs_ok}/{len(gold)}')
    if verbose:
        for key, sg, sm in bad_pairs[:200]:
            print(key, 'gold', sg, 'mine', sm)


if __name__ == '__main__':
    main()
