#!/usr/bin/env python3
"""Developer tool: where is the GPU idle inside one pass?  Reads a rocprofv3 --kernel-trace CSV (output of
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/micro/shard_pass.py` or of a bench run), takes the LAST
`--passes` repetitions of the kernel sequence that starts at --first (a kernel-name substring) and prints, per pass, the busy
time, the idle time between consecutive kernels and the five largest gaps with the kernels on either side.
  python tools/micro/pass_gaps.py DIR --first k_slice_scan --last k_spgemm"""
import argparse, csv, glob, os, re, sys
ap = argparse.ArgumentParser()
ap.add_argument('dir'); ap.add_argument('--first', default='k_part_count'); ap.add_argument('--last', default='k_spgemm'); ap.add_argument('--passes', type=int, default=2)
a = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(a.dir, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f, newline='')):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), (re.search(r'k_\w+(<[^>(]*>)?|rocprim::\w+|__amd_\w+', r['Kernel_Name']) or re.search(r'\S+', r['Kernel_Name'])).group(0)[:60]))
rows.sort()
starts = [i for i, r in enumerate(rows) if a.first in r[2] and (i == 0 or a.first not in rows[i - 1][2])]
for si in starts[-a.passes:]:
    ei = next((j for j in range(si, len(rows)) if a.last in rows[j][2] and (j + 1 == len(rows) or a.last not in rows[j + 1][2])), len(rows) - 1)
    seq = rows[si:ei + 1]
    busy = sum(e - s for s, e, _ in seq) / 1e6
    span = (seq[-1][1] - seq[0][0]) / 1e6
    gaps = sorted(((seq[i + 1][0] - max(x[1] for x in seq[:i + 1])) / 1e6, seq[i][2], seq[i + 1][2]) for i in range(len(seq) - 1))
    idle = sum(max(0.0, g[0]) for g in gaps)
    print(f'pass of {len(seq)} kernels: span {span:.2f} ms, kernels {busy:.2f} ms, idle between kernels {idle:.2f} ms')
    for g in gaps[-6:][::-1]:
        print(f'   gap {g[0]:.3f} ms  after {g[1]}  before {g[2]}')
