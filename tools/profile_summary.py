#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/collect_profiles.sh into the two files kept under profiles/:
  <tag>_bench_kernel_stats_<workload>.csv   (rocprofv3 --kernel-trace --stats summary, copied as is)
  <tag>_pmc_hbm_traffic_<workload>.json     (FETCH_SIZE / WRITE_SIZE per kernel and launch, separate passes)
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes (MI355X_MICROARCH.md "HBM": double it before comparing with a byte count), so
hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; the raw sum is kept beside it.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    if 'rocprim' in name:
        for key in ('onesweep_histograms', 'onesweep_iteration', 'radix_sort', 'scan', 'transform'):
            if key in name:
                return 'rocprim::' + key
        return 'rocprim::other'
    return name.split('(')[0].split('<')[0].strip()


def counters(d, counter):
    per = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                k = short(row['Kernel_Name'])
                per[k][0] += float(row['Counter_Value'])
                per[k][1] += 1
    return per


def main():
    out, tag = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else 'phage-100k'
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    wtag = workload.replace('/', '_')
    dst = os.path.join(out, 'summary')
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(out, 'stats', '**', '*kernel_stats.csv'), recursive=True):
        shutil.copy(f, os.path.join(dst, f'{tag}_bench_kernel_stats_{wtag}.csv'))
    fetch, write = counters(os.path.join(out, 'fetch'), 'FETCH_SIZE'), counters(os.path.join(out, 'write'), 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0]); w, nw = write.get(k, [0.0, 0])
        fk = f / nf if nf else 0.0
        wk = w / nw if nw else 0.0
        kernels[k] = dict(launches=max(nf, nw), FETCH_SIZE_KB_per_launch=round(fk, 1), WRITE_SIZE_KB_per_launch=round(wk, 1),
                          hbm_bytes_per_launch=round((2 * fk + wk) * 1024), hbm_bytes_per_launch_raw=round((fk + wk) * 1024))
    doc = dict(workload=workload, command=f'python bench.py --workload ... --steps {steps} --warmup 1 --no-cpu-baseline --no-cli-wall (1 MI355X)',
               steps_in_run=steps + 1,
               correction='separate --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC slot limits); rocprofv3 units are KB; '
                          'hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, '
                          'MI355X_MICROARCH.md HBM section; an upper bound for narrow random reads)',
               kernels=kernels)
    with open(os.path.join(dst, f'{tag}_pmc_hbm_traffic_{wtag}.json'), 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in kernels.items()}, indent=1))


if __name__ == '__main__':
    main()
