#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/collect_profiles.sh into the two files kept under profiles/:
  <tag>_bench_kernel_stats_<workload>.csv   (rocprofv3 --kernel-trace --stats summary, copied as is)
  <tag>_pmc_hbm_traffic_<workload>.json     (FETCH_SIZE / WRITE_SIZE per kernel and launch, separate passes)
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3.  On gfx950 FETCH_SIZE tallies the 128-byte requests of
wide streaming reads at 64 bytes (MI355X_MICROARCH.md "HBM": double it before comparing with a byte count) -- but
NOT the 64-byte requests of narrow random reads: profiles/r03_pmc_calibration.json (tools/micro/fetch_calib.hip)
measures 8.0 bytes of FETCH_SIZE per streamed 16-byte access and 64.0 per random 16-byte (or 4-byte) read, and 32.0
bytes of WRITE_SIZE per random 4-byte write.  So the factor is per kernel: 1 for the kernels whose reads are random
probes / gathers (RANDOM_READERS below), 2 for the streaming ones;
hbm_bytes_per_launch = (factor * FETCH_SIZE + WRITE_SIZE) * 1024, with the factor and the raw sum kept beside it.
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


# kernels whose FETCH_SIZE is dominated by random 64-byte requests (index probes, list gathers, genome look-ups)
RANDOM_READERS = ('k_lz_parse', 'k_spgemm', 'k_bucket_runs', 'k_bucket_big')


def fetch_factor(kernel):
    return 1 if kernel.startswith(RANDOM_READERS) else 2


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
                          fetch_factor=fetch_factor(k), hbm_bytes_per_launch=round((fetch_factor(k) * fk + wk) * 1024),
                          hbm_bytes_per_launch_raw=round((fk + wk) * 1024))
    doc = dict(workload=workload, command=f'python bench.py --workload ... --steps {steps} --warmup 1 --no-cpu-baseline --no-cli-wall --no-other-workloads --no-out-aln (1 MI355X)',
               steps_in_run=steps + 1,
               correction='separate --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC slot limits); rocprofv3 units are KB; '
                          'hbm_bytes_per_launch = (fetch_factor x FETCH_SIZE + WRITE_SIZE) x 1024: factor 2 for streaming reads (gfx950 '
                          'tallies their 128-B requests at 64 B, MI355X_MICROARCH.md HBM section), factor 1 for kernels of random '
                          '64-B requests -- calibrated in profiles/r03_pmc_calibration.json',
               kernels=kernels)
    with open(os.path.join(dst, f'{tag}_pmc_hbm_traffic_{wtag}.json'), 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in kernels.items()}, indent=1))


if __name__ == '__main__':
    main()
