#!/usr/bin/env python3
"""Summarise the counter passes of tools/collect_pmc.sh: per kernel and launch, every counter that was collected,
plus the FETCH_SIZE / WRITE_SIZE calibration of tools/micro/fetch_calib.hip (bytes the counter reports per access of
a known pattern).  Output: <out>/summary/<tag>_pmc_counters_<workload>.json and <tag>_pmc_calibration.json."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_summary import short  # noqa: E402


def collect(d):
    per = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                k = short(row['Kernel_Name'])
                c = per[k][row['Counter_Name']]
                c[0] += float(row['Counter_Value']); c[1] += 1
    return per


def main():
    out, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    dst = os.path.join(out, 'summary'); os.makedirs(dst, exist_ok=True)
    kernels = defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(out, 'set*'))):
        if not os.path.isdir(d):
            continue
        for k, cs in collect(d).items():
            for c, (v, n) in cs.items():
                kernels[k][c] = round(v / n, 1)
                kernels[k].setdefault('launches_seen', n)
    for k, e in kernels.items():
        wc, av, aa = e.get('SQ_WAVE_CYCLES'), e.get('SQ_ACTIVE_INST_VALU'), e.get('SQ_ACTIVE_INST_ANY')
        if wc and av is not None:
            e['valu_active_share_of_wave_cycles'] = round(av / wc, 4)
        if wc and aa is not None:
            e['any_inst_active_share_of_wave_cycles'] = round(aa / wc, 4)
        if wc and e.get('SQ_WAIT_ANY') is not None:
            e['wait_any_share_of_wave_cycles'] = round(e['SQ_WAIT_ANY'] / wc, 4)
        if wc and e.get('SQ_WAIT_INST_ANY') is not None:
            e['wait_inst_any_share_of_wave_cycles'] = round(e['SQ_WAIT_INST_ANY'] / wc, 4)
        if e.get('SQ_BUSY_CYCLES') and wc:
            e['mean_waves_in_flight_per_SQ_cycle'] = round(wc / e['SQ_BUSY_CYCLES'], 2)
        h, m = e.get('TCC_HIT_sum'), e.get('TCC_MISS_sum')
        if h is not None and m is not None and h + m > 0:
            e['l2_hit_rate'] = round(h / (h + m), 4)
    doc = dict(workload=wl, command='rocprofv3 --pmc <one set per run> --kernel-trace -- python bench.py --workload %s --steps 1 --warmup 1 (1 MI355X)' % wl,
               note='per-launch averages; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); '
                    'FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (see the calibration file for what a KB means per access pattern)',
               kernels=kernels)
    with open(os.path.join(dst, f'{tag}_pmc_counters_{wl}.json'), 'w') as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    # calibration
    cal = {}
    log = {}
    for line in open(os.path.join(out, 'calib_fetch.log'), errors='replace'):
        m = re.match(r'(k_\w+)\s+accesses (\d+) useful_bytes (\d+) ms ([\d.]+)', line)
        if m:
            log[m.group(1)] = dict(accesses=int(m.group(2)), useful_bytes=int(m.group(3)), ms=float(m.group(4)))
    fetch = collect(os.path.join(out, 'calib_fetch')); write = collect(os.path.join(out, 'calib_write'))
    for k, e in log.items():
        f = fetch.get(k, {}).get('FETCH_SIZE', [0.0, 1]); w = write.get(k, {}).get('WRITE_SIZE', [0.0, 1])
        fb, wb = f[0] / max(f[1], 1) * 1024.0, w[0] / max(w[1], 1) * 1024.0
        n_acc = e['useful_bytes'] // (16 if '16' in k else 4)
        cal[k] = dict(e, accesses=n_acc, FETCH_SIZE_bytes=round(fb), WRITE_SIZE_bytes=round(wb),
                      fetch_bytes_per_access=round(fb / n_acc, 2), write_bytes_per_access=round(wb / n_acc, 2),
                      useful_GBps=round(e['useful_bytes'] / e['ms'] / 1e6, 1))
    with open(os.path.join(dst, f'{tag}_pmc_calibration.json'), 'w') as fh:
        json.dump(dict(pool='40 GiB', tool='tools/micro/fetch_calib.hip', kernels=cal,
                       note='FETCH_SIZE / WRITE_SIZE (KB x 1024) per access of a known pattern: k_stream16 = consecutive 16-byte reads, '
                            'k_random16 / k_random4r = random 16- / 4-byte reads of the pool, k_random4w = random 4-byte writes'), fh, indent=1)
    print(json.dumps({k: {c: v for c, v in e.items() if not c.startswith('SQ_INSTS')} for k, e in kernels.items() if 'lz_parse' in k or 'bucket_runs' in k}, indent=1))
    print(json.dumps(cal, indent=1))


if __name__ == '__main__':
    main()
