#!/bin/bash
# Developer tool (GPU box): address-translation and fabric counters of the scattering kernels of the prefilter in their
# placement states (tools/micro/placement_states.py under rocprofv3 --pmc, one counter set per run; every run re-allocates the
# workspace REPS times, so both states occur in it and are told apart by the kernel durations).  usage: placement_pmc.sh <out-dir>
set -u
REPO=$(pwd); OUT=$REPO/${1:-gpurun_out/r5_placement}; export TMPDIR=/tmp REPS=${REPS:-8} NF=${NF:-10000}
mkdir -p "$OUT"; cd /tmp
VG_ALLOC_TRACE=1 python $REPO/tools/micro/placement_states.py > "$OUT/plain.log" 2> "$OUT/plain.err"
HOLD_GB=200 VG_ALLOC_TRACE=1 python $REPO/tools/micro/placement_states.py > "$OUT/plain_after_200GiB.log" 2> "$OUT/plain_after_200GiB.err"
i=0
for SET in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_BUBBLE_sum TCC_REQ_sum" \
           "TCC_EA0_WRREQ TCC_EA0_RDREQ" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/set$i" -- python $REPO/tools/micro/placement_states.py > "$OUT/set$i.log" 2>&1
  echo "$SET" > "$OUT/set$i.names"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
out = sys.argv[1]; doc = {}
KERN = ('k_bucket_runs', 'k_part_scatter_dense', 'k_part_scatter2_narrow', 'k_part_count2', 'k_spgemm')
for d in sorted(glob.glob(os.path.join(out, 'set*'))):
    if not os.path.isdir(d): continue
    dur = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f, newline='')):
            dur[row['Dispatch_Id']] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
    per = defaultdict(lambda: defaultdict(dict))           # kernel -> dispatch -> counter -> value (summed over instances)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f, newline='')):
            m = re.search(r'k_\w+', row['Kernel_Name'])
            if not m or not m.group(0).startswith(KERN): continue
            name = next(k for k in KERN if m.group(0).startswith(k))
            c = per[name][row['Dispatch_Id']]
            c[row['Counter_Name']] = c.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
            c.setdefault('_vals_' + row['Counter_Name'], []).append(float(row['Counter_Value']))
    for name, disp in per.items():
        rows = [(dur.get(did), cs) for did, cs in disp.items() if dur.get(did) and dur[did] > 1.0]      # the full-size launches only
        if len(rows) < 4: continue
        rows.sort(key=lambda r: r[0])
        ds = [r[0] for r in rows]
        # split at the largest gap between consecutive durations
        gaps = [(ds[i + 1] - ds[i], i) for i in range(len(ds) - 1)]; g, gi = max(gaps)
        fast, slow = rows[:gi + 1], rows[gi + 1:]
        e = doc.setdefault(name, {})
        e.setdefault('launches', len(rows))
        for tag, grp in (('fast', fast), ('slow', slow)):
            ee = e.setdefault(tag, {})
            ee['ms_' + os.path.basename(d)] = [round(min(r[0] for r in grp), 2), round(max(r[0] for r in grp), 2), len(grp)]
            for cname in grp[0][1]:
                if cname.startswith('_vals_'):
                    v = [x for r in grp for x in r[1][cname]]
                    if len(grp[0][1][cname]) > 1: ee[cname[6:] + '_per_instance_min_max'] = [min(v), max(v)]
                    continue
                ee[cname] = round(sum(r[1][cname] for r in grp) / len(grp), 1)
json.dump(doc, open(os.path.join(out, 'placement_pmc.json'), 'w'), indent=1, sort_keys=True)
print(json.dumps(doc, indent=1, sort_keys=True))
PY
find "$OUT" -name '*.csv' -size +4M -delete
