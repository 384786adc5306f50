#!/bin/bash
# Developer tool (GPU box): SQ / TCC counters of the parse and index-build kernels
# (tools/micro/parse_ab.py under rocprofv3 --pmc, one counter set per run).  usage: parse_pmc.sh <out-dir> [NF]
# KERNELS="default fused": a variant named `fused` runs with VG_LZ_INDEX=fused (the fused-slot probe experiment of round 6)
set -u
REPO=$(pwd); OUT=$REPO/${1:-gpurun_out/r5_parse_pmc}; export NF=${2:-10000} REPS=1 VG_DEV_SWITCHES=1 TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
# SHORT=1: the instruction counters and the L2 hit / miss / request counters only (two passes per variant)
if [ "${SHORT:-0}" = 1 ]; then
  SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum")
else
  SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU" \
        "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum")
fi
for K in ${KERNELS:-default}; do
  i=0
  for SET in "${SETS[@]}"; do
    i=$((i+1))
    IDX=default; KK=$K; if [ "$K" = fused ]; then IDX=fused; KK=default; fi
    VG_LZ_INDEX=$IDX VG_LZ_KERNEL=$KK rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/$K/set$i" -- python $REPO/tools/micro/parse_ab.py > "$OUT/$K.set$i.log" 2>&1
  done
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
out = sys.argv[1]; doc = {}
for k in [d for d in sorted(os.listdir(out)) if os.path.isdir(os.path.join(out, d))]:
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, k, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f, newline='')):
            m = re.search(r'k_lz_parse\w*|k_build_index\w*|k_index_fuse\w*', row['Kernel_Name']); name = m.group(0) if m else ''
            if name: per[name][row['Counter_Name']].append(float(row['Counter_Value']))
    # the LAST launch of each kernel (the timed repetition; the first is the warm-up)
    doc[k] = {n: {c: v[-1] for c, v in cs.items()} for n, cs in per.items()}
    for f in glob.glob(os.path.join(out, k, '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f, newline='')):
            m = re.search(r'k_lz_parse\w*|k_build_index\w*|k_index_fuse\w*', row['Kernel_Name']); name = m.group(0) if m else ''
            if name in doc[k]: doc[k][name]['last_launch_ms'] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
        break
json.dump(doc, open(os.path.join(out, 'parse_pmc.json'), 'w'), indent=1, sort_keys=True)
print(json.dumps(doc, indent=1, sort_keys=True))
PY
find "$OUT" -name '*.csv' -size +4M -delete
