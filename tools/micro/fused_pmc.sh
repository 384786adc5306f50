#!/bin/bash
# Developer tool: counters of the fused build+parse experiment (VG_LZ_FUSED=1) at phage-100k, one --pmc set per run;
# tools/pmc_summary.py turns them into gpurun_out/pmc_fused/summary/*.json (compare with profiles/r04_pmc_counters_phage-100k.json)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_fused_phage-100k; mkdir -p "$OUT"; export TMPDIR=/tmp
export VG_DEV_SWITCHES=1 VG_LZ_FUSED=1
CMD="python $REPO/bench.py --workload phage-100k --steps 1 --warmup 1 --no-cpu-baseline --no-cli-wall"
cd /tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/set$i" -- $CMD > "$OUT/set$i.log" 2>&1
done
cd "$REPO"
python tools/pmc_summary.py "$OUT" fused phage-100k
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete; find "$OUT" -name '*counter_collection.csv' -size +8M -delete
