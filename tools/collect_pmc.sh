#!/bin/bash
# Run on the GPU box (via gpurun): SQ / TCC counter passes of the bench (one --pmc set per run, never combined with
# a trace domain other than --kernel-trace) and the FETCH_SIZE / WRITE_SIZE calibration of tools/micro/fetch_calib.hip.
# Usage: tools/collect_pmc.sh <round-tag> [workload]   -> gpurun_out/pmc_<tag>_<workload>/ + summary json (copy into profiles/)
set -u
TAG=${1:-r03}
WL=${2:-phage-100k}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
export VG_DEV_SWITCHES=1 VG_PLACEMENT_TRIALS=1      # (one prefilter pass per step in the traces)
CMD="python $REPO/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-cli-wall --no-other-workloads --no-out-aln"
cd /tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/set$i" -- $CMD > "$OUT/set$i.log" 2>&1
  echo "$SET" > "$OUT/set$i.names"
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $REPO/tools/micro/fetch_calib.hip
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/calib_fetch" -- /tmp/fetch_calib 40 1000 > "$OUT/calib_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/calib_write" -- /tmp/fetch_calib 40 1000 > "$OUT/calib_write.log" 2>&1
cd "$REPO"
python tools/pmc_summary.py "$OUT" "$TAG" "$WL"
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
