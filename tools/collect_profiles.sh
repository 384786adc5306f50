#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes of the bench.
# Usage: tools/collect_profiles.sh <round-tag> [workload] [n] [steps]
#   -> gpurun_out/prof_<tag>_<workload>/{stats,fetch,write}/ + summary/ (copy summary/* into profiles/)
set -u
TAG=${1:-r03}
WL=${2:-phage-100k}
N=${3:-}
STEPS=${4:-3}
REPO=$(pwd)
NAME=$WL${N:+/$N}
OUT=$REPO/gpurun_out/prof_${TAG}_${WL}${N:+_$N}
mkdir -p "$OUT"
export TMPDIR=/tmp
# (one prefilter pass per step in the traces: the placement trials of a process's first pass -- DESIGN section 4 -- are off)
export VG_DEV_SWITCHES=1 VG_PLACEMENT_TRIALS=1
CMD="python $REPO/bench.py --workload $WL ${N:+--count $N} --steps $STEPS --warmup 1 --no-cpu-baseline --no-cli-wall --no-other-workloads --no-out-aln"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
cd "$REPO"
python tools/profile_summary.py "$OUT" "$TAG" "$NAME" "$STEPS"
# keep the merge small: raw traces are large
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
