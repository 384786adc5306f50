#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes of the default bench.
# Usage: tools/collect_profiles.sh <round-tag>   -> gpurun_out/prof_<tag>/{stats,fetch,write}/ + summaries
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
cd "$REPO"
python tools/profile_summary.py "$OUT" "$TAG"
# keep the merge small: raw traces are large
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
