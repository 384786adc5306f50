#!/bin/bash
set -u
OUT=gpurun_out/r06d; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
(cd /tmp && SCAN=sliced WORLD=8 REPS=3 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/shard8 -- python $REPO/tools/micro/shard_pass.py > $REPO/$OUT/shard8.log 2>&1)
python tools/micro/pass_gaps.py $OUT/shard8 --first k_slice_scan --last k_spgemm > $OUT/shard8_gaps.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/step -- python $REPO/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-cli-wall --no-other-workloads --no-out-aln > $REPO/$OUT/step.log 2>&1)
python tools/micro/pass_gaps.py $OUT/step --first "k_part_count<" --last k_lz_parse --passes 2 > $OUT/step_gaps.txt 2>&1
find $OUT -name '*.csv' -size +4M -delete
timeout 600 python -m pytest tests -m gpu -x -q -k "accuracy_against or prepared or variants_write or config1" > $OUT/tests.log 2>&1
bash tools/micro/bench_scopes.sh phage-100k 5 > $OUT/bench_scopes.txt 2>&1
cat $OUT/shard8_gaps.txt $OUT/step_gaps.txt; tail -3 $OUT/tests.log; cat $OUT/bench_scopes.txt; tail -2 $OUT/step.log | cut -c1-600
