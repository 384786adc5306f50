#!/bin/bash
# Developer helper: one bench run, step time and per-scope milliseconds only.  usage: bench_scopes.sh [workload] [steps]
W=${1:-phage-100k}; S=${2:-5}
timeout 600 python bench.py --workload $W --steps $S --warmup 2 --no-cpu-baseline --no-cli-wall --no-other-workloads --no-out-aln 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:12], d['ms_per_step'], d['config']['pairs_per_step'], d['roofline']['ms_per_step_by_scope'])"
