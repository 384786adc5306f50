#!/bin/bash
# the driver's bench command on the round's last build, summary on stdout, line under gpurun_out/r06j/
mkdir -p gpurun_out/r06j
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06j/bench.json 2> gpurun_out/r06j/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06j/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['host_ms_per_step'])
for k in d['roofline']['kernels']: print(k)
c = d['cpu_baseline']; print(c['value'], c['runs_pairs_per_s'], c['stage_seconds'], c['busy_threads_per_stage'])
print(d['out_aln']['ratio_parse_and_place'], {k: v['ms_per_step'] for k, v in d['other_workloads'].items()}, d['cli_wall']['runs_total_s'])
PY
