#!/bin/bash
# usage: sweep.sh <label>  (runs bench phage-100k + phage-1k parse timings)
python bench.py --steps 3 --warmup 1 --no-cli-wall --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1 100k', d['ms_per_step'], d['roofline']['ms_per_step_by_scope']['lz_parse'])"
python bench.py --workload phage-1k --steps 10 --warmup 3 --no-cli-wall --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1 1k', d['ms_per_step'], d['roofline']['ms_per_step_by_scope']['lz_parse'])"
