#!/bin/bash
# Developer helper: parse time against the probe widths (positions probed after an event / after a first miss).
for pw in ${PWS:-20 24 28 32 40 48 64}; do for pw2 in ${PW2S:-64}; do
echo -n "PW=$pw PW2=$pw2  "; VG_DEV_SWITCHES=1 VG_LZ_PW=$pw VG_LZ_PW2=$pw2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli-wall 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['ms_per_step_by_scope']['lz_parse'])"
done; done
