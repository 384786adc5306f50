for pw in 32 16 8 4; do for pw2 in 64 32; do
echo "PW=$pw PW2=$pw2"; VG_LZ_PW=$pw VG_LZ_PW2=$pw2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli-wall | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['ms_per_step_by_scope'])"
done; done
