mkdir -p gpurun_out/r5i
(timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "hash_subshards or subshard or fraction or config2" 2>&1 | tail -5) > gpurun_out/r5i/t.log
timeout 1200 python bench.py --workload contigs-1M --steps 2 --warmup 1 --no-cpu-baseline --no-cli-wall > gpurun_out/r5i/contigs.json 2> gpurun_out/r5i/contigs.err
cat gpurun_out/r5i/t.log; python -c "import json; d=json.loads(open('gpurun_out/r5i/contigs.json').read().strip().splitlines()[-1]); print('contigs-1M', d['ms_per_step'], d['config']['pairs_per_step'], d['roofline']['ms_per_step_by_scope'], d['roofline']['host_ms_per_step'])"
