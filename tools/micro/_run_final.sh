# final collection of round 5 (GPU box): gpu suite, profiles of three workloads, the bench lines, the scaling emulation
mkdir -p gpurun_out/r5z
(timeout 2300 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/r5z/gpu_suite.log
tools/collect_profiles.sh r05 phage-100k > gpurun_out/r5z/prof_100k.log 2>&1
tools/collect_profiles.sh r05 phage-1k > gpurun_out/r5z/prof_1k.log 2>&1
tools/collect_profiles.sh r05 imgvr-10k > gpurun_out/r5z/prof_imgvr.log 2>&1
timeout 1200 python bench.py > gpurun_out/r5z/bench_default.json 2> gpurun_out/r5z/bench_default.err
timeout 600 python bench.py --workload phage-1k --no-cli-wall --steps 20 --warmup 3 > gpurun_out/r5z/bench_1k.json 2>/dev/null
timeout 600 python bench.py --workload imgvr-10k --no-cli-wall --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r5z/bench_imgvr.json 2>/dev/null
(SCAN=sliced timeout 600 python tools/strong_scaling_sim.py 2>&1 | tail -6) > gpurun_out/r5z/sim_sliced.log
(SCAN=sliced RANK_SIM=5 timeout 600 python tools/strong_scaling_sim.py 2>&1 | tail -6) > gpurun_out/r5z/sim_sliced_rank5.log
cat gpurun_out/r5z/gpu_suite.log; tail -2 gpurun_out/r5z/bench_default.err
