set -x
mkdir -p gpurun_out
cd tools/micro && hipcc --offload-arch=gfx950 -O2 -o /tmp/ft first_touch.hip && cd ../..
# what the driver's bench does first on a fresh box: the two cold CLI processes
python tools/timing.py cli > gpurun_out/r4_cli_fresh.txt 2>&1
/tmp/ft malloc 8 30 1 > gpurun_out/r4_first_touch.txt 2>&1
/tmp/ft vmm 8 30 1 >> gpurun_out/r4_first_touch.txt 2>&1
/tmp/ft malloc 32 7 0 >> gpurun_out/r4_first_touch.txt 2>&1
python tools/timing.py cli > gpurun_out/r4_cli_warm.txt 2>&1
tail -5 gpurun_out/r4_cli_fresh.txt; cat gpurun_out/r4_first_touch.txt; tail -3 gpurun_out/r4_cli_warm.txt
