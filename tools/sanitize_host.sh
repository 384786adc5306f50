#!/bin/bash
# Developer tool: the host side of libvclust_gpu (FASTA / gzip ingest, filter reader, writers, task lists, the sharded entry
# points' host logic) under AddressSanitizer + UndefinedBehaviorSanitizer: a scratch copy of the repository is built with
# `-Xarch_host -fsanitize=address,undefined` (device code untouched) and the CPU test files that drive the library through
# ctypes run against it with the sanitizer runtime preloaded.  No GPU needed.  Last run (round 4): 46 tests, 0 reports.
set -eu
SRC=$(cd "$(dirname "$0")/.." && pwd); W=${1:-/tmp/vclust_asan}
rm -rf "$W"; mkdir -p "$W"; cp -r "$SRC"/include "$SRC"/vclust_amd "$SRC"/tests "$SRC"/oracle "$SRC"/vclust.py "$W"/
rm -rf "$W"/vclust_amd/libvclust_gpu.so "$W"/vclust_amd/_obj
RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
cd "$W"/vclust_amd/csrc
for f in vg_core.cpp vg_genomes.cpp vg_inflate.cpp vg_io.cpp vg_api.cpp vg_synth.cpp vg_prefilter.hip vg_align.hip vg_dist.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -I"$W"/include -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -x hip -c $f -o "$W"/$f.o &
done; wait
cd "$W"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o vclust_amd/libvclust_gpu.so *.o -lz -lpthread -ldl
make -C oracle > /dev/null
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_abi_host.py tests/test_cli.py tests/test_synth.py tests/test_distributed_cpu.py -q -m "not gpu" -s > "$W"/run.log 2>&1 || true
tail -2 "$W"/run.log
echo "sanitizer reports: $(grep -c 'AddressSanitizer\|runtime error' "$W"/run.log || true)"
