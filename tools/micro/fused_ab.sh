set -x
python -m pytest "tests/test_gpu_cli.py::test_kernel_variants_write_the_same_files" tests/test_gpu_parity.py -m gpu -x -q -k "not allocator" 2>&1 | tail -3
echo "--- separate (default)"; bash tools/micro/bench_scopes.sh phage-100k 5
echo "--- fused"; VG_DEV_SWITCHES=1 VG_LZ_FUSED=1 bash tools/micro/bench_scopes.sh phage-100k 5
