g++ -O2 -pthread -o /tmp/read_cost tools/micro/read_cost.cpp || exit 1
python - <<'PY'
import sys; sys.path.insert(0,'.')
from vclust_amd import synth
codes, offsets, names, _ = synth.make_workload('phage-100k', 10000)
synth.write_fasta('/tmp/g.fna', codes, offsets, names)
PY
ls -la /tmp/g.fna; nproc
for T in 16 64 128 256; do echo "T=$T"; /tmp/read_cost /tmp/g.fna $T; done
