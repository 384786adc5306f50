"""Developer tool: CLI handler time vs process wall (interpreter + HIP teardown) at a given size."""
import os, sys, pathlib, subprocess, tempfile, time
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
from vclust_amd import synth
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
with tempfile.TemporaryDirectory() as td:
    codes, offsets, names = synth.make_families(nf, 10, 40000, seed=1)
    fa = os.path.join(td, 'g.fna'); synth.write_fasta(fa, codes, offsets, names)
    del codes
    code = ("import sys, time; sys.path.insert(0, %r); t0 = time.perf_counter(); from vclust_amd import cli; "
            "sys.argv = ['vclust.py', 'prefilter', '-i', %r, '-o', %r, '-v', '0']; cli.main(); print('handler %%.2f s' %% (time.perf_counter() - t0), flush=True)"
            % (str(root), fa, os.path.join(td, 'f.txt')))
    t0 = time.perf_counter()
    subprocess.run([sys.executable, '-c', code], check=False)
    print('process %.2f s' % (time.perf_counter() - t0))
