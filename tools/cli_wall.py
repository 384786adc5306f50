"""Developer tool: end-to-end CLI wall time (FASTA on disk -> ani.tsv on disk) on a synthetic set."""
import os, sys, pathlib, subprocess, tempfile, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import synth
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
root = pathlib.Path(__file__).resolve().parent.parent
with tempfile.TemporaryDirectory() as td:
    t0 = time.perf_counter()
    codes, offsets, names = synth.make_families(nf, 10, 40000, seed=1)
    fa = os.path.join(td, 'g.fna'); synth.write_fasta(fa, codes, offsets, names)
    print(f'generated {len(names)} genomes, {os.path.getsize(fa)/1e6:.0f} MB FASTA in {time.perf_counter()-t0:.1f} s', flush=True)
    for cmd in (['prefilter', '-i', fa, '-o', os.path.join(td, 'fltr.txt'), '-v', '0'],
                ['align', '-i', fa, '-o', os.path.join(td, 'ani.tsv'), '--filter', os.path.join(td, 'fltr.txt'), '-v', '0']):
        t0 = time.perf_counter()
        subprocess.run([sys.executable, str(root / 'vclust.py'), *cmd], check=True)
        print(f'{cmd[0]:10s} {time.perf_counter()-t0:7.2f} s', flush=True)
    print('rows', sum(1 for _ in open(os.path.join(td, 'ani.tsv'))) - 1)
