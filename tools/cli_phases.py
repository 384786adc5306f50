"""Developer tool: where the CLI's wall time goes at 100k genomes (ingest / upload / first pass / write)."""
import os, sys, pathlib, tempfile, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
with tempfile.TemporaryDirectory() as td:
    codes, offsets, names = synth.make_families(nf, 10, 40000, seed=1)
    fa = os.path.join(td, 'g.fna'); synth.write_fasta(fa, codes, offsets, names)
    del codes
    t = [time.perf_counter()]
    api.set_device(0); t.append(time.perf_counter())
    gs = api.GenomeSet.load([fa], True, n_threads=64); t.append(time.perf_counter())
    gs.to_device(); t.append(time.perf_counter())
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20); t.append(time.perf_counter())
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20); t.append(time.perf_counter())
    gs.write_fltr(os.path.join(td, 'f.txt'), sizes, pairs); t.append(time.perf_counter())
    flt = gs.read_filter(os.path.join(td, 'f.txt'), 0.0); t.append(time.perf_counter())
    tasks = gs.align_tasks(flt); t.append(time.perf_counter())
    st = gs.lz_align(tasks); t.append(time.perf_counter())
    st = gs.lz_align(tasks); t.append(time.perf_counter())
    gs.write_ani(os.path.join(td, 'a.tsv'), tasks, st, columns=None); t.append(time.perf_counter())
    names_ = ['set_device', 'load', 'to_device', 'kmer_shared#1', 'kmer_shared#2', 'write_fltr', 'read_filter', 'align_tasks', 'lz_align#1', 'lz_align#2', 'write_ani']
    for n_, a, b in zip(names_, t, t[1:]):
        print(f'{n_:14s} {b - a:7.2f} s')
