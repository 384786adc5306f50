#!/usr/bin/env python3
"""Developer timing tools (one script, sub-commands; none is part of the product or the tests):

  timing.py cli [families]      end-to-end wall of the drop-in CLI on a phage-100k slice: FASTA on disk -> ani.tsv on
                                disk, two cold processes, with the host-side phase marks (VG_HOST_TRACE) and the
                                allocator trace (VG_ALLOC_TRACE) of each process
  timing.py phases [families]   the same chain inside ONE process through the API, call by call
  timing.py step [families]     wall-clock split of one bench step (host + device)
  timing.py lz                  LZ parse kernel on synthetic pairs of graded divergence
  timing.py gz [families]       host ingest of the same FASTA as plain text, one-member gzip and bgzip (BGZF) blocks
"""
import os
import pathlib
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from vclust_amd import api, synth  # noqa: E402


def _fasta(td, nf):
    codes, offsets, names, _ = synth.make_workload('phage-100k', nf)
    fa = os.path.join(td, 'g.fna')
    synth.write_fasta(fa, codes, offsets, names)
    return fa


def _bare():
    t0 = time.perf_counter()
    subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); import vclust_amd.cli' % str(ROOT)], check=True)
    return time.perf_counter() - t0


def cli(nf):
    with tempfile.TemporaryDirectory(dir=os.environ.get('TMPDIR', '/tmp')) as td:
        fa = _fasta(td, nf)
        print(f'{os.path.getsize(fa) / 1e6:.0f} MB FASTA', flush=True)
        env = dict(os.environ, VG_HOST_TRACE='1', VG_ALLOC_TRACE='1')
        tot = 0.0
        gap = float(os.environ.get('CLI_GAP', '0'))
        thr = ['-t', os.environ['CLI_THREADS']] if os.environ.get('CLI_THREADS') else []      # (default: min(cores, 64))
        for cmd in (['prefilter', '-i', fa, '-o', os.path.join(td, 'fltr.txt'), '-v', '0', *thr],
                    ['align', '-i', fa, '-o', os.path.join(td, 'ani.tsv'), '--filter', os.path.join(td, 'fltr.txt'), '-v', '0', *thr]):
            if cmd[0] == 'align' and gap > 0:
                time.sleep(gap)
            t0 = time.perf_counter(); w0 = time.time()
            p = subprocess.run([sys.executable, str(ROOT / 'vclust.py'), *cmd], env=env, stderr=subprocess.PIPE, text=True)
            dt = time.perf_counter() - t0; tot += dt; w1 = time.time()
            stamps = [float(l.rsplit('@', 1)[1]) for l in p.stderr.splitlines() if l.startswith('[vg host]') and '@' in l]
            if stamps:
                print(f'   process start -> first mark {stamps[0] - w0:.3f} s; last mark -> process gone {w1 - stamps[-1]:.3f} s')
            print(f'== {cmd[0]}: {dt:.3f} s (rc {p.returncode}); bare interpreter + imports: '
                  f'{_bare():.3f} s')
            print(p.stderr, flush=True)
        print(f'== total {tot:.3f} s, rows', sum(1 for _ in open(os.path.join(td, 'ani.tsv'))) - 1)


def phases(nf):
    with tempfile.TemporaryDirectory(dir=os.environ.get('TMPDIR', '/tmp')) as td:
        fa = _fasta(td, nf)
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
        names_ = ['set_device', 'load', 'to_device', 'kmer_shared#1', 'kmer_shared#2', 'write_fltr', 'read_filter', 'align_tasks',
                  'lz_align#1', 'lz_align#2', 'write_ani']
        for n_, a, b in zip(names_, t, t[1:]):
            print(f'{n_:14s} {b - a:7.2f} s')


def step(nf):
    api.set_device(0)
    codes, offsets, names, _ = synth.make_workload('phage-100k', nf)
    gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
    keep = []                 # STEP_KEEP=1: results of earlier iterations stay alive (nothing is unmapped between calls)
    for it in range(int(os.environ.get('STEP_ITERS', '4'))):
        api.profile_enable(True); api.profile_reset()
        t = [time.perf_counter()]
        sizes, pairs = gs.kmer_shared(k=25, min_shared=20); t.append(time.perf_counter())
        k1 = sum(e['total_ms'] for e in api.profile_get()); api.profile_reset()
        cand = gs.filter_pairs(sizes, pairs, k=25, min_kmers=20, min_ident=0.7); t.append(time.perf_counter())
        if os.environ.get('STEP_PREPARE', '1') == '1':
            gs.lz_prepare(cand)            # (as the bench step does: the index build runs beside the task list)
        tasks = gs.align_tasks(cand); t.append(time.perf_counter())
        stats = gs.lz_align(tasks); t.append(time.perf_counter())
        k2 = sum(e['total_ms'] for e in api.profile_get())
        if os.environ.get('STEP_KEEP') == '1':
            keep.append((sizes, pairs, cand, tasks, stats))
        d = np.diff(t) * 1e3
        print('kmer_shared %.2f (kernels %.2f)  candidate %.2f  tasks %.2f  lz_align %.2f (kernels %.2f)  total %.2f ms'
              % (d[0], k1, d[1], d[2], d[3], k2, d.sum()))


def lz():
    def run(label, codes, offsets, pairs, reps=3):
        gs = api.GenomeSet.from_codes(codes, offsets); gs.to_device()
        tasks = gs.align_tasks(pairs)
        gs.lz_align(tasks)
        api.profile_enable(True); api.profile_reset()
        for _ in range(reps):
            st = gs.lz_align(tasks)
        prof = {e['name']: e['total_ms'] / e['launches'] for e in api.profile_get()}
        api.profile_enable(False)
        print(f'{label:28s} tasks {len(tasks):6d}  parse {prof.get("lz_parse", 0):8.3f} ms  build {prof.get("lz_build_index", 0):7.3f} ms  '
              f'per-task {prof.get("lz_parse", 0) * 1e3 / len(tasks):7.3f} us  sumM {int(st["n_match"].sum())} regs {int(st["n_regions"].sum())}')

    def ident(n, L, rate, seed=5):
        rng = np.random.default_rng(seed)
        seqs = []
        for i in range(n):
            a = rng.integers(0, 4, size=L, dtype=np.uint8)
            b = a.copy()
            m = rng.random(L) < rate
            b[m] = (b[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            seqs += [a, b]
        off = np.zeros(len(seqs) + 1, dtype=np.int64); off[1:] = np.cumsum([len(s) for s in seqs])
        pairs = np.array([(2 * i + 1, 2 * i, 0) for i in range(n)], dtype=api.PAIR_DTYPE)
        return np.concatenate(seqs), off, pairs
    api.set_device(0)
    for rate in (0.0, 0.01, 0.05, 0.10, 0.20, 0.30, 0.75):
        c, o, p = ident(4500, 40000, rate)
        run(f'subst rate {rate}', c, o, p)
    c, o, n = synth.make_families(100, 10, 40000, seed=1)
    run('phage-1k families', c, o, synth.family_pairs(100, 10))


def _bgzf(data, block=65280):
    """bgzip's container (no bgzip binary in the image): gzip members of <= 64 KiB, compressed size in a 'BC' subfield"""
    import struct
    import zlib
    out = bytearray()
    for o in list(range(0, len(data), block)) + [len(data)]:
        chunk = data[o:o + block] if o < len(data) else b''
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        out += struct.pack('<BBBBIBBH', 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b'BC' + struct.pack('<HH', 2, 12 + 6 + len(cdata) + 8 - 1)
        out += cdata + struct.pack('<II', zlib.crc32(chunk) & 0xffffffff, len(chunk))
    return bytes(out)


def gz(nf):
    import gzip
    with tempfile.TemporaryDirectory(dir=os.environ.get('TMPDIR', '/tmp')) as td:
        fa = _fasta(td, nf)
        text = open(fa, 'rb').read()
        open(fa + '.gz', 'wb').write(gzip.compress(text, 6))
        open(fa + '.bgz.gz', 'wb').write(_bgzf(text))
        for name in (fa, fa + '.gz', fa + '.bgz.gz'):
            for threads in (1, 16, os.cpu_count() or 16):
                t0 = time.perf_counter()
                gs = api.GenomeSet.load([name], multisample=True, n_threads=threads)
                dt = time.perf_counter() - t0
                print(f'{os.path.basename(name):14s} {os.path.getsize(name) / 1e6:7.0f} MB on disk, {len(text) / 1e6:.0f} MB of text, '
                      f'{threads:3d} threads: {dt:.3f} s ({len(text) / 1e6 / dt:.0f} MB/s of text), {len(gs.names())} genomes', flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'cli'
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    {'cli': lambda: cli(nf), 'phases': lambda: phases(nf), 'step': lambda: step(nf), 'lz': lz, 'gz': lambda: gz(nf)}[what]()
