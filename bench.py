#!/usr/bin/env python3
"""bench.py — genome pairs/sec through prefilter+align on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path over the synthetic genome set, inputs already
resident in HBM: Kmer-db prefilter (k-mer extraction, inverted index, shared-k-mer SpGEMM)
-> host threshold (min-kmers / ani-shorter) -> LZ-ANI parse of every surviving pair in both
directions.  value = unordered pairs aligned per second (whole job, all ranks).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 workload: configs[1] of BASELINE.json, "phage-1k" = 100 families x 10 members x 40 kb.
N > 1: weak scaling, N x 100 families; the prefilter is sharded by k-mer hash range (partial
counts all-gathered over RCCL and summed on the device), the align tasks are dealt by reference
range (each rank indexes 1/N of the genomes), and the per-pair integer rows are all-gathered.
"""
import argparse
import json
import os
import pathlib
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from vclust_amd import api, synth  # noqa: E402
from vclust_amd import distributed as D  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# profile-scope name -> kernel name in the rocprofv3 summaries under profiles/
KERNEL_OF_SCOPE = {'lz_parse': ('k_lz_parse_seg', 'k_lz_parse'), 'lz_build_index': ('k_build_index_lds',),
                   'radix_sort_pairs': ('rocprim::onesweep_iteration',), 'index_runs': ('k_runs',),
                   'spgemm_rows': ('k_spgemm',), 'kmer_extract': ('k_kmer_extract',), 'group_sort': ('k_group_sort',)}


def pmc_traffic(scope):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, tools/collect_profiles.sh): counters cannot
    be read inside the timed run.  Raw FETCH_SIZE + WRITE_SIZE; see the file for the gfx950 x2 bound."""
    files = sorted((ROOT / 'profiles').glob('*_pmc_hbm_traffic.json'))
    if not files:
        return None, None
    doc = json.loads(files[-1].read_text())
    for k in KERNEL_OF_SCOPE.get(scope, ()):
        if k in doc.get('kernels', {}):
            e = doc['kernels'][k]
            per_step = e['launches'] / max(doc.get('steps_in_run', e['launches']), 1)      # e.g. 3 sort iterations per step
            return round(e['hbm_bytes_per_launch_raw'] * per_step), f'profiles/{files[-1].name}:{k}'
    return None, None


def cpu_baseline(sample_families, members, length, seed, threads):
    """Time the CPU oracle (own restatement, not upstream) on a bounded sample of the workload."""
    cli = ROOT / 'oracle' / '_build' / 'oracle_cli'
    if not cli.exists():
        subprocess.run(['make', '-C', str(ROOT / 'oracle')], check=True, stdout=subprocess.DEVNULL)
    codes, offsets, names = synth.make_families(sample_families, members, length=length, seed=seed)
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, 's.fna')
        synth.write_fasta(fa, codes, offsets, names)
        fl, ani = os.path.join(td, 'fltr.txt'), os.path.join(td, 'ani.tsv')
        t0 = time.perf_counter()
        subprocess.run([str(cli), 'prefilter', '-t', str(threads), '-o', fl, fa], check=True)
        subprocess.run([str(cli), 'align', '-t', str(threads), '--filter', fl, '0', '-o', ani, fa], check=True)
        dt = time.perf_counter() - t0
        rows = sum(1 for _ in open(ani)) - 1
    pairs = rows // 2
    return dict(value=pairs / dt, unit='pairs/s', cores=threads, kind='port',
                sample=f'{sample_families} families x {members} x {length} bp = {pairs} pairs, '
                       f'oracle_cli prefilter+align incl. FASTA parse, {dt:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--families', type=int, default=100, help='families per GPU')
    ap.add_argument('--members', type=int, default=10)
    ap.add_argument('--length', type=int, default=40000)
    ap.add_argument('--k', type=int, default=25)
    ap.add_argument('--min-kmers', type=int, default=20)
    ap.add_argument('--min-ident', type=float, default=0.7)
    ap.add_argument('--workload', choices=['phage', 'imgvr'], default='phage',
                    help='phage: families x members x length (configs[1]/[3]); imgvr: --contigs mixed 5-200 kb contigs (configs[2])')
    ap.add_argument('--contigs', type=int, default=10000, help='contigs per GPU for --workload imgvr')
    ap.add_argument('--cpu-sample-families', type=int, default=100)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    import torch
    if api.device_count() < 1:
        raise SystemExit('bench.py needs a HIP device: libvclust_gpu has no CPU fallback')
    if world > 1:
        dist, dev = D.init_process_group()          # nccl (= RCCL) unless VCLUST_DIST_BACKEND says otherwise
    else:
        dev = torch.device('cuda', local_rank)
    api.set_device(local_rank % api.device_count())

    n_fam = args.families * world
    if args.workload == 'imgvr':
        codes, offsets, names, _ = synth.make_contigs(args.contigs * world, seed=2)
        wl = f'imgvr-like x{world}: {len(names)} contigs log-uniform 5-200 kb, families geometric(0.2) <= 20'
    else:
        codes, offsets, names = synth.make_families(n_fam, args.members, length=args.length, seed=1)
        wl = f'phage-1k x{world}: {n_fam} families x {args.members} members x {args.length} bp'
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    gs.to_device()
    lens = gs.lengths()

    def barrier():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    state = {}

    def step():
        # -- prefilter: this rank's k-mer hash range; partial counts add up across ranks (RCCL all-gather)
        if world > 1:
            sizes, pairs = D.prefilter_counts(gs, dist, dev, rank, world, args.k, 1.0)
        else:
            sizes, pairs = gs.kmer_shared(k=args.k, min_shared=args.min_kmers)
        cand = gs.filter_pairs(sizes, pairs, k=args.k, min_kmers=args.min_kmers, min_ident=args.min_ident)
        # -- align: canonical task list, contiguous share per rank, rows gathered over RCCL
        tasks = gs.align_tasks(cand)
        stats, _ = D.align_rows(gs, tasks, dist, dev, rank, world, None, False)
        state.update(n_pairs=len(tasks) // 2, stats=stats, tasks=tasks)

    api.profile_enable(False)
    for _ in range(args.warmup):
        step()
    api.profile_enable(True)
    api.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    prof = api.profile_get()
    api.profile_enable(False)

    if rank == 0:
        n_pairs = state['n_pairs']
        # roofline of the dominant kernel (HIP events on the library stream, vg_profile_*)
        dom = max(prof, key=lambda e: e['total_ms']) if prof else None
        roofline = None
        if dom and dom['launches']:
            avg_ms = dom['total_ms'] / dom['launches']
            alg_bytes = dom['bytes'] / dom['launches']
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(dom['name']) if world == 1 and args.families == 100 and args.workload == 'phage' else (None, None)
            roofline = dict(bound='hbm', kernel=dom['name'], achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit='GB/s',
                            frac=round(achieved / HBM_PEAK_GBS, 6), traffic=traffic, traffic_source=traffic_src,
                            avg_launch_ms=round(avg_ms, 4), algorithmic_bytes_per_launch=round(alg_bytes),
                            kernels={e['name']: round(e['total_ms'] / max(e['launches'], 1), 4) for e in prof})
            if dom['name'] == 'radix_sort_pairs':
                roofline['note'] = ('one "launch" = one rocprim::radix_sort_pairs call = 1 histogram kernel + one onesweep kernel per '
                                    '8 sorted key bits (3 at this size); traffic sums the onesweep kernels')
            # whole path: B_pre = sum(L/4 + 16 (L-k+1)) + 16 P, B_aln = sum over ordered pairs ((Lq+Lr)/4 + 20)
            b_pre = float(np.sum(lens / 4.0 + 16.0 * np.maximum(lens - args.k + 1, 0))) + 16.0 * n_pairs
            tk = state['tasks']
            b_aln = float(np.sum((lens[tk['q']] + lens[tk['r']]) / 4.0 + 20.0))
            step_s = dt / args.steps
            roofline['path'] = dict(algorithmic_bytes_per_step=round(b_pre + b_aln), achieved=round((b_pre + b_aln) / step_s / 1e9, 3),
                                    frac=round((b_pre + b_aln) / step_s / 1e9 / HBM_PEAK_GBS, 6), note='whole step incl. host time, all ranks' if world > 1 else 'whole step incl. host time')
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.workload == 'phage':
            cpu = cpu_baseline(min(args.cpu_sample_families, args.families), args.members, args.length, 1,
                               os.cpu_count() or 1)
        out = {
            'metric': 'genome pairs/sec through prefilter+align (ani.tsv)',
            'value': round(n_pairs * args.steps / dt, 3),
            'unit': 'pairs/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'u64',
            'data': 'synthetic',
            'config': {
                'workload': f'{wl}, k={args.k}, min-kmers={args.min_kmers}, min-ident={args.min_ident}, lz defaults',
                'genomes': int(len(gs)), 'pairs_per_step': int(n_pairs), 'total_bases': int(lens.sum()),
                'parallelism': f'kmer-range x{world} prefilter, reference-range x{world} align',
            },
            'roofline': roofline,
            'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
