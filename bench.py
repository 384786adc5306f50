#!/usr/bin/env python3
"""bench.py — genome pairs/sec through prefilter+align on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path over the synthetic genome set, inputs already resident in HBM:
Kmer-db prefilter (k-mer extraction, inverted index, shared-k-mer SpGEMM) -> thresholds (min-kmers /
ani-shorter) -> LZ-ANI parse of every surviving pair in both directions -> integer rows on the host.
value = unordered pairs aligned per second (whole job, all ranks).

  python bench.py                         # N = 1: phage-100k = 100 000 x 40 kb (the configuration the metric is quoted on)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload: `phage-100k` of SURVEY.md 8(d) (BASELINE configs[3]: 10 000 families x 10 members x 40 kb,
seed 3), which fits one MI355X.  N > 1: the SAME set (strong scaling; `--scaling weak` multiplies the families
by N instead) through the sharded C-ABI entry points (vg_kmer_shared_sharded / vg_lz_align_pairs_sharded): the prefilter
is sharded by k-mer range -- every rank scans 1/N of the bases and an RCCL all-to-all hands each rank the kept masks of its
range; the pairs' keys and counts are all-gathered and summed on the device --, the align tasks are dealt by reference
range (each rank indexes 1/N of the genomes), and the per-pair integer rows are all-gathered.  N > 1 creates the library's
RCCL communicator strictly (no fall-back to host all-gathers); the line says which communicator ran (`comm`).  Other workloads: --workload phage-1k | imgvr-10k | contigs-1M (+ --count).
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
CLI_RUNS = 3              # end-to-end CLI leg: runs back to back (min / median / max in the line)

# profile scope (vg_profile_*) -> (kernel as rocprofv3 names it, stage of SURVEY 8(d) whose algorithmic bytes it is priced on)
SCOPES = {
    'kmer_extract': ('k_kmer_extract', 'extract'),
    'kmer_partition': ('k_part_scatter_dense', 'index'),
    'kmer_partition2': ('k_part_scatter2_narrow', 'index'),
    'radix_sort_pairs': ('rocprim::onesweep_iteration', 'index'),
    'index_runs': ('k_group_runs', 'index'),
    'bucket_sort_runs': ('k_bucket_runs', 'index'),
    'kmer_count': ('k_kmer_count', 'extract'),
    'kmer_emit': ('k_kmer_gather', 'extract'),
    'kmer_emit_recompute': ('k_kmer_emit_sparse', 'extract'),
    'kmer_multi_mask': ('k_multi_mask', 'extract'),
    'kmer_slice_scan': ('k_slice_scan', 'index'),
    'spgemm_rows': ('k_spgemm', 'join'),
    'lz_build_index': ('k_build_index_reg', 'align'),
    'lz_parse': ('k_lz_parse', 'align'),
}


def pmc_traffic(workload, kernel, launches_per_step):
    """HBM bytes per launch of a kernel from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command at the same workload, tools/collect_profiles.sh): counters cannot be
    read inside the timed run.  Corrected as MI355X_MICROARCH.md prescribes (see the file's `correction`).
    A profile whose launches per step differ from this run's was taken on another build (a launch of it is not a
    launch of this one): it is refused, and the line says so instead of printing its figure."""
    for f in sorted((ROOT / 'profiles').glob('r*_pmc_hbm_traffic*.json'), reverse=True):
        doc = json.loads(f.read_text())
        if doc.get('workload') != workload:
            continue
        for k, e in doc.get('kernels', {}).items():
            if kernel.startswith(k) or k.startswith(kernel):
                lps = e['launches'] / max(1, doc.get('steps_in_run', 1))
                if abs(lps - launches_per_step) > 1e-6:
                    return None, f'REFUSED profiles/{f.name}:{k}: {lps:g} launches per step there, {launches_per_step:g} in this run (stale profile)'
                return round(e['hbm_bytes_per_launch']), f'profiles/{f.name}:{k}'
    return None, None


def cpu_baseline(sample_families, members, length, seed, threads, k, min_kmers, min_ident, total_families, reps=3):
    """The CPU oracle (own restatement of the reference path: the reference's native binaries are absent from
    its checkout) on a bounded sample of the same workload, SAME SCOPE as `value`: genomes in memory ->
    integer rows in memory, EVERY stage on all host threads (oracle/align_oracle.c: vo_path_rows_mt -- k-mer sets per
    genome, hash-partitioned index sorted per thread, per-thread pair tables merged by pair hash, LZ parse over
    references; tests/test_oracle_golden.py holds it equal to the serial checker).  `reps` repetitions: value = the MEDIAN
    run, with the spread and every run's stage seconds beside it; busy_threads = CPU seconds / wall seconds of a stage
    (clock of the whole process): the threads that actually worked in it."""
    sys.path.insert(0, str(ROOT / 'tests'))
    import oracle_lib as orc
    codes, offsets, names = synth.make_families(sample_families, members, length=length, seed=seed)
    runs = []
    pairs = 0; ran = threads
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        rows, stage_s, ran = orc.path_rows_mt(codes, offsets, k=k, min_kmers=min_kmers, min_ident=min_ident, threads=threads)
        dt = time.perf_counter() - t0
        cpu_s = orc.last_stage_cpu()
        pairs = len(rows) // 2
        runs.append(dict(seconds=round(dt, 3), pairs_per_s=round(pairs / dt, 1), stage_seconds=stage_s,
                         busy_threads={k_: round(cpu_s[k_] / stage_s[k_], 1) if stage_s[k_] > 0 else None for k_ in stage_s}))
    order = sorted(range(len(runs)), key=lambda i: runs[i]['seconds'])
    med = runs[order[len(order) // 2]]
    rates = [r['pairs_per_s'] for r in runs]
    total_s = sum(r['seconds'] for r in runs)
    return dict(value=med['pairs_per_s'], unit='pairs/s', cores=ran, kind='port',
                seconds=med['seconds'], stage_seconds=med['stage_seconds'], busy_threads_per_stage=med['busy_threads'],
                threads_per_stage={k_: ran for k_ in med['stage_seconds']},
                repetitions=len(runs), runs_pairs_per_s=rates, spread=round((max(rates) - min(rates)) / med['pairs_per_s'], 3),
                runs=runs, total_seconds=round(total_s, 1),
                sample_genomes=sample_families * members, extrapolation=round(total_families / sample_families, 2),
                sample=f'first {sample_families} families ({sample_families * members} genomes x {length} bp) of the same set = '
                       f'{pairs} pairs; genomes in memory -> integer rows in memory (same scope as value), {len(runs)} repetitions of {med["seconds"]:.1f} s (median) on {ran} OpenMP threads '
                       'in every stage (value = the median run; the first run also pays the first touch of its GBs of records); '
                       'pairs/s of the sample stands for the whole set (the work per family is the same: families do not share k-mers); '
                       'own CPU restatement (oracle/), not upstream: kmer-db / lz-ani sources are absent from the reference checkout')


def host_cpus():
    """CPUs this process may actually use: the smallest of the visible CPUs, the affinity mask and the cgroup CPU quota (a
    container on a 256-thread host may be allowed 16: a team of 256 OpenMP threads then time-slices 16 cores)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0]); per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    quota = q / per
            break
        except (OSError, ValueError, IndexError):
            continue
    return dict(visible=os.cpu_count() or 1, usable=n, cgroup_quota=round(quota, 2) if quota else None,
                threads=max(1, min(n, 256, int(quota + 0.5) if quota else n)))


def _run_traced(cmd, env):
    """One CLI process with the library's host-side phase marks (VG_HOST_TRACE): wall seconds, marks {name: ms since
    the previous mark}, and the two stretches the library cannot see (process start -> first mark, last mark -> gone)."""
    w0 = time.time(); t0 = time.perf_counter()
    p = subprocess.run(cmd, check=True, env=dict(env, VG_HOST_TRACE='1', VG_ALLOC_TRACE='1'), stderr=subprocess.PIPE, text=True)
    dt = time.perf_counter() - t0; w1 = time.time()
    import re
    marks, stamps, alloc_ms, blocks = {}, [], 0.0, []
    for line in p.stderr.splitlines():
        m = re.match(r'\[vg host\] (.*?)\s+\+([0-9.]+) ms\s+alloc ([0-9.]+) ms\s+@([0-9.]+)\s*$', line)
        if m:
            name = m.group(1).strip()
            marks[name] = round(marks.get(name, 0.0) + float(m.group(2)), 1)
            alloc_ms += float(m.group(3))
            stamps.append(float(m.group(4)))
            continue
        m = re.match(r'\[vg alloc\] (\S+) of ([0-9.]+) GiB took ([0-9.]+) ms', line)
        if m and float(m.group(2)) >= 1.0:
            blocks.append(dict(path=m.group(1), gib=float(m.group(2)), ms=float(m.group(3))))
    out = dict(wall_s=round(dt, 3), marks_ms=marks, alloc_wait_ms=round(alloc_ms, 1), blocks=blocks)
    if stamps:
        out['start_to_first_mark_s'] = round(stamps[0] - w0, 3)      # interpreter, imports, dlopen of the library + HIP runtime
        out['last_mark_to_exit_s'] = round(w1 - stamps[-1], 3)       # process tear-down (the driver releases the device context)
    return out


def _phase_sums(tr):
    """Condensed view of one process's marks: seconds per phase."""
    m = tr['marks_ms']
    def tot(*prefixes):
        return round(sum(v for k, v in m.items() if k.startswith(prefixes)) / 1e3, 3)
    dev = tot('buckets:', 'index built', 'spgemm', 'pass done', 'sub-shards done', 'vg_kmer_shared', 'vg_lz_align', 'lz:', 'extract:')
    alloc = round(tr.get('alloc_wait_ms', 0.0) / 1e3, 3)
    # alloc_wait = time inside the driver's allocation calls (it clears memory as it hands it out; booked where it was
    # waited for, almost all of it inside the device phases, so it is taken out of that line); device_work = kernels,
    # copies and the host code between them
    return dict(process_start=tr.get('start_to_first_mark_s'), ingest_and_upload=tot('ingest:', 'genomes uploaded', 'device ready'),
                alloc_wait=alloc, device_work=round(max(0.0, dev - alloc), 3),
                filter_and_tasks=tot('filter read', 'align_tasks'), writer_and_release=tot('fltr.txt written', 'ani.tsv written'),
                process_exit=tr.get('last_mark_to_exit_s'),
                blocks_ge_1GiB=[f"{b['path']} {b['gib']:.1f} GiB {b['ms']:.0f} ms" for b in tr.get('blocks', [])])


def cli_wall(codes, offsets, names, n_pairs):
    """End-to-end wall of the drop-in CLI: FASTA on disk -> fltr.txt -> ani.tsv on disk,
    two processes (`vclust.py prefilter`, `vclust.py align --filter`), including process start, ingest and writers,
    with the per-process split of where the wall time goes."""
    tdo = tempfile.TemporaryDirectory(dir=os.environ.get('TMPDIR', '/tmp'))      # (kept by the caller until the timed step's rows have been compared with ani.tsv)
    if True:
        td = tdo.name
        fa = os.path.join(td, 's.fna')
        synth.write_fasta(fa, codes, offsets, names)
        fl, ani = os.path.join(td, 'fltr.txt'), os.path.join(td, 'ani.tsv')
        env = dict(os.environ)
        runs = []
        for _ in range(CLI_RUNS):
            t0 = time.perf_counter()
            pre = _run_traced([sys.executable, str(ROOT / 'vclust.py'), 'prefilter', '-i', fa, '-o', fl, '-v', '0'], env)
            t1 = time.perf_counter()
            aln = _run_traced([sys.executable, str(ROOT / 'vclust.py'), 'align', '-i', fa, '-o', ani, '--filter', fl, '-v', '0'], env)
            t2 = time.perf_counter()
            runs.append(dict(prefilter_s=round(t1 - t0, 3), align_s=round(t2 - t1, 3), total_s=round(t2 - t0, 3),
                             breakdown_s=dict(prefilter=_phase_sums(pre), align=_phase_sums(aln))))
        rows = sum(1 for _ in open(ani)) - 1
        size = os.path.getsize(fa)
        os.unlink(fa)
    order = sorted(range(len(runs)), key=lambda i: runs[i]['total_s'])
    med = runs[order[len(order) // 2]]
    out = dict(prefilter_s=med['prefilter_s'], align_s=med['align_s'], total_s=med['total_s'], rows=rows,
               fasta_bytes=size, pairs_per_s=round(rows / 2 / med['total_s'], 1),
               runs_total_s=[r['total_s'] for r in runs], min_s=runs[order[0]]['total_s'], median_s=med['total_s'], max_s=runs[order[-1]]['total_s'],
               alloc_wait_s_per_run=[round(r['breakdown_s']['prefilter']['alloc_wait'] + r['breakdown_s']['align']['alloc_wait'], 3) for r in runs],
               breakdown_s=med['breakdown_s'],
               note=f'python vclust.py prefilter + align --filter, FASTA on disk -> ani.tsv on disk, two cold processes, {len(runs)} runs back to back '
                    '(a command returns when its files are complete and closed: the tear-down of its device context runs in a detached child '
                    'beside the next command; total_s / breakdown_s = the median run; runs_total_s in run order: the first is the one that meets the device as the '
                    'previous tenant left it); breakdown from the library\'s host-side phase marks (VG_HOST_TRACE); alloc_wait = time inside '
                    'the driver\'s allocation calls (device memory it still has to wipe costs 25-32 ms per GiB, profiles/r04_first_touch.txt), '
                    'taken out of device_work')
    if runs[order[-1]] is not med:
        out['slowest_run_breakdown_s'] = runs[order[-1]]['breakdown_s']
    return out, tdo, ani


def stage_bytes_of(lens, tasks, n_pairs, k):
    """SURVEY 8(d) algorithmic bytes by stage, B_pre and B_aln."""
    n_pos = float(np.sum(np.maximum(lens - k + 1, 0)))
    sb = {
        'extract': float(np.sum(lens)) / 4.0 + 8.0 * n_pos,        # read the packed bases, write each position's u64 k-mer
        'index': 16.0 * n_pos,                                    # the join's key stream: written once, read once
        'join': 8.0 * n_pos + 16.0 * n_pairs,                     # read each k-mer once for the join, 16 B per emitted pair
        'align': float(np.sum((lens[tasks['q']] + lens[tasks['r']]) / 4.0 + 20.0)),
    }
    b_pre = float(np.sum(lens / 4.0 + 16.0 * np.maximum(lens - k + 1, 0))) + 16.0 * n_pairs
    return sb, b_pre, sb['align']


def roofline_of(prof, steps, step_s, stage_bytes, b_pre, b_aln, world, workload_key):
    """The `roofline` object of a run: the dominant kernel of the profile scopes priced on its stage's SURVEY 8(d) bytes."""
    if not prof:
        return None
    per_step = {e['name']: e['total_ms'] / steps for e in prof}
    dom = max((e for e in prof if e['name'] != 'exchange'), key=lambda e: e['total_ms'])      # (a kernel of this rank, not a collective)
    kern, stage = SCOPES.get(dom['name'], (dom['name'], 'align'))
    launches_per_step = dom['launches'] / steps
    avg_ms = dom['total_ms'] / dom['launches']
    stage_ms = sum(v for k, v in per_step.items() if SCOPES.get(k, (k, 'align'))[1] == stage)
    stage_frac = stage_bytes[stage] / max(world, 1) / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if stage == 'index':
        # the inverted index is ONE logical pass of SURVEY 8(d) (16 B per position) implemented as several
        # kernels (partition levels + bucket sort): a kernel of it is priced with its time share of the stage
        alg = stage_bytes[stage] / max(world, 1) * (dom['total_ms'] / steps / stage_ms) / launches_per_step
        basis = ('SURVEY 8(d) bytes of the "index" stage (16 B per position) x this kernel\'s share of the stage time, per launch, '
                 '/ its HIP-event time on the library stream: equals the stage\'s fraction')
    else:
        alg = stage_bytes[stage] / launches_per_step / max(world, 1)
        basis = f'SURVEY 8(d) bytes of the "{stage}" stage, per launch, / this kernel\'s HIP-event time on the library stream'
    achieved = alg / (avg_ms * 1e-3) / 1e9
    impl = dom['bytes'] / dom['launches']
    # the same pricing for every scope of the step (which kernel is the longest flips between k_lz_parse_fast and k_bucket_runs
    # with the placement state of the box: the reader finds both here whichever leads)
    def priced(e):
        kn, stg = SCOPES.get(e['name'], (e['name'], 'align'))
        lps = e['launches'] / steps; a_ms = e['total_ms'] / e['launches']
        s_ms = sum(v for k_, v in per_step.items() if SCOPES.get(k_, (k_, 'align'))[1] == stg)
        # (a kernel of a stage that is ONE logical pass of SURVEY 8(d) implemented as several kernels is priced with its time share of the stage)
        ab = stage_bytes[stg] / max(world, 1) * (e['total_ms'] / steps / s_ms) / lps
        if stg == 'align' and e['name'] == 'lz_parse':
            ab = stage_bytes[stg] / max(world, 1) / lps      # (the parse alone carries B_aln, as the headline kernel of rounds 2-5 did)
        return dict(scope=e['name'], kernel=kn, stage=stg, avg_launch_ms=round(a_ms, 4), launches_per_step=round(lps, 3),
                    algorithmic_bytes_per_launch=round(ab), achieved=round(ab / (a_ms * 1e-3) / 1e9, 3), frac=round(ab / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6))
    all_kernels = [priced(e) for e in sorted((e for e in prof if e['name'] != 'exchange'), key=lambda e: -e['total_ms'])]
    traffic, src = pmc_traffic(workload_key, kern, launches_per_step) if world == 1 else (None, None)
    return dict(
        bound='hbm', kernel=kern, scope=dom['name'], achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit='GB/s',
        frac=round(achieved / HBM_PEAK_GBS, 6), traffic=traffic, traffic_source=src,
        avg_launch_ms=round(avg_ms, 4), launches_per_step=round(launches_per_step, 3),
        algorithmic_bytes_per_launch=round(alg), basis=basis,
        stage=dict(name=stage, ms_per_step=round(stage_ms, 3), algorithmic_bytes_per_step=round(stage_bytes[stage] / max(world, 1)),
                   frac=round(stage_frac, 6)),
        implementation_bytes_per_launch=round(impl),
        implementation_frac=round(impl / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
        kernels=all_kernels,
        ms_per_step_by_scope={k: round(v, 3) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
        host_ms_per_step=round(step_s * 1e3 - sum(per_step.values()), 3),
        path=dict(algorithmic_bytes_per_step=round(b_pre + b_aln), achieved=round((b_pre + b_aln) / step_s / 1e9, 3),
                  frac=round((b_pre + b_aln) / step_s / 1e9 / HBM_PEAK_GBS, 6),
                  note='B_pre + B_aln of SURVEY 8(d) / whole step incl. host time' + (', all ranks' if world > 1 else '')))


def side_workload(name, comm, k, min_ident, steps=8, warmup=3):
    """BASELINE.json configs[1] / [2] in the driver-run record: the same step (prefilter -> thresholds -> align, inputs resident
    in HBM) on another workload, a few steps behind the headline's, with its own roofline object."""
    import torch
    wl = synth.WORKLOADS[name]
    codes, offsets, names, desc = synth.make_workload(name)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    gs.to_device()
    lens = gs.lengths()
    st = {}

    def step():
        sizes, pairs = D.prefilter_counts(gs, comm, k, 1.0, min_shared=20)
        cand = gs.filter_pairs(sizes, pairs, k=k, min_kmers=20, min_ident=min_ident)
        tasks, stats = D.align_pairs(gs, cand, comm)
        st.update(tasks=tasks, n_pairs=len(tasks) // 2)
    api.profile_enable(False)
    for _ in range(warmup):
        step()
    api.profile_enable(True); api.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = api.profile_get(); api.profile_enable(False)
    sb, b_pre, b_aln = stage_bytes_of(lens, st['tasks'], st['n_pairs'], k)
    rf = roofline_of(prof, steps, dt / steps, sb, b_pre, b_aln, 1, name)
    out = dict(workload=f'{desc}, k={k}, min-kmers=20, min-ident={min_ident}, lz defaults', genomes=int(len(gs)), total_bases=int(lens.sum()),
               pairs_per_step=int(st['n_pairs']), steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 3),
               value=round(st['n_pairs'] * steps / dt, 1), unit='pairs/s', roofline=rf)
    gs.close()
    return out


def out_aln_leg(gs, tasks, plain_parse_ms):
    """--out-aln at the headline's size: rows AND regions of all tasks from ONE parse (vg_lz_align with regions);
    kernel times from the library's HIP-event scopes, against the stats-only parse of the timed steps."""
    res = {}
    for rep in range(2):              # (the first call pays for the arena's allocation)
        api.profile_enable(True); api.profile_reset()
        t0 = time.perf_counter()
        stats, regions = gs.lz_align(tasks, want_regions=True)
        wall = time.perf_counter() - t0
        prof = {e['name']: e for e in api.profile_get()}
        api.profile_enable(False)
        parse = prof.get('lz_parse', {}).get('total_ms', 0.0)
        place = prof.get('lz_regions_place', {}).get('total_ms', 0.0)
        res = dict(regions=int(len(regions)), regions_per_task=round(len(regions) / max(1, len(tasks)), 2),
                   parse_with_regions_ms=round(parse, 3), place_ms=round(place, 3), parse_launches=prof.get('lz_parse', {}).get('launches'),
                   stats_only_parse_ms=round(plain_parse_ms, 3), ratio_parse=round(parse / plain_parse_ms, 3) if plain_parse_ms else None,
                   ratio_parse_and_place=round((parse + place) / plain_parse_ms, 3) if plain_parse_ms else None,
                   call_wall_ms=round(wall * 1e3, 1),
                   note='vg_lz_align(..., regions): ONE parse launch writes rows and regions (chunks behind a cursor, placed by k_regions_place once '
                        'the rows are known); call_wall_ms includes the index build, the download of the regions (24 B each) and the host copy')
        assert int(stats['n_regions'].sum()) == len(regions)
        del regions
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', choices=sorted(synth.WORKLOADS), default='phage-100k')
    ap.add_argument('--count', type=int, default=None, help='scale the workload: families (phage sets) or contigs')
    ap.add_argument('--scaling', choices=['strong', 'weak'], default='strong',
                    help='N > 1: strong = the same set on N GPUs; weak = N x the set')
    ap.add_argument('--k', type=int, default=25)
    ap.add_argument('--min-kmers', type=int, default=None, help='default 20 (30 for contigs-1M, large.yml:65-72)')
    ap.add_argument('--min-ident', type=float, default=0.7)
    ap.add_argument('--cpu-sample-families', type=int, default=2000, help='families of the set the CPU baseline runs on (2 000 = 20 000 genomes: a fifth of phage-100k)')
    ap.add_argument('--cpu-threads', type=int, default=0, help='OpenMP threads of the CPU baseline (default: what the process may use: affinity and cgroup quota)')
    ap.add_argument('--cpu-reps', type=int, default=3, help='repetitions of the CPU baseline (value = the median run)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cli-wall', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true', help='skip configs[1] / [2] (phage-1k, imgvr-10k) behind the headline step')
    ap.add_argument('--no-out-aln', action='store_true', help='skip the --out-aln leg (rows + regions from one parse)')
    ap.add_argument('--placement-trials', type=int, default=4, help='placements of the prefilter workspace the first pass may try (vg_set_placement_trials; 1 = none)')
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
    # Placement trials are OPT-IN in the library (vg_set_placement_trials): the first pass of this long-lived process may try
    # up to --placement-trials placements of its workspace (~0.25 s each, once).  They need a warm-up step to
    # fall into (else they would run inside the timed region) and one rank; the line says what was asked for.
    trials = args.placement_trials if (args.warmup > 0 and world == 1) else 1
    api.set_placement_trials(trials)
    # N > 1 measures the RCCL path or nothing: the built-in communicator is created strictly (no silent fall-back to host
    # all-gathers through torch.distributed; VCLUST_COMM / VCLUST_DIST_BACKEND=gloo override it for tests on one GPU)
    kind = os.environ.get('VCLUST_COMM') or ('rccl-strict' if world > 1 and dist is not None and dist.get_backend() == 'nccl' else None)
    rccl_failure = None
    try:
        comm = D.make_comm(dist, dev, kind=kind)
    except RuntimeError as exc:
        # (strict creation failed on EVERY rank -- make_comm agrees on it first --: the run goes on over the callback
        # communicator so that the record holds numbers, and says in so many words that they are NOT an RCCL result)
        if kind != 'rccl-strict':
            raise
        rccl_failure = str(exc)
        comm = D.make_comm(dist, dev, kind='callback')

    wl = synth.WORKLOADS[args.workload]
    base_n = args.count if args.count is not None else (wl['n_families'] if wl['kind'] == 'families' else wl['n'])
    n_units = base_n * (world if args.scaling == 'weak' else 1)
    codes, offsets, names, desc = synth.make_workload(args.workload, n_units)
    min_kmers = args.min_kmers if args.min_kmers is not None else (30 if args.workload == 'contigs-1M' else 20)
    # End-to-end CLI leg FIRST, while this process has not touched the HBM yet: the driver scrubs device memory
    # that moves between processes, so CLI processes started right after this process has used ~150 GB measure
    # the scrub (observed for the same work: 2.7 s on a quiet device, 6.7-8.4 s right after the timed loop)
    e2e = None; cli_td = None; cli_ani = None
    if world == 1 and rank == 0 and not args.no_cli_wall:
        try:
            e2e, cli_td, cli_ani = cli_wall(codes, offsets, names, None)
        except Exception as exc:      # the device-resident figure stands on its own
            e2e = dict(error=str(exc))
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
        # -- prefilter: this rank's k-mer range; partial counts add up across ranks (RCCL all-gather)
        sizes, pairs = D.prefilter_counts(gs, comm, args.k, 1.0, min_shared=min_kmers)
        cand = gs.filter_pairs(sizes, pairs, k=args.k, min_kmers=min_kmers, min_ident=args.min_ident)
        # -- align from the candidate pairs: a rank lists ITS tasks (reference-range share) from the pairs and starts its
        #    index build and parse at once; the canonical task list of the whole set is assembled on the host beside the
        #    kernels and places the rows (gathered over RCCL when there is more than one rank)
        tasks, stats = D.align_pairs(gs, cand, comm)
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
    dt_local = dt
    if dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    prof = api.profile_get()
    api.profile_enable(False)
    # every rank reports its own stage times (stderr, one line each) and rank 0 carries them in the JSON line: a
    # scaling run that goes wrong is diagnosable from its tail
    xch = [e for e in prof if e['name'] == 'exchange']
    mine = dict(rank=rank, s_per_step=round(dt_local / args.steps, 6), pairs=state.get('n_pairs'),
                comm=comm.kind, rccl_ranks=comm.rccl_ranks,
                # RCCL collectives of this rank (HIP events around them on the library stream): ms, calls and bytes received per step
                exchange=dict(ms_per_step=round(sum(e['total_ms'] for e in xch) / args.steps, 3),
                              collectives_per_step=round(sum(e['launches'] for e in xch) / args.steps, 2),
                              bytes_per_step=round(sum(e['bytes'] for e in xch) / args.steps)) if world > 1 or xch else None,
                ms_per_step_by_scope={e['name']: round(e['total_ms'] / args.steps, 3) for e in prof})
    print(f'[bench rank {rank}/{world}] ' + json.dumps(mine), file=sys.stderr, flush=True)
    per_rank = [mine]
    if dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0 and cli_ani is not None:
        # the file the two cold CLI processes wrote against the rows of the timed step: ani.tsv written from the step's
        # integers by the same writer must be the same bytes (the cold path -- RANGE sub-shards under the workspace budget,
        # 6 GiB index batches -- and the resident path agree at the benchmark's own size)
        try:
            import filecmp
            mem_ani = os.path.join(cli_td.name, 'step.ani.tsv')
            gs.write_ani(mem_ani, state['tasks'], state['stats'])
            e2e['cli_rows_equal_step'] = bool(filecmp.cmp(cli_ani, mem_ani, shallow=False))
        except Exception as exc:
            e2e['cli_rows_equal_step'] = f'not compared: {exc}'
        cli_td.cleanup()
    if rank == 0:
        n_pairs = state['n_pairs']
        tk = state['tasks']
        step_s = dt / args.steps
        stage_bytes, b_pre, b_aln = stage_bytes_of(lens, tk, n_pairs, args.k)
        roofline = roofline_of(prof, args.steps, step_s, stage_bytes, b_pre, b_aln, world,
                               args.workload if args.count is None else f'{args.workload}/{args.count}')
        # -- behind the timed region (nothing below touches `value`): --out-aln from one parse at this size, then configs[1] / [2]
        out_aln = None
        if world == 1 and not args.no_out_aln:
            try:
                plain = next((e['total_ms'] / args.steps for e in prof if e['name'] == 'lz_parse'), 0.0)
                out_aln = out_aln_leg(gs, tk, plain)
            except Exception as exc:
                out_aln = dict(error=str(exc))
        others = None
        if world == 1 and not args.no_other_workloads and args.workload == 'phage-100k' and args.count is None:
            others = {}
            for name in ('phage-1k', 'imgvr-10k'):
                try:
                    others[name] = side_workload(name, comm, args.k, args.min_ident)
                except Exception as exc:
                    others[name] = dict(error=str(exc))
        cpu = None
        if world == 1 and not args.no_cpu_baseline and wl['kind'] == 'families':
            hc = host_cpus()
            cpu = cpu_baseline(min(args.cpu_sample_families, n_units), wl['members'], wl['length'], wl['seed'],
                               args.cpu_threads or hc['threads'], args.k, min_kmers, args.min_ident, n_units, reps=args.cpu_reps)
            cpu['host_cpus'] = hc
        out = {
            'metric': 'genome pairs/sec through prefilter+align (ani.tsv)',
            'value': round(n_pairs * args.steps / dt, 3),
            'unit': 'pairs/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(step_s * 1e3, 3),
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'u64',
            'data': 'synthetic',
            'config': {
                'workload': f'{desc}, k={args.k}, min-kmers={min_kmers}, min-ident={args.min_ident}, lz defaults',
                'genomes': int(len(gs)), 'pairs_per_step': int(n_pairs), 'total_bases': int(lens.sum()),
                'sha256': synth.sha256(codes, offsets) if len(codes) <= (1 << 33) else None,      # (the headline set: pinned in tests/golden/synth_sha256.json)
                'parallelism': f'kmer-range x{world} prefilter (bases scanned in {world} slices, kept masks all-to-all), reference-range x{world} align',
            },
            'roofline': roofline,
            'cpu_baseline': cpu,
            'cli_wall': e2e,
            'out_aln': out_aln,
            'other_workloads': others,
            'placement_trials': trials,
            'comm': dict(kind=comm.kind, rccl_ranks=comm.rccl_ranks, strict=(kind == 'rccl-strict'), rccl_failure=rccl_failure,
                         backend=(dist.get_backend() if dist is not None else None)) if world > 1 else None,
            'per_rank': per_rank if world > 1 else None,
        }
        if cpu:
            # the north_star targets as numbers (>= 10x the CPU path at one GPU).  `vs_baseline` stays null: BASELINE.md
            # holds no published figure for this metric, and the reference's own binaries cannot be built or run here
            out['vs_cpu_baseline'] = dict(
                device_resident=round(out['value'] / cpu['value'], 1),
                end_to_end_cli=round(e2e['pairs_per_s'] / cpu['value'], 1) if e2e and 'pairs_per_s' in e2e else None,
                label='GPU pairs/s / cpu_baseline.value (median of its repetitions): vs OWN CPU port (oracle/, every stage on cpu_baseline.cores threads), sample-extrapolated -- not the upstream binaries')
        print(json.dumps(out))
    comm.close()
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
