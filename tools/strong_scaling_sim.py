"""Developer tool: per-rank work of a strong-scaling run, emulated on ONE GPU: EVERY rank's shard of the phage-100k set (NF
families) for world = 1, 2, 4, 8 (RANK_SIM=r: only rank min(r, world - 1)) -- the step of a real run is its SLOWEST rank, which
is what the summary line of each world reports -- its k-mer RANGE shard of the prefilter and its reference range of the align tasks --
with the host work every rank repeats between the stages timed beside it (thresholds and the listing of its own tasks;
the canonical task list runs on a helper thread beside the kernels, as in vg_lz_align_pairs_sharded).

SCAN=sliced (default): the multi-GPU form of the shard pass -- the rank scans 1/world of the BASES (k_slice_scan) and
receives the kept masks + level-1 counts of its k-mer range; the peers' slices are computed by this process
(vg_set_range_scan(1)) and their time (scope emulated_peer_scan) is taken out of the rank's wall time.  SCAN=replicated:
every rank scans all bases (round 4).  The exchanges themselves do not run here; they are MODELLED from their sizes
(XGMI_GBS per link and direction, LAT_US per collective) and printed on their own line."""
import os, sys, pathlib, time, threading
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth, distributed as D
api.set_device(0)
NF = int(os.environ.get('NF', '10000'))
SCAN = os.environ.get('SCAN', 'sliced')
XGMI_GBS = float(os.environ.get('XGMI_GBS', '48'))       # one link, one direction, what RCCL point-to-point reaches (assumed)
LAT_US = float(os.environ.get('LAT_US', '40'))          # launch + completion of one small RCCL collective (assumed)
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
tasks = gs.align_tasks(gs.filter_pairs(sizes, pairs))
P = int(np.sum((np.diff(offsets) + 63) // 64 * 64))
base = None
ONLY = os.environ.get('RANK_SIM')
for world in (1, 2, 4, 8):
  ranks = [min(int(ONLY), world - 1)] if ONLY is not None else list(range(world))
  totals = []
  for rank in ranks:
    best = None
    api.set_range_scan(1 if SCAN == 'sliced' else 0)
    for it in range(3):
        api.profile_enable(True); api.profile_reset()
        t0 = time.perf_counter()
        s, p = gs.kmer_shared(k=25, shard=rank, n_shards=world, min_shared=1 if world > 1 else 20)
        t1 = time.perf_counter()
        peers_ms = sum(e['total_ms'] for e in api.profile_get() if e['name'] == 'emulated_peer_scan')
        # host work of every rank between the stages, as vg_lz_align_pairs_sharded does it: thresholds on the global pair
        # list, the rank's own tasks listed from the pairs (reference ranges from the genomes' task counts), and the
        # canonical task list of the whole set on a helper thread BESIDE the kernels
        cand = gs.filter_pairs(sizes, pairs)
        mine = D.align_pairs_share(gs, cand, world, rank)
        t2 = time.perf_counter()
        th = threading.Thread(target=lambda: gs.align_tasks(cand)); th.start()
        st = gs.lz_align(mine)
        th.join()
        t3 = time.perf_counter()
        prof = {e['name']: round(e['total_ms'], 1) for e in api.profile_get()}
        cur = ((t1 - t0) * 1e3 - peers_ms, (t2 - t1) * 1e3, (t3 - t2) * 1e3, prof, len(p))
        if best is None or cur[0] + cur[1] + cur[2] < best[0] + best[1] + best[2]: best = cur
    api.set_range_scan(0)
    tot = best[0] + best[1] + best[2]
    if base is None: base = tot
    # the exchanges of one step, modelled: masks + level-1 table all-to-all (received bytes over world - 1 links side by
    # side), then set sizes, nomination count, nominations, union counts (all-gathers of a few MB at most) and the rows
    xch = None
    if world > 1:
        recv = (P / 8 + P / 131072 * (2048 / world) * 4) * (world - 1) / world          # bytes this rank receives in the all-to-all
        per_link = recv / (world - 1)
        a2a_ms = per_link / (XGMI_GBS * 1e9) * 1e3 + LAT_US * 1e-3
        small = 8 * len(sizes) + 8 * len(pairs) + 4 * len(pairs) + 12 * 2 * len(pairs)   # sizes, keys, counts, rows: bytes per rank, upper bound
        gathers_ms = 7 * LAT_US * 1e-3 + small * (world - 1) / world / (XGMI_GBS * 1e9) * 1e3
        xch = (a2a_ms if SCAN == 'sliced' else 0.0, gathers_ms, recv if SCAN == 'sliced' else 0)
    kernels = sum(v for k_, v in best[3].items() if k_ != 'emulated_peer_scan')
    line = (f'world {world} rank {rank} [{SCAN}]: prefilter shard {best[0]:.1f} ms ({best[4]} partial pairs)  host between the stages {best[1]:.1f} ms  '
            f'align share {best[2]:.1f} ms  total {tot:.1f} ms = {base / tot:.2f}x  (kernels {kernels:.1f} ms, host + gaps {tot - kernels:.1f} ms)')
    if xch:
        line += (f'  | exchanges MODELLED ({XGMI_GBS:g} GB/s per link and direction, {LAT_US:g} us per collective): mask all-to-all {xch[0]:.2f} ms '
                 f'({xch[2] / 1e6:.0f} MB received), 7 small collectives {xch[1]:.2f} ms -> total {tot + xch[0] + xch[1]:.1f} ms = {base / (tot + xch[0] + xch[1]):.2f}x')
    print(line + f'  {best[3]}', flush=True)
    totals.append((tot, tot + (xch[0] + xch[1] if xch else 0.0), rank, tot - kernels))
  slow = max(totals)
  print(f'== world {world}: SLOWEST of {len(totals)} emulated rank(s) = rank {slow[2]}: {slow[0]:.1f} ms = {base / slow[0]:.2f}x; with the modelled exchanges {max(t[1] for t in totals):.1f} ms = '
        f'{base / max(t[1] for t in totals):.2f}x; fastest rank {min(totals)[0]:.1f} ms; host + gaps of the slowest {slow[3]:.1f} ms', flush=True)
