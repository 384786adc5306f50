"""One process per GPU: sharding of the integer work and the RCCL gather of result rows.

The hot path shards without any data-path collective inside the kernels:
  * prefilter: rank r handles the k-mers whose hash falls into range r of `world` (set sizes
    and shared counts of the shards add up); one variable-length all-gather of
    (a, b, shared) triples + one all-reduce of the per-genome set sizes;
  * align: the canonical ordered-pair list is cut into `world` contiguous pieces (couples kept
    together, consecutive tasks share references); one variable-length gather of the
    12-byte (n_match, aln_len, n_regions) rows (and of the regions when --out-aln is set).
`torch.distributed` with backend "nccl" is RCCL on ROCm; the same code runs with "gloo" on
CPU tensors, which is what the CPU tests use with a stand-in compute backend.
"""
import os

import numpy as np


def dist_env():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (dist, device)."""
    import torch
    import torch.distributed as dist
    rank, world, local_rank = dist_env()
    if backend is None:
        # VCLUST_DIST_BACKEND=gloo: host-side gathers (testing several ranks on one GPU box)
        backend = os.environ.get('VCLUST_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
        if not dist.is_initialized():
            dist.init_process_group('nccl', device_id=device)
    else:
        device = torch.device('cpu')
        if not dist.is_initialized():
            dist.init_process_group('gloo')
    return dist, device


def couple_range(n_couples, rank, world):
    """Contiguous share of the task couples (2 tasks each) for `rank`: [lo, hi) in TASK units."""
    lo = (n_couples * rank // world) * 2
    hi = (n_couples * (rank + 1) // world) * 2
    return lo, hi


def gather_rows(arr, dtype, dist, device, world):
    """Variable-length all-gather of a structured numpy array; rows keep rank order."""
    import torch
    if world == 1 or dist is None:
        return arr
    raw = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).copy()).to(device)
    cnt = torch.tensor([raw.numel()], device=device, dtype=torch.int64)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    sizes = [int(c.item()) for c in cnts]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, device=device, dtype=torch.uint8)
    pad[:raw.numel()] = raw
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    parts = [bufs[r][:sizes[r]].cpu().numpy().view(dtype) for r in range(world)]
    return np.concatenate(parts) if parts else arr


def all_reduce_sum(arr, dist, device, world):
    import torch
    if world == 1 or dist is None:
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()


def merge_pair_counts(pairs):
    """Sum the shared counts of duplicate (a, b) entries (per-shard partial counts)."""
    if len(pairs) == 0:
        return pairs
    key = (pairs['a'].astype(np.uint64) << np.uint64(32)) | pairs['b'].astype(np.uint64)
    uk, inv = np.unique(key, return_inverse=True)
    shared = np.zeros(len(uk), dtype=np.int64)
    np.add.at(shared, inv, pairs['shared'].astype(np.int64))
    out = np.zeros(len(uk), dtype=pairs.dtype)
    out['a'] = (uk >> np.uint64(32)).astype(np.uint32)
    out['b'] = (uk & np.uint64(0xffffffff)).astype(np.uint32)
    out['shared'] = shared.astype(np.uint32)
    return out


# ------------------------------------------------------------------ sharded stages
def prefilter_counts(gs, dist, device, rank, world, k, fraction):
    """All ranks end up with the global set sizes and the global (a, b, shared) triples."""
    from . import api
    sizes, pairs = gs.kmer_shared(k=k, fraction=fraction, shard=rank, n_shards=world, min_shared=1)
    sizes = all_reduce_sum(sizes, dist, device, world)
    pairs = merge_pair_counts(gather_rows(pairs, api.PAIR_DTYPE, dist, device, world))
    return sizes, pairs


def align_rows(gs, tasks, dist, device, rank, world, lz, want_regions):
    """Every rank parses its contiguous share; all ranks receive all rows (rank order == task order)."""
    from . import api
    lo, hi = couple_range(len(tasks) // 2, rank, world)
    if want_regions:
        stats, regions = gs.lz_align(tasks[lo:hi], lz=lz, want_regions=True)
        regions = regions.copy()
        regions['task'] += np.uint32(lo)
        regions = gather_rows(regions, api.REGION_DTYPE, dist, device, world)
    else:
        stats, regions = gs.lz_align(tasks[lo:hi], lz=lz), None
    stats = gather_rows(stats, api.STAT_DTYPE, dist, device, world)
    return stats, regions


def prefilter(paths, out_path, is_multifasta, k=25, min_kmers=20, min_ident=0.7, kmers_fraction=1.0, max_seqs=0,
              num_threads=1):
    from . import api
    dist, device = init_process_group()
    rank, world, local_rank = dist_env()
    api.set_device(local_rank % max(api.device_count(), 1))
    gs = api.GenomeSet.load(paths, is_multifasta, n_threads=num_threads)
    sizes, pairs = prefilter_counts(gs, dist, device, rank, world, k, kmers_fraction)
    if rank == 0:
        gs.write_fltr(out_path, sizes, pairs, k=k, fraction=kmers_fraction, min_kmers=min_kmers, min_ident=min_ident,
                      max_seqs=max_seqs)
    dist.barrier()


def align(paths, out_path, is_multifasta, columns, filter_path=None, filter_threshold=0.0, out_aln=None, lz=None,
          out_filters=None, num_threads=1):
    from . import api
    dist, device = init_process_group()
    rank, world, local_rank = dist_env()
    api.set_device(local_rank % max(api.device_count(), 1))
    gs = api.GenomeSet.load(paths, is_multifasta, n_threads=num_threads)
    tasks = gs.align_tasks(gs.read_filter(filter_path, filter_threshold))
    stats, regions = align_rows(gs, tasks, dist, device, rank, world, lz, out_aln is not None)
    if rank == 0:
        gs.write_ani(out_path, tasks, stats, regions=regions, columns=columns, out_aln=out_aln, lz=lz,
                     out_filters=out_filters)
    dist.barrier()
