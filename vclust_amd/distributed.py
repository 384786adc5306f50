"""One process per GPU: sharding of the integer work and the RCCL gather of result rows.

The hot path shards without any data-path collective inside the kernels:
  * prefilter: rank r handles the k-mers whose (second) hash falls into range r of `world` (set sizes
    and shared counts of the shards add up); one variable-length all-gather of
    (a, b, shared) records (set sizes ride along as diagonal records), summed on the device;
  * align: tasks are dealt by reference range (every rank indexes 1/world of the genomes; the
    partition is a pure function of the task list); one all-gather of the 12-byte
    (n_match, aln_len, n_regions) rows (and of the regions when --out-aln is set).
`torch.distributed` with backend "nccl" is RCCL on ROCm; the same code runs with "gloo" on
CPU tensors, which is what the CPU tests use with a stand-in compute backend.
"""
import os

import numpy as np


# test hook: run the collectives even with a single rank (RCCL smoke test on a one-GPU box)
FORCE_COLLECTIVES = bool(int(os.environ.get('VCLUST_DIST_FORCE', '0')))


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


def _all_gather_flat(pad, dist, world):
    """all_gather of equal-sized 1-D tensors into one [world, len] tensor (one D2H copy afterwards)."""
    import torch
    out = torch.empty(world * pad.numel(), dtype=pad.dtype, device=pad.device)
    try:
        dist.all_gather_into_tensor(out, pad)
    except (RuntimeError, AttributeError, NotImplementedError):
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
        out = torch.cat(bufs)
    return out.view(world, pad.numel())


def gather_known(arr, sizes, dist, device, world):
    """All-gather of per-rank byte arrays whose lengths every rank already knows: one collective."""
    import torch
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, device=device, dtype=torch.uint8)
    if raw.size:
        pad[:raw.size] = torch.from_numpy(raw.copy()).to(device)
    host = _all_gather_flat(pad, dist, world).cpu().numpy()
    return [host[r, :sizes[r]] for r in range(world)]


def gather_rows(arr, dtype, dist, device, world):
    """Variable-length all-gather of a structured numpy array (lengths exchanged first); rank order."""
    import torch
    if dist is None or (world == 1 and not FORCE_COLLECTIVES):
        return arr
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    cnt = torch.tensor([raw.size], device=device, dtype=torch.int64)
    sizes = [int(x) for x in _all_gather_flat(cnt, dist, world).cpu().numpy().reshape(-1)]
    parts = gather_known(arr, sizes, dist, device, world)
    return np.concatenate([p.copy().view(dtype) for p in parts]) if parts else arr


def all_reduce_sum(arr, dist, device, world):
    import torch
    if world == 1 or dist is None:
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()


def merge_pair_counts(pairs):
    """Sum the shared counts of duplicate (a, b) entries (per-shard partial counts); host version."""
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


def ref_owner(tasks, world):
    """Owner rank of every task: references are cut into `world` contiguous id ranges holding about
    the same number of tasks each, so a rank indexes only its own references (1/world of the
    genomes) and the partition is known to every rank without communication."""
    if len(tasks) == 0:
        return np.zeros(0, dtype=np.int64)
    refs = tasks['r'].astype(np.int64)
    per_ref = np.bincount(refs)
    before = np.cumsum(per_ref) - per_ref                       # tasks on references with a smaller id
    owner_of_ref = np.minimum(world - 1, before * world // len(tasks))
    return owner_of_ref[refs]


# ------------------------------------------------------------------ sharded stages
def prefilter_counts(gs, dist, device, rank, world, k, fraction):
    """All ranks end up with the global set sizes and the global (a, b, shared) triples, (a, b) ascending.

    Rank r counts the k-mers of hash range r.  Its partial counts travel as (a << 32 | b, count)
    records; the per-genome set sizes ride along as diagonal records (i, i, size), so one padded
    all-gather carries everything.  The partial counts are summed on the device (sort + segment sum
    via torch), only the merged table comes back to the host."""
    import torch
    from . import api
    sizes, pairs = gs.kmer_shared(k=k, fraction=fraction, shard=rank, n_shards=world, min_shared=1)
    if dist is None or (world == 1 and not FORCE_COLLECTIVES):
        return sizes, pairs
    n = len(sizes)
    rec = np.empty((len(pairs) + n, 2), dtype=np.int64)
    rec[:len(pairs), 0] = (pairs['a'].astype(np.int64) << 32) | pairs['b'].astype(np.int64)
    rec[:len(pairs), 1] = pairs['shared']
    ids = np.arange(n, dtype=np.int64)
    rec[len(pairs):, 0] = (ids << 32) | ids
    rec[len(pairs):, 1] = sizes
    cnt = torch.tensor([rec.shape[0]], device=device, dtype=torch.int64)
    counts = _all_gather_flat(cnt, dist, world).reshape(-1)
    mx = int(counts.max().item())
    pad = torch.zeros(mx * 2, device=device, dtype=torch.int64)
    pad[:rec.size] = torch.from_numpy(rec.reshape(-1)).to(device)
    allrec = _all_gather_flat(pad, dist, world).view(world, mx, 2)
    valid = torch.arange(mx, device=device).unsqueeze(0) < counts.to(device).unsqueeze(1)
    keys, vals = allrec[..., 0][valid], allrec[..., 1][valid]
    uk, inv = torch.unique(keys, return_inverse=True)            # sorted
    sums = torch.zeros(uk.numel(), dtype=torch.int64, device=device).index_add_(0, inv, vals)
    uk, sums = uk.cpu().numpy(), sums.cpu().numpy()
    a, b = uk >> 32, uk & 0xffffffff
    diag = a == b
    gsizes = np.zeros(n, dtype=np.int64)
    gsizes[a[diag]] = sums[diag]
    out = np.zeros(int((~diag).sum()), dtype=api.PAIR_DTYPE)
    out['a'], out['b'], out['shared'] = a[~diag], b[~diag], sums[~diag]
    return gsizes, out


def align_rows(gs, tasks, dist, device, rank, world, lz, want_regions):
    """Every rank parses the tasks of its reference range; all ranks receive all rows in task order."""
    from . import api
    if dist is None or (world == 1 and not FORCE_COLLECTIVES):
        if want_regions:
            stats, regions = gs.lz_align(tasks, lz=lz, want_regions=True)
            return stats, regions
        return gs.lz_align(tasks, lz=lz), None
    owner = ref_owner(tasks, world)
    mine = np.flatnonzero(owner == rank)
    regions = None
    if want_regions:
        part, regions = gs.lz_align(tasks[mine], lz=lz, want_regions=True)
        regions = regions.copy()
        regions['task'] = mine[regions['task']].astype(np.uint32)
        regions = gather_rows(regions, api.REGION_DTYPE, dist, device, world)
    else:
        part = gs.lz_align(tasks[mine], lz=lz)
    row = np.dtype(api.STAT_DTYPE).itemsize
    counts = np.bincount(owner, minlength=world)
    parts = gather_known(part, [int(c) * row for c in counts], dist, device, world)
    stats = np.zeros(len(tasks), dtype=api.STAT_DTYPE)
    for r in range(world):
        stats[np.flatnonzero(owner == r)] = parts[r].copy().view(api.STAT_DTYPE)
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
