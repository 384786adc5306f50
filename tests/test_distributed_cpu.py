"""world_size-2 `gloo` tests of the multi-process plumbing on a box without GPUs: the vg_comm callback
communicator over torch.distributed, the agreement on failures (no rank is left inside a collective), and
the reference-range partition of the align stage (vg_align_owner).  The sharded compute itself needs a GPU
and is covered by tests/test_gpu_cli.py (two ranks on one GPU over gloo; RCCL when two devices are visible)."""
import os
import pathlib
import socket
import subprocess
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys, json
import ctypes as C
sys.path.insert(0, os.environ["VROOT"])
from vclust_amd import api, _lib, distributed as D

dist, device = D.init_process_group("gloo")
rank, world, _ = D.dist_env()
comm = D.make_comm(dist, device)                       # callback communicator: all-gather over gloo, host memory
assert (comm.rank, comm.world) == (rank, world)
comm.selftest(1)
comm.selftest(100003)                                  # odd size, larger than any internal chunk
res = {"selftest": True}
lib = _lib.load()
# a rank that fails during ingest makes EVERY rank return the error (nobody hangs in the next exchange)
fasta = os.environ["VFASTA"] if rank == 0 else os.environ["VFASTA"] + ".missing"
arr = (C.c_char_p * 1)(os.fsencode(fasta))
prm = _lib.PrefilterParams(25, 20, 0.7, 0, 1.0, 0, 1, 0, 1)
rc = lib.vg_prefilter_sharded(arr, 1, os.fsencode(os.environ["VOUT"] + ".fltr"), C.byref(prm), comm.h)
res["ingest_rc"] = rc; res["ingest_msg"] = lib.vg_last_error().decode()
# both ranks ingest fine, but there is no GPU here: every rank reports the same device error
arr = (C.c_char_p * 1)(os.fsencode(os.environ["VFASTA"]))
rc = lib.vg_prefilter_sharded(arr, 1, os.fsencode(os.environ["VOUT"] + ".fltr"), C.byref(prm), comm.h)
res["nodev_rc"] = rc; res["nodev_msg"] = lib.vg_last_error().decode()
p = api.align_params(api.ALIGN_FIELDS[:11])
rc = lib.vg_align_sharded(arr, 1, os.fsencode(os.environ["VOUT"] + ".ani"), C.byref(p), comm.h)
res["align_rc"] = rc
# a set large enough for the sliced k-mer scan (two partition levels: the ranks would exchange kept masks): the agreement in
# front of that exchange is paired by ranks that never reach it (here: both, there is no device), nobody hangs
import numpy as np
rng = np.random.default_rng(7)
lens = np.full(300, 10000); offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
gs = api.GenomeSet.from_codes(rng.integers(0, 4, int(offs[-1])).astype(np.uint8), offs, None)
try:
    D.prefilter_counts(gs, comm, 25, 1.0, min_shared=20); res["sliced_rc"] = 0
except _lib.VclustGpuError as e:
    res["sliced_rc"] = e.code; res["sliced_msg"] = str(e)
res["comm"] = comm.describe()
# the strict form of the built-in RCCL communicator: no device here, so creation fails on every rank and every rank raises
try:
    D.make_comm(dist, device, kind="rccl-strict"); res["strict"] = "created"
except RuntimeError as e:
    res["strict"] = str(e)
json.dump(res, open(os.environ["VOUT"] + f".{rank}.json", "w"))
comm.close()
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_ranks_comm_and_failure_agreement(tmp_path, golden_dir):
    import json
    script = tmp_path / 'worker.py'; script.write_text(WORKER)
    out = tmp_path / 'out'
    env = dict(os.environ, VROOT=str(ROOT), VOUT=str(out), VFASTA=str(golden_dir / 'multifasta.fna'), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(free_port()), WORLD_SIZE='2', HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = [json.load(open(f'{out}.{r}.json')) for r in range(2)]
    for r in range(2):
        assert got[r]['selftest']
        assert got[r]['ingest_rc'] != 0 and 'rank 1 failed' in got[r]['ingest_msg']     # the same verdict on both ranks
        assert got[r]['nodev_rc'] == -3 and 'no HIP device' in got[r]['nodev_msg'] or got[r]['nodev_rc'] != 0
        assert got[r]['align_rc'] != 0
        assert got[r]['sliced_rc'] == -3, got[r]
        assert got[r]['comm'] == dict(comm='callback', rccl_ranks=None, rank=r, world=2)
        assert 'rccl-strict' in got[r]['strict'] and 'not falling back' in got[r]['strict']
    assert got[0]['ingest_rc'] == got[1]['ingest_rc'] and got[0]['nodev_rc'] == got[1]['nodev_rc']
    assert not os.path.exists(f'{out}.fltr')


def _ref_owner(tasks, world):
    """The partition rule, restated: references cut into `world` contiguous id ranges with about equal task counts."""
    refs = tasks['r'].astype(np.int64)
    per_ref = np.bincount(refs)
    before = np.cumsum(per_ref) - per_ref
    return np.minimum(world - 1, before * world // len(tasks))[refs]


def test_align_owner_partitions_by_reference():
    sys.path.insert(0, str(ROOT))
    from vclust_amd import api, distributed as D
    rng = np.random.default_rng(3)
    tasks = np.zeros(5000, dtype=api.TASK_DTYPE)
    tasks['q'] = rng.integers(0, 300, 5000); tasks['r'] = rng.integers(0, 300, 5000) ** 2 // 300     # skewed
    for world in (1, 2, 3, 8):
        owner = D.align_owner(tasks, 300, world)
        assert np.array_equal(owner, _ref_owner(tasks, world))
        assert owner.min() >= 0 and owner.max() <= world - 1
        for g in np.unique(tasks['r']):                          # a reference lives on exactly one rank
            assert len(np.unique(owner[tasks['r'] == g])) == 1
        refs_of = [set(tasks['r'][owner == r].tolist()) for r in range(world)]
        assert all(max(refs_of[i], default=-1) < min(refs_of[i + 1], default=1 << 30) for i in range(world - 1))
        counts = np.bincount(owner, minlength=world)
        assert counts.max() <= len(tasks) / world + np.bincount(tasks['r']).max()
    assert len(D.align_owner(tasks[:0], 300, 4)) == 0


def test_align_pairs_share_is_the_owner_partition_of_the_task_list():
    """vg_align_pairs_share (a rank's tasks straight from the candidate pairs, what vg_lz_align_pairs_sharded lists before
    it launches) against vg_align_tasks + vg_align_owner: for every world size the ranks' shares are exactly the tasks the
    canonical list assigns to them (no GPU: host functions on a set made from codes)."""
    sys.path.insert(0, str(ROOT))
    from vclust_amd import api, distributed as D
    rng = np.random.default_rng(5)
    n = 400
    lens = rng.integers(30, 90, n)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    gs = api.GenomeSet.from_codes(rng.integers(0, 4, int(offsets[-1])).astype(np.uint8), offsets, ['g%d' % i for i in range(n)])
    a = rng.integers(1, n, 3000); b = (rng.integers(0, n, 3000) ** 2 // n) % a                      # a > b, skewed partners
    keys = np.unique(a.astype(np.int64) * n + b)
    cand = np.zeros(len(keys), dtype=api.PAIR_DTYPE); cand['a'] = keys // n; cand['b'] = keys % n
    tasks = gs.align_tasks(cand)
    assert len(tasks) == 2 * len(cand)
    for world in (1, 2, 3, 8):
        owner = D.align_owner(tasks, n, world)
        got_all = 0
        for rank in range(world):
            mine = D.align_pairs_share(gs, cand, world, rank)
            want = tasks[owner == rank]
            assert sorted(zip(mine['q'].tolist(), mine['r'].tolist())) == sorted(zip(want['q'].tolist(), want['r'].tolist())), (world, rank)
            got_all += len(mine)
        assert got_all == len(tasks)
    assert len(D.align_pairs_share(gs, cand[:0], 4, 1)) == 0
    # more than 2^17 pairs: the listing runs on several threads (per-chunk owner counts become starting positions) and must
    # give the same tasks IN THE SAME ORDER as one thread would (pair order, r = a before r = b)
    a = rng.integers(1, n, 400000); b = (rng.integers(0, n, 400000) ** 2 // n) % a
    keys = np.unique(a.astype(np.int64) * n + b)
    cand = np.zeros(len(keys), dtype=api.PAIR_DTYPE); cand['a'] = keys // n; cand['b'] = keys % n
    assert len(cand) >= (1 << 16)
    big = np.concatenate([cand, cand, cand])[:(1 << 17) + 777]              # (repeated pairs are fine for the listing)
    tasks_of = lambda c: np.stack([np.stack([c['b'], c['a']], 1), np.stack([c['a'], c['b']], 1)], 1).reshape(-1, 2)      # (q, r): (b, a) then (a, b)
    per_ref = np.bincount(np.concatenate([big['a'], big['b']]), minlength=n)
    for world in (3, 8):
        own = np.minimum(world - 1, (np.cumsum(per_ref) - per_ref) * world // (2 * len(big)))
        all_t = tasks_of(big)
        for rank in (0, world - 1):
            mine = D.align_pairs_share(gs, big, world, rank)
            want = all_t[own[all_t[:, 1]] == rank]
            assert np.array_equal(np.stack([mine['q'], mine['r']], 1), want), (world, rank)


def test_single_rank_comm_needs_no_process_group():
    sys.path.insert(0, str(ROOT))
    from vclust_amd import distributed as D
    comm = D.make_comm(None, None)
    assert (comm.rank, comm.world) == (0, 1)
    comm.selftest(64)
    comm.close()
