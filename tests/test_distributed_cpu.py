"""world_size-2 `gloo` test of the multi-process path (sharding, variable-length gathers,
partial-count merge).  The GPU kernels are replaced by a stand-in backend built on the CPU
oracle, so that the N>1 plumbing of vclust_amd/distributed.py runs on a box without GPUs."""
import os
import pathlib
import socket
import subprocess
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["VROOT"]); sys.path.insert(0, os.path.join(os.environ["VROOT"], "tests"))
import oracle_lib as orc
from vclust_amd import api, synth, distributed as D

M1, M2 = np.uint64(0xff51afd7ed558ccd), np.uint64(0xc4ceb9fe1a85ec53)
def mix64(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33); x *= M1; x ^= x >> np.uint64(33); x *= M2; x ^= x >> np.uint64(33)
    return x

class OracleBackend:
    """Stands in for api.GenomeSet: same methods, integers from the CPU oracle."""
    def __init__(self, codes, offsets):
        self.codes, self.offsets = codes, offsets
        self.n = len(offsets) - 1
    def seq(self, i):
        return self.codes[self.offsets[i]:self.offsets[i + 1]]
    def kmer_shared(self, k=25, fraction=1.0, shard=0, n_shards=1, min_shared=1):
        sets = []
        for i in range(self.n):
            s = orc.kmer_set(self.seq(i), k)
            h = mix64(s)
            own = ((h & np.uint64(0xffffffff)) * np.uint64(n_shards)) >> np.uint64(32)
            sets.append(s[own == np.uint64(shard)])
        sizes = np.array([len(s) for s in sets], dtype=np.int64)
        pairs = [(a, b, len(np.intersect1d(sets[a], sets[b], assume_unique=True)))
                 for a in range(self.n) for b in range(a)]
        pairs = np.array([p for p in pairs if p[2] >= min_shared], dtype=api.PAIR_DTYPE)
        return sizes, pairs
    def lz_align(self, tasks, lz=None, want_regions=False):
        out = np.zeros(len(tasks), dtype=api.STAT_DTYPE)
        for i, t in enumerate(tasks):
            out[i] = orc.lz_pair_stat(self.seq(int(t["q"])), self.seq(int(t["r"])))
        return out

dist, device = D.init_process_group("gloo")
rank, world, _ = D.dist_env()
codes, offsets, names = synth.make_families(2, 3, length=3000, seed=5)
gs = OracleBackend(codes, offsets)
sizes, pairs = D.prefilter_counts(gs, dist, device, rank, world, 25, 1.0)
cand = pairs[pairs["shared"] >= 20]
order = np.argsort(-np.diff(offsets), kind="stable"); rk = {int(g): r for r, g in enumerate(order)}
couples = sorted((min(rk[int(p["a"])], rk[int(p["b"])]), max(rk[int(p["a"])], rk[int(p["b"])])) for p in cand)
tasks = np.array([t for lo, hi in couples for t in ((order[hi], order[lo]), (order[lo], order[hi]))], dtype=api.TASK_DTYPE)
stats, _ = D.align_rows(gs, tasks, dist, device, rank, world, None, False)
if rank == 0:
    json.dump(dict(sizes=sizes.tolist(), pairs=[[int(x) for x in p] for p in pairs],
                   tasks=[[int(x) for x in t] for t in tasks], stats=[[int(x) for x in s] for s in stats]),
              open(os.environ["VOUT"], "w"))
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_ranks_equal_one_rank(tmp_path):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
    import oracle_lib as orc
    from vclust_amd import synth, distributed as D
    script = tmp_path / 'worker.py'; script.write_text(WORKER)
    out = tmp_path / 'out.json'
    env = dict(os.environ, VROOT=str(ROOT), VOUT=str(out), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    import json
    got = json.load(open(out))
    codes, offsets, names = synth.make_families(2, 3, length=3000, seed=5)
    sizes, pairs = orc.shared_all(codes, offsets, k=25)
    assert got['sizes'] == sizes.tolist()
    assert {(a, b): s for a, b, s in got['pairs']} == pairs
    assert len(got['tasks']) == len(got['stats']) > 0
    for (q, r), st in zip(got['tasks'], got['stats']):
        assert tuple(st) == orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]])


def test_couple_range_partitions_everything():
    sys.path.insert(0, str(ROOT))
    from vclust_amd import distributed as D
    for n in (0, 1, 7, 4500):
        for world in (1, 2, 3, 8):
            cuts = [D.couple_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == 2 * n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert all(lo % 2 == 0 and hi % 2 == 0 for lo, hi in cuts)


def test_merge_pair_counts():
    sys.path.insert(0, str(ROOT))
    from vclust_amd import api, distributed as D
    p = np.array([(3, 1, 5), (2, 0, 1), (3, 1, 7), (2, 0, 2), (4, 3, 9)], dtype=api.PAIR_DTYPE)
    m = D.merge_pair_counts(p)
    assert {(int(x['a']), int(x['b'])): int(x['shared']) for x in m} == {(3, 1): 12, (2, 0): 3, (4, 3): 9}


def test_ref_owner_partitions_by_reference():
    sys.path.insert(0, str(ROOT))
    from vclust_amd import api, distributed as D
    rng = np.random.default_rng(3)
    tasks = np.zeros(5000, dtype=api.TASK_DTYPE)
    tasks['q'] = rng.integers(0, 300, 5000); tasks['r'] = rng.integers(0, 300, 5000) ** 2 // 300     # skewed
    for world in (1, 2, 3, 8):
        owner = D.ref_owner(tasks, world)
        assert owner.min() >= 0 and owner.max() <= world - 1
        for g in np.unique(tasks['r']):                          # a reference lives on exactly one rank
            assert len(np.unique(owner[tasks['r'] == g])) == 1
        refs_of = [set(tasks['r'][owner == r].tolist()) for r in range(world)]
        assert all(max(refs_of[i], default=-1) < min(refs_of[i + 1], default=1 << 30) for i in range(world - 1))
        counts = np.bincount(owner, minlength=world)
        assert counts.max() <= len(tasks) / world + np.bincount(tasks['r']).max()
    assert len(D.ref_owner(tasks[:0], 4)) == 0
