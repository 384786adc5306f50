"""RCCL smoke test of the built-in communicator (vg_comm_rccl_create): the sharded C-ABI entry points on the
real RCCL with the ranks torchrun gives it must reproduce the plain single-process results.  With one rank (a
one-GPU box) VG_DIST_FORCE=1 keeps every exchange and merge on the path: ncclAllGather with nranks = 1, the staging
through HBM, the nomination / union / count-sum protocol of the prefilter and the row / region gathers of align."""
import os, sys, pathlib
os.environ.setdefault('VG_DIST_FORCE', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import ctypes as C
import numpy as np
from vclust_amd import api, synth, _lib, distributed as D
rank, world, local_rank = D.dist_env()
api.set_device(local_rank % api.device_count())
lib = _lib.load()
if world == 1:
    # no process group needed: the unique id never leaves this process
    buf = (C.c_uint8 * 128)(); _lib.check(lib.vg_rccl_unique_id(buf, 128))
    h = C.c_void_p(); _lib.check(lib.vg_comm_rccl_create(0, 1, bytes(buf), 128, C.byref(h)))
    comm = D.Comm(h, lib); dist = None
else:
    dist, device = D.init_process_group('nccl')
    comm = D.make_comm(dist, device, kind='rccl')
comm.selftest(4096)
codes, offsets, names = synth.make_families(6, 5, length=6000, seed=4)
gs = api.GenomeSet.from_codes(codes, offsets, names)
s0, p0 = gs.kmer_shared(k=25, min_shared=1)
s1, p1 = D.prefilter_counts(gs, comm, 25, 1.0)
assert np.array_equal(s0, s1)
s20, p20 = gs.kmer_shared(k=25, min_shared=20)
s21, p21 = D.prefilter_counts(gs, comm, 25, 1.0, min_shared=20)
o0 = np.lexsort((p20['b'], p20['a'])); o1 = np.lexsort((p21['b'], p21['a']))
assert np.array_equal(s20, s21) and np.array_equal(p20[o0], p21[o1]) and len(p21) > 0
k0 = np.sort((p0['a'].astype(np.int64) << 32) | p0['b']); k1 = np.sort((p1['a'].astype(np.int64) << 32) | p1['b'])
assert np.array_equal(k0, k1) and int(p0['shared'].sum()) == int(p1['shared'].sum())
cand = gs.filter_pairs(s1, p1)
tasks = gs.align_tasks(cand)
st0, rg0 = gs.lz_align(tasks, want_regions=True)
st1, rg1 = D.align_rows(gs, tasks, comm, None, True)
assert np.array_equal(st0, st1) and len(rg0) == len(rg1)
st2, _ = D.align_rows(gs, tasks, comm, None, False)
assert np.array_equal(st0, st2)
comm.close()
if dist:
    dist.barrier(); dist.destroy_process_group()
if rank == 0:
    print('rccl ok: world', world, len(p1), 'pairs', len(tasks), 'tasks')
