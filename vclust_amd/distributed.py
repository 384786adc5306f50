"""One process per GPU: the Python side of the sharded stages.

All sharding logic lives behind the C ABI (vclust_amd/csrc/vg_dist.hip: vg_kmer_shared_sharded,
vg_lz_align_sharded, vg_prefilter_sharded, vg_align_sharded); this module only builds the communicator
(vg_comm) they exchange their integer records through:

  * backend "nccl" (= RCCL on ROCm): the library's built-in RCCL communicator -- ncclAllGather on the
    library's own stream, device resident; torch.distributed is used once, to hand rank 0's 128-byte
    unique id to the other ranks;
  * backend "gloo" (CPU tests, several ranks sharing one GPU): a callback communicator whose all-gather runs
    over torch.distributed/gloo on host memory.

`torchrun ... vclust.py prefilter|align` and `bench.py --gpus N` go through here.
"""
import ctypes as C
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
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        device = torch.device('cuda', local_rank % max(torch.cuda.device_count(), 1))
        if not dist.is_initialized():
            dist.init_process_group('nccl', device_id=device)
    else:
        device = torch.device('cpu')
        if not dist.is_initialized():
            dist.init_process_group('gloo')
    return dist, device


class Comm:
    """Owner of a vg_comm handle (and of the Python callback it may point to)."""

    def __init__(self, handle, lib, keep=None):
        self.h, self._lib, self._keep = handle, lib, keep

    @property
    def kind(self):
        """'rccl' (the library's built-in RCCL communicator) or 'callback' (all-gathers handed to torch.distributed)."""
        return 'rccl' if self._lib.vg_comm_kind(self.h) == 1 else 'callback'

    @property
    def rccl_ranks(self):
        """ncclCommCount of the built-in communicator (None for a callback communicator)."""
        n = self._lib.vg_comm_rccl_ranks(self.h)
        return n if n >= 0 else None

    def describe(self):
        return dict(comm=self.kind, rccl_ranks=self.rccl_ranks, rank=self.rank, world=self.world)

    @property
    def rank(self):
        return self._lib.vg_comm_rank(self.h)

    @property
    def world(self):
        return self._lib.vg_comm_world(self.h)

    def selftest(self, nbytes=1000):
        from . import _lib
        _lib.check(self._lib.vg_comm_selftest(self.h, nbytes))

    def close(self):
        if self.h:
            self._lib.vg_comm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _host_view(ptr, nbytes):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(int(nbytes), 1),))[:int(nbytes)]


def make_comm(dist=None, device=None, kind=None):
    """vg_comm for the current process group.  kind: "rccl" (built-in RCCL communicator; any failure to create or
    self-test it sends every rank to the callback communicator with a line on stderr), "rccl-strict" (the same, but a
    failure raises on every rank instead: what a measurement of the RCCL path must use -- `bench.py --gpus N`),
    "callback" (all-gather through torch.distributed); default: $VCLUST_COMM, else rccl on the nccl backend, callback
    otherwise."""
    import torch
    from . import _lib
    lib = _lib.load()
    rank, world, _ = dist_env()
    h = C.c_void_p()
    if dist is None or world == 1:
        _lib.check(lib.vg_comm_create(0, 1, None, None, C.byref(h)))
        return Comm(h, lib)
    backend = dist.get_backend()
    if kind is None:
        kind = os.environ.get('VCLUST_COMM') or ('rccl' if backend == 'nccl' else 'callback')
    strict = kind == 'rccl-strict'
    if kind in ('rccl', 'rccl-strict'):
        # The built-in RCCL communicator, made fail-soft: no step may leave some ranks inside a collective while another
        # has raised.  Rank 0's id travels with a flag; creation and a first real exchange (vg_comm_selftest) are each
        # agreed on by all ranks before the next collective; any failure sends EVERY rank to the callback communicator
        # (all-gathers through torch.distributed) with a line on stderr.
        import sys

        def agree(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1
        uid = torch.zeros(129, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            if lib.vg_rccl_unique_id(buf, 128) == 0:
                uid = torch.from_numpy(np.frombuffer(bytes(buf) + b'\x01', dtype=np.uint8).copy())
            else:
                print(f'vclust_amd.distributed: no RCCL unique id ({lib.vg_last_error().decode()}): callback communicator', file=sys.stderr)
        if backend == 'nccl':
            uid = uid.to(device)
        dist.broadcast(uid, src=0)
        raw = bytes(uid.cpu().numpy().tobytes())
        made = raw[128] == 1 and lib.vg_comm_rccl_create(rank, world, raw[:128], 128, C.byref(h)) == 0
        if raw[128] == 1 and not made:
            print(f'vclust_amd.distributed: rank {rank}: vg_comm_rccl_create failed ({lib.vg_last_error().decode()})', file=sys.stderr)
        if agree(made):
            tested = lib.vg_comm_selftest(h, 1 << 16) == 0
            if not tested:
                print(f'vclust_amd.distributed: rank {rank}: RCCL exchange self-test failed ({lib.vg_last_error().decode()})', file=sys.stderr)
            if agree(tested):
                return Comm(h, lib)
        if made:
            lib.vg_comm_free(h)
        h = C.c_void_p()
        if strict:
            # (every rank is here: the failure was agreed on)
            raise RuntimeError('vclust_amd.distributed: VCLUST_COMM=rccl-strict and the built-in RCCL communicator could not be '
                               'created or failed its self-test on some rank (see stderr): not falling back')
        if rank == 0:
            print('vclust_amd.distributed: falling back to the callback communicator (torch.distributed all-gathers)', file=sys.stderr)

    def allgather(ctx, send, recv, nbytes, on_device):
        try:
            if on_device:
                host_send = np.empty(int(nbytes), dtype=np.uint8)
                _lib.hip_copy(host_send.ctypes.data, send, nbytes, to_host=True)
            else:
                host_send = _host_view(send, nbytes)
            t = torch.from_numpy(np.ascontiguousarray(host_send).copy())
            if backend == 'nccl':
                t = t.to(device)
            out = torch.empty(world * int(nbytes), dtype=torch.uint8, device=t.device)
            try:
                dist.all_gather_into_tensor(out, t)
            except (RuntimeError, AttributeError, NotImplementedError):
                bufs = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(bufs, t)
                out = torch.cat(bufs)
            res = out.cpu().numpy()
            if on_device:
                _lib.hip_copy(recv, res.ctypes.data, world * int(nbytes), to_host=False)
            else:
                _host_view(recv, world * int(nbytes))[:] = res
            return 0
        except Exception as exc:          # never let an exception cross the C boundary
            import sys
            print(f'vclust_amd.distributed: all-gather callback failed: {exc}', file=sys.stderr)
            return 1

    cb = _lib.ALLGATHER_FN(allgather)
    _lib.check(lib.vg_comm_create(rank, world, C.cast(cb, C.c_void_p), None, C.byref(h)))
    return Comm(h, lib, keep=cb)


# ------------------------------------------------------------------ sharded stages (in-memory)
def prefilter_counts(gs, comm, k, fraction, min_shared=1):
    """Global set sizes and (a, b, shared) triples on every rank (vg_kmer_shared_sharded)."""
    from . import _lib, api
    lib = _lib.load()
    sizes = np.zeros(max(len(gs), 1), dtype=np.int64)
    pp = C.POINTER(_lib.PairCount)(); n = C.c_int64()
    _lib.check(lib.vg_kmer_shared_sharded(gs._h, k, float(fraction), int(min_shared), comm.h,
                                          sizes.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(pp), C.byref(n)))
    pairs = api._take(pp, n.value, api.PAIR_DTYPE)
    return sizes[:len(gs)], pairs


def align_pairs_share(gs, cand, world, rank):
    """Rank `rank`'s tasks of the candidate pairs under the reference-range partition (vg_align_pairs_share)."""
    from . import _lib, api
    lib = _lib.load()
    cand = np.ascontiguousarray(cand, dtype=api.PAIR_DTYPE)
    tp = C.POINTER(_lib.Task)(); nt = C.c_int64()
    _lib.check(lib.vg_align_pairs_share(gs._h, cand.ctypes.data_as(C.POINTER(_lib.PairCount)), len(cand), int(world), int(rank), C.byref(tp), C.byref(nt)))
    return api._take(tp, nt.value, api.TASK_DTYPE)


def align_pairs(gs, cand, comm, lz=None):
    """Canonical task list and rows of the candidate pairs on every rank (vg_lz_align_pairs_sharded): a rank starts its
    kernels from the pairs alone, the task list of the whole set is assembled beside them."""
    from . import _lib, api
    lib = _lib.load()
    cand = np.ascontiguousarray(cand, dtype=api.PAIR_DTYPE)
    prm = _lib.LzParams(**{**api.DEFAULT_LZ, **(lz or {})})
    tp = C.POINTER(_lib.Task)(); nt = C.c_int64(); sp = C.POINTER(_lib.PairStat)()
    _lib.check(lib.vg_lz_align_pairs_sharded(gs._h, cand.ctypes.data_as(C.POINTER(_lib.PairCount)), len(cand), C.byref(prm), comm.h,
                                             C.byref(tp), C.byref(nt), C.byref(sp)))
    return api._take(tp, nt.value, api.TASK_DTYPE), api._take(sp, nt.value, api.STAT_DTYPE)


def align_rows(gs, tasks, comm, lz=None, want_regions=False):
    """Rows (and regions) of every task on every rank (vg_lz_align_sharded)."""
    from . import _lib, api
    lib = _lib.load()
    tasks = np.ascontiguousarray(tasks, dtype=api.TASK_DTYPE)
    stats = np.zeros(len(tasks), dtype=api.STAT_DTYPE)
    prm = _lib.LzParams(**{**api.DEFAULT_LZ, **(lz or {})})
    rp = C.POINTER(_lib.Region)(); nr = C.c_int64()
    _lib.check(lib.vg_lz_align_sharded(gs._h, tasks.ctypes.data_as(C.POINTER(_lib.Task)), len(tasks), C.byref(prm), comm.h,
                                       stats.ctypes.data_as(C.POINTER(_lib.PairStat)),
                                       C.byref(rp) if want_regions else None, C.byref(nr)))
    regions = api._take(rp, nr.value, api.REGION_DTYPE) if want_regions else None
    return stats, regions


def align_owner(tasks, n_genomes, world):
    from . import _lib, api
    lib = _lib.load()
    tasks = np.ascontiguousarray(tasks, dtype=api.TASK_DTYPE)
    owner = np.zeros(max(len(tasks), 1), dtype=np.int32)
    _lib.check(lib.vg_align_owner(tasks.ctypes.data_as(C.POINTER(_lib.Task)), len(tasks), int(n_genomes), int(world),
                                  owner.ctypes.data_as(C.POINTER(C.c_int32))))
    return owner[:len(tasks)]


# ------------------------------------------------------------------ whole stages (files)
def prefilter(paths, out_path, is_multifasta, k=25, min_kmers=20, min_ident=0.7, kmers_fraction=1.0, max_seqs=0,
              num_threads=1):
    from . import _lib, api
    lib = _lib.load()
    dist, device = init_process_group()
    rank, world, local_rank = dist_env()
    api.set_device(local_rank % max(api.device_count(), 1))
    comm = make_comm(dist, device)
    try:
        arr = (C.c_char_p * len(paths))(*[os.fsencode(str(p)) for p in paths])
        prm = _lib.PrefilterParams(k, min_kmers, min_ident, 0, kmers_fraction, max_seqs, num_threads, 0, int(bool(is_multifasta)))
        _lib.check(lib.vg_prefilter_sharded(arr, len(paths), os.fsencode(str(out_path)), C.byref(prm), comm.h))
    finally:
        comm.close()


def align(paths, out_path, is_multifasta, columns, filter_path=None, filter_threshold=0.0, out_aln=None, lz=None,
          out_filters=None, num_threads=1):
    from . import _lib, api
    lib = _lib.load()
    dist, device = init_process_group()
    rank, world, local_rank = dist_env()
    api.set_device(local_rank % max(api.device_count(), 1))
    comm = make_comm(dist, device)
    try:
        arr = (C.c_char_p * len(paths))(*[os.fsencode(str(p)) for p in paths])
        p = api.align_params(columns, filter_path, filter_threshold, out_aln, lz, out_filters, num_threads, 0, is_multifasta)
        _lib.check(lib.vg_align_sharded(arr, len(paths), os.fsencode(str(out_path)), C.byref(p), comm.h))
    finally:
        comm.close()
