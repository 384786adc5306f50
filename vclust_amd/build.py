"""Build recipe of libvclust_gpu.so (hand-written HIP for gfx950) and of the CPU oracle.

`python -m vclust_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles gfx950
without a GPU; the .so is kept in-tree (git-ignored) so that it travels with the gpurun
snapshot and is the file the tests load.
"""
import os
import pathlib
import shutil
import subprocess
import sys

PKG_DIR = pathlib.Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
CSRC = PKG_DIR / 'csrc'
LIB_PATH = PKG_DIR / 'libvclust_gpu.so'
ORACLE_DIR = ROOT / 'oracle'

SOURCES = ['vg_core.cpp', 'vg_genomes.cpp', 'vg_inflate.cpp', 'vg_io.cpp', 'vg_api.cpp', 'vg_synth.cpp', 'vg_prefilter.hip', 'vg_align.hip', 'vg_dist.hip']


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and pathlib.Path(cand).exists():
            return cand
    raise RuntimeError('hipcc not found (needs ROCm)')


def _stale(target: pathlib.Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(pathlib.Path(d).stat().st_mtime > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> pathlib.Path:
    """Compile every HIP/C++ source into vclust_amd/libvclust_gpu.so for gfx950."""
    srcs = [CSRC / s for s in SOURCES]
    deps = srcs + [CSRC / 'vg_common.h', ROOT / 'include' / 'vclust_gpu.h']
    if not force and not _stale(LIB_PATH, deps):
        return LIB_PATH
    obj_dir = PKG_DIR / '_obj'
    obj_dir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
             '-Wno-unused-result', f'-I{ROOT / "include"}']
    if os.environ.get('VG_LINES') == '1':        # developer build: line tables for rocprofv3's PC sampling
        flags.append('-gline-tables-only')
    if os.environ.get('VG_DEV') == '1':          # developer build: the instrumented parse kernel (tools/micro/lz_stats.py)
        flags.append('-DVG_DEV_KERNELS')
    objs = []
    procs = []
    for s in srcs:
        o = obj_dir / (s.name + '.o')
        objs.append(o)
        if force or _stale(o, [s, CSRC / 'vg_common.h', ROOT / 'include' / 'vclust_gpu.h']):
            cmd = [hipcc, *flags, '-x', 'hip', '-c', str(s), '-o', str(o)]
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s.name}:\n{out}')
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(LIB_PATH), *map(str, objs), '-lz', '-lpthread', '-ldl']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return LIB_PATH


def build_oracle(verbose: bool = False) -> pathlib.Path:
    """Compile the CPU oracle (test infrastructure) into oracle/_build/."""
    r = subprocess.run(['make', '-C', str(ORACLE_DIR)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'oracle build failed:\n{r.stdout}')
    if verbose:
        print(r.stdout, file=sys.stderr)
    return ORACLE_DIR / '_build' / 'liboracle.so'


if __name__ == '__main__':
    force = '--force' in sys.argv
    print(build_lib(force=force, verbose=True))
    print(build_oracle(verbose=False))
