"""Developer tool: cycle split of one parse task (VG_LZ_ABLATE=128|sel<<8|32)."""
import os, sys, pathlib, subprocess
root = pathlib.Path(__file__).resolve().parent.parent
names = ['probe', 'event loads (masks+gap)', 'bwd extend + region', 'fwd extend']
for sel, nm in enumerate(names):
    env = dict(os.environ, VG_LZ_ABLATE=str(128 | 32 | (sel << 8)))
    out = subprocess.run([sys.executable, str(root / 'tools' / 'one_task.py')], env=env, capture_output=True, text=True).stdout.splitlines()[0]
    f = out.replace('=', ' ').split()
    m = int(out.split(' M ')[1].split()[0]); ev = int(out.split(' A ')[1].split()[0]); us = float(out.split('= ')[1].split()[0])
    print(f'{nm:28s} {m * 16 / 2400:8.1f} us (at 2.4 GHz)   events {ev}   task {us} us')
