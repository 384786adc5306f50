#!/bin/bash
# Developer tool: the two cold CLI processes right behind a process that touched and released 240 GiB of device memory
# (the state in which round 3's footprint of 66 + 50 GB waited 4 s for the driver's wipe): allocation wait per process
set -u
cd tools/micro && hipcc --offload-arch=gfx950 -O2 -o /tmp/ft first_touch.hip && cd ../..
python - <<'PY'
import os, subprocess, sys, tempfile, time, re
sys.path.insert(0, '.')
from vclust_amd import synth
codes, offsets, names, _ = synth.make_workload('phage-100k', 10000)
td = tempfile.mkdtemp(dir=os.environ.get('TMPDIR', '/tmp')); fa = os.path.join(td, 'g.fna')
synth.write_fasta(fa, codes, offsets, names)
for tag, env in (('bounded (default)', {}), ('one pass / one batch (round 3 footprint)', dict(VG_WORKSPACE_GB='1000', VG_ONESHOT_INDEX_GB='64'))):
    subprocess.run(['/tmp/ft', 'malloc', '8', '30', '0'], stdout=subprocess.DEVNULL)      # 240 GiB touched, released, process gone
    tot = 0.0; waits = []
    for cmd in (['prefilter', '-i', fa, '-o', td + '/f.txt', '-v', '0'], ['align', '-i', fa, '-o', td + '/a.tsv', '--filter', td + '/f.txt', '-v', '0']):
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, 'vclust.py', *cmd], env=dict(os.environ, VG_HOST_TRACE='1', VG_ALLOC_TRACE='1', **env), stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0; tot += dt
        w = sum(float(x) for x in re.findall(r'alloc ([0-9.]+) ms', p.stderr))
        waits.append(round(w / 1e3, 3))
        print(f'  {cmd[0]}: {dt:.3f} s, allocation wait {w / 1e3:.3f} s', flush=True)
    print(f'{tag}: total {tot:.3f} s, allocation wait {waits}', flush=True)
PY
