#!/usr/bin/env python3
"""Developer check of the library's own gunzip (vclust_amd/csrc/vg_inflate.cpp) against zlib, outside the library:
   python tools/micro/gunzip_check.py [cases] [seed]
builds tools/micro/gunzip_harness.cpp with AddressSanitizer + UBSan and runs (a) valid streams of many shapes -- skewed
symbol sets (15-bit codes, sub-tables), DNA text, periodic data, random bytes, zeros; levels 0-9, windows 2^9..2^15,
every zlib strategy, several members -- which must be decoded byte for byte, and (b) truncated, corrupted and padded
files, which must be refused (the library then hands the file to zlib) and never crash or be accepted with other bytes."""
import gzip, os, pathlib, struct, subprocess, sys, tempfile, zlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
td = pathlib.Path(tempfile.mkdtemp())
exe = td / 'gunzip_asan'
subprocess.run(['g++', '-O1', '-g', '-fsanitize=address,undefined', '-std=c++17', '-pthread', '-o', str(exe),
                str(ROOT / 'tools/micro/gunzip_harness.cpp'), str(ROOT / 'vclust_amd/csrc/vg_inflate.cpp'), '-lz'], check=True)
def member(data, level=6, wbits=-15, strategy=0, memlevel=8):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return (struct.pack('<BBBBIBB', 0x1f, 0x8b, 8, 0, 0, 0, 3) + co.compress(data) + co.flush()
            + struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff))
def skewed(n, k):
    p = 0.5 ** np.arange(1, k + 1); p /= p.sum()
    return rng.choice(np.arange(k, dtype=np.uint8), size=n, p=p).tobytes()
def dna(n):
    return rng.choice(np.frombuffer(b'ACGT\n', dtype=np.uint8), size=n, p=[.245, .245, .245, .245, .02]).tobytes()
def periodic(n):
    unit = os.urandom(int(rng.integers(1, 40)))
    return (unit * (n // len(unit) + 1))[:n]
def run(blob):
    (td / 'c.gz').write_bytes(blob)
    r = subprocess.run([str(exe), str(td / 'c.gz'), str(td / 'c.out')], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return r.returncode, ((td / 'c.out').read_bytes() if r.returncode == 0 else None), r.stderr
bad = 0
for i in range(n_cases):
    kind = int(rng.integers(0, 6)); n = int(rng.integers(0, 400000))
    d = [lambda: skewed(n, int(rng.integers(2, 200))), lambda: dna(n), lambda: periodic(n), lambda: os.urandom(n),
         lambda: skewed(n // 2, 30) + dna(n // 2), lambda: bytes(n)][kind]()
    gz = member(d, int(rng.integers(0, 10)), -int(rng.integers(9, 16)),
                int(rng.choice([0, 0, 0, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])), int(rng.integers(1, 10)))
    if rng.random() < 0.2:
        gz += member(dna(int(rng.integers(0, 5000))), int(rng.integers(0, 10)))
    rc, got, err = run(gz)
    if rc != 0 or got != gzip.decompress(gz):
        bad += 1; print('VALID STREAM NOT DECODED: case', i, 'rc', rc, err[-300:], flush=True)
base = [gzip.compress(dna(120000), 6), gzip.compress(os.urandom(30000), 0), gzip.compress(dna(50000), 1) + gzip.compress(dna(100), 6)]
refused = 0
for i in range(n_cases):
    b = bytearray(base[i % len(base)]); mode = i % 3
    if mode == 0: b = b[:int(rng.integers(0, len(b)))]
    elif mode == 1:
        for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    else: b += os.urandom(int(rng.integers(1, 40)))
    rc, got, err = run(bytes(b))
    if rc == 2: refused += 1; continue
    try: want = gzip.decompress(bytes(b))
    except Exception: want = None
    if rc != 0 or want is None or got != want:
        bad += 1; print('DAMAGED FILE MISHANDLED: case', i, 'mode', mode, 'rc', rc, err[-300:], flush=True)
print(f'{n_cases} valid streams and {n_cases} damaged files ({refused} refused): {bad} failures')
sys.exit(1 if bad else 0)
