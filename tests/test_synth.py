"""The synthetic workloads of SURVEY.md 8(d): splitmix64 draws, pinned digests, and the C++ generator of
libvclust_gpu.so (vg_synth_plan) agreeing with the numpy definition bit for bit."""
import json

import numpy as np

from vclust_amd import synth


def test_splitmix64_known_answers():
    # reference sequence of splitmix64 seeded with 1234567: state += GOLDEN; output = mix(state)
    out = synth.draws(np.uint64(1234567), 3)
    assert [int(x) for x in out] == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_digests_are_pinned(golden_dir):
    pins = json.loads((golden_dir.parent / 'synth_sha256.json').read_text())
    for key, want in pins.items():
        name, _, n = key.partition('/')
        codes, offsets, names, desc = synth.make_workload(name, int(n) if n else None)
        assert synth.sha256(codes, offsets) == want['sha256'], key
        assert len(names) == want['genomes'] and int(offsets[-1]) == want['bases']


def test_native_generator_equals_numpy_definition():
    a = synth.make_families(7, 4, length=9000, seed=5, native=True)
    b = synth.make_families(7, 4, length=9000, seed=5, native=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    c = synth.make_contigs(120, len_lo=2000, len_hi=20000, seed=9, native=True)
    d = synth.make_contigs(120, len_lo=2000, len_hi=20000, seed=9, native=False)
    assert np.array_equal(c[0], d[0]) and np.array_equal(c[1], d[1]) and np.array_equal(c[3], d[3])


def test_slices_of_one_set_are_independent():
    whole = synth.make_families(6, 3, length=5000, seed=3)
    part = synth.make_families(2, 3, length=5000, seed=3, first_family=4)
    assert np.array_equal(whole[0][whole[1][12]:], part[0]) and whole[2][12:] == part[2]


def test_mutation_model_shape():
    codes, offsets, names = synth.make_families(3, 10, length=40000, seed=1)
    lens = np.diff(offsets)
    assert len(names) == 30 and np.all(np.abs(lens - 40000) <= 5 * 50)
    assert set(np.unique(codes)) <= {0, 1, 2, 3}
