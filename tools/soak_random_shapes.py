#!/usr/bin/env python3
"""Soak run of the random-shape parity test (tests/test_gpu_random_shapes.py) over many more seeds and over the
pair-test variants and knobs the default never picks on such inputs: every seed's pairs, reference test count
and binning table against the CPU oracle, then the pair count again under each other variant.

    python3 tools/soak_random_shapes.py [first_seed] [seeds]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from __graft_entry__ import load_package
import oracle_lib
from test_gpu_random_shapes import _shape, NBB

psa = load_package()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
oracle = oracle_lib.Oracle()
stages = psa.DeviceStages(600_000, "cuda:0")
kinds = ["box", "blobs", "sheet", "rod", "duplicates", "faces"]
# (count variant, knobs) beside the default
others = [(0, {}), (3, {}), (5, {}), (5, {"dense": 0}), (5, {"dense": 2}), (5, {"dense_min": 1}), (7, {}),
          (4, {"clustered_share": 0}), (4, {"clustered_share": 100})]
defaults = {"dense": 1, "dense_min": 11, "clustered_share": 12}
t0 = time.time()
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    parts = [_shape(rng, kinds[int(rng.integers(0, len(kinds)))], int(rng.integers(1, 150_000)))
             for _ in range(int(rng.integers(1, 4)))]
    xyz = np.concatenate(parts)
    rng.shuffle(xyz)
    n = len(xyz)
    want, want_tests, srt, start = oracle.count_pairs(xyz, want_sorted=True)
    t = torch.from_numpy(xyz).to("cuda:0")
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)
    for k, v in defaults.items():
        psa.set_variant(k, v)
    ok = stages.count_pairs(t, 0, 400) == n
    stages.count_tests(0, 400)
    ok = ok and (stages.pairs(), stages.tests()) == (want, want_tests)
    ok = ok and np.array_equal(stages.table[:NBB ** 3 + 1].cpu().numpy().view(np.uint32), start)
    got = {}
    for variant, knobs in others:
        psa.set_variant("count", variant)
        for k, v in defaults.items():
            psa.set_variant(k, knobs.get(k, v))
        stages.count_pairs(t, 0, 400)
        got[(variant, tuple(sorted(knobs.items())))] = stages.pairs()
    wrong = {k: v for k, v in got.items() if v != want}
    # the other binning variant, default pair test
    psa.set_variant("bin", 0)
    psa.set_variant("count", 4)
    for k, v in defaults.items():
        psa.set_variant(k, v)
    stages.count_pairs(t, 0, 400)
    if stages.pairs() != want:
        wrong["bin 0"] = stages.pairs()
    if not ok or wrong:
        bad += 1
        print("seed %d n %d: default ok %s, wrong %s (want %d)" % (seed, n, ok, wrong, want), flush=True)
    if (seed - first) % 20 == 19:
        print("%d seeds, %d bad, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
psa.set_variant("bin", 1)
print("done: %d seeds, %d bad, %.0f s" % (count, bad, time.time() - t0))
sys.exit(1 if bad else 0)
