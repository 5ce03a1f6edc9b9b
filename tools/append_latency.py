#!/usr/bin/env python3
"""Where the first pc_append's time goes: creation, first and later copies, from one thread and from fresh threads."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

psa = load_package()


def ms(t):
    return "%.2f ms" % ((time.time() - t) * 1e3)


t = time.time()
pc = psa.PairCounter(device=0, expected_points=120_000_000)
print("pc_create", ms(t), flush=True)
blk = np.random.default_rng(1).uniform(-9, 9, (65536, 3)).astype(np.float32)
for i in range(4):
    t = time.time()
    pc.append(blk)
    print("main thread append %d" % i, ms(t), flush=True)


def worker(k):
    out = []
    for i in range(4):
        t = time.time()
        pc.append(blk)
        out.append(ms(t))
    print("thread %d appends:" % k, out, flush=True)


for k in range(2):
    th = threading.Thread(target=worker, args=(k,))
    th.start()
    th.join()
ths = [threading.Thread(target=worker, args=(10 + k,)) for k in range(8)]
t = time.time()
[x.start() for x in ths]
[x.join() for x in ths]
print("8 concurrent threads x 4 appends", ms(t), flush=True)
with psa.HostBlock(65536) as hb:
    hb.array[:] = blk
    for i in range(3):
        t = time.time()
        pc.append(hb.array)
        print("page-locked append %d" % i, ms(t), flush=True)
t = time.time()
print("pairs", pc.finish(), ms(t))
