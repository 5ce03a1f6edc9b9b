#!/usr/bin/env python3
"""Do binning (HBM-bound) and the merged-window pair test (vector issue + LDS bound) overlap when they run on two
streams over two independent inputs?  Uniform 100 M points each.  Prints the two alone, both back to back on one
stream, and both on two streams (started in either order)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package

psa = load_package()
ht = importlib.import_module("particle_simulation_amd.hosttools")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
xyz = psa.generate_device(ht.UNIFORM, 7, n)
st1 = psa.DeviceStages(n, "cuda:0")
st2 = psa.DeviceStages(n, "cuda:0")
st1.bin(xyz)
st2.bin(xyz)
st2.count_sorted(n)
torch.cuda.synchronize()
want = st2.pairs()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def on_two_streams(first_bin):
    def fn():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        if first_bin:
            with torch.cuda.stream(sa):
                st1.bin(xyz)
            with torch.cuda.stream(sb):
                st2.count_sorted(n)
        else:
            with torch.cuda.stream(sb):
                st2.count_sorted(n)
            with torch.cuda.stream(sa):
                st1.bin(xyz)
        cur.wait_stream(sa)
        cur.wait_stream(sb)
    return fn


def back_to_back():
    st1.bin(xyz)
    st2.count_sorted(n)


print("binning alone            %.3f ms" % timed(lambda: st1.bin(xyz)), flush=True)
print("pair test alone          %.3f ms" % timed(lambda: st2.count_sorted(n)), flush=True)
print("one stream, back to back %.3f ms" % timed(back_to_back), flush=True)
print("two streams, binning launched first   %.3f ms" % timed(on_two_streams(True)), flush=True)
assert st2.pairs() == want
print("two streams, pair test launched first %.3f ms" % timed(on_two_streams(False)), flush=True)
assert st2.pairs() == want
