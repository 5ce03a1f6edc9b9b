#!/usr/bin/env python3
"""Do the merged-window pair test (vector issue + LDS bound) and the thread-per-particle pair test (cache-gather
bound) overlap when they run on two streams over disjoint x layers?  Uniform 100 M points."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
psa = load_package()
ht = importlib.import_module("particle_simulation_amd.hosttools")
n = 100_000_000
xyz = psa.generate_device(ht.UNIFORM, 7, n)
st = psa.DeviceStages(n, "cuda:0")
st.bin(xyz); torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
res_b = torch.zeros(2, dtype=torch.int64, device="cuda:0")
L = psa.lib()

def run(split):
    st.result.zero_(); res_b.zero_(); torch.cuda.synchronize()
    t = time.time()
    if split > 0:
        psa.set_variant("count", 5)
        psa._check(L.pc_count_sorted_device(st.sorted.data_ptr(), st.table.data_ptr(), n, 0, split, st.result.data_ptr(), sa.cuda_stream))
    if split < 400:
        psa.set_variant("count", 0)
        psa._check(L.pc_count_sorted_device(st.sorted.data_ptr(), st.table.data_ptr(), n, split, 400, res_b.data_ptr(), sb.cuda_stream))
    torch.cuda.synchronize()
    dt = time.time() - t
    return dt * 1e3, int(st.result[0].item()) + int(res_b[0].item())

for split in (400, 0, 360, 340, 320, 300, 280):
    run(split)
    best = min(run(split) for _ in range(5))
    print("layers [0,%d) merged-window kernel, [%d,400) thread-per-particle kernel: %.3f ms, pairs %d" % (split, split, best[0], best[1]), flush=True)
psa.set_variant("count", 4)
