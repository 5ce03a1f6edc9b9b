#!/usr/bin/env python3
"""Rate of the GPU tokenizer kernels on text resident in device memory (10 M six-decimal lines, 285 MB)."""
import importlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package

psa = load_package()
ht = importlib.import_module("particle_simulation_amd.hosttools")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "t.xyz")
    ht.generate_text(path, ht.UNIFORM, 7, n)
    data = open(path, "rb").read()
want = ht.generate(ht.UNIFORM, 7, n)
raw = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
out = torch.empty((n + 8, 3), dtype=torch.float32, device="cuda:0")
count = torch.zeros(1, dtype=torch.int64, device="cuda:0")
status = torch.zeros(1, dtype=torch.int32, device="cuda:0")
wsb = psa.lib().pc_parse_workspace_bytes(len(data))
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream


def run():
    psa._check(psa.lib().pc_parse_text_device(raw.data_ptr(), len(data), out.data_ptr(), n + 8, count.data_ptr(), status.data_ptr(),
                                              ws.data_ptr(), wsb, s))


run()
torch.cuda.synchronize()
assert int(count.item()) == n and int(status.item()) == 0
assert np.array_equal(out[:n].cpu().numpy().view(np.uint32), want.view(np.uint32))
psa.profile_enable(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("%d lines, %d bytes: %.3f ms per call = %.0f GB/s of text, %.2e points/s; floats equal to the host generator's" %
      (n, len(data), ms, len(data) / ms / 1e6, n / ms * 1e3))
for k, (timed, launches, tot) in sorted(psa.profile_read().items()):
    if timed and k.startswith("pc_t"):
        print("    %-24s %.4f ms" % (k, tot / timed))
