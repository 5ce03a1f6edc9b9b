#!/usr/bin/env python3
"""What a GPU tokenizer would have to beat: the time to get the TEXT of a 100 M-point file into device memory,
against the host tokenizer's time on the same file.  Copies: straight from the page-cache mapping (pageable),
and through page-locked staging blocks filled by 16 host threads (the best a host program can do without
registering the mapping); plus the cost of page-locking the whole mapping.

    python3 tools/h2d_text_cost.py [points]
"""
import ctypes
import importlib
import mmap
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package

psa = load_package()
ht = importlib.import_module("particle_simulation_amd.hosttools")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
tmp = tempfile.mkdtemp(prefix="h2dtext_")
path = os.path.join(tmp, "u.xyz")
ht.generate_text(path, ht.UNIFORM, 7, n)
size = os.path.getsize(path)
out = {"points": n, "file_bytes": size}
hip = ctypes.CDLL("libamdhip64.so")
torch.zeros(1, device="cuda:0")
dev = torch.empty(size, dtype=torch.uint8, device="cuda:0")
with open(path, "rb") as f:
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    host = np.frombuffer(mm, dtype=np.uint8)
    host.sum()                                                # page cache warm, mapping faulted in
    for name in ("pageable_copy_first_s", "pageable_copy_again_s"):
        t = time.time()
        hip.hipMemcpy(ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(host.ctypes.data), ctypes.c_size_t(size), 1)
        torch.cuda.synchronize()
        out[name] = time.time() - t
    # page-locking the mapping itself
    t = time.time()
    rc = hip.hipHostRegister(ctypes.c_void_p(host.ctypes.data), ctypes.c_size_t(size), 0)
    out["host_register_rc"] = rc
    out["host_register_s"] = time.time() - t
    if rc == 0:
        t = time.time()
        hip.hipMemcpy(ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(host.ctypes.data), ctypes.c_size_t(size), 1)
        torch.cuda.synchronize()
        out["registered_copy_s"] = time.time() - t
        hip.hipHostUnregister(ctypes.c_void_p(host.ctypes.data))
    # staging through page-locked blocks, 16 threads (memcpy + async copy per 8 MB block)
    import threading
    nthreads, block = 16, 8 << 20
    pinned = [torch.empty(block, dtype=torch.uint8).pin_memory() for _ in range(nthreads)]
    streams = [torch.cuda.Stream() for _ in range(nthreads)]

    def worker(t):
        lo, hi = size * t // nthreads, size * (t + 1) // nthreads
        buf = pinned[t].numpy()
        with torch.cuda.stream(streams[t]):
            for off in range(lo, hi, block):
                m = min(block, hi - off)
                streams[t].synchronize()
                buf[:m] = host[off:off + m]
                dev[off:off + m].copy_(pinned[t][:m], non_blocking=True)
        streams[t].synchronize()
    t0 = time.time()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    [x.start() for x in th]
    [x.join() for x in th]
    out["staged_copy_16_threads_s"] = time.time() - t0
    del host
    mm.close()
# the host tokenizer on the same file, for comparison (whole host program, phases on stderr)
r = subprocess.run([ht.MAIN_PATH, "-f", path, "-p", "16"], capture_output=True, text=True, env=dict(os.environ, PAIRCOUNT_TIMER="1"))
import re
out["host_program_phases_s"] = {m.group(1): float(m.group(2)) / 1e6 for m in re.finditer(r"(\w[\w()]*):\s+(\d+) us", r.stderr)}
os.remove(path)
os.rmdir(tmp)
import json
print(json.dumps(out, indent=1))
