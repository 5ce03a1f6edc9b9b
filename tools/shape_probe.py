#!/usr/bin/env python3
"""Step time of the whole-path call on degenerate shapes: all points in a sheet two cells thick normal to x, y or z,
or spread over a few layers / pencils -- where one bin of a binning level takes nearly every point of a tile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
psa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
g = torch.Generator(device="cuda:0"); g.manual_seed(11)
st = psa.DeviceStages(n, "cuda:0")
for name, thin_axes, width in (("uniform", (), 0), ("sheet normal to x", (0,), 0.1), ("sheet normal to y", (1,), 0.1), ("sheet normal to z", (2,), 0.1),
                               ("slab of 8 layers", (0,), 0.4), ("blobs 0.3", None, 0.3), ("blobs 0.6", None, 0.6), ("blobs 1.0", None, 1.0)):
    if thin_axes is None:                # 64 Gaussian blobs of that width
        centres = torch.rand((64, 3), device="cuda:0", generator=g) * 16 - 8
        xyz = centres[torch.randint(0, 64, (n,), device="cuda:0", generator=g)] + torch.randn((n, 3), device="cuda:0", generator=g) * width
        xyz = xyz.clamp(-9.999, 9.999)
    else:
        xyz = torch.rand((n, 3), device="cuda:0", generator=g) * 19.9 - 9.95
        for a in thin_axes:
            xyz[:, a] = torch.rand(n, device="cuda:0", generator=g) * width + 0.0001
    xyz = xyz.contiguous()
    for variant in (4, 5, 3, 0):
        psa.set_variant("count", variant)
        st.count_pairs(xyz, 0, 400); torch.cuda.synchronize()
        psa.profile_enable(1)
        t = time.time(); st.count_pairs(xyz, 0, 400); torch.cuda.synchronize(); dt = time.time() - t
        psa.profile_enable(False)
        prof = psa.profile_read()
        top = sorted(((ms, k) for k, (timed, launches, ms) in prof.items() if timed), reverse=True)[:3]
        print("%-28s variant %d: %8.2f ms, pairs %d   %s" % (name, variant, dt * 1e3, st.pairs(), ", ".join("%s %.2f" % (k.replace("pc_", ""), ms) for ms, k in top)), flush=True)
    psa.set_variant("count", 4)
