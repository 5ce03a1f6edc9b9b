"""CPU, world_size 2 over gloo: the multi-rank path's host logic -- slab cuts,
routing of points to ranks with their +x halo layer, and the 8-byte all-reduce.
With no GPU here, each rank's slab is counted by the oracle (the checker stands
in for the kernel); the GPU tests cover the kernels' slab arguments."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind_name, q):
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib
    from conftest import load_package
    import oracle_lib
    load_package()
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    sh = importlib.import_module("particle_simulation_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.Oracle()
        kind, sigma, n = (ht.UNIFORM, 0, 400_000) if kind_name == "uniform" else (ht.BLOBS, 300_000, 200_000)
        whole = ht.generate(kind, 11, n, sigma, threads=2)
        if kind_name == "uniform":
            cuts = sh.equal_cuts(world)
        else:
            layer_pop = np.bincount(orc.cell_keys(whole) // (403 * 403), minlength=400)[:400]
            cuts = sh.balanced_cuts(layer_pop.astype(np.float64) ** 2, world)
        lo, hi = cuts[rank], cuts[rank + 1]
        x0, x1 = sh.halo_range(lo, hi)
        mine = ht.generate(kind, 11, n, sigma, x_lo=x0, x_hi=x1, threads=2)      # what this rank would hold
        assert np.array_equal(mine.view(np.uint32), ht.filter_slab(whole, lo, hi).view(np.uint32))
        local, _ = orc.count_pairs(mine, threads=2, x_lo=lo, x_hi=hi)
        total = sh.reduce_count(local)
        want, _ = orc.count_pairs(whole, threads=2)
        q.put((rank, cuts, local, total, want))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind_name", ["uniform", "blobs"])
def test_two_ranks_sum_to_the_single_process_count(kind_name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind_name, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, cuts, local, total, want in res:
        assert total == want, (rank, cuts, local, total, want)
    assert sum(r[2] for r in res) == res[0][4]
    assert 0 < res[0][2] < res[0][4]            # both ranks contributed


def test_cut_helpers(psa):
    import importlib
    sh = importlib.import_module("particle_simulation_amd.sharding")
    assert sh.equal_cuts(8) == [0, 50, 100, 150, 200, 250, 300, 350, 400]
    assert sh.equal_cuts(3)[0] == 0 and sh.equal_cuts(3)[-1] == 400
    cost = np.zeros(400)
    cost[100:110] = 1.0                          # all work in ten layers
    cuts = sh.balanced_cuts(cost, 4)
    assert cuts[0] == 0 and cuts[-1] == 400 and all(b > a for a, b in zip(cuts, cuts[1:]))
    sums = [cost[a:b].sum() for a, b in zip(cuts, cuts[1:])]
    assert max(sums) <= 3.0
    assert sh.halo_range(350, 400) == (350, 399) and sh.halo_range(0, 50) == (0, 50)
    assert sh.reduce_count(41) == 41             # no process group: identity
