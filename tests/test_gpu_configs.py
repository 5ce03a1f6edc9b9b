"""GPU: the task's large configurations at full size.  No oracle run is possible at
these sizes, so each count is pinned through properties the path offers: equality
of independent kernel variants, slab additivity (a slab counted whole equals its
sub-slabs counted separately, each with its own +x halo layer), the reference's
test-count formula evaluated from the table, and the analytic expectation."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NBB = 403


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(autouse=True)
def default_variants(psa):
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)
    yield
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)


def test_config4_clustered_100m(psa, torch_mod):
    """64 Gaussian blobs, 100 M points (the stream bench.py --workload blobs100m uses): the count passes
    2^32 (the 64-bit accumulation path), the default pair test equals the thread-per-particle one, and
    four cost-balanced x slabs add up to the whole."""
    torch = torch_mod
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    sh = importlib.import_module("particle_simulation_amd.sharding")
    n = 100_000_000
    xyz = psa.generate_device(ht.BLOBS, 7, n, 500_000)
    st = psa.DeviceStages(n, "cuda:0")
    psa.set_variant("count", 3)
    st.count_pairs(xyz)
    st.count_tests()
    whole, tests = st.pairs(), st.tests()
    assert whole > 2 ** 32 and tests > whole                      # 7.68e9 pairs, 4.9e10 tests
    assert int(st.table[NBB ** 3].item()) == n
    psa.set_variant("count", 0)
    st.count_sorted(n)
    assert st.pairs() == whole
    psa.set_variant("count", 4)
    del xyz
    # slabs cut from the sorted array where the squared layer populations balance
    layer_start = st.table[:: NBB * NBB][:401].cpu().numpy().view(np.uint32).astype(np.int64)
    pop = np.diff(layer_start).astype(np.float64)
    cuts = sh.balanced_cuts(pop * pop, 4)
    srt = st.sorted[:n].clone()
    total = total_tests = 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        a, b = int(layer_start[lo]), int(layer_start[min(hi + 1, 400)])
        st.count_pairs(srt[a:b].contiguous(), lo, hi)
        st.count_tests(lo, hi)
        total += st.pairs()
        total_tests += st.tests()
    assert total == whole and total_tests == tests


def test_config5_rank_slab_of_one_billion(psa, torch_mod):
    """The slab rank 3 of 8 holds in the 1 B-point configuration (x layers [150,200) plus the halo layer,
    127.5 M points of the uniform stream, 15.6 per cell), made on the device: expectation, variant equality,
    and additivity over two half slabs that each bring their own halo."""
    torch = torch_mod
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    lo, hi = 150, 200
    xyz = psa.generate_device(ht.UNIFORM, 7, 1_000_000_000, 0, x_lo=lo, x_hi=hi)
    n = xyz.shape[0]
    assert 127_000_000 < n < 128_000_000
    st = psa.DeviceStages(n, "cuda:0")
    st.count_pairs(xyz, lo, hi)
    st.count_tests(lo, hi)
    whole, tests = st.pairs(), st.tests()
    del xyz
    # 1/8 of the 3.27e10 pairs and 2.1e11 tests expected for the whole domain
    assert 4.0e9 < whole < 4.2e9 and 2.55e10 < tests < 2.7e10
    psa.set_variant("count", 0)
    st.count_sorted(n, lo, hi)
    assert st.pairs() == whole
    psa.set_variant("count", 4)
    table = st.table.clone()
    srt = st.sorted[:n].clone()
    mid = 175
    total = total_tests = 0
    for a_l, b_l in ((lo, mid), (mid, hi)):
        a = int(table[a_l * NBB * NBB].item()) & 0xFFFFFFFF
        # (a slab call writes the table for its own layers and the halo layer only: the end of the halo layer
        # of the upper half is the end of the array)
        b = n if b_l == hi else int(table[(b_l + 1) * NBB * NBB].item()) & 0xFFFFFFFF
        st.count_pairs(srt[a:b].contiguous(), a_l, b_l)
        st.count_tests(a_l, b_l)
        total += st.pairs()
        total_tests += st.tests()
    assert total == whole and total_tests == tests


def test_more_than_four_gigabytes_of_points(psa, torch_mod):
    """3.7e8 uniform points (4.4 GB of coordinates, 5.8 per cell): the sorted array no longer fits 32-bit byte
    offsets, which the merged-window kernel's gather lists once assumed.  The default pair test, the merged-window
    kernel asked for by name, the thread-per-particle kernel and the sum of two slabs must agree, and the count
    must lie where a uniform density puts it."""
    torch = torch_mod
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    n = 370_000_000
    assert n * 12 > 2 ** 32
    xyz = psa.generate_device(ht.UNIFORM, 5, n, 0)
    st = psa.DeviceStages(n, "cuda:0")
    st.count_pairs(xyz)
    st.count_tests()
    whole, tests = st.pairs(), st.tests()
    expect = n * (n - 1) / 2 * (4.0 / 3.0 * np.pi * 0.05 ** 3) / 8000.0
    assert abs(whole - expect) < 0.01 * expect                      # 4.5e9 pairs: also beyond 2^32
    assert int(st.table[NBB ** 3].item()) & 0xFFFFFFFF == n
    for variant in (5, 0):
        psa.set_variant("count", variant)
        st.count_sorted(n)
        assert st.pairs() == whole, variant
    psa.set_variant("count", 4)
    del xyz
    table = st.table.clone()
    srt = st.sorted[:n].clone()
    total = total_tests = 0
    for a_l, b_l in ((0, 200), (200, 400)):
        a = int(table[a_l * NBB * NBB].item()) & 0xFFFFFFFF
        b = n if b_l == 400 else int(table[(b_l + 1) * NBB * NBB].item()) & 0xFFFFFFFF
        st.count_pairs(srt[a:b].contiguous(), a_l, b_l)
        st.count_tests(a_l, b_l)
        total += st.pairs()
        total_tests += st.tests()
    assert total == whole and total_tests == tests
