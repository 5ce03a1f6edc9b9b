"""GPU: the HIP path, called through the C ABI, against the CPU oracle and the
golden fixtures.  Integer results: the bar is exact equality."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NBB = 403


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def stages(psa, torch_mod):
    return psa.DeviceStages(2_100_000, "cuda:0")


@pytest.fixture(autouse=True)
def default_variants(psa):
    """Every test starts and ends on the library's default kernels."""
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)
    psa.set_variant("dense", 1)
    psa.set_variant("even", 1)
    yield
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)
    psa.set_variant("dense", 1)
    psa.set_variant("even", 1)


def dev(torch_mod, a):
    return torch_mod.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda:0")


def _cell_box(rng, n, lo_cell, hi_cell):
    """n points uniform inside the box of cells [lo_cell, hi_cell) (per axis), kept clear of the box faces."""
    lo = np.asarray(lo_cell, np.float64) * 0.05 - 10.0 + 1e-4
    hi = np.asarray(hi_cell, np.float64) * 0.05 - 10.0 - 1e-4
    return rng.uniform(lo, hi, (n, 3))


VARIANTS = [(1, 4), (1, 5), (1, 3), (0, 0), (1, 1), (0, 2), (1, 7)]      # (binning variant, pair-test variant; 4 = chosen by density, 7 = level Z fused into the pair test)


def gpu_count(psa, stages, torch_mod, xyz, x_lo=0, x_hi=400, bin_v=1, count_v=4):
    psa.set_variant("bin", bin_v)
    psa.set_variant("count", count_v)
    t = dev(torch_mod, xyz).reshape(-1, 3)
    n = stages.count_pairs(t, x_lo, x_hi)
    stages.count_tests(x_lo, x_hi)
    return stages.pairs(), stages.tests(), n


def test_cell_keys_match_reference_formula(psa, stages, torch_mod, oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "cell_index.npz"))
    xs, want = z["x"], z["index"]
    rng = np.random.default_rng(1)
    pts = np.stack([xs, rng.permutation(xs), rng.permutation(xs)], axis=1).astype(np.float32)
    keys = stages.cell_keys(dev(torch_mod, pts)).cpu().numpy().astype(np.int64)
    assert np.array_equal(keys, oracle.cell_keys(pts))
    assert np.array_equal(keys // (NBB * NBB), want)          # golden: reference main.c:134-136
    big = rng.uniform(-9.9999, 9.9999, (1_000_000, 3)).astype(np.float32)
    keys = stages.cell_keys(dev(torch_mod, big)).cpu().numpy().astype(np.int64)
    assert np.array_equal(keys, oracle.cell_keys(big))
    assert int(stages.status.item()) == 0


def test_cell_index_shortcut_is_exact_on_every_float(psa, torch_mod):
    """The kernels' binary32 shortcut for the cell index against the reference's
    binary64 expression (main.c:134-136), over all 2^32 float bit patterns."""
    import ctypes
    bad = torch_mod.zeros(1, dtype=torch_mod.int64, device="cuda:0")
    rc = psa.lib().pc_selfcheck_cell_index_device(ctypes.c_void_p(bad.data_ptr()),
                                                  ctypes.c_void_p(torch_mod.cuda.current_stream().cuda_stream))
    assert rc == 0, psa.lib().pc_last_error()
    torch_mod.cuda.synchronize()
    assert int(bad.item()) == 0


def test_predicate_on_golden_pairs(psa, stages, torch_mod, oracle, golden_dir):
    """Near-threshold pairs from the reference's dist2()<=0.0025 (main.c:85-91,165):
    all 8192 points at once against the oracle, then the exact-threshold cases one
    pair per launch against the reference's own verdict."""
    z = np.load(os.path.join(golden_dir, "dist2_pairs.npz"))
    a, b, within = z["a"][:, :3], z["b"][:, :3], z["within"]
    both = np.concatenate([a, b])
    want_total = oracle.count_pairs(both)[0]
    for bv, cv in VARIANTS:
        total, _, _ = gpu_count(psa, stages, torch_mod, both, bin_v=bv, count_v=cv)
        assert total == want_total, (bv, cv)
    # every vector by itself: one launch per pair, restricted to the x layer of the lower point and its halo
    # layer (all a pair can occupy), so that a launch costs microseconds
    ca = np.floor((a.astype(np.float64) + 10.0) / 0.05).astype(np.int64)
    cb = np.floor((b.astype(np.float64) + 10.0) / 0.05).astype(np.int64)
    adjacent = np.all(np.abs(ca - cb) <= 1, axis=1)
    assert adjacent.sum() > 4000
    psa.set_variant("bin", 1)
    for cv in (5, 3, 0):
        psa.set_variant("count", cv)
        for i in np.nonzero(adjacent)[0]:
            lo = int(min(ca[i, 0], cb[i, 0]))
            t = dev(torch_mod, np.stack([a[i], b[i]]))
            stages.count_pairs(t, lo, lo + 1)
            assert stages.pairs() == int(within[i]), (i, cv)


def test_golden_fixture_counts(psa, stages, torch_mod, golden_dir):
    with open(os.path.join(golden_dir, "counts.json")) as f:
        counts = json.load(f)
    for name, info in counts.items():
        xyz = np.load(os.path.join(golden_dir, name.replace(".xyz", ".f32.npy")))
        for bv, cv in VARIANTS:
            got, tests, n = gpu_count(psa, stages, torch_mod, xyz, bin_v=bv, count_v=cv)
            assert n == info["points"]
            assert got == info["count"], (name, bv, cv)


@pytest.mark.parametrize("kind,n,seed", [("uniform", 200_000, 1), ("uniform", 2_000_000, 2),
                                         ("box", 300_000, 3), ("blobs", 400_000, 4), ("one_cell", 3000, 5),
                                         ("huge_pencils", 120_000, 6)])
def test_seeded_inputs_match_oracle(psa, stages, torch_mod, oracle, kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        xyz = rng.uniform(-9.999, 9.999, (n, 3))
    elif kind == "box":                      # dense enough that most cells hold several points
        xyz = rng.uniform(-1.0, 1.0, (n, 3))
    elif kind == "blobs":
        c = rng.uniform(-8, 8, (16, 3))
        xyz = np.clip(c[rng.integers(0, 16, n)] + rng.normal(0, 0.12, (n, 3)), -9.999, 9.999)
    elif kind == "huge_pencils":             # pencils beyond one workgroup's sorting capacity (8192 points), side by side
        xyz = np.concatenate([_cell_box(rng, 40_000, (77, 200, 0), (78, 201, 400)),
                              _cell_box(rng, 9_000, (77, 201, 0), (78, 202, 400)),
                              _cell_box(rng, 8_193, (78, 200, 0), (79, 201, 400)),
                              _cell_box(rng, 20_000, (399, 399, 100), (400, 400, 103)),      # ... and one with three full cells
                              rng.uniform(-9.999, 9.999, (n - 77_193, 3))])
    else:                                    # everything inside one cell: n(n-1)/2 tests
        xyz = rng.uniform(0.001, 0.049, (n, 3))
    xyz = xyz.astype(np.float32)
    want, want_tests, srt, start = oracle.count_pairs(xyz, want_sorted=True)
    for bv, cv in VARIANTS:
        got, tests, _ = gpu_count(psa, stages, torch_mod, xyz, bin_v=bv, count_v=cv)
        assert got == want, (kind, bv, cv)
        assert tests == want_tests
        # binning: same table, same multiset of points per cell
        table = stages.table[:NBB ** 3 + 1].cpu().numpy().view(np.uint32)
        assert np.array_equal(table, start)
        g = stages.sorted[:n].cpu().numpy()
        keys_sorted = oracle.cell_keys(g)
        assert np.all(np.diff(keys_sorted) >= 0)
        a = np.sort(g.view(np.uint32).astype(np.uint64) @ np.array([1, 1 << 21, 1 << 42], dtype=np.uint64))
        b = np.sort(srt.view(np.uint32).astype(np.uint64) @ np.array([1, 1 << 21, 1 << 42], dtype=np.uint64))
        assert np.array_equal(a, b)


def test_edge_cases(psa, stages, torch_mod, oracle):
    assert gpu_count(psa, stages, torch_mod, np.zeros((0, 3), np.float32))[0] == 0
    assert gpu_count(psa, stages, torch_mod, np.array([[1.0, 2.0, 3.0]], np.float32))[0] == 0
    two = np.array([[0.1, 0.1, 0.1], [0.1, 0.1, 0.1]], np.float32)
    assert gpu_count(psa, stages, torch_mod, two)[0] == 1          # coincident points count (d2 = 0)
    # first/last occupied cells on every axis
    lo, hi = np.float32(-9.999999), np.float32(9.999999)
    pts = np.array([[lo, lo, lo], [lo, lo, lo + 0.01], [hi, hi, hi], [hi - 0.01, hi, hi], [hi, lo, hi]], np.float32)
    assert gpu_count(psa, stages, torch_mod, pts)[0] == oracle.count_pairs(pts)[0] == 2
    # out-of-domain input is an error, not a wrong answer
    with pytest.raises(psa.PairCountError):
        gpu_count(psa, stages, torch_mod, np.array([[10.5, 0.0, 0.0], [0, 0, 0]], np.float32))
    # ... and so are NaN and infinities on any axis, in both binning variants
    for bad in (np.nan, np.inf, -np.inf, 3.0e38):
        for axis in range(3):
            pts = np.zeros((3, 3), np.float32)
            pts[1, axis] = bad
            for bv in (0, 1):
                with pytest.raises(psa.PairCountError):
                    gpu_count(psa, stages, torch_mod, pts, bin_v=bv, count_v=4)
    # the bad point LAST: the binning drops it, and the slot it would have taken at the end of its layer or
    # pencil keeps what an earlier level left there (possibly the bad point itself); the pair test must
    # neither follow such a slot into the table nor crash -- the call reports the domain error
    for bad in (np.nan, np.inf, -np.inf, 3.0e38, -1.0e30):
        for axis in (1, 2, 0):
            for npts in (3, 64, 700):
                pts = np.random.default_rng(npts).uniform(-9.9, 9.9, (npts, 3)).astype(np.float32)
                pts[-1, axis] = bad
                for cv in (0, 4, 3):
                    with pytest.raises(psa.PairCountError):
                        gpu_count(psa, stages, torch_mod, pts, bin_v=1, count_v=cv)

    # the same after a LARGER valid input: the slot the dropped point would have taken now holds a valid point of that
    # earlier call (some other cell), which no kernel may take for a home particle with ranges of its own
    big = np.random.default_rng(5).uniform(-9.9, 9.9, (200_000, 3)).astype(np.float32)
    for cv in (0, 4, 3, 5):
        for axis in (1, 2, 0):
            for bad in (10.025, -10.075):
                assert gpu_count(psa, stages, torch_mod, big, bin_v=1, count_v=cv)[0] == oracle.count_pairs(big)[0]
                pts = big[:10].copy()
                pts[7, axis] = bad
                with pytest.raises(psa.PairCountError):
                    gpu_count(psa, stages, torch_mod, pts, bin_v=1, count_v=cv)
    assert gpu_count(psa, stages, torch_mod, big[:1000], bin_v=1, count_v=4)[0] == oracle.count_pairs(big[:1000])[0]


def test_finish_reports_a_failure_again(psa, torch_mod, oracle):
    """pc_finish after a failed pc_finish returns the same error, not a count of 0; so do pc_tests and
    pc_phase_ms.  A good context next to it is unaffected."""
    pts = np.array([[0.1, 0.2, 0.3], [0.1, 0.2, 0.3], [0.5, 11.0, 0.5]], np.float32)
    pc = psa.PairCounter(device=0, expected_points=16)
    pc.append(pts)
    for _ in range(3):
        with pytest.raises(psa.PairCountError) as e:
            pc.finish()
        assert e.value.code == -3
    with pytest.raises(psa.PairCountError) as e:
        pc.tests()
    assert e.value.code == -3
    with pytest.raises(psa.PairCountError):
        pc.phase_ms()
    with pytest.raises(psa.PairCountError) as e:
        pc.append(pts[:1])
    assert e.value.code == -5                      # closed for appends all the same
    pc.close()
    good = psa.PairCounter(device=0, expected_points=16)
    good.append(pts[:2])
    assert good.finish() == 1 and good.finish() == 1 and good.tests() == 1
    good.close()


def test_slab_shards_sum_to_whole(psa, stages, torch_mod, oracle):
    import oracle_lib
    rng = np.random.default_rng(21)
    xyz = rng.uniform(-2.0, 2.0, (500_000, 3)).astype(np.float32)
    whole, whole_tests, _ = gpu_count(psa, stages, torch_mod, xyz)
    assert whole == oracle.count_pairs(xyz)[0]
    keys = oracle.cell_keys(xyz)
    for cuts in ([0, 200, 400], [0, 170, 185, 200, 215, 230, 400]):
        total = tt = 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sub = oracle_lib.slab_points(xyz, keys, lo, hi)
            c, t, _ = gpu_count(psa, stages, torch_mod, sub, lo, hi)
            total += c
            tt += t
        assert total == whole and tt == whole_tests


def test_huge_pencils_inside_and_outside_a_slab(psa, stages, torch_mod, oracle):
    """Pencils beyond a workgroup's sorting capacity, listed by the level-Y scan for the tile-parallel sort: a slab
    call sorts only those of its own layers and its halo layer, whether the other layers' points are passed or not."""
    import oracle_lib
    rng = np.random.default_rng(61)
    xyz = np.concatenate([_cell_box(rng, 30_000, (77, 200, 0), (78, 201, 400)),
                          _cell_box(rng, 8_200, (78, 200, 0), (79, 201, 400)),
                          _cell_box(rng, 12_000, (79, 201, 50), (80, 202, 350)),
                          rng.uniform(-9.999, 9.999, (60_000, 3))]).astype(np.float32)
    whole, whole_tests, _ = gpu_count(psa, stages, torch_mod, xyz)
    assert whole == oracle.count_pairs(xyz)[0]
    keys = oracle.cell_keys(xyz)
    for cuts in ([0, 78, 400], [0, 77, 79, 80, 400]):
        total = tt = total_all = 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            c, t, _ = gpu_count(psa, stages, torch_mod, oracle_lib.slab_points(xyz, keys, lo, hi), lo, hi)
            total += c
            tt += t
            total_all += gpu_count(psa, stages, torch_mod, xyz, lo, hi)[0]       # every point passed, slab restricted
        assert total == whole and tt == whole_tests and total_all == whole, cuts


def test_streaming_interface_threads(psa, torch_mod, oracle):
    """pc_create / pc_append from several threads / pc_finish: the reference's seams."""
    import threading
    rng = np.random.default_rng(8)
    xyz = rng.uniform(-1.5, 1.5, (300_000, 3)).astype(np.float32)
    want, want_tests = oracle.count_pairs(xyz)
    pc = psa.PairCounter(device=0, expected_points=1000)        # small hint: forces growth
    parts = np.array_split(xyz, 7)
    th = [threading.Thread(target=pc.append, args=(p,)) for p in parts]
    [t.start() for t in th]
    [t.join() for t in th]
    assert pc.size() == len(xyz)
    assert pc.finish() == want
    assert pc.tests() == want_tests
    ms = pc.phase_ms()
    assert ms["binning"] > 0 and ms["count"] > 0
    with pytest.raises(psa.PairCountError):
        pc.append(xyz[:10])
    pc.close()


def test_layers_are_counted_as_blocks_arrive(psa, torch_mod, oracle):
    """The streaming interface takes level X's histogram block by block at pc_append / pc_append_device (blocks
    start and end inside tiles of 4096 points); pc_finish starts from it.  Same count as the oracle for ragged
    blocks from host and device memory; a point outside the grid in an early block is still an error at
    pc_finish; a geometry set after points were appended makes pc_finish count afresh."""
    rng = np.random.default_rng(81)
    xyz = np.concatenate([rng.uniform(-9.999, 9.999, (150_000, 3)),
                          rng.normal(0.0, 0.08, (60_000, 3))]).astype(np.float32)
    rng.shuffle(xyz)
    want, want_tests = oracle.count_pairs(xyz)
    cuts = [0, 1, 4095, 4096, 4097, 12_288, 70_001, 70_002, 131_072, 200_003, len(xyz)]
    with psa.PairCounter(device=0, expected_points=5000) as pc:
        for q, (a, b) in enumerate(zip(cuts, cuts[1:])):
            if q % 3 == 2:
                pc.append_device(dev(torch_mod, xyz[a:b]))
            else:
                pc.append(xyz[a:b])
        assert pc.finish() == want and pc.tests() == want_tests
    bad = xyz.copy()
    bad[100, 0] = 10.5                                  # x outside the grid, first block
    with psa.PairCounter(device=0, expected_points=len(bad)) as pc:
        for a, b in zip(cuts, cuts[1:]):
            pc.append(bad[a:b])
        with pytest.raises(psa.PairCountError):
            pc.finish()
    # geometry after the appends: the counts taken under the default geometry are dropped
    import ctypes
    g = psa.Geometry(ncell=399)
    inner = xyz[(xyz < 9.94).all(axis=1)]
    want_inner = oracle.count_pairs(inner)[0]
    with psa.PairCounter(device=0, expected_points=1000) as pc:
        pc.append(inner[:77_777])
        psa._check(psa.lib().pc_set_geometry(pc._h, ctypes.byref(g)))
        pc.append(inner[77_777:])
        assert pc.finish() == want_inner


def test_append_from_page_locked_blocks(psa, torch_mod, oracle):
    """pc_host_alloc / pc_append from several producer threads, each refilling its own
    page-locked block (what the host program's parser threads do)."""
    import threading
    rng = np.random.default_rng(18)
    xyz = rng.uniform(-1.5, 1.5, (300_000, 3)).astype(np.float32)
    want, want_tests = oracle.count_pairs(xyz)
    pc = psa.PairCounter(device=0, expected_points=1000)
    errors = []

    def producer(part):
        try:
            with psa.HostBlock(4096) as blk:
                for first in range(0, len(part), 4096):
                    m = min(4096, len(part) - first)
                    blk.array[:m] = part[first:first + m]
                    pc.append(blk.array[:m])
        except Exception as e:                               # surfaced after join
            errors.append(e)

    th = [threading.Thread(target=producer, args=(p,)) for p in np.array_split(xyz, 5)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert pc.size() == len(xyz)
    assert pc.finish() == want
    assert pc.tests() == want_tests
    pc.close()


def test_device_generator_equals_host_generator(psa, torch_mod):
    """pc_generate_device against the host generator, bit for bit: the ramp stream walks every
    six-decimal value of the domain (all 19 998 001 of them, both signs, leading-zero quirks
    included), then the two workload streams at an offset, then a slab selection."""
    import importlib
    torch = torch_mod
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    n_ramp = (2 * 9_999_000 + 1 + 2) // 3 + 5
    for kind, n, sigma, first in ((ht.RAMP, n_ramp, 0, 0), (ht.UNIFORM, 2_000_000, 0, 12_345),
                                  (ht.BLOBS, 2_000_000, 120_000, 777), (ht.BLOBS, 500_000, 500_000, 0)):
        want = ht.generate(kind, 42, n, sigma, first=first)
        got = psa.generate_device(kind, 42, n, sigma, first=first).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (kind, n, sigma)
    want = ht.generate(ht.UNIFORM, 9, 1_000_000, 0, x_lo=100, x_hi=150)
    got = psa.generate_device(ht.UNIFORM, 9, 1_000_000, 0, x_lo=100, x_hi=150).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_full_size_properties(psa, torch_mod):
    """10 M uniform points (config 2 size): properties that need no oracle run --
    permutation invariance, shard additivity, and the tests/count relation."""
    torch = torch_mod
    n = 10_000_000
    g = torch.Generator(device="cuda:0")
    g.manual_seed(1234)
    xyz = (torch.rand((n, 3), device="cuda:0", generator=g) * 19.998 - 9.999).contiguous()
    st = psa.DeviceStages(n, "cuda:0")
    st.count_pairs(xyz)
    st.count_tests()
    whole, tests = st.pairs(), st.tests()
    assert 2.9e6 < whole < 3.6e6 and 1.9e7 < tests < 2.3e7          # analytic expectation 3.27e6 / 2.1e7
    perm = torch.randperm(n, device="cuda:0", generator=g)
    st.count_pairs(xyz[perm].contiguous())
    assert st.pairs() == whole
    table = st.table[: NBB ** 3 + 1]
    assert int(table[-1].item()) == n
    assert bool((table[1:] >= table[:-1]).all())
    # two shards of the already sorted array
    srt = st.sorted[:n].clone()
    cut = 200
    first_hi = int(table[(cut + 1) * NBB * NBB].item())
    first_lo = int(table[cut * NBB * NBB].item())
    st.count_pairs(srt[:first_hi].contiguous(), 0, cut)
    a = st.pairs()
    st.count_pairs(srt[first_lo:].contiguous(), cut, 400)
    assert a + st.pairs() == whole


def test_config3_size_properties(psa, torch_mod):
    """100 M uniform points (the metric's configuration, same stream as bench.py):
    no oracle run at this size, so the count is checked through properties --
    the analytic expectation, permutation invariance, the table, and additivity
    over four x slabs each holding only its own points plus the halo layer."""
    import importlib
    torch = torch_mod
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    n = 100_000_000
    host = ht.generate(ht.UNIFORM, 7, n)
    xyz = torch.from_numpy(host).to("cuda:0")
    del host
    st = psa.DeviceStages(n, "cuda:0")
    st.count_pairs(xyz)
    st.count_tests()
    whole, tests = st.pairs(), st.tests()
    assert 3.2e8 < whole < 3.35e8 and 2.05e9 < tests < 2.15e9        # analytic expectation 3.27e8 / 2.1e9
    table = st.table[: NBB ** 3 + 1]
    assert int(table[-1].item()) == n
    assert bool((table[1:] >= table[:-1]).all())
    srt = st.sorted[:n].clone()
    g = torch.Generator(device="cuda:0")
    g.manual_seed(99)
    perm = torch.randperm(n, device="cuda:0", generator=g)
    shuffled = xyz[perm].contiguous()
    del perm, xyz
    st.count_pairs(shuffled)
    assert st.pairs() == whole
    del shuffled
    # four slabs cut out of the sorted array (layers are contiguous runs): own layers + halo layer
    table = st.table[: NBB ** 3 + 1].clone()
    cuts = [0, 100, 200, 300, 400]
    total = total_tests = 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        a = int(table[lo * NBB * NBB].item())
        b = int(table[min(hi + 1, NBB) * NBB * NBB].item())
        st.count_pairs(srt[a:b].contiguous(), lo, hi)
        st.count_tests(lo, hi)
        total += st.pairs()
        total_tests += st.tests()
    assert total == whole and total_tests == tests


@pytest.mark.parametrize("shape", ["pencil", "two_pencils", "sheet", "gradient", "needle_in_dust", "layer_pair", "heavy_block"])
def test_density_regimes(psa, stages, torch_mod, oracle, shape):
    """Inputs that drive the pair test through every window regime (sparse groups,
    wide groups, dense windows, piece-wise streaming) and through the switches
    between them inside one pencil."""
    rng = np.random.default_rng(sum(shape.encode()))
    if shape == "pencil":                  # one z pencil, ~100 points per cell
        xyz = _cell_box(rng, 40_000, (200, 200, 0), (201, 201, 400))
    elif shape == "two_pencils":           # neighbours in y: every point has dense row-1 candidates
        xyz = _cell_box(rng, 60_000, (10, 399 - 1, 0), (11, 400, 400))
    elif shape == "sheet":                 # one x layer: rows 2-4 are empty everywhere
        xyz = _cell_box(rng, 200_000, (399, 0, 0), (400, 400, 400))
    elif shape == "gradient":              # occupancy climbing from 0 to ~250 per cell along z
        z = np.sqrt(rng.uniform(0, 1, 120_000)) * 19.9 - 9.95
        xy = _cell_box(rng, 120_000, (100, 100, 0), (103, 103, 1))[:, :2]
        xyz = np.column_stack([xy, z])
    elif shape == "needle_in_dust":        # one cell with 5000 points inside sparse background
        xyz = np.concatenate([rng.uniform(-9.999, 9.999, (150_000, 3)),
                              _cell_box(rng, 5000, (57, 311, 123), (58, 312, 124))])
    elif shape == "heavy_block":           # an over-full cell (listed for the thread-per-particle kernel) amid full neighbours
        xyz = np.concatenate([_cell_box(rng, 12_000, (120, 77, 301), (121, 78, 302)),
                              _cell_box(rng, 27 * 300, (119, 76, 300), (122, 79, 303)),
                              _cell_box(rng, 9_000, (120, 77, 302), (121, 78, 303))])    # and a second one right above it
    else:                                  # two adjacent x layers, moderately dense: all five rows populated
        xyz = _cell_box(rng, 400_000, (250, 150, 150), (252, 250, 250))
    xyz = xyz.astype(np.float32)
    want, want_tests = oracle.count_pairs(xyz)
    for bv, cv in VARIANTS:
        got, tests, _ = gpu_count(psa, stages, torch_mod, xyz, bin_v=bv, count_v=cv)
        assert got == want, (shape, bv, cv)
        assert tests == want_tests


@pytest.mark.parametrize("shape", ["filled_slab", "pencil_100", "one_cell_700", "ragged", "face_pencils", "over_full"])
def test_dense_segment_kernel(psa, stages, torch_mod, oracle, shape):
    """The segments of cut pencils are worked off by the dense kernel (candidates in registers) or by the group
    kernel -- chosen on the device from how well the listed segments are filled, or forced with
    pc_set_variant("dense", 2 / 0): same count as the oracle every way.  The shapes cover one
    chunk of home particles and one piece of candidates (the usual case), several of each (more than 64 points in
    a cell, more than 256 candidates), empty cells inside a segment, pencils at the faces of the grid (missing
    rows), and cells handed on to the over-full-cell kernel."""
    rng = np.random.default_rng(sum(shape.encode()))
    if shape == "filled_slab":             # 16 points per cell in a 2 x 40 x 400 block: every pencil is cut into segments
        xyz = _cell_box(rng, 2 * 40 * 400 * 16, (30, 100, 0), (32, 140, 400))
    elif shape == "pencil_100":            # ~100 per cell: two chunks of home particles, several pieces of candidates
        xyz = np.concatenate([_cell_box(rng, 40_000, (200, 200, 0), (201, 201, 400)),
                              _cell_box(rng, 40_000, (201, 199, 0), (202, 202, 400))])
    elif shape == "one_cell_700":          # a cell of 700 points (11 chunks of home particles) inside a filled pencil
        xyz = np.concatenate([_cell_box(rng, 8_000, (77, 77, 0), (78, 78, 400)),
                              _cell_box(rng, 700, (77, 77, 200), (78, 78, 201)),
                              _cell_box(rng, 300, (78, 78, 199), (79, 79, 202))])
    elif shape == "ragged":                # occupancy from 0 to ~250 per cell with empty stretches
        z = np.sqrt(rng.uniform(0, 1, 120_000)) * 19.9 - 9.95
        z = z[(np.floor((z + 10) / 0.05).astype(int) % 7) != 3]
        xy = _cell_box(rng, len(z), (100, 100, 0), (103, 103, 1))[:, :2]
        xyz = np.column_stack([xy, z])
    elif shape == "face_pencils":          # pencils on the y = 0, y = 399 and x = 399 faces: rows outside the grid
        xyz = np.concatenate([_cell_box(rng, 12_000, (10, 0, 0), (11, 2, 400)),
                              _cell_box(rng, 12_000, (10, 398, 0), (11, 400, 400)),
                              _cell_box(rng, 12_000, (399, 0, 0), (400, 1, 400)),
                              _cell_box(rng, 12_000, (399, 399, 0), (400, 400, 400))])
    else:                                  # cells whose home particles x candidates pass the over-full limit
        xyz = np.concatenate([_cell_box(rng, 6_000, (120, 77, 301), (121, 78, 302)),
                              _cell_box(rng, 4_000, (120, 77, 302), (121, 78, 303)),
                              _cell_box(rng, 20_000, (120, 77, 0), (121, 78, 400))])
    xyz = xyz.astype(np.float32)
    want, want_tests = oracle.count_pairs(xyz)
    for dense in (1, 0, 2):                # chosen on the device from the listed segments / group kernel / dense kernel
        psa.set_variant("dense", dense)
        got, tests, _ = gpu_count(psa, stages, torch_mod, xyz, bin_v=1, count_v=5)
        assert got == want, (shape, dense)
        assert tests == want_tests
    try:                                   # the device-side choice pushed both ways by its threshold
        psa.set_variant("dense", 1)
        for dense_min in (1, 65535):
            psa.set_variant("dense_min", dense_min)
            assert gpu_count(psa, stages, torch_mod, xyz, bin_v=1, count_v=5)[0] == want, (shape, dense_min)
    finally:
        psa.set_variant("dense_min", 11)


def test_heavy_cell_list_overflow(psa, stages, torch_mod, oracle):
    """Over-full cells are handed to a follow-up kernel through a list; when the list is too
    short (capacity 0, 1, a few) the wave that met the cell processes it itself: same count."""
    rng = np.random.default_rng(77)
    xyz = np.concatenate([_cell_box(rng, 3_000, (5, 5, 5), (6, 6, 6)),           # 12 chunks of 256
                          _cell_box(rng, 2_500, (5, 5, 6), (6, 6, 7)),           # 10 chunks, the cell above
                          _cell_box(rng, 2_000, (300, 399, 399), (301, 400, 400)),
                          rng.uniform(-9.999, 9.999, (50_000, 3))]).astype(np.float32)
    want, want_tests = oracle.count_pairs(xyz)
    try:
        for cap in (65536, 0, 1, 9, 12, 21, 22, 30):
            psa.set_variant("heavy", cap)
            for cv in (3, 5):
                got, tests, _ = gpu_count(psa, stages, torch_mod, xyz, bin_v=1, count_v=cv)
                assert got == want and tests == want_tests, (cap, cv)
    finally:
        psa.set_variant("heavy", 65536)


def test_two_contexts_finish_concurrently(psa, torch_mod, oracle):
    """Two streaming contexts on one device, finished from two threads at once:
    the library serialises them (the pair test has one work area per device)."""
    import threading
    rng = np.random.default_rng(21)
    sets = [rng.uniform(-2.0, 2.0, (250_000, 3)).astype(np.float32),
            _cell_box(rng, 30_000, (50, 60, 0), (51, 61, 400)).astype(np.float32)]   # the second one has planned segments
    want = [oracle.count_pairs(x)[0] for x in sets]
    psa.set_variant("count", 3)                 # the variant with the per-device work area
    for _ in range(3):
        pcs = [psa.PairCounter(device=0, expected_points=len(x)) for x in sets]
        for pc, x in zip(pcs, sets):
            pc.append(x)
        got = [None, None]

        def finish(i):
            got[i] = pcs[i].finish()
        th = [threading.Thread(target=finish, args=(i,)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert got == want
        [pc.close() for pc in pcs]


def test_segment_list_overflow(psa, stages, torch_mod, oracle):
    """Heavy pencils are cut into segments by a planning kernel; when the segment
    list is too small for all of them the rest must be counted whole, slots
    reserved by a pencil that did not fit must be skipped, and capacity 0 must
    switch the mechanism off -- same count every time."""
    rng = np.random.default_rng(5)
    # six heavy pencils of different weight next to each other, plus background
    parts = [rng.uniform(-9.999, 9.999, (50_000, 3))]
    for q, npts in enumerate((3000, 9000, 20000, 5000, 40000, 2500)):
        parts.append(_cell_box(rng, npts, (120, 200 + q, 0), (121, 201 + q, 400)))
    xyz = np.concatenate(parts).astype(np.float32)
    want, want_tests = oracle.count_pairs(xyz)
    try:
        for cap in (262144, 0, 1, 3, 7, 16, 40, 1000):
            psa.set_variant("segments", cap)
            for cv in (3, 5):      # (by density this input would take variant 0; 5 also puts what it defers on the list)
                got, tests, _ = gpu_count(psa, stages, torch_mod, xyz, count_v=cv)
                assert got == want, (cap, cv)
                assert tests == want_tests
        with pytest.raises(psa.PairCountError):
            psa.set_variant("segments", 262145)
    finally:
        psa.set_variant("segments", 262144)
