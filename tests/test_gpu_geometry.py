"""GPU: configurable geometry (domain origin, cell edge, cells per axis, cutoff).
The reference compiles these in (main.c:30-31,165), so nothing of it pins other values:
parity here is "unpinned by the reference".  The checker is the oracle with the same
constants, whose stencil count is itself checked against a brute force over all pairs.
The default geometry passed explicitly must give the classic entry points' results bit for bit."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NBB = 403

GEOMETRIES = [  # origin, cell, ncell, cutoff
    (-10.0, 0.05, 400, 0.03),          # the reference's grid, a smaller cutoff
    (0.0, 0.25, 37, 0.2),              # a coarse grid over [0, 9.25)
    (-3.0, 0.015, 400, 0.0149),        # a fine grid, cutoff just below the cell edge
    (-123.456, 0.7, 111, 0.5),         # an origin binary32 does not represent exactly
]


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(autouse=True)
def default_variants(psa, oracle):
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)
    yield
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)
    oracle.set_geometry()


def _points(rng, origin, cell, ncell, n, frac=0.25):
    lo = origin + 1e-4 * cell
    return rng.uniform(lo, origin + ncell * cell * frac, (n, 3)).astype(np.float32)


def test_default_geometry_given_explicitly_is_the_classic_path(psa, torch_mod, oracle):
    torch = torch_mod
    rng = np.random.default_rng(3)
    xyz = rng.uniform(-2.0, 2.0, (400_000, 3)).astype(np.float32)
    t = torch.from_numpy(xyz).to("cuda:0")
    st = psa.DeviceStages(len(xyz), "cuda:0")
    st.count_pairs(t)
    want, table = st.pairs(), st.table.clone()
    keys = st.cell_keys(t).clone()
    g = psa.Geometry()
    st.count_pairs(t, geometry=g)
    assert st.pairs() == want == oracle.count_pairs(xyz)[0]
    assert torch.equal(st.table[: NBB ** 3 + 1], table[: NBB ** 3 + 1])
    assert torch.equal(st.cell_keys(t, geometry=g), keys)


@pytest.mark.parametrize("origin,cell,ncell,cutoff", GEOMETRIES)
def test_geometry_counts_match_oracle_and_brute_force(psa, torch_mod, oracle, origin, cell, ncell, cutoff):
    torch = torch_mod
    g = psa.Geometry(origin=origin, cell=cell, ncell=ncell, cutoff=cutoff)
    oracle.set_geometry(origin, cell, ncell, g.d2max)
    rng = np.random.default_rng(ncell)
    st = psa.DeviceStages(300_000, "cuda:0")
    # small: all-pairs brute force; larger: the oracle's stencil count (dense enough for the staged pair tests)
    small = _points(rng, origin, cell, ncell, 3000, frac=0.08)
    want_small = oracle.brute_anycell(small)
    assert want_small > 0 and want_small == oracle.count_pairs(small)[0]
    big = _points(rng, origin, cell, ncell, 300_000, frac=0.15 if ncell >= 200 else 0.9)
    want_big, want_tests = oracle.count_pairs(big)
    keys_big = oracle.cell_keys(big)
    for bv, cv in ((1, 4), (1, 5), (1, 3), (0, 0), (1, 0)):
        psa.set_variant("bin", bv)
        psa.set_variant("count", cv)
        st.count_pairs(torch.from_numpy(small).to("cuda:0"), geometry=g)
        assert st.pairs() == want_small, (bv, cv)
        t = torch.from_numpy(big).to("cuda:0")
        st.count_pairs(t, geometry=g)
        st.count_tests()
        assert st.pairs() == want_big and st.tests() == want_tests, (bv, cv)
        assert np.array_equal(st.cell_keys(t, geometry=g).cpu().numpy().astype(np.int64), keys_big)
    # a point in cell ncell (one past the domain) or a whole cell below the origin is a domain error
    # (the reference's (int) conversion truncates toward zero: up to one cell below the origin is still cell 0)
    for bad in (origin + (ncell + 0.5) * cell, origin - 1.5 * cell):
        pts = small[:10].copy()
        pts[7, 1] = bad
        st.count_pairs(torch.from_numpy(pts).to("cuda:0"), geometry=g)
        with pytest.raises(psa.PairCountError):
            st.pairs()


@pytest.mark.parametrize("origin,cell,ncell,cutoff", GEOMETRIES[1:])
def test_cell_index_shortcut_is_exact_for_the_geometry(psa, torch_mod, origin, cell, ncell, cutoff):
    """binary32 shortcut against the binary64 formula on every float, with this geometry's constants."""
    torch = torch_mod
    g = psa.Geometry(origin=origin, cell=cell, ncell=ncell, cutoff=cutoff)
    bad = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    rc = psa.lib().pc_selfcheck_cell_index_geometry_device(ctypes.byref(g), ctypes.c_void_p(bad.data_ptr()),
                                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, psa.lib().pc_last_error()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0


def test_streaming_context_with_geometry_and_rejections(psa, torch_mod, oracle):
    g = psa.Geometry(origin=0.0, cell=0.25, ncell=37, cutoff=0.2)
    oracle.set_geometry(0.0, 0.25, 37, g.d2max)
    rng = np.random.default_rng(5)
    xyz = _points(rng, 0.0, 0.25, 37, 50_000, frac=0.9)
    pc = psa.PairCounter(device=0, expected_points=len(xyz), geometry=g)
    pc.append(xyz)
    assert pc.finish() == oracle.count_pairs(xyz)[0]
    pc.close()
    # cutoff above the cell edge, no cells, a NaN edge: refused
    for kw in (dict(cell=0.05, cutoff=0.0501), dict(ncell=0), dict(ncell=401), dict(cell=float("nan")), dict(d2max=-1.0)):
        with pytest.raises(psa.PairCountError) as e:
            psa.PairCounter(device=0, expected_points=16, geometry=psa.Geometry(**kw))
        assert e.value.code == -1
