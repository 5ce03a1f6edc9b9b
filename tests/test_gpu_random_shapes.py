"""GPU: the default HIP path against the CPU oracle on seeded inputs of mixed shapes -- boxes of many densities,
blobs, sheets, rods along each axis, duplicated points, points on cell faces -- each also cut into random x slabs.
Integer results: the bar is exact equality of pairs, reference tests and the binning table."""
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
    return psa.DeviceStages(400_000, "cuda:0")


def _shape(rng, kind, n):
    c = rng.uniform(-8.0, 8.0, 3)
    if kind == "box":                        # density from 0.01 to hundreds of points per cell
        half = 10.0 ** rng.uniform(-1.2, 0.9)
        p = c + rng.uniform(-half, half, (n, 3))
    elif kind == "blobs":
        k = int(rng.integers(1, 9))
        centres = rng.uniform(-8, 8, (k, 3))
        p = centres[rng.integers(0, k, n)] + rng.normal(0, 10.0 ** rng.uniform(-2.0, -0.3), (n, 3))
    elif kind == "sheet":                    # a thin slab normal to a random axis
        p = rng.uniform(-9.9, 9.9, (n, 3))
        ax = int(rng.integers(0, 3))
        p[:, ax] = c[ax] + rng.uniform(-0.06, 0.06, n)
        other = [a for a in range(3) if a != ax]
        p[:, other] = c[other] + rng.uniform(-1.5, 1.5, (n, 2))
    elif kind == "rod":                      # a rod along a random axis: pencils (z), rows of pencils (y), layers (x)
        ax = int(rng.integers(0, 3))
        p = c + rng.uniform(-0.07, 0.07, (n, 3))
        p[:, ax] = rng.uniform(-9.9, 9.9, n)
    elif kind == "duplicates":               # few distinct positions, many copies (d2 = 0 counts)
        base = c + rng.uniform(-0.3, 0.3, (max(1, n // 50), 3))
        p = base[rng.integers(0, len(base), n)]
    else:                                    # "faces": coordinates on and next to cell boundaries
        cells = rng.integers(150, 250, (n, 3))
        off = rng.choice([0.0, 1e-7, -1e-7, 0.025, 0.049999], (n, 3))
        p = cells * 0.05 - 10.0 + off
    return np.clip(p, -9.9999, 9.9999).astype(np.float32)


@pytest.mark.parametrize("seed", range(30))
def test_random_shapes_match_oracle(psa, stages, torch_mod, oracle, seed):
    import oracle_lib
    rng = np.random.default_rng(1000 + seed)
    kinds = ["box", "blobs", "sheet", "rod", "duplicates", "faces"]
    parts = []
    for _ in range(int(rng.integers(1, 4))):
        parts.append(_shape(rng, kinds[int(rng.integers(0, len(kinds)))], int(rng.integers(1, 90_000))))
    xyz = np.concatenate(parts)
    rng.shuffle(xyz)
    n = len(xyz)
    want, want_tests, srt, start = oracle.count_pairs(xyz, want_sorted=True)
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)
    t = torch_mod.from_numpy(xyz).to("cuda:0")
    assert stages.count_pairs(t, 0, 400) == n
    stages.count_tests(0, 400)
    assert (stages.pairs(), stages.tests()) == (want, want_tests), (seed, n)
    table = stages.table[:NBB ** 3 + 1].cpu().numpy().view(np.uint32)
    assert np.array_equal(table, start)
    # random x slabs: the slab's own points plus its halo layer, and all points with the slab restricted
    keys = oracle.cell_keys(xyz)
    layers = np.unique(keys // (NBB * NBB))
    cuts = sorted(set([0, 400] + [int(v) for v in rng.choice(layers, size=min(3, len(layers)), replace=False)]))
    total = total_all = 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        sub = oracle_lib.slab_points(xyz, keys, lo, hi)
        stages.count_pairs(torch_mod.from_numpy(sub).to("cuda:0"), lo, hi)
        total += stages.pairs()
        stages.count_pairs(t, lo, hi)
        total_all += stages.pairs()
    assert total == want and total_all == want, (seed, cuts)
