"""GPU: the fused pair test (count variant 7: the last binning level inside the pair-test kernel) against the
default route and the CPU oracle -- chunked pencils, pencils handed to the whole-capacity instantiation, the
device-side switch back to the classic kernels, and the streaming interface on top of it.  Exact equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def default_variants(psa):
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)
    yield
    psa.set_variant("bin", 1)
    psa.set_variant("count", 4)


def _slab(rng, n, lo, hi):
    """n uniform points whose x cells lie in [lo, hi] (the layers a slab [lo, hi) and its halo hold)."""
    p = rng.uniform(-9.9999, 9.9999, (n, 3))
    p[:, 0] = rng.uniform(lo * 0.05 - 10.0 + 1e-5, (min(hi, 399) + 1) * 0.05 - 10.0 - 1e-5, n)
    return p.astype(np.float32)


def _both(psa, st, torch, pts, lo, hi):
    t = torch.from_numpy(pts).to("cuda:0")
    out = {}
    for v in (4, 7):
        psa.set_variant("count", v)
        st.count_pairs(t, lo, hi)
        out[v] = st.pairs()
    st.count_tests(lo, hi)
    return out[4], out[7], st.tests()


def test_fused_matches_default_and_oracle(psa, oracle):
    import torch
    rng = np.random.default_rng(11)
    st = psa.DeviceStages(1_300_000, "cuda:0")
    # 1.5 and 1.9 points per cell (two and three chunks per pencil), the last layers of the grid; then the first
    # layers as a whole-grid call, which the oracle can check
    for n, lo, hi in ((int(4 * 160000 * 1.5), 100, 103), (int(2 * 160000 * 1.9), 398, 400)):
        a, b, _ = _both(psa, st, torch, _slab(rng, n, lo, hi), lo, hi)
        assert a == b, (n, lo, hi)
    pts = _slab(rng, 400_000, 0, 2)
    a, b, tests = _both(psa, st, torch, pts, 0, 400)
    want, want_tests = oracle.count_pairs(pts)[:2]
    assert a == b == want and tests == want_tests
    # three full cells of an otherwise thin pencil: more than a chunk holds, less than the kernel's capacity
    thick = np.concatenate([rng.uniform([0.001, 0.001, 1.001], [0.049, 0.049, 1.149], (2600, 3)), _slab(rng, 50_000, 200, 201)]).astype(np.float32)
    a, b, _ = _both(psa, st, torch, thick, 0, 400)
    assert a == b == oracle.count_pairs(thick)[0]
    # a neighbourhood beyond the capacity: the classic kernels take over on the device
    blob = rng.normal(0.0, 0.1, (200_000, 3)).astype(np.float32)
    a, b, _ = _both(psa, st, torch, blob, 0, 400)
    assert a == b == oracle.count_pairs(blob)[0]


def test_fused_streaming_interface(psa, oracle):
    rng = np.random.default_rng(12)
    pts = _slab(rng, 600_000, 50, 52)
    want, want_tests = oracle.count_pairs(pts)[:2]
    psa.set_variant("count", 7)
    with psa.PairCounter(0, len(pts)) as pc:
        pc.append(pts[:250_000])
        pc.append(pts[250_000:])
        assert pc.finish() == want
        assert pc.tests() == want_tests          # counted by the fused kernel itself: there is no table
    # a z coordinate outside the domain is seen by the fused kernel only
    bad = pts.copy()
    bad[1234, 2] = 10.5
    with psa.PairCounter(0, len(bad)) as pc:
        pc.append(bad)
        with pytest.raises(psa.PairCountError) as e:
            pc.finish()
        assert e.value.code == -3


def test_two_contexts_hold_different_variants(psa, oracle):
    """pc_context_set_variant: a context keeps its own choice whatever the process-wide knobs say when it is
    finished -- here the fused pair test in one context (its test count comes from the kernel) and the atomic
    scatter + thread-per-particle kernels in the other."""
    rng = np.random.default_rng(13)
    pts = _slab(rng, 500_000, 10, 12)
    want, want_tests = oracle.count_pairs(pts)[:2]
    a, b = psa.PairCounter(0, len(pts)), psa.PairCounter(0, len(pts))
    a.set_variant("count", 7)
    b.set_variant("bin", 0)
    b.set_variant("count", 0)
    with pytest.raises(psa.PairCountError):
        a.set_variant("segments", 5)            # a process-wide size, not a per-context choice
    a.append(pts)
    b.append(pts)
    psa.set_variant("count", 3)                 # the process-wide choice differs from both
    assert b.finish() == want and a.finish() == want
    assert a.tests() == b.tests() == want_tests
    with pytest.raises(psa.PairCountError):
        a.set_variant("count", 5)               # after pc_finish
    a.close()
    b.close()
