"""GPU: slab routing on the devices (pc_export_layers / pc_import_device) -- points appended unrouted to several
arrival contexts, grouped by x layer there, pulled by the slab owners as layer ranges, counted; against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_layers_exported_and_imported_add_up_to_the_oracle_count(psa, oracle):
    rng = np.random.default_rng(41)
    pts = np.concatenate([rng.uniform(-9.999, 9.999, (300_000, 3)),
                          np.clip(rng.normal([0.02, 1.0, -2.0], 0.3, (200_000, 3)), -9.999, 9.999)]).astype(np.float32)
    rng.shuffle(pts)
    want, want_tests = oracle.count_pairs(pts)
    layer = oracle.cell_keys(pts) // (403 * 403)
    for cuts in ([0, 400], [0, 200, 400], [0, 199, 201, 400], [0, 50, 100, 150, 200, 250, 300, 350, 400]):
        n = len(cuts) - 1
        arrivals = [psa.PairCounter(0, len(pts) // n + 1) for _ in range(n)]
        parts = np.array_split(pts, 3 * n)                      # blocks land on the arrival contexts round robin
        for i, part in enumerate(parts):
            arrivals[i % n].append(part)
        exported = [a.export_layers() for a in arrivals]
        for a, (ptr, starts) in zip(arrivals, exported):
            assert starts[0] == 0 and starts[400] == starts[403] == a.size() and np.all(np.diff(starts) >= 0)
        # the starts are the layer histogram of what each context was given
        mine = np.concatenate([p for i, p in enumerate(parts) if i % n == 0])
        hist = np.bincount(oracle.cell_keys(mine) // (403 * 403), minlength=403)
        assert np.array_equal(np.diff(exported[0][1]), hist[:403])
        total = tests = 0
        for o in range(n):
            lo, hi = cuts[o], (400 if o == n - 1 else cuts[o + 1] + 1)      # layers [lo, hi): the slab and its halo layer
            with psa.PairCounter(0, 1024, slab=(cuts[o], cuts[o + 1])) as owner:
                for ptr, starts in exported:
                    owner.import_device(0, ptr + 12 * int(starts[lo]), int(starts[hi] - starts[lo]))
                assert owner.size() == int(np.count_nonzero((layer >= lo) & (layer < hi)))
                total += owner.finish()
                tests += owner.tests()
        assert total == want and tests == want_tests, cuts
        for a in arrivals:
            a.close()


def test_export_reports_a_point_outside_the_grid(psa):
    pts = np.random.default_rng(3).uniform(-9.9, 9.9, (5000, 3)).astype(np.float32)
    pts[77, 0] = -10.5
    with psa.PairCounter(0, len(pts)) as pc:
        pc.append(pts)
        with pytest.raises(psa.PairCountError) as e:
            pc.export_layers()
        assert e.value.code == -3
