"""CPU: the oracle (oracle/pairs_oracle.c) against the vectors generated from the
unmodified reference (tests/golden/make_golden.py).  This is what pins the
oracle; the GPU parity tests then compare the HIP path with the oracle."""
import json
import os

import numpy as np
import pytest


def test_parser_exhaustive_three_decimals(oracle, golden_dir):
    want = np.load(os.path.join(golden_dir, "parser_d3.npy"))
    i = 0
    for sign in ("", "-"):
        for v in range(10000):
            bits, err = oracle.parse_token("%s%d.%03d" % (sign, v // 1000, v % 1000))
            assert err == 0
            assert bits == int(want[i]), (sign, v, hex(bits), hex(int(want[i])))
            i += 1


def test_parser_sampled_formats_and_quirk_domain(oracle, golden_dir):
    n = 0
    with open(os.path.join(golden_dir, "parser_samples.txt")) as f:
        for line in f:
            tok, hexbits, used = line.split()
            bits, err = oracle.parse_token(tok)
            assert err == 0
            assert bits == int(hexbits, 16), (tok, hex(bits), hexbits)
            n += 1
    assert n > 2000


def test_parser_known_quirks(oracle):
    # reference parse.h:80-88 never looks at the last decimal of an unsigned token:
    # "0.000004" parses as 1e-6 exactly as the reference does (bits pinned in parser_samples.txt)
    bits, _ = oracle.parse_token("0.000004")
    assert np.array([bits], dtype=np.uint32).view(np.float32)[0] == np.float32(1.0e-6)
    # -0.000 is +0.0f (parse.h:72-74)
    assert oracle.parse_token("-0.000000")[0] == 0
    # errors the reference exits on are reported, not parsed
    assert oracle.parse_token("12.5")[1] == 1
    assert oracle.parse_token("1.5e+01")[1] == 2


def test_cell_index_boundaries(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "cell_index.npz"))
    xs, want = z["x"], z["index"]
    pts = np.full((len(xs), 3), -10.0, dtype=np.float32)
    pts[:, 0] = xs
    keys = oracle.cell_keys(pts)
    assert np.array_equal(keys // (403 * 403), want)
    assert want.min() == 0 and want.max() == 399


def test_dist2_and_predicate(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "dist2_pairs.npz"))
    a, b, bits, within = z["a"], z["b"], z["d2_bits"], z["within"]
    for i in range(len(a)):
        got_bits, got_within = oracle.dist2_bits(a[i, :3], b[i, :3])
        assert got_bits == int(bits[i]), i
        assert got_within == bool(within[i]), i
    # both outcomes and the exact-threshold value are represented
    assert 0 < within.sum() < len(within)
    thr = np.array([0.0025], dtype=np.float32).view(np.uint32)[0]
    assert (bits == thr).sum() > 10


def _counts(golden_dir):
    with open(os.path.join(golden_dir, "counts.json")) as f:
        return json.load(f)


def test_end_to_end_fixtures_text(oracle, golden_dir):
    for name, info in _counts(golden_dir).items():
        data = open(os.path.join(golden_dir, name), "rb").read()
        xyz = oracle.parse_text(data)
        assert xyz.shape[0] == info["points"], name
        want_xyz = np.load(os.path.join(golden_dir, name.replace(".xyz", ".f32.npy")))
        assert np.array_equal(xyz.view(np.uint32), want_xyz.view(np.uint32)), name
        got, tests = oracle.count_pairs(xyz, threads=3)
        assert got == info["count"], name
        assert tests >= got


def test_fixture_expectations(golden_dir):
    c = _counts(golden_dir)
    assert c["coincident12.xyz"]["count"] == 66
    assert c["small_lf.xyz"]["count"] == c["small_crlf.xyz"]["count"] == c["small_noeol.xyz"]["count"]
    for info in c.values():
        assert len(set(info["reference_runs"].values())) == 1    # -p/-c invariance of the reference


def test_bruteforce_invariant_matches_binned_count(oracle):
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-0.5, 0.5, (1500, 3)).astype(np.float32)
    xyz[:50] = xyz[50:100]                       # coincident points count too
    assert oracle.brute(xyz) == oracle.count_pairs(xyz)[0]


def test_slab_decomposition_sums_to_whole(oracle):
    import oracle_lib
    rng = np.random.default_rng(11)
    xyz = rng.uniform(-1.0, 1.0, (20000, 3)).astype(np.float32)
    whole, tests = oracle.count_pairs(xyz)
    keys = oracle.cell_keys(xyz)
    for cuts in ([0, 200, 400], [0, 190, 199, 200, 201, 400], [0, 100, 100, 400]):
        total = ttot = 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sub = oracle_lib.slab_points(xyz, keys, lo, hi)
            c, t = oracle.count_pairs(sub, x_lo=lo, x_hi=hi)
            total += c
            ttot += t
        assert total == whole and ttot == tests, cuts


def test_out_of_domain_rejected(oracle):
    with pytest.raises(ValueError):
        oracle.count_pairs(np.array([[10.5, 0, 0]], dtype=np.float32))


def test_other_geometry_stencil_count_equals_all_pairs_brute_force(oracle):
    """With another geometry (nothing of the reference pins it) the binned stencil count of the checker must
    equal the count over ALL pairs, as long as the cutoff does not exceed the cell edge."""
    rng = np.random.default_rng(11)
    try:
        for origin, cell, ncell, cutoff, n in ((-10.0, 0.05, 400, 0.03, 3000), (0.0, 0.25, 37, 0.2, 2500),
                                               (-3.0, 0.015, 400, 0.0149, 4000)):
            d2max = float(np.float32(cutoff) * np.float32(cutoff))
            oracle.set_geometry(origin, cell, ncell, d2max)
            lo, hi = origin + 1e-6, origin + ncell * cell * 0.2
            xyz = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
            got, tests = oracle.count_pairs(xyz)
            assert got == oracle.brute_anycell(xyz) == oracle.brute(xyz) and got > 0
    finally:
        oracle.set_geometry()
    xyz = rng.uniform(-1, 1, (2000, 3)).astype(np.float32)
    assert oracle.count_pairs(xyz)[0] == oracle.brute(xyz)
