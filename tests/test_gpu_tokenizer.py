"""GPU: the tokenizer kernels (text -> float triples on the device) against the reference-generated
vectors, the golden files and the oracle's tokenizer.  Bit patterns: the bar is exact equality."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def bits_of(t):
    return t.cpu().numpy().view(np.uint32)


def test_tokens_match_reference_vectors(psa, torch_mod, golden_dir):
    """Every golden token (20 000 exhaustive d.ddd tokens, the sampled and quirk-domain ones) as the x of a line."""
    want = np.load(os.path.join(golden_dir, "parser_d3.npy"))
    toks = ["%s%d.%03d" % (sign, v // 1000, v % 1000) for sign in ("", "-") for v in range(10000)]
    bits = [int(b) for b in want]
    with open(os.path.join(golden_dir, "parser_samples.txt")) as f:
        for line in f:
            tok, hexbits, _ = line.split()
            toks.append(tok)
            bits.append(int(hexbits, 16))
    text = "".join("%s 1.500000 -2.250000\n" % t for t in toks).encode()
    got, n, status = psa.parse_text_device(text)
    assert status == 0 and n == len(toks)
    b = bits_of(got)
    assert np.array_equal(b[:, 0], np.array(bits, dtype=np.uint32))
    assert (b[:, 1] == np.float32(1.5).view(np.uint32)).all() and (b[:, 2] == np.float32(-2.25).view(np.uint32)).all()


def test_golden_files(psa, torch_mod, golden_dir):
    """The golden .xyz inputs (LF, CRLF, no final newline, 3 and 9 decimals, ...): the reference's floats, in file order."""
    with open(os.path.join(golden_dir, "counts.json")) as f:
        counts = json.load(f)
    for name in counts:
        data = open(os.path.join(golden_dir, name), "rb").read()
        want = np.load(os.path.join(golden_dir, name.replace(".xyz", ".f32.npy")))
        got, n, status = psa.parse_text_device(data)
        assert status == 0 and n == len(want), name
        assert np.array_equal(bits_of(got), want.view(np.uint32).reshape(-1, 3)), name


def test_mixed_token_shapes_against_the_oracle(psa, torch_mod, oracle):
    rng = np.random.default_rng(12)

    def token():
        v = int(rng.integers(0, 10 ** 7))
        sign = "-" if rng.integers(0, 2) else ""
        shape = int(rng.integers(0, 9))
        if shape <= 3:
            return "%s%d.%06d" % (sign, v // 10 ** 6, v % 10 ** 6)
        if shape == 4:
            return "%s%d.%03d" % (sign, v % 10, v % 1000)
        if shape == 5:
            return "%s%d.%017d" % (sign, v % 10, v * 9973 % 10 ** 17)
        if shape == 6:
            return "%s%d.%06de-%d" % (sign, v % 10, v % 10 ** 6, 1 + v % 30)
        if shape == 7:
            return "%s%d.%025d" % (sign, v % 10, v * 7919)              # more than 17 decimals: the rest is skipped
        return "%s0.%06d" % (sign, v % 100)                              # leading zeros: the short-scan domain

    for eol, last in (("\n", "\n"), ("\r\n", "\r\n"), ("\n", ""), ("\n\n", "\n")):
        lines = [" ".join(token() for _ in range(3)) for _ in range(30_000)]
        data = (eol.join(lines) + last).encode()
        want = oracle.parse_text(data)
        got, n, status = psa.parse_text_device(data)
        assert status == 0 and n == len(want) == 30_000
        assert np.array_equal(bits_of(got), want.view(np.uint32)), (eol, last)


def test_errors_and_limits(psa, torch_mod):
    good = b"1.000000 2.000000 3.000000\n"
    for bad, bit in ((b"12.5 1.0 1.0\n", 2), (b"1.5e+01 1.0 1.0\n", 4), (b"1." + b"0" * 400 + b" 1.0 1.0\n", 8)):
        _, n, status = psa.parse_text_device(good * 300 + bad + good * 300)
        assert status & bit, bad[:20]
    got, n, status = psa.parse_text_device(good * 1000, capacity=10)
    assert n == 1000 and status == 16 and got.shape[0] == 10
    got, n, status = psa.parse_text_device(b"")
    assert n == 0 and status == 0


def test_append_text_counts_like_append(psa, torch_mod, oracle, golden_dir):
    """pc_append_text on blocks of whole records from several threads: the oracle's count; text the reference exits on
    is PC_ERR_TEXT there and at pc_finish."""
    import importlib
    import threading
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    xyz = ht.generate(ht.BLOBS, 5, 400_000, 150_000)
    lines = ["%.6f %.6f %.6f\n" % tuple(p) for p in xyz.astype(np.float64)]
    want_pts = oracle.parse_text("".join(lines).encode())
    want, want_tests = oracle.count_pairs(want_pts)
    blocks = ["".join(lines[a:a + 57_143]).encode() for a in range(0, len(lines), 57_143)]
    with psa.PairCounter(device=0, expected_points=1000) as pc:
        got_n = [0] * len(blocks)

        def work(i):
            got_n[i] = pc.append_text(blocks[i])
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(blocks))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert sum(got_n) == len(lines) == pc.size()
        assert pc.finish() == want and pc.tests() == want_tests
    with psa.PairCounter(device=0) as pc:
        pc.append_text(blocks[0])
        with pytest.raises(psa.PairCountError):
            pc.append_text(b"1.0 2.0 33.0\n")
        with pytest.raises(psa.PairCountError):
            pc.finish()
