"""GPU: the C host program (same command line as the reference's ./main) and the
sharded streaming helper, end to end on the golden inputs."""
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ht(psa):
    return importlib.import_module("particle_simulation_amd.hosttools")


def test_main_prints_the_reference_count(ht, golden_dir):
    with open(os.path.join(golden_dir, "counts.json")) as f:
        counts = json.load(f)
    for name, info in counts.items():
        for p in ("1", "4", "7"):
            r = subprocess.run([ht.MAIN_PATH, "-f", os.path.join(golden_dir, name), "-p", p, "-c", "3"],
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, (name, r.stderr)
            assert r.stdout == "%d\n" % info["count"], (name, p, r.stdout)     # exactly "%ld\n" (main.c:336)


def test_main_with_gpu_tokenizer(ht, oracle, golden_dir, tmp_path):
    """PAIRCOUNT_GPU_TOKENIZER=1: the parser threads hand blocks of whole lines to pc_append_text; same output on the
    golden files (CRLF, no final newline, 3 and 9 decimals) and on a file large enough for several blocks per thread;
    text the reference exits on gives its exit status."""
    env = dict(os.environ, PAIRCOUNT_GPU_TOKENIZER="1")
    with open(os.path.join(golden_dir, "counts.json")) as f:
        counts = json.load(f)
    for name, info in counts.items():
        for p in ("1", "5"):
            r = subprocess.run([ht.MAIN_PATH, "-f", os.path.join(golden_dir, name), "-p", p], capture_output=True, text=True,
                               timeout=120, env=env)
            assert r.returncode == 0, (name, r.stderr)
            assert r.stdout == "%d\n" % info["count"], (name, p, r.stdout)
    path = str(tmp_path / "b4m.xyz")
    ht.generate_text(path, ht.BLOBS, 9, 4_000_000, 400_000)              # 114 MB: more than one 32 MB block for a thread
    want = oracle.count_pairs(ht.generate(ht.BLOBS, 9, 4_000_000, 400_000))[0]
    for p in ("2", "7"):
        r = subprocess.run([ht.MAIN_PATH, "-f", path, "-p", p], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and int(r.stdout) == want, (p, r.stderr)
    bad = str(tmp_path / "bad.xyz")
    with open(bad, "w") as f:
        f.write("1.000000 2.000000 3.000000\n" * 1000 + "12.5 1.0 1.0\n" + "1.000000 2.000000 3.000000\n" * 1000)
    r = subprocess.run([ht.MAIN_PATH, "-f", bad, "-p", "3"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 255 and r.stdout == ""
    # a record longer than the GPU tokenizer takes (the reference skips any number of redundant decimals,
    # parse.h:54-56): its block is tokenized on the host, the count is that of the plain run
    long_path = str(tmp_path / "long.xyz")
    with open(long_path, "w") as f:
        f.write("1.000000 2.000000 3.000000\n" * 700)
        f.write("1.00" + "0" * 400 + " 2.000000 3.010000\n")
        f.write("1.000000 2.000000 3.020000\n" * 700)
    plain = subprocess.run([ht.MAIN_PATH, "-f", long_path, "-p", "3"], capture_output=True, text=True, timeout=120)
    r = subprocess.run([ht.MAIN_PATH, "-f", long_path, "-p", "3"], capture_output=True, text=True, timeout=120, env=env)
    assert plain.returncode == 0 and r.returncode == 0, (plain.stderr, r.stderr)
    assert r.stdout == plain.stdout == "%d\n" % (1401 * 1400 // 2)


def test_main_on_generated_file_with_timer(ht, oracle, tmp_path):
    path = str(tmp_path / "u300k.xyz")
    n = ht.generate_text(path, ht.UNIFORM, 3, 300_000)
    xyz = ht.generate(ht.UNIFORM, 3, 300_000)
    # squeeze into a box so that there are pairs: regenerate as blobs instead
    path2 = str(tmp_path / "b300k.xyz")
    ht.generate_text(path2, ht.BLOBS, 3, 300_000, 150_000)
    pts = ht.generate(ht.BLOBS, 3, 300_000, 150_000)
    want, tests = oracle.count_pairs(pts)
    env = dict(os.environ, PAIRCOUNT_TIMER="1")
    r = subprocess.run([ht.MAIN_PATH, "-f", path2, "-p", "8"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    assert int(r.stdout) == want and want > 0
    for phase in ("init", "parsing", "binning", "count"):
        assert phase in r.stderr
    assert ("tests:  %10d" % tests) in r.stderr or str(tests) in r.stderr
    r = subprocess.run([ht.MAIN_PATH, "-f", path, "-p", "3"], capture_output=True, text=True, timeout=120)
    assert int(r.stdout) == oracle.count_pairs(xyz)[0] and n == 300_000


def test_main_on_binary_sidecar(ht, golden_dir, tmp_path):
    """A ".f32" file (the parsed bits of a .xyz file) gives the count of the text file it came from."""
    with open(os.path.join(golden_dir, "counts.json")) as f:
        counts = json.load(f)
    for name, info in counts.items():
        pts = np.load(os.path.join(golden_dir, name.replace(".xyz", ".f32.npy")))
        path = str(tmp_path / name.replace(".xyz", ".f32"))
        pts.astype(np.float32).tofile(path)
        for p in ("1", "5"):
            r = subprocess.run([ht.MAIN_PATH, "-f", path, "-p", p], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, (name, r.stderr)
            assert r.stdout == "%d\n" % info["count"], (name, p, r.stdout)
    empty = tmp_path / "empty.f32"
    empty.write_bytes(b"")
    r = subprocess.run([ht.MAIN_PATH, "-f", str(empty)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout == "0\n"


def test_main_multi_gpu_slabs(ht, oracle, golden_dir, tmp_path):
    """-g N: the parser threads route points to N slab contexts (here all on GPU 0, standing in for N GPUs),
    the slabs are finished concurrently and summed: the golden counts, for text and binary input, for
    equal and for explicit cuts, and with the timer's per-slab report."""
    with open(os.path.join(golden_dir, "counts.json")) as f:
        counts = json.load(f)
    for name, info in counts.items():
        for g, extra in (("2", {}), ("3", {"PAIRCOUNT_CUTS": "0,199,201,400"}), ("8", {}),
                         ("2", {"PAIRCOUNT_ROUTE": "host"}), ("8", {"PAIRCOUNT_ROUTE": "host"})):
            env = dict(os.environ, PAIRCOUNT_DEVICES=",".join("0" * int(g)), **extra)
            r = subprocess.run([ht.MAIN_PATH, "-f", os.path.join(golden_dir, name), "-p", "5", "-g", g],
                               capture_output=True, text=True, timeout=120, env=env)
            assert r.returncode == 0, (name, g, r.stderr)
            assert r.stdout == "%d\n" % info["count"], (name, g, r.stdout)
    path = str(tmp_path / "b500k.xyz")
    ht.generate_text(path, ht.BLOBS, 11, 500_000, 150_000)
    pts = ht.generate(ht.BLOBS, 11, 500_000, 150_000)
    want, tests = oracle.count_pairs(pts)
    binp = str(tmp_path / "b500k.f32")
    pts.tofile(binp)
    for f in (path, binp):
        for route in ({}, {"PAIRCOUNT_ROUTE": "host"}):          # routed on the GPUs (export / import of layers), or by the parser threads
            env = dict(os.environ, PAIRCOUNT_GPUS="4", PAIRCOUNT_DEVICES="0,0,0,0", PAIRCOUNT_TIMER="1", **route)
            r = subprocess.run([ht.MAIN_PATH, "-f", f, "-p", "8"], capture_output=True, text=True, timeout=120, env=env)
            assert r.returncode == 0, r.stderr
            assert int(r.stdout) == want and want > 0
            assert str(tests) in r.stderr and r.stderr.count("slab") == 4
            assert ("export(wall)" in r.stderr) == (not route)
    # an out-of-domain point (only a binary file can hold one) is still an error, whichever slab it was routed to
    for axis in range(3):
        far = pts[:1000].copy()
        far[500, axis] = 10.5
        fp = str(tmp_path / ("far%d.f32" % axis))
        far.tofile(fp)
        for route in ({}, {"PAIRCOUNT_ROUTE": "host"}):
            r = subprocess.run([ht.MAIN_PATH, "-f", fp, "-g", "2"], capture_output=True, text=True, timeout=60,
                               env=dict(os.environ, PAIRCOUNT_DEVICES="0,0", **route))
            assert r.returncode == 255 and "domain" in r.stderr


def test_main_rejects_a_bad_gpu_count(ht, golden_dir):
    """PAIRCOUNT_GPUS is held to the same rule as -g: a positive count, else the reference's failure status."""
    path = os.path.join(golden_dir, "pairs12.xyz")
    if not os.path.exists(path):
        path = os.path.join(golden_dir, sorted(f for f in os.listdir(golden_dir) if f.endswith(".xyz"))[0])
    for bad in ("0", "-1", "two"):
        r = subprocess.run([ht.MAIN_PATH, "-f", path], capture_output=True, text=True, timeout=60,
                           env=dict(os.environ, PAIRCOUNT_GPUS=bad))
        assert r.returncode == 255 and "PAIRCOUNT_GPUS" in r.stderr and r.stdout == "", (bad, r.stderr)
    r = subprocess.run([ht.MAIN_PATH, "-f", path, "-g", "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and r.stdout == ""


def test_main_slab_queue(ht, oracle, golden_dir, tmp_path):
    """PAIRCOUNT_QUEUE=K: K slabs wait in host memory and the GPU threads (here 1, 2 and 3 of them on GPU 0)
    take them from a shared cursor.  The golden counts, the oracle's count on clustered points, every slab
    reported once with the taker that counted it, and a domain error still an error."""
    with open(os.path.join(golden_dir, "counts.json")) as f:
        counts = json.load(f)
    for name, info in counts.items():
        for g, k, extra in (("1", "2", {}), ("2", "7", {}), ("3", "3", {"PAIRCOUNT_CUTS": "0,199,201,400"}), ("2", "64", {})):
            env = dict(os.environ, PAIRCOUNT_DEVICES=",".join("0" * int(g)), PAIRCOUNT_QUEUE=k, **extra)
            r = subprocess.run([ht.MAIN_PATH, "-f", os.path.join(golden_dir, name), "-p", "5", "-g", g],
                               capture_output=True, text=True, timeout=120, env=env)
            assert r.returncode == 0, (name, g, k, r.stderr)
            assert r.stdout == "%d\n" % info["count"], (name, g, k, r.stdout)
    pts = ht.generate(ht.BLOBS, 12, 700_000, 200_000)
    want, tests = oracle.count_pairs(pts)
    binp = str(tmp_path / "b700k.f32")
    pts.tofile(binp)
    path = str(tmp_path / "b700k.xyz")
    ht.generate_text(path, ht.BLOBS, 12, 700_000, 200_000)
    for f in (path, binp):
        env = dict(os.environ, PAIRCOUNT_DEVICES="0,0", PAIRCOUNT_QUEUE="16", PAIRCOUNT_TIMER="1")
        r = subprocess.run([ht.MAIN_PATH, "-f", f, "-p", "8", "-g", "2"], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stderr
        assert int(r.stdout) == want and want > 0
        assert str(tests) in r.stderr and r.stderr.count("slab") == 16 and r.stderr.count("taker") == 2
    far = pts[:1000].copy()
    far[500, 1] = -10.5
    fp = str(tmp_path / "far.f32")
    far.tofile(fp)
    r = subprocess.run([ht.MAIN_PATH, "-f", fp, "-g", "2"], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, PAIRCOUNT_DEVICES="0,0", PAIRCOUNT_QUEUE="5"))
    assert r.returncode == 255 and "domain" in r.stderr
    r = subprocess.run([ht.MAIN_PATH, "-f", binp, "-g", "4"], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, PAIRCOUNT_QUEUE="3"))
    assert r.returncode == 255 and "PAIRCOUNT_QUEUE" in r.stderr


def test_main_error_statuses_of_the_reference(ht, tmp_path):
    """No -f and an empty file fail as the reference does (open: EFAULT; mmap of zero bytes: EINVAL), status 255."""
    r = subprocess.run([ht.MAIN_PATH], capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and r.stderr.startswith("open:")
    empty = tmp_path / "empty.xyz"
    empty.write_bytes(b"")
    r = subprocess.run([ht.MAIN_PATH, "-f", str(empty)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and r.stderr.startswith("mmap:")


def test_main_reports_tokenizer_errors_like_the_reference(ht, tmp_path):
    bad = tmp_path / "bad.xyz"
    bad.write_text("1.000000 2.000000 3.000000\n12.500000 1.000000 1.000000\n" * 8)
    r = subprocess.run([ht.MAIN_PATH, "-f", str(bad)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and "*str != '.'" in r.stderr


def test_sharded_count_single_rank_and_manual_slabs(psa, ht, oracle):
    import torch
    sh = importlib.import_module("particle_simulation_amd.sharding")
    pts = ht.generate(ht.BLOBS, 9, 400_000, 200_000)
    want, _ = oracle.count_pairs(pts)
    dev = torch.device("cuda", 0)
    assert sh.sharded_count(pts, 0, 400, dev) == want
    pop = np.bincount(oracle.cell_keys(pts) // (403 * 403), minlength=400)[:400].astype(np.float64)
    cuts = sh.balanced_cuts(pop * pop, 4)
    total = 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        x0, x1 = sh.halo_range(lo, hi)
        mine = ht.generate(ht.BLOBS, 9, 400_000, 200_000, x_lo=x0, x_hi=x1)
        total += sh.sharded_count(mine, lo, hi, dev)          # one "rank" after another on one GPU
    assert total == want
