#!/usr/bin/env python3
"""Generates the committed golden vectors from the UNMODIFIED reference.

Runs only where /root/reference exists (the build container): it calls
oracle/_ref/libref.so (reference sources compiled where they lie, through
oracle/ref_shim.c) and the reference binary oracle/_ref/main.  Everything it
writes is data -- inputs and expected outputs -- never reference code.

    python tests/golden/make_golden.py

Files written next to this script:
  parser_d3.npy        u32 bits of parse_float for "d.ddd" (10000 unsigned then 10000 signed tokens)
  parser_samples.txt   "<token> <hex bits> <chars consumed>" for sampled formats + the quirk domain
  cell_index.npz       floats around all 401 cell boundaries + reference index (main.c:134-136)
  dist2_pairs.npz      4096 near-threshold point pairs, reference dist2 bits and predicate
  *.xyz                small text inputs; counts.json holds the reference's printed count for
                       each (several -p/-c settings) and the in-process count on the parsed floats
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFDIR = os.path.join(ROOT, "oracle", "_ref")

ref = ctypes.CDLL(os.path.join(REFDIR, "libref.so"))
ref.ref_parse_float_bits.restype = ctypes.c_uint32
ref.ref_parse_float_bits.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]
ref.ref_cell_key.restype = ctypes.c_int64
ref.ref_cell_key.argtypes = [ctypes.POINTER(ctypes.c_float)]
ref.ref_dist2_bits.restype = ctypes.c_uint32
ref.ref_dist2_bits.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
ref.ref_count_points.restype = ctypes.c_long
ref.ref_count_points.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
ref.ref_parse_line.restype = ctypes.c_long
ref.ref_parse_line.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_float)]


def parse_bits(tok):
    used = ctypes.c_long()
    b = ref.ref_parse_float_bits(tok.encode() + b"\n", ctypes.byref(used))
    return b, used.value


def d3_tokens():
    for s in ("", "-"):
        for i in range(10000):
            yield "%s%d.%03d" % (s, i // 1000, i % 1000)


def gen_parser():
    bits = np.array([parse_bits(t)[0] for t in d3_tokens()], dtype=np.uint32)
    np.save(os.path.join(HERE, "parser_d3.npy"), bits)
    rng = np.random.default_rng(20250314)
    toks = []
    for dec in (1, 2, 4, 6, 8, 10, 17, 20):
        for v in rng.uniform(-9.999, 9.999, 150):
            t = "%.*f" % (dec, v)
            if abs(float(t)) < 10.0:       # "10.0" has two integer digits: outside the grammar
                toks.append(t)
    for fmt in ("%.3e", "%.6e", "%.16e"):
        for v in rng.uniform(-0.9999, 0.9999, 150):        # negative exponents only (positive exits)
            s = fmt % v
            if "e+" in s:
                continue
            toks.append(s)
    # the leading-zero scan quirk (parse.h:80-88): every shape around it
    for last in range(0, 100):
        toks.append("0.%06d" % last)
        toks.append("-0.%06d" % last)
        toks.append("0.%04d" % last)
        toks.append("-0.%04d" % last)
    for k in range(1, 10):
        toks += ["0." + "0" * (k - 1) + "5", "-0." + "0" * (k - 1) + "5", "%d.0" % k, "-%d.0" % k,
                 "%d.%s" % (k, "9" * 17), "0." + "0" * k + "1" * (17 - k)]
    toks += ["0.0", "-0.0", "0.000000", "-0.000000", "9.999999", "-9.999999", "9.99999999999999999",
             "1.000000", "0.050000", "-0.050000", "0.1", "0.5", "0.25", "5.000000e-01", "1.0e-05", "-3.25e-10"]
    with open(os.path.join(HERE, "parser_samples.txt"), "w") as f:
        for t in toks:
            b, used = parse_bits(t)
            f.write("%s %08x %d\n" % (t, b, used))
    return len(bits) + len(toks)


def cell_of(f):
    p = (ctypes.c_float * 3)(f, -10.0, -10.0)
    key = ref.ref_cell_key(p)
    return int(key // (403 * 403))


def gen_cells():
    vals = []
    for k in range(0, 401):
        b = np.float32(-10.0 + 0.05 * k)
        cur = b
        for _ in range(3):
            cur = np.nextafter(cur, np.float32(-100))
        for _ in range(7):
            vals.append(cur)
            cur = np.nextafter(cur, np.float32(100))
    rng = np.random.default_rng(7)
    vals += list(rng.uniform(-9.9999, 9.9999, 4000).astype(np.float32))
    vals = np.array(vals, dtype=np.float32)
    vals = vals[(vals > -10.0) & (vals < 10.0)]
    idx = np.array([cell_of(float(v)) for v in vals], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "cell_index.npz"), x=vals, index=idx)
    return len(vals)


def gen_dist2():
    rng = np.random.default_rng(99)
    n = 4096
    a = np.zeros((n, 4), dtype=np.float32)
    b = np.zeros((n, 4), dtype=np.float32)
    a[:, :3] = rng.uniform(-9.9, 9.9, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = 0.05 * (1.0 + rng.uniform(-5e-7, 5e-7, n))
    r[: n // 8] = 0.05                     # as close to the threshold as float allows
    b[:, :3] = (a[:, :3].astype(np.float64) + d * r[:, None]).astype(np.float32)
    # coincident and axis-aligned exact cases
    b[n - 64:n - 32, :3] = a[n - 64:n - 32, :3]
    a[n - 32:, :3] = 0.0
    b[n - 32:, :3] = 0.0
    b[n - 32:, 0] = np.float32(0.05)
    b[n - 16:, 0] = np.nextafter(np.float32(0.05), np.float32(1))
    # the 4th lane is junk in the reference (main.c:130): make it hostile
    a[:, 3] = rng.choice(np.array([np.nan, np.inf, -np.inf, 1e30, 0.0], dtype=np.float32), n)
    b[:, 3] = rng.choice(np.array([np.nan, np.inf, -np.inf, -1e30, 1.0], dtype=np.float32), n)
    bits = np.zeros(n, dtype=np.uint32)
    within = np.zeros(n, dtype=np.uint8)
    w = ctypes.c_int()
    for i in range(n):
        pa = a[i].ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        pb = b[i].ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        bits[i] = ref.ref_dist2_bits(pa, pb, ctypes.byref(w))
        within[i] = w.value
    np.savez_compressed(os.path.join(HERE, "dist2_pairs.npz"), a=a, b=b, d2_bits=bits, within=within)
    return n, int(within.sum())


def fmt_line(p, dec=6):
    return " ".join("%.*f" % (dec, v) for v in p)


def write_xyz(name, pts, dec=6, eol="\n", trailing=True):
    lines = [fmt_line(p, dec) for p in pts]
    data = eol.join(lines) + (eol if trailing else "")
    path = os.path.join(HERE, name)
    with open(path, "w", newline="") as f:
        f.write(data)
    return path


def run_ref(path, p, c):
    out = subprocess.run([os.path.join(REFDIR, "main"), "-f", path, "-p", str(p), "-c", str(c)],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    if out.returncode != 0:
        raise RuntimeError("reference failed on %s: rc=%d %s" % (path, out.returncode, out.stderr))
    return int(out.stdout.strip().splitlines()[-1])


def parse_file_with_ref(path):
    """Parsed float triples exactly as the reference's parse_line yields them."""
    data = open(path, "rb").read() + b"\0"
    pos, pts = 0, []
    buf = (ctypes.c_float * 3)()
    end = len(data) - 1
    while pos < end:
        used = ref.ref_parse_line(data[pos:pos + 256], buf)
        pts.append((buf[0], buf[1], buf[2]))
        pos += used
    return np.array(pts, dtype=np.float32)


def gen_files():
    rng = np.random.default_rng(4242)
    files = {}
    # 12 points, 2 distinct positions 0.01 apart in one cell: every pair counts -> C(12,2)=66
    pts = [(0.1, 0.1, 0.1), (0.11, 0.1, 0.1)] * 6
    files["coincident12.xyz"] = write_xyz("coincident12.xyz", pts)
    # sparse uniform inside a 1.2-wide box straddling the origin (about 0.3 points per cell)
    files["uniform_3k.xyz"] = write_xyz("uniform_3k.xyz", rng.uniform(-0.6, 0.6, (3000, 3)))
    # dense blob: cells holding more than 8 and more than 64 points (several ChunkLists per cell)
    blob = np.clip(rng.normal(0.0, 0.045, (1500, 3)) + np.array([3.21, -7.77, 0.02]), -9.999, 9.999)
    files["blob_1500.xyz"] = write_xyz("blob_1500.xyz", blob)
    # first and last occupied cell on every axis, with partners across the corner cells
    corner = []
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                base = np.array([sx, sy, sz]) * 9.97
                corner.append(base + rng.uniform(-0.029, 0.029, (40, 3)))
    corner = np.clip(np.concatenate(corner), -9.999999, 9.999999)
    files["corners_320.xyz"] = write_xyz("corners_320.xyz", corner)
    # a thin sheet one cell thick in x at the slab seam used by the sharding tests (x index 199/200)
    sheet = rng.uniform(-1.0, 1.0, (2500, 3))
    sheet[:, 0] = rng.uniform(-0.06, 0.06, 2500)
    files["seam_sheet_2500.xyz"] = write_xyz("seam_sheet_2500.xyz", sheet)
    # formatting variants of one small input
    small = rng.uniform(-0.2, 0.2, (400, 3))
    files["small_lf.xyz"] = write_xyz("small_lf.xyz", small)
    files["small_crlf.xyz"] = write_xyz("small_crlf.xyz", small, eol="\r\n")
    files["small_noeol.xyz"] = write_xyz("small_noeol.xyz", small, trailing=False)
    files["small_dec3.xyz"] = write_xyz("small_dec3.xyz", small, dec=3)
    files["small_dec9.xyz"] = write_xyz("small_dec9.xyz", small, dec=9)

    counts = {}
    for name, path in sorted(files.items()):
        runs = {"%d/%d" % (p, c): run_ref(path, p, c) for p, c in ((1, 1), (4, 4), (3, 5), (8, 8))}
        vals = set(runs.values())
        xyz = parse_file_with_ref(path)
        inproc = int(ref.ref_count_points(xyz.ctypes.data, xyz.shape[0], 4))
        assert len(vals) == 1 and inproc in vals, (name, runs, inproc)
        counts[name] = {"count": inproc, "points": int(xyz.shape[0]), "reference_runs": runs}
        np.save(os.path.join(HERE, name.replace(".xyz", ".f32.npy")), xyz)
    with open(os.path.join(HERE, "counts.json"), "w") as f:
        json.dump(counts, f, indent=1, sort_keys=True)
    return counts


if __name__ == "__main__":
    if not os.path.exists(os.path.join(REFDIR, "libref.so")):
        sys.exit("oracle/_ref is not built (needs /root/reference): run make -C oracle")
    print("parser tokens:", gen_parser())
    print("cell boundary floats:", gen_cells())
    print("dist2 pairs (n, within):", gen_dist2())
    for k, v in gen_files().items():
        print(k, v["points"], v["count"])
