#!/usr/bin/env python3
"""End-to-end wall time on a full-size text file: the C host program (tokenize on
the host, bin + count on the GPU) next to the reference binary, same file, same
thread count, best of two runs each.

    python3 tools/e2e_full.py [--workload uniform100m] [--threads 16] [--out gpurun_out/e2e.json]
"""
import argparse
import importlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(cmd, env=None, runs=2, limit=None):
    """Best wall time of `runs` runs; None if a run exceeds `limit` seconds."""
    best = None
    for _ in range(runs):
        t = time.time()
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        while True:                      # (a line a minute while a long run is under way)
            try:
                so, se = p.communicate(timeout=60)
                break
            except subprocess.TimeoutExpired:
                print("  %s: running for %d s" % (os.path.basename(cmd[0]), time.time() - t), flush=True)
                if limit and time.time() - t > limit:
                    p.kill()
                    p.communicate()
                    print("  %s: no result within %d s" % (os.path.basename(cmd[0]), limit), flush=True)
                    return None
        w = time.time() - t
        r = subprocess.CompletedProcess(cmd, p.returncode, so, se)
        if r.returncode != 0:
            raise SystemExit("%s failed: %s" % (cmd[0], r.stderr[-300:]))
        if best is None or w < best[0]:
            best = (w, r)
        print("  %s: %.3f s" % (os.path.basename(cmd[0]), w), flush=True)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="uniform100m")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--points", type=int, default=None, help="another number of points than the workload's")
    ap.add_argument("--reference-runs", type=int, default=2)
    ap.add_argument("--reference-limit", type=int, default=None, help="give the reference up after this many seconds")
    ap.add_argument("--skip-reference", action="store_true", help="time only the product's host program")
    ap.add_argument("--gpu-tokenizer", type=int, nargs="*", default=[],
                    help="also time the host program with PAIRCOUNT_GPU_TOKENIZER=K for each K given (K threads at a time "
                         "send their block of text through the GPU tokenizer, the others tokenize on the host)")
    ap.add_argument("--slabs", type=int, nargs="*", default=[],
                    help="also time the host program with -g S for each S given, all S slabs on device 0 "
                         "(the multi-GPU path of the host program on a one-GPU box)")
    ap.add_argument("--routes", nargs="*", default=["device"], choices=["device", "host"],
                    help="with --slabs: where the points are routed to their slabs (device: layers exported and imported "
                         "on the GPUs, the default of the host program; host: point by point in the parser threads)")
    ap.add_argument("--sidecar", action="store_true",
                    help="also time the host program on the .f32 form of the same points (no tokenizing)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "e2e_full.json"))
    args = ap.parse_args()
    from __graft_entry__ import load_package
    import bench
    load_package()
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    kind_name, points, sigma, _ = bench.WORKLOADS[args.workload]
    kind = ht.UNIFORM if kind_name == "uniform" else ht.BLOBS
    if args.points:
        points = args.points
    tmp = tempfile.mkdtemp(prefix="pce2e_")
    path = os.path.join(tmp, "input.xyz")
    t = time.time()
    n = ht.generate_text(path, kind, 7, points, sigma)
    print("generated %d points, %d bytes in %.1f s" % (n, os.path.getsize(path), time.time() - t), flush=True)
    out = {"workload": args.workload, "points": n, "file_bytes": os.path.getsize(path), "threads": args.threads}
    try:
        w, r = timed([ht.MAIN_PATH, "-f", path, "-p", str(args.threads)], env=dict(os.environ, PAIRCOUNT_TIMER="1"))
        out["gpu_host_program"] = {"wall_s": w, "count": int(r.stdout.strip()),
                                   "phases_s": {m.group(1): float(m.group(2)) / 1e6
                                                for m in re.finditer(r"(\w[\w()]*):\s+(\d+) us", r.stderr)}}
        for k in args.gpu_tokenizer:
            w, r = timed([ht.MAIN_PATH, "-f", path, "-p", str(args.threads)],
                         env=dict(os.environ, PAIRCOUNT_TIMER="1", PAIRCOUNT_GPU_TOKENIZER=str(k)))
            out["gpu_host_program_gpu_tokenizer_%d" % k] = {
                "wall_s": w, "count": int(r.stdout.strip()),
                "phases_s": {m.group(1): float(m.group(2)) / 1e6 for m in re.finditer(r"(\w[\w()]*):\s+(\d+) us", r.stderr)}}
        for k, route in ((k, route) for k in args.slabs for route in args.routes):
            w, r = timed([ht.MAIN_PATH, "-f", path, "-p", str(args.threads), "-g", str(k)],
                         env=dict(os.environ, PAIRCOUNT_TIMER="1", PAIRCOUNT_DEVICES=",".join(["0"] * k), PAIRCOUNT_ROUTE=route))
            out["gpu_host_program_%d_slabs_on_one_gpu%s" % (k, "" if route == "device" else "_routed_on_host")] = {
                "wall_s": w, "count": int(r.stdout.strip()),
                "phases_s": {m.group(1): float(m.group(2)) / 1e6 for m in re.finditer(r"(\w[\w()]*):\s+(\d+) us", r.stderr)},
                "slabs": [ln.strip() for ln in r.stderr.splitlines() if ln.strip().startswith("slab")]}
        ref = os.path.join(ROOT, "oracle", "_ref", "main_timer")
        if os.path.exists(ref) and not args.skip_reference:
            got = timed([ref, "-f", path, "-p", str(args.threads), "-c", str(args.threads)], runs=args.reference_runs,
                        limit=args.reference_limit)
            if got is None:
                out["reference"] = {"wall_s": None, "note": "no result within %d s" % args.reference_limit}
            else:
                w, r = got
                out["reference"] = {"wall_s": w, "count": int([ln for ln in r.stdout.splitlines() if ln.strip().isdigit()][-1]),
                                    "phases_s": {m.group(1): int(m.group(2)) / 1e6 for m in re.finditer(r"(\w+):\s+(\d+) us", r.stdout)}}
                out["counts_equal"] = out["reference"]["count"] == out["gpu_host_program"]["count"]
        if args.sidecar:
            side = os.path.join(tmp, "input.f32")
            ht.generate(kind, 7, points, sigma).tofile(side)
            try:
                w, r = timed([ht.MAIN_PATH, "-f", side, "-p", str(args.threads)], env=dict(os.environ, PAIRCOUNT_TIMER="1"))
                out["gpu_host_program_f32_sidecar"] = {
                    "wall_s": w, "count": int(r.stdout.strip()), "file_bytes": os.path.getsize(side),
                    "phases_s": {m.group(1): float(m.group(2)) / 1e6 for m in re.finditer(r"(\w[\w()]*):\s+(\d+) us", r.stderr)}}
            finally:
                os.remove(side)
    finally:
        os.remove(path)
        os.rmdir(tmp)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
