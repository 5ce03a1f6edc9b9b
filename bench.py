#!/usr/bin/env python3
"""Benchmark of the pair-counting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One step = one pass of the hot path (cell keys, binning, pair test, and for
N > 1 the 8-byte all-reduce) over synthetic points already resident in HBM.
Workloads (the configurations of the task): uniform100m (default: the configuration the
metric is quoted on), uniform10m, blobs100m, uniform1b (125 M points per GPU,
weak scaling).  For N > 1 the x range is cut into N slabs, one rank per GPU;
each rank holds its slab's points plus the +x halo layer.

Prints ONE JSON line on rank 0.  value = reference dist2() evaluations covered
per second by the whole job (tests of all ranks / max-over-ranks step time).

`python bench.py --gpus N` with N > 1 and no rank environment starts the N ranks
itself: the parent touches no GPU, starts N fresh child processes of this file
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank
0's line and exits with the worst child status.  Under torch.distributed.run the
ranks already exist and nothing is spawned.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak
VALU_NOFMA_TOPS = 78.6           # fp32 vector peak without FMA credit (157.3 TFLOP/s / 2)

WORKLOADS = {
    # name: (kind, total points at N=1, sigma in micro-units, scaling, points multiply with ranks)
    "uniform10m": ("uniform", 10_000_000, 0, "strong"),
    "uniform100m": ("uniform", 100_000_000, 0, "strong"),
    "blobs100m": ("blobs", 100_000_000, 500_000, "strong"),
    "uniform1b": ("uniform", 125_000_000, 0, "weak"),
}


def kernel_bytes(name, n, layers):
    """Algorithmic HBM bytes of one launch of each kernel, n points, `layers` x layers of cells."""
    cells = (layers + 1) * 403 * 403
    all_cells = 403 ** 3 + 2
    table = {
        "pc_a1_keys_hist": 12 * n + 4 * n + 8 * n,          # read xyz, write key, histogram read-modify-write
        "pc_a2_scan_reduce": 4 * all_cells,
        "pc_a2_scan_apply": 8 * all_cells,
        "pc_a3_scatter": 4 * n + 12 * n + 8 * n + 12 * n,   # key, xyz in, cursor rmw, xyz out
        "pc_c0_count_global": 12 * n + 4 * cells,
        "pc_c1_count_tiles": 12 * n + 4 * cells,
        "pc_c2_count_waves": 12 * n + 4 * cells,
        "pc_c3_count_groups": 12 * n + 4 * cells,
        "pc_c5_count_merged": 12 * n + 4 * cells,
        "pc_l1_hist_x": 12 * n,
        "pc_l3_partition_x": 24 * n + 2 * n,                # points in and out, y cells out
        "pc_l4_hist_y": 2 * n,                              # y cells in
        "pc_l6_partition_y": 24 * n,
        "pc_l7_sort_pencils": 24 * n + 4 * all_cells,       # (with pc_l7b_sort_big_pencils, which takes the pencils too big for a wave)
    }
    return table.get(name)


def measured_valu_ceiling():
    """What tools/micro/valu_rate.hip measured on an MI355X (committed under profiles/): the chip's rate of plain
    binary32 vector operations at eight waves per SIMD, and the rate of the pair test's nine-operation distance
    arithmetic alone (no loads, no bookkeeping).  None if no such profile is in the tree."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*valu_issue_rate.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        rows = json.load(f).get("rows", [])
    pick = lambda kind: max((r for r in rows if r["kind"] == kind), key=lambda r: r["waves_per_simd"], default=None)
    mul, test = pick("v_mul_f32"), pick("pair test, 9 plain instructions")
    if not mul or not test:
        return None
    return {"source": os.path.relpath(files[-1], ROOT), "plain_ops_tops": mul["lane_ops_per_s"] / 1e12,
            "clock_ghz": mul["clock_ghz"], "cycles_per_wave_instruction": mul["ns_per_wave_instruction_per_simd"] * mul["clock_ghz"],
            "pair_test_arithmetic_tops": 9 * test["tests_per_s"] / 1e12}


def spawn_ranks(n, cmd=None, timeout_s=None, grace_s=5.0):
    """The parent of a self-launched multi-rank run: no torch, no HIP, only child processes.

    Every child is watched.  When one exits with a non-zero status -- a bad device ordinal, an import error,
    a failed RCCL start-up -- the others would wait for it in init_process_group or a barrier until the
    collective's own timeout; instead they are ended here (SIGTERM, SIGKILL after a grace period), the tail of
    the failing rank's stderr is relayed, and the parent exits with that rank's status.  All ranks keep their
    stderr (temporary files); rank 0's stdout carries the result line."""
    import signal
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = cmd or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    timeout_s = timeout_s or float(os.environ.get("PAIRCOUNT_BENCH_TIMEOUT", "1500"))
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=errs[r], text=True))

    def tail(r, limit=3000):
        errs[r].seek(0)
        return errs[r].read()[-limit:]

    def end_all():
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        t_end = time.time() + grace_s
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t0 = time.time()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc is not None and rc != 0:
                failed = (r, rc, "rank %d exited with status %d" % (r, rc))
                break
        if failed is None and time.time() - t0 > timeout_s:
            failed = (0, 1, "no result within %.0f s" % timeout_s)
        if failed is None:
            time.sleep(0.05)
    if failed is None:
        for r, p in enumerate(procs):            # all have exited: any late non-zero status still counts
            if p.returncode != 0:
                failed = (r, p.returncode, "rank %d exited with status %d" % (r, p.returncode))
                break
    if failed is not None:
        r, rc, what = failed
        end_all()
        sys.stderr.write("bench.py: %s; the other ranks were stopped.  Its stderr ends:\n%s\n" % (what, tail(r)))
        sys.exit(rc if 0 < rc < 256 else 1)
    out0 = procs[0].stdout.read()
    for r in range(n):                           # warnings of the ranks, for the log
        t = tail(r, 600)
        if t.strip():
            sys.stderr.write("[rank %d stderr] %s\n" % (r, t.strip()))
    sys.stdout.write(out0)
    sys.stdout.flush()
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="uniform100m", choices=sorted(WORKLOADS))
    ap.add_argument("--points", type=int, default=0, help="override the workload's point count")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--sigma-micro", type=int, default=None,
                    help="clustered workloads: another blob width than the workload's, in millionths (measurements)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generate", default="host", choices=["host", "device"],
                    help="where the synthetic points are made: the host generator (then copied), or the same "
                         "stream produced on the GPU bit for bit (faster set-up at 1 B points)")
    ap.add_argument("--bin-variant", type=int, default=None)
    ap.add_argument("--count-variant", type=int, default=None)
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="any other knob of pc_set_variant (measurements), e.g. --set split=1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks "
                         "(ranks share devices, the 8-byte sum goes through host memory)")
    ap.add_argument("--as-rank", type=int, nargs=2, metavar=("R", "W"), default=None,
                    help="single process, no collective: time the slab that rank R of a W-rank run would hold "
                         "(a per-rank preview of the N-GPU line on a one-GPU box; not a bench result)")
    ap.add_argument("--event-period", type=int, default=4,
                    help="per-kernel HIP events around every k-th launch of each kernel inside the timed region (1: all)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the process group and run the 8-byte all-reduce also with ONE rank (the RCCL "
                         "communicator, the int64 all_reduce and its teardown then execute on a one-GPU box)")
    ap.add_argument("--no-e2e-full", action="store_true",
                    help="skip the full-size end-to-end leg of cpu_baseline (reference program and the product's host "
                         "program on the whole workload written as a text file)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.as_rank:
        spawn_ranks(args.gpus)

    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    psa = load_package()
    import importlib
    ht = importlib.import_module("particle_simulation_amd.hosttools")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if os.environ.get("PAIRCOUNT_TEST_FAIL_RANK") == str(rank):      # (launcher test: a rank that dies at start-up)
        sys.exit("bench.py: rank %d fails on request (PAIRCOUNT_TEST_FAIL_RANK)" % rank)
    collective = world > 1 or args.force_collective
    if args.backend == "nccl" and world > torch.cuda.device_count():
        sys.exit("bench.py: %d ranks but %d GPUs; --backend gloo rehearses more ranks than GPUs" % (world, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_device = device if args.backend == "nccl" else torch.device("cpu")

    kind_name, base_points, sigma, scaling = WORKLOADS[args.workload]
    if args.sigma_micro is not None:
        sigma = args.sigma_micro
    total_points = args.points or (base_points * world if scaling == "weak" else base_points)
    kind = ht.UNIFORM if kind_name == "uniform" else ht.BLOBS
    sh = importlib.import_module("particle_simulation_amd.sharding")
    slabs, slab = world, rank                    # how many x slabs the domain is cut into, and which one is ours
    if args.as_rank and world == 1:
        slab, slabs = args.as_rank
        total_points = args.points or (base_points * slabs if scaling == "weak" else base_points)
    if slabs > 1 and kind_name == "blobs":
        # clustered input: cut where the estimated work balances (the reference's test count per layer from a
        # block histogram of the stream, plus a per-point share for binning); the per-rank rows of the result
        # line show how well it did
        cost = ht.layer_cost(kind, args.seed, total_points, sigma,
                             threads=max(1, min(16, (os.cpu_count() or 1) // max(world, 1))))
        cuts = sh.balanced_cuts(cost, slabs)
    else:
        cuts = sh.equal_cuts(slabs)
    x_lo, x_hi = cuts[slab], cuts[slab + 1]
    if args.bin_variant is not None:
        psa.set_variant("bin", args.bin_variant)
    if args.count_variant is not None:
        psa.set_variant("count", args.count_variant)
    for kv in args.set:
        name, _, value = kv.partition("=")
        psa.set_variant(name, int(value))

    # ---- synthetic input: this rank's slab plus its +x halo layer, resident in HBM ----
    t_gen = time.time()
    # generator threads of this rank: its share of the CPUs this job may use, at most 16
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    threads = max(1, min(16, cpus // world))
    if args.generate == "device":
        xyz = psa.generate_device(kind, args.seed, total_points, sigma, device=device,
                                  x_lo=x_lo if slabs > 1 else 0, x_hi=min(x_hi, 399) if slabs > 1 else 399)
        torch.cuda.synchronize()
    else:
        if slabs == 1:
            host = ht.generate(kind, args.seed, total_points, sigma, threads=threads)
        else:
            host = ht.generate(kind, args.seed, total_points, sigma, x_lo=x_lo, x_hi=min(x_hi, 399), threads=threads)
        xyz = torch.from_numpy(host).to(device)
        del host
    n = xyz.shape[0]
    t_gen = time.time() - t_gen

    st = psa.DeviceStages(n, device)
    total = torch.zeros(1, dtype=torch.int64, device=red_device)

    def step():
        st.count_pairs(xyz, x_lo, x_hi)          # keys + binning + pair test, one C-ABI call
        if collective:
            total.copy_(st.result[0:1])      # (gloo rehearsal: a synchronising copy to host)
            dist.all_reduce(total)           # the path's only collective: one 8-byte sum

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # per-kernel HIP events on every 4th step of the timed region (on every step they cost 0.15 ms of a 3.7 ms
    # step: 34 event records); PAIRCOUNT_BENCH_NO_EVENTS=1 switches them off to measure just that
    psa.profile_enable(0 if os.environ.get("PAIRCOUNT_BENCH_NO_EVENTS") else args.event_period)
    if collective:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if collective:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    psa.profile_enable(False)
    prof = psa.profile_read()
    # the collective by itself: the copy of the count and the 8-byte sum, K times, on an otherwise idle device
    all_reduce_ms = None
    if collective:
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            total.copy_(st.result[0:1])
            dist.all_reduce(total)
        torch.cuda.synchronize()
        all_reduce_ms = (time.perf_counter() - t1) / args.steps * 1e3

    pairs_local = st.pairs()
    st.count_tests(x_lo, x_hi)
    tests_local = st.tests()

    def phase_ms(prefix_is_count):
        return sum(ms / timed * (launches / args.steps) for nm, (timed, launches, ms) in prof.items()
                   if timed and nm.startswith("pc_c") == prefix_is_count)
    # one row per rank: what it held and how long its steps took (imbalance shows here, not in the max)
    mine = torch.tensor([rank, x_lo, x_hi, n, tests_local, pairs_local, elapsed / args.steps * 1e3,
                         phase_ms(False), phase_ms(True)], dtype=torch.float64, device=red_device)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    if collective:
        dist.all_gather(rows, mine)
    else:
        rows = [mine]
    ranks = [{"rank": int(r[0]), "x_lo": int(r[1]), "x_hi": int(r[2]), "points_held": int(r[3]), "tests": int(r[4]),
              "pairs": int(r[5]), "ms_per_step": float(r[6]), "binning_ms": float(r[7]), "pair_test_ms": float(r[8])}
             for r in (t.tolist() for t in rows)]
    pairs = sum(r["pairs"] for r in ranks)
    tests = sum(r["tests"] for r in ranks)
    points_held = sum(r["points_held"] for r in ranks)
    if collective:
        assert int(total.item()) == pairs, "all-reduced count differs from the sum of rank counts"
    ms_per_step = max(r["ms_per_step"] for r in ranks)
    elapsed = ms_per_step * 1e-3 * args.steps

    if rank == 0:
        # ---- per-kernel timings (HIP events on the launch stream, inside the timed region) ----
        kernels = {}
        for name, (timed, launches, ms) in prof.items():
            if timed:
                kernels[name] = {"avg_ms": ms / timed, "launches_per_step": launches / args.steps, "timed_launches": timed}
        if not kernels:            # events switched off (PAIRCOUNT_BENCH_NO_EVENTS: what the events themselves cost)
            print(json.dumps({"ms_per_step": ms_per_step, "events": False, "pairs": pairs, "tests": tests}))
            if collective:
                dist.destroy_process_group()
            return
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches_per_step"])
        layers = x_hi - x_lo
        for name, k in kernels.items():
            b = kernel_bytes(name, n, layers)
            if b:
                k["algorithmic_gb"] = b / 1e9
                k["achieved_gbs"] = b / 1e9 / (k["avg_ms"] * 1e-3)
        bin_ms = sum(k["avg_ms"] * k["launches_per_step"] for nm, k in kernels.items() if not nm.startswith("pc_c"))
        cnt_ms = sum(k["avg_ms"] * k["launches_per_step"] for nm, k in kernels.items() if nm.startswith("pc_c"))
        dk = kernels[dom]
        # HBM traffic per launch of the dominant kernel: not measurable from inside this process; taken from
        # the committed rocprofv3 --pmc passes of the same workload (profiles/*traffic_<workload>.json,
        # produced by tools/pmc_to_traffic.py from the request counters, see there), else null
        traffic = None
        if world == 1 and not args.points:
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic_%s.json" % args.workload)))
            if files:
                with open(files[-1]) as f:
                    prof_file = json.load(f)
                tk = prof_file.get("kernels", {}).get(dom)
                dk["traffic_source"] = os.path.relpath(files[-1], ROOT)
                # measured on these very kernels?  (a profile of older device sources says nothing about this build)
                if prof_file.get("device_source_hash") != ht.device_source_hash():
                    dk["traffic_stale"] = "device sources changed since %s was measured" % dk["traffic_source"]
                elif tk:
                    traffic = tk["hbm_bytes"] / 1e9
        # The dominant kernel against the roofline that bounds it.  A pair-test kernel is priced both ways --
        # 9 unfused binary32 operations per reference test against the vector peak without FMA credit, and its
        # compulsory bytes (12 N + 4 C) against HBM -- and the larger fraction names the bound.
        hbm_frac = (dk.get("achieved_gbs") or 0.0) / HBM_PEAK_GBS
        roofline = {"kernel": dom, "bound": "hbm", "achieved": dk.get("achieved_gbs"), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": hbm_frac}
        if dom.startswith("pc_c"):
            valu_tops = 9 * tests_local / (dk["avg_ms"] * 1e-3) / 1e12
            roofline["hbm_leg"] = {"achieved_gbs": dk.get("achieved_gbs"), "frac": hbm_frac}
            roofline["valu_leg"] = {"achieved_tops": valu_tops, "peak_tops": VALU_NOFMA_TOPS, "frac": valu_tops / VALU_NOFMA_TOPS,
                                    "ops_per_test": 9}
            # beside the nominal peak (one wave64 instruction per 2 cycles and SIMD at 2.4 GHz): what the same
            # chip was measured to issue, and the distance arithmetic of the test by itself
            ceiling = measured_valu_ceiling()
            if ceiling:
                roofline["valu_leg"]["measured"] = dict(ceiling, frac_of_plain_ops=valu_tops / ceiling["plain_ops_tops"],
                                                        frac_of_pair_test_arithmetic=valu_tops / ceiling["pair_test_arithmetic_tops"])
            if valu_tops / VALU_NOFMA_TOPS > hbm_frac:
                roofline.update({"bound": "valu", "achieved": valu_tops, "peak": VALU_NOFMA_TOPS, "unit": "TFLOP/s",
                                 "frac": valu_tops / VALU_NOFMA_TOPS})
        roofline.update({"traffic": traffic, "traffic_unit": "GB per launch (PMC, committed profile)",
                         "algorithmic_gb": dk.get("algorithmic_gb"), "avg_launch_ms": dk["avg_ms"]})
        binning_alg_gb = (44 * n + 8 * (layers + 1) * 403 * 403) / 1e9
        out = {
            "metric": "pairs-tested/s (reference dist2 evaluations covered per second, whole hot path: keys+binning+pair test)",
            "value": tests / (elapsed / args.steps),
            "unit": "pairs-tested/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            **({"preview_of_rank": args.as_rank} if args.as_rank and world == 1 else {}),
            "ms_per_step": ms_per_step, "all_reduce_ms": all_reduce_ms, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "points": total_points, "points_held_incl_halo": points_held,
                       "cutoff": 0.05, "slabs": slabs, "slab_cuts": cuts, "seed": args.seed,
                       "generator": args.generate, "generator_s": round(t_gen, 2), "backend": args.backend if collective else None},
            "pairs": pairs, "tests": tests,
            "pairs_tested_per_s_pair_test_kernel_only": (tests_local / (cnt_ms * 1e-3)) if cnt_ms else None,
            "binning_ms": bin_ms, "pair_test_ms": cnt_ms,
            "binning_roofline": {"bound": "hbm", "algorithmic_gb": binning_alg_gb,
                                 "achieved": binning_alg_gb / (bin_ms * 1e-3) if bin_ms else None, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s",
                                 "frac": binning_alg_gb / (bin_ms * 1e-3) / HBM_PEAK_GBS if bin_ms else None,
                                 "note": "all binning kernels of a step together against 44 N + 8 C bytes"},
            "pair_test_valu_frac": (9 * tests_local / (cnt_ms * 1e-3) / 1e12 / VALU_NOFMA_TOPS) if cnt_ms else None,
            "ranks": ranks,
            "kernels": kernels,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and not args.as_rank:
            out["cpu_baseline"] = cpu_baseline(psa, ht, st, torch, kind, args.seed, total_points, sigma, device)
            if not args.no_e2e_full and not args.points and args.workload in ("uniform100m", "blobs100m"):
                del xyz
                out["cpu_baseline"]["end_to_end_full"] = end_to_end_full(ht, kind, args.seed, total_points, sigma, pairs)
        print(json.dumps(out))
    if collective:
        dist.destroy_process_group()


def end_to_end_full(ht, kind, seed, total_points, sigma, hip_pairs):
    """The whole workload as a text file through both programs, process start to count printed: the
    reference binary once (it takes a quarter of a minute at 100 M points) and the product's host program
    twice (best run).  Both counts are compared with the count of the timed steps."""
    cores = ht.host_threads()
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "main_timer")
    if not (os.path.exists(ref_bin) and os.path.exists(ht.MAIN_PATH)):
        return {"error": "oracle/_ref/main_timer or the host program is not built"}
    tmp = tempfile.mkdtemp(prefix="pcbench_full_", dir=os.environ.get("PAIRCOUNT_TMP") or None)
    path = os.path.join(tmp, "full.xyz")
    try:
        t0 = time.time()
        npts = ht.generate_text(path, kind, seed, total_points, sigma)
        t_write = time.time() - t0
        out = {"points": npts, "file_bytes": os.path.getsize(path), "threads": cores, "file_written_s": round(t_write, 2)}
        best = None
        for _ in range(2):
            t1 = time.time()
            g = subprocess.run([ht.MAIN_PATH, "-f", path, "-p", str(cores)], capture_output=True, text=True,
                               timeout=600, env=dict(os.environ, PAIRCOUNT_TIMER="1"))
            w = time.time() - t1
            if g.returncode != 0:
                return dict(out, error="host program failed rc=%d: %s" % (g.returncode, g.stderr[-200:]))
            if best is None or w < best[0]:
                best = (w, g)
        gph = {m.group(1): float(m.group(2)) / 1e6 for m in re.finditer(r"(\w[\w()]*):\s+(\d+) us", best[1].stderr)}
        out.update({"gpu_host_program_wall_s": best[0], "gpu_phases_s": gph, "gpu_count": int(best[1].stdout.strip())})
        t1 = time.time()
        r = subprocess.run([ref_bin, "-f", path, "-p", str(cores), "-c", str(cores)], capture_output=True, text=True, timeout=900)
        wall = time.time() - t1
        if r.returncode != 0:
            return dict(out, error="reference binary failed rc=%d: %s" % (r.returncode, r.stderr[-200:]))
        phases = {m.group(1): int(m.group(2)) / 1e6 for m in re.finditer(r"(\w+):\s+(\d+) us", r.stdout)}
        count = int([ln for ln in r.stdout.splitlines() if ln.strip().isdigit()][-1])
        out.update({"reference_wall_s": wall, "reference_phases_s": phases, "reference_count": count,
                    "counts_equal": count == out["gpu_count"] == hip_pairs,
                    "speedup": wall / best[0],
                    "note": "process start to count printed on the same file; reference one run, host program best of two"})
        return out
    finally:
        try:
            os.remove(path)
            os.rmdir(tmp)
        except OSError:
            pass


def cpu_baseline(psa, ht, st, torch, kind, seed, total_points, sigma, device):
    """The reference binary (oracle/_ref, built from /root/reference in the build
    container) on a bounded sample of the same workload: an x-slab of the stream
    holding about 10 M points, same density.  Its count is compared with the HIP
    count on the identical points.  Falls back to the C restatement (oracle/)
    when the reference build is absent."""
    import numpy as np
    cores = ht.host_threads()
    want = min(total_points, 10_000_000)
    layers = max(1, min(400, round(400 * want / total_points)))
    lo = 200 - layers // 2
    hi = lo + layers - 1                       # inclusive cell range of the sample
    tmp = tempfile.mkdtemp(prefix="pcbench_")
    path = os.path.join(tmp, "sample.xyz")
    try:
        npts = ht.generate_text(path, kind, seed, total_points, sigma, x_lo=lo, x_hi=hi)
        sample = ht.generate(kind, seed, total_points, sigma, x_lo=lo, x_hi=hi)
        assert sample.shape[0] == npts
        t = torch.from_numpy(sample).to(device)
        st2 = st if npts <= st.capacity else psa.DeviceStages(npts, device)
        st2.count_pairs(t)
        st2.count_tests()
        gpu_pairs, tests = st2.pairs(), st2.tests()
        sample_desc = "x-slab cells [%d,%d] of the workload stream: %d points, %d tests" % (lo, hi, npts, tests)
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "main_timer")
        if os.path.exists(ref_bin):
            # both programs: best of two runs (the first touch of a freshly written file is slow for either)
            wall, r = None, None
            for _ in range(2):
                t0 = time.time()
                rr = subprocess.run([ref_bin, "-f", path, "-p", str(cores), "-c", str(cores)],
                                    capture_output=True, text=True, timeout=900)
                w = time.time() - t0
                if rr.returncode != 0:
                    return {"error": "reference binary failed rc=%d: %s" % (rr.returncode, rr.stderr[-200:])}
                if wall is None or w < wall:
                    wall, r = w, rr
            phases = {m.group(1): int(m.group(2)) / 1e6 for m in re.finditer(r"(\w+):\s+(\d+) us", r.stdout)}
            count = int([ln for ln in r.stdout.splitlines() if ln.strip().isdigit()][-1])
            # the same file through the product's C host program (tokenize on the host, bin+count on the GPU)
            e2e = None
            main_bin = ht.MAIN_PATH
            if os.path.exists(main_bin):
                gwall, g = None, None
                for _ in range(2):
                    t1 = time.time()
                    gg = subprocess.run([main_bin, "-f", path, "-p", str(cores)], capture_output=True, text=True,
                                        timeout=900, env=dict(os.environ, PAIRCOUNT_TIMER="1"))
                    w = time.time() - t1
                    if gwall is None or w < gwall or g.returncode != 0:
                        gwall, g = w, gg
                if g.returncode == 0:
                    gph = {m.group(1): float(m.group(2)) / 1e6 for m in re.finditer(r"(\w[\w()]*):\s+(\d+) us", g.stderr)}
                    e2e = {"file_bytes": os.path.getsize(path), "gpu_host_program_wall_s": gwall, "gpu_phases_s": gph,
                           "gpu_count": int(g.stdout.strip()), "reference_wall_s": wall,
                           "note": "process start to count printed, same file, best of two runs each; the GPU program's "
                                   "wall time includes HIP start-up and a pageable-memory PCIe copy of the parsed points"}
            return {"end_to_end": e2e, "value": tests / phases["count"], "unit": "pairs-tested/s", "cores": cores, "kind": "reference",
                    "sample": sample_desc, "phases_s": phases, "wall_s": wall, "pairs": count,
                    "matches_hip_count": count == gpu_pairs,
                    "value_whole_path": tests / (phases["parsing"] + phases["flatten"] + phases["count"]),
                    "note": "value = tests / reference 'count' phase; value_whole_path also charges its "
                            "parsing (tokenize+insert) and flatten phases"}
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        orc = oracle_lib.Oracle()
        t0 = time.time()
        pairs, tests2 = orc.count_pairs(sample, threads=cores)
        wall = time.time() - t0
        return {"value": tests / wall, "unit": "pairs-tested/s", "cores": cores, "kind": "port", "sample": sample_desc,
                "wall_s": wall, "pairs": pairs, "matches_hip_count": pairs == gpu_pairs}
    finally:
        try:
            os.remove(path)
            os.rmdir(tmp)
        except OSError:
            pass


if __name__ == "__main__":
    main()
