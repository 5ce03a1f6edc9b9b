"""GPU: pc_count_pairs_noalloc (workspace inside the library, no allocation) against
pc_count_pairs_device (workspace passed in) on the same inputs, through the C ABI.
Two calls in a row share the built-in workspace, the second on a smaller input:
count, status and the cell table must be equal, the sorted array equal cell by cell
(the order inside a cell is unspecified for both entry points)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NBB = 403
TABLE_LEN = NBB ** 3 + 2


def _canonical(sorted_xyz, table, n):
    """(n,3) float32 in table order -> rows ordered by (cell, x bits, y bits, z bits)."""
    counts = np.diff(table[: NBB ** 3 + 1].astype(np.int64))
    cell = np.repeat(np.arange(NBB ** 3, dtype=np.int64), counts)
    assert len(cell) == n
    bits = sorted_xyz[:n].view(np.uint32)
    order = np.lexsort((bits[:, 2], bits[:, 1], bits[:, 0], cell))
    return bits[order]


def _run(psa, torch, entry, xyz, x_lo, x_hi, bufs):
    L = psa.lib()
    n = xyz.shape[0]
    bufs["sorted"].zero_()
    bufs["result"].zero_()
    bufs["status"].zero_()
    s = torch.cuda.current_stream().cuda_stream
    vp = ctypes.c_void_p
    if entry == "noalloc":
        rc = L.pc_count_pairs_noalloc(vp(xyz.data_ptr()), n, x_lo, x_hi, vp(bufs["sorted"].data_ptr()),
                                      vp(bufs["table"].data_ptr()), vp(bufs["result"].data_ptr()),
                                      vp(bufs["status"].data_ptr()), vp(s))
    else:
        rc = L.pc_count_pairs_device(vp(xyz.data_ptr()), n, x_lo, x_hi, vp(bufs["sorted"].data_ptr()),
                                     vp(bufs["table"].data_ptr()), vp(bufs["ws"].data_ptr()), bufs["ws"].numel(),
                                     vp(bufs["result"].data_ptr()), vp(bufs["status"].data_ptr()), vp(s))
    assert rc == 0, L.pc_last_error()
    torch.cuda.synchronize()
    return (int(bufs["result"][0].item()), int(bufs["status"].item()),
            bufs["table"].cpu().numpy().view(np.uint32).copy(), bufs["sorted"].cpu().numpy().copy())


def test_noalloc_entry_point_equals_workspace_entry_point(psa, oracle):
    import torch
    assert torch.cuda.is_available()
    L = psa.lib()
    rng = np.random.default_rng(2024)
    big = np.concatenate([rng.uniform(-9.999, 9.999, (700_000, 3)),
                          rng.normal(0.0, 0.15, (200_000, 3)).clip(-9.999, 9.999)]).astype(np.float32)
    small = np.concatenate([rng.uniform(-3.0, 3.0, (40_000, 3)),
                            rng.uniform(1.0, 1.04, (900, 3))]).astype(np.float32)      # sparse + one crowded cell
    cap = len(big)
    bufs = {
        "sorted": torch.empty((cap, 3), dtype=torch.float32, device="cuda:0"),
        "table": torch.empty(TABLE_LEN, dtype=torch.int32, device="cuda:0"),
        "ws": torch.empty(L.pc_workspace_bytes(cap), dtype=torch.uint8, device="cuda:0"),
        "result": torch.zeros(2, dtype=torch.int64, device="cuda:0"),
        "status": torch.zeros(1, dtype=torch.int32, device="cuda:0"),
    }
    # two calls in a row on the built-in workspace, the second smaller; then the reference entry point on each
    got = []
    for pts, (lo, hi) in ((big, (0, 400)), (small, (0, 400))):
        got.append(_run(psa, torch, "noalloc", torch.from_numpy(pts).to("cuda:0"), lo, hi, bufs))
    for (pts, (lo, hi)), g in zip(((big, (0, 400)), (small, (0, 400))), got):
        want = _run(psa, torch, "device", torch.from_numpy(pts).to("cuda:0"), lo, hi, bufs)
        n = len(pts)
        assert g[0] == want[0] and g[1] == want[1] == 0
        assert g[0] == oracle.count_pairs(pts)[0]
        assert np.array_equal(g[2][: NBB ** 3 + 1], want[2][: NBB ** 3 + 1])
        assert np.array_equal(_canonical(g[3], g[2], n), _canonical(want[3], want[2], n))
    # a slab call after the whole-grid calls (only its layers of the table are written: compare those)
    lo, hi = 180, 200
    keys = oracle.cell_keys(small)
    layer = keys // (NBB * NBB)
    sub = small[(layer >= lo) & (layer <= hi)]
    a = _run(psa, torch, "noalloc", torch.from_numpy(sub).to("cuda:0"), lo, hi, bufs)
    b = _run(psa, torch, "device", torch.from_numpy(sub).to("cuda:0"), lo, hi, bufs)
    assert a[0] == b[0] and a[1] == b[1] == 0
    sl = slice(lo * NBB * NBB, (hi + 1) * NBB * NBB + 1)
    assert np.array_equal(a[2][sl], b[2][sl])
    # sizes beyond the built-in workspace are refused with a clear error, nothing is launched
    vp = ctypes.c_void_p
    too_many = 1048576 + 1
    rc = L.pc_count_pairs_noalloc(vp(bufs["sorted"].data_ptr()), too_many, 0, 400, vp(bufs["sorted"].data_ptr()),
                                  vp(bufs["table"].data_ptr()), vp(bufs["result"].data_ptr()),
                                  vp(bufs["status"].data_ptr()), vp(0))
    assert rc == -1 and b"exceed" in L.pc_last_error()


def _hip():
    """The HIP runtime the library itself is linked against (already loaded by torch)."""
    return ctypes.CDLL("libamdhip64.so")


def test_noalloc_allocates_nothing_and_can_be_captured(psa, oracle):
    """After pc_noalloc_prepare(): free device memory is the same before and after a call, and a call recorded
    into a HIP graph (stream capture) gives the same count when the graph is launched -- twice, on new input in
    the same buffers."""
    import torch
    L, hip = psa.lib(), _hip()
    vp = ctypes.c_void_p
    rng = np.random.default_rng(77)
    pts = np.concatenate([rng.uniform(-9.999, 9.999, (300_000, 3)), rng.normal(2.0, 0.1, (100_000, 3))]).astype(np.float32)
    other = rng.normal(-3.0, 0.2, (400_000, 3)).astype(np.float32)
    want, want_other = oracle.count_pairs(pts)[0], oracle.count_pairs(other)[0]
    n = len(pts)
    xyz = torch.from_numpy(pts).to("cuda:0")
    srt = torch.empty((n, 3), dtype=torch.float32, device="cuda:0")
    table = torch.empty(TABLE_LEN, dtype=torch.int32, device="cuda:0")
    result = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    status = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    assert L.pc_noalloc_prepare() == 0, L.pc_last_error()
    L.pc_profile_enable(0)
    stream = torch.cuda.Stream()
    args = (vp(xyz.data_ptr()), n, 0, 400, vp(srt.data_ptr()), vp(table.data_ptr()), vp(result.data_ptr()),
            vp(status.data_ptr()), vp(stream.cuda_stream))

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    assert L.pc_count_pairs_noalloc(*args) == 0, L.pc_last_error()      # (warm: the code object's kernels have run once)
    stream.synchronize()
    before = free_bytes()
    assert L.pc_count_pairs_noalloc(*args) == 0, L.pc_last_error()
    stream.synchronize()
    assert free_bytes() == before
    assert int(result[0].item()) == want and int(status.item()) == 0

    # the same call under stream capture: nothing in it may be refused, and nothing runs until the graph is launched
    result.zero_()
    torch.cuda.synchronize()
    graph, graph_exec = vp(), vp()
    assert hip.hipStreamBeginCapture(vp(stream.cuda_stream), 0) == 0          # hipStreamCaptureModeGlobal
    rc = L.pc_count_pairs_noalloc(*args)
    end = hip.hipStreamEndCapture(vp(stream.cuda_stream), ctypes.byref(graph))
    assert rc == 0, L.pc_last_error()
    assert end == 0 and graph.value
    assert int(result[0].item()) == 0                                          # captured, not executed
    assert hip.hipGraphInstantiate(ctypes.byref(graph_exec), graph, None, None, 0) == 0
    try:
        assert hip.hipGraphLaunch(graph_exec, vp(stream.cuda_stream)) == 0
        stream.synchronize()
        assert int(result[0].item()) == want and int(status.item()) == 0
        xyz.copy_(torch.from_numpy(other))                                     # new points, same buffers, same graph
        torch.cuda.synchronize()
        assert hip.hipGraphLaunch(graph_exec, vp(stream.cuda_stream)) == 0
        stream.synchronize()
        assert int(result[0].item()) == want_other and int(status.item()) == 0
    finally:
        hip.hipGraphExecDestroy(graph_exec)
        hip.hipGraphDestroy(graph)
