"""x-slab sharding of the pair count across GPUs (one process per GPU).

Every neighbour offset the reference visits has dx in {0,+1} (reference
src/main.c:195-209), so the occupied x layers 0..399 can be cut into contiguous
slabs [lo, hi): the rank owning a slab counts every pair whose home (lower-key)
cell lies in it and only needs, besides its own points, read-only copies of
layer hi (a one-sided +x halo).  Slabs are independent; the only exchange is
one 8-byte sum, which replaces the reference's `cstate.npairs += npairs`
(src/main.c:231) across ranks.
"""
import numpy as np

LAYERS = 400


def equal_cuts(n_ranks, layers=LAYERS):
    """[c0..cN]: rank r owns home layers [c[r], c[r+1]); equal widths."""
    return [layers * r // n_ranks for r in range(n_ranks + 1)]


def balanced_cuts(layer_cost, n_ranks):
    """Contiguous cuts whose per-slab cost sums are as even as a greedy sweep
    over the prefix sum gets them; `layer_cost` has one entry per x layer (for
    clustered inputs use an estimate of the pair tests homed in the layer, e.g.
    the squared layer population).  Every slab gets at least one layer."""
    cost = np.asarray(layer_cost, dtype=np.float64)
    layers = len(cost)
    assert 1 <= n_ranks <= layers
    prefix = np.concatenate([[0.0], np.cumsum(cost)])
    total = prefix[-1]
    cuts = [0]
    for r in range(1, n_ranks):
        target = total * r / n_ranks
        c = int(np.searchsorted(prefix, target, side="left"))
        c = max(c, cuts[-1] + 1)                 # at least one layer per slab
        c = min(c, layers - (n_ranks - r))       # leave a layer for each later slab
        cuts.append(c)
    cuts.append(layers)
    return cuts


def halo_range(lo, hi, layers=LAYERS):
    """Inclusive x-cell range of the points a rank owning [lo,hi) must hold."""
    return lo, min(hi, layers - 1)


def reduce_count(local_count, device=None, group=None):
    """Sum of the ranks' counts: the path's only collective (8 bytes)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def sharded_count(points_of_rank, lo, hi, device, group=None):
    """Counts on this rank's GPU the pairs homed in layers [lo,hi) among
    `points_of_rank` (its slab plus halo layer, host array) and returns the
    all-reduced total.  Needs the HIP library and a GPU."""
    from . import PairCounter
    pc = PairCounter(device=device.index if hasattr(device, "index") else int(device),
                     expected_points=len(points_of_rank), slab=(lo, hi))
    try:
        pc.append(points_of_rank)
        local = pc.finish()
    finally:
        pc.close()
    return reduce_count(local, device=device, group=group)
