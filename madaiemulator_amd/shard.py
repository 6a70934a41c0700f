"""One-process-per-GPU sharding of the embarrassingly parallel axes.

The reference parallelises with a pthread pool over independent optimiser restarts and reduces with a
mutex-guarded arg-max (libEmu/estimate_threaded.c:172-188, 295-323); multi-output models are a serial loop over
independent PCA components (multivar_support.c:23-25).  Here every rank owns a contiguous/cyclic share of the
independent units, does all of its work with no data-path collective, and ONE all-gather at the end assembles
the result (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
"""
import os

import numpy as np


def world():
    """(rank, world_size, local_rank) from the torch.distributed.run environment"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def cyclic_share(n_units, rank, world_size):
    """unit u -> rank u mod world_size (component c -> GPU c mod 8; restart r -> GPU r mod 8)"""
    return list(range(rank, n_units, world_size))


def block_share(n_units, rank, world_size):
    """contiguous split of a query list: (start, stop)"""
    base, rem = divmod(n_units, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def all_gather_rows(local_rows, ncols, device=None):
    """Gather a variable number of fixed-width float64 rows from every rank (one padded all_gather).

    Returns a list (one entry per rank) of (n_r, ncols) numpy arrays, identical on all ranks."""
    import torch
    dist = _dist()
    local = np.ascontiguousarray(local_rows, dtype=np.float64).reshape(-1, ncols)
    if dist is None:
        return [local]
    ws = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    counts = torch.zeros(ws, dtype=torch.int64, device=dev)
    counts[dist.get_rank()] = local.shape[0]
    dist.all_reduce(counts)
    nmax = int(counts.max().item())
    buf = torch.zeros((nmax, ncols), dtype=torch.float64, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf)
    return [o[:int(c)].cpu().numpy() for o, c in zip(out, counts.tolist())]


def farm_evaluations(eval_fn, thetas, rank=None, world_size=None):
    """Axis 2: independent likelihood evaluations of a theta list, restart r -> rank r mod W.

    eval_fn(theta) -> -logL (float).  Returns the full vector of values (NaN where the evaluation failed), the same
    on every rank -- the batched entry point the reference exposes to R (libRbind/rbind.c:626-724)."""
    r0, w0, _ = world()
    rank = r0 if rank is None else rank
    world_size = w0 if world_size is None else world_size
    thetas = np.asarray(thetas, dtype=np.float64)
    mine = cyclic_share(len(thetas), rank, world_size)
    rows = np.array([[i, eval_fn(thetas[i])] for i in mine], dtype=np.float64).reshape(-1, 2)
    vals = np.full(len(thetas), np.nan)
    for part in all_gather_rows(rows, 2):
        for i, v in part:
            vals[int(i)] = v
    return vals


def farm_evaluations_batched(batch_fn, thetas, batch=16, rank=None, world_size=None):
    """As farm_evaluations, but a rank sends its share to the device in lock-step batches:
    batch_fn(theta_rows) -> array of -logL, one per row (Context.loglik_batch(rows)["value"]; NaN where the
    evaluation failed).  Same partition (restart r -> rank r mod W), same single all-gather."""
    r0, w0, _ = world()
    rank = r0 if rank is None else rank
    world_size = w0 if world_size is None else world_size
    thetas = np.asarray(thetas, dtype=np.float64)
    mine = cyclic_share(len(thetas), rank, world_size)
    rows = []
    for s0 in range(0, len(mine), max(1, batch)):
        idx = mine[s0:s0 + max(1, batch)]
        vals = np.asarray(batch_fn(thetas[idx]), dtype=np.float64).reshape(-1)
        rows += [[i, v] for i, v in zip(idx, vals)]
    rows = np.array(rows, dtype=np.float64).reshape(-1, 2)
    out = np.full(len(thetas), np.nan)
    for part in all_gather_rows(rows, 2):
        for i, v in part:
            out[int(i)] = v
    return out


def best_of(values, thetas):
    """the reference's arg-max under results_mutex (estimate_threaded.c:308-313): NaN/inf are skipped
    (maxmultimin.c:110); values are -logL so the best is the smallest"""
    values = np.asarray(values, dtype=np.float64)
    ok = np.isfinite(values)
    if not ok.any():
        return None, None
    i = int(np.argmin(np.where(ok, values, np.inf)))
    return i, np.asarray(thetas)[i]


def farm_components(component_fn, n_components, width, rank=None, world_size=None):
    """Axis 1: independent PCA components of a multi-output model, component c -> rank c mod W.

    component_fn(c) -> 1-D float64 array of length `width` (e.g. best thetas + likelihood).  Returns an
    (n_components, width) array, identical on all ranks; rank 0 would write the snapshot."""
    r0, w0, _ = world()
    rank = r0 if rank is None else rank
    world_size = w0 if world_size is None else world_size
    mine = cyclic_share(n_components, rank, world_size)
    rows = np.array([np.concatenate([[c], np.asarray(component_fn(c), dtype=np.float64)]) for c in mine],
                    dtype=np.float64).reshape(-1, width + 1)
    out = np.full((n_components, width), np.nan)
    for part in all_gather_rows(rows, width + 1):
        for row in part:
            out[int(row[0])] = row[1:]
    return out


def farm_queries(predict_fn, Xq, rank=None, world_size=None):
    """Axis 3: query points split in contiguous blocks; gather 2 doubles per query."""
    r0, w0, _ = world()
    rank = r0 if rank is None else rank
    world_size = w0 if world_size is None else world_size
    Xq = np.asarray(Xq, dtype=np.float64)
    a, b = block_share(len(Xq), rank, world_size)
    if b > a:
        m, v = predict_fn(Xq[a:b])
        rows = np.column_stack([m, v])
    else:
        rows = np.zeros((0, 2))
    parts = all_gather_rows(rows, 2)
    allr = np.vstack(parts) if parts else np.zeros((0, 2))
    return allr[:, 0].copy(), allr[:, 1].copy()
