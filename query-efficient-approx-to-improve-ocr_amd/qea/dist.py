"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo"
for the CPU rehearsal in tests).  The minibatch is sharded along N, BatchNorm statistics stay local
(no SyncBN), and each optimiser step exchanges ONE flat fp32 gradient buffer:
CRNN grads (35.0 MB) after Phase A, UNet grads (31.1 MB) after Phase B — two all-reduces per step,
because the CRNN update sits between the phases."""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(device=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); no-op for 1 process."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or (dist.is_available() and dist.is_initialized()):
        return world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on these hosts
    # QEA_DIST_BACKEND=gloo: rehearsal of the N-rank path on fewer GPUs than ranks (several ranks sharing one card, which RCCL
    # refuses); gloo all-reduces / broadcasts CUDA tensors through the host, the CER all-gather already runs on CPU tensors then
    backend = os.environ.get("QEA_DIST_BACKEND", "nccl" if (device is not None and device.type == "cuda") else "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return world()


def allreduce_mean_(flat):
    """In-place average of one flat gradient buffer over the ranks (sum all-reduce, then 1/world)."""
    w = world()
    if w > 1:
        dist.all_reduce(flat)
        flat.mul_(1.0 / w)
    return flat


def allreduce_module_grads(module):
    """Average a module's gradients: one collective on the flat buffer when the module is flat
    (qea/params.py), else one per parameter."""
    if world() == 1:
        return
    fs = module.__dict__.get("_qea_flat_state")
    if fs is not None and fs.intact():
        fs.attach_grads()
        allreduce_mean_(fs.grad)
        return
    for p in module.parameters():
        if p.grad is None:                                      # a rank without work this step still joins every collective
            p.grad = torch.zeros_like(p)
        allreduce_mean_(p.grad)


def global_topk(local_cers, k_global, with_counts=False):
    """TopKCER over the WHOLE minibatch when it is sharded: all-gather the per-shard CERs,
    rank them in the global stable-descending order (rank-major index as the tie-break), and return
    (local indices of the winners that live in this shard, k) with k = the number of strips actually picked
    = min(k_global, minibatch rows) on EVERY path (a minibatch with fewer rows than k_global picks them all).
    The caller weights its loss by len(local)/k so that averaged gradients equal the global-batch mean.
    with_counts: also return how many winners live on each rank (the input of rebalance_rows); k == sum(counts)."""
    w, r = world(), rank()
    vals = torch.as_tensor(local_cers, dtype=torch.float32)
    if w == 1:
        order = torch.argsort(-vals, stable=True)[:k_global]
        return (order, order.numel(), [order.numel()]) if with_counts else (order, order.numel())
    n_local = torch.tensor([vals.numel()], dtype=torch.int64)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(w)]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    sizes = [s.to(dev) for s in sizes]
    dist.all_gather(sizes, n_local.to(dev))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    padded = torch.full((mx,), float("-inf"), device=dev)
    padded[:vals.numel()] = vals.to(dev)
    gathered = [torch.empty(mx, device=dev) for _ in range(w)]
    dist.all_gather(gathered, padded)
    allv = torch.cat([g[:s].cpu() for g, s in zip(gathered, sizes)])
    order = torch.argsort(-allv, stable=True)[:k_global]
    start = sum(sizes[:r])
    mine = order[(order >= start) & (order < start + sizes[r])] - start
    if not with_counts:
        return mine, order.numel()
    bounds = torch.tensor([0] + sizes).cumsum(0)
    counts = [int(((order >= bounds[i]) & (order < bounds[i + 1])).sum()) for i in range(w)]
    return mine, order.numel(), counts


def balanced_slice(k, w, r):
    """rows [lo, hi) of a k-row list that rank r of w takes: sizes differ by at most one"""
    return (k * r) // w, (k * (r + 1)) // w


def rebalance_rows(rows, counts):
    """Global-TopK load balance ("re-balances them with an all-to-all of <= 410 x 16 KB images"): rank i holds
    counts[i] winner rows (worst case: all of them on one rank, which would then run the whole of Phase A while the others
    wait in the all-reduce).  The winners, listed rank-major, are dealt out again in equal slices: every rank writes its rows
    into its range of one zero-filled [k, ...] buffer, ONE all-reduce (sum with zeros: exact) makes the list whole everywhere
    — 6.7 MB at k = 410, against the 35 MB gradient exchange that follows — and rank r keeps rows balanced_slice(k, w, r)."""
    w, r = world(), rank()
    if w == 1:
        return rows
    k = sum(counts)
    if rows.shape[0] != counts[r]:
        raise ValueError(f"rank {r} holds {rows.shape[0]} winner rows, the plan says {counts[r]}")
    buf = rows.new_zeros((k,) + tuple(rows.shape[1:]))
    off = sum(counts[:r])
    buf[off:off + counts[r]].copy_(rows)
    dist.all_reduce(buf)
    lo, hi = balanced_slice(k, w, r)
    return buf[lo:hi].contiguous()


def equal_shards(indices, per_step):
    """Shard a (rank-identical) index list so that EVERY rank gets the same number of optimiser steps: the list is TRUNCATED to a
    multiple of world * per_step (the tail is dropped, nothing wraps around — like the reference's drop_last loader,
    train_nn_area.py:131-133) and step i of rank r takes the r-th group of `per_step` indices of the i-th global batch.
    Ranks that ran different step counts would leave the others blocked in their next all-reduce."""
    w, r = world(), rank()
    if w == 1:
        return indices
    idx = torch.as_tensor(indices)
    steps = idx.numel() // (w * per_step)
    return idx[: steps * w * per_step].view(steps, w, per_step)[:, r].reshape(-1)


def deal_batches(batches):
    """Round-robin deal of a (rank-identical) list of batches, cut to a multiple of the world size."""
    w, r = world(), rank()
    if w == 1:
        return batches
    n = len(batches) // w * w
    return batches[:n][r::w]
