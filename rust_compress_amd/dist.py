"""Multi-GPU sharding of a batch of independent blocks.

Blocks never talk to each other, so the compute needs no collective: rank g gets the contiguous range of
block indices [b_g, b_{g+1}) balanced by output bytes.  A collective appears only when the caller's data
lives on ONE rank: `scatter_blocks` / `gather_blocks` move variable-length byte ranges root -> peers and
peers -> root with grouped point-to-point transfers (RCCL over xGMI on GPUs: the root drives all seven
links at once, which a ring collective would not; gloo on CPU in the tests).
One process per GPU, `torch.distributed` initialised by the caller.
"""
import numpy as np


def partition(weights, world):
    """Contiguous ranges balanced by `weights` (e.g. decoded bytes per block).
    -> int64 array `bounds` of length world+1 with bounds[0] = 0, bounds[-1] = len(weights)."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if n == 0:
        return np.zeros(world + 1, dtype=np.int64)
    csum = np.concatenate([[0.0], np.cumsum(w)])
    total = csum[-1]
    bounds = np.zeros(world + 1, dtype=np.int64)
    for g in range(1, world):
        target = total * g / world
        bounds[g] = int(np.searchsorted(csum, target, side="left"))
    bounds[world] = n
    return np.maximum.accumulate(np.minimum(bounds, n))


def _sizes_for(bounds, off, lens):
    """byte span [lo, hi) of each rank's block range in the packed buffer"""
    spans = []
    for g in range(len(bounds) - 1):
        a, b = int(bounds[g]), int(bounds[g + 1])
        if a == b:
            spans.append((0, 0))
        else:
            spans.append((int(off[a]), int(off[b - 1] + lens[b - 1])))
    return spans


def scatter_blocks(base, off, lens, bounds, root=0, device=None):
    """Root holds (base uint8 tensor, off, lens numpy); every rank returns (local_base tensor, local_off, local_len).
    Descriptors travel with broadcast_object_list (tiny); payload bytes with grouped isend/irecv."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = [None]
    if rank == root:
        spans = _sizes_for(bounds, off, lens)
        meta = [(np.asarray(bounds), np.asarray(off), np.asarray(lens), spans)]
    dist.broadcast_object_list(meta, src=root)
    bounds, off, lens, spans = meta[0]
    lo, hi = spans[rank]
    a, b = int(bounds[rank]), int(bounds[rank + 1])
    dev = device if device is not None else (base.device if base is not None else "cpu")
    if rank == root:
        local = base[lo:hi].clone()
        reqs = []
        for g in range(world):
            if g == root or spans[g][1] == spans[g][0]:
                continue
            reqs.append(dist.isend(base[spans[g][0]:spans[g][1]].contiguous(), dst=g))
        for r in reqs:
            r.wait()
    else:
        local = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
        if hi > lo:
            dist.recv(local, src=root)
    return local, (np.asarray(off[a:b]) - lo).astype(np.uint64), np.asarray(lens[a:b]).astype(np.uint64)


def gather_blocks(local_out, local_off, local_len, bounds, root=0):
    """Inverse of scatter for the outputs: root returns (list of per-block bytes-like tensors in global order)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    # compact local outputs (offsets may have gaps)
    parts = [local_out[int(o):int(o) + int(l)] for o, l in zip(local_off, local_len)]
    packed = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=local_out.device)
    lens_all = [None] * world
    dist.all_gather_object(lens_all, [int(l) for l in local_len])
    if rank == root:
        out = []
        for g in range(world):
            tot = sum(lens_all[g])
            if g == root:
                buf = packed
            else:
                buf = torch.empty(tot, dtype=torch.uint8, device=local_out.device)
                if tot:
                    dist.recv(buf, src=g)
            p = 0
            for l in lens_all[g]:
                out.append(buf[p:p + l])
                p += l
        return out
    if packed.numel():
        dist.send(packed, dst=root)
    return None
