"""List-id sharding across the GPUs of one node (BASELINE configs[3]; reference: IndexShards over
sub-indexes that share the coarse quantizer, Auncel/IndexShards.cpp:261-311).

One process per GPU.  Rank r owns the inverted lists with owner[l] == r (balanced by bytes); every
rank quantizes the whole query batch (the 2 MB centroid table is replicated), scans only the probed
lists it owns -- non-owned lists have size 0 and are skipped exactly like empty lists in the reference
(Auncel/IndexIVF.cpp:450-455) -- and the per-rank (D, I) tables (n x k x 12 B each) are gathered on
rank 0 and merged on the host with merge_tables semantics.  No data-path collective is needed."""
import numpy as np


def assign_owners(list_sizes, nshard):
    """greedy longest-processing-time balance of list bytes; deterministic"""
    sizes = np.asarray(list_sizes, dtype=np.int64)
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(nshard, dtype=np.int64)
    owner = np.empty(len(sizes), dtype=np.int64)
    for l in order:
        r = int(np.argmin(load))
        owner[l] = r
        load[r] += sizes[l]
    return owner


def local_assignment(assign, owner, rank):
    """per-vector list number with every vector of a non-owned list dropped (-1)"""
    assign = np.asarray(assign, dtype=np.int64)
    keep = (assign >= 0) & (owner[np.clip(assign, 0, len(owner) - 1)] == rank)
    return np.where(keep, assign, -1)


def gather_and_merge(local_D, local_I, metric, merge_fn, dist=None, dst=0):
    """gather the per-rank result tables on `dst` and k-way merge them there.

    merge_fn(metric, all_D[nshard,n,k], all_I[nshard,n,k]) -> (D, I); returns None on other ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return merge_fn(metric, local_D[None], local_I[None])
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    tD = torch.from_numpy(np.ascontiguousarray(local_D)).to(dev)
    tI = torch.from_numpy(np.ascontiguousarray(local_I)).to(dev)
    if rank == dst:
        gD = [torch.empty_like(tD) for _ in range(world)]
        gI = [torch.empty_like(tI) for _ in range(world)]
    else:
        gD = gI = None
    dist.gather(tD, gD, dst=dst)
    dist.gather(tI, gI, dst=dst)
    if rank != dst:
        return None
    all_D = np.stack([t.cpu().numpy() for t in gD])
    all_I = np.stack([t.cpu().numpy() for t in gI])
    return merge_fn(metric, all_D, all_I)


def allgather_rows(local, counts, dist=None):
    """rank r holds rows [r n / N, (r + 1) n / N) of a table (the coarse ranking of its share of the batch, SURVEY 8e: "coarse
    quantization is computed once"); every rank gets the whole table.  n x nprobe x 8 B: 2.6 MB at n = 10 000, nprobe = 32.
    RCCL all-gather on the GPUs (backend nccl), gloo in the CPU rehearsal.  counts[r] = rows of rank r (they differ by at most
    one: shares are padded to the longest)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    m = max(counts)
    if t.shape[0] < m:
        t = torch.cat([t, torch.zeros((m - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)])
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)])
