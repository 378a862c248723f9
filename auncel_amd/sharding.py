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


def gather_tables(local_D, local_I, dist=None, dst=0):
    """the per-rank (D, I) tables stacked on `dst` ([nshard, n, k] each; copies, the callers' buffers are free again); None elsewhere"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.array(local_D, copy=True)[None], np.array(local_I, copy=True)[None]
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    tD = torch.from_numpy(np.ascontiguousarray(local_D)).to(dev)
    tI = torch.from_numpy(np.ascontiguousarray(local_I)).to(dev)
    gD = [torch.empty_like(tD) for _ in range(world)] if rank == dst else None
    gI = [torch.empty_like(tI) for _ in range(world)] if rank == dst else None
    dist.gather(tD, gD, dst=dst)
    dist.gather(tI, gI, dst=dst)
    if rank != dst:
        return None
    return np.stack([t.cpu().numpy() for t in gD]), np.stack([t.cpu().numpy() for t in gI])


def run_pipelined(h, metric, merge_fn, nq, k, nprobe, counts, rank, dist, nsteps, lag=3, coarse_ahead=2, dst=0):
    """`nsteps` sharded searches of the whole resident batch with `lag` of them in flight (IndexShards::search over list-id shards,
    Auncel/IndexShards.cpp:261-311, one step after the other in the reference; ThreadedIndex keeps the shards busy, not the steps).

    A step = coarse ranking of this rank's share of the batch -> all-gather of the key rows (the path's one exchange) -> search of
    every query over the lists this rank owns -> gather of the (D, I) tables on `dst` -> merge_tables there.  The searches and the
    coarse rankings are tickets of the engine's asynchronous entry points (amd_ivf_submit_coarse_resident /
    amd_ivf_submit_search_resident_preassigned: they run on the handle's internal contexts, set_async_depth of them at a time); ONE
    thread per rank issues both collectives in a fixed order -- keys of step i, then tables of step i - lag -- so every rank issues
    them in the same order without any agreement at run time; the merge runs on a thread of its own on `dst` (C++,
    amd_ivf_merge_tables) while the next steps' scans are on the GPU.

    Returns (merged (D, I) of the last step on `dst` else None, dict of per-rank sums: coarse_ms, exchange_ms, scan_ms, select_ms,
    scan_launches, scan_min_bytes, merge_ms, steps_merged)."""
    import queue
    import threading
    import time
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    q0 = int(sum(counts[:rank]))
    ring = lag + 4
    bufs = [(np.empty((nq, k), np.float32), np.empty((nq, k), np.int64)) for _ in range(ring)]
    acc = {"coarse_ms": 0.0, "exchange_ms": 0.0, "scan_ms": 0.0, "select_ms": 0.0, "scan_launches": 0.0, "scan_min_bytes": 0.0,
           "merge_ms": 0.0, "steps_merged": 0}
    coarse_t, search_t, last = {}, {}, {}
    merge_q = queue.Queue(maxsize=2)
    errs = []

    def merger():
        while True:
            item = merge_q.get()
            if item is None:
                return
            try:
                t0 = time.perf_counter()
                last["out"] = merge_fn(metric, item[0], item[1])
                acc["merge_ms"] += (time.perf_counter() - t0) * 1e3
                acc["steps_merged"] += 1
            except Exception as e:  # noqa: BLE001
                errs.append(e)

    mt = None
    if rank == dst:
        mt = threading.Thread(target=merger, daemon=True)
        mt.start()

    def submit_coarse(i):
        coarse_t[i] = h.submit_coarse_resident(q0, counts[rank], nprobe, mode=0)

    for i in range(min(coarse_ahead, nsteps)):
        submit_coarse(i)
    for i in range(nsteps + lag):
        if i < nsteps:
            _, ck, tm, _ = h.wait(coarse_t.pop(i))
            acc["coarse_ms"] += tm["coarse_ms"]
            if i + coarse_ahead < nsteps:
                submit_coarse(i + coarse_ahead)
            tx = time.perf_counter()
            keys = allgather_rows(ck, counts, dist if world > 1 else None)
            acc["exchange_ms"] += (time.perf_counter() - tx) * 1e3
            search_t[i] = h.submit_search_resident_preassigned(0, nq, k, keys, out=bufs[i % ring])
        j = i - lag
        if j >= 0:
            D, I, tm, _ = h.wait(search_t.pop(j))
            for key in ("scan_ms", "select_ms", "scan_launches", "scan_min_bytes"):
                acc[key] += tm[key]
            tx = time.perf_counter()
            tabs = gather_tables(D, I, dist if world > 1 else None, dst)
            acc["exchange_ms"] += (time.perf_counter() - tx) * 1e3
            if rank == dst:
                merge_q.put(tabs)
    if mt is not None:
        merge_q.put(None)
        mt.join()
    if errs:
        raise errs[0]
    return last.get("out"), acc
