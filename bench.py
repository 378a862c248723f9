#!/usr/bin/env python3
"""bench.py -- queries/sec of the Auncel error-bounded IVF-Flat search on MI355X.

Workload (BASELINE.json configs[1]): SIFT-10M-like (10M x 128 uint8-valued fp32 vectors), IVF4096,Flat,
max_topk = 100 heap, top-10 asked with error bound 0.05 (recall@10 >= 0.95), per-query adaptive
nprobe (Auncel ELP).  One "step" = Error_sys::search over the whole batch of 5000 resident test
queries (inputs already in HBM), exactly the call the reference's eval/bound.cpp times.

Multi-GPU (--gpus N, launched by torch.distributed.run): the adaptive rule needs the global top-k
after every probe, so it does not shard by list; every rank holds a replica of the
index and searches its own 5000 queries -- weak scaling, no data-path collective.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# spread of the synthetic blobs: with the reference's IVF training (25 k-means iterations) fixed-nprobe recall@10 reaches
# 0.95 at nprobe ~28 of 4096 (0.934 @ 16, 0.955 @ 32, 0.974 @ 64: profiles/r01_sigma_sweep.txt), as on real SIFT
SIGMA = 38.0


def host_cores():
    """hardware threads this process may actually use: the affinity mask, cut by a cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:  # noqa: BLE001 -- no cgroup v2 file: the mask is all there is
        pass
    return n


def kfd_gpu_count():
    """GPUs of this node by the kernel driver's topology (nodes with SIMDs), honouring ROCR / HIP_VISIBLE_DEVICES as a count;
    None where the topology is not readable.  No HIP call: the launcher parent must not initialise the GPU."""
    import glob
    n = 0
    try:
        paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
        if not paths:
            return None
        for p in paths:
            for line in open(p):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def gen_data(torch, dev, nb, nq, d, nblobs, sigma, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    centres = torch.rand((nblobs, d), generator=g, device=dev) * 160.0

    def draw(n, gg):
        out = torch.empty((n, d), device=dev, dtype=torch.float32)
        for i0 in range(0, n, 1 << 20):
            i1 = min(n, i0 + (1 << 20))
            c = torch.randint(0, nblobs, (i1 - i0,), generator=gg, device=dev)
            x = centres[c] + torch.randn((i1 - i0, d), generator=gg, device=dev) * sigma
            out[i0:i1] = torch.floor(torch.clamp(x, 0, 255))
        return out

    xb = draw(nb, g)
    return xb, centres, draw


def find_data_files(root):
    """--data DIR: base / query / ground-truth files of the reference's harness layouts (Auncel/eval/bound.cpp:29-113,225-340):
    *base*.fvecs|.fbin|.u8bin, *query*.fvecs|.fbin|.u8bin, *groundtruth*|*gt*.ivecs|.ibin"""
    names = sorted(os.listdir(root))

    def pick(words, exts):
        for n in names:
            low = n.lower()
            if any(w in low for w in words) and low.endswith(exts):
                return os.path.join(root, n)
        return None

    vec = (".fvecs", ".fbin", ".u8bin", ".bvecs.fbin")
    return pick(("base", "learn_as_base"), vec), pick(("query",), vec), pick(("groundtruth", "gt"), (".ivecs", ".ibin"))


def read_vectors(capi, path, limit=0):
    """one of the harness's vector files -> float32 (n, d); .u8bin keeps the unsigned byte values (the harness's own fbin_read
    widens them as signed chars, eval/bound.cpp:83-92 -- capi.read_fbin(nbytes=1) -- which would fold 128..255 below zero)"""
    low = path.lower()
    if low.endswith(".fvecs"):
        x = capi.read_fvecs(path)
        return x[:limit] if limit else x
    if low.endswith(".u8bin"):
        x, _ = capi.read_fbin(path, num=limit, nbytes=1)
        return np.where(x < 0, x + 256.0, x).astype(np.float32)
    x, _ = capi.read_fbin(path, num=limit, nbytes=4)
    return x


def read_ids(capi, path):
    return capi.read_ivecs(path) if path.lower().endswith(".ivecs") else capi.read_ibin(path)[0]


def kmeans_centroids(torch, xb, nlist, iters, seed):
    """bench infrastructure only (k-means is outside the hot path): a few Lloyd steps on a sample"""
    g = torch.Generator(device=xb.device)
    g.manual_seed(seed)
    ns = min(xb.shape[0], 256 * nlist)
    samp = xb[torch.randperm(xb.shape[0], generator=g, device=xb.device)[:ns]]
    cen = samp[:nlist].clone()
    for _ in range(iters):
        cn = (cen * cen).sum(1)
        assign = torch.empty(ns, dtype=torch.long, device=xb.device)
        for i0 in range(0, ns, 1 << 17):
            x = samp[i0:i0 + (1 << 17)]
            assign[i0:i0 + x.shape[0]] = (cn[None, :] - 2 * x @ cen.T).argmin(1)
        sums = torch.zeros_like(cen).index_add_(0, assign, samp)
        cnt = torch.bincount(assign, minlength=nlist).float()
        cen = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1)[:, None], cen)
    return cen.contiguous()


def ground_truth(torch, xb, xq, K):
    """exact on uint8-valued data: every fp32 partial sum stays below 2**24"""
    nq = xq.shape[0]
    bn = (xb * xb).sum(1)
    D = torch.full((nq, K), float("inf"), device=xb.device)
    I = torch.full((nq, K), -1, dtype=torch.long, device=xb.device)
    qs = 1000
    for q0 in range(0, nq, qs):
        q = xq[q0:q0 + qs]
        qn = (q * q).sum(1)
        bd, bi = D[q0:q0 + qs], I[q0:q0 + qs]
        for b0 in range(0, xb.shape[0], 1 << 20):
            b = xb[b0:b0 + (1 << 20)]
            dist = qn[:, None] + bn[None, b0:b0 + b.shape[0]] - 2 * (q @ b.T)
            cd, ci = dist.topk(K, dim=1, largest=False)
            md = torch.cat([bd, cd], 1)
            mi = torch.cat([bi, ci + b0], 1)
            sd, si = md.topk(K, dim=1, largest=False)
            bd, bi = sd, mi.gather(1, si)
        D[q0:q0 + qs], I[q0:q0 + qs] = bd, bi
    return D.cpu().numpy(), I.cpu().numpy()


def recall_dist(D, gtD, topk):
    """the reference's own recall (eval/bound.cpp:117-128): returned distances within the true k-th"""
    thr = gtD[:, topk - 1:topk] + 1e-6
    return (D[:, :topk] <= thr).sum(1) / float(topk)


class StepResults:
    """(D, I, my_nprobe) of EVERY step of a timed region, for the parity check after the clock has stopped: each step writes into a
    result buffer of its own (nothing is copied or compared while the clock runs), `collect` then folds steps over the same slice
    that returned identical bytes into one stored variant with a count, and `check` compares every variant with the reference's
    result for its slice.  What is compared is therefore what the timed searches themselves returned -- six in flight, whatever
    kernels the engine picks under that concurrency -- not a re-run."""
    KEEP = 128  # result buffers (6 MB each at the headline shape): of a longer region the last KEEP steps are checked

    def __init__(self):
        self.pending, self.variants, self.steps, self.unchecked = [], {}, 0, 0

    def note(self, sn, start, buf, np_):
        self.pending.append((sn, start, buf, np_))

    def collect(self, nbuf, ses):
        last = max(sn for sn, _, _, _ in self.pending) if self.pending else -1
        for sn, start, (D, I), np_ in self.pending:
            if sn + nbuf <= last:  # (its buffer was written again by a later step)
                self.unchecked += 1
                continue
            self.steps += 1
            npq = None if np_ is None else np_[start:start + ses]
            for v in self.variants.setdefault(start, []):
                if np.array_equal(v["D"].view(np.uint32), D.view(np.uint32)) and np.array_equal(v["I"], I) and (npq is None or np.array_equal(v["np"], npq)):
                    v["count"] += 1
                    break
            else:
                self.variants[start].append({"D": D.copy(), "I": I.copy(), "np": None if npq is None else npq.copy(), "count": 1})
        self.pending = []
        return self

    def check(self, ref):
        """ref(start) -> (D, I, my_nprobe or None) of the reference for the slice that starts at `start` (columns as the steps')"""
        bad = bad_q = 0
        for start, vs in self.variants.items():
            rD, rI, rnp = ref(start)
            for v in vs:
                diff = (v["I"] != rI).any(1) | (v["D"].view(np.uint32) != np.ascontiguousarray(rD).view(np.uint32)).any(1)
                if rnp is not None and v["np"] is not None:
                    diff |= v["np"] != rnp
                if diff.any():
                    bad += v["count"]
                    bad_q += int(diff.sum()) * v["count"]
        return {"timed_steps_checked": self.steps, "timed_steps_differing": bad, "queries_differing_over_those_steps": bad_q,
                "steps_not_kept": self.unchecked, "slices": len(self.variants),
                "distinct_results_per_slice_max": max([len(v) for v in self.variants.values()] or [0])}


def run_shards(args, torch, dist, capi, rank, world, local, dev, red_dev):
    """BASELINE configs[3]: SIFT-10M-like, IVF4096,Flat, k = topk, fixed nprobe; the inverted lists are sharded by list id over
    the ranks (the reference's IndexShards over sub-indexes that share one coarse quantizer, Auncel/IndexShards.cpp:261-311;
    owner balanced by list bytes).  The coarse quantization is computed once: rank r ranks the queries
    [r n / N, (r + 1) n / N) and the key rows (n x nprobe x 8 bytes) are all-gathered -- the path's one exchange step, RCCL over
    xGMI; every rank then scans the probed lists it owns for the whole batch, the per-rank (D, I) tables (n x k x 12 bytes) are
    gathered on rank 0 and merged there with merge_tables semantics (IndexShards.cpp:44-105).  Strong scaling: database and
    batch are fixed as N grows.  N = 1 runs the same code with one shard.  A step = coarse + all-gather + search + merge over
    the whole batch of `--test` resident queries.  Returns the result line (rank 0) or None."""
    from auncel_amd import sharding
    d, nlist, k, nq = args.d, args.nlist, args.topk, args.test
    t0 = time.time()
    xb_t, _, draw = gen_data(torch, dev, args.nb, 0, d, args.blobs, args.sigma, 1235)
    g = torch.Generator(device=dev)
    g.manual_seed(7)  # the same queries on every rank
    xq_t = draw(nq, g)
    nvalid = min(1000, nq)
    gtD, _ = ground_truth(torch, xb_t, xq_t[:nvalid], k)
    xb = xb_t.cpu().numpy()
    xq = xq_t.cpu().numpy()
    del xb_t, xq_t
    torch.cuda.empty_cache()
    cen, _ = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=25, coarse_mode=0, device=local)  # the reference's IVF training
    # every rank derives the same list assignment with the engine's exact assignment kernel, then keeps the lists it owns
    q = capi.Handle(d, nlist, capi.METRIC_L2, local)
    q.set_centroids(cen)
    assign = np.empty(args.nb, dtype=np.int64)
    for i0 in range(0, args.nb, 1 << 20):
        assign[i0:i0 + (1 << 20)] = q.coarse(xb[i0:i0 + (1 << 20)], 1, mode=0)[1][:, 0]
    del q
    sizes = np.bincount(assign, minlength=nlist)
    owner = sharding.assign_owners(sizes, world)
    mine = sharding.local_assignment(assign, owner, rank)
    h = capi.Handle(d, nlist, capi.METRIC_L2, local)
    h.set_centroids(cen)
    keep = mine >= 0
    h.add(xb[keep], xids=np.nonzero(keep)[0].astype(np.int64), precomputed_idx=mine[keep])
    if not (world > 1 and rank == 0):
        del xb  # (rank 0 of a sharded run searches the undivided index once at the end: the result the shards must reproduce)
    h.set_queries(xq)
    log(f"shards: data, k-means, assignment, {int(keep.sum())} of {args.nb} vectors on rank 0: {time.time() - t0:.1f}s")

    counts = [(r + 1) * nq // world - r * nq // world for r in range(world)]
    # steps in flight (sharding.run_pipelined): `lag` searches at a time on the engine's internal contexts, the coarse rankings of the
    # next steps ahead of them, one thread issuing the two collectives in a fixed order, the merge on its own thread on rank 0
    lag = max(1, args.shard_lag)
    h.set_async_depth(min(16, lag + 2))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(nsteps):
        return sharding.run_pipelined(h, capi.METRIC_L2, capi.merge_tables, nq, k, args.nprobe, counts, rank, dist if world > 1 else None,
                                      nsteps, lag=lag)

    if args.warmup:
        run(max(args.warmup, lag + 1))
    barrier()
    t0 = time.perf_counter()
    out, acc = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    h.set_async_depth(0)
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same step one at a time (after the clock, never part of `value`): the kernels alone on the chip -- what the roofline is
    # quoted on, as in the headline -- and the serial form's step time next to the pipelined one
    solo = {"coarse_ms": 0.0, "scan_ms": 0.0, "select_ms": 0.0, "scan_launches": 0.0, "scan_min_bytes": 0.0}
    nsolo = max(2, min(4, args.steps))
    barrier()
    ts0 = time.perf_counter()
    for _ in range(nsolo):
        _, ck_ = h.coarse_resident(sum(counts[:rank]), counts[rank], args.nprobe, mode=0, want_dis=False)
        solo["coarse_ms"] += h.last_timing()["coarse_ms"]
        keys_ = sharding.allgather_rows(ck_, counts, dist if world > 1 else None)
        Ds, Is = h.search_resident_preassigned(0, nq, k, keys_)
        tm_ = h.last_timing()
        for key in ("scan_ms", "scan_launches", "scan_min_bytes", "select_ms"):
            solo[key] += tm_[key]
        out_solo = sharding.gather_and_merge(Ds, Is, capi.METRIC_L2, capi.merge_tables, dist if world > 1 else None)
    barrier()
    solo_ms = 1e3 * (time.perf_counter() - ts0) / nsolo
    # per-rank kernel times of the timed region (ms per step): what divides by N and what does not
    mine = [acc.get(key, 0.0) / args.steps for key in ("coarse_ms", "exchange_ms", "scan_ms", "select_ms")]
    if world > 1:
        tt = torch.tensor(mine, device=red_dev, dtype=torch.float64)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = [[float(v) for v in t_.tolist()] for t_ in allt]
    else:
        per_rank = [mine]
    line = None
    single_sha = None
    if rank == 0 and world > 1:
        # the same search over the undivided index, on this rank: what IndexShards over N sub-indexes must give (IndexShards.cpp:261-311)
        _, ckf = h.coarse_resident(0, nq, args.nprobe, mode=0, want_dis=False)
        hf = capi.Handle(d, nlist, capi.METRIC_L2, local)
        hf.set_centroids(cen)
        hf.add(xb, precomputed_idx=assign)
        del xb
        hf.set_queries(xq)
        Df, _ = hf.search_resident_preassigned(0, nq, k, ckf)
        single_sha = __import__("hashlib").sha256(np.ascontiguousarray(Df).tobytes()).hexdigest()
        del hf
    if rank == 0:
        D, I = out
        rec = recall_dist(D[:nvalid], gtD, k)
        load = np.bincount(owner, weights=sizes, minlength=world)
        launches = max(solo["scan_launches"], 1)
        per_launch = solo["scan_min_bytes"] / launches
        achieved = (per_launch / 1e9) / (solo["scan_ms"] / launches / 1e3) if solo["scan_ms"] > 0 else None
        same_as_serial = bool(np.array_equal(out_solo[0].view(np.uint32), D.view(np.uint32)) and np.array_equal(out_solo[1], I))
        line = {
            "metric": "queries/sec, SIFT-10M d=128 IVF4096 k=10 fixed nprobe, IndexShards (lists sharded by list id, host top-k merge)",
            "value": nq * args.steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8" if h.scan_arith() == 2 else "f32", "data": "synthetic",
            "config": {"workload": f"SIFT-{args.nb // 1000000}M-like d={d} IVF{nlist},Flat k={k} nprobe={args.nprobe}, IndexShards over {world} GPU(s): "
                                   f"lists sharded by list id, batch {nq} resident queries searched by every shard, host merge_tables on rank 0",
                       "nb": args.nb, "sigma": args.sigma, "nprobe": args.nprobe, "recall_at_k_mean": float(rec.mean()),
                       "ranks_seen": getattr(args, "ranks_seen", world), "collective_backend": os.environ.get("BENCH_DIST_BACKEND", "nccl") if world > 1 else None,
                       "coarse_queries_per_rank": [int(c) for c in counts],
                       "steps_in_flight": lag, "steps_merged": int(acc["steps_merged"]),
                       "pipeline": "sharding.run_pipelined: amd_ivf_submit_coarse_resident / amd_ivf_submit_search_resident_preassigned tickets, one "
                                   "thread issuing key all-gather and table gather in a fixed order, amd_ivf_merge_tables on its own thread on rank 0",
                       "per_rank_ms_per_step": {"columns": ["coarse (own share of the batch)", "all-gather of the key rows (host wall)", "scan", "select"],
                                                "rows": per_rank},
                       # distances of the merged result (sorted rows: identical for any number of shards, whatever the order
                       # the merge gives equal distances)
                       "distances_sha256": __import__("hashlib").sha256(np.ascontiguousarray(D).tobytes()).hexdigest(),
                       "single_index_distances_sha256": single_sha,  # (N > 1: recomputed on rank 0 over the undivided index)
                       "shard_bytes_max_over_min": float(load.max() / max(load.min(), 1))},
            "roofline": {"bound": "hbm", "kernel": "scan_mfma_thr_kernel + scan_mfma_pair_kernel" if h.scan_arith() == 2 else "scan_lanes_kernel", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0 if achieved else None, "traffic": None,
                         "traffic_source": "engine's lower bound (every probed list once per round + rows written), rank 0's shard",
                         "measured": "one step at a time, same run, after the timed region",
                         "min_bytes_per_launch": per_launch, "avg_launch_ms": solo["scan_ms"] / launches, "launches_per_step": launches / nsolo,
                         "other_kernels_ms_per_step": {"coarse": solo["coarse_ms"] / nsolo, "select": solo["select_ms"] / nsolo},
                         "in_flight": {"steps": lag, "avg_launch_ms": acc["scan_ms"] / max(acc["scan_launches"], 1),
                                       "coarse_ms_per_step": acc["coarse_ms"] / args.steps, "select_ms_per_step": acc["select_ms"] / args.steps,
                                       "merge_ms_per_step": acc["merge_ms"] / max(acc["steps_merged"], 1)}},
            "one_step_at_a_time": {"ms_per_step": solo_ms, "value": nq / solo_ms * 1e3, "unit": "queries/s", "same_results_as_pipelined": same_as_serial},
        }
        if world == 1 and not args.no_cpu:
            # CPU side: the pinned restatement of IndexIVF::search (coarse + search_preassigned) on a bounded sample, all host cores
            from oracle import pyoracle
            S = min(args.cpu_sample, 2000, nq)
            codes, ids, off = [], [], np.zeros(nlist + 1, dtype=np.uintp)
            for l in range(nlist):
                c, i_ = h.get_list(l)
                codes.append(c)
                ids.append(i_)
                off[l + 1] = off[l] + len(i_)
            lists = pyoracle.Lists.__new__(pyoracle.Lists)
            lists.metric, lists.centroids, lists.nlist, lists.d = pyoracle.METRIC_L2, cen, nlist, d
            lists.off, lists.codes, lists.ids = off, np.concatenate(codes), np.concatenate(ids)
            del codes, ids
            lists.struct = pyoracle.OrcIndex(lists.metric, d, nlist, pyoracle._s(lists.off), pyoracle._f(lists.codes), pyoracle._i(lists.ids))
            cores = host_cores()
            tc = time.perf_counter()
            cd, ck = pyoracle.knn(pyoracle.METRIC_L2, xq[:S], cen, args.nprobe, nthreads=cores)
            oD, oI, _ = pyoracle.search_preassigned(lists, xq[:S], k, ck, cd, nthreads=cores)
            cpu_s = time.perf_counter() - tc
            line["cpu_baseline"] = {"value": S / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                                    "sample": f"first {S} of the {nq} queries, same lists, coarse + search_preassigned, OpenMP over queries",
                                    "gpu_matches_cpu_on_sample": bool(np.array_equal(oI, I[:S]) and np.array_equal(oD, D[:S]))}
            # the compiled reference itself where its harness travelled (oracle/_ref/ref_harness fixedbench: IndexIVF::search, one query per
            # call, OpenMP over queries -- the same lists, centroids and queries)
            from oracle import refbench
            if refbench.available() and not args.no_ref:
                try:
                    ro = refbench.run_fixed(pyoracle.METRIC_L2, cen, lists.off, lists.codes, lists.ids, xq[:S], k, args.nprobe, threads=cores)
                    port = line["cpu_baseline"]
                    line["cpu_baseline"] = {"value": S / ro["seconds_all_threads"], "unit": "queries/s", "cores": ro["threads"], "kind": "reference",
                                            "sample": f"first {S} of the {nq} queries: the compiled reference (Auncel/*.cpp, -O3 -msse4) on the engine's lists and "
                                                      "centroids, IndexIVF::search(1, ...) per query at the same nprobe, OpenMP over queries",
                                            "gpu_matches_cpu_on_sample": bool(np.array_equal(ro["I"], I[:S]) and np.array_equal(ro["D"].view(np.uint32), D[:S].view(np.uint32))),
                                            "port": port}
                except Exception as e:  # noqa: BLE001 -- the port's figures stay
                    log("reference harness not usable here:", repr(e))
    del h
    torch.cuda.empty_cache()
    if line is not None and single_sha is not None and single_sha != line["config"]["distances_sha256"]:
        log("SHARDS MISMATCH: the merged distances of", world, "shards differ from the single index's")
        line["config"]["shards_equal_single_index"] = False
    elif line is not None and single_sha is not None:
        line["config"]["shards_equal_single_index"] = True
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--nb", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--train", type=int, default=5000)
    ap.add_argument("--test", type=int, default=5000)
    ap.add_argument("--sigma", type=float, default=SIGMA)
    ap.add_argument("--blobs", type=int, default=20000)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--maxtopk", type=int, default=100)
    ap.add_argument("--bound", type=float, default=0.95)
    ap.add_argument("--std-m", type=float, default=1.0)
    ap.add_argument("--cpu-sample", type=int, default=5000)
    ap.add_argument("--mode", choices=["adaptive", "shards"], default="adaptive",
                    help="adaptive: BASELINE config 2, the headline (replicas at N > 1).  shards: config 4 -- fixed nprobe, the inverted "
                         "lists sharded by list id over the N GPUs (IndexShards), per-GPU partial top-k merged on the host; strong scaling")
    ap.add_argument("--nprobe", type=int, default=32, help="--mode shards: probes per query")
    ap.add_argument("--shard-lag", type=int, default=2, help="--mode shards: steps in flight (searches on the GPU while earlier steps' tables are gathered and merged)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the untimed extra legs (one batch at a time, fp32 path, guaranteed-bound point): profiling runs")
    ap.add_argument("--no-other", action="store_true",
                    help="skip the other_configs block (BASELINE configs 1, 3, 5 at fixed nprobe, each compared with the compiled reference)")
    ap.add_argument("--no-ref", action="store_true", help="cpu_baseline from the CPU restatement only (skip oracle/_ref/ref_harness)")
    ap.add_argument("--kmeans", choices=["engine", "torch"], default="engine",
                    help="coarse centroids: the engine's Clustering::train restatement (amd_ivf_kmeans, the reference's IVF "
                         "training: 25 iterations, 256 points per centroid) or a 4-step torch Lloyd")
    ap.add_argument("--stagger-ms", type=float, default=0.0,
                    help="context j issues its first step j x this many ms after context 0 (inside the timed region): out of phase, the "
                         "scan of one batch runs under the selection of another; started together they tend to stay in step")
    ap.add_argument("--pinned-out", type=int, default=1, help="1: result buffers in page-locked host memory, 0: pageable")
    ap.add_argument("--slices", type=int, default=4,
                    help="distinct resident query slices of --test queries each; step s searches slice s mod this (no step repeats the "
                         "previous step's batch)")
    ap.add_argument("--data", default=None,
                    help="directory with base / query / ground-truth files in the reference harness's layouts (.fvecs/.ivecs, .fbin/.ibin, "
                         ".u8bin): the same pipeline on real data instead of the synthetic blobs")
    ap.add_argument("--data-rows", type=int, default=0, help="--data: base vectors to read (0: the whole file)")
    ap.add_argument("--coarse-ties", choices=["redo", "id", "heap"], default="redo",
                    help="order inside runs of bit-equal coarse distances: redo = the reference's (its heap's history) for every query, in "
                         "one pass (include/auncel_amd.h: amd_ivf_coarse_tie_rows) -- the regime in which every result equals the "
                         "reference's by construction; id = centroid number (no heap: what the engine does for large calls when the "
                         "option is unset)")
    ap.add_argument("--runner", choices=["async", "threads"], default="async",
                    help="how --in-flight batches are kept in flight: async = ONE caller thread through amd_ivf_submit_adaptive / "
                         "amd_ivf_wait (the engine's internal contexts), threads = one host thread + amd_ivf_clone context per batch")
    ap.add_argument("--coalesce", type=int, default=1,
                    help="async runner: queued steps over adjacent resident slices that the engine may serve in ONE pass over the lists "
                         "(option \"coalesce\" of include/auncel_amd.h; every step stays one amd_ivf_submit_adaptive call with its own result)")
    ap.add_argument("--in-flight", type=int, default=6,
                    help="batches kept in flight per GPU, each from its own host thread on its own search context "
                         "(amd_ivf_clone); 1 = one batch at a time")
    args = ap.parse_args()
    # --gpus N > 1 without a launcher: start the N ranks ourselves, exactly as the driver would (one process per GPU, RCCL),
    # BEFORE anything in this process touches the GPU (the parent only counts devices, which does not initialise HIP), wait for
    # them and leave with their exit code.  A run that was asked for N GPUs never degrades to fewer silently: too few visible
    # devices is an error here, and a rank whose process group has another size than --gpus is an error below.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        if "BENCH_DEVICE" not in os.environ:  # (BENCH_DEVICE: the rehearsal with every rank on one device, gloo)
            ndev = kfd_gpu_count()  # (sysfs: the parent never touches the HIP runtime)
            if ndev is None:  # (no driver topology to read -- a container without the device nodes: torch's count, which does not start HIP)
                import torch
                ndev = torch.cuda.device_count()
            if ndev < args.gpus:
                sys.exit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible on this node")
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"--gpus {args.gpus} without a launcher: starting {' '.join(cmd[1:8])} ...")
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    # hardware queues the HIP runtime maps its streams onto: the engine lays its streams out for 8 per priority class; the library does
    # not touch the environment, so it is said here, before HIP starts, whichever of torch and the engine touches the GPU first
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    import torch.distributed as dist

    from auncel_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_DIST_BACKEND=gloo BENCH_DEVICE=0: rehearsal of the multi-rank flow on a box with one GPU (all ranks on it)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if "BENCH_DEVICE" in os.environ:
        local = int(os.environ["BENCH_DEVICE"])
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: launched with WORLD_SIZE {world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    ranks_seen = dist.get_world_size() if world > 1 else 1
    if world != args.gpus or ranks_seen != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: WORLD_SIZE is {world} and the process group has {ranks_seen} rank(s)")
    if backend == "nccl" and "BENCH_DEVICE" not in os.environ and torch.cuda.device_count() < world:
        sys.exit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
    args.ranks_seen = ranks_seen
    # Host waits: every batch in flight has a host thread waiting on its stream between rounds.  Spinning waits (the HIP default)
    # are ~2 % faster but need a core each; when the ranks' threads outnumber the cores this process tree may use (cgroup quota),
    # the engine sleeps on blocking events instead (include/auncel_amd.h: AUNCEL_AMD_BLOCKING_SYNC).
    if host_cores() < world * (args.in_flight + 1):
        os.environ.setdefault("AUNCEL_AMD_BLOCKING_SYNC", "1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the max-over-ranks reduction lives

    if args.mode == "shards":
        line = run_shards(args, torch, dist, capi, rank, world, local, dev, red_dev)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.destroy_process_group()
        if rank == 0 and line["config"].get("shards_equal_single_index") is False:
            sys.exit(3)  # (a sharded result that is not the single index's is not a measurement)
        return

    d, nlist, K, topk, ts, ses = args.d, args.nlist, args.maxtopk, args.topk, args.train, args.test
    nsl = max(1, args.slices)
    t0 = time.time()
    data_desc, gt_check = "synthetic", None
    if args.data:
        fb, fq, fg = find_data_files(args.data)
        assert fb and fq, f"--data {args.data}: need *base* and *query* files (.fvecs / .fbin / .u8bin)"
        xb_h = read_vectors(capi, fb, args.data_rows)
        xq_h = read_vectors(capi, fq)
        args.nb, d = xb_h.shape
        assert xq_h.shape[1] == d
        # the query file is cut like the reference's run (eval/bound.cpp:337-340): a training half, then the timed slices
        nsl = max(1, min(nsl, (xq_h.shape[0] - ts) // ses))
        assert xq_h.shape[0] >= ts + ses, f"{fq}: {xq_h.shape[0]} queries, need {ts} training + {ses} timed"
        xq_h = xq_h[:ts + nsl * ses]
        xb_t = torch.from_numpy(xb_h).to(dev)
        xq_t = torch.from_numpy(xq_h).to(dev)
        del xb_h, xq_h
        data_desc = f"real: {os.path.basename(fb)}, {os.path.basename(fq)}"
    else:
        xb_t, _, draw = gen_data(torch, dev, args.nb, 0, d, args.blobs, args.sigma, 1235)
        gq = torch.Generator(device=dev)
        gq.manual_seed(777 + rank)  # every replica searches its own query set
        xq_t = draw(ts + nsl * ses, gq)
    cen_t = kmeans_centroids(torch, xb_t, nlist, 4, 99)
    gtD, gtI = ground_truth(torch, xb_t, xq_t, K)
    if args.data and fg:
        # the file's ids against the brute-force kernel on 200 queries: a true neighbour is within the computed k-th distance
        gi = read_ids(capi, fg)
        m = min(200, gi.shape[0], xq_t.shape[0])
        kk = min(topk, gi.shape[1])
        dd = ((xb_t[torch.from_numpy(gi[:m, :kk].astype(np.int64)).to(dev)] - xq_t[:m, None, :]) ** 2).sum(2).cpu().numpy()
        gt_check = {"file": os.path.basename(fg), "queries": int(m), "k": int(kk),
                    "agrees_with_brute_force": bool((dd <= gtD[:m, kk - 1:kk] * (1 + 1e-6) + 1e-3).all())}
        log("ground-truth file check:", gt_check)
    xb, xq, cen = xb_t.cpu().numpy(), xq_t.cpu().numpy(), cen_t.cpu().numpy()
    del xb_t, xq_t, cen_t
    torch.cuda.empty_cache()
    log(f"data + ground truth: {time.time() - t0:.1f}s")
    if args.kmeans == "engine":
        t0 = time.time()
        cen, kobj = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=25, coarse_mode=0, device=local)
        log(f"k-means (amd_ivf_kmeans, 25 iterations on {min(len(xb), 256 * nlist)} points): {time.time() - t0:.1f}s, "
            f"objective {kobj[0]:.4g} -> {kobj[-1]:.4g}")

    # ---- index build through the C ABI (add = exact assignment on the GPU + append)
    t0 = time.time()
    h = capi.Handle(d, nlist, capi.METRIC_L2, local)
    h.set_centroids(cen)
    h.add(xb)
    del xb
    h.set_interdis(None)
    h.set_queries(xq)
    # every search of this run -- the grid below, the timed region, the legs -- in the reference's exact regime
    ties_opt = {"id": 0, "heap": 1, "redo": 2}[args.coarse_ties]
    if "AUNCEL_AMD_COARSE_TIES" not in os.environ:
        h.set_option("coarse_ties", ties_opt)
    sizes = np.array([h.list_size(l) for l in range(nlist)])
    log(f"index build: {time.time() - t0:.1f}s; list size mean {sizes.mean():.0f} max {sizes.max()} empty {(sizes == 0).sum()}")

    # ---- offline trace training (Error_sys::sys_train)
    t0 = time.time()
    ntr = 0
    while (1 << ntr) <= nlist // 8:
        ntr += 1
    # the training half is split: traces from its first 80 % (Error_sys::sys_train), hyper-parameters chosen on the remaining
    # 20 % -- queries that shaped neither the traces nor the timing
    tfit = (ts * 4 // 5) // 10 * 10
    raw = [np.full((tfit * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
    h.train_samples(0, tfit, K, gtD, tfit, raw)
    traces = [capi.trace_sb(r) for r in raw]
    h.set_tuner(K, traces, capi.arcos_table())
    log(f"trace training on {tfit} queries: {time.time() - t0:.1f}s; bins per trace {[len(t[0]) for t in traces]}")

    # ---- hyper-parameters: the reference ships hand-tuned (multipler, std_m) rows for IVF1024 only (hyperparameter.txt) and
    # has no tuner.  Both operating points below are chosen on the validation part of the training half (queries tfit..ts);
    # the timed half is never looked at while choosing:
    #   headline    the first pair, from the most aggressive on, whose MEAN recall@topk holds the bound plus one standard
    #               error of that mean (the metric's "queries/s @ recall@10 >= 0.95")
    #   guaranteed  the first pair whose MINIMUM over queries holds it -- the reference's own acceptance check,
    #               "Error bound is guaranteed" (eval/bound.cpp:404-414)
    # std_m scales the spread term of the k-scaling estimate (Trace::search: mean + std_m * std), multipler the probe count
    # at which a fired query stops.
    nall = ts + nsl * ses
    req = np.full(nall, args.bound, dtype=np.float32)
    grid = [(1.0, sm) for sm in (0.0, 0.25, 0.5, 0.75) if sm < args.std_m]
    grid += [(m, args.std_m) for m in (1.0, 1.25, 1.5, 1.75, 2.0, 2.5, 3.0, 4.0, 5.0, 6.0, 8.0, 12.0, 16.0, 24.0)]
    nval = ts - tfit
    headline, guaranteed, best_min = None, None, (0.0, None)
    for mult, sm in grid:
        np_ = np.zeros(nall, dtype=np.uint64)
        tr_ = np.zeros(nall, dtype=np.float32)
        D, I = h.search_adaptive(tfit, nval, topk, mult, sm, req, np_, tr_)
        rec = recall_dist(D, gtD[tfit:ts], topk)
        se = float(rec.std() / np.sqrt(nval))
        log(f"  multipler {mult} std_m {sm}: validation recall@{topk} mean {rec.mean():.4f} (s.e. {se:.4f}) min {rec.min():.2f} "
            f"nprobe mean {np_[tfit:ts].mean():.1f}")
        if headline is None and rec.mean() - se >= args.bound:
            headline = (mult, sm)
        if rec.min() > best_min[0]:
            best_min = (float(rec.min()), (mult, sm))
        if guaranteed is None and rec.min() >= args.bound:
            guaranteed = (mult, sm)
        if headline is not None and guaranteed is not None:
            break
    if headline is None:
        headline = grid[-1]
    chosen, chosen_std = headline
    args.std_m = chosen_std

    # ---- timed region: K steps over the resident test batch.  A step is one search_adaptive call over the whole batch;
    # with --in-flight N the K calls are issued from N host threads, each on its own search context over the same
    # device-resident index, so that one batch's latency-bound phases (ordered selection, round planning) overlap
    # another batch's scans.  Every step is a complete, independent search either way.
    import threading
    nfl = max(1, min(args.in_flight, args.steps))
    use_async = args.runner == "async" and nfl > 1
    if use_async:
        # one caller thread: submit returns a ticket at once, the engine runs up to nfl searches at a time on internal contexts
        # (its statistics and settings cover them), wait collects a result
        h.set_async_depth(nfl)
        ctxs = [h]
    else:
        ctxs = [h] + [h.clone() for _ in range(nfl - 1)]
        for c in ctxs[1:]:
            c.set_queries(xq)

    stagger_s = float(os.environ.get("BENCH_STAGGER_MS", args.stagger_ms)) / 1e3

    # result buffers of every context: page-locked host memory (the caller's choice in the reference's API too), so that
    # the 6 MB of (D, I) of a step come back by direct DMA instead of through the runtime's staging copies
    # (async: as many tickets again as searches run at a time wait in the engine's queue, so that a context that finishes out of
    # order finds its next search there instead of idling until the caller has collected the oldest)
    coalesce = max(1, args.coalesce) if use_async else 1
    if "AUNCEL_AMD_COALESCE" not in os.environ:
        h.set_option("coalesce", coalesce)
    nslots = 2 * nfl * coalesce if use_async else nfl

    def result_buffers(count):
        """`count` (D, I) pairs cut from ONE page-locked block each, so that the buffers of consecutive steps follow each other in
        memory: what lets the engine serve queued tickets over adjacent query ranges in one pass (option "coalesce")"""
        if not args.pinned_out:
            return [None] * count
        Dall = torch.empty((count, ses, K), dtype=torch.float32).pin_memory().numpy()
        Iall = torch.empty((count, ses, K), dtype=torch.int64).pin_memory().numpy()
        return [(Dall[i], Iall[i]) for i in range(count)]

    outs = result_buffers(nslots)  # one pair per batch in flight (or queued)
    async_outs = outs if use_async else None  # (the single-caller leg of a threads run makes its own)
    # ... and one pair per TIMED step (StepResults): what the timed searches return is what the parity leg checks
    step_bufs = result_buffers(max(min(args.steps, StepResults.KEEP), min(args.steps, nslots))) if args.pinned_out else None

    hyper = {"mult": chosen, "std_m": chosen_std}

    def step(ctx, stepno, slot=0, keep=None):
        """step s searches slice s mod nsl of the resident queries: consecutive steps never see the same batch"""
        # (stepno // nfl: with as many slices as contexts, stepno % nsl alone would hand a context the same slice every time)
        start = ts + ((stepno + stepno // nfl) % nsl) * ses
        np_ = np.zeros(nall, dtype=np.uint64)
        tr_ = np.zeros(nall, dtype=np.float32)
        buf = step_bufs[stepno % len(step_bufs)] if keep is not None and step_bufs else outs[slot]
        D, I = ctx.search_adaptive(start, ses, topk, hyper["mult"], hyper["std_m"], req, np_, tr_, out=buf)
        if keep is not None and step_bufs:
            keep.note(stepno, start, buf, np_)
        return D, I, np_, start

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def account(acc, tm, hinted, short, redone, res, det=None):
        if det:  # per phase: (ms by HIP events on the phase's own stream, launches); bytes the dense / threshold rounds had to move
            ph = acc.setdefault("phases", {})
            for key, val in det.items():
                if isinstance(val, tuple):
                    cur = ph.get(key, (0.0, 0.0))
                    ph[key] = (cur[0] + val[0], cur[1] + val[1])
                else:
                    ph[key] = ph.get(key, 0.0) + val
        acc["hinted_launches"] = acc.get("hinted_launches", 0) + hinted
        acc["short_hints"] = acc.get("short_hints", 0) + short
        acc["tie_redone"] = acc.get("tie_redone", 0) + redone
        for key in ("scan_ms", "scan_bytes", "scan_launches", "coarse_ms", "select_ms", "scan_min_bytes"):
            acc[key] = acc.get(key, 0.0) + tm[key]
        acc["slot_eff"] = acc.get("slot_eff", 0.0) + tm["slot_efficiency"]
        acc["last"] = res

    def run_steps_async(nsteps, acc, keep=None):
        """one caller thread keeps up to nfl steps in flight: submit step s, and once nfl are out wait for the oldest first.
        keep (a StepResults): every step returns into a result buffer of its own and is noted for the parity check"""
        pending = []
        # (a buffer must not come round again while its step is still out: at most len(async_outs) are)
        kept_bufs = step_bufs if keep is not None and step_bufs and len(step_bufs) >= min(nsteps, len(async_outs)) else None

        def finish():
            ticket, np_, start, sn, buf = pending.pop(0)
            D, I, tm, dg = h.wait(ticket)
            if os.environ.get("AUNCEL_BENCH_TRACE_ASYNC"):
                log(f"async step done at {1e3 * (time.perf_counter() - t_async0):.2f} ms: engine wall {tm['total_ms']:.2f} ms")
            if kept_bufs is not None:
                keep.note(sn, start, buf, np_)
            account(acc, tm, dg["hinted_launches"], dg["short_hints"], dg["tie_redone"], (D, I, np_, start))

        aslots = len(async_outs)
        t_async0 = time.perf_counter()
        for sn in range(nsteps):
            if len(pending) == aslots:
                finish()
            elif sn and sn < nfl and stagger_s:
                time.sleep(stagger_s)  # the first nfl submissions start out of phase (scan of one under selection of the other)
            start = ts + (sn % nsl) * ses
            np_ = np.zeros(nall, dtype=np.uint64)
            tr_ = np.zeros(nall, dtype=np.float32)
            buf = kept_bufs[sn % len(kept_bufs)] if kept_bufs is not None else async_outs[sn % aslots]
            pending.append((h.submit_adaptive(start, ses, topk, hyper["mult"], hyper["std_m"], req, np_, tr_, out=buf), np_, start, sn, buf))
        while pending:
            finish()

    def run_steps(nsteps, acc, keep=None):
        """steps j, j + nfl, ... on context j; acc collects per-step kernel timings and the last result"""
        if use_async and nfl > 1:
            return run_steps_async(nsteps, acc, keep)
        errs = []

        def worker(j):
            try:
                if j and stagger_s:
                    time.sleep(j * stagger_s)  # start the contexts out of phase (scan of one under selection of the other)
                # (dealt, not pulled: with steps pulled from a shared counter a context can come out of a short warm-up without
                # having searched at all, and its first search -- workspaces, streams -- lands in the timed region: 2.4-2.5 vs
                # 2.7 M q/s at --steps 20 --warmup 5)
                for sn in range(j, nsteps, nfl):
                    res = step(ctxs[j], sn, j, keep)
                    tm = ctxs[j].last_timing()
                    hints = ctxs[j].last_round_hints()
                    det = ctxs[j].last_timing_detail()
                    with lock:
                        account(acc, tm, hints[0], hints[1], ctxs[j].last_tie_redone(), res, det)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        lock = threading.Lock()
        if nfl == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(j,)) for j in range(nfl)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errs:
            raise errs[0]

    run_steps(max(args.warmup, nfl if args.warmup else 0), {})
    for c in ctxs:
        c.stats(reset=True)
    acc = {}
    kept_timed = StepResults()
    barrier()
    counts0 = h.async_counts()
    t0 = time.perf_counter()
    run_steps(args.steps, acc, kept_timed)
    barrier()
    elapsed = time.perf_counter() - t0
    counts1 = h.async_counts()
    kept_timed.collect(len(step_bufs) if step_bufs else 1, ses)  # (after the clock: copies of the distinct results, the buffers are free again)
    per_rank_value = [ses * args.steps / elapsed]
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_value = [ses * args.steps / float(v.item()) for v in allt]  # every rank's own clock over its own replica
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = {}
    for c in ctxs:
        for key, val in c.stats().items():
            st[key] = st.get(key, 0) + val
    workload = (f"SIFT-{args.nb // 1000000}M-like d={d} IVF{nlist},Flat max_topk={K} topk={topk} Auncel error-bound nprobe "
                f"(bound {args.bound}), batch {ses} resident queries per GPU, index replicated per GPU")
    D, I, my_np, q_start = acc["last"]
    D, I, my_np = D.copy(), I.copy(), my_np.copy()  # (the result buffers are reused by the legs below)
    my_sl = my_np[q_start:q_start + ses]  # my_nprobe of the slice the last timed step searched
    gt_sl = gtD[q_start:q_start + ses]
    skip_legs = set(filter(None, os.environ.get("AUNCEL_BENCH_SKIP_LEGS", "").split(",")))  # (diagnosis: one_batch, fp32, exact_ties)
    def timed_leg(nsteps):
        """nsteps more steps with the current settings, after the timed region (never part of `value`)"""
        leg = {}
        run_steps(min(nsteps, max(nfl, 2)), {})
        kept = StepResults()
        barrier()
        tl = time.perf_counter()
        run_steps(nsteps, leg, kept)
        barrier()
        leg["elapsed"] = time.perf_counter() - tl
        leg["kept"] = kept.collect(len(step_bufs) if step_bufs else 1, ses)
        return leg

    # the same steps one batch at a time (after the timed region, not part of `value`): per-launch kernel figures without
    # another batch sharing the chip
    solo = {}
    if nfl > 1 and not args.no_legs and "one_batch" not in skip_legs:
        nfl_keep, nfl = nfl, 1
        barrier()
        ts0 = time.perf_counter()
        run_steps(args.steps, solo)
        barrier()
        solo["elapsed"] = time.perf_counter() - ts0
        nfl = nfl_keep
    # the same workload on the fp32 lists (byte codes switched off on every context): what the engine does on data that is
    # not uint8-valued, and a cross-check of the byte-code path (results must be identical)
    fp32 = None
    kept_fp32 = kept_fixed = None
    arith = h.scan_arith()  # of the timed region
    if arith == 2 and not args.no_legs and "fp32" not in skip_legs:
        for c in ctxs:
            c.set_byte_codes(False)
        # (at least two rounds of the searches in flight: six steps were a third of the leg's own spread)
        nst = int(os.environ.get("AUNCEL_BENCH_FP32_STEPS", max(4 * nfl, args.steps)))  # (12 steps still read 0.9 once where 24 read 1.4-1.5)
        leg = timed_leg(nst)
        kept_fp32 = leg["kept"]
        fD, fI, f_np, f_start = leg["last"]
        if f_start != q_start:  # the same slice as the timed region's last step, for the comparison
            fD, fI, f_np, f_start = step(ctxs[0], (q_start - ts) // ses)
            fD, fI = fD.copy(), fI.copy()
        fp32 = {"value": ses * nst / leg["elapsed"], "unit": "queries/s", "ms_per_step": 1000.0 * leg["elapsed"] / nst,
                "scan_arith": {0: "fp32 reference order", 1: "fp32 fused"}.get(ctxs[0].scan_arith(), "?"),
                "scan_avg_launch_ms": leg["scan_ms"] / max(leg["scan_launches"], 1),
                "filter_copy": {2: "fp16", 1: "fp32", 0: "none"}.get(int(h.get_option("filter")), "?"),
                "same_results_as_byte_codes": bool(np.array_equal(fD, D) and np.array_equal(fI, I) and
                                                   np.array_equal(f_np[q_start:q_start + ses], my_sl))}
        # its own roofline, by the headline's convention: the kernels alone on the chip (a few steps one batch at a time, HIP events on
        # the streams the kernels run on).  The fp32 lists are four times the byte codes, and every round is a pass over them -- the
        # dense round on the vector ALU in the reference's rounding sequence, the threshold rounds as matrix-core filter + exact
        # recomputation.  Bytes = what the rounds could not avoid moving (every probed list once per round, as fp32 or as the fp16
        # copy, + rows / mask bits written: the planner's count).
        fsolo = {}
        nfl_keep, nfl = nfl, 1
        run_steps(2, {})
        barrier()
        tf0 = time.perf_counter()
        run_steps(max(4, nst // 2), fsolo)
        barrier()
        fsolo["elapsed"] = time.perf_counter() - tf0
        nfl = nfl_keep
        fp32["one_batch_at_a_time_ms"] = 1000.0 * fsolo["elapsed"] / max(4, nst // 2)
        ph = fsolo.get("phases", {})
        nst_r = max(4, nst // 2)
        fr = {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "measured": "one batch at a time, same run", "per_launch": []}
        tot_b = tot_ms = 0.0
        for key, what in (("scan_dense", "dense round: scan_lanes_kernel (vector ALU, reference order, lists in lane order)"),
                          ("scan_thr", "threshold rounds: scan_filter_kernel (matrix cores over the fp16 copy of the lists; option filter = 1: the "
                                       "fp32 copy) + rescore_kernel (exact recomputation of what it keeps)")):
            ms_l, n_l = ph.get(key, (0.0, 0.0))
            if not n_l:
                continue
            mb = ph.get("min_bytes_dense" if key == "scan_dense" else "min_bytes_thr", 0.0)
            tot_b += mb
            tot_ms += ms_l
            fr["per_launch"].append({"what": what, "launches_per_step": n_l / nst_r, "ms": ms_l / n_l, "min_bytes": mb / n_l,
                                     "GBps": mb / 1e9 / (ms_l / 1e3), "frac": mb / 1e9 / (ms_l / 1e3) / 8000.0})
        if tot_ms > 0:
            fr["achieved"] = tot_b / 1e9 / (tot_ms / 1e3)
            fr["frac"] = fr["achieved"] / 8000.0
            # all list passes of a step against the HBM peak, over the step time of the leg with the searches in flight
            fr["list_bytes_per_step"] = tot_b / nst_r
            fr["step_hbm_frac"] = tot_b / nst_r / 1e9 / (leg["elapsed"] / nst) / 8000.0
        fr["phases_ms_per_step"] = {k: v[0] / nst_r for k, v in ph.items() if isinstance(v, tuple)}
        fp32["roofline"] = fr
        for c in ctxs:
            c.set_byte_codes(True)
    # what the exact regime costs: the same steps with runs of bit-equal coarse distances left in centroid-number order (no heap; the
    # engine's choice for large calls when the option is unset).  Not the reference's result for the few queries that read such a run.
    id_ties = None
    if not args.no_legs and ties_opt == 2 and "AUNCEL_AMD_COARSE_TIES" not in os.environ and "id_ties" not in skip_legs:
        h.set_option("coarse_ties", 0)
        try:
            nst = max(nfl, args.steps // 3)
            leg = timed_leg(nst)
            id_ties = {"value": ses * nst / leg["elapsed"], "unit": "queries/s", "ms_per_step": 1000.0 * leg["elapsed"] / nst,
                       "setting": "coarse_ties = 0 (centroid-number order inside runs of equal coarse distances): NOT the timed configuration"}
        finally:
            h.set_option("coarse_ties", ties_opt)
    # the same index at fixed nprobe 32 (recall@10 ~ 0.955 on this data) through the same runner: what the adaptive rule buys over it
    fixed32 = None
    if not args.no_legs and "fixed" not in skip_legs:
        fnp, fk = 32, topk
        # (a step is 1 ms: twelve of them were a window that one hiccup moved by a third -- 2.9 to 5.2 M q/s on one box)
        nst = max(6 * nfl, 2 * args.steps)
        # (600 KB a step: every step of the leg keeps its own, for the parity check)
        fouts = [(np.empty((ses, fk), np.float32), np.empty((ses, fk), np.int64)) for _ in range(max(2 * nfl, nst))]

        def run_fixed(nsteps, keep=None):
            pend, last = [], None
            for sn in range(nsteps):
                if len(pend) == 2 * nfl:
                    last = h.wait(pend.pop(0)[0])
                st0 = ts + (sn % nsl) * ses
                buf = fouts[sn % len(fouts)]
                pend.append((h.submit_search_resident(st0, ses, fk, fnp, out=buf), st0))
                if keep is not None:
                    keep.note(sn, st0, buf, None)
            st_last = pend[-1][1]
            while pend:
                last = h.wait(pend.pop(0)[0])
            return last, st_last

        if nfl > 1:
            h.set_async_depth(nfl)
        run_fixed(max(2 * nfl, 4))
        kept_fixed = StepResults()
        barrier()
        tf0 = time.perf_counter()
        (fD, fI, _, _), f_st = run_fixed(nst, kept_fixed)
        barrier()
        f_el = time.perf_counter() - tf0
        kept_fixed.collect(len(fouts), ses)
        frec = recall_dist(fD, gtD[f_st:f_st + ses, :fk], fk)
        fixed32 = {"nprobe": fnp, "k": fk, "value": ses * nst / f_el, "unit": "queries/s", "ms_per_step": 1000.0 * f_el / nst,
                   "recall_at_k_mean": float(frec.mean()), "how": "amd_ivf_submit_search_resident / amd_ivf_wait on the same index and slices"}
        if not use_async:
            h.set_async_depth(0)
    # one query per call: the reference's own evaluation protocol (eval/bound.cpp:391-396 issues Error_sys::search(D, I, i, 1) for every
    # test query).  Wall time of the call through the C ABI (ctypes), and the time inside the entry point; results kept and compared
    # with the reference's below (parity leg).
    lat1 = None
    lat_keep = None
    if not args.no_legs and "latency1" not in skip_legs:
        ncall = min(400, nsl * ses)
        stride = (nsl * ses) // ncall  # (spread over every resident slice)
        ids1 = ts + np.arange(ncall) * stride
        np1 = np.zeros(nall, dtype=np.uint64)
        tr1 = np.zeros(nall, dtype=np.float32)
        lD = np.empty((ncall, K), np.float32)
        lI = np.empty((ncall, K), np.int64)
        # (calls of fewer than 20 queries: the engine's own policy for runs of equal coarse distances -- centroid-number order first, the
        # call repeated with the reference's heap order only if a run lies where the query read -- gives the reference's result as well
        # and does not send every ranking with a run in reach through the 1.6 ms heap: the option is left unset for this leg)
        if "AUNCEL_AMD_COARSE_TIES" not in os.environ:
            h.set_option("coarse_ties", float("nan"))
        for i in range(20):
            h.search_adaptive(int(ids1[i]), 1, topk, chosen, chosen_std, req, np1, tr1)
        wall, inside = np.zeros(ncall), np.zeros(ncall)
        for i in range(ncall):
            q = int(ids1[i])
            np1[q] = 0
            tq = time.perf_counter()
            D1, I1 = h.search_adaptive(q, 1, topk, chosen, chosen_std, req, np1, tr1)
            wall[i] = (time.perf_counter() - tq) * 1e3
            inside[i] = h.last_timing()["total_ms"]
            lD[i], lI[i] = D1[0], I1[0]
        if "AUNCEL_AMD_COARSE_TIES" not in os.environ:
            h.set_option("coarse_ties", ties_opt)
        lat_keep = (ids1, lD, lI, np1[ids1].copy())
        lat1 = {"what": "amd_ivf_search_adaptive over ONE resident query per call, the reference's protocol (eval/bound.cpp:391-396)", "calls": int(ncall),
                "ms_median": float(np.median(wall)), "ms_p90": float(np.percentile(wall, 90)), "ms_p99": float(np.percentile(wall, 99)), "ms_min": float(wall.min()),
                "inside_the_entry_point": {"ms_median": float(np.median(inside)), "ms_p90": float(np.percentile(inside, 90))},
                "queries_per_s_one_caller": float(1e3 / wall.mean()), "my_nprobe_mean": float(np1[ids1].mean())}
    # the reference's acceptance check at the operating point chosen for it on the training half
    guar = {"validation_min_recall_best": best_min[0], "validation_best_point": best_min[1]}
    if guaranteed is not None and not args.no_legs:
        hyper["mult"], hyper["std_m"] = guaranteed
        nst = max(4, args.steps // 3)
        leg = timed_leg(nst)
        gD, gI, g_np, g_start = leg["last"]
        grec = recall_dist(gD, gtD[g_start:g_start + ses], topk)
        guar.update({"multipler": guaranteed[0], "std_m": guaranteed[1], "value": ses * nst / leg["elapsed"], "unit": "queries/s",
                     "recall_min_test": float(grec.min()), "recall_mean_test": float(grec.mean()),
                     "bound_guaranteed_on_test": bool(grec.min() >= args.bound), "nprobe_mean": float(g_np[g_start:g_start + ses].mean())})
        hyper["mult"], hyper["std_m"] = chosen, chosen_std
    elif guaranteed is None:
        guar["note"] = "no grid point up to multipler 24 holds the bound for every validation query"
    # the same steps kept in flight by ONE caller thread through the asynchronous entry points (amd_ivf_submit_adaptive /
    # amd_ivf_wait: the engine's own contexts and worker threads) instead of one host thread + context per batch.  Last of the legs, and
    # the one that pays for it: which hardware queue a new stream shares with which is decided when it is created and depends on
    # every stream the process created before -- right after the timed region this leg reaches 3.0 M q/s, here 2.0-2.8 (and
    # whichever leg is put behind it instead loses as much: measured in every order)
    single_caller = None
    if nfl > 1 and not args.no_legs and not use_async:
        while len(ctxs) > 1:  # (the caller threads' contexts go first: their streams would share hardware queues with the engine's own)
            ctxs.pop().close()
        h.set_async_depth(nfl)
        async_outs = result_buffers(2 * nfl)
        run_steps_async(max(2 * nfl, 8), {})
        barrier()
        ta0 = time.perf_counter()
        aleg = {}
        run_steps_async(args.steps, aleg)
        barrier()
        a_el = time.perf_counter() - ta0
        aD, aI, a_np, a_start = aleg["last"]
        single_caller = {"value": ses * args.steps / a_el, "unit": "queries/s", "ms_per_step": 1000.0 * a_el / args.steps,
                         "how": f"one caller thread, amd_ivf_submit_adaptive / amd_ivf_wait, {nfl} searches at a time on the engine's "
                                f"internal contexts, {2 * nfl} tickets out; last leg of the run (see DESIGN.md section 5 on the order of the legs)"}
        h.set_async_depth(0)  # (the contexts' streams would crowd the hardware queues of anything run after)
    scan_ms, scan_bytes, scan_launches = acc["scan_ms"], acc["scan_bytes"], acc["scan_launches"]
    coarse_ms, select_ms, slot_eff = acc["coarse_ms"], acc["select_ms"], acc["slot_eff"] / args.steps

    rec = recall_dist(D, gt_sl, topk)
    log("my_nprobe of the timed queries: percentiles 10/25/50/75/90/95/99 =", np.percentile(my_sl, [10, 25, 50, 75, 90, 95, 99]).tolist(),
        "; share <= 12:", float((my_sl <= 12).mean()), "<= 42:", float((my_sl <= 42).mean()))
    # algorithmic bytes: the reference's own ndis counter (codes actually visited by the probe loops,
    # IndexIVF.cpp:676,733) x d x 4; `scan_bytes` (distances the tiles computed, incl. the probes a round
    # ran past a query's stop point) is reported beside it as computed_over_algorithmic
    alg_bytes = float(st["ndis"]) * d * 4.0
    scan_kernel = "scan_mfma_thr_kernel (dense round) + scan_mfma_pair_kernel (threshold round): see per_launch" if arith == 2 else "scan_lanes_kernel"
    min_bytes = acc["scan_min_bytes"]
    # HBM traffic of the scan per launch.  PMC counters cannot be read from inside this process: the measured figure comes
    # from the committed rocprofv3 --pmc passes over this same command (profiles/collect.sh -> summarize.py; FETCH_SIZE
    # doubled for the scan's wide loads as MI355X_MICROARCH.md prescribes), used only when that profile was taken on this
    # workload.  Without it the engine's own lower bound stands in (every probed list once per round + the rows written,
    # counted by the planning kernels): `achieved` is then a lower bound of the kernel's real rate.
    traffic, traffic_source = None, None
    cands = []
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
        if cands:
            pj = json.load(open(cands[-1]))
            if pj.get("_workload") == workload:
                # (average over the scan launches of a step: dense and threshold rounds; roofline.per_launch has them apart)
                ks_ = [k for k in pj if k.startswith(("scan_mfma", "scan_tiles_kernel", "scan_lanes_kernel")) and not k.endswith("[coarse]")]
                nd_ = sum(pj[k]["dispatches"] for k in ks_)
                if nd_:
                    traffic = sum(pj[k]["hbm_bytes_per_dispatch"] * pj[k]["dispatches"] for k in ks_) / nd_
                    traffic_source = os.path.relpath(cands[-1], ROOT)
    except Exception as e:  # a missing or stale profile leaves traffic null
        log("no PMC traffic figure:", e)
    launches = max(scan_launches, 1)
    # The kernel's own figure needs the kernel alone on the chip: with several batches in flight a launch shares HBM and CUs
    # with the other batches' kernels and its event span says little about the kernel.  The top-level roofline therefore comes
    # from the one-batch-at-a-time leg of this same run (same steps, same data, right after the timed region; `measured`
    # says which), and the spans of the timed region itself are kept beside it as `in_flight`.
    alone = solo if solo else acc
    a_ms, a_l = alone["scan_ms"], max(alone["scan_launches"], 1)
    bytes_per_launch = traffic if traffic is not None else alone["scan_min_bytes"] / a_l
    out = {
        "metric": "queries/sec @ recall@10>=0.95, SIFT-10M d=128 IVF4096, 1/2/4/8 GPU",
        "value": ses * args.steps * world / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # arithmetic of the dominant kernel: byte codes and exact integer contractions on the i8 matrix cores when the data is
        # uint8-valued (bit-identical to the reference's fp32 results there), fp32 in the reference's
        # summation order otherwise
        "dtype": "u8" if arith == 2 else "f32",
        "data": data_desc,
        "config": {
            "workload": workload,
            "query_slices": nsl, "queries_per_slice": ses,
            "ranks_seen": args.ranks_seen, "collective_backend": backend if world > 1 else None,
            "per_rank_value": per_rank_value,  # (value = all ranks' queries / the slowest rank's time)
            # scan grids of the device-chained rounds are sized from the previous search's counts (+ 12 %); a round that needs
            # more runs on fewer workgroups than it would have been given (it is still complete: the workgroups stride)
            "round_hint": {"launches_sized_by_a_hint": int(acc.get("hinted_launches", 0)), "hint_too_small": int(acc.get("short_hints", 0))},
            "in_flight": nfl, "coalesce": coalesce, "queries_searched_again_in_timed_region": int(acc.get("tie_redone", 0)),
            # (timed region: steps submitted, and the passes over the lists that served them)
            "tickets_and_passes": {"tickets": counts1[0] - counts0[0], "passes": counts1[1] - counts0[1]} if use_async else None,
            "runner": ("one caller thread: amd_ivf_submit_adaptive / amd_ivf_wait, the engine's internal contexts" if use_async
                       else "one host thread and one amd_ivf_clone context per batch in flight" if nfl > 1 else "one synchronous call at a time"),
            "host_wait": "blocking events" if os.environ.get("AUNCEL_AMD_BLOCKING_SYNC", "0") not in ("", "0") else "spin", "host_cores": host_cores(), "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "stagger_ms": stagger_s * 1e3,
            "scan_arith": {0: "fp32 reference order", 1: "fp32 fused", 2: "byte codes, v_mfma_i32_32x32x32_i8"}[arith],
            "nb": args.nb, "sigma": args.sigma, "multipler": chosen, "std_m": args.std_m,
            "hyper_parameters_chosen_on": "validation fifth of the training half (mean recall@%d - 1 s.e. >= %.2f); traces from the other four fifths" % (topk, args.bound),
            "recall_at_10_mean": float(rec.mean()), "recall_at_10_min": float(rec.min()),
            "recall_target_met_on_timed_half": bool(rec.mean() >= args.bound),
            "bound_guaranteed": bool(rec.min() >= args.bound),  # eval/bound.cpp:404-414 at the headline point
            "nprobe_mean": float(my_sl.mean()), "nprobe_max": int(my_sl.max()),
            "ndis_per_query": st["ndis"] / float(ses * args.steps),
        },
        # The list scan is bound by HBM: it streams every probed list once per round (shared by all the queries probing it)
        # and writes the distance rows.  achieved = HBM bytes per launch / average launch duration (HIP events on the engine's
        # streams around every scan launch of the timed region; with several batches in flight a launch shares the chip, so
        # this is the contended figure -- one_batch_at_a_time has the kernel alone).
        "roofline": {
            "bound": "hbm",
            "kernel": scan_kernel,
            "achieved": (bytes_per_launch / 1e9) / (a_ms / a_l / 1e3) if a_ms > 0 else None,
            "peak": 8000.0,
            "unit": "GB/s",
            "measured": "one batch at a time, same run, after the timed region" if solo else "timed region (one batch at a time)",
            "traffic": traffic,
            "traffic_source": traffic_source if traffic is not None else "none committed for this workload: achieved uses min_bytes_per_launch",
            "min_bytes_per_launch": alone["scan_min_bytes"] / a_l,
            "avg_launch_ms": a_ms / a_l,
            "launches_per_step": scan_launches / args.steps,
            "in_flight": {"batches": nfl, "avg_launch_ms": scan_ms / launches,
                          "scan_share_of_hbm_peak_over_timed_region": (min_bytes / 1e9) / elapsed / 8000.0},
            # SURVEY 8(d)'s streaming model (one fp32 row per distance the reference computes): far above the HBM peak because a
            # list fetched once serves every query probing it, and is stored as bytes
            "algorithmic_bytes_per_launch": alg_bytes / launches,
            "algorithmic_frac": ((alg_bytes / launches / 1e9) / (a_ms / a_l / 1e3)) / 8000.0 if a_ms > 0 else None,
            "computed_over_algorithmic": scan_bytes / alg_bytes if alg_bytes else None,
            "tile_slot_efficiency": slot_eff,
            "other_kernels_ms_per_step": {"coarse": alone["coarse_ms"] / args.steps, "select": alone["select_ms"] / args.steps},
            # the arithmetic side: distances computed per second against the measured issue rate of the instruction
            # (profiles/r02_ubench_mfma_i8.txt: 15 992 G distances/s at d = 128; fp32: 550 G wave-instructions/s)
            "compute": {
                "unit": "G distances/s",
                "op": "v_mfma_i32_32x32x32_i8" if arith == 2 else "v_pk_fma_f32 / v_pk_mul+add",
                "achieved": (alone["scan_bytes"] / (4.0 * d)) / (a_ms / 1e3) / 1e9 if a_ms > 0 else None,
                "peak": 15992.0 * 128.0 / d if arith == 2 else 550.0 * 64 / (d if arith == 1 else 1.5 * d),
            },
        },
    }
    rf = out["roofline"]
    rf["frac"] = rf["achieved"] / rf["peak"] if rf["achieved"] else None
    # ---- the two scan launches of a step, each by itself (an average over a dense and a threshold launch hides
    # the worse one): ms = HIP events around the launch on the stream it runs on (one batch at a time), bytes = the committed
    # per-dispatch PMC figure of that kernel (FETCH_SIZE x 2 for its 16-byte-per-lane streams + WRITE_SIZE) when the profile
    # was taken on this workload, else the engine's lower bound for the round (every probed list once + rows / mask bits written)
    pipe = int(h.get_option("scan_pipelined")) if arith == 2 else 0
    ksn = (d + 31) // 32
    knames = {"scan_dense": (f"scan_mfma_thr_kernel<1, {ksn}, true>" if pipe & 1 else f"scan_mfma_kernel<1, false, {ksn}>") if arith == 2 else "scan_lanes_kernel",
              "scan_thr": (f"scan_mfma_pair_kernel<1, {ksn}>" if pipe & 4 else f"scan_mfma_thr_kernel<1, {ksn}, false>" if pipe & 2
                           else f"scan_mfma_kernel<1, true, {ksn}>") if arith == 2 else "scan_filter_kernel"}
    prof = {}
    try:
        if cands:
            pj2 = json.load(open(cands[-1]))
            if pj2.get("_workload") == workload:
                prof = pj2
    except Exception:  # noqa: BLE001
        prof = {}
    phases = alone.get("phases", {})
    per_launch = []
    for ph, kind in (("scan_dense", "dense round (round 0: every distance stored)"), ("scan_thr", "threshold rounds (mask bits + the distances that beat the threshold)")):
        ms_l, n_l = phases.get(ph, (0.0, 0.0))
        if not n_l:
            continue
        lb = phases.get("min_bytes_dense" if ph == "scan_dense" else "min_bytes_thr", 0.0) / n_l
        pm = prof.get(knames[ph], {})
        by = pm.get("hbm_bytes_per_dispatch")
        per_launch.append({"kernel": knames[ph], "what": kind, "launches_per_step": n_l / args.steps, "ms": ms_l / n_l,
                           "bytes": by if by else lb, "bytes_source": (os.path.relpath(cands[-1], ROOT) + " (PMC per dispatch)") if by else "engine lower bound",
                           "min_bytes": lb, "GBps": (by if by else lb) / 1e9 / (ms_l / n_l / 1e3),
                           "frac": (by if by else lb) / 1e9 / (ms_l / n_l / 1e3) / 8000.0})
    rf["per_launch"] = per_launch
    # every phase of a step, one batch at a time (events on the phases' own streams; tie_fix_kernel runs beside the next round)
    rf["phases_ms_per_step"] = {k: v[0] / args.steps for k, v in phases.items() if isinstance(v, tuple)}
    rf["phase_launches_per_step"] = {k: v[1] / args.steps for k, v in phases.items() if isinstance(v, tuple)}
    # the whole timed region against the HBM peak: the bytes a step moves through HBM by the committed PMC profile (all kernels) /
    # the step time of the TIMED region
    if prof.get("_hbm_bytes_per_step"):
        rf["step_hbm_bytes"] = prof["_hbm_bytes_per_step"]
        rf["step_hbm_frac"] = prof["_hbm_bytes_per_step"] / 1e9 / (elapsed / args.steps) / 8000.0
    cp = rf["compute"]
    cp["frac"] = cp["achieved"] / cp["peak"] if cp["achieved"] else None
    if solo:
        s_ms, s_l = solo["scan_ms"], max(solo["scan_launches"], 1)
        s_bytes = traffic if traffic is not None else solo["scan_min_bytes"] / s_l
        out["one_batch_at_a_time"] = {
            "value": ses * args.steps / solo["elapsed"], "unit": "queries/s", "ms_per_step": 1000.0 * solo["elapsed"] / args.steps,
            "scan_avg_launch_ms": s_ms / s_l,
            "scan_hbm_GBps": (s_bytes / 1e9) / (s_ms / s_l / 1e3), "scan_frac_of_hbm_peak": (s_bytes / 1e9) / (s_ms / s_l / 1e3) / 8000.0,
            "scan_G_distances_per_s": (solo["scan_bytes"] / (4.0 * d)) / (s_ms / 1e3) / 1e9,
            "other_kernels_ms_per_step": {"coarse": solo["coarse_ms"] / args.steps, "select": solo["select_ms"] / args.steps},
        }
    if single_caller is not None:
        out["single_caller_async"] = single_caller
    if lat1 is not None:
        out["latency_batch1"] = lat1
    if fp32 is not None:
        out["fp32_path"] = fp32
        # the reference's own arithmetic as first-class fields: the same workload with byte codes off (what float data gets)
        out["value_fp32"] = fp32["value"]
        out["roofline_fp32"] = fp32.get("roofline")
    if id_ties is not None:
        out["centroid_number_tie_order"] = id_ties
    if fixed32 is not None:
        out["fixed_nprobe_32"] = fixed32
    # What the round schedule scans beyond what the rule needed, by probe counts (list lengths average out): round 0 runs probes
    # [0, first) of every query, round 1 [first, min(first * grow, ...)) of the queries that did not stop in round 0 (none of them
    # has fired at multipler 1: the rule fires at the stage it stops at), a third round whatever is left
    first_r = int(h.get_option("round_first")) if h.get_option("round_first") > 0 else 12
    grow_r = h.get_option("round_grow") if h.get_option("round_grow") > 0 else 12.0
    e0, e1 = first_r, int(max(first_r * grow_r, first_r + 12))
    npq = my_sl.astype(np.int64)
    need = [np.minimum(npq, e0).sum(), np.clip(npq - e0, 0, e1 - e0).sum(), np.clip(npq - e1, 0, None).sum()]
    comp = [e0 * len(npq), int((npq > e0).sum()) * (e1 - e0), np.clip(npq - e1, 0, None).sum()]
    out["roofline"]["computed_over_algorithmic_by_round"] = {
        "by": "probe counts of the last timed slice (my_nprobe per query against the round schedule)",
        "rounds": [{"probes": [0, e0], "scanned": int(comp[0]), "needed": int(need[0]), "ratio": float(comp[0] / max(need[0], 1))},
                   {"probes": [e0, e1], "scanned": int(comp[1]), "needed": int(need[1]), "ratio": float(comp[1] / max(need[1], 1))},
                   {"probes": [e1, "my_nprobe"], "scanned": int(comp[2]), "needed": int(need[2]), "ratio": 1.0}]}
    out["guaranteed_bound_point"] = guar

    # ---- CPU baseline and parity: the compiled reference (and the pinned CPU restatement) on EVERY resident slice the timed region
    # searched -- all nsl x ses queries -- against the engine in the timed configuration (there is only one: --coarse-ties redo)
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle
        S = nsl * ses
        try:
            log(f"host: os.cpu_count {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, cgroup cpu.max {open('/sys/fs/cgroup/cpu.max').read().strip()}")
        except Exception:  # noqa: BLE001
            pass
        t0 = time.time()
        # the engine's results for every slice, one call per slice as in the timed region
        gD = np.empty((S, K), dtype=np.float32)
        gI = np.empty((S, K), dtype=np.int64)
        g_np = np.zeros(nall, dtype=np.uint64)
        g_tr = np.zeros(nall, dtype=np.float32)
        patched = redone = 0
        for sl in range(nsl):
            D_, I_ = h.search_adaptive(ts + sl * ses, ses, topk, chosen, args.std_m, req, g_np, g_tr)
            gD[sl * ses:(sl + 1) * ses], gI[sl * ses:(sl + 1) * ses] = D_, I_
            patched += h.last_tie_patched()
            redone += h.last_tie_redone()
        g_np = g_np[ts:ts + S].copy()
        codes, ids, off = [], [], np.zeros(nlist + 1, dtype=np.uintp)
        for l in range(nlist):
            c, i_ = h.get_list(l)
            codes.append(c)
            ids.append(i_)
            off[l + 1] = off[l] + len(i_)
        lists = pyoracle.Lists.__new__(pyoracle.Lists)
        lists.metric, lists.centroids, lists.nlist, lists.d = pyoracle.METRIC_L2, cen, nlist, d
        lists.off, lists.codes, lists.ids = off, np.concatenate(codes), np.concatenate(ids)
        del codes, ids
        lists.struct = pyoracle.OrcIndex(lists.metric, d, nlist, pyoracle._s(lists.off), pyoracle._f(lists.codes), pyoracle._i(lists.ids))
        cores = host_cores()  # threads beyond a CPU quota only get throttled
        xs = xq[ts:ts + S]
        tun = pyoracle.Tuner(h.get_interdis(), traces, K, nall, arcos=capi.arcos_table())
        stt = tun.struct(topk, req, chosen, args.std_m)
        tc = time.perf_counter()
        cd, ck = pyoracle.knn(pyoracle.METRIC_L2, xs, cen, nlist, nthreads=cores)
        oD, oI, _ = pyoracle.search_preassigned(lists, xs, K, ck, cd, tuner=stt, offset=ts, nthreads=cores)
        cpu_s = time.perf_counter() - tc
        del cd, ck

        def differing(rD, rI, rnp):
            return int(((rI != gI).any(1) | (rD != gD).any(1) | (rnp != g_np)).sum())

        port_np = tun.my_nprobe[ts:ts + S].astype(np.uint64)
        port_diff = differing(oD, oI, port_np)

        def lat_check(rD, rI, rnp):
            """the one-query-per-call results against the reference's for the same queries"""
            if lat_keep is None:
                return None
            ids_, lD_, lI_, lnp_ = lat_keep
            rows = ids_ - ts
            bad = (lI_ != rI[rows]).any(1) | (lD_.view(np.uint32) != np.ascontiguousarray(rD[rows]).view(np.uint32)).any(1) | (lnp_ != rnp[rows])
            return {"calls_checked": int(len(ids_)), "calls_differing": int(bad.sum())}

        def slices_of(rD, rI, rnp, cols=K):
            return lambda start: (rD[start - ts:start - ts + ses, :cols], rI[start - ts:start - ts + ses, :cols],
                                  None if rnp is None else rnp[start - ts:start - ts + ses])

        # ... and what the TIMED searches themselves returned (six in flight: StepResults), step by step, against the same results
        timed_par = dict(kept_timed.check(slices_of(oD, oI, port_np)), against="CPU restatement (pinned)")
        fp32_par = dict(kept_fp32.check(slices_of(oD, oI, port_np)), against="CPU restatement (pinned)") if kept_fp32 is not None else None
        fixed_par = None
        if kept_fixed is not None:
            fcd, fck = pyoracle.knn(pyoracle.METRIC_L2, xs, cen, fixed32["nprobe"], nthreads=cores)
            pfD, pfI, _ = pyoracle.search_preassigned(lists, xs, fixed32["k"], fck, fcd, nthreads=cores)
            fixed_par = dict(kept_fixed.check(slices_of(pfD, pfI, None, fixed32["k"])), against="CPU restatement (pinned)")
            del fcd, fck
        parity = {"regime": {2: "coarse_ties = 2: the reference's order inside runs of bit-equal coarse distances for every query, the heap's order "
                                "patched into the one pass (include/auncel_amd.h)", 1: "coarse_ties = 1", 0: "coarse_ties = 0 (centroid-number order)"}[ties_opt],
                  "queries": S, "slices": nsl,
                  "rankings_the_heap_changed": int(patched), "queries_searched_again": int(redone),
                  # the results of the timed steps themselves (every step of the timed region: D, I and my_nprobe as the searches in flight
                  # returned them) and of the legs' steps, against the reference's result for the slice each searched
                  "timed_steps_checked": timed_par["timed_steps_checked"], "timed_steps_differing": timed_par["timed_steps_differing"],
                  "timed_region": timed_par, "fp32_path_steps": fp32_par, "fixed_nprobe_32_steps": fixed_par,
                  "latency_batch1_calls": lat_check(oD, oI, port_np)}
        if port_diff:
            log("PARITY MISMATCH vs the CPU restatement:", port_diff, "of", S, "queries differ in I / D / my_nprobe")
        log(f"parity vs the CPU restatement on {S} queries ({nsl} slices): {port_diff} differ; {patched} rankings changed by the heap, {redone} queries searched again")
        # one thread: what the shipped reference does -- its IndexIVF.cpp cannot be built with OpenMP (Auncel/IndexIVF.cpp:484-486)
        # and eval/bound.cpp issues one search() per query
        S1 = min(64, S)
        tun1 = pyoracle.Tuner(h.get_interdis(), traces, K, nall, arcos=capi.arcos_table())
        st1 = tun1.struct(topk, req, chosen, args.std_m)
        t1 = time.perf_counter()
        cd1, ck1 = pyoracle.knn(pyoracle.METRIC_L2, xs[:S1], cen, nlist, nthreads=1)
        pyoracle.search_preassigned(lists, xs[:S1], K, ck1, cd1, tuner=st1, offset=ts, nthreads=1)
        cpu1_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": S / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": f"all {S} resident queries of the timed region ({nsl} slices of {ses}), same index, coarse + adaptive scan, OpenMP over queries",
                               "gpu_matches_cpu_on_sample": port_diff == 0, "queries_differing": port_diff, "parity": parity,
                               "one_thread": {"value": S1 / cpu1_s, "unit": "queries/s", "cores": 1, "sample": f"first {S1} of the timed queries"}}
        log(f"cpu baseline: {S / cpu_s:.1f} q/s on {cores} threads, {S1 / cpu1_s:.1f} q/s on one (setup {time.time() - t0:.1f}s); "
            f"parity on the sample: {port_diff == 0}")
        # The compiled reference itself (oracle/_ref/ref_harness, built from /root/reference where that exists; the binary
        # travels, the sources do not): same lists, centroids and traces, eval/bound.cpp's one-search-per-query loop on one
        # thread -- as the reference runs -- and the same calls spread over the host cores.
        from oracle import refbench
        if refbench.available() and not args.no_ref:
            try:
                t0 = time.time()
                ro = refbench.run(cen, lists.off, lists.codes, lists.ids, traces, xs, ts, K, topk, args.bound, chosen, args.std_m,
                                  single_thread_queries=S1, threads=cores)
                rnp = ro["my_nprobe"].astype(np.uint64)
                ref_diff = differing(ro["D"], ro["I"], rnp)
                against = "compiled reference (oracle/_ref/ref_harness)"
                timed_par = dict(kept_timed.check(slices_of(ro["D"], ro["I"], rnp)), against=against)
                parity.update({"timed_steps_checked": timed_par["timed_steps_checked"], "timed_steps_differing": timed_par["timed_steps_differing"],
                               "timed_region": timed_par})
                parity["latency_batch1_calls"] = lat_check(ro["D"], ro["I"], rnp)
                if kept_fp32 is not None:
                    parity["fp32_path_steps"] = dict(kept_fp32.check(slices_of(ro["D"], ro["I"], rnp)), against=against)
                if kept_fixed is not None:
                    rf_ = refbench.run_fixed(pyoracle.METRIC_L2, cen, lists.off, lists.codes, lists.ids, xs, fixed32["k"], fixed32["nprobe"], threads=cores)
                    parity["fixed_nprobe_32_steps"] = dict(kept_fixed.check(slices_of(rf_["D"], rf_["I"], None, fixed32["k"])), against=against,
                                                           reference_qps=S / rf_["seconds_all_threads"])
                port = out["cpu_baseline"]
                out["cpu_baseline"] = {
                    "value": S / ro["seconds_all_threads"], "unit": "queries/s", "cores": ro["threads"], "kind": "reference",
                    "sample": f"all {S} resident queries of the timed region ({nsl} slices of {ses}); the compiled reference (Auncel/*.cpp, -O3 -msse4) on the "
                              "engine's lists / centroids / traces, one IndexIVF::search(1, ...) per query in tune mode, OpenMP over queries",
                    "gpu_matches_cpu_on_sample": ref_diff == 0, "queries_differing": ref_diff, "parity": parity,
                    "one_thread": {"value": ro["queries_one_thread"] / ro["seconds_one_thread"], "unit": "queries/s", "cores": 1,
                                   "sample": f"first {ro['queries_one_thread']} of the timed queries, Error_sys::search(D, I, i, 1) per query "
                                             "(eval/bound.cpp:380-386) -- what the shipped reference does"},
                    "port": {k: port[k] for k in ("value", "cores", "gpu_matches_cpu_on_sample", "queries_differing", "one_thread")},
                }
                log("timed steps against the reference:", json.dumps({k: parity[k] for k in ("timed_region", "fp32_path_steps", "fixed_nprobe_32_steps")}))
                log(f"reference on the host: {S / ro['seconds_all_threads']:.1f} q/s on {ro['threads']} threads, "
                    f"{ro['queries_one_thread'] / ro['seconds_one_thread']:.1f} q/s on one ({time.time() - t0:.1f}s incl. hand-over); "
                    f"GPU == reference on all {S} queries: {ref_diff == 0} ({ref_diff} differ)")
            except Exception as e:  # noqa: BLE001 -- the port's figures stay
                log("reference harness not usable here:", repr(e))
    if gt_check is not None:
        out["config"]["ground_truth_file"] = gt_check
    # ---- BASELINE configs 1, 3, 5 under the same clock: fixed nprobe, one batch of 10 000 resident queries per
    # call, value / per-phase times / roofline of the scan, and ids + distances compared bit for bit with the compiled reference
    # (oracle/_ref/ref_harness fixedbench, 2000 queries each) and the pinned CPU restatement.  After the headline: never part of it.
    if rank == 0 and world == 1 and not args.no_legs and not args.no_other and not args.data:
        try:
            while len(ctxs) > 1:  # (the search contexts go before the index they were cloned from)
                ctxs.pop()
            del ctxs, h
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_configs
            oc = []
            for c, npb in (("1", (8,)), ("3", (32,)), ("5", (32,))):
                t0 = time.time()
                for line in bench_configs.run_config(torch, capi, dev, c, npb, 64, 0 if args.no_cpu or args.no_ref else 2000, log):
                    oc.append(line)
                    log(f"other config {c} nprobe {line['nprobe']}: {line['value'] / 1e6:.3f} M q/s one call at a time ({(line.get('in_flight') or {}).get('value', 0) / 1e6:.3f} with "
                        f"{(line.get('in_flight') or {}).get('searches_at_a_time')} at a time), {line['ms_per_batch']:.2f} ms per batch of {line['batch']}, "
                        f"recall@{line['k']} {line['recall_at_k']:.4f}, == reference: {line['gpu_equals_reference']}, == CPU restatement: "
                        f"{line['gpu_equals_cpu_on_sample']} ({time.time() - t0:.0f}s)")
            out["other_configs"] = oc
        except Exception as e:  # noqa: BLE001 -- the headline stands without them
            log("other_configs failed:", repr(e))
            out["other_configs"] = {"error": repr(e)}
        # ---- the same pipeline on data where the reference's own acceptance check holds (scripts/guaranteed_workload.py): the
        # headline's data never satisfies it (bound_guaranteed false at every grid point), so the q/s at which Auncel's actual claim --
        # every query within the error bound -- is met on this engine is measured on tighter blobs, next to the headline
        try:
            import guaranteed_workload
            t0 = time.time()
            gw = guaranteed_workload.run(torch, capi, dev, log, sigma=20.0, nb=args.nb, d=d, nlist=nlist, blobs=args.blobs, K=K, topk=topk, bound=0.9,
                                         ts=ts, ses=ses, steps=max(12, args.steps // 2), in_flight=nfl, check_parity=not args.no_cpu)
            out["guaranteed_bound_workload"] = gw
            log(f"guaranteed-bound workload: guaranteed {gw.get('bound_guaranteed')} at multipler {gw.get('multipler')}: {gw.get('value', 0) / 1e6:.3f} M q/s "
                f"({time.time() - t0:.0f}s)")
        except Exception as e:  # noqa: BLE001
            log("guaranteed_bound_workload failed:", repr(e))
            out["guaranteed_bound_workload"] = {"error": repr(e)}
    if world > 1:
        # north_star's split next to the replicas: the inverted lists sharded by list id over the same N GPUs (BASELINE config 4,
        # Auncel/IndexShards.cpp:261-311), fixed nprobe, run after the replica timing; one driver command records both
        while len(ctxs) > 1:  # (the search contexts go before the index they were cloned from)
            ctxs.pop()
        del ctxs, h
        torch.cuda.empty_cache()
        sargs = argparse.Namespace(**vars(args))
        sargs.test = 2 * ses  # the reference's 10 000-query batch
        sl = run_shards(sargs, torch, dist, capi, rank, world, local, dev, red_dev)
        if rank == 0 and sl is not None:
            out["shards"] = {"metric": sl["metric"], "value": sl["value"], "unit": sl["unit"], "ms_per_step": sl["ms_per_step"],
                             "single_index_distances_sha256": sl["config"].get("single_index_distances_sha256"),
                             "shards_equal_single_index": sl["config"].get("shards_equal_single_index"),
                             "scaling": sl["scaling"], "n_gpus": sl["n_gpus"], "nprobe": sargs.nprobe, "batch": sargs.test,
                             "distances_sha256": sl["config"]["distances_sha256"], "recall_at_k_mean": sl["config"]["recall_at_k_mean"],
                             "ranks_seen": sl["config"]["ranks_seen"],
                             "per_rank_ms_per_step": sl["config"]["per_rank_ms_per_step"],
                             "shard_bytes_max_over_min": sl["config"]["shard_bytes_max_over_min"], "roofline": sl["roofline"]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and world > 1 and out.get("shards", {}).get("shards_equal_single_index") is False:
        sys.exit(3)
    # a timed step (or a leg's step) whose result is not the reference's is not a measurement
    cb = out.get("cpu_baseline") or {}
    par = cb.get("parity") or {}
    bad = int(cb.get("queries_differing") or 0) + sum(int((par.get(k) or {}).get("timed_steps_differing") or 0)
                                                        for k in ("timed_region", "fp32_path_steps", "fixed_nprobe_32_steps"))
    bad += int((par.get("latency_batch1_calls") or {}).get("calls_differing") or 0)
    gw = out.get("guaranteed_bound_workload") or {}
    bad += int((gw.get("parity") or {}).get("timed_steps_differing") or 0)
    if rank == 0 and bad:
        log("PARITY FAILURE:", bad, "differences between the engine's results and the reference's (see cpu_baseline.parity)")
        sys.exit(4)


if __name__ == "__main__":
    main()
