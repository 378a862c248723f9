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
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ref", action="store_true", help="cpu_baseline from the CPU restatement only (skip oracle/_ref/ref_harness)")
    ap.add_argument("--kmeans", choices=["engine", "torch"], default="engine",
                    help="coarse centroids: the engine's Clustering::train restatement (amd_ivf_kmeans, the reference's IVF "
                         "training: 25 iterations, 256 points per centroid) or a 4-step torch Lloyd")
    ap.add_argument("--stagger-ms", type=float, default=0.8,
                    help="context j issues its first step j x this many ms after context 0 (inside the timed region): out of phase, the "
                         "scan of one batch runs under the selection of another; started together they tend to stay in step")
    ap.add_argument("--pinned-out", type=int, default=1, help="1: result buffers in page-locked host memory, 0: pageable")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="batches kept in flight per GPU, each from its own host thread on its own search context "
                         "(amd_ivf_clone); 1 = one batch at a time")
    args = ap.parse_args()
    # hardware queues the HIP runtime maps its streams onto (ROCm's default, pinned here because the figure is sensitive to it:
    # with 8 queues four batches in flight lose 8 % and six collapse, profiles/r01_in_flight_sweep.txt); must precede HIP start-up
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"
    # Host waits: every batch in flight has a host thread waiting on its stream between rounds.  Spinning waits (the HIP default)
    # are ~2 % faster but need a core each; when the ranks' threads outnumber the cores this process tree may use (cgroup quota),
    # the engine sleeps on blocking events instead (include/auncel_amd.h: AUNCEL_AMD_BLOCKING_SYNC).
    if host_cores() < world * (args.in_flight + 1):
        os.environ.setdefault("AUNCEL_AMD_BLOCKING_SYNC", "1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the max-over-ranks reduction lives

    d, nlist, K, topk, ts, ses = args.d, args.nlist, args.maxtopk, args.topk, args.train, args.test
    t0 = time.time()
    xb_t, _, draw = gen_data(torch, dev, args.nb, 0, d, args.blobs, args.sigma, 1235)
    gq = torch.Generator(device=dev)
    gq.manual_seed(777 + rank)  # every replica searches its own query set
    xq_t = draw(ts + ses, gq)
    cen_t = kmeans_centroids(torch, xb_t, nlist, 4, 99)
    gtD, gtI = ground_truth(torch, xb_t, xq_t, K)
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
    sizes = np.array([h.list_size(l) for l in range(nlist)])
    log(f"index build: {time.time() - t0:.1f}s; list size mean {sizes.mean():.0f} max {sizes.max()} empty {(sizes == 0).sum()}")

    # ---- offline trace training (Error_sys::sys_train)
    t0 = time.time()
    ntr = 0
    while (1 << ntr) <= nlist // 8:
        ntr += 1
    raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
    h.train_samples(0, ts, K, gtD, ts, raw)
    traces = [capi.trace_sb(r) for r in raw]
    h.set_tuner(K, traces, capi.arcos_table())
    log(f"trace training: {time.time() - t0:.1f}s; bins per trace {[len(t[0]) for t in traces]}")

    # ---- hyper-parameters: the reference ships hand-tuned (multipler, std_m) rows for IVF1024 only (hyperparameter.txt).
    # Here: the first pair, from the most aggressive on, that holds the bound on the training half and also on the timed
    # half -- the metric is quoted AT recall@10 >= bound.  std_m scales the spread term of the k-scaling estimate
    # (Trace::search: mean + std_m * std), multipler the probe count at which a fired query stops.
    req = np.full(ts + ses, args.bound, dtype=np.float32)
    grid = [(1.0, sm) for sm in (0.0, 0.25, 0.5, 0.75) if sm < args.std_m]
    grid += [(m, args.std_m) for m in (1.0, 1.25, 1.5, 1.75, 2.0, 2.5, 3.0, 4.0, 5.0, 6.0, 8.0, 12.0)]
    chosen, chosen_std = grid[-1]
    for mult, sm in grid:
        np_ = np.zeros(ts + ses, dtype=np.uint64)
        tr_ = np.zeros(ts + ses, dtype=np.float32)
        D, I = h.search_adaptive(0, ts, topk, mult, sm, req, np_, tr_)
        rec = recall_dist(D, gtD[:ts], topk)
        D2, _ = h.search_adaptive(ts, ses, topk, mult, sm, req, np_, tr_)
        rec2 = recall_dist(D2, gtD[ts:], topk)
        log(f"  multipler {mult} std_m {sm}: recall@{topk} train {rec.mean():.4f} test {rec2.mean():.4f} nprobe mean {np_[:ts].mean():.1f}")
        if rec.mean() >= args.bound and rec2.mean() >= args.bound:
            chosen, chosen_std = mult, sm
            break
    args.std_m = chosen_std

    # ---- timed region: K steps over the resident test batch.  A step is one search_adaptive call over the whole batch;
    # with --in-flight N the K calls are issued from N host threads, each on its own search context over the same
    # device-resident index, so that one batch's latency-bound phases (ordered selection, round planning) overlap
    # another batch's scans.  Every step is a complete, independent search either way.
    import threading
    nfl = max(1, min(args.in_flight, args.steps))
    ctxs = [h] + [h.clone() for _ in range(nfl - 1)]
    for c in ctxs[1:]:
        c.set_queries(xq)

    stagger_s = float(os.environ.get("BENCH_STAGGER_MS", args.stagger_ms)) / 1e3

    # result buffers of every context: page-locked host memory (the caller's choice in the reference's API too), so that
    # the 6 MB of (D, I) of a step come back by direct DMA instead of through the runtime's staging copies
    outs = {}
    for c in ctxs:
        if args.pinned_out:
            outs[id(c)] = (torch.empty((ses, K), dtype=torch.float32).pin_memory().numpy(),
                           torch.empty((ses, K), dtype=torch.int64).pin_memory().numpy())
        else:
            outs[id(c)] = None

    def step(ctx):
        np_ = np.zeros(ts + ses, dtype=np.uint64)
        tr_ = np.zeros(ts + ses, dtype=np.float32)
        D, I = ctx.search_adaptive(ts, ses, topk, chosen, args.std_m, req, np_, tr_, out=outs[id(ctx)])
        return D, I, np_

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(nsteps, acc):
        """steps j, j + nfl, ... on context j; acc collects per-step kernel timings and the last result"""
        errs = []

        def worker(j):
            try:
                if j and stagger_s:
                    time.sleep(j * stagger_s)  # start the contexts out of phase (scan of one under selection of the other)
                for _ in range(j, nsteps, nfl):
                    res = step(ctxs[j])
                    tm = ctxs[j].last_timing()
                    with lock:
                        for key in ("scan_ms", "scan_bytes", "scan_launches", "coarse_ms", "select_ms"):
                            acc[key] = acc.get(key, 0.0) + tm[key]
                        acc["slot_eff"] = acc.get("slot_eff", 0.0) + tm["slot_efficiency"]
                        acc["last"] = res
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
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, acc)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = {}
    for c in ctxs:
        for key, val in c.stats().items():
            st[key] = st.get(key, 0) + val
    # the same steps one batch at a time (after the timed region, not part of `value`): per-launch kernel figures without
    # another batch sharing the chip
    solo = {}
    if nfl > 1:
        nfl_keep, nfl = nfl, 1
        barrier()
        ts0 = time.perf_counter()
        run_steps(args.steps, solo)
        barrier()
        solo["elapsed"] = time.perf_counter() - ts0
        nfl = nfl_keep
    D, I, my_np = acc["last"]
    scan_ms, scan_bytes, scan_launches = acc["scan_ms"], acc["scan_bytes"], acc["scan_launches"]
    coarse_ms, select_ms, slot_eff = acc["coarse_ms"], acc["select_ms"], acc["slot_eff"] / args.steps

    rec = recall_dist(D, gtD[ts:], topk)
    log("my_nprobe of the timed queries: percentiles 10/25/50/75/90/95/99 =", np.percentile(my_np[ts:], [10, 25, 50, 75, 90, 95, 99]).tolist(),
        "; share <= 12:", float((my_np[ts:] <= 12).mean()), "<= 42:", float((my_np[ts:] <= 42).mean()))
    # algorithmic bytes: the reference's own ndis counter (codes actually visited by the probe loops,
    # IndexIVF.cpp:676,733) x d x 4; `scan_bytes` (distances the tiles computed, incl. the probes a round
    # ran past a query's stop point) is reported beside it as computed_over_algorithmic
    alg_bytes = float(st["ndis"]) * d * 4.0
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
        # arithmetic of the dominant kernel: byte codes and integer dot products when the data is uint8-valued (bit-identical
        # to the reference's fp32 results there), fp32 in the reference's summation order otherwise
        "dtype": "u8" if h.scan_arith() == 2 else "f32",
        "data": "synthetic",
        "config": {
            "workload": f"SIFT-{args.nb // 1000000}M-like d={d} IVF{nlist},Flat max_topk={K} topk={topk} Auncel error-bound nprobe "
                        f"(bound {args.bound}), batch {ses} resident queries per GPU, index replicated per GPU",
            "in_flight": nfl, "host_wait": "blocking events" if os.environ.get("AUNCEL_AMD_BLOCKING_SYNC", "0") not in ("", "0") else "spin", "host_cores": host_cores(), "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "stagger_ms": stagger_s * 1e3, "scan_arith": {0: "fp32 reference order", 1: "fp32 fused", 2: "byte codes, v_dot4_u32_u8"}[h.scan_arith()],
            "nb": args.nb, "sigma": args.sigma, "multipler": chosen, "std_m": args.std_m,
            "recall_at_10_mean": float(rec.mean()), "recall_at_10_min": float(rec.min()),
            "nprobe_mean": float(my_np[ts:].mean()), "nprobe_max": int(my_np[ts:].max()),
            "ndis_per_query": st["ndis"] / float(ses * args.steps),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": (alg_bytes / 1e9) / (scan_ms / 1e3) if scan_ms > 0 else None,
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": ((alg_bytes / 1e9) / (scan_ms / 1e3)) / 8000.0 if scan_ms > 0 else None,
            "traffic": None,
            "kernel": "scan_tiles_kernel",
            "avg_launch_ms": scan_ms / max(scan_launches, 1),
            "algorithmic_bytes_per_launch": alg_bytes / max(scan_launches, 1),
            "computed_over_algorithmic": scan_bytes / alg_bytes if alg_bytes else None,
            "launches_per_step": scan_launches / args.steps,
            "tile_slot_efficiency": slot_eff,
            "other_kernels_ms_per_step": {"coarse": coarse_ms / args.steps, "select": select_ms / args.steps},
            # The lists are shared by the queries of a round, so the kernel is not HBM-bound (traffic << algorithmic bytes):
            # its own ceiling is the VALU issue rate.  Byte codes: d/4 v_dot4_u32_u8 per distance; scratch/ubench/dot4_rate.hip
            # measures 550 G wave-instructions/s for it on this chip (1024 SIMDs x 2.15 GHz / 4), i.e. 550e9 * 64 / (d/4)
            # distances/s.  `achieved` counts every distance the tiles computed (scan_bytes / (4 d)).
            "compute": {
                "bound": "valu", "op": "v_dot4_u32_u8" if h.scan_arith() == 2 else "v_pk_fma_f32 / v_pk_mul+add",
                "unit": "G distances/s",
                "achieved": (scan_bytes / (4.0 * d)) / (scan_ms / 1e3) / 1e9 if scan_ms > 0 else None,
                "peak": 550.0 * 64 / ((d / 4.0) if h.scan_arith() == 2 else (d if h.scan_arith() == 1 else 1.5 * d)),
            },
        },
    }

    cp = out["roofline"]["compute"]
    cp["frac"] = cp["achieved"] / cp["peak"] if cp["achieved"] else None
    if solo:
        s_ms = solo["scan_ms"]
        out["one_batch_at_a_time"] = {
            "value": ses * args.steps / solo["elapsed"], "unit": "queries/s", "ms_per_step": 1000.0 * solo["elapsed"] / args.steps,
            "scan_avg_launch_ms": s_ms / max(solo["scan_launches"], 1),
            "scan_algorithmic_GBps": (alg_bytes / 1e9) / (s_ms / 1e3),
            "scan_G_distances_per_s": (solo["scan_bytes"] / (4.0 * d)) / (s_ms / 1e3) / 1e9,
            "scan_frac_of_valu_peak": (solo["scan_bytes"] / (4.0 * d)) / (s_ms / 1e3) / 1e9 / cp["peak"],
            "other_kernels_ms_per_step": {"coarse": solo["coarse_ms"] / args.steps, "select": solo["select_ms"] / args.steps},
        }

    # HBM traffic of the scan per launch: PMC counters cannot be read from inside this process; the figure comes
    # from the committed rocprofv3 --pmc passes over this same command (profiles/collect.sh -> summarize.py),
    # FETCH_SIZE doubled for the scan's wide loads as MI355X_MICROARCH.md prescribes, and only when that profile
    # was taken on this workload
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
        if cands:
            pj = json.load(open(cands[-1]))
            if pj.get("_workload") == out["config"]["workload"]:
                out["roofline"]["traffic"] = pj["scan_tiles_kernel [lists]"]["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = os.path.relpath(cands[-1], ROOT)
    except Exception as e:  # a missing or stale profile leaves traffic null
        log("no PMC traffic figure:", e)

    # ---- CPU baseline: the pinned CPU restatement of the reference path, all host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle
        S = min(args.cpu_sample, ses)
        try:
            log(f"host: os.cpu_count {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, cgroup cpu.max {open('/sys/fs/cgroup/cpu.max').read().strip()}")
        except Exception:  # noqa: BLE001
            pass
        t0 = time.time()
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
        tun = pyoracle.Tuner(h.get_interdis(), traces, K, ts + ses, arcos=capi.arcos_table())
        stt = tun.struct(topk, req, chosen, args.std_m)
        tc = time.perf_counter()
        cd, ck = pyoracle.knn(pyoracle.METRIC_L2, xs, cen, nlist, nthreads=cores)
        oD, oI, _ = pyoracle.search_preassigned(lists, xs, K, ck, cd, tuner=stt, offset=ts, nthreads=cores)
        cpu_s = time.perf_counter() - tc
        same = bool(np.array_equal(oI, I[:S]) and np.array_equal(oD, D[:S]) and np.array_equal(tun.my_nprobe[ts:ts + S], my_np[ts:ts + S]))
        # one thread: what the shipped reference does -- its IndexIVF.cpp cannot be built with OpenMP (Auncel/IndexIVF.cpp:484-486)
        # and eval/bound.cpp issues one search() per query
        S1 = min(64, S)
        tun1 = pyoracle.Tuner(h.get_interdis(), traces, K, ts + ses, arcos=capi.arcos_table())
        st1 = tun1.struct(topk, req, chosen, args.std_m)
        t1 = time.perf_counter()
        cd1, ck1 = pyoracle.knn(pyoracle.METRIC_L2, xs[:S1], cen, nlist, nthreads=1)
        pyoracle.search_preassigned(lists, xs[:S1], K, ck1, cd1, tuner=st1, offset=ts, nthreads=1)
        cpu1_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": S / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": f"first {S} of the {ses} timed queries, same index, coarse + adaptive scan, OpenMP over queries",
                               "gpu_matches_cpu_on_sample": same,
                               "one_thread": {"value": S1 / cpu1_s, "unit": "queries/s", "cores": 1, "sample": f"first {S1} of the timed queries"}}
        log(f"cpu baseline: {S / cpu_s:.1f} q/s on {cores} threads, {S1 / cpu1_s:.1f} q/s on one (setup {time.time() - t0:.1f}s); "
            f"parity on sample: {same}")
        # The compiled reference itself (oracle/_ref/ref_harness, built from /root/reference where that exists; the binary
        # travels, the sources do not): same lists, centroids and traces, eval/bound.cpp's one-search-per-query loop on one
        # thread -- as the reference runs -- and the same calls spread over the host cores.
        from oracle import refbench
        if refbench.available() and not args.no_ref:
            try:
                t0 = time.time()
                ro = refbench.run(cen, lists.off, lists.codes, lists.ids, traces, xs, ts, K, topk, args.bound, chosen, args.std_m,
                                  single_thread_queries=S1, threads=cores)
                same_ref = bool(np.array_equal(ro["I"], I[:S]) and np.array_equal(ro["D"], D[:S])
                                and np.array_equal(ro["my_nprobe"].astype(np.uint64), my_np[ts:ts + S]))
                port = out["cpu_baseline"]
                out["cpu_baseline"] = {
                    "value": S / ro["seconds_all_threads"], "unit": "queries/s", "cores": ro["threads"], "kind": "reference",
                    "sample": f"first {S} of the {ses} timed queries; the compiled reference (Auncel/*.cpp, -O3 -msse4) on the engine's "
                              "lists / centroids / traces, one IndexIVF::search(1, ...) per query in tune mode, OpenMP over queries",
                    "gpu_matches_cpu_on_sample": same_ref,
                    "one_thread": {"value": ro["queries_one_thread"] / ro["seconds_one_thread"], "unit": "queries/s", "cores": 1,
                                   "sample": f"first {ro['queries_one_thread']} of the timed queries, Error_sys::search(D, I, i, 1) per query "
                                             "(eval/bound.cpp:380-386) -- what the shipped reference does"},
                    "port": {k: port[k] for k in ("value", "cores", "gpu_matches_cpu_on_sample", "one_thread")},
                }
                log(f"reference on the host: {S / ro['seconds_all_threads']:.1f} q/s on {ro['threads']} threads, "
                    f"{ro['queries_one_thread'] / ro['seconds_one_thread']:.1f} q/s on one ({time.time() - t0:.1f}s incl. hand-over); "
                    f"GPU == reference on the sample: {same_ref}")
            except Exception as e:  # noqa: BLE001 -- the port's figures stay
                log("reference harness not usable here:", repr(e))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
