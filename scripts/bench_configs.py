#!/usr/bin/env python3
"""Fixed-nprobe runs of the other BASELINE.json configs (parity-at-scale + throughput), one JSON line each.

  cfg1: SIFT-1M-like  d=128 IVF1024 k=10  nprobe 8          (the reference's own CPU-runnable case)
  cfg3: DEEP-10M-like d=96  IVF4096 k=100 IP, nprobe 16/32/64
  cfg5: GIST-1M-like  d=960 IVF4096 k=10  batch 10000, nprobe 32/64

For every run a sample of the queries is re-searched with the pinned CPU oracle on the same lists and coarse
ranking: ids and distances must be bit-identical (float data included: the scan kernel keeps the reference's
summation order)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def normalize(t):
    return t / t.norm(dim=1, keepdim=True)


def gen(torch, dev, kind, nb, nq):
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    if kind == "sift":
        xb, _, draw = bench.gen_data(torch, dev, nb, 0, 128, max(2000, nb // 500), 35.0, 1234)
        return xb, draw(nq, g)
    if kind == "deep":
        d, blobs = 96, 20000
        c = normalize(torch.randn((blobs, d), generator=g, device=dev))

        def draw(n):
            out = torch.empty((n, d), device=dev)
            for i0 in range(0, n, 1 << 20):
                i1 = min(n, i0 + (1 << 20))
                idx = torch.randint(0, blobs, (i1 - i0,), generator=g, device=dev)
                out[i0:i1] = normalize(c[idx] + torch.randn((i1 - i0, d), generator=g, device=dev) * (0.6 / d ** 0.5))
            return out
        return draw(nb), draw(nq)
    if kind == "gist":
        d, blobs = 960, 2000
        c = torch.rand((blobs, d), generator=g, device=dev) * 0.5

        def draw(n):
            out = torch.empty((n, d), device=dev)
            for i0 in range(0, n, 1 << 17):
                i1 = min(n, i0 + (1 << 17))
                idx = torch.randint(0, blobs, (i1 - i0,), generator=g, device=dev)
                out[i0:i1] = torch.clamp(c[idx] + torch.randn((i1 - i0, d), generator=g, device=dev) * 0.12, 0, 1.5)
            return out
        return draw(nb), draw(nq)
    raise ValueError(kind)


def gt_ids(torch, xb, xq, k, ip):
    nq = xq.shape[0]
    bn = None if ip else (xb * xb).sum(1)
    outI = torch.empty((nq, k), dtype=torch.long, device=xb.device)
    qs = 500
    for q0 in range(0, nq, qs):
        q = xq[q0:q0 + qs]
        best_d = best_i = None
        for b0 in range(0, xb.shape[0], 1 << 20):
            b = xb[b0:b0 + (1 << 20)]
            s = q @ b.T
            dist = -s if ip else (bn[None, b0:b0 + b.shape[0]] - 2 * s)
            cd, ci = dist.topk(k, dim=1, largest=False)
            ci = ci + b0
            if best_d is None:
                best_d, best_i = cd, ci
            else:
                md, mi = torch.cat([best_d, cd], 1), torch.cat([best_i, ci], 1)
                best_d, si = md.topk(k, dim=1, largest=False)
                best_i = mi.gather(1, si)
        outI[q0:q0 + qs] = best_i
    return outI.cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="1,3,5")
    ap.add_argument("--sample", type=int, default=64)
    ap.add_argument("--ref-sample", type=int, default=2000, help="queries the compiled reference runs (oracle/_ref/ref_harness fixedbench)")
    ap.add_argument("--nprobes", default="", help="comma list: only these nprobe values")
    args = ap.parse_args()
    import torch
    from auncel_amd import capi
    from oracle import pyoracle, refbench
    dev = torch.device("cuda", 0)
    cfgs = {"1": ("sift", 1_000_000, 10000, 1024, 10, (8,), capi.METRIC_L2),
            "3": ("deep", 10_000_000, 10000, 4096, 100, (16, 32, 64), capi.METRIC_IP),
            "5": ("gist", 1_000_000, 10000, 4096, 10, (32, 64), capi.METRIC_L2)}
    for c in args.cfg.split(","):
        kind, nb, nq, nlist, k, nprobes, metric = cfgs[c]
        xb_t, xq_t = gen(torch, dev, kind, nb, nq)
        d = xb_t.shape[1]
        gtI = gt_ids(torch, xb_t, xq_t, k, metric == capi.METRIC_IP)
        xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
        del xb_t, xq_t
        torch.cuda.empty_cache()
        # coarse centroids by the reference's IVF training (Clustering::train, 25 iterations) on the GPU
        cen, _ = capi.kmeans(metric, xb, nlist, niter=25)
        h = capi.Handle(d, nlist, metric, 0)
        h.set_centroids(cen)
        if metric == capi.METRIC_IP:
            # add() assigns by the index metric (max inner product), like the reference's quantizer->assign
            h.add(xb)
        else:
            h.add(xb)
        del xb
        h.set_queries(xq)
        # oracle lists for the parity sample
        codes, ids, off = [], [], np.zeros(nlist + 1, dtype=np.uintp)
        for l in range(nlist):
            cc, ii = h.get_list(l)
            codes.append(cc)
            ids.append(ii)
            off[l + 1] = off[l] + len(ii)
        lists = pyoracle.Lists.__new__(pyoracle.Lists)
        lists.metric, lists.centroids, lists.nlist, lists.d = metric, cen, nlist, d
        lists.off, lists.codes, lists.ids = off, np.concatenate(codes), np.concatenate(ids)
        del codes, ids
        lists.struct = pyoracle.OrcIndex(metric, d, nlist, pyoracle._s(lists.off), pyoracle._f(lists.codes), pyoracle._i(lists.ids))
        if args.nprobes:
            nprobes = tuple(int(v) for v in args.nprobes.split(","))
        for nprobe in nprobes:
            h.search_resident(0, nq, k, nprobe)
            best = None
            for _ in range(3):
                h.stats(reset=True)
                t0 = time.perf_counter()
                D, I = h.search_resident(0, nq, k, nprobe)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, h.last_timing(), h.stats())
            dt, tm, st = best
            recall = np.mean([len(set(I[i]) & set(gtI[i])) / k for i in range(0, nq, 10)])
            S = args.sample
            cores = bench.host_cores()
            tc = time.perf_counter()
            cd, ck = pyoracle.knn(metric, xq[:S], cen, nprobe, nthreads=cores)
            oD, oI, _ = pyoracle.search_preassigned(lists, xq[:S], k, ck, cd, nthreads=cores)
            cpu = S / (time.perf_counter() - tc)
            same = bool(np.array_equal(oI, I[:S]) and np.array_equal(oD.view(np.uint32), D[:S].view(np.uint32)))
            # the compiled reference itself, where its harness is present (oracle/_ref): IndexIVF::search, one query per call
            ref = None
            if refbench.available():
                try:
                    SR = min(nq, args.ref_sample)
                    ro = refbench.run_fixed(metric, cen, lists.off, lists.codes, lists.ids, xq[:SR], k, nprobe, threads=cores)
                    ref = {"qps": SR / ro["seconds_all_threads"], "threads": ro["threads"], "queries": SR,
                           "gpu_equals_reference": bool(np.array_equal(ro["I"], I[:SR]) and np.array_equal(ro["D"].view(np.uint32), D[:SR].view(np.uint32)))}
                except Exception as e:  # noqa: BLE001
                    ref = {"error": repr(e)}
            alg = st["ndis"] * d * 4.0
            print(json.dumps({"config": c, "data": kind + "-like synthetic", "nb": nb, "d": d, "nlist": nlist, "k": k, "nprobe": nprobe,
                              "metric": "IP" if metric == 0 else "L2", "batch": nq, "qps": nq / dt, "recall_at_k": float(recall),
                              "scan_ms": tm["scan_ms"], "select_ms": tm["select_ms"], "coarse_ms": tm["coarse_ms"],
                              "scan_algorithmic_GBps": alg / 1e6 / max(tm["scan_ms"], 1e-9), "tile_slot_efficiency": tm["slot_efficiency"],
                              "cpu_oracle_qps": cpu, "cpu_threads": cores, "gpu_equals_cpu_on_sample": same, "reference": ref}), flush=True)
        h.close()
        del lists


if __name__ == "__main__":
    main()
