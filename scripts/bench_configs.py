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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (before anything starts the HIP runtime: auncel_amd/__init__.py)
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


CFGS = {"1": ("sift", 1_000_000, 10000, 1024, 10, (8,), 1),
        "3": ("deep", 10_000_000, 10000, 4096, 100, (16, 32, 64), 0),
        "5": ("gist", 1_000_000, 10000, 4096, 10, (32, 64), 1)}


def run_config(torch, capi, dev, c, nprobes=None, sample=64, ref_sample=2000, log=None, in_flight=3):
    """one BASELINE config (build + timed fixed-nprobe searches + parity against the pinned oracle and, where its harness is
    present, the compiled reference): yields one dict per nprobe"""
    from oracle import pyoracle, refbench
    kind, nb, nq, nlist, k, default_nprobes, metric = CFGS[c]
    nprobes = tuple(nprobes) if nprobes else default_nprobes
    t0 = time.time()
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
    h.add(xb)  # (assigns by the index metric, like the reference's quantizer->assign)
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
    if log:
        log(f"config {c}: data, ground truth, k-means, add: {time.time() - t0:.1f}s")
    for nprobe in nprobes:
        h.search_resident(0, nq, k, nprobe)
        best = None
        for _ in range(3):
            h.stats(reset=True)
            t0 = time.perf_counter()
            D, I = h.search_resident(0, nq, k, nprobe)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, h.last_timing(), h.stats(), h.last_timing_detail())
        dt, tm, st, det = best
        # ... and with searches in flight, as the headline is measured: the same batch submitted again and again through the asynchronous
        # entry points (amd_ivf_submit_search_resident / amd_ivf_wait), `lag` searches at a time, results of every call compared with the
        # synchronous call's
        flight = None
        if in_flight > 1:
            h.set_async_depth(in_flight)
            nrun = 4 * in_flight
            bufs = [(np.empty((nq, k), np.float32), np.empty((nq, k), np.int64)) for _ in range(2 * in_flight)]

            def run(nsteps):
                pend, same_all = [], True
                for sn in range(nsteps):
                    if len(pend) == len(bufs):
                        Dp, Ip, _, _ = h.wait(pend.pop(0))
                        same_all &= bool(np.array_equal(Ip, I) and np.array_equal(Dp.view(np.uint32), D.view(np.uint32)))
                    pend.append(h.submit_search_resident(0, nq, k, nprobe, out=bufs[sn % len(bufs)]))
                while pend:
                    Dp, Ip, _, _ = h.wait(pend.pop(0))
                    same_all &= bool(np.array_equal(Ip, I) and np.array_equal(Dp.view(np.uint32), D.view(np.uint32)))
                return same_all

            run(2 * in_flight)
            t0 = time.perf_counter()
            ok = run(nrun)
            el = time.perf_counter() - t0
            flight = {"searches_at_a_time": in_flight, "value": nq * nrun / el, "unit": "queries/s", "ms_per_batch": 1e3 * el / nrun,
                      "same_results_as_the_synchronous_call": ok}
            h.set_async_depth(0)
        recall = np.mean([len(set(I[i]) & set(gtI[i])) / k for i in range(0, nq, 10)])
        S = sample
        cores = bench.host_cores()
        tc = time.perf_counter()
        cd, ck = pyoracle.knn(metric, xq[:S], cen, nprobe, nthreads=cores)
        oD, oI, _ = pyoracle.search_preassigned(lists, xq[:S], k, ck, cd, nthreads=cores)
        cpu = S / (time.perf_counter() - tc)
        same = bool(np.array_equal(oI, I[:S]) and np.array_equal(oD.view(np.uint32), D[:S].view(np.uint32)))
        # the compiled reference itself, where its harness is present (oracle/_ref): IndexIVF::search, one query per call
        ref = None
        if refbench.available() and ref_sample:
            try:
                SR = min(nq, ref_sample)
                ro = refbench.run_fixed(metric, cen, lists.off, lists.codes, lists.ids, xq[:SR], k, nprobe, threads=cores)
                ref = {"qps": SR / ro["seconds_all_threads"], "threads": ro["threads"], "queries": SR,
                       "gpu_equals_reference": bool(np.array_equal(ro["I"], I[:SR]) and np.array_equal(ro["D"].view(np.uint32), D[:SR].view(np.uint32)))}
            except Exception as e:  # noqa: BLE001
                ref = {"error": repr(e)}
        alg = st["ndis"] * d * 4.0
        arith = h.scan_arith()
        filt = h.last_filter()[0]
        # roofline of the run's scan: the bytes its rounds could not avoid moving (every probed list once per round + rows / mask
        # bits written; dense and threshold rounds apart) over the HIP-event time of their launches
        phases = {p: {"ms": det[p][0], "launches": det[p][1]} for p in capi.Handle.PHASES if det[p][1]}
        roof = {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "traffic": None, "per_launch": []}
        for ph, mb in (("scan_dense", det["min_bytes_dense"]), ("scan_thr", det["min_bytes_thr"])):
            if det[ph][1] and det[ph][0] > 0:
                roof["per_launch"].append({"phase": ph, "launches": det[ph][1], "ms": det[ph][0] / det[ph][1], "min_bytes": mb / det[ph][1],
                                           "GBps": mb / 1e9 / (det[ph][0] / 1e3), "frac": mb / 1e9 / (det[ph][0] / 1e3) / 8000.0})
        # HBM traffic by the committed counter profile of this configuration (profiles/collect_cfg.sh -> profiles/r*_pmc_cfg<c>.json: the
        # kernels' FETCH_SIZE x 2 + WRITE_SIZE over one search of this batch at nprobe 32), per phase
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_cfg{c}.json")))
            if cands and nprobe == 32:
                pj = json.load(open(cands[-1]))
                dense_b = sum(e["hbm_read_bytes_x2"] + e["hbm_write_bytes"] for k_, e in pj.items()
                              if isinstance(e, dict) and k_.startswith(("scan_tiles_kernel", "scan_lanes_kernel")))
                thr_b = sum(e["hbm_read_bytes_x2"] + e["hbm_write_bytes"] for k_, e in pj.items()
                            if isinstance(e, dict) and (k_.startswith("scan_filter") or k_.startswith("rescore_kernel")))
                for pl in roof["per_launch"]:
                    tb = dense_b if pl["phase"] == "scan_dense" else thr_b
                    pl["traffic"] = tb / max(pl["launches"], 1)
                    pl["traffic_frac"] = tb / 1e9 / (pl["ms"] * pl["launches"] / 1e3) / 8000.0 if pl["ms"] else None
                roof["traffic"] = (dense_b + thr_b) / max(sum(pl["launches"] for pl in roof["per_launch"]), 1)
                roof["traffic_source"] = os.path.relpath(cands[-1], ROOT) + " (PMC, per search; divided by the launches of the phase)"
        except Exception:  # noqa: BLE001 -- no profile: traffic stays null
            pass
        tot_ms = det["scan_dense"][0] + det["scan_thr"][0]
        tot_b = det["min_bytes_dense"] + det["min_bytes_thr"]
        roof["achieved"] = tot_b / 1e9 / (tot_ms / 1e3) if tot_ms > 0 else None
        roof["frac"] = roof["achieved"] / 8000.0 if roof["achieved"] else None
        yield {"config": c, "data": kind + "-like synthetic", "nb": nb, "d": d, "nlist": nlist, "k": k, "nprobe": nprobe,
               "metric": "IP" if metric == 0 else "L2", "batch": nq, "value": nq / dt, "unit": "queries/s", "ms_per_batch": dt * 1e3, "qps": nq / dt,
               "recall_at_k": float(recall),
               "dtype": {0: "f32", 1: "f32", 2: "u8"}[arith], "threshold_rounds_through_the_fp32_filter": int(filt),
               "coarse_rankings_from_matrix_core_distances": int(h.last_coarse_pick()),
               "scan_ms": tm["scan_ms"], "select_ms": tm["select_ms"], "coarse_ms": tm["coarse_ms"], "phases": phases, "roofline": roof,
               "scan_algorithmic_GBps": alg / 1e6 / max(tm["scan_ms"], 1e-9), "tile_slot_efficiency": tm["slot_efficiency"],
               "in_flight": flight,
               "cpu_oracle_qps": cpu, "cpu_threads": cores, "gpu_equals_cpu_on_sample": same, "reference": ref,
               "gpu_equals_reference": (ref or {}).get("gpu_equals_reference")}
    h.close()
    del lists


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="1,3,5")
    ap.add_argument("--sample", type=int, default=64)
    ap.add_argument("--ref-sample", type=int, default=2000, help="queries the compiled reference runs (oracle/_ref/ref_harness fixedbench)")
    ap.add_argument("--nprobes", default="", help="comma list: only these nprobe values")
    ap.add_argument("--in-flight", type=int, default=3, help="searches at a time in the in_flight block (1: skip it)")
    args = ap.parse_args()
    import torch
    from auncel_amd import capi
    dev = torch.device("cuda", 0)
    for c in args.cfg.split(","):
        nprobes = tuple(int(v) for v in args.nprobes.split(",")) if args.nprobes else None
        for line in run_config(torch, capi, dev, c, nprobes, args.sample, args.ref_sample, in_flight=args.in_flight):
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
