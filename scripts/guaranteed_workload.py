#!/usr/bin/env python3
"""The bench pipeline on a data setting where Auncel's own acceptance check holds: every query's recall@k >= 1 - error bound
("Error bound is guaranteed", Auncel/eval/bound.cpp:404-414; its run.sh: `./bound sift10M 5000 5000 10 0.1 6`, i.e. 9 of every
query's 10).  bench.py's headline data (blobs of sigma 38) is too hard for that at any grid point up to multipler 24 -- the minimum
recall over its queries stays at 0.1-0.8 -- so the headline satisfies the metric's mean-recall reading only; here the same
10M x 128 / IVF4096 index shape is built over tighter blobs (sigma 20), the grid is walked until the MINIMUM recall holds on the
validation fifth AND on every query the timed steps searched, and the q/s at that point is reported.
    python scripts/guaranteed_workload.py [--sigma 30] [--steps 24]          (one JSON line; bench.py calls run() for its line)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def parity_of(kept, h, cen, traces, xs, ts, ses, K, topk, bound, mult, sm, nlist, d, capi, log):
    """every timed step's (D, I, my_nprobe) against the compiled reference (oracle/_ref/ref_harness; the pinned CPU restatement where
    the harness is absent) on the same lists / centroids / traces: test infrastructure, after the clock"""
    from oracle import pyoracle, refbench
    codes, ids, off = [], [], np.zeros(nlist + 1, dtype=np.uintp)
    for l in range(nlist):
        c, i_ = h.get_list(l)
        codes.append(c)
        ids.append(i_)
        off[l + 1] = off[l] + len(i_)
    codes, ids = np.concatenate(codes), np.concatenate(ids)
    S = len(xs)
    if refbench.available():
        ro = refbench.run(cen, off, codes, ids, traces, xs, ts, K, topk, bound, mult, sm, single_thread_queries=1, threads=bench.host_cores())
        rD, rI, rnp, against = ro["D"], ro["I"], ro["my_nprobe"].astype(np.uint64), "compiled reference (oracle/_ref/ref_harness)"
    else:
        lists = pyoracle.Lists.__new__(pyoracle.Lists)
        lists.metric, lists.centroids, lists.nlist, lists.d = pyoracle.METRIC_L2, cen, nlist, d
        lists.off, lists.codes, lists.ids = off, codes, ids
        lists.struct = pyoracle.OrcIndex(lists.metric, d, nlist, pyoracle._s(lists.off), pyoracle._f(lists.codes), pyoracle._i(lists.ids))
        nall = ts + S
        tun = pyoracle.Tuner(h.get_interdis(), traces, K, nall, arcos=capi.arcos_table())
        stt = tun.struct(topk, np.full(nall, bound, dtype=np.float32), mult, sm)
        cd, ck = pyoracle.knn(pyoracle.METRIC_L2, xs, cen, nlist, nthreads=bench.host_cores())
        rD, rI, _ = pyoracle.search_preassigned(lists, xs, K, ck, cd, tuner=stt, offset=ts, nthreads=bench.host_cores())
        rnp, against = tun.my_nprobe[ts:ts + S].astype(np.uint64), "CPU restatement (pinned)"
    res = kept.check(lambda start: (rD[start - ts:start - ts + ses], rI[start - ts:start - ts + ses], rnp[start - ts:start - ts + ses]))
    res["against"] = against
    log("    guaranteed workload, timed steps against the reference:", json.dumps(res))
    return res


def run(torch, capi, dev, log, sigma=30.0, nb=10_000_000, d=128, nlist=4096, blobs=20000, K=100, topk=10, bound=0.9, ts=5000, ses=5000,
        steps=24, in_flight=6, nsl=2, check_parity=True):
    t0 = time.time()
    xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, blobs, sigma, 4321)
    gq = torch.Generator(device=dev)
    gq.manual_seed(99)
    xq_t = draw(ts + nsl * ses, gq)
    gtD, _ = bench.ground_truth(torch, xb_t, xq_t, K)
    xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
    del xb_t, xq_t
    torch.cuda.empty_cache()
    cen, _ = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=25, coarse_mode=0, device=dev.index or 0)
    h = capi.Handle(d, nlist, capi.METRIC_L2, dev.index or 0)
    h.set_centroids(cen)
    h.add(xb)
    del xb
    h.set_interdis(None)
    h.set_queries(xq)
    h.set_option("coarse_ties", 2)
    ntr = 0
    while (1 << ntr) <= nlist // 8:
        ntr += 1
    tfit = (ts * 4 // 5) // 10 * 10
    raw = [np.full((tfit * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
    h.train_samples(0, tfit, K, gtD, tfit, raw)
    traces = [capi.trace_sb(r) for r in raw]
    h.set_tuner(K, traces, capi.arcos_table())
    nall = ts + nsl * ses
    req = np.full(nall, bound, dtype=np.float32)
    nval = ts - tfit
    grid = [(m, 1.0) for m in (1.0, 1.5, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0, 16.0, 24.0)]
    out = {"what": "the same pipeline on data where the reference's acceptance check (every query's recall@k >= 1 - error bound, eval/bound.cpp:404-414; "
                   "the setting of its run.sh line `./bound sift10M 5000 5000 10 0.1 6`) holds",
           "data": f"{nb // 1000000}M x {d} uint8-valued blobs ({blobs} centres, sigma {sigma}), IVF{nlist}", "bound": bound, "topk": topk, "grid": [],
           "setup_seconds": time.time() - t0, "bound_guaranteed": False}
    h.set_async_depth(in_flight)
    outs = [(np.empty((ses, K), np.float32), np.empty((ses, K), np.int64)) for _ in range(2 * in_flight)]

    step_outs = [(np.empty((ses, K), np.float32), np.empty((ses, K), np.int64)) for _ in range(max(steps, 2 * in_flight))]

    def run_steps(n, mult, sm, keep=None):
        """n steps over the timed slices, in_flight at a time; returns the per-query recall of EVERY timed slice's last search.
        keep (bench.StepResults): every step returns into a buffer of its own and is noted for the parity check"""
        pend, recs, nps = [], {}, {}

        def finish():
            tk, npq, st0, sn, buf = pend.pop(0)
            D, _, _, _ = h.wait(tk)
            if keep is not None:
                keep.note(sn, st0, buf, npq)
            else:
                recs[st0] = bench.recall_dist(D, gtD[st0:st0 + ses], topk)
                nps[st0] = npq[st0:st0 + ses].copy()

        for sn in range(n):
            if len(pend) == 2 * in_flight:
                finish()
            st0 = ts + (sn % nsl) * ses
            np_ = np.zeros(nall, dtype=np.uint64)
            tr_ = np.zeros(nall, dtype=np.float32)
            buf = step_outs[sn % len(step_outs)] if keep is not None else outs[sn % len(outs)]
            pend.append((h.submit_adaptive(st0, ses, topk, mult, sm, req, np_, tr_, out=buf), np_, st0, sn, buf))
        while pend:
            finish()
        if keep is not None:  # (after the clock: recall of every slice's last search from the kept results)
            for sn, st0, (D, _), npq in keep.pending:
                recs[st0] = bench.recall_dist(D, gtD[st0:st0 + ses], topk)
                nps[st0] = npq[st0:st0 + ses].copy()
            keep.collect(len(step_outs), ses)
        return np.concatenate([recs[k] for k in sorted(recs)]), np.concatenate([nps[k] for k in sorted(nps)])

    for mult, sm in grid:
        np_ = np.zeros(nall, dtype=np.uint64)
        tr_ = np.zeros(nall, dtype=np.float32)
        D, _ = h.search_adaptive(tfit, nval, topk, mult, sm, req, np_, tr_)
        rec = bench.recall_dist(D, gtD[tfit:ts], topk)
        row = {"multipler": mult, "std_m": sm, "validation_recall_min": float(rec.min()), "validation_recall_mean": float(rec.mean()),
               "validation_nprobe_mean": float(np_[tfit:ts].mean())}
        out["grid"].append(row)
        log(f"  guaranteed workload sigma {sigma}: multipler {mult}: validation recall@{topk} min {rec.min():.2f} mean {rec.mean():.4f} nprobe mean {np_[tfit:ts].mean():.1f}")
        if rec.min() < bound:
            continue
        # holds on the validation fifth: the timed steps at this point, and the reference's check on the queries they searched (it is
        # made on the searched queries, with hand-tuned hyper-parameters -- hyperparameter.txt; a point that fails it is followed by the next)
        run_steps(2 * in_flight, mult, sm)
        torch.cuda.synchronize()
        kept = bench.StepResults()
        t1 = time.perf_counter()
        rec_t, np_t = run_steps(steps, mult, sm, kept)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        row.update({"value": ses * steps / el, "timed_recall_min": float(rec_t.min()), "timed_recall_mean": float(rec_t.mean()), "timed_nprobe_mean": float(np_t.mean())})
        log(f"    timed: {ses * steps / el / 1e6:.3f} M q/s, recall min {rec_t.min():.2f} mean {rec_t.mean():.5f} over {len(rec_t)} queries")
        if rec_t.min() >= bound:
            out.update({"multipler": mult, "std_m": sm, "value": ses * steps / el, "unit": "queries/s", "ms_per_step": 1e3 * el / steps, "in_flight": in_flight,
                        "recall_min": float(rec_t.min()), "recall_mean": float(rec_t.mean()), "queries_checked": int(len(rec_t)), "bound_guaranteed": True,
                        "nprobe_mean": float(np_t.mean()), "coarse_ties": "redo (the exact regime, as the headline)"})
            if check_parity:
                out["parity"] = parity_of(kept, h, cen, traces, xq[ts:ts + nsl * ses], ts, ses, K, topk, bound, mult, sm, nlist, d, capi, log)
            break
    h.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sigma", default="20")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--nb", type=int, default=10_000_000)
    a = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    from auncel_amd import capi
    for s in a.sigma.split(","):
        print(json.dumps(run(torch, capi, torch.device("cuda", 0), bench.log, sigma=float(s), nb=a.nb, steps=a.steps)), flush=True)
