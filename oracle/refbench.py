"""TEST INFRASTRUCTURE ONLY (oracle/): runs the compiled reference (oracle/_ref/ref_harness, built by oracle/Makefile
from /root/reference in the build container; the binary travels to the GPU box, the sources do not) in its `bench` mode:
the reference's own Error_sys::search, one query per call as eval/bound.cpp issues them, timed on the host.  Used by
bench.py's cpu_baseline leg (kind "reference") and by tests/test_oracle_golden.py."""
import os
import subprocess
import tempfile

import numpy as np

from . import tbundle

HERE = os.path.dirname(os.path.abspath(__file__))
HARNESS = os.path.join(HERE, "_ref", "ref_harness")


def available():
    return os.path.exists(HARNESS) and os.access(HARNESS, os.X_OK)


def run(centroids, list_off, codes, ids, traces, xq, id0, max_topk, topk, require_acc, multipler, std_m,
        single_thread_queries=64, threads=None, timeout=900, tmpdir=None):
    """traces: [(x, y, std)] as amd_ivf_set_tuner takes them; xq: the queries id0 .. id0 + len(xq).
    Returns dict(D, I, my_nprobe, seconds_one_thread, queries_one_thread, seconds_all_threads, threads)."""
    t = {"d": int(centroids.shape[1]), "nlist": int(centroids.shape[0]), "max_topk": int(max_topk), "topk": int(topk),
         "id0": int(id0), "single_thread_queries": int(single_thread_queries),
         "centroids": np.ascontiguousarray(centroids, dtype=np.float32),
         "list_off": np.ascontiguousarray(list_off).astype(np.int64), "codes": codes, "ids": np.ascontiguousarray(ids, dtype=np.int64),
         "xq": np.ascontiguousarray(xq, dtype=np.float32), "require_acc": np.array([require_acc], dtype=np.float32),
         "multipler": float(multipler), "std_m": float(std_m)}
    for i, (x, y, s) in enumerate(traces):
        t[f"trace_x{i}"] = np.ascontiguousarray(x, dtype=np.float32)
        t[f"trace_y{i}"] = np.ascontiguousarray(y, dtype=np.float32)
        t[f"trace_std{i}"] = np.ascontiguousarray(s, dtype=np.float32)
    with tempfile.TemporaryDirectory(dir=tmpdir) as tmp:
        fin, fout = os.path.join(tmp, "in.tb"), os.path.join(tmp, "out.tb")
        tbundle.save(fin, t)
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS=str(threads or os.cpu_count() or 1))
        subprocess.run([HARNESS, "bench", fin, fout], check=True, cwd=tmp, env=env, timeout=timeout, stdout=subprocess.DEVNULL)
        os.remove(fin)
        out = tbundle.load(fout)
    return {"D": out["D"], "I": out["I"], "my_nprobe": out["my_nprobe"], "seconds_one_thread": float(out["seconds_one_thread"][0]),
            "queries_one_thread": int(out["queries_one_thread"][0]), "seconds_all_threads": float(out["seconds_all_threads"][0]),
            "threads": int(out["threads"][0])}


def run_fixed(metric, centroids, list_off, codes, ids, xq, k, nprobe, threads=None, timeout=900, tmpdir=None):
    """IndexIVF::search(1, x, k, ...) of the compiled reference for every row of xq (nprobe fixed), OpenMP over queries"""
    with tempfile.TemporaryDirectory(dir=tmpdir) as tmp:
        fin, fout = os.path.join(tmp, "in.tb"), os.path.join(tmp, "out.tb")
        tbundle.save(fin, {"d": int(centroids.shape[1]), "nlist": int(centroids.shape[0]), "k": int(k), "nprobe": int(nprobe),
                           "metric": int(metric), "centroids": np.ascontiguousarray(centroids, dtype=np.float32),
                           "list_off": np.ascontiguousarray(list_off).astype(np.int64), "codes": codes,
                           "ids": np.ascontiguousarray(ids, dtype=np.int64), "xq": np.ascontiguousarray(xq, dtype=np.float32)})
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS=str(threads or os.cpu_count() or 1))
        subprocess.run([HARNESS, "fixedbench", fin, fout], check=True, cwd=tmp, env=env, timeout=timeout, stdout=subprocess.DEVNULL)
        os.remove(fin)
        out = tbundle.load(fout)
    return {"D": out["D"], "I": out["I"], "seconds_all_threads": float(out["seconds_all_threads"][0]), "threads": int(out["threads"][0])}
