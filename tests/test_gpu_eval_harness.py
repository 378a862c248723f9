"""The reference's own harnesses -- Auncel/eval/bound.cpp (the north-star caller), effect_error.cpp, overhead.cpp,
effect_time.cpp -- compiled UNMODIFIED against the mirror headers and linked with libfaiss_amd + libauncel_amd
(oracle/Makefile `evalbin`, built where /root/reference exists; the binaries travel like ref_harness), RUN on the GPU on small
synthetic files: index_factory("IVF1024,Flat"), train, add, Error_sys::sys_train / set_queries / search / time_search with
their hard-coded dataset paths redirected by an LD_PRELOAD shim (tests/cpp/path_remap.c)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
BINS = {n: os.path.join(REFDIR, "eval_" + n) for n in ("bound", "effect_error", "overhead", "effect_time")}


def write_fvecs(path, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows = np.empty((x.shape[0], x.shape[1] + 1), dtype=np.float32)
    rows[:, 0] = np.array([x.shape[1]], dtype=np.int32).view(np.float32)[0]
    rows[:, 1:] = x
    rows.tofile(path)


def write_ivecs(path, x):
    x = np.ascontiguousarray(x, dtype=np.int32)
    rows = np.empty((x.shape[0], x.shape[1] + 1), dtype=np.int32)
    rows[:, 0] = x.shape[1]
    rows[:, 1:] = x
    rows.tofile(path)


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    if not all(os.path.exists(b) for b in BINS.values()):
        pytest.skip("oracle/_ref/eval_* not built (they are built where the reference sources are: make -C oracle evalbin)")
    tmp = tmp_path_factory.mktemp("evalrun")
    shim = str(tmp / "path_remap.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", shim, os.path.join(ROOT, "tests", "cpp", "path_remap.c"), "-ldl"], check=True)
    # a small SIFT-like world under <tmp>/data/sift/sift10M/ (the paths eval/*.cpp compile in for "sift10M")
    rs = np.random.RandomState(3)
    d, nb, ts, ses = 32, 60000, 200, 100
    cen = rs.randint(0, 160, size=(600, d))
    xb = np.clip(cen[rs.randint(0, 600, nb)] + rs.randn(nb, d) * 18, 0, 255).astype(np.uint8).astype(np.float32)
    xq = np.clip(cen[rs.randint(0, 600, ts + ses)] + rs.randn(ts + ses, d) * 18, 0, 255).astype(np.uint8).astype(np.float32)
    dist = (xq ** 2).sum(1)[:, None] + (xb ** 2).sum(1)[None, :] - 2.0 * (xq @ xb.T)
    gi = np.argsort(dist, axis=1, kind="stable")[:, :100]
    gd = np.take_along_axis(dist, gi, 1).astype(np.float32)
    dd = tmp / "data" / "sift" / "sift10M"
    dd.mkdir(parents=True)
    write_fvecs(dd / "sift10M.fvecs", xb)
    write_fvecs(dd / "query.fvecs", xq)
    write_ivecs(dd / "idx.ivecs", gi)
    write_fvecs(dd / "dis.fvecs", gd)
    run = tmp / "w" / "run"  # (the harnesses read ../hyperparameter.txt and write ./trained_index/)
    (run / "trained_index").mkdir(parents=True)
    (tmp / "w" / "hyperparameter.txt").write_text("\n".join(["9.3 1.0", "6.9 1.0", "2.7 12.0", "11.0 8.0", "6.7 1.0", "7.9 6.0", "10.2 6.0", "26.5 1.0",
                                                             "10.0 0.2", "4.2 1.0", "4.5 1.0", "15.0 1.0"]) + "\n")
    env = dict(os.environ, LD_PRELOAD=shim, AUNCEL_DATA_ROOT=str(tmp / "data"), OMP_NUM_THREADS="4")
    return {"run": run, "env": env, "ts": ts, "ses": ses}


def _go(world, name, args):
    r = subprocess.run(["stdbuf", "-o0", BINS[name]] + [str(a) for a in args], cwd=world["run"], env=world["env"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_bound_runs_on_the_gpu(world):
    """./bound sift10M <train> <test> 10 0.1 9: one Error_sys::search(D, I, i, 1) per query (eval/bound.cpp:391-396)"""
    out = _go(world, "bound", ["sift10M", world["ts"], world["ses"], 10, 0.1, 9])
    assert "Finish error profile system search" in out and "Error Bound :" in out, out[-2000:]
    eb = float(out.split("Error Bound :")[1].split()[0])
    assert 0.0 <= eb <= 1.0
    assert "Error bound is guaranteed" in out, out[-1500:]  # (well separated blobs, multipler 10: every query finds its top 10)
    lat = (world["run"] / "Auncel_Latency_sift10M_10_10.log").read_text().split()
    assert len(lat) == world["ses"] and all(float(v) > 0 for v in lat)
    assert (world["run"] / "trained_index" / "sift10M_IVF1024,Flat_trained.index").stat().st_size > 1024 * 32 * 4


def test_effect_error_runs_on_the_gpu(world):
    out = _go(world, "effect_error", ["sift10M", 100, world["ts"], world["ses"]])
    assert "Finish error profile system search" in out
    rows = [l.split() for l in (world["run"] / "Effective_error_sift10M.log").read_text().splitlines() if l.strip()]
    assert len(rows) == world["ses"]
    acc = np.array([float(r[0]) for r in rows])
    rec = np.array([float(r[1]) for r in rows])
    assert set(np.round(acc, 1)) <= {0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3}
    assert ((rec >= 0) & (rec <= 1)).all() and (rec >= acc - 1e-6).mean() > 0.7  # the true recall at the stop mostly holds the asked one


def test_overhead_runs_on_the_gpu(world):
    out = _go(world, "overhead", ["sift10M", 100, world["ts"], world["ses"]])
    assert "With ELP search Time:" in out


def test_effect_time_runs_on_the_gpu(world):
    out = _go(world, "effect_time", ["sift10M", 100, world["ts"], world["ses"]])
    assert "Finish error profile system search" in out
    rows = [l.split() for l in (world["run"] / "Effective_time_sift10M.log").read_text().splitlines() if l.strip()]
    assert len(rows) == world["ses"] and all(float(r[1]) > 0 for r in rows)
