"""The small synthetic world the reference's eval/ harnesses are run on -- by tests/test_gpu_eval_harness.py (the binaries built
against the mirror, on the GPU) and by tests/golden/make_harness_golden.py (the binaries built from the reference itself, on the
CPU, whose outputs are the goldens): files in the layouts and at the compiled-in paths of Auncel/eval/*.cpp for "sift10M"."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TS, SES = 200, 100
HYPER = ["9.3 1.0", "6.9 1.0", "2.7 12.0", "11.0 8.0", "6.7 1.0", "7.9 6.0", "10.2 6.0", "26.5 1.0", "10.0 0.2", "4.2 1.0", "4.5 1.0", "15.0 1.0"]


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


def build(tmp):
    """tmp: a pathlib directory.  Returns {run, env, ts, ses}: the working directory of the harnesses and their environment."""
    shim = str(tmp / "path_remap.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", shim, os.path.join(ROOT, "tests", "cpp", "path_remap.c"), "-ldl"], check=True)
    # a small SIFT-like world under <tmp>/data/sift/sift10M/ (the paths eval/*.cpp compile in for "sift10M")
    rs = np.random.RandomState(3)
    d, nb, ts, ses = 32, 60000, TS, SES
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
    (tmp / "w" / "hyperparameter.txt").write_text("\n".join(HYPER) + "\n")
    env = dict(os.environ, LD_PRELOAD=shim, AUNCEL_DATA_ROOT=str(tmp / "data"), OMP_NUM_THREADS="4")
    return {"run": run, "env": env, "ts": ts, "ses": ses}


def outputs(world, name, stdout):
    """the parts of a harness run that do not depend on the clock"""
    run = world["run"]
    if name == "bound":
        return {"error_bound": stdout.split("Error Bound :")[1].split()[0], "guaranteed": "Error bound is guaranteed" in stdout}
    if name == "effect_error":
        return {"rows": [l.split() for l in (run / "Effective_error_sift10M.log").read_text().splitlines() if l.strip()]}
    return {}
