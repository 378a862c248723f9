#!/usr/bin/env python3
"""Goldens for the reference's eval/ harnesses: the four programs built FROM THE REFERENCE (oracle/Makefile refevalbin:
oracle/_ref/refeval_*, real headers, libfaiss_ref.a, MKL) run on the CPU on the synthetic world of tests/eval_world.py; what they
print and write that does not depend on the clock goes to tests/golden/harness_sift_d32.json.  tests/test_gpu_eval_harness.py
compares the same programs built against the mirror and run on the GPU with it.
    make -C oracle refevalbin && python tests/golden/make_harness_golden.py"""
import hashlib
import json
import os
import pathlib
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import eval_world  # noqa: E402

REFDIR = os.path.join(eval_world.ROOT, "oracle", "_ref")


def main():
    out = {}
    with tempfile.TemporaryDirectory() as t:
        w = eval_world.build(pathlib.Path(t))
        w["env"]["MKL_THREADING_LAYER"] = "SEQUENTIAL"  # (as oracle/refbench.py runs the reference: libgomp and MKL's OpenMP layer do not mix)
        for name, args in (("bound", ["sift10M", w["ts"], w["ses"], 10, 0.1, 9]), ("effect_error", ["sift10M", 100, w["ts"], w["ses"]])):
            r = subprocess.run([os.path.join(REFDIR, "refeval_" + name)] + [str(a) for a in args], cwd=w["run"], env=w["env"], capture_output=True,
                               text=True, timeout=1800)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            out[name] = eval_world.outputs(w, name, r.stdout)
            print(name, "done:", {k: (v if not isinstance(v, list) else f"{len(v)} rows") for k, v in out[name].items()})
        idx = (w["run"] / "trained_index" / "sift10M_IVF1024,Flat_trained.index").read_bytes()
        out["trained_index_sha256"] = hashlib.sha256(idx).hexdigest()
        out["trained_index_bytes"] = len(idx)
    with open(os.path.join(HERE, "harness_sift_d32.json"), "w") as f:
        import re
        f.write(re.sub(r'\[\s+"([^"]+)",\s+"([^"]+)"\s+\]', r'["\1", "\2"]', json.dumps(out, indent=1)) + "\n")


if __name__ == "__main__":
    main()
