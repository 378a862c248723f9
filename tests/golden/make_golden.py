#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the compiled reference (this container only).

    make -C oracle ref            # builds oracle/_ref/ref_harness from /root/reference/Auncel
    python tests/golden/make_golden.py [case ...]

For each case in cases.py the inputs are written to a scratch tbundle, the reference harness
(oracle/ref_harness.cpp linked with the reference objects) is run on it, and every tensor it
returns is stored, together with the SHA-256 of the inputs, in <case>.npz.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases  # noqa: E402
from oracle import tbundle  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def main():
    names = sys.argv[1:] or list(cases.CASES)
    for name in names:
        case = cases.CASES[name]()
        kind = case.pop("kind")
        big = case.pop("big", 0)
        with tempfile.TemporaryDirectory() as tmp:
            fin, fout = os.path.join(tmp, "in.tb"), os.path.join(tmp, "out.tb")
            tbundle.save(fin, case)
            env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="4")
            subprocess.run([HARNESS, kind, fin, fout], check=True, cwd=tmp, env=env,
                           stdout=subprocess.DEVNULL)
            out = tbundle.load(fout)
        if big:
            out = cases.trim_big(out, case)
        out["input_sha"] = np.array(cases.input_sha(case))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        sz = os.path.getsize(os.path.join(HERE, name + ".npz"))
        print(f"{name}: {len(out)} tensors, {sz/1e6:.2f} MB")


if __name__ == "__main__":
    main()
