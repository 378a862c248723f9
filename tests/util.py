"""shared helpers for the tests: golden loading, list building"""
import functools
import os

import numpy as np

import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def load_case(name):
    """returns (inputs, golden) after checking that the regenerated inputs are the bytes
    the golden outputs were computed from"""
    case = cases.CASES[name]()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    gold = {k: g[k] for k in g.files}
    assert str(gold["input_sha"]) == cases.input_sha(case), f"{name}: regenerated inputs differ from golden inputs"
    return case, gold


FIXED = [n for n in cases.CASES if n.startswith("fixed_")]
AUNCEL_BIG = ["auncel_sift_nl4096"]  # nlist = 4096 (BASELINE config 2's shape); golden trimmed by cases.trim_big
AUNCEL = [n for n in cases.CASES if n.startswith("auncel_") and n not in AUNCEL_BIG]
KMEANS = [n for n in cases.CASES if n.startswith("kmeans_")]


def traces_from_gold(gold, prefix="sb_"):
    out = []
    i = 0
    while f"{prefix}trace{i}" in gold:
        t = gold[f"{prefix}trace{i}"]
        out.append((t[:, 0].copy(), t[:, 1].copy(), gold[f"{prefix}stds{i}"].copy()))
        i += 1
    return out
