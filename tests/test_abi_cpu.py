"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
fails loudly without a GPU, and its host-side shard merge matches the reference's merge_tables."""
import os
import re

import numpy as np
import pytest

from util import FIXED, load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from auncel_amd import build, capi
    build.build()
    return capi


def test_exports_match_header(capi):
    hdr = open(os.path.join(ROOT, "include", "auncel_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(amd_ivf_[a-z_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = capi.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == declared


def test_option_table_matches_header():
    """every key amd_ivf_set_option accepts (the engine's OPT_TABLE) is in the header's table, and nothing else is"""
    hdr = open(os.path.join(ROOT, "include", "auncel_amd.h")).read()
    block = hdr[hdr.index(" *   key "):hdr.index("amd_ivf_set_option(h, key, NAN)")]
    documented = set(re.findall(r'"([a-z_]+)"', block))
    eng = open(os.path.join(ROOT, "auncel_amd", "csrc", "ivf_engine.hip")).read()
    table = eng[eng.index("const OptSpec OPT_TABLE[N_OPT] = {"):]
    table = table[:table.index("};")]
    accepted = set(re.findall(r'\{"([a-z_]+)", "AUNCEL_AMD_', table))
    assert accepted and accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))


def test_no_cpu_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.EngineError) as e:
        capi.Handle(16, 4)
    assert e.value.code == -4


@pytest.mark.parametrize("name", [n for n in FIXED if n not in ("fixed_gist_l2_d960", "fixed_odd_d30")])
def test_merge_tables_host(capi, oracle, name):
    case, gold = load_case(name)
    nshard, a = case["nshard"], gold["assign"]
    for k in case["ks"]:
        allD, allI = [], []
        for s in range(nshard):
            sub = oracle.Lists(case["metric"], case["centroids"], case["xb"], np.where(a % nshard == s, a, -1))
            D, I, _ = oracle.search_preassigned(sub, case["xq"], int(k), gold["coarse_keys_sse"], gold["coarse_dis_sse"])
            allD.append(D)
            allI.append(I)
        D, I = capi.merge_tables(case["metric"], np.stack(allD), np.stack(allI))
        assert np.array_equal(I, gold[f"I_shards_k{k}"])
        assert np.array_equal(D.view(np.uint32), gold[f"D_shards_k{k}"].view(np.uint32))


@pytest.mark.parametrize("name", ["auncel_sift_d32", "auncel_gauss_d64"])
def test_trace_sb_host(capi, name):
    """Trace::SB bucketing (host side of the product) against the reference's traces"""
    case, gold = load_case(name)
    i = 0
    while f"raw_trace{i}" in gold:
        x, y, s = capi.trace_sb(gold[f"raw_trace{i}"])
        assert np.array_equal(x.view(np.uint32), gold[f"sb_trace{i}"][:, 0].view(np.uint32))
        assert np.array_equal(y.view(np.uint32), gold[f"sb_trace{i}"][:, 1].view(np.uint32))
        assert np.array_equal(s.view(np.uint32), gold[f"sb_stds{i}"].view(np.uint32))
        i += 1
    assert i == 8
    # numpy arccos vs the reference's LUT (std::acos on float)
    assert np.array_equal(capi.arcos_table().view(np.uint32), gold["arcos_list"].view(np.uint32))


def test_loading_the_library_leaves_the_environment_alone():
    """the HIP runtime's GPU_MAX_HW_QUEUES is the process's business: neither the package nor the library's static constructors set it"""
    import subprocess
    import sys
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import auncel_amd; from auncel_amd import capi; capi.lib(); "
            "import ctypes; libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "print('ENV', os.environ.get('GPU_MAX_HW_QUEUES'), libc.getenv(b'GPU_MAX_HW_QUEUES'))")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ENV None None" in r.stdout, r.stdout
