"""The host-side C++ mirror of the reference's classes (auncel_amd/csrc/host, namespace faiss) run
through the flows of the reference's own callers by tests/cpp/host_mirror_driver.cpp."""
import os
import subprocess

import numpy as np
import pytest

from util import load_case, traces_from_gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER_SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_driver.cpp")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    from auncel_amd import build
    build.build_host()
    exe = str(tmp_path_factory.mktemp("drv") / "host_mirror_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", DRIVER_SRC, "-o", exe, "-L" + build.LIBDIR, "-lfaiss_amd", "-launcel_amd",
                    "-Wl,-rpath," + build.LIBDIR, "-pthread"], check=True)
    return exe


def test_driver_builds_and_links(driver):
    """CPU-side: the mirror and a caller written against the reference's class names compile and link"""
    assert os.path.exists(driver)


def _run(driver, kind, tensors, tmp_path):
    from oracle import tbundle
    f = str(tmp_path / "in.tb")
    tbundle.save(f, tensors)
    r = subprocess.run([driver, kind, f], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fixed_sift_l2", "fixed_gauss_l2_d96", "fixed_deep_ip_d96", "fixed_ragged", "fixed_dups"])
def test_fixed_flows(driver, tmp_path, name):
    case, gold = load_case(name)
    t = {k: v for k, v in case.items() if k != "kind"}
    t.update({k: v for k, v in gold.items() if k != "input_sha"})
    _run(driver, "fixed", t, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kmeans_toy", "kmeans_void", "kmeans_toy_ip", "kmeans_sub_int"])
def test_clustering(driver, tmp_path, name):
    """faiss::Clustering / IndexIVF::train of the mirror: assignment on the GPU, the reference's update procedure; centroids
    bit for bit against the compiled reference (goldens)"""
    case, gold = load_case(name)
    t = {k: v for k, v in case.items() if k != "kind"}
    t.update({k: v for k, v in gold.items() if k != "input_sha"})
    _run(driver, "kmeans", t, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["auncel_sift_d32", "auncel_gauss_d64"])
def test_error_sys_flow(driver, oracle, tmp_path, name):
    case, gold = load_case(name)
    K, ts = case["max_topk"], case["train_num"]
    # expectation for sys_train: pinned oracle fed with the exact coarse ranking (see test_gpu_parity)
    lists = oracle.Lists(case["metric"], gold["centroids"], case["xb"], gold["assign"])
    ntr = len(traces_from_gold(gold))
    raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
    oracle.train_samples(lists, case["xq"][:ts], K, gold["coarse_keys_sse"][:ts], gold["coarse_dis_sse"][:ts],
                         gold["interdis_cem"], gold["arcos_list"], gold["gtD"], 0, ts, raw)
    t = {k: v for k, v in case.items() if k != "kind"}
    for k in ("centroids", "interdis_cem", "gtD", "gtI"):
        t[k] = gold[k]
    for i in range(ntr):
        x, y, s = oracle.trace_sb(raw[i])
        t[f"exp_sb_trace{i}"] = np.stack([x, y], 1)
        t[f"exp_sb_stds{i}"] = s
        t[f"sb_trace{i}"] = gold[f"sb_trace{i}"]
        t[f"sb_stds{i}"] = gold[f"sb_stds{i}"]
    for r in range(len(case["topks"])):
        for k in ("I", "D", "my_nprobe"):
            t[f"{k}_r{r}"] = gold[f"{k}_r{r}"]
    _run(driver, "auncel", t, tmp_path)


@pytest.mark.parametrize("name", ["io_ragged", "io_sift"])
def test_index_io_bytes(tmp_path, name):
    """write_index / read_index of the mirror against index files written by the reference (CPU only)"""
    from auncel_amd import build
    from oracle import tbundle
    build.build_host()
    exe = str(tmp_path / "index_io_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "index_io_driver.cpp"), "-o", exe, "-L" + build.LIBDIR,
                    "-lfaiss_amd", "-launcel_amd", "-Wl,-rpath," + build.LIBDIR, "-pthread"], check=True)
    case, gold = load_case(name)
    t = {k: v for k, v in case.items() if k != "kind"}
    t.update({k: v for k, v in gold.items() if k != "input_sha"})
    f = str(tmp_path / "in.tb")
    tbundle.save(f, t)
    r = subprocess.run([exe, f, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists("/root/reference/Auncel/eval/bound.cpp"), reason="reference tree not present")
def test_reference_harness_builds_against_mirror(tmp_path):
    """drop-in check: the reference's own eval/bound.cpp (the north-star caller) compiles unmodified against the
    mirror headers and links with libfaiss_amd + libauncel_amd, and so does effect_time.cpp; effect_error / overhead compile as well"""
    from auncel_amd import build
    build.build_host()
    root = tmp_path / "Auncel"
    root.mkdir()
    # the sources stay where they are; the directory is a real one (through a symlinked directory "../IndexIVF.h" would resolve
    # next to the link's target: to the reference's own headers, not to the mirror's)
    (root / "eval").mkdir()
    for src in ("bound", "effect_error", "overhead", "effect_time"):
        os.symlink(f"/root/reference/Auncel/eval/{src}.cpp", root / "eval" / f"{src}.cpp")
    for f in os.listdir(build.HOST_DIR):
        if f.endswith(".h"):
            os.symlink(os.path.join(build.HOST_DIR, f), root / f)
    for src in ("bound", "effect_error", "overhead", "effect_time"):
        subprocess.run(["g++", "-std=c++17", "-O0", "-fopenmp", "-w", "-c", str(root / "eval" / f"{src}.cpp"), "-o", str(tmp_path / f"{src}.o")],
                       check=True)
    subprocess.run(["g++", "-fopenmp", str(tmp_path / "bound.o"), "-L" + build.LIBDIR, "-lfaiss_amd", "-launcel_amd",
                    "-Wl,-rpath," + build.LIBDIR, "-o", str(tmp_path / "bound")], check=True)
    r = subprocess.run([str(tmp_path / "bound")], capture_output=True, text=True)  # no argv: usage path, no GPU touched
    assert r.returncode != 127
    # effect_time.cpp is the caller of Error_sys::time_search
    subprocess.run(["g++", "-fopenmp", str(tmp_path / "effect_time.o"), "-L" + build.LIBDIR, "-lfaiss_amd", "-launcel_amd",
                    "-Wl,-rpath," + build.LIBDIR, "-o", str(tmp_path / "effect_time")], check=True)
