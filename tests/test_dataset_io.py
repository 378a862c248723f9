"""The harness's dataset readers (Auncel/eval/bound.cpp:29-113) behind the C ABI (amd_ivf_read_*) and under their own names in
the class mirror (auncel_amd/csrc/host/dataset_io.h): files written here by a few lines of numpy, read back, compared."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from auncel_amd import build, capi
    build.build()
    return capi


def write_vecs(path, x):
    """.fvecs / .ivecs: every row = int32 d + d 4-byte values"""
    n, d = x.shape
    rows = np.empty((n, d + 1), dtype=np.int32)
    rows[:, 0] = d
    rows[:, 1:] = x.view(np.int32)
    rows.tofile(path)


def write_bin(path, x, n_header=None):
    """.fbin / .ibin / .u8bin: int32 n, int32 d, payload"""
    with open(path, "wb") as f:
        np.array([x.shape[0] if n_header is None else n_header, x.shape[1]], dtype=np.int32).tofile(f)
        x.tofile(f)


def test_fvecs_ivecs_round_trip(capi, tmp_path):
    rs = np.random.RandomState(3)
    x = rs.randn(1237, 96).astype(np.float32)
    g = rs.randint(0, 1 << 30, size=(211, 100)).astype(np.int32)
    write_vecs(tmp_path / "base.fvecs", x)
    write_vecs(tmp_path / "gt.ivecs", g)
    rx = capi.read_fvecs(str(tmp_path / "base.fvecs"))
    rg = capi.read_ivecs(str(tmp_path / "gt.ivecs"))
    assert rx.dtype == np.float32 and rx.shape == x.shape and np.array_equal(rx.view(np.uint32), x.view(np.uint32))
    assert rg.dtype == np.int32 and np.array_equal(rg, g)


def test_fbin_widths_and_row_count(capi, tmp_path):
    rs = np.random.RandomState(4)
    x = rs.randn(500, 24).astype(np.float32)
    write_bin(tmp_path / "x.fbin", x)
    r, n = capi.read_fbin(str(tmp_path / "x.fbin"))
    assert n == 500 and np.array_equal(r.view(np.uint32), x.view(np.uint32))
    r, n = capi.read_fbin(str(tmp_path / "x.fbin"), num=123)  # the harness asks for the rows it wants, *n stays the header's
    assert n == 500 and r.shape == (123, 24) and np.array_equal(r, x[:123])
    # one byte per value: read as signed chars and widened (bound.cpp:83-92) -- 200 comes back as -56
    u = rs.randint(0, 256, size=(300, 128)).astype(np.uint8)
    write_bin(tmp_path / "x.u8bin", u)
    r, n = capi.read_fbin(str(tmp_path / "x.u8bin"), nbytes=1)
    assert n == 300 and np.array_equal(r, u.view(np.int8).astype(np.float32))
    ids = rs.randint(0, 1 << 31, size=(50, 10)).astype(np.int32)
    write_bin(tmp_path / "gt.ibin", ids)
    r, n = capi.read_ibin(str(tmp_path / "gt.ibin"))
    assert n == 50 and r.dtype == np.int32 and np.array_equal(r, ids)


def test_malformed_files_are_errors_not_aborts(capi, tmp_path):
    with pytest.raises(capi.EngineError) as e:
        capi.read_fvecs(str(tmp_path / "missing.fvecs"))
    assert e.value.code == -2 and "could not open" in str(e.value)
    x = np.zeros((10, 8), dtype=np.float32)
    write_vecs(tmp_path / "t.fvecs", x)
    with open(tmp_path / "t.fvecs", "ab") as f:
        f.write(b"\0" * 5)
    with pytest.raises(capi.EngineError) as e:
        capi.read_fvecs(str(tmp_path / "t.fvecs"))
    assert "weird file size" in str(e.value)
    write_bin(tmp_path / "short.fbin", x, n_header=20)  # header promises more rows than the file holds
    with pytest.raises(capi.EngineError) as e:
        capi.read_fbin(str(tmp_path / "short.fbin"))
    assert "could not read whole file" in str(e.value)
    with pytest.raises(capi.EngineError):
        capi.read_fbin(str(tmp_path / "short.fbin"), nbytes=2)


def test_mirror_header_keeps_the_harness_names(capi, tmp_path):
    """fvecs_read / ivecs_read / fbin_read / ibin_read with the harness's signatures, from a caller compiled against the mirror"""
    from auncel_amd import build
    rs = np.random.RandomState(5)
    x = rs.randn(64, 12).astype(np.float32)
    u = rs.randint(0, 256, size=(40, 16)).astype(np.uint8)
    write_vecs(tmp_path / "a.fvecs", x)
    write_bin(tmp_path / "b.u8bin", u)
    src = tmp_path / "drv.cpp"
    src.write_text(r'''
#include <cstdio>
#include "dataset_io.h"
int main(int argc, char** argv) {
    size_t d, n;
    float* a = faiss::fvecs_read(argv[1], &d, &n);
    double s = 0;
    for (size_t i = 0; i < d * n; i++) s += a[i];
    printf("%zu %zu %.9g\n", d, n, s);
    delete[] a;
    float* b = faiss::fbin_read(argv[2], &d, &n, 40, 1);
    s = 0;
    for (size_t i = 0; i < d * 40; i++) s += b[i];
    printf("%zu %zu %.9g\n", d, n, s);
    delete[] b;
    try {
        faiss::fvecs_read("/nonexistent.fvecs", &d, &n);
        return 1;
    } catch (const faiss::FaissException& e) {
        printf("threw\n");
    }
    return 0;
}
''')
    exe = str(tmp_path / "drv")
    subprocess.run(["g++", "-std=c++17", "-O1", str(src), "-I" + build.HOST_DIR, "-o", exe, "-L" + build.LIBDIR, "-launcel_amd",
                    "-Wl,-rpath," + build.LIBDIR], check=True)
    out = subprocess.run([exe, str(tmp_path / "a.fvecs"), str(tmp_path / "b.u8bin")], capture_output=True, text=True, check=True).stdout.split("\n")
    assert out[0].split()[:2] == ["12", "64"] and abs(float(out[0].split()[2]) - float(x.astype(np.float64).sum())) < 1e-3
    assert out[1].split()[:2] == ["16", "40"] and float(out[1].split()[2]) == float(u.view(np.int8).astype(np.float64).sum())
    assert out[2] == "threw"
