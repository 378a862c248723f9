"""INTEGRATION.md route B: integration/AmdIndexIVFFlat.h is the subclass a maintainer adds inside the reference tree.  It
is compiled here against the reference's own, unmodified headers (C++11, as the reference builds) and linked into a small
program with the compiled reference and the engine -- the override of IndexIVF::search_preassigned, the trace upload and
the per-thread contexts are real code, not prose.  Needs the reference tree (this container only)."""
import os
import subprocess

import pytest

REF = "/root/reference/Auncel"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

PROGRAM = r"""
#include "Auncel/gpu_amd/AmdIndexIVFFlat.h"
#include <cstdio>
int main() {
    faiss::IndexFlatL2 quantizer(8);
    faiss::AmdIndexIVFFlat index(&quantizer, 8, 4, faiss::METRIC_L2, 0);
    faiss::IndexIVF* base = &index;  // the reference's callers hold IndexIVF*: the override is reached through the vtable
    std::printf("%zu %d\n", base->nlist, (int)index.device_coarse);
    return 0;
}
"""


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "IndexIVFFlat.h")), reason="reference tree not present")
def test_subclass_compiles_against_reference_headers(tmp_path):
    tree = tmp_path / "Auncel"
    (tree / "gpu_amd").mkdir(parents=True)
    for f in os.listdir(REF):
        if f.endswith(".h"):
            os.symlink(os.path.join(REF, f), tree / f)  # the reference's headers stay where they are
    os.symlink(os.path.join(ROOT, "integration", "AmdIndexIVFFlat.h"), tree / "gpu_amd" / "AmdIndexIVFFlat.h")
    src = tmp_path / "main.cpp"
    src.write_text(PROGRAM)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wno-sign-compare", "-Wno-unused-variable", "-Wno-unknown-pragmas", "-Werror=return-type",
                    "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "main.o")], check=True, cwd=tmp_path)
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libfaiss_ref.a")
    from auncel_amd import build
    if os.path.exists(ref_lib) and os.path.exists(build.LIB):
        # link with the compiled reference and the engine: every symbol the subclass uses exists on both sides
        mkl = "/opt/conda/lib/libmkl_rt.so.1"
        cmd = ["g++", "-fopenmp", str(tmp_path / "main.o"), ref_lib, "-L" + build.LIBDIR, "-launcel_amd", "-Wl,-rpath," + build.LIBDIR,
               "-o", str(tmp_path / "main")]
        if os.path.exists(mkl):
            cmd.insert(4, mkl)
        subprocess.run(cmd, check=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["float", "bytes"])
def test_reference_classes_drive_the_subclass(kind):
    """oracle/_ref/subclass_driver (tests/cpp/subclass_driver.cpp, built where the reference is): the compiled reference's
    IndexIVF::search, Error_sys::sys_train and Error_sys::search run once on its own CPU index and once on
    integration/AmdIndexIVFFlat.h, and every output is compared bit for bit"""
    exe = os.path.join(ROOT, "oracle", "_ref", "subclass_driver")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/subclass_driver not built (needs the reference tree at build time)")
    env = dict(os.environ, MKL_THREADING_LAYER="GNU", OMP_NUM_THREADS="8")
    r = subprocess.run([exe, kind], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "SUBCLASS PARITY OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
