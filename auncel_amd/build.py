"""Builds libauncel_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libauncel_amd.so")
SOURCES = ["ivf_kernels.hip", "ivf_engine.hip"]
# -ffp-contract=off: products and sums are rounded separately, as in the reference's SSE build
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "auncel_amd.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
