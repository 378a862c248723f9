"""Builds libauncel_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libauncel_amd.so")
SOURCES = ["ivf_kernels.hip", "ivf_plan.hip", "ivf_kmeans.hip", "ivf_engine.hip"]
# -ffp-contract=off: products and sums are rounded separately, as in the reference's SSE build
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))] + [os.path.join(HERE, "..", "include", "auncel_amd.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False):
    extra = os.environ.get("AUNCEL_AMD_CXXFLAGS", "").split()
    stamp = os.path.join(LIBDIR, "flags.txt")  # a library built with other flags (experiments) is rebuilt
    want = " ".join(FLAGS + extra)
    same_flags = os.path.exists(stamp) and open(stamp).read() == want
    if not force and same_flags and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(want)
    return LIB


HOST_LIB = os.path.join(LIBDIR, "libfaiss_amd.so")
HOST_DIR = os.path.join(CSRC, "host")


def build_host(force=False):
    """C++ mirror of the reference's classes (namespace faiss) on top of the C ABI"""
    build(force)
    srcs = [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR)]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(p) <= os.path.getmtime(HOST_LIB) for p in srcs + [LIB]):
        return HOST_LIB
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", os.path.join(HOST_DIR, "faiss_amd.cpp"), "-o", HOST_LIB,
           "-L" + LIBDIR, "-launcel_amd", "-Wl,-rpath,$ORIGIN", "-pthread"]
    subprocess.run(cmd, check=True)
    return HOST_LIB


if __name__ == "__main__":
    print(build(force=True))
    print(build_host(force=True))
