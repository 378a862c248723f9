"""Builds libauncel_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libauncel_amd.so")
SOURCES = ["ivf_kernels.hip", "ivf_select.hip", "ivf_filter.hip", "dataset_io.cpp", "ivf_plan.hip", "ivf_kmeans.hip", "ivf_engine.hip"]
# -ffp-contract=off: products and sums are rounded separately, as in the reference's SSE build
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]


LAST = {"compiled": [], "reused": 0}  # what the last build() call did (printed by __graft_entry__.build)


def _deps(src):
    """a source and the headers next to it (plus the public header)"""
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return [os.path.join(CSRC, src)] + hdrs + [os.path.join(HERE, "..", "include", "auncel_amd.h")]


def build(force=False):
    """one object per source, compiled in parallel and only when stale, then linked"""
    from concurrent.futures import ThreadPoolExecutor
    extra = os.environ.get("AUNCEL_AMD_CXXFLAGS", "").split()
    stamp = os.path.join(LIBDIR, "flags.txt")  # objects built with other flags (experiments) are rebuilt
    want = " ".join(FLAGS + extra)
    same_flags = os.path.exists(stamp) and open(stamp).read() == want
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in FLAGS if f != "-shared"]
    todo = []
    for s in SOURCES:
        o = os.path.join(objdir, s + ".o")
        stale = force or not same_flags or not os.path.exists(o) or any(os.path.getmtime(p) > os.path.getmtime(o) for p in _deps(s))
        if stale:
            todo.append([hipcc] + cflags + extra + ["-c", os.path.join(CSRC, s), "-o", o])
    LAST["compiled"], LAST["reused"] = [os.path.basename(c[-3]) for c in todo], len(SOURCES) - len(todo)
    if not todo and os.path.exists(LIB) and same_flags:
        return LIB
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        for r in ex.map(lambda c: subprocess.run(c), todo):
            if r.returncode != 0:
                raise RuntimeError("hipcc failed")
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB], check=True)
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
