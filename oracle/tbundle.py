"""TEST INFRASTRUCTURE ONLY (oracle/): Python side of the named-tensor container
described in oracle/tbundle.h.  Used by tests/ and tests/golden/make_golden.py to
talk to the C++ oracle programs; never imported by the product package."""
import struct

import numpy as np

_DT = {0: np.float32, 1: np.int64, 2: np.int32, 3: np.uint8, 4: np.float64, 5: np.uint64}
_RDT = {np.dtype(v): k for k, v in _DT.items()}
MAGIC = b"TBND1\0\0\0"


def save(path, tensors):
    """tensors: dict name -> array-like (python ints -> i64, floats -> f64)."""
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(tensors)))
        for name, v in tensors.items():
            if isinstance(v, bool):
                v = np.array([int(v)], dtype=np.int64)
            elif isinstance(v, (int, np.integer)):
                v = np.array([v], dtype=np.int64)
            elif isinstance(v, float):
                v = np.array([v], dtype=np.float64)
            a = np.ascontiguousarray(v)
            if a.dtype not in _RDT:
                raise TypeError(f"{name}: unsupported dtype {a.dtype}")
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<II", _RDT[a.dtype], a.ndim))
            f.write(struct.pack(f"<{a.ndim}Q", *a.shape))
            f.write(memoryview(a.reshape(-1)).cast("B") if a.size else b"")  # no copy: bench.py passes a 5 GB matrix


def load(path):
    out = {}
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("bad tbundle magic: " + str(path))
        (count,) = struct.unpack("<I", f.read(4))
        for _ in range(count):
            (nl,) = struct.unpack("<I", f.read(4))
            name = f.read(nl).decode()
            dt, ndim = struct.unpack("<II", f.read(8))
            dims = struct.unpack(f"<{ndim}Q", f.read(8 * ndim)) if ndim else ()
            n = int(np.prod(dims)) if ndim else 1
            dtype = np.dtype(_DT[dt])
            a = np.frombuffer(f.read(n * dtype.itemsize), dtype=dtype).reshape(dims)
            out[name] = a.copy()
    return out
