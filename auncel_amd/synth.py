"""Deterministic synthetic datasets for the IVF-Flat hot path.

No real SIFT/DEEP/GIST data ships with the reference or this image, so every test and
benchmark uses blob mixtures shaped like those datasets.  Generators use the legacy
``numpy.random.RandomState`` streams (frozen by NumPy's compatibility policy) so that a
(seed, shape) pair always yields the same bytes; golden fixtures store the SHA-256 of the
inputs they were made from.
"""
import hashlib

import numpy as np


def _blobs(rs, n, d, centres, sigma, chunk=1 << 18):
    out = np.empty((n, d), dtype=np.float32)
    g = centres.shape[0]
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        c = rs.randint(0, g, size=i1 - i0)
        out[i0:i1] = centres[c] + rs.standard_normal((i1 - i0, d)).astype(np.float32) * np.float32(sigma)
    return out


def sift_like(nb, nq, d=128, nblobs=2000, sigma=18.0, seed=1234):
    """uint8-valued fp32 vectors (every partial L2 sum < 2**24 -> fp32 arithmetic is exact)."""
    rs = np.random.RandomState(seed)
    centres = rs.uniform(0, 160, size=(nblobs, d)).astype(np.float32)
    xb = np.floor(np.clip(_blobs(rs, nb, d, centres, sigma), 0, 255)).astype(np.float32)
    xq = np.floor(np.clip(_blobs(rs, nq, d, centres, sigma), 0, 255)).astype(np.float32)
    return xb, xq


def gauss_like(nb, nq, d=96, nblobs=200, sigma=0.6, seed=7):
    """generic float data (no exact arithmetic: summation order matters)."""
    rs = np.random.RandomState(seed)
    centres = rs.standard_normal((nblobs, d)).astype(np.float32)
    return _blobs(rs, nb, d, centres, sigma), _blobs(rs, nq, d, centres, sigma)


def deep_like(nb, nq, d=96, nblobs=20000, sigma=0.25, seed=1236):
    """L2-normalised float vectors, used with the inner-product metric."""
    rs = np.random.RandomState(seed)
    centres = rs.standard_normal((nblobs, d)).astype(np.float32)
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    xb = _blobs(rs, nb, d, centres, sigma / np.sqrt(d))
    xq = _blobs(rs, nq, d, centres, sigma / np.sqrt(d))
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    return xb.astype(np.float32), xq.astype(np.float32)


def gist_like(nb, nq, d=960, nblobs=2000, sigma=0.06, seed=1237):
    rs = np.random.RandomState(seed)
    centres = rs.uniform(0, 0.5, size=(nblobs, d)).astype(np.float32)
    xb = np.clip(_blobs(rs, nb, d, centres, sigma), 0, 1.5).astype(np.float32)
    xq = np.clip(_blobs(rs, nq, d, centres, sigma), 0, 1.5).astype(np.float32)
    return xb, xq


def sample_centroids(xb, nlist, seed=99):
    """centroids = distinct database rows (k-means itself is out of scope: SURVEY.md §2.1)."""
    rs = np.random.RandomState(seed)
    idx = rs.choice(xb.shape[0], size=nlist, replace=False)
    return np.ascontiguousarray(xb[np.sort(idx)])


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()
