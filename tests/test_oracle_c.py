"""The plain-C restatement of the VQ search (oracle/vq_search.c) reproduces the real reference's
indices on the golden known-answer vectors, and agrees with the torch oracle on random data."""
import ctypes
import os
import subprocess

import numpy as np
import torch

from oracle import facodec_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "-s"], check=True)
    lib = ctypes.CDLL(os.path.join(REPO, "oracle", "_build", "libvq_oracle.so"))
    lib.oracle_vq_search.restype = ctypes.c_int
    lib.oracle_vq_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    return lib


def _search(lib, lat, cb):
    lat = np.ascontiguousarray(lat, dtype=np.float32)
    cb = np.ascontiguousarray(cb, dtype=np.float32)
    idx = np.empty(lat.shape[0], dtype=np.int64)
    assert lib.oracle_vq_search(lat.ctypes.data, cb.ctypes.data, idx.ctypes.data, lat.shape[0], cb.shape[0]) == 0
    return idx


def test_c_oracle_on_reference_known_answers(golden_dir):
    lib = _lib()
    d = np.load(os.path.join(golden_dir, "vq_kat.npz"))
    lat = d["latents"].transpose(0, 2, 1).reshape(-1, 8)
    assert np.array_equal(_search(lib, lat, d["codebook"]).reshape(4, 300), d["indices"].astype(np.int64))
    g = np.random.Generator(np.random.Philox(key=int(d["sweep_key"])))
    g.standard_normal((1024, 8)); g.standard_normal((4, 8, 300))
    big = g.standard_normal((1, 8, 1 << 18)).astype(np.float32)[0].T
    assert np.array_equal(_search(lib, big, d["codebook"]), d["sweep_indices"].astype(np.int64))


def test_c_oracle_equals_torch_oracle_on_random_codebooks():
    lib = _lib()
    g = torch.Generator().manual_seed(3)
    for K in (7, 64, 1024):
        cb = torch.randn(K, 8, generator=g)
        lat = torch.randn(1, 8, 5000, generator=g) * 0.3
        _, idx = O.vq_nearest(lat, cb)
        assert np.array_equal(_search(lib, lat[0].t().numpy(), cb.numpy()), idx.reshape(-1).numpy())
