"""ORACLE k-NN: ctypes front-end of oracle/knn_exact.c plus the reference's own op chain.

`knn_exact`  — platform-independent exact arithmetic (the oracle of record for indices).
`knn_torch`  — the literal op chain of sv_util.py:19-25 on torch CPU (depends on the host's
               BLAS rounding order; used for the cpu_baseline stopwatch and as a cross-check).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsvnet_oracle.so")
_lib = None

XX_OUTER = 0
XX_INNER = 1


def build(force=False):
    """Compile knn_exact.c (gcc). Called by __graft_entry__.build()."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "knn_exact.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.svnet_oracle_knn.restype = ctypes.c_int
        _lib.svnet_oracle_knn.argtypes = [ctypes.c_void_p] + [ctypes.c_int64] * 6 + [
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def xx_mode_of(x):
    """Which ATen reduction path `sum(x**2, dim=1)` takes for this [B,C,N] tensor (Appendix A)."""
    B, C, N = x.shape
    if x.stride(1) == 1 and C > 1:
        return XX_INNER           # transposed view of a [B,N,C] tensor (layers 2-4)
    return XX_OUTER               # contiguous [B,C,N] (layer 1)


def knn_exact(x, k, return_pd=False):
    """x: [B,C,N] float32 CPU tensor with arbitrary strides (as the reference passes it).
    Returns idx [B,N,k] int64 (cloud-local, nearest first)."""
    assert x.dtype == torch.float32 and x.device.type == "cpu" and x.dim() == 3
    B, C, N = x.shape
    idx = np.empty((B, N, k), dtype=np.int64)
    pd = np.empty((B, N, N), dtype=np.float32) if return_pd else None
    base = x if x.numel() == 0 else x
    rc = _load().svnet_oracle_knn(
        ctypes.c_void_p(base.data_ptr()), B, N, C, x.stride(0), x.stride(2), x.stride(1),
        xx_mode_of(x), int(k), idx.ctypes.data_as(ctypes.c_void_p),
        pd.ctypes.data_as(ctypes.c_void_p) if return_pd else None)
    if rc != 0:
        raise ValueError("svnet_oracle_knn failed with code %d" % rc)
    out = torch.from_numpy(idx)
    return (out, torch.from_numpy(pd)) if return_pd else out


def knn_torch(x, k):
    """sv_util.py:19-25 as written there: dense [B,N,N] matrix + topk."""
    gram = torch.matmul(x.transpose(2, 1), x)
    sq = (x ** 2).sum(dim=1, keepdim=True)
    neg_d2 = -sq - (-2 * gram) - sq.transpose(2, 1)
    return neg_d2.topk(k=k, dim=-1)[1]


def tie_aware_mismatches(idx_a, idx_b, pd):
    """Number of slots where two index sets REALLY disagree.  topk's order among exactly equal
    distances is implementation-defined, so a differing slot only counts
    when the distance values it selects differ as well."""
    idx_a, idx_b = idx_a.long(), idx_b.long()
    differ = idx_a != idx_b
    if not differ.any():
        return 0
    va = torch.gather(pd, -1, idx_a)
    vb = torch.gather(pd, -1, idx_b)
    return int((differ & (va != vb)).sum())
