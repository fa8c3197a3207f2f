"""CPU tests of the host side: synthetic data, the C-ABI library surface, the no-fallback rule and the
data-parallel gradient bucket (gloo, world size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_is_deterministic_and_normalised():
    from svnet_amd import synth
    a = synth.cloud_batch(1234, 0, 0, 4, 256)
    b = synth.cloud_batch(1234, 0, 0, 4, 256)
    c = synth.cloud_batch(1234, 0, 1, 4, 256)
    assert a.dtype == np.float32 and a.shape == (4, 3, 256)
    assert (a == b).all() and not (a == c).all()
    assert abs(np.sqrt((a ** 2).sum(1)).max(axis=1) - 1.0).max() < 1e-6
    assert abs(a.mean(axis=2)).max() < 1e-6
    lab = synth.class_labels(1234, 0, 0, 64)
    assert lab.min() >= 0 and lab.max() < 40
    R = synth.random_rotation(1, 2)
    assert abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12


def test_library_exports_every_declared_symbol():
    """The .so loads without a GPU and exports exactly what include/svnet_hip.h declares."""
    from svnet_amd import _lib
    _lib.build()
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "svnet_hip.h")).read()
    declared = set(re.findall(r"\b(svnet_[a-z0-9_]+)\s*\(", header))
    declared.discard("svnet_gemm_desc")
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name
    assert L.svnet_version() == 100
    assert L.svnet_knn_workspace_bytes(2, 8, 3) >= (2 * 8 * 3 + 2 * 8) * 4


def test_argument_errors_do_not_need_a_gpu():
    from svnet_amd import _lib
    L = _lib.lib()
    assert L.svnet_knn_f32(None, 1, 8, 3, 24, 1, 8, 0, 2, None, None, 0, None) == -1
    assert b"null" in L.svnet_last_error()


def test_product_path_has_no_cpu_fallback():
    from svnet_amd.models.utils.sv_util import knn, svpool
    from svnet_amd.models.sv_layers import Linear
    with pytest.raises(RuntimeError):
        knn(torch.zeros(1, 3, 8), 2)
    with pytest.raises(RuntimeError):
        svpool((torch.zeros(1, 4, 2, 3), torch.zeros(1, 4, 2, 3, 2)))
    with pytest.raises(RuntimeError):
        Linear(4, 2, False, bw=True, ba=True)(torch.zeros(3, 4))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "svnet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_flat_gradient_bucket_allreduce_gloo_world2():
    """One process per rank, gloo on CPU: after the bucket all-reduce every rank holds the mean gradient,
    and parameter .grad tensors are views of the flat bucket (zero-copy)."""
    script = os.path.join(ROOT, "tests", "dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "OK" in o, o
