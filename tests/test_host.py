"""CPU tests of the host side: synthetic data, the C-ABI library surface, the no-fallback rule and the
data-parallel gradient bucket (gloo, world size 2)."""
import os
import json
import re
import socket
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
    header_abi = int(re.search(r"#define SVNET_ABI_VERSION (\d+)", header).group(1))
    assert L.svnet_version() == _lib.ABI_VERSION == header_abi          # library, binding and header speak the same ABI
    assert L.svnet_knn_workspace_bytes(2, 8, 3) >= (2 * 8 * 3 + 2 * 8) * 4


def test_argument_errors_do_not_need_a_gpu():
    from svnet_amd import _lib
    L = _lib.lib()
    assert L.svnet_knn_f32(None, 1, 8, 3, 24, 1, 8, 0, 2, None, None, 0, None) == -1
    assert b"null" in L.svnet_last_error()


def test_shape_queries_of_the_fused_launches_answer_without_a_gpu():
    """The host-side "is this shape taken" queries the Python layer asks before it chooses a fused launch (round 5: the k-NN table
    prepared by the previous level's apply pass, the one-launch block tail, the concatenation with the gate's sums): pure host
    functions - same answers here as on the GPU box, and the refusals the GPU tests rely on."""
    from svnet_amd import _lib
    L = _lib.lib()
    # k-NN table from the apply pass: whole 32-point tiles per cloud, 8 <= C <= 384, channel-major table (default build: always)
    assert L.svnet_knn_table_fusable(32, 1024, 62) == 1 and L.svnet_knn_table_fusable(32, 2048, 127) == 1
    assert L.svnet_knn_table_fusable(2, 100, 62) == 0 and L.svnet_knn_table_fusable(2, 1024, 4) == 0 and L.svnet_knn_table_fusable(0, 1024, 62) == 0
    assert L.svnet_knn_table_fusable(2, 1024, 385) == 0 and L.svnet_knn_table_fusable(2, 8192, 62) == 0
    # block tail: P a whole number of clouds of N % 32 == 0 points, widths <= 256
    assert L.svnet_block_tail_supported(32768, 1024, 128, 42, 0) == 1 and L.svnet_block_tail_supported(32768, 1024, 128, 42, 1) == 1
    assert L.svnet_block_tail_supported(200, 100, 32, 10, 0) == 0 and L.svnet_block_tail_supported(1000, 1024, 32, 10, 0) == 0
    assert L.svnet_block_tail_supported(2048, 1024, 300, 10, 0) == 0 and L.svnet_block_tail_supported(2048, 1024, 2, 1, 1) == 0      # (C = 5 < 8: no table)
    assert L.svnet_block_tail_supported(2048, 1024, 2, 1, 0) == 1
    # concatenation + per-cloud sums: 24 < C <= 192, pre_cols <= 8 lanes' worth, whole clouds of whole row blocks
    assert L.svnet_v2s_cat_sum_supported(32768, 83, 256, 1024) == 1 and L.svnet_v2s_cat_sum_supported(65536, 80, 256, 2048) == 1
    assert L.svnet_v2s_cat_sum_supported(96, 83, 256, 48) == 0 and L.svnet_v2s_cat_sum_supported(32768, 10, 64, 1024) == 0
    assert L.svnet_v2s_cat_sum_supported(32768, 83, 300, 1024) == 0 and L.svnet_v2s_cat_sum_supported(32768, 340, 512, 1024) == 0


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


def test_reference_checkpoints_load_with_and_without_the_dataparallel_prefix(tmp_path):
    """a checkpoint written the way the reference writes it (`module.`-prefixed keys inside
    {epoch, state_dict, optimizer, scheduler, best_test_acc}, utils.py:141-171) loads into the drop-in model, round-trips through
    save_checkpoint / load_checkpoint (latest.txt, model_best.pth, removal of the previous file), and keys are checked strictly."""
    import argparse
    import contextlib
    import io
    import torch
    import svnet_amd.models as M
    from svnet_amd import checkpoint as ck
    from oracle import params as oparams
    P = oparams.synthetic_params("sv_dgcnn_cls", binary=True, seed=5)
    with contextlib.redirect_stdout(io.StringIO()):
        a = M.SV_DGCNN_CLS(argparse.Namespace(k=20, binary=True), 40)
        b = M.SV_DGCNN_CLS(argparse.Namespace(k=20, binary=True), 40)
    a.load_state_dict(P)
    ref_style = ck.reference_state_dict(a)
    assert all(k.startswith("module.") for k in ref_style) and len(ref_style) == 112          # 112 keys
    state = {"epoch": 7, "state_dict": ref_style, "optimizer": {"lr": 1e-3}, "scheduler": {"last_epoch": 7}, "best_test_acc": 0.5}
    root = str(tmp_path)
    assert ck.save_checkpoint(state, 6, root, False, None) == 6
    assert ck.save_checkpoint(state, 7, root, True, 6) == 7
    files = sorted(os.listdir(os.path.join(root, "save_models")))
    assert files == ["checkpoint_007.pth", "latest.txt", "model_best.pth"]                      # epoch 6 removed: (6 + 1) % 20 > 0
    loaded = ck.load_checkpoint(root, resume=True)
    rest = ck.load_reference_checkpoint(b, loaded)
    assert rest["epoch"] == 7 and rest["best_test_acc"] == 0.5
    for (n, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), n
    ck.load_reference_checkpoint(b, P)                                                            # bare, un-prefixed state_dict
    bad = dict(ref_style)
    bad.pop("module.conv4.linear1.beta")
    with pytest.raises(RuntimeError):
        ck.load_reference_checkpoint(b, {"state_dict": bad})
    assert ck.load_checkpoint(root, test=os.path.join(root, "nope.pth")) is None


def test_params_macs_counters_reproduce_the_reference_numbers():
    """Params / MACs / ADDs / BOPs per cloud of the SV models as printed by the reference's params_macs scripts
    (probed in the survey: SV_DGCNN binary 50.84 M MACs + 207.26 M ADDs + 1 175.58 M BOPs, 3.43 Mbit; fp 1 433.69 M MACs,
    49.71 Mbit; part-seg 243.5 / 974.8 / 6 006.5 and 7 224.8; SV_PointNet 29.6 / 206.1 / 1 222.1 and 1 457.8)."""
    import argparse
    import contextlib
    import io
    import svnet_amd.models as M
    from svnet_amd import params_macs as pm

    def close(got, want, places):
        return all(abs(g - w) < 0.51 * 10 ** (-places) for g, w in zip(got, want))
    assert close(pm.sv_dgcnn_cls(True), (50.84, 207.26, 1175.58), 2)
    assert close(pm.sv_dgcnn_cls(False), (1433.69, 0.0, 0.0), 2)
    assert close(pm.sv_dgcnn_pseg(True), (243.5, 974.8, 6006.5), 1)
    assert close(pm.sv_dgcnn_pseg(False), (7224.8, 0.0, 0.0), 1)
    assert close(pm.sv_pointnet_cls(True), (29.6, 206.1, 1222.1), 1)
    assert close(pm.sv_pointnet_cls(False), (1457.8, 0.0, 0.0), 1)
    with contextlib.redirect_stdout(io.StringIO()):
        mb = M.SV_DGCNN_CLS(argparse.Namespace(k=20, binary=True), 40)
        mf = M.SV_DGCNN_CLS(argparse.Namespace(k=20, binary=False), 40)
    assert abs(pm.get_param(mb) - 3.43) < 0.005 and abs(pm.get_param(mf) - 49.71) < 0.005
    assert len(pm.report()) == 8


def test_deferred_weight_gradients_only_when_autograd_merely_adopts_them():
    """TrainStep hands autograd parameter gradients the side stream has not written yet (config.DEFER_WGRAD): sound only while every
    parameter belongs to one module, has no hooks, and .grad is None at backward.  The check itself needs no GPU."""
    from svnet_amd.train import TrainStep

    class Two(torch.nn.Module):
        def __init__(self, shared):
            super().__init__()
            self.a = torch.nn.Linear(4, 4, bias=False)
            self.b = torch.nn.Linear(4, 4, bias=False)
            if shared:
                self.b.weight = self.a.weight

    x, y = torch.zeros(2, 4), torch.zeros(2, dtype=torch.int64)
    plain = TrainStep(Two(False), (x,), y)
    assert plain._deferral_is_safe()
    assert not TrainStep(Two(True), (x,), y)._deferral_is_safe()          # one Parameter used by two modules: its gradients are ADDED
    hooked = Two(False)
    hooked.a.weight.register_hook(lambda g: g)
    assert not TrainStep(hooked, (x,), y)._deferral_is_safe()             # a tensor hook runs on the main stream, on unwritten memory
    if hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
        post = Two(False)
        post.b.weight.register_post_accumulate_grad_hook(lambda p: None)
        assert not TrainStep(post, (x,), y)._deferral_is_safe()


def _run_bench(argv, env_extra=None, launcher=False, timeout=300):
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable]
    if launcher:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    p = subprocess.run(cmd + [os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr.decode()


@pytest.mark.parametrize("launcher", [False, True], ids=["self_spawned", "torch_distributed_run"])
def test_bench_rank_plumbing_runs_end_to_end_on_two_gloo_ranks(launcher):
    """The N > 1 path of bench.py (the thing that replaces nn.DataParallel, main_cls_dgcnn.py:125,182-184) has never run on RCCL from this
    container.  Its control flow is the same function for both backends (bench.rank_main): two ranks on gloo / CPU with a stand-in step
    go through spawn (or the driver's `python -m torch.distributed.run` launch), the world-size check on the collective's own group, the
    barrier-bracketed timed region, the all_gather of the per-rank clocks, the max over ranks, the product's GradBucket.all_reduce_mean()
    and the rank-0-only JSON line with `rccl_world` and `collective_us`."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--selftest-cpu"], launcher=launcher)
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, (lines, err[-2000:])                      # rank 0 only, ONE line
    out = json.loads(lines[0])
    assert out["selftest"] is True and out["data"] == "selftest"      # can never be mistaken for a measurement
    assert out["n_gpus"] == 2 and out["config"]["rccl_world"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "weak" and out["config"]["global_batch"] == 64
    per_rank = out["per_rank_ms_per_step"]
    assert len(per_rank) == 2 and abs(out["ms_per_step"] - max(per_rank)) < 1e-9     # the job's time is the slowest rank's
    assert abs(out["value"] - 64 * 1e3 / out["ms_per_step"]) < 1e-2 * out["value"]   # whole-job throughput: all ranks' clouds / that time
    assert out["config"]["collective_us"] is not None and out["config"]["collective_us"] > 0
    assert "cpu_baseline" not in out and "other_workloads" not in out                # world-1-only legs stay out of the N > 1 line


def test_bench_parent_fails_when_a_rank_fails_and_refuses_a_wrong_world_size():
    """A rank that dies while the other sits in a collective must end the job with ITS exit code at once (the parent polls every child and
    kills the rest - waiting for rank 0 first would hang until the collective's timeout); `--gpus N` with another WORLD_SIZE is refused."""
    import time as _time
    t0 = _time.time()
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--selftest-cpu", "--selftest-fail-rank", "1"])
    assert rc == 7 and not lines, (rc, lines, err[-1000:])
    assert _time.time() - t0 < 120
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--selftest-cpu"],
                                env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and not lines and "WORLD_SIZE" in err


def test_a_second_backward_of_the_same_weights_is_not_deferred():
    """a module applied twice in one forward hands autograd TWO gradients of the same weights, which it adds on the main
    stream - the second must not be an unwritten side-stream tensor.  _Deferred.first_use() is the run-time consumer count (CPU logic)."""
    from svnet_amd import _ops
    d = _ops._Deferred()
    a, b, c = torch.zeros(3), torch.zeros(3), torch.zeros(3)
    assert d.first_use(a, b)
    assert d.first_use(c, None)
    assert not d.first_use(a)              # the same weights again in this backward: joined schedule
    assert not d.first_use(c, torch.zeros(2))
    d.join(torch.device("cpu"))            # (nothing kept: no stream is touched) - the next backward starts counting afresh
    assert d.first_use(a, b, c)
