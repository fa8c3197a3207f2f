#!/usr/bin/env python3
"""Headline benchmark: point-clouds/sec, forward (train mode) + cal_loss + backward (+ gradient all-reduce for
N > 1) of sv_dgcnn_cls --binary, B=32 per GPU, N=1024, k=20, synthetic clouds resident in HBM.

    python bench.py [--gpus N --steps K --warmup W] [--mode train|eval] [--workload dgcnn_cls|pointnet_fp|pointnet_bin|partseg]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--workload` selects another BASELINE.json config (same JSON contract, `config.workload` names it): pointnet_fp = config 2
(sv_pointnet_cls full precision, B=32 N=1024), pointnet_bin = config 1's model at B=32, partseg = config 5's per-GPU workload
(sv_dgcnn_partseg --binary, B=32 N=2048 k=40).  The default line (config 3 / 4) is unchanged.

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself: N fresh child processes, one per GPU, before
this process touches a GPU.  Rank 0 prints ONE JSON line (contract in the task statement).
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3    # f32-input MFMA = the f32 vector rate (MI355X_MICROARCH.md, matrix cores)
MFMA_I8_PEAK_TOPS = 5000.0      # int8 MFMA: 2x the bf16 rate per clock (same table)
B_PER_GPU, N_POINTS, K_NN = 32, 1024, 20
# the other BASELINE.json configs (see module docstring).  stage_bytes: SURVEY.md §8(d) algorithmic bytes of the k-NN + gather
# stages of one forward batch at B=32 (API-compatible, materialising form).
WORKLOADS = {
    "dgcnn_cls": dict(model="sv_dgcnn_cls", binary=True, B=32, N=1024, k=20, stage_bytes=1440.2e6,
                      name="sv_dgcnn_cls --binary"),
    "pointnet_fp": dict(model="sv_pointnet_cls", binary=False, B=32, N=1024, k=20, stage_bytes=34.9e6,
                        name="sv_pointnet_cls full precision"),
    "pointnet_bin": dict(model="sv_pointnet_cls", binary=True, B=32, N=1024, k=20, stage_bytes=34.9e6,
                         name="sv_pointnet_cls --binary"),
    "partseg": dict(model="sv_dgcnn_pseg", binary=True, B=32, N=2048, k=40, stage_bytes=6595.0e6,
                    name="sv_dgcnn_partseg --binary"),
}
# algorithmic bytes of the four k-NN + gather stages of one forward batch of B=32 (API-compatible,
# materialising form: x read twice, idx written and read, edge features written) = 27.0 + 351.8 + 351.8 + 709.6 MB
KNN_GATHER_STAGE_BYTES = 1440.2e6


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=("train", "eval"), default="train",
                    help="train: fwd+loss+bwd (the headline metric); eval: forward only, eval(), no_grad (SURVEY §8d secondary)")
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="dgcnn_cls",
                    help="which BASELINE.json config to run (default: the headline one, sv_dgcnn_cls --binary B=32 N=1024 k=20)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="default workload only: skip the other BASELINE configs' legs (other_workloads) and the train-loop leg")
    ap.add_argument("--force-collective", action="store_true",
                    help="--gpus 1 only: run the RCCL all-reduce(avg) of the real gradient bucket inside the timed loop at world size 1 "
                         "(the collective's fixed cost on this box: collective_us)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="NOT a benchmark: drive this file's rank plumbing (spawn, world-size check, process group, barrier, all_gather of the "
                         "per-rank clocks, max over ranks, rank-0-only JSON line, collective_us) on gloo / CPU with a stand-in step - "
                         "tests/test_host.py rehearses the N > 1 path with it where no GPU exists")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help="--selftest-cpu: this rank exits with code 7 before the timed region")
    return ap.parse_args()


# ----------------------------------------------------------------------------- multi-GPU launch

def spawn_ranks(args):
    """Start one fresh process per GPU (this process has not touched a GPU and never does), wait for all of them."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        # poll ALL children: a rank that dies while the others sit in a collective must take the job down at once (waiting for rank 0
        # first would hang until the collective's timeout)
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = rc or code
            if live and rc == 0:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                   # exact PIDs we started, never a pattern
                p.wait()
    return rc


# ----------------------------------------------------------------------------- CPU baseline (the checker, timed)

def cpu_info():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1))


def cpu_share():
    """Host cores this process may really use: affinity mask and cgroup CPU quota.  A 1-GPU box of the pool shows 256 logical CPUs and
    a quota of 16 cores: torch's default 128 threads ran the oracle 2.7x SLOWER there than 16 (profiles/r05_cpu_threads.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(wl, sample_b=None, timed=3, threads=None):
    """The oracle (CPU restatement of the reference, kind "port") timed on this box's host cores on the SAME workload (same model,
    N, k and - round 5 - the same B; `sample_b` bounds it for the slower workloads): SURVEY §8(d) / BASELINE.md §3 protocol, 1 warm-up +
    3 timed, median, both fwd+cal_loss+bwd (train) and forward only (eval, no_grad).  Threads = the cores this process may use
    (cpu_share(): the cgroup quota), not the logical CPU count."""
    import torch
    from svnet_amd import synth
    from oracle import params as oparams, sv_ref
    model, phys = cpu_info()
    if sample_b is None:
        sample_b = wl["B"]
    share = cpu_share()
    threads = int(threads) if threads else min(torch.get_num_threads(), share)
    torch.set_num_threads(threads)
    name, binary, N, k = wl["model"], wl["binary"], wl["N"], wl["k"]
    P = oparams.synthetic_params(name, binary=binary, seed=1234, requires_grad=True)
    x = torch.from_numpy(synth.cloud_batch(1234, 0, 0, sample_b, N))
    seg = name == "sv_dgcnn_pseg"
    if seg:
        lab = torch.from_numpy(synth.category_onehot(1234, 0, 0, sample_b))
        y = torch.from_numpy(synth.seg_labels(1234, 0, 0, sample_b, N))
    else:
        y = torch.from_numpy(synth.class_labels(1234, 0, 0, sample_b))

    def fwd(ctx):
        if seg:
            lo = sv_ref.sv_dgcnn_pseg(x, lab, P, k, binary, ctx)
            return lo.permute(0, 2, 1).reshape(-1, lo.shape[1])
        return (sv_ref.sv_dgcnn_cls if name == "sv_dgcnn_cls" else sv_ref.sv_pointnet_cls)(x, P, k, binary, ctx)

    def train_once():
        for p in P.values():
            p.grad = None
        sv_ref.cal_loss(fwd(sv_ref.Ctx(train=True, knn="torch")), y.reshape(-1)).backward()

    def eval_once():
        with torch.no_grad():
            fwd(sv_ref.Ctx(train=False, knn="torch"))

    def median_time(fn):
        ts = []
        for it in range(1 + timed):
            t0 = time.perf_counter()
            fn()
            if it > 0:
                ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    t_train, t_eval = median_time(train_once), median_time(eval_once)
    return {"value": round(sample_b / t_train, 4), "unit": "point-clouds/sec", "cores": threads, "kind": "port",
            "forward_only_value": round(sample_b / t_eval, 4), "cpu_model": model, "physical_cores": phys, "cpu_share": share,
            "sample": "oracle (torch CPU ops, %d threads = this process's CPU share of %d cores) of %s at B=%d N=%d k=%d (the bench's own batch: "
                      "profiles/r05_cpu_baseline_table.txt shows the per-cloud rate flat in B): median of %d after 1 warm-up; "
                      "fwd+loss+bwd %.2f s, forward only (eval, no_grad) %.2f s per batch"
                      % (threads, share, wl["name"], sample_b, N, k, timed, t_train, t_eval)}


# ----------------------------------------------------------------------------- roofline legs

def measured_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of THIS build (profiles/r05_pmc_traffic.json, written by
    tools/make_traffic_json.py from separate FETCH_SIZE / WRITE_SIZE passes; units and gfx950 corrections as MI355X_MICROARCH.md
    prescribes: 2 x FETCH_SIZE + WRITE_SIZE)."""
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
            k = rec["kernels"][kernel_key]
            return k["traffic_bytes"], "profiles/%s (commit %s)" % (name, rec.get("commit", "?"))
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def avg_ms(timer):
    ms = timer.elapsed_ms()
    return (sum(ms) / len(ms)) if ms else None


def dgcnn_cls_legs(args, model, x, train, work, torch, _lib):
    """Roofline legs of the headline workload: returns (per_launch, stages)."""
    stages = None
    per_launch = None
    rank = 0
    # ---- roofline legs: HIP events around single entry points, eager launches on the stream each kernel is launched on,
    # after the timed region (same kernels, same arguments as the replayed graph)
    stages = None
    per_launch = None
    if rank == 0:
        P_, E = B_PER_GPU * N_POINTS, B_PER_GPU * N_POINTS * K_NN
        t_tile = _lib.KernelTimer("svnet_edgeblock_bwd_f32", lambda a: a[0]._obj.parts == 2 and a[0]._obj.Os == 128)
        t_knn = _lib.KernelTimer("svnet_knn_f32")
        t_knn2 = _lib.KernelTimer("svnet_knn_sv_f32")
        t_efwd = _lib.KernelTimer("svnet_edgeblock_fwd_f32")
        t_xfwd = _lib.KernelTimer("svnet_xyzblock_fwd_f32")
        t_rows = _lib.KernelTimer("svnet_gemm_f32", lambda a: (a[0]._obj.M == P_ and a[0]._obj.N == 505 and a[0]._obj.K == 512
                                                               and a[0]._obj.b_exact and not a[0]._obj.a_sign))
        t_tn = _lib.KernelTimer("svnet_edgeblock_wgrad_f32", lambda a: a[9] == 128)            # conv4: Os = 128
        t_i8 = _lib.KernelTimer("svnet_binlinear_i8_fwd_f32", lambda a: a[6] == P_ and a[7] == 505 and a[8] == 512)   # conv5.linear1
        _lib.TIMERS[:] = [t_tile, t_knn, t_knn2, t_efwd, t_xfwd, t_rows, t_tn, t_i8]
        reps = 3
        # (the stage legs time the k-NN calls in their SELF-CONTAINED form - table preparation, distances, selection: in the step itself
        #  the preparation of the three feature-space graphs runs inside the previous level's apply pass, config.KNN_TABLE_AHEAD, and
        #  would drop out of every timer here)
        from svnet_amd import config as _cfg
        ahead, _cfg.KNN_TABLE_AHEAD = _cfg.KNN_TABLE_AHEAD, False
        try:
            for _ in range(reps):
                if args.mode == "train":
                    train.fwd_bwd()
                else:
                    work.forward()
            torch.cuda.synchronize()
        finally:
            _cfg.KNN_TABLE_AHEAD = ahead
        _lib.TIMERS[:] = []

        # (iii) dominant kernel of the step: edgeblock_bwd_kernel<0,8,44>, the 32-edge tile kernel of conv4's fused backward (one launch
        # per step).  ALGORITHMIC bytes = SURVEY.md §8(d)'s figure for the backward of this gather stage (K10: the fp32 edge-feature
        # gradient read, the neighbour ids read, the point gradients written - what an API-compatible implementation must move):
        # E*(2Cs+6Cv)*4 + E*8 + P*(Cs+3Cv)*4.  `kernel_bytes` = what THIS kernel's design moves per launch: per edge the
        # kept n (2 B x Os), planes (120 B) and neighbour id read, the row-sliced planes (80 B) and the message row (Cs+3Cv+9 floats)
        # written; per point the pooled-edge operands / v / zz read and the centre sums written (round 1 also wrote dL/dy, 4 B x Os per
        # edge: the weight-gradient GEMM now recomputes it from n).
        ms = avg_ms(t_tile)
        if ms:
            Cs, Cv, Os, Ov = 64, 21, 128, 42
            dur = ms * 1e-3
            alg = E * (2 * Cs + 6 * Cv) * 4 + E * 8 + P_ * (Cs + 3 * Cv) * 4
            reads = E * (2 * Os + 120 + 8) + P_ * (Os * (4 + 2) + 4 * (3 * Cv + 18))
            writes = E * (80 + 4 * (Cs + 3 * Cv + 9)) + P_ * 4 * (Cs + 3 * Cv + 9)
            own = reads + writes
            traffic, src = measured_traffic("edgeblock_bwd_conv4")
            per_launch = {"bound": "hbm", "achieved": round(alg / dur / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(alg / dur / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                          "kernel": "edgeblock_bwd_kernel<0,8,44> (conv4 backward tile kernel: Cs=64 Cv=21 -> Os=128 Ov=42)",
                          "avg_launch_us": round(dur * 1e6, 1), "algorithmic_bytes": alg,
                          "algorithmic_bytes_source": "SURVEY.md §8(d) K10, conv4 stage: 665.9 MB dEdge + 5.2 MB idx + 16.7 MB dx",
                          "kernel_bytes": own, "kernel_bytes_frac": round(own / dur / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "instruction/latency-bound (64-lane waves carry 21..64 channels), not bandwidth-bound; see DESIGN.md"}
        # (i) the stage north_star's ">= 40 % HBM" is attached to: the four k-NN + gather stages of one forward batch, which here are
        # the 4 k-NN launches plus the fused gather+block+pool kernels (the gather is not a kernel of its own any more)
        stages = {}
        knn_ms, ef_ms, xf_ms = t_knn.elapsed_ms() + t_knn2.elapsed_ms(), t_efwd.elapsed_ms(), t_xfwd.elapsed_ms()
        if knn_ms and ef_ms and xf_ms:
            t_stage = (sum(knn_ms) + sum(ef_ms) + sum(xf_ms)) / reps * 1e-3
            t_knn_only = sum(knn_ms) / reps * 1e-3
            stages["knn_gather_forward"] = {
                "bound": "hbm", "algorithmic_bytes": KNN_GATHER_STAGE_BYTES, "time_ms": round(t_stage * 1e3, 4),
                "knn_only_ms": round(t_knn_only * 1e3, 4),
                # the stage with the k-NN term split out (VERDICT r4 #6c): the four fused gather + SVBlock + pooling kernels alone
                # against the same bytes - what the gather half reaches; the exact k-NN's own floor is profiles/r05_knn_floor.txt
                "gather_block_pool_ms": round((t_stage - t_knn_only) * 1e3, 4),
                "frac_without_knn": round(KNN_GATHER_STAGE_BYTES / (t_stage - t_knn_only) / 1e9 / HBM_PEAK_GBS, 4),
                "achieved": round(KNN_GATHER_STAGE_BYTES / t_stage / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(KNN_GATHER_STAGE_BYTES / t_stage / 1e9 / HBM_PEAK_GBS, 4),
                "what": "SURVEY §8(d) bytes of the 4 k-NN + gather stages / (svnet_knn_f32 + 3 x svnet_knn_sv_f32 + xyzblock_fwd + 3 x edgeblock_fwd); the fused "
                        "kernels also do the SVBlock and the pooling of each stage, so this UNDER-states the gather's own bandwidth.  The k-NN calls "
                        "are timed self-contained (own table preparation); in the step that preparation runs inside the previous level's apply pass"}
            # the exact k-NN against the pipe it runs on since round 4: the fmaf chain of every (query, candidate, channel) on
            # v_mfma_f32_16x16x4_f32 (csrc/knn.hip knn_mf8_kernel); the selection and the table preparation are in the time, not in the flops
            knn_flops = 2.0 * B_PER_GPU * N_POINTS * N_POINTS * (3 + 62 + 62 + 127)
            stages["knn_exact_f32_mfma"] = {
                "bound": "mfma", "flops_f32": knn_flops, "time_ms": round(t_knn_only * 1e3, 4),
                "achieved": round(knn_flops / t_knn_only / 1e12, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(knn_flops / t_knn_only / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                "what": "2 B N^2 sum(C) flops of the four graphs (C = 3, 62, 62, 127) / (4 k-NN calls: table preparation + distances + exact "
                        "top-k selection); the f32 matrix pipe equals the f32 vector rate and does not overlap vector work (DESIGN.md 4.6)"}
        # (i') the same stage in its API-compatible, MATERIALISING form (what SURVEY §8(d)'s byte count describes): the k-NN calls as
        # measured above plus the tier-1 gather kernels (svnet_edge_xyz_f32 / svnet_edge_diffcat_fwd_f32) writing the fp32 edge
        # tensors of the four stages, on this model's own point tables and graphs.  The fused path never performs this traffic; the
        # entry shows what the gather kernels themselves reach when they do.
        try:
            from svnet_amd import _ops as ops
            import svnet_amd.models.sv_dgcnn_cls as MD
            taps, pool = [], MD.svpool

            def tapped(*a, **kw):
                out = pool(*a, **kw)
                taps.append(out)
                return out
            MD.svpool = tapped
            try:
                with torch.no_grad():
                    model(x)
            finally:
                MD.svpool = pool
            graphs = [ops.knn(x, K_NN)] + [ops.knn_sv(s_.contiguous(), v_.contiguous(), K_NN) for s_, v_ in taps[:3]]
            t_g = [_lib.KernelTimer("svnet_edge_xyz_f32"), _lib.KernelTimer("svnet_edge_diffcat_fwd_f32")]
            _lib.TIMERS[:] = t_g
            for _ in range(reps):
                ops.edge_xyz(x, graphs[0], 0)
                for (s_, v_), g_ in zip(taps[:3], graphs[1:]):
                    B_, N_, Cs_ = s_.shape
                    ops.EdgeDiffcat.apply(s_.reshape(B_, N_, 1, Cs_), g_, False, K_NN)
                    ops.EdgeDiffcat.apply(v_, g_, False, K_NN)
            torch.cuda.synchronize()
            _lib.TIMERS[:] = []
            t_gather = sum(sum(t.elapsed_ms()) for t in t_g) / reps * 1e-3
            if knn_ms and t_gather > 0:
                t_mat = t_gather + sum(knn_ms) / reps * 1e-3
                gather_bytes = KNN_GATHER_STAGE_BYTES - 4 * 2 * P_ * K_NN * 8 / 2      # without the id write of the k-NN kernels
                stages["knn_gather_forward_materialising"] = {
                    "bound": "hbm", "algorithmic_bytes": KNN_GATHER_STAGE_BYTES, "time_ms": round(t_mat * 1e3, 4),
                    "gather_only_ms": round(t_gather * 1e3, 4), "achieved": round(KNN_GATHER_STAGE_BYTES / t_mat / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(KNN_GATHER_STAGE_BYTES / t_mat / 1e9 / HBM_PEAK_GBS, 4),
                    "gather_only_frac": round(gather_bytes / t_gather / 1e9 / HBM_PEAK_GBS, 4),
                    "what": "same bytes / (4 k-NN calls + the tier-1 gather kernels that write the fp32 edge tensors); not part of the timed step"}
        except Exception as e:                                        # diagnostic leg only
            print("materialising-gather leg skipped: %r" % (e,), file=sys.stderr)
        # (ii) MFMA utilisation of the two dense products north_star names (vs the 2.5 PFLOP/s dense bf16 peak; fp32 operands are
        # split exactly into 3 bf16 pieces, so 3 MFMA passes per fp32 product)
        ms = avg_ms(t_rows)
        if ms:
            fl = 2.0 * P_ * 505 * 512 * 3
            stages["mfma_rows_conv5_dx"] = {"bound": "mfma", "flops_bf16": fl, "time_us": round(ms * 1e3, 1),
                                            "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                            "fp32_equivalent_tflops": round(fl / 3 / (ms * 1e-3) / 1e12, 1),
                                            "what": "dx = (g*scale) . sign(W1) of conv5.linear1: [32768 x 512] x [512 x 505]"}
        ms = avg_ms(t_i8)
        if ms:
            ops = 2.0 * P_ * 505 * 512
            stages["binlinear_i8_conv5"] = {"bound": "mfma", "ops_int8": ops, "time_us": round(ms * 1e3, 1),
                                            "achieved": round(ops / (ms * 1e-3) / 1e12, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                                            "frac": round(ops / (ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, 4),
                                            "hbm_gbs": round((P_ * 505 * 4 + P_ * 512 * 4 + 3 * P_ * 505 / 8) / (ms * 1e-3) / 1e9, 1),
                                            "what": "conv5.linear1 forward: ternary [32768 x 505] x [505 x 512] on v_mfma_i32_32x32x32_i8 (exact integer counts); "
                                                    "the kernel is bound by staging its fp32 operand (hbm_gbs = x read + y written + planes written), not by the matrix pipe"}
        ms = avg_ms(t_tn)
        if ms:
            fl = 2.0 * 320 * 128 * E * 3
            stages["mfma_tn_tern_conv4_dW"] = {"bound": "mfma", "flops_bf16": fl, "time_us": round(ms * 1e3, 1),
                                               "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                               "what": "GX[320 x 128] = x_b^T . dy over E = 655 360 edge rows (ternary planes x fp32, padding columns included)"}

    return per_launch, stages


def graph_kernel_nodes(step):
    """Kernel nodes of the captured HIP graph (= launches per replayed step), or None when the graph handle is not reachable."""
    try:
        import ctypes
        import torch
        g = torch.cuda.CUDAGraph(keep_graph=True)
    except Exception:
        return None
    try:
        with torch.cuda.graph(g, stream=step._stream if getattr(step, "_stream", None) is not None else torch.cuda.Stream()):
            step.fwd_bwd(planes_external=True) if hasattr(step, "fwd_bwd") else step.forward(planes_external=True)
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(ctypes.c_void_p(g.raw_cuda_graph()), None, ctypes.byref(n)) != 0:
            return None
        nodes = (ctypes.c_void_p * n.value)()
        hip.hipGraphGetNodes(ctypes.c_void_p(g.raw_cuda_graph()), nodes, ctypes.byref(n))
        kernels = 0
        for nd in nodes:
            t = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t)) == 0 and t.value == 0:    # hipGraphNodeTypeKernel
                kernels += 1
        return {"kernel_nodes": kernels, "all_nodes": int(n.value)}
    except Exception:
        return None


def other_workload_legs(args, wl, model, inputs, train, work, torch, _lib):
    """Roofline legs of the non-headline workloads: the k-NN + gather stage against SURVEY §8(d)'s bytes (hbm) and the dominant
    dense product against the matrix-core peak of its arithmetic type (mfma).  Eager launches after the timed region, HIP events
    on the launch stream."""
    B, N, k = wl["B"], wl["N"], wl["k"]
    P_ = B * N
    t_knn, t_knn2 = _lib.KernelTimer("svnet_knn_f32"), _lib.KernelTimer("svnet_knn_sv_f32")
    t_efwd, t_xfwd = _lib.KernelTimer("svnet_edgeblock_fwd_f32"), _lib.KernelTimer("svnet_xyzblock_fwd_f32")
    t_exyz = _lib.KernelTimer("svnet_edge_xyz_f32")
    timers = [t_knn, t_knn2, t_efwd, t_xfwd, t_exyz]
    dense = None
    if wl["model"] == "sv_pointnet_cls" and not wl["binary"]:
        # config 2's dominant dense product: conv_fuse.linear1, [B*N, 2044] x [2044, 512] in fp32 (sv_pointnet_cls.py:26,53)
        # (priced at what the kernel issues: SIX bf16 MFMA products per fp32 product - the leading terms of the exact three-way splits of
        #  both operands - against the dense bf16 peak.  Against the f32-input MFMA peak the same time reads > 1.0: the split product is
        #  faster than the fp32 matrix pipe, which is why it is used.)
        dense = ("svnet_gemm_f32", lambda a: a[0]._obj.M == P_ and a[0]._obj.N == 512 and a[0]._obj.K == 2044 and not a[0]._obj.b_exact,
                 6 * 2.0 * P_ * 512 * 2044, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s",
                 "conv_fuse.linear1 forward: [%d x 2044] x [2044 x 512], fp32 operands as six bf16 MFMA products per fp32 product "
                 "(fp32-equivalent rate = achieved / 6)" % P_)
    elif wl["model"] == "sv_pointnet_cls":
        dense = ("svnet_binlinear_i8_fwd_f32", lambda a: a[6] == P_ and a[7] == 2044 and a[8] == 512,
                 2.0 * P_ * 512 * 2044, MFMA_I8_PEAK_TOPS, "TOP/s",
                 "conv_fuse.linear1 forward: ternary [%d x 2044] x [2044 x 512] on v_mfma_i32_32x32x32_i8 (exact integer counts)" % P_)
    elif wl["model"] == "sv_dgcnn_pseg":
        dense = ("svnet_binlinear_i8_fwd_f32", lambda a: a[6] == P_ and a[7] == 2144 and a[8] == 256,
                 2.0 * P_ * 256 * 2144, MFMA_I8_PEAK_TOPS, "TOP/s",
                 "conv8 (seg head) forward: ternary [%d x 2144] x [2144 x 256] on v_mfma_i32_32x32x32_i8" % P_)
    t_dense = None
    if dense:
        t_dense = _lib.KernelTimer(dense[0], dense[1])
        timers.append(t_dense)
    _lib.TIMERS[:] = timers
    reps = 3
    from svnet_amd import config as _cfg
    ahead, _cfg.KNN_TABLE_AHEAD = _cfg.KNN_TABLE_AHEAD, False          # (the k-NN calls in their self-contained form: see roofline_legs)
    try:
        for _ in range(reps):
            if args.mode == "train":
                train.fwd_bwd()
            else:
                work.forward()
        torch.cuda.synchronize()
    finally:
        _cfg.KNN_TABLE_AHEAD = ahead
    _lib.TIMERS[:] = []
    stages = {}
    knn_ms = t_knn.elapsed_ms() + t_knn2.elapsed_ms()
    gat_ms = t_efwd.elapsed_ms() + t_xfwd.elapsed_ms() + t_exyz.elapsed_ms()
    if knn_ms and gat_ms:
        t_stage = (sum(knn_ms) + sum(gat_ms)) / reps * 1e-3
        stages["knn_gather_forward"] = {
            "bound": "hbm", "algorithmic_bytes": wl["stage_bytes"], "time_ms": round(t_stage * 1e3, 4),
            "knn_only_ms": round(sum(knn_ms) / reps, 4), "achieved": round(wl["stage_bytes"] / t_stage / 1e9, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(wl["stage_bytes"] / t_stage / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
            "what": "SURVEY §8(d) bytes of the k-NN + gather stages of one forward batch / (the k-NN launches + the gather kernels: fused "
                    "gather+SVBlock+pool kernels where the layer is fused, svnet_edge_xyz_f32 where the edges are materialised)"}
    if t_dense is not None:
        ms = avg_ms(t_dense)
        if ms:
            stages["dominant_dense_product"] = {
                "bound": "mfma", "ops": dense[2], "time_us": round(ms * 1e3, 1), "achieved": round(dense[2] / (ms * 1e-3) / 1e12, 2),
                "peak": dense[3], "unit": dense[4], "frac": round(dense[2] / (ms * 1e-3) / 1e12 / dense[3], 4), "traffic": None,
                "what": dense[5]}
    primary = stages.get("dominant_dense_product") if wl["model"] == "sv_pointnet_cls" else stages.get("knn_gather_forward")
    return primary, stages


def build_workload(name, dev, rank=0, B=None):
    """(wl, model, inputs, target, loss_fn) of one BASELINE.json config: random-init weights of that architecture (manual_seed(0)),
    rank-indexed synthetic clouds / labels resident on `dev`."""
    import torch
    import svnet_amd.models as M
    from svnet_amd import synth
    from svnet_amd.train import cal_loss, seg_loss
    wl = dict(WORKLOADS[name])
    if B is not None:
        wl["B"] = int(B)
    B, N, k = wl["B"], wl["N"], wl["k"]
    torch.manual_seed(0)
    cls, nc = {"sv_dgcnn_cls": (M.SV_DGCNN_CLS, 40), "sv_pointnet_cls": (M.SV_PointNet_CLS, 40), "sv_dgcnn_pseg": (M.SV_DGCNN_PSEG, 50)}[wl["model"]]
    with contextlib.redirect_stdout(io.StringIO()):
        model = cls(argparse.Namespace(k=k, binary=wl["binary"], dropout=0.5), nc).to(dev)
    x = torch.from_numpy(synth.cloud_batch(1234, 0, rank, B, N)).to(dev)              # rank-indexed synthetic clouds
    if wl["model"] == "sv_dgcnn_pseg":
        inputs = (x, torch.from_numpy(synth.category_onehot(1234, 0, rank, B)).to(dev))
        y = torch.from_numpy(synth.seg_labels(1234, 0, rank, B, N)).to(dev)
        loss_fn = seg_loss
    else:
        inputs = (x,)
        y = torch.from_numpy(synth.class_labels(1234, 0, rank, B)).to(dev)
        loss_fn = cal_loss
    return wl, model, inputs, y, loss_fn


def other_workloads_leg(args, dev, torch, _lib, steps=10):
    """BASELINE.json configs 1 / 2 / 5's per-GPU workloads next to the headline line: capture + `steps` replays each,
    no CPU leg.  {name: {ms_per_step, value, graph_nodes, roofline, ...}}; a workload that fails reports its error instead."""
    from svnet_amd.train import TrainStep
    import gc
    out = {}
    for name in ("pointnet_fp", "pointnet_bin", "partseg"):
        try:
            wl, model, inputs, y, loss_fn = build_workload(name, dev)
            step = TrainStep(model.train(), inputs, y, loss_fn)
            step.capture()
            for _ in range(2):
                step.run(all_reduce=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step.run(all_reduce=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            loss = float(step.loss)
            sub = argparse.Namespace(mode="train")
            primary, stages = other_workload_legs(sub, wl, model, inputs, step, step, torch, _lib)
            nodes = graph_kernel_nodes(step)
            out[name] = {"workload": "%s fwd+loss+bwd, B=%d N=%d k=%d" % (wl["name"], wl["B"], wl["N"], wl["k"]),
                         "ms_per_step": round(dt / steps * 1e3, 3), "value": round(wl["B"] * steps / dt, 2), "unit": "point-clouds/sec",
                         "steps": steps, "launch": "hipGraph replay", "graph_nodes": nodes, "loss_finite": bool(loss == loss and abs(loss) < 1e6),
                         "roofline": primary, "roofline_stages": stages}
            del step, model, inputs, y
        except Exception as e:                                        # a secondary leg must not take the headline line down
            out[name] = {"error": repr(e)[:300]}
        gc.collect()
        torch.cuda.empty_cache()
    return out


def train_loop_leg(model, inputs, y, loss_fn, torch, steps=20):
    """What a TRAINING step costs beyond the timed fwd+loss+bwd replay: the flat Adam step (one kernel over
    the flat parameter / gradient buffers, main_cls_dgcnn.py:132-133,185) and, before the next replay, the re-pack of the binarized
    weights it changed (_ops.PLANES.refresh: what sv_layers.py:44-48 re-derives inside every forward).  The parameters are re-homed
    into one flat buffer first, so the step is captured anew (its graph bakes the parameter addresses in)."""
    from svnet_amd.train import FlatAdam, FlatParams, TrainStep
    flat = FlatParams(model)
    step = TrainStep(model.train(), inputs, y, loss_fn)
    step.capture()
    opt = FlatAdam(flat, step.bucket, lr=1e-3)

    def loop(n, with_opt):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step.run(all_reduce=False)
            if with_opt:
                opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    loop(3, True)
    t_eager = loop(steps, True)
    opt.capture()                                         # the update kernel + every re-pack as ONE graph (svnet_amd.train)
    loop(3, True)
    t_train = loop(steps, True)
    loss = float(step.loss)
    t_replay = loop(steps, False)
    return {"train_loop_ms_per_step": round(t_train, 3), "replay_only_ms_per_step": round(t_replay, 3),
            "optimizer_and_repack_ms": round(t_train - t_replay, 3), "eager_optimizer_train_loop_ms_per_step": round(t_eager, 3),
            "steps": steps, "loss_after": round(loss, 6),
            "what": "hipGraph replay of fwd+loss+bwd + a second graph [FlatAdam's update kernel -> re-pack of the binarized weights] per step, "
                    "%d steps on one batch; eager_optimizer = the same with the update and the ~25 re-pack launches issued one by one; "
                    "replay_only = the first graph alone (nothing stale, nothing re-packed)" % steps}


class _SelftestStep:
    """Stand-in for svnet_amd.train.TrainStep in --selftest-cpu runs (same surface: capture / run / fwd_bwd / bucket / loss): a tiny
    torch CPU model whose gradients live in the product's GradBucket and go through the product's all_reduce_mean() on gloo."""

    def __init__(self, rank):
        import torch
        from svnet_amd.dist import GradBucket
        torch.manual_seed(0)                                          # same weights on every rank
        self.model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.BatchNorm1d(32), torch.nn.ReLU(), torch.nn.Linear(32, 40))
        gen = torch.Generator().manual_seed(100 + rank)               # rank-indexed inputs
        self.x, self.y = torch.randn(32, 16, generator=gen), torch.randint(0, 40, (32,), generator=gen)
        self.bucket = GradBucket(self.model.parameters())
        self.graph, self.loss = None, None

    def capture(self):
        return self

    def fwd_bwd(self, planes_external=False):
        import torch
        self.bucket.zero()
        loss = torch.nn.functional.cross_entropy(self.model(self.x), self.y)
        loss.backward()
        self.loss = loss.detach()
        return self.loss

    def run(self, all_reduce=True):
        loss = self.fwd_bwd()
        if all_reduce:
            self.bucket.all_reduce_mean()
        return loss


def rank_main(args):
    """One rank of the job (the whole job at --gpus 1).  The product run talks RCCL ("nccl") from cuda:LOCAL_RANK; --selftest-cpu runs
    the SAME control flow - process group, world-size check, barrier, per-rank clocks, max over ranks, rank-0-only line - on gloo / CPU
    with a stand-in step, so that the N > 1 path has executed before the driver's 8-GPU node runs it (tests/test_host.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    selftest = args.selftest_cpu
    wl = WORKLOADS[args.workload]
    B, N, k = wl["B"], wl["N"], wl["k"]
    if args.force_collective and world != 1:
        print("bench.py: --force-collective is the world-size-1 leg (the N > 1 runs always reduce)", file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist
    backend = "gloo" if selftest else "nccl"
    collective = world > 1 or args.force_collective
    if not selftest and local >= torch.cuda.device_count():          # (device_count() does not initialise the GPU)
        print("bench.py: rank %d wants cuda:%d but this node shows %d GPU(s)" % (rank, local, torch.cuda.device_count()), file=sys.stderr)
        return 2
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if selftest:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        if dist.get_world_size() != args.gpus:                          # the collective's own group, not the environment
            print("bench.py: --gpus %d but the %s group has %d ranks" % (args.gpus, backend, dist.get_world_size()), file=sys.stderr)
            return 2
    if selftest:
        dev = torch.device("cpu")

        def device_sync():
            pass
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        device_sync = torch.cuda.synchronize

    if selftest:
        _lib = config = None
        model = inputs = y = loss_fn = x = None
        train = _SelftestStep(rank)
        if rank == args.selftest_fail_rank:
            print("bench.py: rank %d fails on request (--selftest-fail-rank)" % rank, file=sys.stderr)
            return 7
    else:
        from svnet_amd import _lib, config
        from svnet_amd.train import ForwardStep, TrainStep
        _lib.lib()                                                        # fail loudly if the HIP library is missing
        wl, model, inputs, y, loss_fn = build_workload(args.workload, dev, rank)
        x = inputs[0]
        train = TrainStep(model.train(), inputs, y, loss_fn)
    if args.mode == "train" or selftest:
        work = train
    else:
        work = ForwardStep(model, inputs)
    graph_ok = False
    if not args.no_graph and not selftest:
        try:
            work.capture()
            graph_ok = True
        except Exception as e:                                        # capture is an optimisation, not a requirement
            if rank == 0:
                print("graph capture failed, running eagerly: %r" % (e,), file=sys.stderr)
            work.graph = None
            device_sync()

    def barrier():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    def timed(step_fn, steps, warmup):
        for _ in range(warmup):
            step_fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier()
        mine = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        every = [mine]
        if world > 1:
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        return max(per_rank), per_rank                                # max over the ranks is the job's time

    if args.force_collective and args.mode == "train":
        def run_step():
            work.run(all_reduce=False)
            train.bucket.all_reduce_mean(force=True)                   # the real bucket through RCCL at world size 1
    else:
        run_step = work.run
    elapsed, per_rank = timed(run_step, args.steps, args.warmup)
    collective_us = None
    if collective and args.mode == "train":
        # the collective alone: `steps` back-to-back all-reduces of the bucket after the timed region (HIP events on the device)
        for _ in range(3):
            train.bucket.all_reduce_mean(force=True)
        barrier()
        if selftest:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                train.bucket.all_reduce_mean(force=True)
            collective_us = round((time.perf_counter() - t0) / args.steps * 1e6, 1)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                train.bucket.all_reduce_mean(force=True)
            e1.record()
            torch.cuda.synchronize()
            collective_us = round(e0.elapsed_time(e1) / args.steps * 1e3, 1)
    loss = float(train.loss) if (args.mode == "train" and train.loss is not None) else None
    if loss is not None and not (loss == loss and abs(loss) < 1e6):
        print("bench.py: non-finite loss %r in the timed region" % loss, file=sys.stderr)
        return 3

    # secondary number: forward-only eval throughput next to the headline one (rank-local graph, after the timed region)
    fwd_only = None
    if args.mode == "train" and world == 1 and not selftest:
        fs = ForwardStep(model, inputs)
        try:
            if not args.no_graph:
                fs.capture()
        except Exception:
            fs.graph = None
            torch.cuda.synchronize()
        t_f, _ = timed(fs.run, args.steps, 2)
        fwd_only = {"value": round(B * args.steps / t_f, 2), "unit": "point-clouds/sec", "ms_per_step": round(t_f / args.steps * 1e3, 3),
                    "what": "forward only, eval(), no_grad, same model and batch"}
        model.train()

    stages = per_launch = primary = None
    launches = None
    bucket_mb = train.bucket.flat.numel() * 4 / 1e6
    if rank == 0 and not selftest:
        if args.workload == "dgcnn_cls":
            per_launch, stages = dgcnn_cls_legs(args, model, x, train, work, torch, _lib)
            primary = per_launch if args.mode == "train" else (stages or {}).get("knn_gather_forward")
        else:
            primary, stages = other_workload_legs(args, wl, model, inputs, train, work, torch, _lib)
        if graph_ok:
            launches = graph_kernel_nodes(work)

    extras = (args.workload == "dgcnn_cls" and args.mode == "train" and world == 1 and not args.no_extras and not args.no_graph
              and not selftest)
    others = loop_leg = None
    if rank == 0 and extras:
        try:
            loop_leg = train_loop_leg(model, inputs, y, loss_fn, torch, steps=20)
        except Exception as e:
            loop_leg = {"error": repr(e)[:300]}
        del work, train
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        others = other_workloads_leg(args, dev, torch, _lib)
        train = None

    if rank == 0:
        clouds = B * world * args.steps
        what = "fwd+bwd" if args.mode == "train" else "forward (eval, no_grad)"
        shape = "B=%d N=%d k=%d" % (B, N, k)
        out = {
            "metric": "point-clouds/sec %s, %s %s" % (what, wl["model"].replace("sv_dgcnn_pseg", "sv_dgcnn_partseg"), shape),
            "value": round(clouds / elapsed, 2), "unit": "point-clouds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (ternary bit-planes in the binarized layers)" if wl["binary"] else "f32",
            "data": "synthetic",
            "config": {"workload": "%s %s, B=%d per GPU, N=%d, k=%d"
                                   % (wl["name"], "fwd+loss+bwd" if args.mode == "train" else "forward only (eval)", B, N, k),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "collective": ("%s all-reduce(avg) of one %.2f MB gradient bucket per step, world size %d"
                                      % ("RCCL" if backend == "nccl" else backend, bucket_mb, dist.get_world_size())) if collective else "none (1 rank)",
                       "rccl_world": dist.get_world_size() if collective else 1,      # the collective group's own size (1 = no group)
                       "collective_us": collective_us,
                       "launch": "hipGraph replay" if graph_ok else "eager",
                       "graph_nodes": launches},
            "per_rank_ms_per_step": [round(t / args.steps * 1e3, 3) for t in per_rank],
            "roofline": primary,
        }
        if selftest:
            # NOT a measurement of the product: the line only shows that the rank plumbing ran end to end
            out.update({"metric": "SELFTEST (gloo/CPU stand-in step, not the product)", "data": "selftest", "dtype": "f32",
                        "selftest": True})
            out["config"]["workload"] = "rank-plumbing self-test: stand-in torch CPU step, B=%d per rank" % B
        else:
            # which binarized-linear kernel serves the dense layers with >= 1024 rows (both give identical integer counts)
            out["config"]["binlinear"] = (("i8_mfma (>= 1024 rows) + xnor (head)" if config.BINLINEAR_MFMA else "xnor")
                                          if wl["binary"] else "none (fp)")
        if args.workload == "dgcnn_cls" and args.mode == "train" and not selftest:
            out["metric"] = "point-clouds/sec fwd+bwd, sv_dgcnn_cls B=32 N=1024 k=20"           # BASELINE.json's wording
        if loss is not None:
            out["loss"] = round(loss, 6)
        if stages:
            out["roofline_stages"] = stages
        if fwd_only:
            out["forward_only"] = fwd_only
        if loop_leg:
            out["train_loop"] = loop_leg
            if "train_loop_ms_per_step" in loop_leg:
                out["train_loop_ms_per_step"] = loop_leg["train_loop_ms_per_step"]
        if others:
            out["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline and not selftest:
            # the headline workload at its own B = 32; the secondary workloads on a bounded sample (--workload runs only)
            out["cpu_baseline"] = cpu_baseline(wl, sample_b=None if args.workload == "dgcnn_cls" else (8 if wl["model"] != "sv_dgcnn_pseg" else 2))
        print(json.dumps(out))
        sys.stdout.flush()
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    sys.exit(rank_main(args))


if __name__ == "__main__":
    main()
