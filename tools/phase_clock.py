"""Per-phase cycle counts of edgeblock_bwd_kernel (diagnostic): builds a -DSVNET_PHASE_CLOCK copy of the library into
gpurun_out/, runs one fused backward per layer shape and prints the mean cycles each workgroup spent per phase."""
import os, subprocess, sys, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1 and os.path.exists(os.path.join(root, "_ab", "libphase.so")):
    # built beforehand: make -C svnet_amd/csrc BUILD=_build_phase OUT=../../_ab/libphase.so EXTRA=-DSVNET_PHASE_CLOCK
    env = dict(os.environ, SVNET_HIP_LIB=os.path.join(root, "_ab", "libphase.so"))
    sys.exit(subprocess.call([sys.executable, __file__, "run"], env=env))
if len(sys.argv) == 1:
    out = os.path.join(root, "gpurun_out", "phase_lib")
    os.makedirs(out, exist_ok=True)
    src = os.path.join(root, "svnet_amd", "csrc")
    objs = []
    for f in sorted(os.listdir(src)):
        if f.endswith(".hip") or f == "error.cpp":
            o = os.path.join(out, f + ".o")
            flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DSVNET_PHASE_CLOCK"]
            if f == "knn.hip":
                flags.append("-ffp-contract=off")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(src, f), "-o", o])
            objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libsvnet_hip.so")] + objs)
    env = dict(os.environ, SVNET_HIP_LIB=os.path.join(out, "libsvnet_hip.so"))
    sys.exit(subprocess.call([sys.executable, __file__, "run"], env=env))
import torch, contextlib, io
sys.path.insert(0, root)
from svnet_amd import _lib, _ops, config
_lib.LIB_PATH = os.environ["SVNET_HIP_LIB"]
from svnet_amd.models.sv_layers import SVBlock
from svnet_amd.models.utils.sv_util import get_graph_feature_sv, svpool
config.FUSE_EDGE_BLOCKS = True
names = {11: "A_issue", 12: "A_wait", 10: "C_setup", 0: "A_compute", 1: "transpose", 5: "B_mfma", 6: "B_barrier", 2: "B_epilogue", 7: "C_pass1", 9: "C_pass23", 4: "C_pass4", 3: "C1_old"}
order = [11, 12, 0, 1, 5, 6, 2, 10, 7, 9, 4]
for (Cs, Cv, Os, Ov) in [(32, 10, 32, 10), (32, 10, 64, 21), (64, 21, 128, 42)]:
    with contextlib.redirect_stdout(io.StringIO()):
        blk = SVBlock((2 * Cs, 2 * Cv), (Os, Ov), binary=True).cuda().train()
    s = torch.randn(32, 1024, Cs, device="cuda", requires_grad=True)
    v = torch.randn(32, 1024, 3, Cv, device="cuda", requires_grad=True)
    timer = _lib.KernelTimer("svnet_edgeblock_bwd_f32", lambda a: a[0]._obj.parts == 2)
    _lib.TIMERS[:] = [timer]
    for it in range(3):
        _ops.DEBUG_BUFFER = torch.zeros(8 + 256 * 16, dtype=torch.int64, device="cuda")
        so, vo = svpool(blk(get_graph_feature_sv((s, v), k=20)))
        (so.sum() + vo.sum()).backward()
        torch.cuda.synchronize()
    _lib.TIMERS[:] = []
    kms = min(timer.elapsed_ms())
    cyc = [0] * 8 + _ops.DEBUG_BUFFER[8:].view(256, 16).sum(0).cpu().tolist()
    tiles = 32 * 1024 * 20 // 32
    per = {i: cyc[8 + i] / tiles for i in order}
    tot = sum(per.values())
    life, wall = cyc[8 + 14] / tiles, cyc[8 + 15] / tiles
    kms_ = kms
    print("kernel %.1f us (this instrumented build) -> %.1f workgroups alive per CU; " % (kms * 1e3, tiles * wall / 100.0 / (256 * kms * 1e3)) +
          "Os=%d Cs=%d Cv=%d: workgroup lifetime %d clock64 ticks = %d ticks of the 100 MHz clock (%.2f us; 1 clock64 tick = %.3f ns); marks sum to %d: " % (
        Os, Cs, Cv, life, wall, wall / 100.0, wall * 10.0 / max(life, 1), tot)
          + " ".join("%s=%d (%.0f%%)" % (names[i], per[i], 100.0 * per[i] / tot) for i in order), flush=True)
