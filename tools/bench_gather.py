"""Times edgeblock_bwd_gather_kernel alone on a real kNN graph (diagnostic).  With arguments: -D flags for variant builds of
edgeblock_post.hip (built on the box into gpurun_out/gather_var/)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
if len(sys.argv) > 1 and sys.argv[1] != "run":
    out = os.path.join(root, "gpurun_out", "gather_var"); os.makedirs(out, exist_ok=True)
    src = os.path.join(root, "svnet_amd", "csrc")
    o = os.path.join(out, "edgeblock_post.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-slp-vectorize"] + sys.argv[1:] +
                          ["-c", os.path.join(src, "edgeblock_post.hip"), "-o", o])
    objs = [os.path.join(src, "_build", f) for f in os.listdir(os.path.join(src, "_build")) if f.endswith(".o") and f != "edgeblock_post.o"]
    lib = os.path.join(out, "libsvnet_hip.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o] + objs)
    sys.exit(subprocess.call([sys.executable, __file__, "run"], env=dict(os.environ, SVNET_HIP_LIB=lib, SVNET_VARIANT=" ".join(sys.argv[1:]))))
import torch
from svnet_amd import _lib
if os.environ.get("SVNET_HIP_LIB"): _lib.LIB_PATH = os.environ["SVNET_HIP_LIB"]
from svnet_amd import _ops
from svnet_amd._ops import _p, _stream, call
B, N, k = 32, 1024, 20
P, E = B * N, B * N * k
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, int(os.environ.get("SVNET_KNN_DIM", "3")), N, device=dev)
idx = _ops.knn(x, k).reshape(P, k).contiguous()
rng = torch.empty(2 * P, dtype=torch.int32, device=dev); red = torch.empty(E, dtype=torch.int32, device=dev); src = torch.empty(E, dtype=torch.int32, device=dev)
call("svnet_knn_reverse_i32", _p(idx), B, N, k, _p(rng), _p(red), _p(src), _stream())
deg = (rng.view(P, 2)[:, 1] - rng.view(P, 2)[:, 0]).float()
print("in-degree: mean %.1f max %d  p99 %d" % (deg.mean().item(), int(deg.max().item()), int(deg.kthvalue(int(P * 0.99)).values.item())))
for (Cs, Cv, Os, Ov) in [(64, 21, 128, 42), (32, 10, 64, 21)]:
    R = _lib.lib().svnet_edgeblock_msg_stride(Cs, Cv, Ov)
    f = lambda *s: torch.randn(*s, device=dev)
    msg, ut, ub, ge = f(E, R), f(P, 3, 2 * Ov), f(P, 3, Ov), f(P, 3, Ov)
    coef, bcoef = f(4 * Os + 2 * Ov), f(8 * Os + 2 * Ov + 4)
    dvc, dzc = f(P, 3, Ov), f(P, 9)
    Rp = (2 * Ov + 6 + 3) // 4 * 4
    acat, ds, dv = f(3 * P, Rp), f(P, Cs), f(P, 3, Cv)
    dbp, db1 = f(512), f(2 * Cs + 6 * Cv)
    def run():
        call("svnet_edgeblock_bwd_gather_f32", _p(msg), _p(rng), _p(red), _p(src), _p(ut), _p(ub), _p(ge), _p(coef), _p(bcoef), Os, _p(dvc), _p(dzc),
             P, N, Cs, Cv, Ov, _p(acat), Rp, _p(ds), _p(dv), _p(dbp), _p(db1), _stream())
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    print("gather Cs=%d Cv=%d R=%d: %.1f us  [%s]" % (Cs, Cv, R, a.elapsed_time(b) / 20 * 1e3, os.environ.get("SVNET_VARIANT", "product build")), flush=True)
