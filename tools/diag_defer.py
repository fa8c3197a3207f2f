import sys, os, argparse, contextlib, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import svnet_amd.models as M
from svnet_amd import synth, config, _ops
from svnet_amd.train import TrainStep
for kv in sys.argv[1:]:
    k_, v_ = kv.split('='); setattr(config, k_, bool(int(v_)))
dev = torch.device("cuda:0")
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = M.SV_DGCNN_CLS(argparse.Namespace(k=16, binary=True), 40).to(dev).train()
B, N = 4, 512
x = torch.from_numpy(synth.cloud_batch(1234, 0, 0, B, N)).to(dev)
y = torch.from_numpy(synth.class_labels(1234, 0, 0, B)).to(dev)
step = TrainStep(model, (x,), y)
res = {}
for flag in (False, True):
    config.DEFER_WGRAD = flag
    loss = float(step.fwd_bwd()); torch.cuda.synchronize()
    res[flag] = step.bucket.flat.clone()
config.DEFER_WGRAD = True
step.capture()
for r in range(3):
    step.run(all_reduce=False); torch.cuda.synchronize()
    res["replay%d" % r] = step.bucket.flat.clone()
scale = float(res[False].abs().max())
for key in (True, "replay0", "replay1", "replay2"):
    off = 0
    bad = []
    for n, p in model.named_parameters():
        k = p.numel()
        e = float((res[key][off:off+k] - res[False][off:off+k]).abs().max()) / scale
        if e > 5e-5: bad.append((n, round(e, 4)))
        off += k
    print(key, bad[:12])
