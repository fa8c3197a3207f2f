"""GPU test (-m gpu): a caller written against the PUBLIC module namespace only, in the reference's call order.

INTEGRATION.md §1 promises that a model file written for the reference (`from .sv_layers import *`, `from .utils.sv_util import *`)
runs on the drop-in after swapping those two imports.  svnet_amd's own SV_DGCNN_CLS does not prove that: its forward is restructured
(CatSink, pooled parts, the BatchNorm inside the global pooling, fused classifier heads).  The class below is what a user's model file
looks like - every call the reference's forward makes (models/sv_dgcnn_cls.py:46-82), in that order, on the names the two modules
export and nothing else: get_graph_feature -> init_scalar -> conv1 -> svpool -> 3 x (get_graph_feature_sv -> conv -> svpool) -> svcat
-> conv5 -> svfuse -> transpose -> adaptive_max/avg_pool1d -> cat -> leaky_relu(bn(linear)) x 2 -> linear3.  The lazy handles the
drop-in's functions return (XyzEdges, LazyInitScalar, EdgeFeatures, Pending*Block) must behave as the tuples / tensors they stand for.
Same state_dict in both, same input: eval logits to 1e-5, train-step loss and every parameter gradient to 1e-4, with the fused edge
layers on and off (both models under the same switch)."""
import argparse
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import params as oparams
from tests.common import case_errors
from tests.golden import cases as C

pytestmark = pytest.mark.gpu


def _user_model_class():
    # what the top of a user's model file looks like after the two-line swap
    from svnet_amd.models.sv_layers import Linear, SVBlock, SVFuse, Vector2Scalar, nn, F, torch as T
    from svnet_amd.models.utils.sv_util import get_graph_feature, get_graph_feature_sv, svcat, svpool

    class UserDGCNN(nn.Module):
        def __init__(self, args, num_class=40):
            super().__init__()
            self.k, b = args.k, args.binary
            self.init_scalar = Vector2Scalar(2, 3)
            self.conv1 = SVBlock((6, 2), (32, 10))
            self.conv2 = SVBlock((64, 20), (32, 10), b)
            self.conv3 = SVBlock((64, 20), (64, 21), b)
            self.conv4 = SVBlock((128, 42), (128, 42), b)
            self.conv5 = SVBlock((256, 83), (512, 170), b)
            self.svfuse = SVFuse(170, 3, b)
            self.linear1 = Linear(2044, 512, bias=False, bw=b, ba=b)
            self.bn1 = nn.BatchNorm1d(512)
            self.dp1 = nn.Dropout(p=0.0)
            self.linear2 = Linear(512, 256, bias=False, bw=b, ba=b)
            self.bn2 = nn.BatchNorm1d(256)
            self.dp2 = nn.Dropout(p=0.0)
            self.linear3 = nn.Linear(256, num_class)

        def forward(self, cloud):
            nb = cloud.size(0)
            vec = get_graph_feature(cloud.unsqueeze(1), k=self.k)
            sca = self.init_scalar(vec)
            feats = [svpool(self.conv1((sca, vec)))]
            for conv in (self.conv2, self.conv3, self.conv4):
                edges = get_graph_feature_sv(feats[-1], k=self.k)
                feats.append(svpool(conv(edges)))
            fused = self.svfuse(self.conv5(svcat(feats)))
            fused = fused.transpose(-1, -2).contiguous()
            hi = F.adaptive_max_pool1d(fused, 1).view(nb, -1)
            av = F.adaptive_avg_pool1d(fused, 1).view(nb, -1)
            h = T.cat((hi, av), 1)
            h = self.dp1(F.leaky_relu(self.bn1(self.linear1(h)), negative_slope=0.2))
            h = self.dp2(F.leaky_relu(self.bn2(self.linear2(h)), negative_slope=0.2))
            return self.linear3(h)

    return UserDGCNN


def _pair(binary, k, dev):
    import svnet_amd.models as M
    state = oparams.synthetic_params("sv_dgcnn_cls", binary=binary, seed=C.SEED)
    args = argparse.Namespace(k=k, binary=binary)
    with contextlib.redirect_stdout(io.StringIO()):
        ours, user = M.SV_DGCNN_CLS(args, 40), _user_model_class()(args, 40)
    for m in ours.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    ours.load_state_dict(state, strict=True)
    user.load_state_dict(state, strict=True)            # the same keys: a reference checkpoint loads into a user's file unchanged
    return ours.to(dev), user.to(dev)


def _step(model, x, y):
    from svnet_amd.train import cal_loss
    for p in model.parameters():
        p.grad = None
    logits = model(x)
    loss = cal_loss(logits, y)
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach().cpu().numpy(), float(loss), {"d:" + n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}


@pytest.mark.parametrize("fuse", [True, False], ids=["fused_edges", "layerwise_edges"])
@pytest.mark.parametrize("binary", [True, False], ids=["binary", "fp"])
def test_a_reference_order_caller_runs_unchanged_on_the_drop_in(binary, fuse, hip_device):
    from svnet_amd import config
    B, N, k = 4, 128, 8
    x, _, y = C.model_inputs("reforder_%d" % binary, "sv_dgcnn_cls", B, N)
    x, y = x.to(hip_device), y.to(hip_device)
    old = config.FUSE_EDGE_BLOCKS
    config.FUSE_EDGE_BLOCKS = fuse
    try:
        ours, user = _pair(binary, k, hip_device)
        with torch.no_grad():
            lo, lu = ours.eval()(x).cpu().numpy(), user.eval()(x).cpu().numpy()
        scale = float(np.abs(lo).max())
        assert np.isfinite(lu).all() and float(np.abs(lu - lo).max()) <= 1e-5 * scale, float(np.abs(lu - lo).max()) / scale
        lo, loss_o, g_o = _step(ours.train(), x, y)
        lu, loss_u, g_u = _step(user.train(), x, y)
    finally:
        config.FUSE_EDGE_BLOCKS = old
    assert float(np.abs(lu - lo).max()) <= 1e-5 * float(np.abs(lo).max())
    assert abs(loss_u - loss_o) <= 1e-5 * max(1.0, abs(loss_o))
    errs = case_errors(g_u, g_o)
    bad = sorted(((e, n) for n, e in errs.items() if e > 1e-4), reverse=True)
    assert not bad, bad[:5]
    # BatchNorm buffers went the same way in both (the user's file calls nn.BatchNorm1d itself for the head, ours the fused head kernels)
    for (n, a), (_, b_) in zip(ours.state_dict().items(), user.state_dict().items()):
        if "running_" in n:
            a, b_ = a.cpu().numpy(), b_.cpu().numpy()
            assert float(np.abs(a - b_).max()) <= 1e-5 * max(float(np.abs(a).max()), 1e-3), n
        if n.endswith("num_batches_tracked"):
            assert int(a) == int(b_) == 1, n
