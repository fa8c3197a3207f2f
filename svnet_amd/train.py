"""Minimal train-step pieces for the SV classifiers on MI355X: the loss of the reference's training loop
(utils.py:33-50 cal_loss; used at main_cls_dgcnn.py:182) as a HIP kernel, and a fwd+bwd step helper."""
from . import _ops


def cal_loss(pred, target, smoothing=True):
    """Label-smoothed cross entropy (eps = 0.2), mean over rows.  pred [R,C] float32, target [R] int64."""
    return _ops.SmoothCE.apply(pred, target.contiguous().view(-1), 0.2 if smoothing else 0.0)


def forward_backward(model, x, y, bucket=None):
    """One fwd + cal_loss + bwd pass; gradients accumulate into the (pre-zeroed) .grad tensors."""
    logits = model(x)
    loss = cal_loss(logits, y)
    loss.backward()
    return loss
