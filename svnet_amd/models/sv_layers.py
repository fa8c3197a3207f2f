"""Scalar/vector layer library of SVNet on MI355X.

Drop-in for the reference module `models/sv_layers.py`: same class names, constructor signatures,
sub-module / parameter names (state_dict compatible) and train/eval behaviour:
  Linear :20-53, Conv1d :55-78, VectorBN :81-102, Vector2Scalar :104-129, VectorReLU :131-149,
  SVBlock :151-196, SVFuse :198-220, SV_STNkd :222-244.
Every forward runs hand-written gfx950 kernels through svnet_amd._ops; tensors must be on a HIP
device (no CPU fallback).
"""
import os
import sys
import copy
import math
import numpy as np

import torch
import torch.nn as nn
import torch.nn.init as init
import torch.nn.functional as F

from .. import _ops
from .. import config
from .utils.sv_util import svpool, EdgeFeatures, XyzEdges

EPS = 1e-6

__all__ = ["EPS", "linear_bn_act", "Linear", "Conv1d", "VectorBN", "Vector2Scalar", "VectorReLU", "SVBlock", "SVFuse", "SV_STNkd",
           "svpool", "torch", "nn", "F", "np", "math", "os", "sys", "copy", "init"]

_ACT_NONE, _ACT_LEAKY, _ACT_RELU = 0, 1, 2


def _bn_momentum(bn):
    """nn.BatchNorm1d's momentum (None = cumulative moving average, which the kernels do not implement)."""
    if bn.momentum is None:
        raise NotImplementedError("svnet_amd: BatchNorm1d(momentum=None) (cumulative average) is not supported")
    return float(bn.momentum)


def batch_norm_act(bn, x, act=_ACT_NONE, slope=0.2):
    """nn.BatchNorm1d `bn` (+ activation) over the rows of x [..., C] in one fused pass."""
    training = bn.training or bn.running_mean is None
    nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None    # += 1 inside the finalize kernel
    return _ops.BNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, act, slope, nbt, bn.eps, _bn_momentum(bn))


def linear_bn_act(lin, bn, x, act=_ACT_NONE, slope=0.2):
    """act(bn(lin(x))) - the classifier heads' layer pattern (sv_dgcnn_cls.py:76-78, sv_pointnet_cls.py:59-60).  A binarized
    layer over a few rows (x [M <= 64, K]) runs as one fused pass forward and two passes backward (_ops.BinLinearBNAct,
    csrc/head.hip); anything else is the layer-wise chain."""
    if (config.FUSE_HEAD and isinstance(lin, Linear) and lin.bw and lin.ba and lin.bias is None and x.dim() == 2 and x.is_cuda
            and _ops.BinLinearBNAct.supported(x.shape[0], lin.in_features, lin.out_features) and bn.affine
            and lin.training == bn.training):
        training = bn.training or bn.running_mean is None
        grads = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in lin.parameters())
                                             or any(p.requires_grad for p in bn.parameters()))
        if training or not grads:                      # (eval-mode gradients - bare sign(): zero - stay on the layer-wise ops)
            nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
            return _ops.BinLinearBNAct.apply(x, lin.weight, lin.beta, lin.scale, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                             training, act, slope, nbt, bn.eps, _bn_momentum(bn), grads)
    return batch_norm_act(bn, lin(x), act, slope)


class Linear(nn.Linear):
    """nn.Linear whose weights (bw) and/or activations (ba) are binarized with sign():
    y = (sign(x + beta) . sign(W)^T) * scale (+ bias).  sign(0) = 0, so operands are ternary.
    train(): backward is the clamp(-1.2, 1.2) straight-through estimator; eval(): bare sign(), whose
    gradient is zero (only `scale` and `bias` receive one), exactly as in the reference."""

    def __init__(self, in_channels, out_channels, bias, bw=False, ba=False):
        super(Linear, self).__init__(in_channels, out_channels, bias)
        self.bw, self.ba = bw, ba
        if ba:
            self.beta = nn.Parameter(torch.zeros(1, in_channels))
        if bw:
            self.scale = nn.Parameter(torch.ones(1, out_channels) / math.sqrt(in_channels))

    def forward(self, x, vstats=False):
        """vstats: a hint from SVBlock - x holds [..., 3, K] vectors and a VectorBN takes the output next (see _ops.BwLinear)."""
        if self.bw and self.ba:
            return _ops.BinLinear.apply(x, self.weight, self.beta, self.scale, self.bias, self.training)
        if self.bw:
            y = _ops.BwLinear.apply(x, self.weight, self.scale, self.training, bool(vstats) and self.bias is None)
            return y if self.bias is None else y + self.bias
        if self.ba:
            raise AttributeError("'Linear' object has no attribute 'scale'")  # same failure as the reference (:49)
        return _ops.FpLinear.apply(x, self.weight, self.bias)


class Conv1d(nn.Conv1d):
    """1x1 convolution on channel-first [B,C,N] rows, optionally binarized like Linear(bw, ba)."""

    def __init__(self, in_channels, out_channels, binary=False):
        super(Conv1d, self).__init__(in_channels, out_channels, 1, bias=False)
        print('Conv1d: ', in_channels, out_channels)
        self.binary = binary
        if binary:
            self.beta = nn.Parameter(torch.zeros(1, in_channels, 1))
            self.scale = nn.Parameter(torch.ones(1, out_channels, 1) / math.sqrt(in_channels))

    def forward_rows(self, rows):
        """The same layer on channel-LAST rows [..., C] -> [..., O] (what the kernels work on; callers that keep their
        activations channel-last skip the two transposed copies of forward())."""
        if self.binary:   # (the raw [O,C,1] / [1,C,1] / [1,O,1] parameters: the op views them, and caches the packed weight by parameter)
            return _ops.BinLinear.apply(rows, self.weight, self.beta, self.scale, None, self.training)
        return _ops.FpLinear.apply(rows, self.weight.view(self.out_channels, self.in_channels), None)

    def forward(self, x):
        return self.forward_rows(x.transpose(1, 2)).transpose(1, 2).contiguous()     # [B,C,N] -> rows view -> [B,O,N]


class VectorBN(nn.Module):
    """Batch-normalise the length of each vector channel: v * BN(|v| + EPS) / (|v| + EPS)."""

    def __init__(self, dim):
        super(VectorBN, self).__init__()
        self.bn = nn.BatchNorm1d(dim)

    def forward(self, v, gate=None):
        '''
        shape of v: B, N_points, [k,] 3, dim ; gate (optional): B, dim
        '''
        bn = self.bn
        training = bn.training or bn.running_mean is None
        nbt = bn.num_batches_tracked if bn.training else None                                 # += 1 inside the finalize kernel
        rows = v.numel() // (3 * v.shape[-1])
        rows_per_batch = max(rows // v.shape[0], 1)
        return _ops.VBN.apply(v, bn.weight, bn.bias, bn.running_mean, bn.running_var, gate, rows_per_batch, training, nbt, bn.eps,
                              _bn_momentum(bn))


class Vector2Scalar(nn.Module):
    """Rotation-invariant scalars s[d*multi+j] = sum_i v[i,d] * (v W^T)[i,j]."""

    def __init__(self, v_dim, multi, binary=False, trans_back=False):
        super(Vector2Scalar, self).__init__()
        self.trans_back = trans_back
        self.linear = Linear(v_dim, multi, bias=False, bw=binary)

    def forward(self, v):
        '''
        shape of v: B, N_points, [k,] 3, dim
        '''
        assert v.ndim in [3, 4, 5], 'dim of v should be in [4, 5], got {}'.format(v.ndim)
        if isinstance(v, XyzEdges):
            if not self.linear.bw and not self.trans_back and tuple(self.linear.weight.shape) == (3, v.nc):
                return LazyInitScalar(self, v)
            v = v.materialize()
        s, z = _ops.V2S.apply(v, self.linear.weight, self.linear.scale if self.linear.bw else None, self.training)
        return (s, z) if self.trans_back else s


class VectorReLU(nn.Module):
    """Keeps the vectors whose norm exceeds the (n/10)-th smallest norm of their sample.  Never instantiated
    by any SV model (API surface only): a thin composition of torch ops on the device, no dedicated kernel."""

    def __init__(self):
        super(VectorReLU, self).__init__()
        self.div = 10

    def forward(self, x):
        shape_x = x.shape
        rows = x.reshape(shape_x[0], -1, 3, shape_x[-1])
        kth = rows.shape[1] // self.div
        length = torch.linalg.vector_norm(rows, dim=2, keepdim=True).detach()
        threshold = torch.kthvalue(length, kth, dim=1, keepdim=True)[0]
        return torch.where(length > threshold, rows, torch.zeros_like(rows)).view(shape_x)


class LazyInitScalar:
    """Vector2Scalar(2,3) / (3,3) applied to lazy XyzEdges (the `init_scalar` of the DGCNN / PointNet models): stands for its [B,N,k,6 or 9]
    output; computed on demand, or folded into the fused first-layer kernel by SVBlock + svpool."""

    def __init__(self, v2s, edges):
        self.v2s, self.edges, self._s = v2s, edges, None

    def materialize(self):
        if self._s is None:
            self._s = self.v2s(self.edges.materialize())
        return self._s

    def __getattr__(self, name):
        return getattr(self.materialize(), name)


class PendingXyzBlock:
    """SVBlock (fp) applied to (LazyInitScalar, XyzEdges): svpool(max over k) runs the fused first-layer kernel."""

    def __init__(self, block, s_lazy, edges):
        self.block, self.s_lazy, self.edges, self._out = block, s_lazy, edges, None

    def pooled(self, dim, keepdim, spool):
        if dim != 2 or spool != 'max' or self._out is not None:
            return None
        b, e = self.block, self.edges
        bn1, bn2 = b.bn1, b.bn2.bn
        s, v, s_view, v_view = _ops.XyzBlock.apply(
            e.pts, e.idx, e.k, b.training, self.s_lazy.v2s.linear.weight, b.v2s.linear.weight, b.linear1.weight, bn1.weight,
            bn1.bias, bn1.running_mean, bn1.running_var, b.linear2.weight, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var,
            b.gate[0].weight, b.gate[2].weight, bn1.num_batches_tracked, bn2.num_batches_tracked)
        if s_view is not None:
            _ops._SINK.tracked(s_view, v_view)          # (its slice of a pyramid's concatenation: _ops.CatSink)
        return (s.unsqueeze(2), v.unsqueeze(2)) if keepdim else (s, v)

    def materialize(self):
        if self._out is None:
            self._out = self.block._forward_rows((self.s_lazy.materialize(), self.edges.materialize()))
        return self._out

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, i):
        return self.materialize()[i]

    def __len__(self):
        return 2


class PendingEdgeBlock:
    """SVBlock applied to lazy EdgeFeatures: a stand-in for the tuple (s, v) of per-edge outputs.  svpool(max over
    the k neighbours) consumes it with the fused kernel; any other use materialises the edges and runs the
    layer-by-layer path, so behaviour is unchanged."""

    def __init__(self, block, edges):
        self.block, self.edges, self._out = block, edges, None

    def pooled(self, dim, keepdim, spool):
        if dim != 2 or spool != 'max' or self._out is not None:
            return None
        b, e = self.block, self.edges
        bn1, bn2 = b.bn1, b.bn2.bn
        training = b.training
        if not training and torch.is_grad_enabled() and (e.s.requires_grad or any(p.requires_grad for p in b.parameters())):
            return None            # eval-mode gradients (bare sign(): zero STE gradient) take the layer-wise path
        s, v, s_view, v_view = _ops.EdgeBlock.apply(
            e.s, e.v, e, e.k, training, b.v2s.linear.weight, b.v2s.linear.scale, b.linear1.weight, b.linear1.beta,
            b.linear1.scale, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, b.linear2.weight, b.linear2.scale,
            bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, b.gate[0].weight, b.gate[2].weight,
            bn1.num_batches_tracked, bn2.num_batches_tracked)
        if s_view is not None:
            _ops._SINK.tracked(s_view, v_view)
        return (s.unsqueeze(2), v.unsqueeze(2)) if keepdim else (s, v)

    def materialize(self):
        if self._out is None:
            self._out = self.block._forward_rows(self.edges.materialize())
        return self._out

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, i):
        return self.materialize()[i]

    def __len__(self):
        return 2


class SVBlock(nn.Module):
    """One scalar/vector layer: gate from the mean scalar, invariant scalars from the vectors, (binarized)
    scalar linear + BN + LeakyReLU, (sign-weight) vector linear + VectorBN, vectors scaled by the gate."""

    def __init__(self, in_dims, out_dims, binary=False):
        super(SVBlock, self).__init__()
        print('SVBlock: ', in_dims, out_dims)

        self.gate = nn.Sequential(
            nn.Linear(in_dims[0], out_dims[1] // 2, bias=False),
            nn.ReLU(inplace=True),
            nn.Linear(out_dims[1] // 2, out_dims[1], bias=False),
            nn.Sigmoid(),
        )
        self.v2s = Vector2Scalar(in_dims[1], 3, binary=binary)

        self.linear1 = Linear(in_dims[0] + in_dims[1] * 3, out_dims[0], bias=False, bw=binary, ba=binary)
        self.bn1 = nn.BatchNorm1d(out_dims[0])
        self.relu = nn.LeakyReLU(negative_slope=0.2)

        self.linear2 = Linear(in_dims[1], out_dims[1], bias=False, bw=binary)
        self.bn2 = VectorBN(out_dims[1])

    def _gate(self, s):
        small = self.gate[0].out_features <= 256 and self.gate[2].out_features <= 256
        s3 = s.reshape(s.shape[0], -1, s.shape[-1])
        if small and config.GATE_MEAN_INSIDE and _ops.GateMLPRows.supported(s3):
            return _ops.GateMLPRows.apply(s3, self.gate[0].weight, self.gate[2].weight)   # (the mean over the rows inside the MLP's launch)
        pooled = _ops.Pool.apply(s3, 1, 1)                                               # mean over all rows of a cloud
        if small:
            return _ops.GateMLP.apply(pooled, self.gate[0].weight, self.gate[2].weight)  # -> [B, Cv_out]
        h = _ops.Act.apply(_ops.FpLinear.apply(pooled, self.gate[0].weight, None), 1)   # ReLU
        _ops._tap_act(self.gate[0].weight, 2, h)
        return _ops.Act.apply(_ops.FpLinear.apply(h, self.gate[2].weight, None), 2)     # Sigmoid -> [B, Cv_out]

    def _can_fuse(self, edges):
        lin1, lin2 = self.linear1, self.linear2
        if not (lin1.bw and lin1.ba and lin2.bw and self.v2s.linear.bw) or edges.idx_is_global:
            return False
        Cs, Cv = edges.s.shape[-1], edges.v.shape[-1]
        return (Cs <= 64 and 2 * Cv <= 64 and lin1.out_features <= 128 and lin2.out_features <= 64 and 2 <= edges.k <= 64
                and edges.s.shape[1] <= 8192 and lin1.out_features in (8, 16, 32, 64, 128) and lin1.in_features == 2 * Cs + 6 * Cv
                and self._default_bn())

    def _default_bn(self):
        """The fused kernels take nn.BatchNorm1d's default eps / momentum (what every SV model uses); anything else runs layer-wise."""
        return all(bn.track_running_stats and bn.eps == _ops.BN_EPS and bn.momentum == _ops.BN_MOMENTUM for bn in (self.bn1, self.bn2.bn))

    def forward(self, x):
        '''
        shape of s: B, N_points, [k,] s_dim
        shape of v: B, N_points, [k,] 3, v_dim
        '''
        if isinstance(x, EdgeFeatures) and self._can_fuse(x):
            return PendingEdgeBlock(self, x)
        if isinstance(x, (tuple, list)) and len(x) == 2 and (isinstance(x[0], LazyInitScalar) or isinstance(x[1], XyzEdges)):
            s, v = x
            lin1, lin2 = self.linear1, self.linear2
            if (isinstance(s, LazyInitScalar) and s.edges is v and not (lin1.bw or lin1.ba or lin2.bw or self.v2s.linear.bw)
                    and lin1.in_features == 6 * v.nc and lin2.in_features == v.nc and lin1.out_features <= 64 and lin2.out_features <= 64
                    and v.k <= 64 and self._default_bn()
                    and (self.training or not torch.is_grad_enabled()
                         or not any(p.requires_grad for p in self.parameters()))):
                return PendingXyzBlock(self, s, v)
            x = (s.materialize() if isinstance(s, LazyInitScalar) else s, v.materialize() if isinstance(v, XyzEdges) else v)
        return self._forward_rows(x)

    def _v2s_cat_fusable(self, s, v):
        lin = self.v2s.linear
        return (config.FUSE_V2S_CAT and torch.is_tensor(s) and torch.is_tensor(v) and s.is_cuda and not self.v2s.trans_back
                and lin.weight.shape[0] == 3 and v.shape[-1] <= 768 and s.shape[:-1] == v.shape[:-2])

    def _cat_s_v2s(self, s, v):
        """cat[s, Vector2Scalar(v)] (sv_layers.py:187-188); on the GPU the Vector2Scalar kernel writes the concatenation in place."""
        lin = self.v2s.linear
        if self._v2s_cat_fusable(s, v):
            return _ops.V2SCat.apply(s, v, lin.weight, lin.scale if lin.bw else None, self.training)
        return torch.cat([s, self.v2s(v)], dim=-1)

    def _cat_and_gate(self, s, v):
        """(cat[s, Vector2Scalar(v)], gate) with s consumed ONCE: the concatenation op also returns the per-cloud mean of s the gate MLP
        starts from (sv_layers.py:179), so that its backward writes dL/ds once (the cat gradient's s columns + the mean's broadcast)
        instead of autograd adding a broadcast tensor and a strided slice.  None when the pieces are not fusable."""
        lin = self.v2s.linear
        # (wide s only: the fused backward pass gives a thread a column - at the PointNet callers' 32 .. 64 columns most of a workgroup
        #  idles and the step was 4 % slower, 6.19 against 5.92 ms; at conv5 of the DGCNN callers, 256 columns, it is the faster form)
        if not (self._v2s_cat_fusable(s, v) and s.dim() >= 3 and s.shape[-1] >= 128 and self.gate[0].out_features <= 256
                and self.gate[2].out_features <= 256):
            return None
        if config.FUSE_CAT_MEAN:      # the gate MLP inside the op, fed by column sums the concatenation kernel forms from its own copy of s
            return _ops.V2SCat.apply(s, v, lin.weight, lin.scale if lin.bw else None, self.training, s.shape[0], self.gate[0].weight,
                                     self.gate[2].weight)
        cat, s_mean = _ops.V2SCat.apply(s, v, lin.weight, lin.scale if lin.bw else None, self.training, s.shape[0])
        return cat, _ops.GateMLP.apply(s_mean, self.gate[0].weight, self.gate[2].weight)

    def forward_prebn(self, x):
        """(y, v_out) with y = linear1(cat[s, v2s(v)]) BEFORE bn1 + LeakyReLU: for a consumer that folds them into what it
        does next (the classifier's global pooling, _ops.GlobalMaxMeanPoolBN).  Rows path only."""
        return self._forward_rows(x, prebn=True)

    def forward_pretail(self, x):
        """(y, v_lin, gate): both products of the block BEFORE their normalisations - y = linear1(cat[s, v2s(v)]) before bn1 + LeakyReLU,
        v_lin = linear2(v) before VectorBN and the gate - for a consumer that runs bn1, bn2 and what follows them in its own passes
        (the classifier's tail, _ops.GlobalMaxMeanPoolBNV).  Rows path only; v_lin may still be in flight on the side stream."""
        return self._forward_rows(x, prebn=True, pretail=True)

    def forward_cloud_s(self, s_point, s_cloud, v):
        """The block on (cat[s_point, expand(s_cloud)], v) - the reference's `svcat([x, expand_as(pooled)])` of sv_pointnet_cls.py:50-52 -
        WITHOUT the concatenation of the scalar halves: linear1 counts the per-cloud columns once per cloud (_ops.BinLinearCloud, identical
        outputs), the gate's mean over the rows of a broadcast column is the column.  s_point [B,N,Cp], s_cloud [B,Cc], v [B,N,3,Cv] (whole).
        Binarized blocks on device rows only; anything else concatenates and takes forward()."""
        lin1 = self.linear1
        B, N, Cp = s_point.shape
        Cc = s_cloud.shape[-1]
        binary = lin1.bw and lin1.ba
        ok = (config.SPLIT_BROADCAST and (binary or not (lin1.bw or lin1.ba)) and lin1.bias is None and s_point.is_cuda
              and self._v2s_cat_fusable(s_point, v) and lin1.in_features == Cp + Cc + 3 * v.shape[-1] and B * N >= 1024
              and lin1.out_features >= 64 and self.gate[0].out_features <= 256 and self.gate[2].out_features <= 256)
        if not ok:
            return self.forward((torch.cat([s_point, s_cloud.unsqueeze(1).expand(B, N, Cc)], dim=-1), v))
        lz = self.v2s.linear
        two = config.TWO_STREAM_BLOCKS and B * N >= config.TWO_STREAM_MIN_ROWS
        if two:
            main, side = torch.cuda.current_stream(s_point.device), _ops._side_stream(s_point.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                v_lin = self.linear2(v, vstats=True)
        # cat[s_point, Vector2Scalar(v)] in place + the per-cloud mean of s_point (one consumer of s_point in the autograd graph)
        cat_pt, mean_pt = _ops.V2SCat.apply(s_point, v, lz.weight, lz.scale if lz.bw else None, self.training, B)
        v_scale = _ops.GateMLP.apply(torch.cat([mean_pt, s_cloud], dim=-1), self.gate[0].weight, self.gate[2].weight)
        if binary:
            y = _ops.BinLinearCloud.apply(s_cloud, cat_pt, lin1.weight, lin1.beta, lin1.scale, self.training, Cp)
        else:
            # full precision: the product is linear in its columns - per-point columns over B*N rows, per-cloud columns over B rows,
            # one broadcast add (the per-cloud block's gradients are then sums over each cloud's rows, formed by autograd)
            W = lin1.weight
            y = _ops.FpLinear.apply(cat_pt, torch.cat([W[:, :Cp], W[:, Cp + Cc:]], dim=1), None) \
                + _ops.FpLinear.apply(s_cloud, W[:, Cp:Cp + Cc], None).unsqueeze(1)
        s_out = batch_norm_act(self.bn1, y, _ACT_LEAKY, self.relu.negative_slope)
        if two:
            side.wait_stream(main)                       # (the gate came from the main stream)
            with torch.cuda.stream(side):
                v_out = self.bn2(v_lin, gate=v_scale)
            main.wait_stream(side)
            v_out.record_stream(main)
        else:
            v_out = self.bn2(self.linear2(v, vstats=True), gate=v_scale)
        return (s_out, v_out)

    def _forward_rows(self, x, prebn=False, pretail=False):
        s, v = x
        rows = s.numel() // max(s.shape[-1], 1)
        if config.TWO_STREAM_BLOCKS and rows >= config.TWO_STREAM_MIN_ROWS and s.is_cuda:
            # the two paths only share their inputs: the vector path goes to the side stream (fork / join with events, so the
            # pattern is captured into a HIP graph as two branches); its output is handed to the main stream's allocator view.
            # linear2 needs neither the gate nor s: its product starts at the fork, the gate (a pooling pass + a tiny MLP, ~50 us of
            # latency-bound launches on conv5 of the classifier) is computed on the main stream beside it, and only the VectorBN waits
            # for it - the vector path is the longer of the two and used to start after the gate
            main, side = torch.cuda.current_stream(s.device), _ops._side_stream(s.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                v_lin = self.linear2(v, vstats=True)
            fused = self._cat_and_gate(s, v)
            if fused is not None:
                s_cat, v_scale = fused
            else:
                v_scale = self._gate(s)
            if pretail:                                  # (the consumer forks / joins the side stream itself)
                return (self.linear1(s_cat if fused is not None else self._cat_s_v2s(s, v)), v_lin, v_scale)
            side.wait_stream(main)                       # (the gate came from the main stream)
            with torch.cuda.stream(side):
                v_out = self.bn2(v_lin, gate=v_scale)
            s_out = self.linear1(s_cat if fused is not None else self._cat_s_v2s(s, v))
            if not prebn:
                s_out = batch_norm_act(self.bn1, s_out, _ACT_LEAKY, self.relu.negative_slope)
            main.wait_stream(side)
            v_out.record_stream(main)
            return (s_out, v_out)

        v_scale = self._gate(s)
        s = self.linear1(self._cat_s_v2s(s, v))
        if not prebn:
            s = batch_norm_act(self.bn1, s, _ACT_LEAKY, self.relu.negative_slope)

        v = self.linear2(v, vstats=True)
        if pretail:
            return (s, v, v_scale)
        v = self.bn2(v, gate=v_scale)
        return (s, v)


class SVFuse(nn.Module):
    """cat[s, Vector2Scalar(v)] (optionally also returning the 3 x multi frame z)."""

    def __init__(self, v_dim, multi, binary, trans_back=False):
        super(SVFuse, self).__init__()
        print('SVFuse: ', v_dim)
        self.trans_back = trans_back
        self.v2s = Vector2Scalar(v_dim, multi, binary=binary, trans_back=trans_back)

    def parts(self, x):
        """(s, Vector2Scalar(v)) without the concatenation, for consumers that reduce over the points right away
        (pooling the parts and concatenating the small results equals pooling the concatenation)."""
        s, v = x
        return s, self.v2s(v)

    def forward(self, x):
        s, v = x
        if self.trans_back:
            s_v, trans = self.v2s(v)
            return torch.cat([s, s_v], dim=-1), trans
        lin = self.v2s.linear
        if (config.FUSE_V2S_CAT and torch.is_tensor(s) and torch.is_tensor(v) and s.is_cuda and lin.weight.shape[0] == 3 and v.shape[-1] <= 768
                and s.shape[:-1] == v.shape[:-2] and s.dim() >= 2):
            # the concatenation written in place by the Vector2Scalar kernel (no [.., 3C] intermediate, no cat pass; its backward reads the
            # Vector2Scalar columns of the gradient where they lie) - as SVBlock does for its linear1 input
            return _ops.V2SCat.apply(s, v, lin.weight, lin.scale if lin.bw else None, self.training)
        return torch.cat([s, self.v2s(v)], dim=-1)


class SV_STNkd(nn.Module):
    """Three per-point SVBlocks, pooling over the points, three per-cloud SVBlocks."""

    def __init__(self, dim, binary):
        super(SV_STNkd, self).__init__()
        widths = [dim, (64 // 2, 64 // 6), (128 // 2, 128 // 6), (1024 // 2, 1024 // 6),
                  (512 // 2, 512 // 6), (256 // 2, 256 // 6), dim]
        names = ['conv1', 'conv2', 'conv3', 'fc1', 'fc2', 'fc3']
        for i, name in enumerate(names):
            setattr(self, name, SVBlock(widths[i], widths[i + 1], binary=binary))

    def forward(self, x):
        for name in ('conv1', 'conv2', 'conv3'):
            x = getattr(self, name)(x)
        x = svpool(x, dim=1)
        for name in ('fc1', 'fc2', 'fc3'):
            x = getattr(self, name)(x)
        return x
