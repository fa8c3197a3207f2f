"""SV-DGCNN classifier (caller of the hot path).

Same constructor `(args, num_class)`, sub-module names and forward order as the reference
models/sv_dgcnn_cls.py:22-82, so its state_dicts load unchanged; composition only — the arithmetic is
in .sv_layers / .utils.sv_util (HIP kernels).
"""
from .sv_layers import *
from .utils.sv_util import *
from .sv_layers import batch_norm_act, linear_bn_act, _ACT_LEAKY, _bn_momentum
from .. import _ops, config


class SV_DGCNN_CLS(nn.Module):
    def __init__(self, args, num_class=40):
        super(SV_DGCNN_CLS, self).__init__()
        self.k = args.k
        self.binary = args.binary
        drop = 0 if self.binary else 0.5
        b = self.binary

        self.init_scalar = Vector2Scalar(2, 3)
        self.conv1 = SVBlock((6, 2), (32, 10))                       # never binarized (reference :30)
        self.conv2 = SVBlock((64, 20), (32, 10), b)
        self.conv3 = SVBlock((64, 20), (64, 21), b)
        self.conv4 = SVBlock((128, 42), (128, 42), b)
        self.conv5 = SVBlock((32 + 32 + 64 + 128, 10 + 10 + 21 + 42), (512, 170), b)
        self.svfuse = SVFuse(170, 3, b)

        self.linear1 = Linear((512 + 170 * 3) * 2, 512, bias=False, bw=b, ba=b)
        self.bn1 = nn.BatchNorm1d(512)
        self.dp1 = nn.Dropout(p=drop)
        self.linear2 = Linear(512, 256, bias=False, bw=b, ba=b)
        self.bn2 = nn.BatchNorm1d(256)
        self.dp2 = nn.Dropout(p=drop)
        self.linear3 = nn.Linear(256, num_class)

    def forward(self, x):
        blocks = (self.conv1, self.conv2, self.conv3, self.conv4)
        # svcat([x1, x2, x3, x4]) (reference :68) is written in place: every fused level's apply kernel stores its pooled (s, v) a second
        # time as its column slice of the concatenation (no concatenation pass in front of conv5; torch.cat when a level is not fused)
        sink = _ops.CatSink([b.linear1.out_features for b in blocks], [b.linear2.out_features for b in blocks])
        with sink:
            v = get_graph_feature(x.unsqueeze(1), k=self.k)               # [B,N,k,3,2]
            # (a level whose output the next level's k-NN reads prepares that k-NN's table in its apply pass: _ops.knn_table_ahead)
            with _ops.knn_table_ahead():
                level = svpool(self.conv1((self.init_scalar(v), v)))
            pyramid = [level]
            for block in (self.conv2, self.conv3, self.conv4):             # dynamic feature-space graph per level
                edges = get_graph_feature_sv(level, k=self.k)
                if block is self.conv4:
                    level = svpool(block(edges))
                else:
                    with _ops.knn_table_ahead():
                        level = svpool(block(edges))
                pyramid.append(level)

        # feat = svfuse(conv5(.)) is [B,N,1022] = [s | s_v]; it is only ever pooled over the points, so its two parts are pooled
        # where they are ([max | mean] with one shared backward pass each) and the [B,.] results are put in the reference's order
        x5 = sink.result(pyramid)
        bn = self.conv5.bn1
        if (config.FUSE_BN_POOL and x5[0].is_cuda and x5[0].dim() == 3 and bn.track_running_stats
                and _ops.GlobalMaxMeanPoolBN.supported(x5[0].shape[0], x5[0].shape[1], self.conv5.linear1.out_features)):
            # bn1 + LeakyReLU of conv5 run inside the pooling pass (the activated [B,N,512] tensor and its gradient are never written)
            bn2, fz = self.conv5.bn2.bn, self.svfuse.v2s.linear
            if (config.FUSE_VTAIL and not self.svfuse.trans_back and self.conv5._default_bn() and bn2.affine and fz.weight.shape[0] == 3
                    and bn.training == bn2.training
                    and _ops.GlobalMaxMeanPoolBNV.supported(x5[0].shape[0], x5[0].shape[1], self.conv5.linear1.out_features, fz.weight.shape[1])):
                # ... and so do VectorBN, the gate, svfuse's Vector2Scalar and the pooling of ITS half, in one pass over linear2's product
                with _ops.defer_rows_wgrad():                # (its weight-gradient chain may leave the backward's critical path)
                    y5, v_lin, gate = self.conv5.forward_pretail(x5)
                pooled = _ops.GlobalMaxMeanPoolBNV.apply(
                    y5, v_lin, gate, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn2.weight, bn2.bias, bn2.running_mean,
                    bn2.running_var, fz.weight, fz.scale if fz.bw else None, bn.training, _ACT_LEAKY, self.conv5.relu.negative_slope,
                    bn.num_batches_tracked if bn.training else None, bn2.num_batches_tracked if bn2.training else None, bn.eps, _bn_momentum(bn))
                h = self.dp1(linear_bn_act(self.linear1, self.bn1, pooled, _ACT_LEAKY, 0.2))
                h = self.dp2(linear_bn_act(self.linear2, self.bn2, h, _ACT_LEAKY, 0.2))
                return _ops.FpLinear.apply(h, self.linear3.weight, self.linear3.bias)
            with _ops.defer_rows_wgrad():
                y5, v5 = self.conv5.forward_prebn(x5)
            sv5 = self.svfuse.v2s(v5)
            nbt = bn.num_batches_tracked if bn.training else None
            pooled = _ops.GlobalMaxMeanPoolBN.apply(y5, sv5, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, _ACT_LEAKY,
                                                    self.conv5.relu.negative_slope, nbt, bn.eps, _bn_momentum(bn))
        else:
            s5, sv5 = self.svfuse.parts(self.conv5(x5))
            pooled = _ops.GlobalMaxMeanPool.apply(s5, sv5)               # [max s | max s_v | mean s | mean s_v] = max | mean of cat[s, s_v]

        h = self.dp1(linear_bn_act(self.linear1, self.bn1, pooled, _ACT_LEAKY, 0.2))
        h = self.dp2(linear_bn_act(self.linear2, self.bn2, h, _ACT_LEAKY, 0.2))
        return _ops.FpLinear.apply(h, self.linear3.weight, self.linear3.bias)
