"""SV-DGCNN part segmentation (caller of the hot path).

Same constructor `(args, num_part)`, channel rounding, sub-module names and forward order as the
reference models/sv_dgcnn_partseg.py:18-128; composition only.
"""
from .sv_layers import *
from .utils.sv_util import *
from .sv_layers import batch_norm_act, _ACT_LEAKY
from .. import _ops, config


def _round8(v, divisor=8):
    """Channel counts rounded to a multiple of 8, never more than 10 % below v (reference :18-31)."""
    r = max(divisor, int(v + divisor / 2) // divisor * divisor)
    return r + divisor if r < 0.9 * v else r


_V = _round8


class _ConvBNAct(nn.Sequential):
    """[conv, BatchNorm1d, LeakyReLU]; keys `.0.*`, `.1.*` as in the reference's nn.Sequential.  The kernels work on channel-LAST rows:
    forward_rows keeps them that way (the head below stays in rows from the concatenation to the logits - the reference's
    channel-first tensors cost two 0.5 GB transposed copies per layer at B=32, N=2048); forward() is the reference's [B,C,N] form."""

    def forward_rows(self, rows):                                    # rows: [..., C] -> [..., O]
        # (the deferred weight-gradient schedule of sv_dgcnn_cls's conv5 - _ops.defer_rows_wgrad - was measured here: head layers 15.8 -> 16.0 ms,
        #  conv5 15.79 / 15.96 -> 15.78 / 15.86: not used)
        y = self[0].forward_rows(rows) if isinstance(self[0], Conv1d) else \
            _ops.FpLinear.apply(rows, self[0].weight.view(self[0].out_channels, -1), None)
        return batch_norm_act(self[1], y, _ACT_LEAKY, 0.2)

    def forward_rows_split(self, x_cloud, x_point):
        """The layer on cat[expand(x_cloud [B,Kc]), x_point [B,N,Kp]] (the reference's `repeat` + `cat`, sv_dgcnn_partseg.py:115-121)
        WITHOUT the concatenation: a binarized layer adds the per-cloud block's integer counts - B rows - to the per-point block's
        (_ops.BinLinearCloud: identical outputs); anything else concatenates."""
        conv = self[0]
        if config.SPLIT_BROADCAST and isinstance(conv, Conv1d) and conv.binary and x_point.is_cuda and x_point.dim() == 3:
            y = _ops.BinLinearCloud.apply(x_cloud, x_point, conv.weight, conv.beta, conv.scale, conv.training)
            return batch_norm_act(self[1], y, _ACT_LEAKY, 0.2)
        B, N = x_point.shape[:2]
        return self.forward_rows(torch.cat([x_cloud.unsqueeze(1).expand(B, N, x_cloud.shape[-1]), x_point], dim=-1))

    def forward(self, x):                                            # x: [B,C,N]
        return self.forward_rows(x.transpose(1, 2)).transpose(1, 2).contiguous()


class SV_DGCNN_PSEG(nn.Module):
    def __init__(self, args, num_part):
        super(SV_DGCNN_PSEG, self).__init__()
        self.args = args
        self.k = args.k
        self.binary = args.binary
        self.dropout = 0 if self.binary else args.dropout
        self.emb = 1024
        b, emb = self.binary, self.emb
        s1, v1 = _V(64 // 2), _V(64 // 6)
        s3, v3 = _V(128 // 2), _V(128 // 6)
        s4, v4 = _V(256 // 2), _V(256 // 6)
        cs, cv = s1 * 2 + s3 + s4, v1 * 2 + v3 + v4

        self.init_scalar = Vector2Scalar(2, 3)
        self.conv1 = SVBlock((6, 2), (s1, v1))
        self.conv2 = SVBlock((s1 * 2, v1 * 2), (s1, v1), b)
        self.conv3 = SVBlock((s1 * 2, v1 * 2), (s3, v3), b)
        self.conv4 = SVBlock((s3 * 2, v3 * 2), (s4, v4), b)

        self.svfuse1 = SVFuse(cv, 3, b)
        self.conv5 = SVBlock((cs, cv), (_V(emb // 2), _V(emb // 6)), b)
        self.conv6 = SVBlock((_V(emb // 2), _V(emb // 6)), (_V(emb // 4), _V(emb // 12)), b)
        self.svfuse2 = SVFuse(_V(emb // 12), 3, b)
        self.svfuse3 = SVFuse(_V(emb // 6), 3, b)
        self.conv7 = _ConvBNAct(nn.Conv1d(16, 64, kernel_size=1, bias=False), nn.BatchNorm1d(64),
                                nn.LeakyReLU(negative_slope=0.2))
        head_in = _V(emb // 2) + _V(emb // 4) + (_V(emb // 6) + _V(emb // 12)) * 3 + 64 + cs + cv * 3
        self.conv8 = _ConvBNAct(Conv1d(head_in, 256, b), nn.BatchNorm1d(256), nn.LeakyReLU(negative_slope=0.2))
        self.dp1 = nn.Dropout(p=self.dropout)
        self.conv9 = _ConvBNAct(Conv1d(256, 256, b), nn.BatchNorm1d(256), nn.LeakyReLU(negative_slope=0.2))
        self.dp2 = nn.Dropout(p=self.dropout)
        self.conv10 = _ConvBNAct(Conv1d(256, 128, b), nn.BatchNorm1d(128), nn.LeakyReLU(negative_slope=0.2))
        self.conv11 = nn.Conv1d(128, num_part, kernel_size=1, bias=False)

    def forward(self, x, l):
        B, N = x.size(0), x.size(2)
        blocks = (self.conv1, self.conv2, self.conv3, self.conv4)
        # (the pyramid's concatenation is written in place by the fused levels' apply kernels: _ops.CatSink, as in sv_dgcnn_cls)
        sink = _ops.CatSink([b.linear1.out_features for b in blocks], [b.linear2.out_features for b in blocks])
        with sink:
            v = get_graph_feature(x.unsqueeze(1), k=self.k)
            with _ops.knn_table_ahead():                                  # (the next level's k-NN table from this level's apply pass)
                level = svpool(self.conv1((self.init_scalar(v), v)))
            pyramid = [level]
            for block in (self.conv2, self.conv3, self.conv4):
                edges = get_graph_feature_sv(level, k=self.k)
                if block is self.conv4:
                    level = svpool(block(edges))
                else:
                    with _ops.knn_table_ahead():
                        level = svpool(block(edges))
                pyramid.append(level)

        x = sink.result(pyramid)
        fine = self.svfuse1(x)                                        # [B,N,cs+3cv]
        x = self.conv5(x)
        pooled = self.svfuse2(self.conv6(svpool(x, dim=1, keepdim=True)))          # [B,1,emb/2]
        # max over the points of cat[s, Vector2Scalar(v)] = cat of the two parts' maxima: the [B,N,1016] concatenation is never built
        s5, sv5 = self.svfuse3.parts(x)
        glob = _ops.PoolMaxParts.apply(s5, sv5).unsqueeze(-1)                      # [B,emb,1]

        lab = self.conv7(l.view(B, -1, 1))                            # [B,64,1]
        # the head on channel-LAST rows [B,N,head_in] = [glob | pooled | lab (one row per cloud, broadcast) | fine]: the same columns
        # in the same order as the reference's channel-first cat (sv_dgcnn_partseg.py:117-121), written once
        percloud = torch.cat([glob, pooled.transpose(-1, -2), lab], dim=1).transpose(1, 2)          # [B,1,1600]
        # (the 1 600 per-cloud columns are NOT repeated over the N points: conv8 counts them once per cloud, _ops.BinLinearCloud)
        rows = self.dp1(self.conv8.forward_rows_split(percloud.reshape(B, -1), fine))
        rows = self.dp2(self.conv9.forward_rows(rows))
        rows = self.conv10.forward_rows(rows)
        w = self.conv11.weight.view(self.conv11.out_channels, -1)
        return _ops.FpLinear.apply(rows, w, None).transpose(1, 2).contiguous()                       # [B,num_part,N]
