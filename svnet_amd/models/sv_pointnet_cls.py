"""SV-PointNet classifier (caller of the hot path).

Same constructor signatures, sub-module names and forward order as the reference
models/sv_pointnet_cls.py:12-81 (SVPointNetEncoder, SV_PointNet_CLS); composition only.
"""
from .sv_layers import *
from .utils.sv_util import *
from .sv_layers import batch_norm_act, linear_bn_act, _ACT_RELU
from .. import _ops


def _broadcast_like(g, x):
    return (g[0].expand_as(x[0]), g[1].expand_as(x[1]))


class SVPointNetEncoder(nn.Module):
    def __init__(self, k, binary):
        super(SVPointNetEncoder, self).__init__()
        self.k = k
        self.binary = binary

        self.init_scalar = Vector2Scalar(3, 3)
        self.conv_pos = SVBlock((9, 3), (32, 10))                     # never binarized (reference :19)
        self.conv1 = SVBlock((32, 10), (32, 10), binary=binary)
        self.fstn = SV_STNkd((32, 10), binary=binary)
        self.conv2 = SVBlock((64, 20), (64, 21), binary=binary)
        self.conv3 = SVBlock((64, 21), (512, 170), binary=binary)
        self.conv_fuse = SVBlock((1024, 340), (512, 170), binary=binary)
        self.svfuse = SVFuse(170, 3, binary=binary)

    def forward(self, x):
        v = get_graph_feature_cross(x.unsqueeze(1), k=self.k)         # [B,N,k,3,3]
        x = svpool(self.conv_pos((self.init_scalar(v), v)))
        x = self.conv1(x)

        g = self.fstn(x)                                              # per-cloud (s [B,32], v [B,3,10])
        x = svcat([x, _broadcast_like((g[0].unsqueeze(1), g[1].unsqueeze(1)), x)])
        x = self.conv3(self.conv2(x))

        # svcat([x, expand_as(pooled)]) -> conv_fuse (reference :50-52): the scalar halves are not concatenated - conv_fuse.linear1 counts the
        # 512 per-cloud columns of its 2 044 once per cloud (SVBlock.forward_cloud_s); the vector halves are (Vector2Scalar mixes them per point)
        gs, gv = svpool(x, dim=1, keepdim=True)
        v_cat = torch.cat([x[1], gv.expand_as(x[1])], dim=-1)
        x = svpool(self.conv_fuse.forward_cloud_s(x[0], gs.reshape(gs.shape[0], -1), v_cat), dim=1)
        return self.svfuse(x)                                         # [B,1022]


class SV_PointNet_CLS(nn.Module):
    def __init__(self, args, num_class=40):
        super(SV_PointNet_CLS, self).__init__()
        self.binary = args.binary
        self.k = args.k
        drop = 0 if self.binary else 0.4

        self.feat = SVPointNetEncoder(k=self.k, binary=self.binary)
        self.fc1 = Linear(512 + 170 * 3, 512, bias=False, bw=self.binary, ba=self.binary)
        self.fc2 = Linear(512, 256, bias=False, bw=self.binary, ba=self.binary)
        self.fc3 = nn.Linear(256, num_class)
        self.dropout = nn.Dropout(p=drop)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()

    def forward(self, x):
        x = self.feat(x)
        x = linear_bn_act(self.fc1, self.bn1, x, _ACT_RELU)
        if self.dropout.p == 0:                                        # (binary: dropout(p=0) between fc2 and bn2 is the identity)
            x = linear_bn_act(self.fc2, self.bn2, x, _ACT_RELU)
        else:
            x = batch_norm_act(self.bn2, self.dropout(self.fc2(x)), _ACT_RELU)
        return _ops.FpLinear.apply(x, self.fc3.weight, self.fc3.bias)
