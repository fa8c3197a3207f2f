"""Run-time switches of the HIP path."""
import os

# Debug: check the k-NN inputs for NaN / Inf and raise, instead of relying on the kernels' id clamp (a host sync per call).
DEBUG_FINITE = os.environ.get("SVNET_DEBUG_FINITE", "0") not in ("", "0")

# Fuse get_graph_feature_sv -> binarized SVBlock -> svpool into one pass over the edges (csrc/edgeblock.hip).
# Off = tier 1: every tensor the reference materialises is materialised (used as the on-device cross-check).
FUSE_EDGE_BLOCKS = True

# SVBlock on materialised rows (conv5 of the DGCNN models, every per-point block of the PointNet models): run the vector path
# (linear2 -> VectorBN) on the side stream beside the scalar path (Vector2Scalar -> linear1 -> BatchNorm + LeakyReLU); autograd
# runs each path's backward on the stream of its forward, so the two backward chains overlap as well.
TWO_STREAM_BLOCKS = True
TWO_STREAM_MIN_ROWS = 4096

# Backward of the fused edge layers: entries of a reverse neighbour list that one wave of the gather kernel sums; longer lists
# (hub points of the feature-space graphs) are cut into chunks summed by other waves.  0 = one wave per whole list.
# Round 4, re-measured on the current kernels (two alternating runs each, profiles/r04_ab_gather_chunk.log): 32: 4.396 / 4.388 ms,
# 24: 4.400 / 4.404, 48: 4.378 / 4.381, 64: 4.379 / 4.368, 128: 4.376 / 4.371, 0: 4.367 / 4.363.  The deepest graph of the bench has 44 %
# of its edges in lists longer than 32 (max in-degree 206: tools/diag_indegree.py), so at 32 nearly half of conv4's messages went
# through the second, atomic launch; 128 keeps a bound on what one wave may be handed (a pathological hub) at the price of 0.008 ms.
GATHER_CHUNK = 128

# Backward of a dense (sign-weight / binarized) layer with many rows: the weight-gradient product on a helper stream beside the
# input-gradient product.
DW_BESIDE = False      # measured: +0.12 ms per step on sv_dgcnn_cls B=32 (5.75 against 5.62): the two products compete for the same CUs

# Classifier: conv5's BatchNorm + LeakyReLU inside the global [max | mean] pooling pass (no activated [B,N,512] tensor, no gradient of it).
FUSE_BN_POOL = True

# ... and conv5's VectorBN + gate, svfuse's Vector2Scalar and the pooling of its half as ONE pass over linear2's product each way
# (csrc/vtail.hip, _ops.GlobalMaxMeanPoolBNV): VectorBN's output, the [B,N,510] scalars and their gradient are never written.
FUSE_VTAIL = True
# ... and its backward's second pass recomputes dL/d(VectorBN's output) per point instead of reading the copy the first pass would store
FUSE_VTAIL_APPLY = True

# SVBlock on rows: cat[s, Vector2Scalar(v)] written in place by the Vector2Scalar kernel (no intermediate, no cat pass).
FUSE_V2S_CAT = True

# Binarized dense layers with >= 1024 rows: int8 ternary operands on the matrix cores instead of XNOR-popcounts on the vector ALU.
BINLINEAR_MFMA = True

# ... and its kernel leaves the column sums of the output for the BatchNorm that follows (no second pass over the output for the statistics).
FUSE_BN_STATS = True

# Rows layers whose input is cat[expand(per-cloud columns), per-point columns] (sv_dgcnn_partseg.py:115-121): the per-cloud block of a
# binarized layer is counted once per cloud and added to the per-point block's counts (identical outputs, _ops.BinLinearCloud) instead of
# being repeated over the N points - conv8 of sv_dgcnn_partseg: 1 600 of 2 144 columns.
SPLIT_BROADCAST = True

# Classifier heads (a binarized dense layer + BatchNorm + activation over batch-size rows): one fused pass forward, two backward
# (csrc/head.hip) instead of ~12 launch-bound kernels per layer.
FUSE_HEAD = True
# TrainStep: the fused edge layers' weight-gradient chains stay on the side stream until the one join before the gradients are packed
# (svnet_amd._ops._Deferred); False = every backward joins before it returns
DEFER_WGRAD = True
VEC_EARLY = True
# ... and so do the weight-gradient chains of the big rows layers (conv5.linear1 of the classifier: 111 + 11 us at the tail of the side
# stream instead of in front of the launch-bound end of conv5's backward): 4.29 -> 4.23 ms (profiles/r05_ab_defer_rows.log).  Opt-in per
# call site (_ops.defer_rows_wgrad): on every rows layer of sv_pointnet_cls it cost 0.38 ms per step
DEFER_ROWS_WGRAD = True
# (round 4 measured the gate's chain - per-cloud mean of s, MLP - of an SVBlock on rows on the side stream behind linear2's product:
#  4.62 ms against 4.57 - 4.60 with the gate on the main stream; the switch and its code path are gone.)
# sign-weight products with many rows and more than 128 columns go to the LDS-tiled rows kernel from this K on (it needs K >= 64)
ROWS2_MIN_K = 64

# Stream priorities (torch: lower number = higher priority; -1 is the highest this build hands out, 0 the default).  MAIN_PRIORITY is the
# priority of the stream a step is captured on (svnet_amd.train), SIDE_PRIORITY that of the side stream (_ops._side_stream: the vector
# path of an SVBlock on rows and the deferred weight-gradient chains).
MAIN_PRIORITY = 0
SIDE_PRIORITY = 0

# SVBlock on rows: linear2's product also forms the batch sums of the VectorBN behind it (csrc/vlinear.hip; K <= 96, O <= 256) - no
# statistics pass over its output.  Measured at conv5 of the classifier (32 768 points, 83 -> 170): alone 51 us + 57 us for the apply
# pass (which then reads the product cold) against 72 + 21 + 44 for rows GEMM + statistics + apply, but IN the step 4.44 ms against 4.40
# (two alternating runs: one 8-wave workgroup per CU holding 100 KB of LDS shares the CUs worse with the scalar path beside it than the
# three lighter kernels did), and nothing on sv_pointnet_cls (5.61 against 5.62): off.
FUSE_VBN_STATS = False

# SVBlock on rows, narrow s (the PointNet callers): the per-cloud mean of s the gate MLP starts from is formed inside the MLP's launch
# (_ops.GateMLPRows) instead of by a pooling pass of two launches in front of it
GATE_MEAN_INSIDE = True

# Fused forward edge kernels: when no weight of linear1 is exactly 0 (a device-side flag the packing kernel sets) the popcount products skip
# the weights' non-zero plane: the edge's own non-zero count is one scalar per edge, a word costs xor + and + bcnt instead of five instructions
EDGE_DENSE_WEIGHTS = True
# A fused level followed by get_graph_feature_sv prepares that k-NN's candidate table in its apply pass (_ops.knn_table_ahead)
KNN_TABLE_AHEAD = True
# ... and the coefficients + gate MLP + apply pass of a fused level are ONE launch (svnet_*_tail_f32) instead of two on the critical path
FUSE_BLOCK_TAIL = True
# conv5's concatenation kernel also sums the s columns it copies per cloud; the gate MLP starts from those sums (no pooling pass over s)
FUSE_CAT_MEAN = True
