"""ctypes binding of libsvnet_hip.so (the C ABI declared in include/svnet_hip.h).

There is no CPU fallback: if the library is missing, or a tensor is not on a HIP device, the call
raises.  `build()` compiles the library in-tree with hipcc (gfx950 cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVNET_DIAG_LIB") or os.path.join(_HERE, "libsvnet_hip.so")    # (SVNET_DIAG_LIB: an ablation build, tools/ only)
_lib = None
ABI_VERSION = 417       # include/svnet_hip.h SVNET_ABI_VERSION: argument lists / buffer-length contracts this binding was written against

c_p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f = ctypes.c_float
c_sz = ctypes.c_size_t


class GemmDesc(ctypes.Structure):
    """struct svnet_gemm_desc (include/svnet_hip.h)."""
    _fields_ = [
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("A", c_p), ("a_rs", c_i64), ("a_cs", c_i64),
        ("a_scale", c_p),
        ("a_sign", c_p), ("a_nz", c_p),
        ("B", c_p), ("b_rs", c_i64), ("b_cs", c_i64),
        ("b_exact", c_int),
        ("C", c_p), ("ldc", c_i64), ("c_cs", c_i64),
        ("alpha", c_f),
        ("col_scale", c_p),
        ("bias", c_p),
        ("mask", c_p),
        ("col_sum", c_p),
        ("split_k", c_int),
        ("accumulate", c_int),
        ("tern_tile_mask", ctypes.c_uint32),
        ("workspace", c_p),
        ("workspace_bytes", ctypes.c_size_t),
    ]


class GateFwdJob(ctypes.Structure):
    """struct svnet_gate_fwd_job (include/svnet_hip.h): a gate MLP run beside a coefficient launch."""
    _fields_ = [("gin", c_p), ("gin_f64", c_p), ("gin_out", c_p), ("in_scale", c_f), ("W0", c_p), ("W2", c_p),
                ("B", c_i64), ("Cin", c_i64), ("H", c_i64), ("Ov", c_i64), ("h", c_p), ("gate", c_p), ("rows", c_p), ("R", c_i64)]


class BlockTailDesc(ctypes.Structure):
    """struct svnet_block_tail_desc (include/svnet_hip.h): coefficients + gate MLP + apply (+ k-NN table) of a fused level in one launch."""
    _fields_ = [("stat1", c_p), ("stat_v", c_p), ("E", c_i64), ("Os", c_i64), ("Ov", c_i64), ("scale1", c_p),
                ("gamma1", c_p), ("beta1", c_p), ("running_mean1", c_p), ("running_var1", c_p),
                ("gamma2", c_p), ("beta2", c_p), ("running_mean2", c_p), ("running_var2", c_p),
                ("training", c_int), ("eps", c_f), ("momentum", c_f),
                ("coef", c_p), ("num_batches_tracked1", c_p), ("num_batches_tracked2", c_p),
                ("gate", GateFwdJob),
                ("hi", c_p), ("lo", c_p), ("mv", c_p), ("mvn", c_p), ("P", c_i64), ("N", c_i64), ("slope", c_f),
                ("s_out", c_p), ("v_out", c_p), ("s_cat", c_p), ("s_ld", c_i64), ("v_cat", c_p), ("v_ld", c_i64),
                ("knn_workspace", c_p), ("knn_workspace_bytes", c_sz)]


class GateBwdJob(ctypes.Structure):
    """struct svnet_gate_bwd_job (include/svnet_hip.h)."""
    _fields_ = [("dgate", c_p), ("gate", c_p), ("h", c_p), ("gin", c_p), ("in_scale", c_f), ("W0", c_p), ("W2", c_p),
                ("B", c_i64), ("Cin", c_i64), ("H", c_i64), ("Ov", c_i64), ("out_scale", c_f), ("dgin", c_p), ("dW0", c_p), ("dW2", c_p)]


class EdgeBlockDesc(ctypes.Structure):
    """struct svnet_edgeblock_desc (include/svnet_hip.h)."""
    _fields_ = [
        ("B", c_i64), ("N", c_i64), ("k", c_i64),
        ("Cs", c_int), ("Cv", c_int), ("Os", c_int), ("Ov", c_int),
        ("s", c_p), ("v", c_p), ("idx", c_p),
        ("zz", c_p), ("ut", c_p),
        ("w_sign", c_p), ("w_nz", c_p), ("beta_perm", c_p),
        ("n_max", c_p), ("n_min", c_p), ("slot_max", c_p), ("slot_min", c_p),
        ("mv", c_p), ("mvn", c_p),
        ("stat_n", c_p), ("stat_v", c_p), ("gate_sum", c_p),
        ("n16", c_p), ("planes", c_p),
        ("w_dense", c_p),
    ]


class EdgeBlockBwdDesc(ctypes.Structure):
    """struct svnet_edgeblock_bwd_desc (include/svnet_hip.h)."""
    _fields_ = [
        ("B", c_i64), ("N", c_i64), ("k", c_i64),
        ("Cs", c_int), ("Cv", c_int), ("Os", c_int), ("Ov", c_int),
        ("v", c_p), ("idx", c_p), ("zz", c_p), ("ut", c_p),
        ("n16", c_p), ("planes", c_p),
        ("w1bt", c_p),
        ("scale1", c_p),
        ("slot_max", c_p), ("slot_min", c_p),
        ("coef", c_p), ("gate", c_p), ("gy", c_p), ("bcoef", c_p), ("gv", c_p), ("gconst", c_p),
        ("dn_out", c_p), ("x_sign32", c_p), ("x_nz32", c_p),
        ("msg", c_p),
        ("ub_tab", c_p), ("ge_tab", c_p),
        ("ds_acc", c_p), ("dv_acc", c_p), ("dvc", c_p), ("dzc", c_p), ("dbeta_perm", c_p),
        ("debug", c_p),
        ("parts", c_int),
    ]


class XyzBlockDesc(ctypes.Structure):
    """struct svnet_xyzblock_desc (include/svnet_hip.h)."""
    _fields_ = [
        ("B", c_i64), ("N", c_i64), ("k", c_i64),
        ("Os", c_int), ("Ov", c_int),
        ("x", c_p), ("idx", c_p),
        ("w0", c_p), ("wz", c_p), ("w1", c_p), ("w2", c_p),
        ("y_max", c_p), ("y_min", c_p), ("slot_max", c_p), ("slot_min", c_p),
        ("mv", c_p), ("mvn", c_p),
        ("stat_y", c_p), ("stat_v", c_p), ("gate_sum", c_p),
        ("nc", c_i64),
    ]


class XyzBlockBwdDesc(ctypes.Structure):
    """struct svnet_xyzblock_bwd_desc (include/svnet_hip.h)."""
    _fields_ = [
        ("B", c_i64), ("N", c_i64), ("k", c_i64),
        ("Os", c_int), ("Ov", c_int),
        ("x", c_p), ("idx", c_p),
        ("w0", c_p), ("wz", c_p), ("w1", c_p), ("w2", c_p),
        ("slot_max", c_p), ("slot_min", c_p),
        ("coef", c_p), ("bcoef", c_p), ("gate", c_p), ("gy", c_p), ("gv", c_p), ("gconst", c_p),
        ("gw", c_p),
        ("nc", c_i64),
    ]


class BinHeadDesc(ctypes.Structure):
    """struct svnet_binhead_desc (include/svnet_hip.h)."""
    _fields_ = [
        ("M", c_i64), ("K", c_i64), ("O", c_i64),
        ("W", c_p), ("w_sign", c_p), ("w_nz", c_p), ("wld", c_i64), ("w_b", c_p),
        ("scale", c_p), ("gamma", c_p), ("bn_beta", c_p),
        ("running_mean", c_p), ("running_var", c_p), ("nbt", c_p),
        ("training", c_int), ("eps", c_f), ("momentum", c_f), ("act", c_int), ("slope", c_f),
        ("x_sign", c_p), ("x_nz", c_p), ("x_ste", c_p), ("xc_sign", c_p), ("xc_nz", c_p),
        ("y", c_p), ("mean", c_p), ("invstd", c_p), ("out", c_p),
        ("g", c_p), ("dnT", c_p), ("dW", c_p), ("dscale", c_p), ("dgamma", c_p), ("dbn_beta", c_p),
        ("dx", c_p), ("dbeta_in", c_p),
    ]


# name -> (restype, argtypes); every symbol include/svnet_hip.h declares
SIGNATURES = {
    "svnet_version": (c_int, []),
    "svnet_stamp_u64": (c_int, [c_p, c_p]),
    "svnet_slices_sum_f32": (c_int, [c_p, c_i64, c_p]),
    "svnet_slices_sum_f64": (c_int, [c_p, c_i64, c_p]),
    "svnet_last_error": (ctypes.c_char_p, []),
    "svnet_knn_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "svnet_knn_f32": (c_int, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p, c_p, c_sz, c_p]),
    "svnet_knn_sv_f32": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_int, c_p, c_p, c_sz, c_p]),
    "svnet_knn_table_fusable": (c_int, [c_i64, c_i64, c_i64]),
    "svnet_block_tail_supported": (c_int, [c_i64, c_i64, c_i64, c_i64, c_int]),
    "svnet_edgeblock_tail_f32": (c_int, [c_p, c_p]),
    "svnet_xyzblock_tail_f32": (c_int, [c_p, c_p]),
    "svnet_knn_from_table_f32": (c_int, [c_p, c_sz, c_i64, c_i64, c_i64, c_int, c_p, c_p]),
    "svnet_edge_xyz_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_p, c_p]),
    "svnet_edge_diffcat_fwd_f32": (c_int, [c_p, c_p, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "svnet_edge_diffcat_bwd_f32": (c_int, [c_p, c_p, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "svnet_gemm_workspace_bytes": (ctypes.c_size_t, [c_i64, c_i64]),
    "svnet_gemm_f32": (c_int, [ctypes.POINTER(GemmDesc), c_p]),
    "svnet_binweight_prepare_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "svnet_binlinear_fwd_f32": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "svnet_binweight_grad_f32": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_int, c_int, c_p, c_i64, c_p]),
    "svnet_edgeblock_prepare_vec_f32": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "svnet_knn_reverse_i32": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p, c_p, c_p]),
    "svnet_edgeblock_msg_stride": (c_i64, [c_i64, c_i64, c_i64]),
    "svnet_edgeblock_bwd_gather_f32": (c_int, [c_p] * 9 + [c_i64] + [c_p, c_p] + [c_i64] * 5 + [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_p]),
    "svnet_edgeblock_bwd_params_f32": (c_int, [c_p] * 8 + [c_i64] * 4 + [c_p] * 6 + [c_p]),
    "svnet_edgeblock_prepare_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p]),
    "svnet_edgeblock_fwd_f32": (c_int, [ctypes.POINTER(EdgeBlockDesc), c_p]),
    "svnet_edgeblock_coeffs_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_f, c_f, c_p, c_p, c_p, c_p, c_p]),
    "svnet_edgeblock_apply_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p]),
    "svnet_edgeblock_apply_knn_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p, c_sz, c_p]),
    "svnet_edgeblock_wbt_bf16": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "svnet_edgeblock_bwd_prelude_f32": (c_int, [c_p] * 9 + [c_i64, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p]),
    "svnet_edgeblock_bwd_coeffs_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "svnet_edgeblock_bwd_f32": (c_int, [ctypes.POINTER(EdgeBlockBwdDesc), c_p]),
    "svnet_edgeblock_wgrad_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, ctypes.c_uint32, c_p]),
    "svnet_xyzblock_fwd_f32": (c_int, [ctypes.POINTER(XyzBlockDesc), c_p]),
    "svnet_xyzblock_coeffs_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_f, c_f, c_p, c_p, c_p, c_p, c_p]),
    "svnet_xyzblock_apply_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p]),
    "svnet_xyzblock_apply_knn_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p, c_sz, c_p]),
    "svnet_xyzblock_bwd_prelude_f32": (c_int, [c_p] * 8 + [c_i64, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p]),
    "svnet_xyzblock_bwd_f32": (c_int, [ctypes.POINTER(XyzBlockBwdDesc), c_p]),
    "svnet_binweight_i8_bytes": (c_sz, [c_i64, c_i64]),
    "svnet_binweight_pack_i8": (c_int, [c_p, c_i64, c_i64, c_p, c_p]),
    "svnet_binlinear_i8_fwd_f32": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    "svnet_binlinear_i8_cloud_fwd_f32": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "svnet_v2s_fwd_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "svnet_v2s_bwd_f32": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "svnet_v2s_cat_fwd_f32": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_p]),
    "svnet_v2s_cat_sum_supported": (c_int, [c_i64, c_i64, c_i64, c_i64]),
    "svnet_v2s_cat_sum_fwd_f32": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p]),
    "svnet_v2s_bwd_ld_f32": (c_int, [c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "svnet_vproject_fwd_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p]),
    "svnet_vproject_bwd_f32": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "svnet_colstats_f64": (c_int, [c_p, c_i64, c_i64, c_int, c_p, c_p]),
    "svnet_bn_finalize_f32": (c_int, [c_p, c_i64, c_i64, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    "svnet_bn_eval_stats_f32": (c_int, [c_p, c_p, c_i64, c_f, c_p, c_p, c_p]),
    "svnet_bn_act_fwd_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_f, c_p, c_p]),
    "svnet_bn_act_bwd_reduce_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_f, c_p, c_p]),
    "svnet_bn_act_bwd_apply_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_f, c_int, c_p, c_p]),
    "svnet_vbn_fwd_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p]),
    "svnet_vbn_fwd_stats_f32": (c_int, [c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p]),
    "svnet_vbn_bwd_reduce_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "svnet_vbn_bwd_apply_f32": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_int, c_p, c_p]),
    "svnet_pool_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_int]),
    "svnet_pool_fwd_f32": (c_int, [c_p, c_i64, c_i64, c_i64, c_int, c_p, c_i64, c_p, c_p, c_sz, c_p]),
    "svnet_pool_maxmean_fwd_f32": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_sz, c_p]),
    "svnet_bn_pool_fwd_f32": (c_int, [c_p] * 5 + [c_i64] * 3 + [c_int, c_f, c_p, c_p, c_i64, c_p, c_p, c_sz, c_int, c_p]),
    "svnet_bn_pool_bwd_f32": (c_int, [c_p, c_p, c_i64] + [c_p] * 6 + [c_i64] * 3 + [c_int, c_f, c_int, c_p, c_p, c_p]),
    "svnet_pool_bwd_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_int, c_p, c_p]),
    "svnet_pool_mean_bwd_add_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "svnet_pool_maxmean_bwd_f32": (c_int, [c_p, c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p]),
    "svnet_vtail_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "svnet_vtail_fwd_f32": (c_int, [c_p, c_p, c_f, c_f] + [c_p] * 9 + [c_i64] * 3 + [c_p, c_p, c_i64, c_p, c_p, c_sz, c_int, c_p]),
    "svnet_vtail_bwd_f32": (c_int, [c_p] * 9 + [c_i64, c_p] + [c_i64] * 3 + [c_p] * 5),
    "svnet_vtail_bwd_apply_f32": (c_int, [c_p] * 9 + [c_i64, c_p] + [c_i64] * 3 + [c_p, c_int, c_p, c_p]),
    "svnet_act_fwd_f32": (c_int, [c_p, c_i64, c_int, c_p, c_p]),
    "svnet_act_bwd_f32": (c_int, [c_p, c_p, c_i64, c_int, c_p, c_p]),
    "svnet_vlinear_stats_f32": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p]),
    "svnet_gate_mlp_fwd_f32": (c_int, [c_p, c_p, c_p, c_f, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p]),
    "svnet_gate_mlp_bwd_f32": (c_int, [c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p, c_p]),
    "svnet_adam_step_f32": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_p]),
    "svnet_sgd_step_f32": (c_int, [c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_int, c_p]),
    "svnet_adam_step_dev_f32": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_p, c_p]),
    "svnet_sgd_step_dev_f32": (c_int, [c_p, c_p, c_p, c_i64, c_p, c_p]),
    "svnet_smooth_ce_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_f, c_p, c_p, c_p, c_i64, c_p]),
    "svnet_binhead_pack_f32": (c_int, [c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "svnet_binhead_fwd_f32": (c_int, [ctypes.POINTER(BinHeadDesc), c_p]),
    "svnet_binhead_bwd_f32": (c_int, [ctypes.POINTER(BinHeadDesc), c_p]),
    "svnet_fplinear_small_bwd_f32": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p]),
}


class SvnetHipError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile svnet_amd/csrc/*.hip into svnet_amd/libsvnet_hip.so (hipcc, --offload-arch=gfx950)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise SvnetHipError("building libsvnet_hip.so failed")
    return LIB_PATH


def lib():
    """The loaded library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SvnetHipError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C svnet_amd/csrc). "
                "svnet_amd has no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.svnet_version() != ABI_VERSION:
            raise SvnetHipError("%s was built for ABI %d, this binding speaks %d (include/svnet_hip.h SVNET_ABI_VERSION): rebuild it"
                                % (LIB_PATH, handle.svnet_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().svnet_last_error()
        raise SvnetHipError("%s failed (%d): %s" % (what, code, msg.decode() if msg else "?"))


class KernelTimer:
    """Optional HIP-event stopwatch around ONE entry point (bench.py's roofline legs): events are recorded on
    the stream the kernel is launched on, immediately before and after the launch call.  `select(args)`
    may narrow the timing to launches with particular arguments (e.g. one layer's shape)."""

    def __init__(self, name, select=None):
        self.name, self.select, self.pairs = name, select, []

    def elapsed_ms(self):
        return [a.elapsed_time(b) for a, b in self.pairs]


TIMERS = []          # KernelTimer objects that are live (bench.py only; empty in normal operation)


class StepClock:
    """Diagnostic (tools/step_clock.py): start times of a step's C-ABI launches without a profiler.  While `CLOCK` is set, call()
    puts a one-thread kernel that stores the device's 100 MHz clock in front of every launch whose entry point passes `select`, on
    the launch's stream; captured into a HIP graph the stamps are replayed with it, and `read()` returns, for the last replay,
    [(microseconds since the first stamp, entry point, stream id)] in time order.  Each stamp costs ~2 us of stream time."""

    def __init__(self, device, select=None, slots=4096):
        import torch
        self.buf = torch.zeros(slots, dtype=torch.int64, device=device)
        self.names, self.select = [], select

    def mark(self, name):
        import torch
        if self.select is not None and not self.select(name):
            return
        i = len(self.names)
        if i >= self.buf.numel():
            return
        st = torch.cuda.current_stream(self.buf.device)
        self.names.append((name, st.cuda_stream))
        lib().svnet_stamp_u64(ctypes.c_void_p(self.buf.data_ptr() + 8 * i), ctypes.c_void_p(st.cuda_stream))

    def read(self):
        t = self.buf[:len(self.names)].cpu().tolist()
        t0 = min(t) if t else 0
        return sorted(((x - t0) / 100.0, n, s) for x, (n, s) in zip(t, self.names))


CLOCK = None         # a StepClock while tools/step_clock.py measures; None in normal operation


def call(name, *args):
    """Invoke one C-ABI entry point and raise on a non-zero return code."""
    fn = getattr(lib(), name)
    if CLOCK is not None:
        CLOCK.mark(name)
    hit = [t for t in TIMERS if t.name == name and (t.select is None or t.select(args))] if TIMERS else None
    if hit:
        import torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        a.record(st)
        rc = fn(*args)
        b.record(st)
        for t in hit:
            t.pairs.append((a, b))
    else:
        rc = fn(*args)
    if rc != 0:
        msg = lib().svnet_last_error()
        raise SvnetHipError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
