"""ctypes binding of libdiffassemble_hip.so (C ABI: include/diffassemble_hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every
operator raises.  ``import torch`` happens first on purpose -- PyTorch-ROCm bundles its own
libamdhip64 (SONAME libamdhip64.so.7); loading ours afterwards makes the dynamic loader
bind our DT_NEEDED entry to that already-loaded runtime, so streams and device pointers are
shared with torch instead of living in a second HIP runtime.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

LIB_PATH = os.environ.get("DA_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdiffassemble_hip.so")
# (DA_LIB_PATH: an alternate BUILD of the same library -- compile-flag A/Bs, tools/build_ab.sh; never a different implementation)
DA_MAX_LAYERS = 8
ABI_VERSION = 19
PREC_F32, PREC_BF16 = 0, 1
VARIANT_2D, VARIANT_3D = 0, 1
ARCH_TRANSFORMER, ARCH_EXOPHORMER = 0, 1
MEAN_EPSILON, MEAN_START_X = 0, 1
ACT_NONE, ACT_GELU, ACT_LEAKY02 = 0, 1, 2
CONV_Q_PRESCALED, CONV_FOLDED_V32 = 1, 2
TRAIN_MMA_FP32, TRAIN_MMA_BF16 = 0, 1
TRAIN_BWD_ALL, TRAIN_BWD_EARLY, TRAIN_BWD_LATE = 0, 1, 2
DBG_COUNTERS = ("opt_gen_workgroups", "dense_fast_exits", "dual_gen_slabs", "opt_masked_gen_workgroups")          # fallback events (indices 0 .. 3)
DBG_LAUNCH_COUNTERS = {"resident_launches": 4, "virtual_rows_in_launch": 5}          # which kernel took a layer (DA_DBG_RES_LAUNCHES, DA_DBG_VIRT_IN_LAUNCH)
PROF_CLASSES = ("embed", "linear_mlp", "linear_qkvs", "attn_hidden", "attn_last", "head", "update", "conv_fused")

_fp = C.c_void_p        # device pointers travel as integers
_FPL = _fp * DA_MAX_LAYERS


class DaWeights(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("arch", C.c_int32), ("steps", C.c_int32), ("c_in", C.c_int32),
        ("c_out", C.c_int32), ("feat_dim", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32),
        ("n_layers", C.c_int32), ("virt_nodes", C.c_int32),
        ("time_emb", _fp),
        ("pos_w0", _fp), ("pos_b0", _fp), ("pos_w1", _fp), ("pos_b1", _fp),
        ("mlp_w0", _fp), ("mlp_b0", _fp), ("mlp_w1", _fp), ("mlp_b1", _fp),
        ("conv_wq", _FPL), ("conv_bq", _FPL), ("conv_wk", _FPL), ("conv_bk", _FPL),
        ("conv_wv", _FPL), ("conv_bv", _FPL), ("conv_ws", _FPL), ("conv_bs", _FPL),
        ("virt_emb", _fp),
        ("head_w0", _fp), ("head_b0", _fp), ("head_w1", _fp), ("head_b1", _fp),
        ("head_r_w0", _fp), ("head_r_b0", _fp), ("head_r_w1", _fp), ("head_r_b1", _fp),
    ]


class DaGraph(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("n_real", C.c_int32), ("n_graphs", C.c_int32), ("dense", C.c_int32),
        ("n_edges", C.c_int64),
        ("row_ptr", _fp), ("col_src", _fp), ("edge_id", _fp), ("graph_ptr", _fp),
        ("max_graph_nodes", C.c_int32), ("n_pad", C.c_int32),
        ("pad_ptr", _fp), ("row_map", _fp),
        ("out_ptr", _fp), ("out_dst", _fp),
        ("hybrid", C.c_int32), ("reserved0", C.c_int32),
        ("mask", _fp), ("mask_ptr", _fp), ("irr_row_ptr", _fp), ("irr_col_src", _fp),
        ("slot_node", _fp), ("blk_class", _fp), ("blk_class_ptr", _fp), ("blk_class_stride", C.c_int32), ("reserved1", C.c_int32),
        ("rm_meta", _fp), ("agg_row_ptr", _fp), ("agg_col_src", _fp), ("agg_mult", _fp),
    ]


class DaLoopOpts(C.Structure):
    """da_loop_opts: sampler 0 = DDIM / 1 = DDPM, eta, classifier-free guidance switch + weight, per-iteration noise."""
    _fields_ = [("sampler", C.c_int32), ("eta", C.c_float), ("cfg", C.c_int32), ("cfg_w", C.c_float), ("noise", C.c_void_p)]


class DaSchedule(C.Structure):
    _fields_ = [
        ("steps", C.c_int32),
        ("betas", _fp), ("alphas_cumprod", _fp), ("sqrt_recip_alphas", _fp),
        ("sqrt_recip_alphas_cumprod", _fp), ("sqrt_recipm1_alphas_cumprod", _fp),
        ("sqrt_one_minus_alphas_cumprod", _fp), ("posterior_variance", _fp),
    ]


ENCODER_CONVS, ENCODER_FEATS = 19, 1088


class DaEncoderWeights(C.Structure):
    _fields_ = [
        ("n_convs", C.c_int32), ("reserved0", C.c_int32),
        ("stem_w", _fp), ("stem_b", _fp),
        ("conv_w", _fp * ENCODER_CONVS), ("conv_b", _fp * ENCODER_CONVS),
        ("lin1_w", _fp), ("lin1_b", _fp), ("lin2_w", _fp), ("lin2_b", _fp),
    ]


PCD_STAGES, PCD_K = 3, 20


class DaPcdEncoderWeights(C.Structure):
    _fields_ = [
        ("feat_dim", C.c_int32), ("reserved0", C.c_int32),
        ("premap", _fp * PCD_STAGES), ("bn_a", _fp * PCD_STAGES), ("conv_b", _fp * PCD_STAGES),
        ("conv6", _fp), ("linear0", _fp),
    ]


class DaConfig(C.Structure):
    """include/diffassemble_hip.h `da_config`: the library's switches (one environment variable each, settable at run time)."""
    _fields_ = [(n, C.c_int32) for n in ("struct_bytes", "disable_mfma", "disable_dense", "disable_folds", "attn_level", "xpanel", "tail_next",
                                         "pair_split", "train_attn", "train_side_streams")]


# name -> (restype, argtypes); every symbol include/diffassemble_hip.h declares
PROTOTYPES = {
    "da_abi_version": (C.c_int, []),
    "da_last_error": (C.c_char_p, []),
    "da_config_get": (C.c_int, [C.POINTER(DaConfig)]),
    "da_config_set": (C.c_int, [C.POINTER(DaConfig)]),
    "da_build_flags": (C.c_int, []),
    "da_denoiser_create": (C.c_int, [C.POINTER(DaWeights), C.c_int, _fp, C.POINTER(_fp)]),
    "da_denoiser_destroy": (None, [_fp]),
    "da_denoiser_flags": (C.c_int, [_fp]),
    "da_denoiser_workspace_bytes": (C.c_size_t, [_fp, C.POINTER(DaGraph)]),
    "da_denoiser_set_features": (C.c_int, [_fp, C.POINTER(DaGraph), _fp, _fp, C.c_size_t, _fp]),
    "da_denoiser_forward": (C.c_int, [_fp, C.POINTER(DaGraph), _fp, _fp, C.c_int64, _fp, _fp, C.c_int, _fp, _fp,
                                      C.c_size_t, _fp]),
    "da_ddim_step": (C.c_int, [C.POINTER(DaSchedule), C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp,
                               C.c_int64, C.c_int, C.c_int, C.c_float, _fp, _fp, _fp]),
    "da_ddpm_step": (C.c_int, [C.POINTER(DaSchedule), C.c_int, C.c_int, _fp, _fp, _fp, C.c_int64, _fp, _fp, _fp]),
    "da_sample_loop": (C.c_int, [_fp, C.POINTER(DaGraph), C.POINTER(DaSchedule), C.c_int, C.c_int, C.c_int,
                                 _fp, _fp, _fp, _fp, C.c_size_t, C.c_int, _fp]),
    "da_sample_loop_ex": (C.c_int, [_fp, C.POINTER(DaGraph), C.POINTER(DaSchedule), C.c_int, C.c_int, C.c_int,
                                    _fp, _fp, _fp, _fp, C.c_size_t, C.c_int, C.POINTER(DaLoopOpts), _fp]),
    "da_sample_loop_pair": (C.c_int, [_fp, C.POINTER(DaSchedule), C.c_int, C.c_int, C.c_int,
                                      C.POINTER(DaGraph), _fp, _fp, _fp, C.c_size_t,
                                      C.POINTER(DaGraph), _fp, _fp, _fp, C.c_size_t, _fp]),
    "da_sample_loop_pair_traj": (C.c_int, [_fp, C.POINTER(DaSchedule), C.c_int, C.c_int, C.c_int,
                                           C.POINTER(DaGraph), _fp, _fp, _fp, C.c_size_t,
                                           C.POINTER(DaGraph), _fp, _fp, _fp, C.c_size_t, _fp, _fp, C.c_size_t, _fp]),
    "da_sample_loop_pair_ex": (C.c_int, [_fp, C.POINTER(DaSchedule), C.c_int, C.c_int, C.c_int,
                                         C.POINTER(DaGraph), _fp, _fp, _fp, C.c_size_t,
                                         C.POINTER(DaGraph), _fp, _fp, _fp, C.c_size_t, _fp, _fp, C.c_size_t,
                                         C.POINTER(DaLoopOpts), _fp, C.c_size_t, _fp]),
    "da_profile_enable": (C.c_int, [_fp, C.c_int]),
    "da_profile_read": (C.c_int, [_fp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "da_linear": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, _fp, _fp,
                            C.c_int, _fp]),
    "da_linear_packed_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "da_linear_pack": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp]),
    "da_linear_packed": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, _fp, C.c_int, _fp, _fp,
                                   C.c_int, _fp]),
    "da_attn_csr": (C.c_int, [C.c_int, C.POINTER(DaGraph), C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, _fp, _fp]),
    "da_attn_dense_scratch_bytes": (C.c_size_t, [C.c_int, C.POINTER(DaGraph), C.c_int, C.c_int]),
    "da_conv_dense": (C.c_int, [C.c_int, C.POINTER(DaGraph), C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int,
                                _fp, _fp, _fp]),
    "da_conv_dense_ex": (C.c_int, [C.c_int, C.POINTER(DaGraph), C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int,
                                   _fp, _fp, C.c_int, _fp]),
    "da_debug_counters": (C.c_int, [C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "da_train_workspace_bytes": (C.c_size_t, [C.POINTER(DaWeights), C.POINTER(DaGraph)]),
    "da_train_workspace_bytes_ex": (C.c_size_t, [C.POINTER(DaWeights), C.POINTER(DaGraph), C.c_int]),
    "da_train_forward": (C.c_int, [C.POINTER(DaWeights), C.POINTER(DaGraph), _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "da_train_forward_ex": (C.c_int, [C.POINTER(DaWeights), C.POINTER(DaGraph), _fp, _fp, _fp, _fp, _fp, C.c_size_t, C.c_int, _fp]),
    "da_train_backward_ex": (C.c_int, [C.POINTER(DaWeights), C.POINTER(DaWeights), C.POINTER(DaGraph), _fp, _fp, _fp, _fp,
                                       _fp, C.c_size_t, C.c_int, _fp]),
    "da_train_backward_stage": (C.c_int, [C.POINTER(DaWeights), C.POINTER(DaWeights), C.POINTER(DaGraph), _fp, _fp, _fp, _fp,
                                          _fp, C.c_size_t, C.c_int, C.c_int, _fp]),
    "da_q_sample": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "da_loss_grad": (C.c_int, [C.c_int, C.c_size_t, _fp, _fp, _fp, _fp, _fp]),
    "da_adafactor_step": (C.c_int, [C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_size_t, C.c_int, C.c_float,
                                    C.c_float, C.c_float, C.c_float, _fp]),
    "da_greedy_assign": (C.c_int, [C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, _fp, _fp]),
    "da_encoder_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "da_encoder_forward": (C.c_int, [C.c_int, C.POINTER(DaEncoderWeights), C.c_int, _fp, _fp, C.c_int, _fp, C.c_size_t,
                                     C.c_int, C.c_int, _fp]),
    "da_pcd_encoder_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "da_pcd_encoder_forward": (C.c_int, [C.POINTER(DaPcdEncoderWeights), C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp,
                                         C.c_size_t, C.c_int, _fp]),
    "da_knn": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp]),
    "da_nearest_sq": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp]),
    "da_enc_train_scratch_bytes": (C.c_size_t, [C.c_int]),
    "da_enc_conv": (C.c_int, [C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "da_enc_stem": (C.c_int, [C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int, _fp]),
    "da_enc_stem_im2col": (C.c_int, [C.c_int, C.c_int, _fp, _fp, _fp]),
    "da_enc_bn_stats": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp]),
    "da_enc_bn_apply": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, _fp, _fp]),
    "da_enc_bn_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    "da_enc_upsample2": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "da_gemm_tn_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp]),
    "da_gemm_tn_bf16": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp]),
    "da_colsum_f32": (C.c_int, [C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, _fp]),
    "da_enc_bank_grad": (C.c_int, [C.c_int, _fp, _fp, _fp, _fp]),
    "da_expander_mask": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, _fp]),
    "da_train_backward": (C.c_int, [C.POINTER(DaWeights), C.POINTER(DaWeights), C.POINTER(DaGraph), _fp, _fp, _fp, _fp,
                                    _fp, C.c_size_t, _fp]),
}

_lib = None


class DaError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DaError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  diffassemble_amd has no CPU / eager fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.da_abi_version() != ABI_VERSION:
            raise DaError(f"ABI mismatch: library reports {h.da_abi_version()}")
        _lib = h
    return _lib


def config():
    """The library's current switches as a ``DaConfig``."""
    c = DaConfig()
    check(lib().da_config_get(C.byref(c)))
    return c


def set_config(**fields):
    """Change switches at run time (``set_config(xpanel=0, tail_next=0)``); returns the previous ``DaConfig``.  Takes effect for the calls
    that follow: a denoiser keeps the folds it was created with, a captured loop graph is recorded again under the new switches."""
    old = config()
    new = DaConfig.from_buffer_copy(old)
    for k, v in fields.items():
        if k == "struct_bytes" or k not in dict(DaConfig._fields_):
            raise DaError(f"da_config has no field {k!r}")
        setattr(new, k, int(v))
    check(lib().da_config_set(C.byref(new)))
    return old


def experiments_build():
    """True when the loaded library is the EXPERIMENTS build (DA_EXPERIMENTS=1 python __graft_entry__.py; DA_LIB_PATH=.../lib_exp/...)."""
    return bool(lib().da_build_flags() & 1)


def check(rc):
    if rc != 0:
        raise DaError(f"libdiffassemble_hip error {rc}: {lib().da_last_error().decode(errors='replace')}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
