"""ctypes binding of libcst_hip.so (C ABI: include/cst.h).

The product path fails LOUDLY when the library is missing or a call returns an error: there is
no CPU or PyTorch fallback anywhere behind these wrappers."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcst_hip.so")
ABI_VERSION = 6  # include/cst.h: CST_ABI_VERSION

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
BIAS_NONE, BIAS_COL, BIAS_ROW = 0, 1, 2
K_GEMM, K_ATTN_FWD, K_ATTN_BWD, K_LAYERNORM, K_CONV0, K_ELEMENTWISE, K_LOSS, K_OPTIM = range(8)
KERNEL_CLASSES = ["gemm", "attn_fwd", "attn_bwd", "layernorm", "conv0", "elementwise", "loss", "optim"]

c_i64, c_int, c_f, c_p = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", c_int), ("c_dtype", c_int), ("a_kmajor", c_int), ("b_kmajor", c_int),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("A", c_p), ("lda", c_i64), ("a_seg", c_i64), ("a_seg_stride", c_i64),
        ("B", c_p), ("ldb", c_i64), ("b_seg", c_i64), ("b_seg_stride", c_i64),
        ("C", c_p), ("ldc", c_i64),
        ("bias", c_p), ("bias_mode", c_int), ("sbias0", c_i64), ("sbias1", c_i64),
        ("act", c_int),
        ("aux_out", c_p), ("ld_aux_out", c_i64),
        ("dact", c_int),
        ("aux_in", c_p), ("ld_aux_in", c_i64),
        ("resid", c_p), ("ld_resid", c_i64),
        ("alpha", c_f),
        ("drop_p", c_f), ("drop_key", ctypes.c_uint32),
        ("batch0", c_i64), ("batch1", c_i64),
        ("sa0", c_i64), ("sa1", c_i64), ("sb0", c_i64), ("sb1", c_i64), ("sc0", c_i64), ("sc1", c_i64),
        ("split_k", c_int),
        ("workspace", c_p), ("workspace_bytes", c_i64),
        ("k_live", c_p), ("k_epoch", ctypes.c_uint32),
        ("m_live", c_p), ("m_epoch", ctypes.c_uint32),
        ("k_len", c_p),
        ("m_len", c_p),
        ("colsum", c_p),
        ("defer_reduce", c_int),
    ]


class ReduceItem(ctypes.Structure):
    _fields_ = [("src", c_p), ("dst", c_p), ("stride", c_i64), ("L", c_i64), ("P", ctypes.c_int32), ("dst_dtype", ctypes.c_int32),
                ("block0", ctypes.c_int32), ("order", ctypes.c_int32)]


class AttnDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", c_int),
        ("B", c_i64), ("H", c_i64), ("Tq", c_i64), ("Tk", c_i64), ("D", c_i64),
        ("Q", c_p), ("q_sb", c_i64), ("q_sh", c_i64), ("q_st", c_i64),
        ("K", c_p), ("k_sb", c_i64), ("k_sh", c_i64), ("k_st", c_i64),
        ("V", c_p), ("v_sb", c_i64), ("v_sh", c_i64), ("v_st", c_i64),
        ("O", c_p), ("o_sb", c_i64), ("o_sh", c_i64), ("o_st", c_i64),
        ("lse", c_p),
        ("key_padding_mask", c_p), ("kpm_stride", c_i64),
        ("causal", c_int), ("scale", c_f),
        ("drop_p", c_f), ("drop_key", ctypes.c_uint32),
        ("dO", c_p), ("do_sb", c_i64), ("do_sh", c_i64), ("do_st", c_i64),
        ("dQ", c_p), ("dq_sb", c_i64), ("dq_sh", c_i64), ("dq_st", c_i64),
        ("dK", c_p), ("dk_sb", c_i64), ("dk_sh", c_i64), ("dk_st", c_i64),
        ("dV", c_p), ("dv_sb", c_i64), ("dv_sh", c_i64), ("dv_st", c_i64),
        ("delta", c_p),
        ("kv_len", c_p),
        ("q_flags", c_p),
        ("seq_offsets", c_p),
        ("kpm_bits", c_p),
        ("bwd_ws", c_p),
    ]


class BeamDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", c_int),
        ("bsz", c_i64), ("beam", c_i64), ("vocab", c_i64), ("max_len", c_i64),
        ("pad", c_i64), ("unk", c_i64), ("eos", c_i64), ("min_len", c_i64),
        ("unk_penalty", c_f), ("len_penalty", c_f), ("temperature", c_f),
        ("normalize_scores", c_int),
        ("logits", c_p), ("ld_logits", c_i64),
        ("step", c_p),
        ("tokens", c_p), ("scores", c_p), ("anc", c_p),
        ("cands_to_ignore", c_p), ("finished", c_p), ("nfinal", c_p), ("num_remaining", c_p),
        ("fin_tokens", c_p), ("fin_pos", c_p), ("fin_score", c_p), ("fin_len", c_p),
        ("workspace", c_p),
    ]


# every symbol include/cst.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("cst_last_error", ctypes.c_char_p, []),
    ("cst_version", c_int, []),
    ("cst_device_arch_ok", c_int, []),
    ("cst_prof_enable", None, [c_int]),
    ("cst_prof_query", c_i64, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    ("cst_layernorm_fwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_f, c_int, c_p]),
    ("cst_layernorm_bwd_workspace", c_i64, [c_i64, c_i64]),
    ("cst_layernorm_bwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_int, c_p]),
    ("cst_layernorm_bwd_tiles", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_int, c_p, ctypes.c_uint32, c_p]),
    ("cst_gemm_workspace", c_i64, [ctypes.POINTER(GemmDesc)]),
    ("cst_gemm_colsum_is_fused", c_int, [ctypes.POINTER(GemmDesc)]),
    ("cst_gemm_splits", c_int, [ctypes.POINTER(GemmDesc)]),
    ("cst_reduce_multi", c_int, [ctypes.POINTER(ReduceItem), c_int, c_p]),
    ("cst_gemm", c_int, [ctypes.POINTER(GemmDesc), c_p]),
    ("cst_prof_dump", c_i64, [c_int, ctypes.c_char_p, c_i64]),
    ("cst_gemm_reserve_cus", c_int, [c_int]),
    ("cst_transpose2d", c_int, [c_p, c_p, c_i64, c_i64, c_int, c_p]),
    ("cst_transpose2d_multi", c_int, [c_p, c_int, c_i64, c_int, c_p]),
    ("cst_weight_norm_workspace", c_i64, [c_i64, c_i64]),
    ("cst_weight_norm_fwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_p]),
    ("cst_weight_norm_bwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_p]),
    ("cst_attn_bwd_workspace", c_i64, [ctypes.POINTER(AttnDesc)]),
    ("cst_attn_fwd", c_int, [ctypes.POINTER(AttnDesc), c_p]),
    ("cst_attn_bwd", c_int, [ctypes.POINTER(AttnDesc), c_p]),
    ("cst_conv0_fwd_workspace", c_i64, [c_i64, c_i64, c_int, c_int]),
    ("cst_conv0_gn_gelu_fwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_int, c_int, c_f, c_int, c_p]),
    ("cst_conv0_bwd_workspace", c_i64, [c_i64, c_i64, c_i64, c_int, c_int]),
    ("cst_conv0_gn_gelu_bwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_p]),
    ("cst_glu_fwd", c_int, [c_p, c_p, c_i64, c_i64, c_int, c_p]),
    ("cst_glu_bwd", c_int, [c_p, c_p, c_p, c_i64, c_i64, c_int, c_p]),
    ("cst_act_bwd", c_int, [c_p, c_p, c_p, c_i64, c_int, c_int, c_p]),
    ("cst_act_fwd", c_int, [c_p, c_p, c_i64, c_int, c_int, c_p]),
    ("cst_colsum", c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_int, c_p]),
    ("cst_colsum_workspace", c_i64, [c_i64, c_i64]),
    ("cst_colsum_typed", c_int, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_int, c_int, c_p]),
    ("cst_colsum_typed_live", c_int, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_int, c_int, c_p, ctypes.c_uint32, c_p]),
    ("cst_col2im1d", c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_p]),
    ("cst_mask_rows", c_int, [c_p, c_p, c_p, c_i64, c_i64, c_int, c_p]),
    ("cst_dropout", c_int, [c_p, c_p, c_i64, c_f, ctypes.c_uint32, c_int, c_p]),
    ("cst_rows_pack", c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_int, c_int, c_p]),
    ("cst_rows_unpack", c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_int, c_int, c_p]),
    ("cst_dec_linear", c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_p, c_i64, c_int, c_p]),
    ("cst_dec_ln_linear", c_int, [c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_p, c_i64, c_int, c_p]),
    ("cst_dropout_colsum", c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_int, c_f, ctypes.c_uint32, c_p, ctypes.c_uint32, c_p]),
    ("cst_conv_row_limits", c_int, [c_p, c_p, c_p, c_int, c_i64, c_p, c_i64, c_int, c_p]),
    ("cst_dropout_scale", c_int, [c_p, c_p, c_i64, c_f, c_f, ctypes.c_uint32, c_int, c_p]),
    ("cst_embed_pos_fwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_f, c_i64, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, ctypes.c_uint32, c_int, c_p]),
    ("cst_embed_bwd", c_int, [c_p, c_p, c_p, c_f, c_i64, c_i64, c_i64, c_i64, c_f, ctypes.c_uint32, c_int, c_int, c_p]),
    ("cst_ls_ce_fwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_f, c_i64, c_int, c_p]),
    ("cst_ls_ce_bwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_f, c_i64, c_int, c_p]),
    ("cst_contrastive_fwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
    ("cst_contrastive_bwd", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
    ("cst_sumsq_workspace", c_i64, []),
    ("cst_sumsq", c_int, [c_p, c_i64, c_p, c_p, c_int, c_p]),
    ("cst_adam_step", c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_p, c_int, c_int, c_p]),
    ("cst_beam_workspace", c_i64, [c_i64, c_i64]),
    ("cst_beam_init", c_int, [ctypes.POINTER(BeamDesc), c_p]),
    ("cst_beam_step", c_int, [ctypes.POINTER(BeamDesc), c_p]),
    ("cst_dec_embed", c_int, [c_p, c_p, c_p, c_p, c_f, c_i64, c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    ("cst_dec_self_attn", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
    ("cst_batch_by_size", c_i64, [c_p, c_i64, c_i64, c_i64, ctypes.c_int32, c_p]),
    ("cst_wav_info", c_int, [ctypes.c_char_p, c_p, c_p, c_p, c_p]),
    ("cst_wav_read_f32", c_i64, [ctypes.c_char_p, c_i64, c_i64, c_p, c_i64]),
    ("cst_dec_cross_attn", c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
    ("cst_dec_ln_q_cross_attn", c_int, [c_p, c_i64, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
]

_lib = None


def load():
    """Load libcst_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "chimera-st_amd: %s is missing — the HIP extension is the product path and there is no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C chimera-st_amd/csrc`." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    built = lib.cst_version()
    if built != ABI_VERSION:
        raise RuntimeError("chimera-st_amd: %s was built for ABI version %d, this package binds version %d (include/cst.h "
                           "CST_ABI_VERSION) — rebuild it with `make -C chimera-st_amd/csrc`." % (LIB_PATH, built, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().cst_last_error()
        raise RuntimeError("libcst_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError("chimera-st_amd kernels compute in float32 or bfloat16 storage, got %s" % t)


_DEV_INDEX = None


def stream_ptr():
    """Raw hipStream_t of torch's CURRENT stream on this process's device (one process per GPU).  torch.cuda.current_stream()
    costs ~20 us of Python per call (device-index plumbing); the private raw-stream getter is the same lookup in ~0.3 us —
    at ~850 C-ABI launches per update that is 17 ms of host time, enough to starve the GPU in the small-kernel decoder phases."""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(_DEV_INDEX))


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("chimera-st_amd: tensor is not on the GPU — the HIP path has no CPU fallback")
    return ctypes.c_void_p(t.data_ptr())


def prof_enable(on: bool):
    load().cst_prof_enable(1 if on else 0)


def prof_query():
    """{class: dict(launches, ms, flops, bytes)} of everything recorded since prof_enable(True)."""
    lib = load()
    out = {}
    for i, name in enumerate(KERNEL_CLASSES):
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = lib.cst_prof_query(i, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
        out[name] = dict(launches=int(n), ms=ms.value, flops=fl.value, bytes=by.value)
    return out


def prof_dump(cls):
    """Per-launch records of one kernel class of the profiling table: [(ms, flops, bytes, tag)] in launch order."""
    lib = load()
    n = lib.cst_prof_dump(cls, None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    lib.cst_prof_dump(cls, buf, len(buf))
    out = []
    for line in buf.value.decode().splitlines():
        ms, fl, by, *tag = line.split(" ", 3)
        out.append((float(ms), float(fl), float(by), tag[0] if tag else ""))
    return out
