"""ctypes binding of libcris_hip.so (include/cris_hip.h).

`import torch` happens first so the library binds to torch's bundled libamdhip64 (one HIP runtime in
the process).  The product path FAILS LOUDLY when the library is missing or was
not built - there is no CPU / eager fallback anywhere in cris.pytorch_amd.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcris_hip.so")
# A/B builds only (tools/build_variants.sh: the same sources compiled with extra -D switches into csrc/variants/libcris_hip_<tag>.so,
# cross-compiled in the container so that one GPU call can compare several builds): CRIS_LIB_VARIANT=<tag> loads that file instead.
# The product, the tests and the driver's bench never set it.
if os.environ.get("CRIS_LIB_VARIANT"):
    LIB_PATH = os.path.join(_HERE, "csrc", "variants", "libcris_hip_%s.so" % os.environ["CRIS_LIB_VARIANT"])

P, I, L, F, U = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint32


class ConvGemmParams(C.Structure):
    _fields_ = [("A", P), ("Wt", P), ("bias", P), ("resid", P), ("out", P), ("outT", P), ("colsum", P), ("colsq", P),
                ("T_sec_stride", L),
                ("lda", I), ("a_coff", I),
                ("Bn", I), ("H", I), ("W", I), ("C", I),
                ("OH", I), ("OW", I), ("KH", I), ("KW", I), ("stride", I), ("pad", I),
                ("ldb", I),
                ("M", I), ("N", I), ("K", I),
                ("act", I),
                ("ldr", I), ("r_coff", I), ("resid_f32", I),
                ("ldc", I), ("c_coff", I), ("out_f32", I),
                ("T_L", I), ("T_Lpad", I), ("T_E", I),
                ("drop_p", F), ("drop_thresh", U), ("drop_seed", U), ("drop_stream", U),
                ("drop_seed_dev", P),
                ("ws", P),
                ("bnr_y", P), ("bnr_mean", P), ("bnr_invstd", P), ("bnr_scale", P), ("bnr_shift", P),
                ("bnr_ldy", I), ("bnr_coff", I), ("stat_ld", I), ("pad_", I)]


GEMM_GROUP_MAX = 12


class ConvGemmGroup(C.Structure):
    _fields_ = [("n", I), ("block_start", I * (GEMM_GROUP_MAX + 1)), ("prob", ConvGemmParams * GEMM_GROUP_MAX)]


class WgradParams(C.Structure):
    _fields_ = [("dY", P), ("X", P), ("dW", P),
                ("ldy", I), ("y_coff", I), ("N_ld", I),
                ("ldx", I), ("x_coff", I),
                ("Bn", I), ("H", I), ("W", I), ("C", I),
                ("OH", I), ("OW", I), ("KH", I), ("KW", I), ("stride", I), ("pad", I),
                ("M", I), ("N", I), ("K", I),
                ("ldw", I), ("splits", I), ("tile", I), ("defer_reduce", I),
                ("dbias", P), ("ws", P)]


WGRAD_GROUP_MAX = 24


class WgradGroup(C.Structure):
    _fields_ = [("n", I), ("block_start", I * (WGRAD_GROUP_MAX + 1)), ("prob", WgradParams * WGRAD_GROUP_MAX)]


class PackDesc(C.Structure):
    _fields_ = [("src", P), ("dstF", P), ("dstD", P), ("row_scale", P),
                ("N", I), ("Cin", I), ("taps", I), ("Cpad", I), ("Npad", I), ("src_transposed", I),
                ("block_start", I), ("ldF", I), ("ldD", I), ("pad_", I)]


class BnApplyParams(C.Structure):
    _fields_ = [("y", P), ("ldy", I), ("y_coff", I),
                ("scale", P), ("shift", P),
                ("y2", P), ("ldy2", I), ("y2_coff", I),
                ("scale2", P), ("shift2", P),
                ("ident", P), ("ldi", I), ("i_coff", I),
                ("mul", P),
                ("z", P), ("ldz", I), ("z_coff", I),
                ("Bn", I), ("H", I), ("W", I), ("C", I),
                ("relu", I), ("pool", I)]


class BnBwdParams(C.Structure):
    _fields_ = [("dz", P), ("lddz", I), ("dz_coff", I),
                ("z", P), ("ldz", I), ("z_coff", I),
                ("y", P), ("ldy", I), ("y_coff", I),
                ("scale", P), ("shift", P), ("mean", P), ("invstd", P),
                ("y2", P), ("ldy2", I), ("y2_coff", I),
                ("mean2", P), ("invstd2", P), ("scale2", P),
                ("mul", P),
                ("sums", P), ("part", P),
                ("dmul", P),
                ("dy", P), ("lddy", I), ("dy_coff", I),
                ("dy2", P), ("lddy2", I), ("dy2_coff", I),
                ("dident", P), ("lddi", I), ("di_coff", I),
                ("dident_accum", I),
                ("Bn", I), ("H", I), ("W", I), ("C", I),
                ("relu", I), ("pool", I),
                ("count", F)]


class LnFwdParams(C.Structure):
    _fields_ = [("x", P), ("x_f32", I), ("ldx", I),
                ("gamma", P), ("beta", P),
                ("pos", P), ("pos_rows", I),
                ("resid", P),
                ("y", P), ("ypos", P), ("out_f32", P),
                ("mean", P), ("rstd", P),
                ("rows", I), ("C", I),
                ("in_relu", I),
                ("in_drop_p", F), ("in_thresh", U), ("in_seed", U), ("in_stream", U),
                ("out_drop_p", F), ("out_thresh", U), ("out_seed", U), ("out_stream", U),
                ("eps", F),
                ("seed_dev", P)]


class LnBwdParams(C.Structure):
    _fields_ = [("x", P), ("x_f32", I), ("ldx", I),
                ("gamma", P),
                ("mean", P), ("rstd", P),
                ("dy", P), ("dypos", P), ("dout_f32", P),
                ("part", P),
                ("dx", P), ("dx_f32", I), ("dx_accum", I),
                ("rows", I), ("C", I),
                ("in_relu", I),
                ("in_drop_p", F), ("in_thresh", U), ("in_seed", U), ("in_stream", U),
                ("out_drop_p", F), ("out_thresh", U), ("out_seed", U), ("out_stream", U),
                ("seed_dev", P)]


class AttnParams(C.Structure):
    _fields_ = [("Q", P), ("ldq", I),
                ("K", P), ("ldk", I),
                ("V", P), ("ldv", I),
                ("Vt", P), ("Kt", P), ("Qt", P), ("Lk_pad", I), ("Lq_pad", I),
                ("key_tokens", P),
                ("O", P), ("ldo", I),
                ("lse", P),
                ("dO", P), ("lddo", I), ("dOt", P),
                ("delta", P),
                ("dQ", P), ("lddq", I),
                ("dK", P), ("lddk", I),
                ("dV", P), ("lddv", I),
                ("B", I), ("Hn", I), ("Lq", I), ("Lk", I),
                ("causal", I),
                ("scale", F),
                ("drop_p", F), ("drop_thresh", U), ("drop_seed", U), ("drop_stream", U),
                ("drop_seed_dev", P)]


SUM_GROUP_MAX = 64


class SumEntry(C.Structure):
    _fields_ = [("part", P), ("out", P), ("nparts", I), ("ncol", I), ("ld", I), ("pad_", I)]


class SumGroup(C.Structure):
    _fields_ = [("n", I), ("block_start", I * (SUM_GROUP_MAX + 1)), ("e", SumEntry * SUM_GROUP_MAX)]


class AdamDesc(C.Structure):
    _fields_ = [("p", P), ("g", P), ("m", P), ("v", P),
                ("n", L),
                ("lr", F), ("pad_", F),
                ("block_start", I), ("taps", I),
                ("cin", I), ("cpad", I),
                ("dstF", P), ("dstD", P),
                ("N", I), ("npad", I),
                ("transposed", I), ("ldF", I),
                ("row_live", P), ("row_len", I), ("ldD", I)]


ZERO_RANGES_MAX = 16


class ZeroRange(C.Structure):
    _fields_ = [("p", P), ("nbytes", C.c_size_t)]


class ZeroRanges(C.Structure):
    _fields_ = [("n", I), ("pad_", I), ("r", ZeroRange * ZERO_RANGES_MAX)]


class SampleDesc(C.Structure):
    _fields_ = [("img", P), ("mask", P), ("H", I), ("W", I), ("inv", C.c_double * 6)]


class JpegInfo(C.Structure):
    _fields_ = [("width", I), ("height", I), ("ncomp", I), ("hmax", I), ("vmax", I), ("mcus_x", I), ("mcus_y", I),
                ("restart_interval", I),
                ("comp_h", I * 3), ("comp_v", I * 3), ("blocks_w", I * 3), ("blocks_h", I * 3), ("down_w", I * 3), ("down_h", I * 3),
                ("quant", (C.c_ushort * 64) * 3),
                ("coef_offset", L * 3), ("coef_count", L), ("plane_offset", L * 3), ("plane_bytes", L), ("scan_offset", L),
                ("total_blocks", I), ("multiscan", I)]


class JpegImage(C.Structure):
    _fields_ = [("coef", P), ("planes", P), ("rgb", P), ("info", JpegInfo)]


class P2PParams(C.Structure):
    _fields_ = [("data", P), ("boxes", P), ("gen_dev", P), ("err", P),
                ("n", I), ("rank", I), ("world", I),
                ("slot", I), ("slots", I),
                ("max_floats", I),
                ("gen_host", I),
                ("spin_limit", I)]


class P2PLink(C.Structure):
    _fields_ = [("boxes", P), ("err", P), ("gen_dev", P),
                ("gen_host", I), ("rank", I), ("world", I),
                ("slot", I), ("slots", I), ("max_floats", I), ("spin_limit", I)]


class P2PArenaParams(C.Structure):
    _fields_ = [("arenas", P), ("lo", C.c_long), ("n", C.c_long), ("link", P2PLink), ("blocks", I)]


STRUCTS = {
    "cris_conv_gemm_params": ConvGemmParams, "cris_conv_gemm_group": ConvGemmGroup, "cris_wgrad_params": WgradParams, "cris_wgrad_group": WgradGroup, "cris_pack_desc": PackDesc,
    "cris_bn_apply_params": BnApplyParams, "cris_bn_bwd_params": BnBwdParams, "cris_ln_fwd_params": LnFwdParams,
    "cris_ln_bwd_params": LnBwdParams, "cris_sum_entry": SumEntry, "cris_sum_group": SumGroup, "cris_attn_params": AttnParams, "cris_adam_desc": AdamDesc, "cris_p2p_params": P2PParams, "cris_p2p_link": P2PLink, "cris_p2p_arena_params": P2PArenaParams, "cris_zero_ranges": ZeroRanges,
    "cris_sample_desc": SampleDesc, "cris_jpeg_info": JpegInfo, "cris_jpeg_image": JpegImage,
}

# name -> (restype, argtypes); struct launchers take (struct*, stream)
_SIGS = {
    "cris_last_error": (C.c_char_p, []),
    "cris_abi_version": (I, []),
    "cris_sizeof": (I, [C.c_char_p]),
    "cris_echo_conv_gemm": (L, [P]),
    "cris_conv_gemm": (I, [P, P]),
    "cris_conv_wgrad": (I, [P, P]),
    "cris_conv_wgrad_group": (I, [P, P]),
    "cris_conv_wgrad_tile": (I, [P]),
    "cris_wgrad_reduce": (I, [P, P]),
    "cris_wgrad_reduce_group": (I, [P, P]),
    "cris_wgrad_ws_floats": (L, [I, I, I, I]),
    "cris_pack_weights": (I, [P, I, I, P]),
    "cris_pack_blocks": (I, [P]),
    "cris_pack_block_elems": (I, []),
    "cris_conv_gemm_stat_rows": (I, [P]),
    "cris_conv_gemm_variant": (I, [P, I, P]),
    "cris_conv_gemm_variant_stat_rows": (I, [P, I]),
    "cris_conv_gemm_ws_floats": (L, [P, I]),
    "cris_conv_gemm_num_variants": (I, []),
    "cris_conv_gemm_plan": (I, [P, I, P]),
    "cris_conv_gemm_group_launch": (I, [P, I, P]),
    "cris_conv_gemm_variant_name": (C.c_char_p, [I]),
    "cris_bn_partials_rows": (I, [I]),
    "cris_bn_finalize": (I, [P, P, I, I, F, F, P, P, P, P, F, F, I, P, P, P, P, P, P, P]),
    "cris_bn_sync_pack": (I, [P, P, P, F, I, P]),
    "cris_bn_sync_unpack": (I, [P, P, F, I, P]),
    "cris_colstats_bf16": (I, [P, I, I, I, I, I, P, P, P]),
    "cris_bn_eval_coeffs": (I, [P, P, P, P, F, I, P, P, P]),
    "cris_bn_apply": (I, [P, P]),
    "cris_bn_bwd_reduce": (I, [P, P]),
    "cris_bn_bwd_apply": (I, [P, P]),
    "cris_bn_bwd_ws_floats": (L, [P]),
    "cris_ln_bwd_parts": (I, [I]),
    "cris_sum_tables": (I, [P, P]),
    "cris_ln_fwd": (I, [P, P]),
    "cris_ln_bwd": (I, [P, P]),
    "cris_attn_fwd": (I, [P, P]),
    "cris_attn_bwd_dq": (I, [P, P]),
    "cris_attn_bwd_dkv": (I, [P, P]),
    "cris_stem_im2col": (I, [P, I, I, I, P, P]),
    "cris_avgpool2_fwd": (I, [P, I, I, I, I, I, I, P, I, I, P]),
    "cris_avgpool2_bwd": (I, [P, I, I, I, I, I, I, P, I, I, I, P]),
    "cris_upsample2_fwd": (I, [P, I, I, I, I, I, I, P, I, I, P]),
    "cris_upsample2_bwd": (I, [P, I, I, I, I, I, I, P, I, I, I, P]),
    "cris_fill_coords": (I, [P, I, I, I, I, I, I, P]),
    "cris_add_bf16": (I, [P, I, I, P, I, I, P, I, I, I, I, P]),
    "cris_add_rowtable": (I, [P, I, P, I, P, I, I, I, P]),
    "cris_cast_f32_bf16": (I, [P, P, L, P]),
    "cris_cast_bf16_f32": (I, [P, P, L, I, P]),
    "cris_cast_f32_bf16_drop": (I, [P, P, L, F, U, U, U, P, P]),
    "cris_step_advance": (I, [P, P, P, P]),
    "cris_axpy_f32": (I, [P, P, F, L, P]),
    "cris_quickgelu_fwd": (I, [P, P, L, P]),
    "cris_quickgelu_bwd": (I, [P, P, P, L, P]),
    "cris_embed_fwd": (I, [P, P, P, I, I, I, P, P]),
    "cris_embed_bwd": (I, [P, P, I, I, I, P, P, P, P]),
    "cris_eot_gather": (I, [P, P, I, I, I, P, P, P]),
    "cris_eot_scatter_add": (I, [P, P, I, I, I, P, P]),
    "cris_eot_gather_ln_f32": (I, [P, P, P, P, P, P, I, I, I, P, P, P]),
    "cris_eot_scatter_add_f32": (I, [P, P, I, I, I, P, P]),
    "cris_linear_f32_small": (I, [P, I, P, I, I, P, I, I, I, P, I, I, P]),
    "cris_outer_sum_f32_small": (I, [P, I, P, I, I, I, I, P, I, P, P]),
    "cris_colstats_f32_small": (I, [P, I, I, I, P, P, P]),
    "cris_bn_relu_f32_small": (I, [P, I, P, P, I, I, P, I, P]),
    "cris_bn_relu_bwd_reduce_f32_small": (I, [P, I, P, I, P, P, P, P, I, I, P, P]),
    "cris_bn_relu_bwd_apply_f32_small": (I, [P, I, P, I, P, P, P, P, P, F, I, I, P, I, P]),
    "cris_posresize_fwd": (I, [P, P, I, I, I, P, P]),
    "cris_posresize_bwd": (I, [P, P, I, I, I, P, P]),
    "cris_batch_rowsum": (I, [P, I, I, I, I, P, P]),
    "cris_dynconv_fwd": (I, [P, I, I, I, I, P, I, P, P]),
    "cris_dynconv_bwd": (I, [P, P, I, I, I, I, P, I, P, P, P, P]),
    "cris_dynconv_bwd_ws_floats": (L, [I, I, I, I]),
    "cris_mask_resize_nearest": (I, [P, I, I, I, I, I, P, P]),
    "cris_bce_fwd": (I, [P, P, L, P, P, P]),
    "cris_bce_ws_floats": (I, []),
    "cris_bce_bwd": (I, [P, P, L, P, P, P]),
    "cris_train_metric": (I, [P, P, I, I, F, F, P, P]),
    "cris_sigmoid_bicubic_up": (I, [P, I, I, I, I, I, P, P]),
    "cris_warp_affine_cubic": (I, [P, I, I, P, I, I, F, P, P]),
    "cris_threshold_iou": (I, [P, P, L, F, P, P]),
    "cris_preprocess_batch": (I, [P, I, I, I, P, P, P, P, P, P, P, P]),
    "cris_invert_affine": (I, [P, P]),
    "cris_remap_tables_u8": (I, [P, P]),
    "cris_zero_bytes": (I, [P, C.c_size_t, P]),
    "cris_zero_many": (I, [P, P]),
    "cris_adam_step": (I, [P, I, I, F, F, F, F, F, F, F, P, I, P]),
    "cris_adam_step_amp": (I, [P, I, I, F, F, F, F, F, F, F, P, P, P, I, P]),
    "cris_counter_advance_unless": (I, [P, P, P]),
    "cris_adam_blocks": (I, [P]),
    "cris_adam_block_elems": (I, []),
    "cris_unpack_grads": (I, [P, I, I, P]),
    "cris_p2p_mailbox_bytes": (C.c_size_t, [I, I, I]),
    "cris_p2p_alloc": (I, [C.c_size_t, P]),
    "cris_p2p_free": (I, [P]),
    "cris_p2p_export": (I, [P, P]),
    "cris_p2p_import": (I, [P, P]),
    "cris_p2p_close": (I, [P]),
    "cris_p2p_allreduce_sum": (I, [P, P]),
    "cris_p2p_ll_allreduce_sum": (I, [P, P, I, P]),
    "cris_p2p_arena_allreduce": (I, [P, P]),
    "cris_bn_finalize_sync": (I, [P, P, I, I, F, F, P, P, P, P, F, F, I, P, P, P, P, P, P]),
    "cris_bn_bwd_reduce_sync": (I, [P, P, P, P]),
    "cris_bn_bwd_sum": (I, [P, I, P]),
    "cris_bn_bwd_sum_sync": (I, [P, I, P, P, P]),
    "cris_jpeg_read_header": (I, [P, C.c_size_t, P]),
    "cris_jpeg_decode_coefficients": (I, [P, C.c_size_t, P, P]),
    "cris_jpeg_decode_coefficients_batch": (I, [I, P, P, P, P, I]),
    "cris_jpeg_reconstruct": (I, [P, I, I, L, P]),
    "cris_png_gray8_size": (I, [P, C.c_size_t, P, P]),
    "cris_png_decode_gray8": (I, [P, C.c_size_t, P, I, I]),
    "cris_comm_rccl_path": (C.c_char_p, []),
    "cris_comm_unique_id": (I, [P]),
    "cris_comm_init": (I, [I, I, P, P]),
    "cris_comm_destroy": (I, [P]),
    "cris_comm_rank": (I, [P]),
    "cris_comm_world": (I, [P]),
    "cris_comm_syncbn_exchange": (I, [P, P, C.c_size_t, P]),
    "cris_comm_allreduce_bucket": (I, [P, P, C.c_size_t, P]),
    "cris_comm_wait": (I, [P, P]),
    "cris_comm_broadcast": (I, [P, P, C.c_size_t, I, P]),
}
EXPORTS = sorted(_SIGS)

_lib = None


class HipLibraryError(RuntimeError):
    pass


ABI_VERSION = 4      # == CRIS_ABI_VERSION of include/cris_hip.h (tests/test_abi.py compares the two)


def load():
    """Load libcris_hip.so (building is __graft_entry__.build()'s job).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "libcris_hip.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the CRIS HIP path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.cris_abi_version.restype = C.c_int
    if lib.cris_abi_version() != ABI_VERSION:
        raise HipLibraryError("ABI mismatch: %s reports cris_abi_version() = %d, this binding was written against %d (include/cris_hip.h "
                              "CRIS_ABI_VERSION) - a stale build; run `python -c 'import __graft_entry__ as g; g.build()'`"
                              % (LIB_PATH, lib.cris_abi_version(), ABI_VERSION))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for cname, st in STRUCTS.items():
        n = lib.cris_sizeof(cname.encode())
        if n != C.sizeof(st):
            raise HipLibraryError("ABI mismatch: sizeof(%s) C=%d ctypes=%d" % (cname, n, C.sizeof(st)))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().cris_last_error()
        raise HipLibraryError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


class CommandList:
    """Host-side command list of one step: every libcris_hip launch (function + ctypes arguments, stream included) and
    every torch-level op (stream wait, collective) in issue order.  Replaying it costs a few microseconds of Python per
    entry instead of the schedule's Python (closures, struct filling, allocations) - the launch mode used where a HIP graph
    is not (collectives inside the step).  All buffers the commands point to must stay allocated (trainer.py: MemPool)."""

    def __init__(self):
        self.cmds = []

    def replay(self):
        for fn, args, name in self.cmds:
            if args is None:
                fn()
            else:
                rc = fn(*args)
                if rc != 0:
                    check(rc, name)


RECORDER = None          # a CommandList while a step is being recorded


def call(name, *args):
    """Invoke an exported launcher and raise on a non-zero return code."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)
    if RECORDER is not None:
        RECORDER.cmds.append((fn, args, name))


def ptr(t):
    """Device (or host) address of a tensor, None -> NULL."""
    return None if t is None else t.data_ptr()


def dropout_threshold(p: float) -> int:
    """keep iff hash >= threshold (same rule as oracle/dropout_hash.py threshold())."""
    return min(int(p * 4294967296.0), 0xFFFFFFFF) if p > 0 else 0
