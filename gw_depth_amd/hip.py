"""ctypes binding of libgwdepth_hip.so (C ABI: include/gwdepth.h).

There is no CPU fallback: importing works anywhere (so host-side logic can be unit-tested), but
every compute entry point raises HipUnavailable unless the shared library is present AND the
tensors live on a HIP device.  Tests may install a fake device library with `set_library()` to
exercise the host logic on CPU; the product never does.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GWD_LIB") or os.path.join(_HERE, "libgwdepth_hip.so")     # GWD_LIB: another build of the same ABI (same-box A/B of kernel variants)

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3, 4
WS_INORM_GELU, WS_RESAMPLE_BWD, WS_EVAL, WS_PLANE = 0, 1, 2, 3          # gwd_query_workspace ops
GATHER_CONV, GATHER_TRANSPOSED, GATHER_UPSAMPLED = 0, 1, 2
RESAMPLE_BILINEAR_AC, RESAMPLE_NEAREST = 0, 1

ENTRY_POINTS = [
    "gwd_version", "gwd_arch", "gwd_conv_forward", "gwd_conv_wgrad", "gwd_weight_prep", "gwd_act_backward",
    "gwd_colsum", "gwd_layernorm_forward", "gwd_layernorm_backward", "gwd_softmax_forward",
    "gwd_softmax_backward", "gwd_silog_sums", "gwd_silog_backward", "gwd_seg_ce_sum", "gwd_seg_ce_backward",
    "gwd_sqnorm", "gwd_adamw_step", "gwd_resample_forward", "gwd_resample_backward", "gwd_avgpool_forward",
    "gwd_avgpool_backward", "gwd_winattn_forward", "gwd_winattn_backward", "gwd_tokattn_forward",
    "gwd_tokattn_backward", "gwd_tokattn_pair_forward", "gwd_tokattn_pair_backward", "gwd_upsample_taps_collapse", "gwd_upsample_taps_fold", "gwd_certain_sample", "gwd_lsap", "gwd_window_map", "gwd_window_map_multi",
    "gwd_inorm_gelu_forward", "gwd_inorm_gelu_backward", "gwd_weight_prep_batch",
    "gwd_point_sample_forward", "gwd_point_sample_backward", "gwd_act_backward_colsum", "gwd_resample_backward_sep",
    "gwd_softmax_masked_forward", "gwd_softmax_scaled_backward", "gwd_query_workspace", "gwd_eval_accumulate", "gwd_colsum_batch", "gwd_conv_wgrad_batch",
    "gwd_plane_loss_forward", "gwd_plane_loss_backward", "gwd_collate",
    "gwd_anchor_depth_forward", "gwd_anchor_depth_backward", "gwd_mha_flash_forward", "gwd_mha_flash_backward",
    "gwd_ref_scores_forward", "gwd_ref_scores_backward", "gwd_ref_mix_forward", "gwd_ref_mix_backward", "gwd_unpad_add_batch", "gwd_stem_pack", "gwd_stem_forward", "gwd_pos_sine", "gwd_silog_finalize", "gwd_psp_pool_forward", "gwd_psp_pool_backward",
    "gwd_match_cost", "gwd_set_losses_forward", "gwd_set_losses_backward", "gwd_resample_u8_pass", "gwd_gather2d", "gwd_point_sample_backward_gather", "gwd_point_sample_framed_forward", "gwd_point_sample_framed_backward", "gwd_stride_place", "gwd_color_adjust", "gwd_bmm",
]


class HipUnavailable(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("y", ctypes.c_void_p), ("z", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("residual", ctypes.c_void_p),
                ("zero_page", ctypes.c_void_p), ("mult", ctypes.c_void_p), ("gate", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("B", "Hi", "Wi", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad",
                                              "gather", "Hv", "Wv", "act")] + \
               [("act_scale", ctypes.c_float), ("dtype", ctypes.c_int32), ("gate_act", ctypes.c_int32),
                ("ln_mean", ctypes.c_void_p), ("ln_rstd", ctypes.c_void_p), ("ln_C", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class BmmDesc(ctypes.Structure):
    """gwd_bmm_desc (include/gwdepth.h)."""
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("c", ctypes.c_void_p)] + \
               [(n, ctypes.c_int64) for n in ("a_sb0", "a_sb1", "a_ld", "b_sb0", "b_sb1", "b_ld", "c_sb0", "c_sb1", "c_ld")] + \
               [(n, ctypes.c_int32) for n in ("M", "N", "K", "nb0", "nb1", "a_kmajor", "b_kmajor", "c_is_f32_accumulate", "splits")] + \
               [("alpha", ctypes.c_float), ("dtype", ctypes.c_int32)]


class PrepJob(ctypes.Structure):
    """gwd_prep_job (include/gwdepth.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("row_scale", ctypes.c_void_p), ("w_fwd", ctypes.c_void_p), ("w_dgrad", ctypes.c_void_p),
                ("N", ctypes.c_int32), ("taps", ctypes.c_int32), ("C", ctypes.c_int32), ("block0", ctypes.c_int32),
                ("Np", ctypes.c_int32), ("Cg", ctypes.c_int32), ("Cgp", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class UnpadJob(ctypes.Structure):
    """gwd_unpad_job (include/gwdepth.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("N", ctypes.c_int32), ("taps", ctypes.c_int32), ("G", ctypes.c_int32),
                ("Cg", ctypes.c_int32), ("Cgp", ctypes.c_int32), ("block0", ctypes.c_int32)]


UNPAD_BATCH = 24
STEM_PACKED_ELEMS = 14 * 2 * 64 * 8


def stem_out(n):
    """Output size of conv 7/2/3 followed by max-pool 3/2/1 along one axis."""
    return ((n - 1) // 2 + 1 - 1) // 2 + 1


class ColsumJob(ctypes.Structure):
    """gwd_colsum_job (include/gwdepth.h)."""
    _fields_ = [("g", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_int64), ("C", ctypes.c_int32),
                ("block0", ctypes.c_int32), ("blocks", ctypes.c_int32)]


COLSUM_BATCH = 16
LSAP_MAX_TARGETS = 64     # gwd_lsap: targets per image (MAXT in csrc/lsap.hip)
COLLATE_BATCH = 16


class ImageJob(ctypes.Structure):
    """gwd_image_job (include/gwdepth.h)."""
    _fields_ = [("rgb", ctypes.c_void_p), ("depth_mm", ctypes.c_void_p), ("labels", ctypes.c_void_p),
                ("h", ctypes.c_int32), ("w", ctypes.c_int32)]


class Strided(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("ws", ctypes.c_int64), ("ts", ctypes.c_int64), ("hs", ctypes.c_int64)]


def _strided(t):
    """(windows, tokens, heads, head_dim) tensor or view with unit channel stride."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise ValueError("window-attention operand must be (W, N, heads, hd) with unit last stride")
    return Strided(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


_ZERO_PAGES = {}


def _zero_page(device):
    """256 zero bytes per device: the LDS-DMA convolution pipeline fetches out-of-image taps from here."""
    if device.type != "cuda":
        return None
    z = _ZERO_PAGES.get(device)
    if z is None:
        z = _ZERO_PAGES[device] = torch.zeros(256, dtype=torch.uint8, device=device)
    return ctypes.c_void_p(z.data_ptr())


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("gw_depth_amd kernels take float32 or bfloat16, got %s" % t.dtype)


def _ptr_pitched(t):
    """Address of an operand whose pixel pitch the entry point takes separately (a channel slice of a pixel-major map)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _ptr(t):
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("kernel operand must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


class HipLibrary:
    """Thin typed wrapper: tensors in, raw pointers + sizes + current stream out."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise HipUnavailable(
                "libgwdepth_hip.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        self.lib = ctypes.CDLL(path)
        for name in ENTRY_POINTS:
            if not hasattr(self.lib, name):
                raise HipUnavailable("libgwdepth_hip.so lacks symbol " + name)
        self.lib.gwd_arch.restype = ctypes.c_char_p
        for name in ENTRY_POINTS[2:]:
            getattr(self.lib, name).restype = ctypes.c_int
        L = self.lib
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        L.gwd_conv_forward.argtypes = [ctypes.POINTER(ConvDesc), vp]
        L.gwd_conv_wgrad.argtypes = [ctypes.POINTER(ConvDesc), vp, vp]
        L.gwd_weight_prep.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.gwd_act_backward.argtypes = [vp, vp, vp, vp, i64, i32, i32, f32, i32, vp]
        L.gwd_colsum.argtypes = [vp, vp, i64, i32, i32, vp]
        L.gwd_conv_wgrad_batch.argtypes = [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_void_p), i32, vp]
        L.gwd_plane_loss_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp]
        L.gwd_plane_loss_backward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp]
        L.gwd_collate.argtypes = [ctypes.POINTER(ImageJob), i32, i32, i32, ctypes.POINTER(ctypes.c_float),
                                  ctypes.POINTER(ctypes.c_float), vp, vp, vp, vp, i32, vp]
        L.gwd_mha_flash_forward.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp, vp, i64, vp, i32, i32, i32, i32, f32, i32, vp]
        L.gwd_mha_flash_backward.argtypes = [vp] * 5 + [i64] * 5 + [vp] * 7 + [i64] * 3 + [i32, i32, i32, i32, f32, i32, vp]
        L.gwd_anchor_depth_forward.argtypes = [vp, vp, vp, i32, i64, i32, i32, vp]
        L.gwd_anchor_depth_backward.argtypes = [vp, vp, vp, vp, vp, i32, i64, i32, i32, vp]
        L.gwd_colsum_batch.argtypes = [ctypes.POINTER(ColsumJob), i32, i32, vp]
        L.gwd_layernorm_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
        L.gwd_layernorm_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, i32, vp]
        L.gwd_unpad_add_batch.argtypes = [ctypes.POINTER(UnpadJob), i32, vp]
        L.gwd_silog_finalize.argtypes = [vp, f32, f32, vp, vp]
        L.gwd_resample_u8_pass.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i32, i32, i32, i32, vp]
        L.gwd_gather2d.argtypes = [vp, vp, vp, vp, i32, i32, i64, i32, vp]
        L.gwd_match_cost.argtypes = [vp] * 5 + [i32] * 6 + [f32, f32, vp]
        L.gwd_set_losses_forward.argtypes = [vp] * 9 + [f32] + [vp] * 4 + [i32] * 6 + [vp]
        L.gwd_set_losses_backward.argtypes = [vp] * 8 + [f32] + [vp] * 6 + [i32] * 6 + [vp]
        L.gwd_color_adjust.argtypes = [vp, vp, vp, i64, i32, f32, vp]
        L.gwd_bmm.argtypes = [ctypes.POINTER(BmmDesc), vp]
        L.gwd_stride_place.argtypes = [vp, vp, vp] + [i32] * 8 + [vp]
        L.gwd_psp_pool_forward.argtypes = [vp] * 5 + [i32] * 5 + [vp]
        L.gwd_psp_pool_backward.argtypes = [vp] * 6 + [i32] * 6 + [vp]
        L.gwd_pos_sine.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_stem_pack.argtypes = [vp, vp, vp, vp]
        L.gwd_stem_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.gwd_softmax_forward.argtypes = [vp, vp, i64, i32, i32, vp]
        L.gwd_softmax_backward.argtypes = [vp, vp, vp, i64, i32, i32, vp]
        L.gwd_silog_sums.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_silog_backward.argtypes = [vp, vp, vp, vp, f32, f32, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_seg_ce_sum.argtypes = [vp, vp, vp, i64, i32, vp]
        L.gwd_seg_ce_backward.argtypes = [vp, vp, vp, f32, vp, i64, i32, vp]
        L.gwd_resample_forward.argtypes = [vp, vp] + [i32] * 9 + [vp]
        L.gwd_resample_backward.argtypes = [vp, vp] + [i32] * 7 + [vp, i32, i32, vp]
        L.gwd_avgpool_forward.argtypes = [vp, vp] + [i32] * 6 + [vp]
        L.gwd_avgpool_backward.argtypes = [vp, vp] + [i32] * 6 + [vp]
        sp = ctypes.POINTER(Strided)
        L.gwd_winattn_forward.argtypes = [sp, sp, sp, sp, vp, vp, i32, vp, i64, i32, i32, i32, f32, i32, vp]
        L.gwd_winattn_backward.argtypes = [sp] * 7 + [vp, vp, vp, i32, vp, i64, i32, i32, i32, f32, i32, i32, vp]
        L.gwd_ref_scores_forward.argtypes = [sp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
        L.gwd_ref_scores_backward.argtypes = [sp, vp, vp, sp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
        L.gwd_ref_mix_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_ref_mix_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_tokattn_forward.argtypes = [sp] * 4 + [i64, i32, i32, f32, i32, vp]
        L.gwd_tokattn_backward.argtypes = [sp] * 7 + [i64, i32, i32, f32, i32, vp]
        L.gwd_upsample_taps_collapse.argtypes = [vp, vp, i32, i32, i32, vp]
        L.gwd_upsample_taps_fold.argtypes = [vp, vp, i32, i32, vp]
        L.gwd_tokattn_pair_forward.argtypes = [sp] * 6 + [i64, i32, i32, f32, i32, vp]
        L.gwd_tokattn_pair_backward.argtypes = [sp] * 10 + [i64, i32, i32, f32, i32, vp]
        L.gwd_certain_sample.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, vp]
        L.gwd_lsap.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
        L.gwd_inorm_gelu_forward.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, ctypes.c_float, i32, vp]
        L.gwd_inorm_gelu_backward.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp]
        L.gwd_query_workspace.argtypes = [i32, ctypes.POINTER(ctypes.c_int64), i32]
        L.gwd_query_workspace.restype = ctypes.c_int64
        L.gwd_eval_accumulate.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, vp, i32, i64, f32, f32, i32, i32, vp]
        L.gwd_softmax_masked_forward.argtypes = [vp, vp, vp, i64, i32, i64, ctypes.c_float, i32, vp]
        L.gwd_softmax_scaled_backward.argtypes = [vp, vp, vp, i64, i32, ctypes.c_float, i32, vp]
        L.gwd_resample_backward_sep.argtypes = [vp, vp, vp] + [i32] * 9 + [vp]
        L.gwd_act_backward_colsum.argtypes = [vp, vp, vp, vp, i64, i32, i32, ctypes.c_float, vp, i32, vp]
        L.gwd_point_sample_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_point_sample_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_point_sample_backward_gather.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_point_sample_framed_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_point_sample_framed_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_weight_prep_batch.argtypes = [vp, i32, i32, vp, vp]
        L.gwd_window_map.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_window_map_multi.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.gwd_sqnorm.argtypes = [vp, vp, i64, vp]
        L.gwd_adamw_step.argtypes = [vp, vp, vp, vp, vp, vp, i64] + [f32] * 9 + [vp]

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _stream(*tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise HipUnavailable("gw_depth_amd ops run on a HIP device only (got a %s tensor); "
                                     "there is no CPU path" % t.device)
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with status %d" % (what, rc))

    def workspace_bytes(self, op, *dims):
        """gwd_query_workspace: bytes of caller-provided scratch for op (WS_INORM_GELU / WS_RESAMPLE_BWD)."""
        arr = (ctypes.c_int64 * len(dims))(*[int(d) for d in dims])
        n = self.lib.gwd_query_workspace(op, arr, len(dims))
        if n < 0:
            raise ValueError("gwd_query_workspace(%d, %r) -> %d" % (op, dims, n))
        return n

    def version(self):
        return self.lib.gwd_version()

    # ------------------------------------------------------------------ entry points
    @staticmethod
    def _desc(x, w, y, dims, z=None, scale=None, shift=None, residual=None, stride=1, pad=0,
              gather=GATHER_CONV, virt=(0, 0), act=ACT_NONE, act_scale=1.0, mult=None, gate=None, gate_act=ACT_NONE, ln=None):
        B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW = dims
        d = ConvDesc()
        d.x, d.w, d.y, d.z = _ptr(x), _ptr(w), _ptr(y), _ptr(z)
        d.scale, d.shift, d.residual = _ptr(scale), _ptr(shift), _ptr(residual)
        d.zero_page = _zero_page(x.device)
        d.mult = _ptr(mult)
        d.gate, d.gate_act = _ptr(gate), (gate_act if gate is not None else ACT_NONE)
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW = B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW
        d.stride, d.pad, d.gather, d.Hv, d.Wv = stride, pad, gather, virt[0], virt[1]
        d.act, d.act_scale, d.dtype = act, act_scale, dtype_code(x)
        if ln is not None:                              # ConvLn mode: (mean, rstd, real channel count), gwd_conv_desc.ln_*
            d.ln_mean, d.ln_rstd, d.ln_C = _ptr(ln[0]), _ptr(ln[1]), int(ln[2])
        return d

    def conv_forward(self, x, w, y, dims, **kw):
        """dims = (B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW); kw: z scale shift residual stride pad gather virt act act_scale mult
        gate gate_act (the epilogue's last step: backward of the activation whose output is `gate`); ln=(mean, rstd, C): the ConvLn
        epilogue (LayerNorm over the first C channels, scale / shift = gamma / beta) - returns False when the library has no fused
        kernel for the shape (nothing was launched)."""
        d = self._desc(x, w, y, dims, **kw)
        rc = self.lib.gwd_conv_forward(ctypes.byref(d), self._stream(x, w, y))
        if rc == -4 and kw.get("ln") is not None:
            return False                                # no fused ConvLn kernel for this shape: the caller runs the two kernels
        if rc == -4 and kw.get("gate") is not None and kw.get("gate_act") == ACT_GELU:
            return False                                # no kernel with the GELU gate for this shape: the caller applies it in a pass of its own
        self._check(rc, "gwd_conv_forward")

    def conv_wgrad(self, x, gy, dw, dims, **kw):
        d = self._desc(x, None, gy, dims, **kw)
        self._check(self.lib.gwd_conv_wgrad(ctypes.byref(d), _ptr(dw), self._stream(x, gy, dw)), "gwd_conv_wgrad")

    def conv_wgrad_batch(self, jobs):
        """jobs: tuples (x, gy, dw, dims, kw) as for conv_wgrad; one library call, grouped launches (gwd_conv_wgrad_batch)."""
        descs = (ConvDesc * len(jobs))()
        dws = (ctypes.c_void_p * len(jobs))()
        ts = []
        for i, (x, gy, dw, dims, kw) in enumerate(jobs):
            descs[i] = self._desc(x, None, gy, dims, **kw)
            dws[i] = _ptr(dw)
            ts += [x, gy, dw]
        self._check(self.lib.gwd_conv_wgrad_batch(descs, dws, len(jobs), self._stream(*ts)), "gwd_conv_wgrad_batch")

    def weight_prep(self, w, row_scale, w_fwd, w_dgrad, N, taps, C, dtype):
        self._check(self.lib.gwd_weight_prep(_ptr(w), _ptr(row_scale), _ptr(w_fwd), _ptr(w_dgrad), N, taps, C, dtype,
                                             self._stream(w, w_fwd, w_dgrad)), "gwd_weight_prep")

    def act_backward(self, gy, ref, gx, scale, rows, C, act, act_scale):
        self._check(self.lib.gwd_act_backward(_ptr(gy), _ptr(ref), _ptr(gx), _ptr(scale), rows, C, act, act_scale,
                                              dtype_code(gy), self._stream(gy, ref, gx)), "gwd_act_backward")

    def act_backward_colsum(self, gy, ref, gx, dbias, rows, C, act, act_scale, mult=None):
        """Fused activation backward + bias gradient (gy first multiplied by `mult` when given); False when the shape is not
        supported (use the separate calls)."""
        rc = self.lib.gwd_act_backward_colsum(_ptr(gy), _ptr(ref), _ptr(gx), _ptr(dbias), rows, C, act, act_scale, _ptr(mult),
                                              dtype_code(gy), self._stream(gy, ref, gx))
        if rc == -4:
            return False
        self._check(rc, "gwd_act_backward_colsum")
        return True

    def colsum(self, g, out, rows, C):
        self._check(self.lib.gwd_colsum(_ptr(g), _ptr(out), rows, C, dtype_code(g), self._stream(g, out)), "gwd_colsum")

    def colsum_batch(self, jobs):
        """jobs: up to COLSUM_BATCH tuples (g, out, rows, C) of ONE dtype with vector-shaped C (colsum_batchable)."""
        recs = (ColsumJob * len(jobs))()
        for r, (g, out, rows, C) in zip(recs, jobs):
            r.g, r.out, r.rows, r.C = g.data_ptr(), out.data_ptr(), rows, C
            if not (g.is_contiguous() and out.is_contiguous()):
                raise ValueError("kernel operand must be contiguous")
        ts = [t for j in jobs for t in j[:2]]
        self._check(self.lib.gwd_colsum_batch(recs, len(jobs), dtype_code(jobs[0][0]), self._stream(*ts)), "gwd_colsum_batch")

    @staticmethod
    def colsum_batchable(g, C):
        vec = 8 if g.dtype == torch.bfloat16 else 4
        return C % vec == 0 and C // vec <= 256

    def layernorm_forward(self, x, gamma, beta, y, mean, rstd, rows, C, gelu, residual=None, ld=0):
        """ld: row pitch in elements (0 = C); channels C..ld-1 are zero padding, written as zeros."""
        self._check(self.lib.gwd_layernorm_forward(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(residual), _ptr(y), _ptr(mean), _ptr(rstd),
                                                   rows, C, ld, int(gelu), dtype_code(x), self._stream(x, y)),
                    "gwd_layernorm_forward")

    def layernorm_backward(self, gy, x, gamma, beta, mean, rstd, gx, dgamma, dbeta, rows, C, gelu, ld=0, gskip=None, elu_input=False):
        """gskip: a second gradient of x, added to gx in the kernel; elu_input: x is an ELU output whose backward is applied to gx
        as well (GWD_LN_ELU_INPUT).  Returns False when that could not be done (no vector kernel for the shape: the call has then
        run WITHOUT both and the caller does them)."""
        flags = int(bool(gelu)) | (2 if elu_input else 0)
        rc = self.lib.gwd_layernorm_backward(_ptr(gy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(gx), _ptr(dgamma),
                                             _ptr(dbeta), rows, C, ld, flags, _ptr(gskip), dtype_code(x), self._stream(gy, x, gx))
        if rc == -4 and (gskip is not None or elu_input):
            self._check(self.lib.gwd_layernorm_backward(_ptr(gy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(gx), _ptr(dgamma),
                                                        _ptr(dbeta), rows, C, ld, int(bool(gelu)), None, dtype_code(x), self._stream(gy, x, gx)),
                        "gwd_layernorm_backward")
            return False
        self._check(rc, "gwd_layernorm_backward")
        return True

    def pos_counts(self, mask_full, mask_level, counts):
        """mask_full (B,H,W) bool/u8 -> mask_level (B,h,w) bool (nearest), counts (B,h,w,2) int16 (gwd_pos_sine, first stage)."""
        B, H, W = mask_full.shape
        _, h, w = mask_level.shape
        if tuple(counts.shape) != (B, h, w, 2) or counts.dtype != torch.int16 or mask_full.element_size() != 1 or mask_level.element_size() != 1:
            raise ValueError("pos_counts: byte masks and int16 counts (B,h,w,2) expected")
        self._check(self.lib.gwd_pos_sine(_ptr(mask_full), _ptr(mask_level), _ptr(counts), None, None, B, H, W, h, w, 0, 0,
                                          self._stream(mask_full, mask_level, counts)), "gwd_pos_sine")

    def pos_emit(self, counts, dim_t, out, normalize):
        """counts (B,h,w,2) int16 + dim_t (F) fp32 -> out (B,h,w,2F) fp32 (gwd_pos_sine, second stage)."""
        B, h, w, _ = counts.shape
        F = dim_t.numel()
        if tuple(out.shape) != (B, h, w, 2 * F) or out.dtype != torch.float32 or dim_t.dtype != torch.float32:
            raise ValueError("pos_emit: out fp32 (B,h,w,2F) expected")
        self._check(self.lib.gwd_pos_sine(None, None, _ptr(counts), _ptr(dim_t), _ptr(out), B, 0, 0, h, w, F, int(bool(normalize)),
                                          self._stream(counts, dim_t, out)), "gwd_pos_sine")

    def stem_pack(self, w, scale, packed):
        """w fp32 (64,7,7,3) [* scale (64)] -> packed bf16 (STEM_PACKED_ELEMS,) in gwd_stem_forward's operand order."""
        if tuple(w.shape) != (64, 7, 7, 3) or w.dtype != torch.float32 or packed.numel() != STEM_PACKED_ELEMS or packed.dtype != torch.bfloat16:
            raise ValueError("stem_pack: w fp32 (64,7,7,3) and a bf16 buffer of %d values expected" % STEM_PACKED_ELEMS)
        self._check(self.lib.gwd_stem_pack(_ptr(w), _ptr(scale), _ptr(packed), self._stream(w, packed)), "gwd_stem_pack")

    def stem_forward(self, x, packed, shift, y):
        """conv 7x7 s2 p3 (3 -> 64) + shift + ReLU + max-pool 3x3 s2 p1: x bf16 (B,H,W,3) -> y bf16 (B,Hp,Wp,64)."""
        B, H, W, C = x.shape
        Hp, Wp = stem_out(H), stem_out(W)
        if C != 3 or tuple(y.shape) != (B, Hp, Wp, 64) or not x.is_contiguous() or not y.is_contiguous():
            raise ValueError("stem_forward: x (B,H,W,3) and y (B,%d,%d,64) contiguous expected" % (Hp, Wp))
        self._check(self.lib.gwd_stem_forward(_ptr(x), _ptr(packed), _ptr(shift), _ptr(y), B, H, W, dtype_code(x), self._stream(x, packed, y)),
                    "gwd_stem_forward")

    def unpad_add_batch(self, jobs):
        """jobs: tuples (src, dst, N, taps, G, Cg, Cgp): dst (N, taps, G*Cg) += src (.., taps, G*Cgp), fp32; one launch per
        UNPAD_BATCH jobs (gwd_unpad_add_batch)."""
        for i0 in range(0, len(jobs), UNPAD_BATCH):
            part = jobs[i0:i0 + UNPAD_BATCH]
            recs = (UnpadJob * len(part))()
            ts = []
            for i, (src, dst, N, taps, G, Cg, Cgp) in enumerate(part):
                if src.dtype != torch.float32 or dst.dtype != torch.float32 or not src.is_contiguous() or not dst.is_contiguous():
                    raise ValueError("unpad_add_batch: contiguous fp32 tensors expected")
                if dst.numel() != N * taps * G * Cg or src.numel() < N * taps * G * Cgp:
                    raise ValueError("unpad_add_batch: shapes do not match the job")
                r = recs[i]
                r.src, r.dst, r.N, r.taps, r.G, r.Cg, r.Cgp, r.block0 = src.data_ptr(), dst.data_ptr(), N, taps, G, Cg, Cgp, 0
                ts += [src, dst]
            self._check(self.lib.gwd_unpad_add_batch(recs, len(part), self._stream(*ts)), "gwd_unpad_add_batch")

    def softmax_forward(self, x, y, rows, L):
        self._check(self.lib.gwd_softmax_forward(_ptr(x), _ptr(y), rows, L, dtype_code(x), self._stream(x, y)),
                    "gwd_softmax_forward")

    def softmax_masked_forward(self, x, key_mask, y, rows, L, rows_per_mask, scale):
        self._check(self.lib.gwd_softmax_masked_forward(_ptr(x), _ptr(key_mask), _ptr(y), rows, L, rows_per_mask, scale,
                                                        dtype_code(x), self._stream(x, y)), "gwd_softmax_masked_forward")

    def softmax_scaled_backward(self, gy, y, gx, rows, L, scale):
        self._check(self.lib.gwd_softmax_scaled_backward(_ptr(gy), _ptr(y), _ptr(gx), rows, L, scale, dtype_code(y),
                                                         self._stream(gy, y, gx)), "gwd_softmax_scaled_backward")

    def softmax_backward(self, gy, y, gx, rows, L):
        self._check(self.lib.gwd_softmax_backward(_ptr(gy), _ptr(y), _ptr(gx), rows, L, dtype_code(y),
                                                  self._stream(gy, y, gx)), "gwd_softmax_backward")

    def eval_accumulate(self, pred, gt, seg, seg_strides, seg_gt, workspace, measures, running, confusion, B, HW, dmin, dmax):
        """gwd_eval_accumulate.  seg may be a strided view: seg_strides = (image, pixel, class) element strides."""
        sb, sp, sc = (int(v) for v in seg_strides) if seg is not None else (0, 0, 0)
        self._check(self.lib.gwd_eval_accumulate(
            _ptr(pred), _ptr(gt), None if seg is None else ctypes.c_void_p(seg.data_ptr()), sb, sp, sc, _ptr(seg_gt),
            _ptr(workspace), _ptr(measures), _ptr(running), _ptr(confusion), B, HW, float(dmin), float(dmax),
            dtype_code(pred) if pred is not None else F32, dtype_code(seg) if seg is not None else F32,
            self._stream(pred, gt, seg, seg_gt, workspace, measures, running, confusion)), "gwd_eval_accumulate")

    def plane_loss_forward(self, depth, valid, tri, n_planes, P, H, W, min_area, workspace, stats, loss):
        self._check(self.lib.gwd_plane_loss_forward(_ptr(depth), _ptr(valid), _ptr(tri), _ptr(n_planes), P, H, W, min_area,
                                                    _ptr(workspace), _ptr(stats), _ptr(loss), dtype_code(depth),
                                                    self._stream(depth, valid, tri, n_planes, workspace, stats, loss)), "gwd_plane_loss_forward")

    def plane_loss_backward(self, depth, valid, tri, n_planes, P, H, W, stats, gloss, gdepth):
        self._check(self.lib.gwd_plane_loss_backward(_ptr(depth), _ptr(valid), _ptr(tri), _ptr(n_planes), P, H, W, _ptr(stats),
                                                     _ptr(gloss), _ptr(gdepth), dtype_code(depth),
                                                     self._stream(depth, valid, tri, n_planes, stats, gloss, gdepth)), "gwd_plane_loss_backward")

    def collate(self, samples, H, W, mean, std, images, mask, depth, seg):
        """gwd_collate.  samples: (rgb uint8 (h,w,3), depth_mm int32 (h,w) | None, labels uint8 (h,w) | None) device tensors."""
        jobs = (ImageJob * len(samples))()
        ts = [images, mask, depth, seg]
        for j, (rgb, dmm, lab) in zip(jobs, samples):
            j.rgb, j.depth_mm, j.labels = _ptr(rgb), _ptr(dmm), _ptr(lab)
            j.h, j.w = rgb.shape[0], rgb.shape[1]
            ts += [rgb, dmm, lab]
        f3 = ctypes.c_float * 3
        self._check(self.lib.gwd_collate(jobs, len(samples), H, W, f3(*mean), f3(*std), _ptr(images), _ptr(mask), _ptr(depth),
                                         _ptr(seg), dtype_code(images), self._stream(*ts)), "gwd_collate")

    @staticmethod
    def _tok(t):
        """(B, tokens, E) tensor or last-dim slice: raw pointer + token stride; dense batches, unit channel stride."""
        if t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            raise ValueError("attention operand must be (B, tokens, E) with unit channel stride and dense token rows")
        return ctypes.c_void_p(t.data_ptr()), t.stride(1)

    def mha_flash_forward(self, q, k, v, key_padding_mask, mult, out, lse, H, scale):
        """gwd_mha_flash_forward: q (B,L,32H), k / v (B,S,32H) bf16 tensors or channel slices of a packed projection."""
        B, L, S = q.shape[0], q.shape[1], k.shape[1]
        (qp, qs), (kp, ks), (vp_, vs), (op, os_) = self._tok(q), self._tok(k), self._tok(v), self._tok(out)
        self._check(self.lib.gwd_mha_flash_forward(qp, kp, vp_, qs, ks, vs, _ptr(key_padding_mask), _ptr(mult), op, os_, _ptr(lse),
                                                   B, H, L, S, float(scale), dtype_code(q),
                                                   self._stream(q, k, v, key_padding_mask, mult, out, lse)), "gwd_mha_flash_forward")

    def mha_flash_backward(self, q, k, v, go, out, key_padding_mask, mult, lse, delta, gq, gk, gv, H, scale):
        B, L, S = q.shape[0], q.shape[1], k.shape[1]
        ts = [self._tok(t) for t in (q, k, v, go, out, gq, gk, gv)]
        self._check(self.lib.gwd_mha_flash_backward(ts[0][0], ts[1][0], ts[2][0], ts[3][0], ts[4][0], ts[0][1], ts[1][1], ts[2][1],
                                                    ts[3][1], ts[4][1], _ptr(key_padding_mask), _ptr(mult), _ptr(lse), _ptr(delta),
                                                    ts[5][0], ts[6][0], ts[7][0], ts[5][1], ts[6][1], ts[7][1], B, H, L, S,
                                                    float(scale), dtype_code(q),
                                                    self._stream(q, k, v, go, out, key_padding_mask, mult, lse, delta, gq, gk, gv)),
                    "gwd_mha_flash_backward")

    def anchor_depth_forward(self, att, anchor, pred, B, P, R):
        self._check(self.lib.gwd_anchor_depth_forward(_ptr(att), _ptr(anchor), _ptr(pred), B, P, R, dtype_code(att),
                                                      self._stream(att, anchor, pred)), "gwd_anchor_depth_forward")

    def anchor_depth_backward(self, att, anchor, gpred, datt, danchor, B, P, R):
        self._check(self.lib.gwd_anchor_depth_backward(_ptr(att), _ptr(anchor), _ptr(gpred), _ptr(datt), _ptr(danchor), B, P, R,
                                                       dtype_code(att), self._stream(att, anchor, gpred, datt, danchor)),
                    "gwd_anchor_depth_backward")

    def silog_sums(self, pred, gt, sums, B, h, w, H, W, log_err):
        self._check(self.lib.gwd_silog_sums(_ptr(pred), _ptr(gt), _ptr(sums), B, h, w, H, W, int(log_err),
                                            dtype_code(pred), self._stream(pred, gt, sums)), "gwd_silog_sums")

    def silog_finalize(self, sums, lam, scale, loss):
        self._check(self.lib.gwd_silog_finalize(_ptr(sums), lam, scale, _ptr(loss), self._stream(sums, loss)), "gwd_silog_finalize")

    def silog_backward(self, pred, gt, sums, gloss, weight, lam, gpred, B, h, w, H, W, log_err):
        self._check(self.lib.gwd_silog_backward(_ptr(pred), _ptr(gt), _ptr(sums), _ptr(gloss), weight, lam, _ptr(gpred),
                                                B, h, w, H, W, int(log_err), dtype_code(pred),
                                                self._stream(pred, gt, gpred)), "gwd_silog_backward")

    def seg_ce_sum(self, logits, target, out, P):
        self._check(self.lib.gwd_seg_ce_sum(_ptr(logits), _ptr(target), _ptr(out), P, dtype_code(logits),
                                            self._stream(logits, target, out)), "gwd_seg_ce_sum")

    def seg_ce_backward(self, logits, target, gloss, scale, glogits, P):
        self._check(self.lib.gwd_seg_ce_backward(_ptr(logits), _ptr(target), _ptr(gloss), scale, _ptr(glogits), P,
                                                 dtype_code(logits), self._stream(logits, glogits)), "gwd_seg_ce_backward")

    @staticmethod
    def _pixel_pitch(t, H, W, C):
        """Element pitch between pixels of a (B,H,W,C) tensor that may be a channel slice of a wider pixel-major map."""
        ld = t.stride(2) if t.dim() == 4 and W > 1 else C
        if t.dim() == 4 and (t.stride(3) != 1 or (H > 1 and t.stride(1) != W * ld) or (t.shape[0] > 1 and t.stride(0) != H * W * ld)):
            raise ValueError("pixel-major tensor or a channel slice of one expected, got strides %r" % (t.stride(),))
        return ld

    def resample_forward(self, x, y, B, Hs, Ws, Ho, Wo, C, mode):
        """y may be a channel slice of a wider (B,Ho,Wo,*) map: the result is written in place there."""
        self._check(self.lib.gwd_resample_forward(_ptr(x), _ptr_pitched(y), B, Hs, Ws, Ho, Wo, C, mode, self._pixel_pitch(y, Ho, Wo, C), dtype_code(x),
                                                  self._stream(x, y)), "gwd_resample_forward")

    def resample_backward(self, gy, gx, B, Hs, Ws, Ho, Wo, C, mode, gate=None, gate_act=ACT_NONE):
        """gate / gate_act: gx *= act'(.) of the activation whose output is `gate` (B,Hs,Ws,C); False when the shape cannot (the call
        has then run WITHOUT the gate)."""
        rc = self.lib.gwd_resample_backward(_ptr(gy), _ptr(gx), B, Hs, Ws, Ho, Wo, C, mode, _ptr(gate), gate_act if gate is not None else 0,
                                            dtype_code(gy), self._stream(gy, gx))
        if rc == -4 and gate is not None:
            self._check(self.lib.gwd_resample_backward(_ptr(gy), _ptr(gx), B, Hs, Ws, Ho, Wo, C, mode, None, 0, dtype_code(gy),
                                                       self._stream(gy, gx)), "gwd_resample_backward")
            return False
        self._check(rc, "gwd_resample_backward")
        return True

    def resample_backward_sep(self, gy, tmp, gx, B, Hs, Ws, Ho, Wo, C, mode):
        """Separable backward through the fp32 scratch `tmp` (B,Ho,Ws,C); False when C is not vector-sized."""
        rc = self.lib.gwd_resample_backward_sep(_ptr_pitched(gy), _ptr(tmp), _ptr(gx), B, Hs, Ws, Ho, Wo, C, mode, self._pixel_pitch(gy, Ho, Wo, C),
                                                dtype_code(gy), self._stream(gy, gx))
        if rc == -4:
            return False
        self._check(rc, "gwd_resample_backward_sep")
        return True

    def stride_place(self, src, residual, dst, stride):
        """dst (B,H,W,C) = [residual +] src (B,Ho,Wo,C) on the pixels (stride*i, stride*j), zeros elsewhere; False when C is not vector-sized."""
        B, H, W, C = dst.shape
        Ho, Wo = src.shape[1], src.shape[2]
        rc = self.lib.gwd_stride_place(_ptr(src), _ptr(residual), _ptr(dst), B, H, W, Ho, Wo, C, stride, dtype_code(dst), self._stream(src, dst))
        if rc == -4:
            return False
        self._check(rc, "gwd_stride_place")
        return True

    def psp_pool_forward(self, x, p16, p8, p4, p2):
        """x (B,H,W,C) -> its 16/8/4/2 average pools in one pass; False when the shape is not supported (use avgpool_forward)."""
        B, H, W, C = x.shape
        rc = self.lib.gwd_psp_pool_forward(_ptr(x), _ptr(p16), _ptr(p8), _ptr(p4), _ptr(p2), B, H, W, C, dtype_code(x), self._stream(x, p16, p8, p4, p2))
        if rc == -4:
            return False
        self._check(rc, "gwd_psp_pool_forward")
        return True

    def psp_pool_backward(self, g_pass, g16, g8, g4, g2, gx):
        """gx (B,H,W,C) = g_pass (may be a channel slice of a wider map, or None) + the four pools' gradients spread back."""
        B, H, W, C = gx.shape
        ld = self._pixel_pitch(g_pass, H, W, C) if g_pass is not None else 0
        self._check(self.lib.gwd_psp_pool_backward(_ptr_pitched(g_pass), _ptr(g16), _ptr(g8), _ptr(g4), _ptr(g2), _ptr(gx), B, H, W, C, ld,
                                                   dtype_code(gx), self._stream(gx, g16, g8, g4, g2)), "gwd_psp_pool_backward")

    def avgpool_forward(self, x, y, B, H, W, C, k):
        self._check(self.lib.gwd_avgpool_forward(_ptr(x), _ptr(y), B, H, W, C, k, dtype_code(x), self._stream(x, y)),
                    "gwd_avgpool_forward")

    def avgpool_backward(self, gy, gx, B, H, W, C, k):
        self._check(self.lib.gwd_avgpool_backward(_ptr(gy), _ptr(gx), B, H, W, C, k, dtype_code(gy),
                                                  self._stream(gy, gx)), "gwd_avgpool_backward")

    def winattn_forward(self, q, k, v, o, bias, region, wpi, scale, rel_index=None):
        """q,k,v,o: (W, 49, heads, hd) tensors or views; bias (heads,49,49) fp32 - or, with rel_index (49*49 int32), the
        (n_rel, heads) relative-position TABLE itself; region (wpi,49) int32 or None."""
        W, N, H, D = q.shape
        s = [_strided(t) for t in (q, k, v, o)]
        n_rel = bias.shape[0] if rel_index is not None else 0
        self._check(self.lib.gwd_winattn_forward(*[ctypes.byref(x) for x in s], _ptr(bias), _ptr(rel_index), n_rel, _ptr(region), W, wpi,
                                                 H, D, scale, dtype_code(q), self._stream(q, k, v, o, bias, rel_index)), "gwd_winattn_forward")

    def winattn_backward(self, q, k, v, go, gq, gk, gv, bias, dbias, region, wpi, scale, rel_index=None, head_major=False):
        """dbias: same layout as bias (dense, or the table's gradient with rel_index); ACCUMULATED into.  head_major (rel_index only):
        dbias is a (heads, n_rel) scratch instead of the (n_rel, heads) table gradient."""
        W, N, H, D = q.shape
        s = [_strided(t) for t in (q, k, v, go, gq, gk, gv)]
        n_rel = bias.shape[0] if rel_index is not None else 0
        self._check(self.lib.gwd_winattn_backward(*[ctypes.byref(x) for x in s], _ptr(bias), _ptr(dbias), _ptr(rel_index), n_rel,
                                                  _ptr(region), W, wpi, H, D, scale, int(bool(head_major)), dtype_code(q),
                                                  self._stream(q, go, gq, bias, dbias, rel_index)), "gwd_winattn_backward")

    def ref_scores_forward(self, q, ref_k, ra, B, nwin, scale):
        """q (B*nwin, 49, H, hd) operand / view, ref_k (B, R, H*hd), ra (B, nwin*49, R, H) out."""
        H, hd, R = q.shape[2], q.shape[3], ref_k.shape[1]
        sq = _strided(q)
        self._check(self.lib.gwd_ref_scores_forward(ctypes.byref(sq), _ptr(ref_k), _ptr(ra), B, nwin, R, H, hd, float(scale),
                                                    dtype_code(q), self._stream(q, ref_k, ra)), "gwd_ref_scores_forward")

    def ref_scores_backward(self, q, ref_k, g, dq, d_ref_k, B, nwin, scale):
        H, hd, R = q.shape[2], q.shape[3], ref_k.shape[1]
        sq, sdq = _strided(q), _strided(dq)
        self._check(self.lib.gwd_ref_scores_backward(ctypes.byref(sq), _ptr(ref_k), _ptr(g), ctypes.byref(sdq), _ptr(d_ref_k), B, nwin,
                                                     R, H, hd, float(scale), dtype_code(q), self._stream(q, ref_k, g, dq, d_ref_k)),
                    "gwd_ref_scores_backward")

    def ref_mix_forward(self, ra, ref_v, q_new, att, H):
        """ra (B, T, R, H), ref_v (B, R, C) -> q_new (B, T, C), att (B, T, R, H) or None."""
        B, T, R = ra.shape[0], ra.shape[1], ra.shape[2]
        self._check(self.lib.gwd_ref_mix_forward(_ptr(ra), _ptr(ref_v), _ptr(q_new), _ptr(att), B, T, R, H, ref_v.shape[2] // H,
                                                 dtype_code(ra), self._stream(ra, ref_v, q_new, att)), "gwd_ref_mix_forward")

    def ref_mix_backward(self, att, ref_v, g, d_ra, d_ref_v, H):
        B, T, R = att.shape[0], att.shape[1], att.shape[2]
        self._check(self.lib.gwd_ref_mix_backward(_ptr(att), _ptr(ref_v), _ptr(g), _ptr(d_ra), _ptr(d_ref_v), B, T, R, H,
                                                  ref_v.shape[2] // H, dtype_code(att), self._stream(att, ref_v, g, d_ra, d_ref_v)),
                    "gwd_ref_mix_backward")

    def tokattn_forward(self, q, k, v, o, scale):
        """q,o: (W,49,heads,4); k,v: (W,49,heads,e) tensors or views."""
        W, N, H, _ = q.shape
        s = [_strided(t) for t in (q, k, v, o)]
        self._check(self.lib.gwd_tokattn_forward(*[ctypes.byref(x) for x in s], W, H, k.shape[3], scale, dtype_code(q),
                                                 self._stream(q, k, v, o)), "gwd_tokattn_forward")

    def tokattn_backward(self, q, k, v, go, gq, gk, gv, scale):
        W, N, H, _ = q.shape
        s = [_strided(t) for t in (q, k, v, go, gq, gk, gv)]
        self._check(self.lib.gwd_tokattn_backward(*[ctypes.byref(x) for x in s], W, H, k.shape[3], scale, dtype_code(q),
                                                  self._stream(q, go, gq)), "gwd_tokattn_backward")

    def upsample_taps_collapse(self, w, wk):
        """w (Cout,3,3,Cin) fp32 -> wk (Cin,4,4,Cout): the 4x4 / stride-2 taps of a 3x3 convolution over a 2x nearest-upsampled map."""
        Cout, KH, KW, Cin = w.shape
        if (KH, KW) != (3, 3) or tuple(wk.shape) != (Cin, 4, 4, Cout) or w.dtype != torch.float32 or not (w.is_contiguous() and wk.is_contiguous()):
            raise ValueError("upsample_taps_collapse: w (Cout,3,3,Cin) fp32 -> wk (Cin,4,4,Cout), both contiguous")
        self._check(self.lib.gwd_upsample_taps_collapse(_ptr(w), _ptr(wk), Cout, Cin, dtype_code(wk), self._stream(w, wk)), "gwd_upsample_taps_collapse")

    def upsample_taps_fold(self, D, dw):
        """dw (Cout,3,3,Cin) fp32 += the 4x4-form weight gradient D (Cin,4,4,Cout) fp32 folded back onto the nine taps."""
        Cout, KH, KW, Cin = dw.shape
        if (KH, KW) != (3, 3) or tuple(D.shape) != (Cin, 4, 4, Cout) or D.dtype != torch.float32 or dw.dtype != torch.float32 \
                or not (D.is_contiguous() and dw.is_contiguous()):
            raise ValueError("upsample_taps_fold: D (Cin,4,4,Cout) fp32 -> dw (Cout,3,3,Cin) fp32, both contiguous")
        self._check(self.lib.gwd_upsample_taps_fold(_ptr(D), _ptr(dw), Cout, Cin, self._stream(D, dw)), "gwd_upsample_taps_fold")

    def tokattn_pair_forward(self, q, q2, k, v, o, o2, scale):
        """Both class tokens against the same k / v in one launch (bf16): q, q2 -> o, o2."""
        W, N, H, _ = q.shape
        s = [_strided(t) for t in (q, q2, k, v, o, o2)]
        self._check(self.lib.gwd_tokattn_pair_forward(*[ctypes.byref(x) for x in s], W, H, k.shape[3], scale, dtype_code(q),
                                                      self._stream(q, q2, k, v, o, o2)), "gwd_tokattn_pair_forward")

    def tokattn_pair_backward(self, q, q2, k, v, go, go2, gq, gq2, gk, gv, scale):
        """gq, gq2 per token; gk, gv summed over both tokens."""
        W, N, H, _ = q.shape
        s = [_strided(t) for t in (q, q2, k, v, go, go2, gq, gq2, gk, gv)]
        self._check(self.lib.gwd_tokattn_pair_backward(*[ctypes.byref(x) for x in s], W, H, k.shape[3], scale, dtype_code(q),
                                                       self._stream(q, go, gq)), "gwd_tokattn_pair_backward")

    def certain_sample(self, small, large, coords, edges, sample_num):
        """small (B,1,hs,ws), large (B,1,H,W) fp32; edges (I+1,) fp32; coords (B,S,1,2) fp32 out."""
        B, _, hs, ws = small.shape
        H, W = large.shape[-2:]
        self._check(self.lib.gwd_certain_sample(_ptr(small), _ptr(large), _ptr(coords), B, hs, ws, H, W, _ptr(edges),
                                                edges.numel() - 1, sample_num, self._stream(small, large, coords)),
                    "gwd_certain_sample")

    def bmm(self, a, b, c, M, N, K, a_kmajor=False, b_kmajor=False, alpha=1.0, accumulate=False, splits=1):
        """c[b0][b1] (M x N) = alpha * A (M x K) @ B (N x K)^T over two batch dims (gwd_bmm).  a, b, c: 4-D tensors / views with unit
        inner stride, dims (b0, b1, outer, inner); an operand is (rows, K) - or (K, rows) when *_kmajor.  A batch dim of size 1 in a
        or b broadcasts.  accumulate: c is fp32 and is ADDED to (required for splits > 1)."""
        for t in (a, b, c):
            if t.dim() != 4 or t.stride(3) != 1:
                raise ValueError("bmm: 4-D operands with unit inner stride expected")
        nb0, nb1 = c.shape[0], c.shape[1]
        d = BmmDesc()
        d.a, d.b, d.c = a.data_ptr(), b.data_ptr(), c.data_ptr()
        bs = lambda t, i: 0 if t.shape[i] == 1 else t.stride(i)
        d.a_sb0, d.a_sb1, d.a_ld = bs(a, 0), bs(a, 1), a.stride(2)
        d.b_sb0, d.b_sb1, d.b_ld = bs(b, 0), bs(b, 1), b.stride(2)
        d.c_sb0, d.c_sb1, d.c_ld = c.stride(0), c.stride(1), c.stride(2)
        d.M, d.N, d.K, d.nb0, d.nb1 = M, N, K, nb0, nb1
        d.a_kmajor, d.b_kmajor, d.c_is_f32_accumulate, d.splits = int(a_kmajor), int(b_kmajor), int(accumulate), int(splits)
        if accumulate and c.dtype != torch.float32:
            raise ValueError("bmm: an accumulated result is fp32")
        d.alpha, d.dtype = float(alpha), dtype_code(a)
        self._check(self.lib.gwd_bmm(ctypes.byref(d), self._stream(a, b, c)), "gwd_bmm")

    COLOR_MODES = {"brightness": 0, "contrast": 1, "saturation": 2, "hue": 3}

    def color_adjust(self, rgb, out, mode, factor, scratch=None):
        """One ColorJitter adjustment on a uint8 (h,w,3) device image (gwd_color_adjust); scratch: 1-element int64 tensor for 'contrast'."""
        if rgb.dtype != torch.uint8 or out.dtype != torch.uint8 or rgb.shape != out.shape or rgb.shape[-1] != 3:
            raise ValueError("color_adjust: uint8 (h,w,3) images expected")
        self._check(self.lib.gwd_color_adjust(_ptr(rgb), _ptr(out), _ptr(scratch), rgb.numel() // 3, self.COLOR_MODES[mode], float(factor),
                                              self._stream(rgb, out)), "gwd_color_adjust")

    def resample_u8_pass(self, src, dst, bounds, kk, axis, row_stride, base0, step0, base1, step1):
        """One pass of Pillow's BILINEAR resize over uint8 pixels (gwd_resample_u8_pass); src may be a window / flipped view given by
        the index maps, dst dense: (other, n_out, C) for axis 1, (n_out, other, C) for axis 0."""
        n_out, ksize = kk.shape
        other, C = (dst.shape[0], dst.shape[2]) if axis == 1 else (dst.shape[1], dst.shape[2])
        if src.dtype != torch.uint8 or dst.dtype != torch.uint8 or bounds.dtype != torch.int32 or kk.dtype != torch.int32 or not dst.is_contiguous():
            raise ValueError("resample_u8_pass: uint8 images and int32 tables expected")
        if dst.shape[1 if axis == 1 else 0] != n_out or tuple(bounds.shape) != (n_out, 2):
            raise ValueError("resample_u8_pass: table / output shapes do not match")
        self._check(self.lib.gwd_resample_u8_pass(_ptr_pitched(src), _ptr(dst), _ptr(bounds), _ptr(kk), ksize, axis, n_out, other, C, row_stride,
                                                  base0, step0, base1, step1, self._stream(src, dst, bounds, kk)), "gwd_resample_u8_pass")

    def gather2d(self, src, dst, ytab, xtab, row_stride_bytes, elem_bytes):
        oh, ow = ytab.numel(), xtab.numel()
        if ytab.dtype != torch.int32 or xtab.dtype != torch.int32 or not dst.is_contiguous() or dst.numel() * dst.element_size() != oh * ow * elem_bytes:
            raise ValueError("gather2d: int32 tables and a dense (oh, ow) output expected")
        self._check(self.lib.gwd_gather2d(_ptr_pitched(src), _ptr(dst), _ptr(ytab), _ptr(xtab), oh, ow, row_stride_bytes, elem_bytes,
                                          self._stream(src, dst, ytab, xtab)), "gwd_gather2d")

    def match_cost(self, logits, lines, tgt_lines, tgt_labels, cost, w_line, w_class):
        """cost (L,B,Q,cap) of matcher.py:52-70 from logits (L,B,Q,K), lines (L,B,Q,D), padded targets (cap,D) / (cap,) int64."""
        L_, B, Q, K = logits.shape
        D, cap = lines.shape[-1], tgt_lines.shape[0]
        if tgt_labels.dtype != torch.int64 or tuple(cost.shape) != (L_, B, Q, cap) or any(t.dtype != torch.float32 for t in (logits, lines, tgt_lines, cost)):
            raise ValueError("match_cost: fp32 tensors, int64 labels and cost (L,B,Q,cap) expected")
        self._check(self.lib.gwd_match_cost(_ptr(logits), _ptr(lines), _ptr(tgt_lines), _ptr(tgt_labels), _ptr(cost), L_, B, Q, cap, K, D,
                                            w_line, w_class, self._stream(logits, lines, cost)), "gwd_match_cost")

    def set_losses_forward(self, logits, lines, tgt_lines, tgt_labels, bidx, valid, qot, class_weight, num_items, world, target_class, ce, l1, wsum):
        L_, B, Q, K = logits.shape
        D, cap = lines.shape[-1], tgt_lines.shape[0]
        if any(t.dtype != torch.int32 for t in (bidx, valid, qot, target_class)) or tgt_labels.dtype != torch.int64:
            raise ValueError("set_losses_forward: int32 bidx / valid / qot / target_class and int64 labels expected")
        self._check(self.lib.gwd_set_losses_forward(_ptr(logits), _ptr(lines), _ptr(tgt_lines), _ptr(tgt_labels), _ptr(bidx), _ptr(valid), _ptr(qot),
                                                    _ptr(class_weight), _ptr(num_items), float(world), _ptr(target_class), _ptr(ce), _ptr(l1), _ptr(wsum),
                                                    L_, B, Q, cap, K, D, self._stream(logits, lines, ce, l1)), "gwd_set_losses_forward")

    def set_losses_backward(self, logits, lines, tgt_lines, bidx, valid, qot, class_weight, num_items, world, target_class, wsum, g_ce, g_l1,
                            dlogits, dlines):
        L_, B, Q, K = logits.shape
        D, cap = lines.shape[-1], tgt_lines.shape[0]
        self._check(self.lib.gwd_set_losses_backward(_ptr(logits), _ptr(lines), _ptr(tgt_lines), _ptr(bidx), _ptr(valid), _ptr(qot), _ptr(class_weight),
                                                     _ptr(num_items), float(world), _ptr(target_class), _ptr(wsum), _ptr(g_ce), _ptr(g_l1),
                                                     _ptr(dlogits), _ptr(dlines), L_, B, Q, cap, K, D, self._stream(logits, lines, dlogits, dlines)),
                    "gwd_set_losses_backward")

    def lsap(self, cost, col_offsets, out, max_targets):
        """cost (layers,B,Q,sumT) fp32; col_offsets (B+1,) int32 DEVICE data (image b owns columns [off[b], off[b+1]),
        at most max_targets <= 64 of them; columns from off[B] on are padding); out (layers,sumT) int32: the query
        assigned to every target column, Q for padding columns."""
        L_, B, Q, sumT = cost.shape
        self._check(self.lib.gwd_lsap(_ptr(cost), _ptr(col_offsets), _ptr(out), L_, B, Q, sumT, max_targets,
                                      self._stream(cost, col_offsets, out)), "gwd_lsap")

    def inorm_gelu_forward(self, a, u, y, part, stat, B, L, C, S, eps):
        self._check(self.lib.gwd_inorm_gelu_forward(_ptr(a), _ptr(u), _ptr(y), _ptr(part), _ptr(stat), B, L, C, S, eps,
                                                    dtype_code(u), self._stream(a, u, y)), "gwd_inorm_gelu_forward")

    def inorm_gelu_backward(self, gy, u, stat, part, du, B, L, C, S):
        self._check(self.lib.gwd_inorm_gelu_backward(_ptr(gy), _ptr(u), _ptr(stat), _ptr(part), _ptr(du), B, L, C, S,
                                                     dtype_code(u), self._stream(gy, u, du)), "gwd_inorm_gelu_backward")

    def point_sample_forward(self, fmap, coords, out, B, H, W, C, S, mode):
        self._check(self.lib.gwd_point_sample_forward(_ptr(fmap), _ptr(coords), _ptr(out), B, H, W, C, S, mode, dtype_code(fmap),
                                                      self._stream(fmap, out)), "gwd_point_sample_forward")

    def point_sample_backward_gather(self, gout, coords, gmap, B, H, W, C, S, mode):
        """Writes every element of gmap (no pre-zeroing); False when the shape is not supported (zero gmap, point_sample_backward)."""
        rc = self.lib.gwd_point_sample_backward_gather(_ptr(gout), _ptr(coords), _ptr(gmap), B, H, W, C, S, mode, dtype_code(gmap),
                                                       self._stream(gout, gmap))
        if rc == -4:
            return False
        self._check(rc, "gwd_point_sample_backward_gather")
        return True

    def point_sample_framed_forward(self, fmap, coords, out, B, H, W, C, S, frame):
        """Nearest sampling in the padded / rolled (Hf, Wf, shift) frame of the map, without building it."""
        Hf, Wf, shift = frame
        self._check(self.lib.gwd_point_sample_framed_forward(_ptr(fmap), _ptr(coords), _ptr(out), B, H, W, C, S, Hf, Wf, shift, dtype_code(fmap),
                                                             self._stream(fmap, out)), "gwd_point_sample_framed_forward")

    def point_sample_framed_backward(self, gout, coords, gmap, B, H, W, C, S, frame):
        """Every element of gmap written; S <= 256 (the caller checks before choosing the framed form)."""
        Hf, Wf, shift = frame
        self._check(self.lib.gwd_point_sample_framed_backward(_ptr(gout), _ptr(coords), _ptr(gmap), B, H, W, C, S, Hf, Wf, shift, dtype_code(gmap),
                                                              self._stream(gout, gmap)), "gwd_point_sample_framed_backward")

    def point_sample_backward(self, gout, coords, gmap, B, H, W, C, S, mode):
        self._check(self.lib.gwd_point_sample_backward(_ptr(gout), _ptr(coords), _ptr(gmap), B, H, W, C, S, mode, dtype_code(gmap),
                                                       self._stream(gout, gmap)), "gwd_point_sample_backward")

    def weight_prep_batch(self, table, n_jobs, total_blocks, block_job=None):
        """table: device uint8 tensor holding n_jobs packed gwd_prep_job records (see PrepJob); block_job: device int32 (total_blocks,)
        job index per block (optional, saves every block a binary search)."""
        self._check(self.lib.gwd_weight_prep_batch(_ptr(table), n_jobs, total_blocks, _ptr(block_job), self._stream(table)), "gwd_weight_prep_batch")

    def window_map(self, src, dst, B, H, W, C, shift, gather, residual=None):
        self._check(self.lib.gwd_window_map(_ptr(src), _ptr(dst), _ptr(residual), B, H, W, C, shift, int(gather), dtype_code(src),
                                            self._stream(src, dst)), "gwd_window_map")

    def window_map_multi(self, srcs, dsts, B, H, W, Cs, shift, gather, residuals=None):
        """gwd_window_map for up to 4 maps of one geometry in one launch (residuals: list with None entries, or None)."""
        n = len(srcs)
        vpn, i32n = ctypes.c_void_p * n, ctypes.c_int32 * n
        res = [None] * n if residuals is None else list(residuals)
        addr = lambda t: None if t is None else _ptr(t).value
        self._check(self.lib.gwd_window_map_multi(vpn(*[addr(t) for t in srcs]), vpn(*[addr(t) for t in dsts]), vpn(*[addr(t) for t in res]),
                                                  i32n(*[int(c) for c in Cs]), n, B, H, W, shift, int(gather), dtype_code(srcs[0]),
                                                  self._stream(*srcs, *dsts)), "gwd_window_map_multi")

    def sqnorm(self, g, sq, n):
        self._check(self.lib.gwd_sqnorm(_ptr(g), _ptr(sq), n, self._stream(g, sq)), "gwd_sqnorm")

    def adamw_step(self, p, g, m, v, p16, sq, n, lr, b1, b2, eps, wd, bc1, bc2, max_norm, grad_scale):
        self._check(self.lib.gwd_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(p16), _ptr(sq), n, lr, b1, b2, eps,
                                            wd, bc1, bc2, max_norm, grad_scale, self._stream(p, g, m, v)),
                    "gwd_adamw_step")


_LIB = None


def library():
    """The process-wide device library; raises HipUnavailable when it cannot be loaded."""
    global _LIB
    if _LIB is None:
        _LIB = HipLibrary()
    return _LIB


def set_library(lib):
    """TEST HOOK: install a stand-in device library (tests/fake_device.py) or None to reset."""
    global _LIB
    _LIB = lib
