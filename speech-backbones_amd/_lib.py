"""ctypes binding of libgradtts_gfx950.so (C ABI: include/gradtts_abi.h).

PyTorch is only plumbing here: it owns device memory (packed weights, workspace, tensors) and the stream.
There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import weakref
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GTTS_LIB", os.path.join(_HERE, "libgradtts_gfx950.so"))   # GTTS_LIB: tuning variants

PREC_BF16X3 = 0
PREC_BF16 = 1
PREC_BF16_STORE = 2
PREC_F16F8 = 3          # fp16 hi*hi + both cross terms in one fp8 MFMA on the 3x3 Block convolutions (ABI 5); fp32-grade like bf16x3

_lib = None
_lock = threading.Lock()


class UnetCfg(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int), ("n_feats", ctypes.c_int), ("n_spks", ctypes.c_int),
                ("spk_emb_dim", ctypes.c_int), ("groups", ctypes.c_int), ("pe_scale", ctypes.c_float),
                ("beta_min", ctypes.c_float), ("beta_max", ctypes.c_float), ("precision", ctypes.c_int),
                ("keep_intermediates", ctypes.c_int), ("arch", ctypes.c_int), ("dim_cond", ctypes.c_int),
                ("use_ref_t", ctypes.c_int), ("c_dim", ctypes.c_int), ("vc_beta_min", ctypes.c_double),
                ("vc_beta_max", ctypes.c_double), ("conv_ws", ctypes.c_int)]


class EncCfg(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("n_vocab", ctypes.c_int), ("n_feats", ctypes.c_int), ("channels", ctypes.c_int),
                ("filter_channels", ctypes.c_int), ("filter_channels_dp", ctypes.c_int), ("n_heads", ctypes.c_int),
                ("n_layers", ctypes.c_int), ("kernel_size", ctypes.c_int), ("window_size", ctypes.c_int)]


class VocCfg(ctypes.Structure):
    _fields_ = [("n_mels", ctypes.c_int), ("upsample_initial_channel", ctypes.c_int), ("n_ups", ctypes.c_int),
                ("upsample_rates", ctypes.c_int * 8), ("upsample_kernel_sizes", ctypes.c_int * 8),
                ("n_kernels", ctypes.c_int), ("resblock_kernel_sizes", ctypes.c_int * 8),
                ("resblock_dilations", (ctypes.c_int * 3) * 8), ("resblock_type", ctypes.c_int)]


class PackItem(ctypes.Structure):           # gtts_pack_item
    _fields_ = [("w", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("kind", ctypes.c_int), ("cin", ctypes.c_int),
                ("cout", ctypes.c_int), ("transposed", ctypes.c_int)]


def lib():
    """Load the HIP library (once).  Fails loudly when it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libgradtts_gfx950.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python speech-backbones_amd/build.py`. There is no CPU/PyTorch fallback for the sampling path."
                % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, i, f, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
        L.gtts_abi_version.restype = i
        L.gtts_last_error.restype = ctypes.c_char_p
        L.gtts_plan_create.argtypes = [ctypes.POINTER(UnetCfg), ctypes.POINTER(vp)]
        L.gtts_plan_destroy.argtypes = [vp]
        L.gtts_plan_destroy.restype = None
        L.gtts_plan_num_params.argtypes = [vp]
        L.gtts_plan_param_info.argtypes = [vp, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i),
                                           ctypes.POINTER(i * 4)]
        L.gtts_packed_weight_bytes.argtypes = [vp]
        L.gtts_packed_weight_bytes.restype = sz
        L.gtts_workspace_bytes.argtypes = [vp, i, i]
        L.gtts_workspace_bytes.restype = sz
        L.gtts_plan_set_streams.argtypes = [vp, ctypes.POINTER(vp), i]
        L.gtts_plan_set_graph.argtypes = [vp, i]
        L.gtts_mas_maximum_path_cpu.argtypes = [vp, vp, vp, vp, vp, i, i, i]
        L.gtts_bcast_weights.argtypes = [vp, sz, i, vp, vp]
        L.gtts_pack_weights.argtypes = [vp, ctypes.POINTER(vp), i, vp, vp, vp]
        L.gtts_estimator_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i, i, vp]
        L.gtts_euler_step.argtypes = [vp, vp, vp, vp, vp, f, f, i, i, i, vp]
        L.gtts_reverse_diffusion.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i, i, i, i, i, vp]
        L.gtts_mas_scratch_bytes.argtypes = [i, i, i]
        L.gtts_mas_scratch_bytes.restype = sz
        L.gtts_mas_maximum_path.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, vp]
        L.gtts_expand_alignment.argtypes = [vp, vp, vp, vp, vp, f, vp, vp, vp, i, i, i, i, vp]
        L.gtts_log_prior.argtypes = [vp, vp, vp, i, i, i, i, vp]
        L.gtts_diffusion_noising.argtypes = [vp, vp, vp, vp, vp, f, f, vp, vp, i, i, i, vp]
        L.gtts_score_loss_partials.argtypes = [i, i, i]
        L.gtts_score_loss_partials.restype = sz
        L.gtts_score_loss.argtypes = [vp, vp, vp, f, f, f, vp, vp, i, i, i, vp]
        L.gtts_conv3x3_packed_bytes.argtypes = [i, i]
        L.gtts_conv3x3_packed_bytes.restype = sz
        L.gtts_conv3x3_pack.argtypes = [vp, vp, i, i, i, vp]
        L.gtts_conv3x3_masked.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.gtts_conv3x3_wgrad.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.gtts_conv3x3_wgrad_workspace_bytes.argtypes = [i, i, i, i, i]
        L.gtts_conv3x3_wgrad_workspace_bytes.restype = sz
        L.gtts_conv3x3_wgrad_tiled.argtypes = [vp, vp, vp, vp, vp, vp, sz, i, i, i, i, i, vp]
        L.gtts_conv1x1_packed_bytes.argtypes = [i, i]
        L.gtts_conv1x1_packed_bytes.restype = sz
        L.gtts_conv1x1_pack.argtypes = [vp, vp, i, i, i, vp]
        L.gtts_conv1x1_masked.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.gtts_conv1x1_wgrad_workspace_bytes.argtypes = [i, i, i, i, i]
        L.gtts_conv1x1_wgrad_workspace_bytes.restype = sz
        L.gtts_conv1x1_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, sz, i, i, i, i, i, vp]
        L.gtts_conv3x3_masked2.argtypes = [vp, vp, i, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.gtts_conv3x3_masked3.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.gtts_conv3x3_wgrad_tiled2.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, sz, i, i, i, i, i, vp]
        L.gtts_conv_resample_packed_bytes.argtypes = [i, i, i]
        L.gtts_conv_resample_packed_bytes.restype = sz
        L.gtts_conv_resample_pack.argtypes = [vp, vp, i, i, i, vp]
        L.gtts_conv_resample.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
        L.gtts_gn_mish_forward_tb.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, f, vp]
        L.gtts_gn_mish_backward_tb.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.gtts_add_masked.argtypes = [vp, vp, vp, vp, i, i, i, i, i, vp]
        L.gtts_conv_wgrad_small_scratch_floats.argtypes = [i, i, i, i]
        L.gtts_conv_wgrad_small_scratch_floats.restype = sz
        L.gtts_conv_wgrad_small.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
        L.gtts_zero_insert2.argtypes = [vp, vp, i, i, i, i, vp]
        L.gtts_space_to_depth2.argtypes = [vp, vp, i, i, i, i, vp]
        L.gtts_final_conv_forward.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
        L.gtts_final_conv_scratch_floats.argtypes = [i, i, i, i]
        L.gtts_final_conv_scratch_floats.restype = sz
        L.gtts_final_conv_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp]
        L.gtts_attn_train_scratch_floats.argtypes = [i, i]
        L.gtts_attn_train_scratch_floats.restype = sz
        L.gtts_attn_train_forward.argtypes = [vp, vp, vp, vp, vp, i, i, vp]
        L.gtts_attn_train_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
        L.gtts_rezero_forward.argtypes = [vp, vp, vp, vp, sz, vp]
        L.gtts_rezero_scratch_bytes.argtypes = [sz]
        L.gtts_rezero_scratch_bytes.restype = sz
        L.gtts_rezero_backward.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
        L.gtts_gn_mish_forward.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, f, vp]
        L.gtts_gn_mish_stats_floats.argtypes = [i, i]
        L.gtts_gn_mish_stats_floats.restype = sz
        L.gtts_gn_mish_scratch_bytes.argtypes = [i, i]
        L.gtts_gn_mish_scratch_bytes.restype = sz
        L.gtts_gn_mish_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.gtts_enc_create.argtypes = [ctypes.POINTER(EncCfg), ctypes.POINTER(vp)]
        L.gtts_enc_destroy.argtypes = [vp]
        L.gtts_enc_destroy.restype = None
        L.gtts_enc_num_params.argtypes = [vp]
        L.gtts_enc_param_info.argtypes = [vp, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i), ctypes.POINTER(i * 4)]
        L.gtts_enc_packed_bytes.argtypes = [vp]
        L.gtts_enc_packed_bytes.restype = sz
        L.gtts_enc_pack.argtypes = [vp, ctypes.POINTER(vp), i, vp, vp]
        L.gtts_enc_workspace_bytes.argtypes = [vp, i, i]
        L.gtts_enc_workspace_bytes.restype = sz
        L.gtts_enc_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, i, i, vp]
        L.gtts_postnet_create.argtypes = [i, i, i, ctypes.POINTER(vp)]
        L.gtts_postnet_destroy.argtypes = [vp]
        L.gtts_postnet_destroy.restype = None
        L.gtts_postnet_num_params.argtypes = [vp]
        L.gtts_postnet_param_info.argtypes = [vp, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i), ctypes.POINTER(i * 4)]
        L.gtts_postnet_packed_bytes.argtypes = [vp]
        L.gtts_postnet_packed_bytes.restype = sz
        L.gtts_postnet_pack.argtypes = [vp, ctypes.POINTER(vp), i, vp, vp]
        L.gtts_postnet_workspace_bytes.argtypes = [vp, i, i]
        L.gtts_postnet_workspace_bytes.restype = sz
        L.gtts_postnet_forward.argtypes = [vp, vp, vp, vp, vp, vp, sz, i, i, vp]
        L.gtts_voc_create.argtypes = [ctypes.POINTER(VocCfg), ctypes.POINTER(vp)]
        L.gtts_voc_destroy.argtypes = [vp]
        L.gtts_voc_destroy.restype = None
        L.gtts_voc_num_params.argtypes = [vp]
        L.gtts_voc_param_info.argtypes = [vp, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i), ctypes.POINTER(i * 4)]
        L.gtts_voc_packed_bytes.argtypes = [vp]
        L.gtts_voc_packed_bytes.restype = sz
        L.gtts_voc_pack.argtypes = [vp, ctypes.POINTER(vp), i, vp, vp]
        L.gtts_voc_workspace_bytes.argtypes = [vp, i, i]
        L.gtts_voc_workspace_bytes.restype = sz
        L.gtts_voc_hop.argtypes = [vp]
        L.gtts_voc_forward.argtypes = [vp, vp, vp, vp, vp, sz, i, i, vp]
        L.gtts_plan_num_tensors.argtypes = [vp]
        L.gtts_plan_tensor_info.argtypes = [vp, i, i, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz),
                                            ctypes.POINTER(i * 4)]
        L.gtts_vc_workspace_bytes.argtypes = [vp, i, i, i]
        L.gtts_vc_workspace_bytes.restype = sz
        L.gtts_vc_estimator_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i, i, i, vp]
        L.gtts_vc_reverse_diffusion.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i, i, i, i, i, i, i, vp]
        L.gtts_vc_tensor_info.argtypes = [vp, i, i, i, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz),
                                          ctypes.POINTER(i * 4)]
        L.gtts_plan_num_ops.argtypes = [vp]
        L.gtts_plan_op_info.argtypes = [vp, i, i, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.gtts_profile_enable.argtypes = [vp, i]
        L.gtts_profile_collect.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]
        L.gtts_profile_timeline.argtypes = [vp, i, ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i)]
        L.gtts_pack_batch_desc_bytes.argtypes = [i]
        L.gtts_pack_batch_desc_bytes.restype = sz
        L.gtts_pack_batch_describe.argtypes = [ctypes.POINTER(PackItem), i, vp, ctypes.POINTER(i)]
        L.gtts_pack_batch.argtypes = [vp, i, i, vp]
        L.gtts_ubench_mfma_out_floats.argtypes = [i]
        L.gtts_ubench_mfma_out_floats.restype = sz
        L.gtts_ubench_mfma.argtypes = [vp, sz, vp, i, i, ctypes.POINTER(ctypes.c_double), vp]
        L.gtts_ubench_hbm.argtypes = [vp, vp, vp, sz, i, i, ctypes.POINTER(ctypes.c_double), vp]
        L.gtts_workspace_status.argtypes = [vp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(f), vp]
        L.gtts_in_glu_stats_floats.argtypes = [i, i]
        L.gtts_in_glu_stats_floats.restype = sz
        L.gtts_in_glu_scratch_floats.argtypes = [i, i]
        L.gtts_in_glu_scratch_floats.restype = sz
        L.gtts_in_glu_forward.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, f, vp]
        L.gtts_in_glu_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp]
        if L.gtts_abi_version() != 6:
            raise RuntimeError("libgradtts_gfx950.so ABI version mismatch")
        _lib = L
        return _lib


class RangeError(RuntimeError):
    """GTTS_E_RANGE: a value lies outside what the plan's precision represents (f16f8: a 3x3 conv weight with |w| >= 63.97)."""


def _check(rc, what):
    if rc != 0:
        exc = RangeError if rc == -7 else RuntimeError
        raise exc("%s failed (%d): %s" % (what, rc, lib().gtts_last_error().decode()))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """Context that makes `device` current for the launches inside -- free when it already is (the training wrappers run
    ~600 launches per step; torch.cuda.device() costs more host time than the launch it guards)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s must live on a HIP device (got %s); the sampling path has no CPU fallback" %
                           (name, t.device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class Plan:
    """Host-side plan of one score U-Net (mirrors GradLogPEstimator2d.__init__, diffusion.py:129-172)."""

    def __init__(self, dim=64, n_feats=80, n_spks=1, spk_emb_dim=64, groups=8, pe_scale=1000.0, beta_min=0.05,
                 beta_max=20.0, precision=PREC_BF16X3, keep_intermediates=False, arch=0, dim_cond=128, use_ref_t=True,
                 c_dim=256, streams=None, conv_ws=None):
        """arch=0: Grad-TTS GradLogPEstimator2d; arch=1: DiffVC GradLogPEstimator (dim = dim_base).

        conv_ws: which kernel runs the wide Block 3x3 convolutions in bf16x3 (gtts_unet_cfg.conv_ws).  False: the uniform-wave
        kernel of conv_mfma.hip, which shares CUs with other streams' kernels.  True: the persistent wave-specialised kernel of
        conv_ws.hip -- 11 % faster per launch, but it owns every CU's registers while it runs.  None (default): True where these
        convolutions dominate (dim >= 128: the DiffVC decoder, +6 % end to end), False on the Grad-TTS dim-64 network, where a
        third of the call is bandwidth-bound kernels that three sub-batch streams hide under the convolutions of another
        sub-batch (measured on one box: 7.05 ms per U-Net call against 7.47 with the persistent kernel on two streams).

        precision PREC_F16F8 (the drop-in modules' default): the 128-channel-and-wider Block convolutions run the f16 + fp8 split on
        the persistent kernel (conv_ws defaults to True), unsplit -- measured on one box, ms per U-Net call at B = 16: 6.68 against
        7.16 for bf16x3 on three streams; with the persistent kernel two sub-batch streams are slower (6.80 vs 6.64), and sub-batch
        streams confined to disjoint halves of the CUs (hipExtStreamCreateWithCUMask) slower still (7.07; bf16x3: 7.70 vs 7.16).

        streams: number of sub-batches gtts_reverse_diffusion runs side by side on torch side streams owned by this
        object and registered with gtts_plan_set_streams (0 / 1: no split; $GTTS_STREAMS overrides the default).  Default 3;
        2 with conv_ws in bf16x3 (8 + 8 utterances fill the chip in whole rounds of persistent workgroups); 0 with PREC_F16F8."""
        if conv_ws is None:
            conv_ws = int(dim) >= 128 or int(precision) == PREC_F16F8
        conv_ws = bool(conv_ws) and int(precision) in (PREC_BF16X3, PREC_F16F8)
        self._kw = dict(dim=dim, n_feats=n_feats, n_spks=n_spks, spk_emb_dim=spk_emb_dim, groups=groups,
                        pe_scale=pe_scale, beta_min=beta_min, beta_max=beta_max, precision=precision,
                        keep_intermediates=keep_intermediates, arch=arch, dim_cond=dim_cond, use_ref_t=use_ref_t,
                        c_dim=c_dim, streams=streams, conv_ws=conv_ws)
        self.conv_ws = conv_ws
        self.cfg = UnetCfg(int(dim), int(n_feats), int(n_spks), int(spk_emb_dim), int(groups), float(pe_scale),
                           float(beta_min), float(beta_max), int(precision), 1 if keep_intermediates else 0, int(arch),
                           int(dim_cond), 1 if use_ref_t else 0, int(c_dim), float(beta_min), float(beta_max),
                           1 if conv_ws else 0)
        self._h = ctypes.c_void_p()
        _check(lib().gtts_plan_create(ctypes.byref(self.cfg), ctypes.byref(self._h)), "gtts_plan_create")
        self._ws = {}
        if streams is None:
            streams = int(os.environ.get("GTTS_STREAMS", ("0" if int(precision) == PREC_F16F8 else "2") if conv_ws else "3"))
        self._nstreams = 0 if int(streams) < 2 else min(int(streams), 4)
        self._side = None           # (device, [torch.cuda.Stream])
        self._graph = False
        self._gstream = None
        self._stage = {}            # graph mode: persistent argument buffers (stable addresses -> graph cache hits)

    # a Plan is host metadata: copies / pickles rebuild it from its constructor arguments (EMA deep copies,
    # torch.save(model) of a module that already sampled)
    def __reduce__(self):
        return (_rebuild_plan, (self._kw,))

    def __deepcopy__(self, memo):
        return Plan(**self._kw)

    def set_graph(self, on=True):
        """hipGraph replay of whole reverse_diffusion calls (launch-bound small batches).  Inputs are copied into
        persistent buffers owned by this object so that every call presents the same addresses to the library."""
        _check(lib().gtts_plan_set_graph(self._h, 1 if on else 0), "gtts_plan_set_graph")
        self._graph = bool(on)
        self._stage = {}
        self._gstream = None        # capture / replay stream (stream capture is not allowed on the default stream)

    def _use_streams(self, device):
        """Register this plan's side streams for `device` (created once; they belong to this object)."""
        if self._nstreams < 2:
            return
        if self._side is not None and self._side[0] == device:
            return
        side = [torch.cuda.Stream(device=device) for _ in range(self._nstreams)]
        arr = (ctypes.c_void_p * len(side))(*[s.cuda_stream for s in side])
        _check(lib().gtts_plan_set_streams(self._h, arr, len(side)), "gtts_plan_set_streams")
        self._side = (device, side)
        self._ws.clear()            # the workspace size depends on the number of sub-batches

    def __del__(self):
        try:
            if self._h:
                lib().gtts_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- state_dict layout
    def param_layout(self):
        L = lib()
        out = []
        for k in range(L.gtts_plan_num_params(self._h)):
            name, rank, dims = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int * 4)()
            _check(L.gtts_plan_param_info(self._h, k, ctypes.byref(name), ctypes.byref(rank), ctypes.byref(dims)),
                   "gtts_plan_param_info")
            out.append((name.value.decode(), tuple(dims[:rank.value])))
        return out

    def packed_bytes(self):
        return int(lib().gtts_packed_weight_bytes(self._h))

    def workspace_bytes(self, B, T):
        n = int(lib().gtts_workspace_bytes(self._h, int(B), int(T)))
        if n == 0:
            raise RuntimeError("gtts_workspace_bytes: %s" % lib().gtts_last_error().decode())
        return n

    def workspace(self, B, T, device):
        key = (int(B), int(T), str(device))
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()      # one shape at a time: the caching allocator recycles the old block
            ws = torch.empty(self.workspace_bytes(B, T), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def range_status(self, device=None):
        """(events, max |x|) of the activation range record of the last estimator / sampler call (PREC_F16F8: staging lanes x
        launches that split an activation with |x| >= 1024, which keeps fp16-grade cross terms only; always (0, 0.0) in the other
        precisions).  Synchronises the current stream."""
        ws = None
        for key, w in self._ws.items():
            if device is None or key[-1] == str(device):
                ws = w
        if ws is None:
            return 0, 0.0
        n, mx = ctypes.c_uint(0), ctypes.c_float(0.0)
        with torch.cuda.device(ws.device):
            _check(lib().gtts_workspace_status(_ptr(ws), ctypes.byref(n), ctypes.byref(mx), _stream()), "gtts_workspace_status")
        return int(n.value), float(mx.value)

    # ---- weights
    def pack(self, state, device):
        """state: mapping name -> tensor with the estimator-level names of the reference state_dict.
        PREC_F16F8: raises RangeError when a 3x3 Block-convolution weight does not fit the format (|w| >= 63.97)."""
        layout = self.param_layout()
        keep = []
        for name, shape in layout:
            if name not in state:
                raise RuntimeError("state_dict is missing '%s'" % name)
            t = state[name].detach().to(device=device, dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise RuntimeError("parameter %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
            keep.append(t)
        arr = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
        half = self.cfg.dim // 2
        # exactly SinusoidalPosEmb's frequency table, computed on the host CPU like the reference (diffusion.py:121-122)
        import math
        freq = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1))).to(device)
        blob = torch.empty(self.packed_bytes(), dtype=torch.uint8, device=device)
        with torch.cuda.device(blob.device):
            _check(lib().gtts_pack_weights(self._h, arr, len(keep), _ptr(freq), _ptr(blob), _stream()),
                   "gtts_pack_weights")
            torch.cuda.current_stream().synchronize()     # sources in `keep` may be temporaries
        return blob

    # ---- GradLogPEstimator2d.forward
    def estimator_forward(self, blob, x, mask, mu, t, spk=None):
        x, mask, mu, t, spk = (_f32c(x, "x"), _f32c(mask, "mask"), _f32c(mu, "mu"), _f32c(t, "t"),
                               _f32c(spk, "spk"))
        B, F, T = x.shape
        if F != self.cfg.n_feats:
            raise RuntimeError("expected %d mel bins, got %d" % (self.cfg.n_feats, F))
        if mask.numel() != B * T or mu.shape != x.shape or t.numel() != B:
            raise RuntimeError("shape mismatch: x %s mask %s mu %s t %s" % (tuple(x.shape), tuple(mask.shape),
                                                                         tuple(mu.shape), tuple(t.shape)))
        out = torch.empty_like(x)
        ws = self.workspace(B, T, x.device)
        with torch.cuda.device(x.device):
            _check(lib().gtts_estimator_forward(self._h, _ptr(blob), _ptr(x), _ptr(mask), _ptr(mu), _ptr(t), _ptr(spk),
                                                _ptr(out), _ptr(ws), ws.numel(), B, T, _stream()),
                   "gtts_estimator_forward")
        return out

    # ---- Diffusion.reverse_diffusion (whole loop)
    def reverse_diffusion(self, blob, z, mask, mu, n_timesteps, spk=None, noise=None):
        z, mask, mu, spk, noise = (_f32c(z, "z"), _f32c(mask, "mask"), _f32c(mu, "mu"), _f32c(spk, "spk"),
                                   _f32c(noise, "noise"))
        B, F, T = z.shape
        if F != self.cfg.n_feats:
            raise RuntimeError("expected %d mel bins, got %d" % (self.cfg.n_feats, F))
        if mask.numel() != B * T or mu.shape != z.shape:
            raise RuntimeError("shape mismatch: z %s mask %s mu %s" % (tuple(z.shape), tuple(mask.shape), tuple(mu.shape)))
        if noise is not None and tuple(noise.shape) != (int(n_timesteps), B, F, T):
            raise RuntimeError("noise must be [n_timesteps, B, F, T]")
        self._use_streams(z.device)
        ws = self.workspace(B, T, z.device)
        if self._graph:
            key = (B, T, str(z.device), spk is not None, None if noise is None else tuple(noise.shape))
            st = self._stage.get(key)
            if st is None:
                self._stage.clear()
                st = {"z": torch.empty_like(z), "mask": torch.empty_like(mask), "mu": torch.empty_like(mu),
                      "out": torch.empty_like(z), "spk": None if spk is None else torch.empty_like(spk),
                      "noise": None if noise is None else torch.empty_like(noise)}
                self._stage[key] = st
            for k, v in (("z", z), ("mask", mask), ("mu", mu), ("spk", spk), ("noise", noise)):
                if v is not None:
                    st[k].copy_(v)
            z, mask, mu, spk, noise, out = st["z"], st["mask"], st["mu"], st["spk"], st["noise"], st["out"]
        else:
            out = torch.empty_like(z)
        with torch.cuda.device(z.device):
            if self._graph:
                cur = torch.cuda.current_stream()
                if self._gstream is None or self._gstream.device != z.device:
                    self._gstream = torch.cuda.Stream(device=z.device)
                self._gstream.wait_stream(cur)
                with torch.cuda.stream(self._gstream):
                    _check(lib().gtts_reverse_diffusion(self._h, _ptr(blob), _ptr(z), _ptr(mask), _ptr(mu), _ptr(spk),
                                                        _ptr(noise), _ptr(out), _ptr(ws), ws.numel(), B, T, int(n_timesteps),
                                                        0, int(n_timesteps), _stream()), "gtts_reverse_diffusion")
                cur.wait_stream(self._gstream)
                return out.clone()
            _check(lib().gtts_reverse_diffusion(self._h, _ptr(blob), _ptr(z), _ptr(mask), _ptr(mu), _ptr(spk),
                                                _ptr(noise), _ptr(out), _ptr(ws), ws.numel(), B, T, int(n_timesteps),
                                                0, int(n_timesteps), _stream()), "gtts_reverse_diffusion")
        return out

    # ---- DiffVC (arch=1)
    def vc_workspace(self, B, T, Tr, device):
        key = ("vc", int(B), int(T), int(Tr), str(device))
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            n = int(lib().gtts_vc_workspace_bytes(self._h, int(B), int(T), int(Tr)))
            if n == 0:
                raise RuntimeError("gtts_vc_workspace_bytes: %s" % lib().gtts_last_error().decode())
            ws = torch.empty(n, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def vc_estimator_forward(self, blob, x, x_mask, mean, xt_ref, ref_mask, c, t):
        """DiffVC GradLogPEstimator.forward(x, x_mask, mean, ref, ref_mask, c, t) (DiffVC/model/diffusion.py:61-106)."""
        x, x_mask, mean, xt_ref, ref_mask, c, t = (_f32c(v, n) for v, n in (
            (x, "x"), (x_mask, "x_mask"), (mean, "mean"), (xt_ref, "ref"), (ref_mask, "ref_mask"), (c, "c"), (t, "t")))
        B, F, T = x.shape
        Tr = int(ref_mask.shape[-1]) if ref_mask is not None else T
        out = torch.empty_like(x)
        ws = self.vc_workspace(B, T, Tr, x.device)
        with torch.cuda.device(x.device):
            _check(lib().gtts_vc_estimator_forward(self._h, _ptr(blob), _ptr(x), _ptr(x_mask), _ptr(mean), _ptr(xt_ref),
                                                   _ptr(ref_mask), _ptr(c), _ptr(t), _ptr(out), _ptr(ws), ws.numel(), B, T, Tr,
                                                   _stream()), "gtts_vc_estimator_forward")
        return out

    def vc_reverse_diffusion(self, blob, z, mask, mean, ref, ref_mask, mean_ref, c, n_timesteps, mode, noise=None,
                             noise_chunk_bytes=256 << 20):
        """DiffVC Diffusion.reverse_diffusion (DiffVC/model/diffusion.py:164-196); mode in {'pf','em','ml'}.

        noise (em / ml): a [n_timesteps, B, F, T] tensor, or a callable `draw(k)` returning the next k steps' N(0,1)
        draws as [k, B, F, T] -- then the loop runs in step ranges of at most noise_chunk_bytes of noise, so long
        schedules never hold more than one chunk (the reference holds one [B,F,T] draw at a time)."""
        modes = {"pf": 0, "em": 1, "ml": 2}
        if mode not in modes:
            raise RuntimeError("mode must be one of pf / em / ml")
        draw = noise if callable(noise) else None
        z, mask, mean, ref, ref_mask, mean_ref, c = (_f32c(v, n) for v, n in (
            (z, "z"), (mask, "mask"), (mean, "mean"), (ref, "ref"), (ref_mask, "ref_mask"), (mean_ref, "mean_ref"), (c, "c")))
        B, F, T = z.shape
        N = int(n_timesteps)
        Tr = int(ref_mask.shape[-1]) if ref_mask is not None else T
        if draw is None:
            noise = _f32c(noise, "noise")
            if mode != "pf" and (noise is None or tuple(noise.shape) != (N, B, F, T)):
                raise RuntimeError("em / ml sampling needs noise of shape [n_timesteps, B, F, T] (or a callable)")
        out = torch.empty_like(z)
        ws = self.vc_workspace(B, T, Tr, z.device)
        per = max(1, int(noise_chunk_bytes) // max(1, B * F * T * 4)) if (draw is not None and mode != "pf") else N
        with torch.cuda.device(z.device):
            i0 = 0
            while i0 < N:
                i1 = min(N, i0 + per)
                nz = None
                if mode != "pf":
                    nz = _f32c(draw(i1 - i0), "noise") if draw is not None else noise[i0:i1]
                    if tuple(nz.shape) != (i1 - i0, B, F, T):
                        raise RuntimeError("noise chunk must be [%d, B, F, T]" % (i1 - i0))
                _check(lib().gtts_vc_reverse_diffusion(self._h, _ptr(blob), _ptr(z), _ptr(mask), _ptr(mean), _ptr(ref),
                                                       _ptr(ref_mask), _ptr(mean_ref), _ptr(c), _ptr(nz), _ptr(out), _ptr(ws),
                                                       ws.numel(), B, T, Tr, N, modes[mode], i0, i1, _stream()),
                       "gtts_vc_reverse_diffusion")
                i0 = i1
        return out

    def vc_tensors(self, B, T, Tr, device):
        """Named intermediates of the last DiffVC estimator call (keep_intermediates plans); ref tensors use T_ref."""
        L = lib()
        ws = self.vc_workspace(B, T, Tr, device)
        out = {}
        for k in range(L.gtts_plan_num_tensors(self._h)):
            name, off, dims = ctypes.c_char_p(), ctypes.c_size_t(), (ctypes.c_int * 4)()
            _check(L.gtts_vc_tensor_info(self._h, k, int(B), int(T), int(Tr), ctypes.byref(name), ctypes.byref(off),
                                         ctypes.byref(dims)), "gtts_vc_tensor_info")
            out[name.value.decode()] = (off.value, tuple(dims))
        return ws, out

    # ---- measurement (bench.py): per-op HIP-event timing
    def ops(self, B, T):
        """[(label, kernel, algorithmic flops, algorithmic bytes)] of the launches of one estimator call."""
        L = lib()
        out = []
        for k in range(L.gtts_plan_num_ops(self._h)):
            label, kern, fl, by = ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double()
            _check(L.gtts_plan_op_info(self._h, k, int(B), int(T), ctypes.byref(label), ctypes.byref(kern),
                                       ctypes.byref(fl), ctypes.byref(by)), "gtts_plan_op_info")
            out.append((label.value.decode(), kern.value.decode(), fl.value, by.value))
        return out

    def profile(self, on):
        """on = True / 1: per-op durations (the sampler runs unsplit); on = 2: timeline mode (sub-batch streams stay on, see
        profile_timeline); False / 0: off."""
        _check(lib().gtts_profile_enable(self._h, 2 if on == 2 else (1 if on else 0)), "gtts_profile_enable")

    def profile_timeline(self, cap=1 << 18):
        """[(op index, stream index, start ms, end ms)] of every launch recorded in timeline mode; clears the record."""
        op, st = (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
        t0, t1 = (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
        n = ctypes.c_int(0)
        _check(lib().gtts_profile_timeline(self._h, cap, op, st, t0, t1, ctypes.byref(n)), "gtts_profile_timeline")
        return [(op[k], st[k], t0[k], t1[k]) for k in range(n.value)]

    def profile_collect(self):
        """(ms per op, launches per op) accumulated since profiling was enabled; clears the record."""
        n = lib().gtts_plan_num_ops(self._h)
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_longlong * n)()
        _check(lib().gtts_profile_collect(self._h, ms, cnt), "gtts_profile_collect")
        return list(ms), list(cnt)

    # ---- debugging: named intermediates (keep_intermediates plans)
    def tensors(self, B, T, device):
        if int(self.cfg.precision) == PREC_BF16_STORE:
            raise RuntimeError("Plan.tensors(): the named-intermediate views are fp32; a keep_intermediates plan with bf16 "
                               "activation storage stores 2-byte activations -- use PREC_BF16 / PREC_BF16X3 for tap tests")
        L = lib()
        ws = self.workspace(B, T, device)
        out = {}
        for k in range(L.gtts_plan_num_tensors(self._h)):
            name, off, dims = ctypes.c_char_p(), ctypes.c_size_t(), (ctypes.c_int * 4)()
            _check(L.gtts_plan_tensor_info(self._h, k, int(B), int(T), ctypes.byref(name), ctypes.byref(off),
                                           ctypes.byref(dims)), "gtts_plan_tensor_info")
            n = dims[0] * dims[1] * dims[2] * dims[3]
            if n <= 0:
                continue
            view = ws[off.value: off.value + 4 * n].view(torch.float32).view(*dims)
            out[name.value.decode()] = view
        return out


class Vocoder:
    """HiFi-GAN generator on the HIP kernels (csrc/voc.hip): Generator(h).forward of Grad-TTS/hifi-gan/models.py:77-120."""

    def __init__(self, upsample_rates=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4), upsample_initial_channel=512,
                 resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)), resblock="1",
                 n_mels=80):
        self._kw = dict(upsample_rates=tuple(upsample_rates), upsample_kernel_sizes=tuple(upsample_kernel_sizes),
                        upsample_initial_channel=int(upsample_initial_channel),
                        resblock_kernel_sizes=tuple(resblock_kernel_sizes),
                        resblock_dilation_sizes=tuple(tuple(d) for d in resblock_dilation_sizes), resblock=str(resblock),
                        n_mels=int(n_mels))
        cfg = VocCfg()
        cfg.n_mels = int(n_mels)
        cfg.upsample_initial_channel = int(upsample_initial_channel)
        cfg.n_ups = len(upsample_rates)
        cfg.n_kernels = len(resblock_kernel_sizes)
        cfg.resblock_type = int(resblock)
        if cfg.n_ups > 8 or cfg.n_kernels > 8:
            raise RuntimeError("at most 8 upsamplers / ResBlock kernels")
        for k, (u, ks) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cfg.upsample_rates[k] = int(u)
            cfg.upsample_kernel_sizes[k] = int(ks)
        want = 3 if str(resblock) == "1" else 2       # ResBlock1 walks three dilations, ResBlock2 two (models.py:11-74)
        for k, (ks, dil) in enumerate(zip(resblock_kernel_sizes, resblock_dilation_sizes)):
            cfg.resblock_kernel_sizes[k] = int(ks)
            if len(dil) != want:
                raise RuntimeError("resblock_dilation_sizes[%d] has %d entries; the HIP ResBlock%s runs exactly %d (the torch "
                                   "module would accept any count, the kernel does not)" % (k, len(dil), resblock, want))
            for j in range(3):
                cfg.resblock_dilations[k][j] = int(dil[j]) if j < len(dil) else 1
        self.cfg = cfg
        self._h = ctypes.c_void_p()
        L = lib()
        _check(L.gtts_voc_create(ctypes.byref(cfg), ctypes.byref(self._h)), "gtts_voc_create")
        self.hop = int(L.gtts_voc_hop(self._h))
        self._ws = {}

    @classmethod
    def from_config(cls, h):
        """h: the AttrDict of hifigan-config.json (Grad-TTS/inference.py:57-59)."""
        return cls(h["upsample_rates"], h["upsample_kernel_sizes"], h["upsample_initial_channel"],
                   h["resblock_kernel_sizes"], h["resblock_dilation_sizes"], h["resblock"], h.get("num_mels", 80))

    def __reduce__(self):
        return (_rebuild_voc, (self._kw,))

    def __deepcopy__(self, memo):
        return Vocoder(**self._kw)

    def __del__(self):
        try:
            if self._h:
                lib().gtts_voc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def param_layout(self):
        L = lib()
        out = []
        for k in range(L.gtts_voc_num_params(self._h)):
            name, rank, dims = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int * 4)()
            _check(L.gtts_voc_param_info(self._h, k, ctypes.byref(name), ctypes.byref(rank), ctypes.byref(dims)),
                   "gtts_voc_param_info")
            out.append((name.value.decode(), tuple(dims[:rank.value])))
        return out

    def pack(self, state, device):
        """state: name -> tensor with weight normalisation folded (`<module>.weight`, `<module>.bias`)."""
        keep = []
        for name, shape in self.param_layout():
            if name not in state:
                raise RuntimeError("state_dict is missing '%s' (call remove_weight_norm() or fold weight_g / weight_v)" % name)
            t = state[name].detach().to(device=device, dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise RuntimeError("parameter %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
            keep.append(t)
        arr = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
        blob = torch.empty(int(lib().gtts_voc_packed_bytes(self._h)), dtype=torch.uint8, device=device)
        with torch.cuda.device(blob.device):
            _check(lib().gtts_voc_pack(self._h, arr, len(keep), _ptr(blob), _stream()), "gtts_voc_pack")
            torch.cuda.current_stream().synchronize()
        return blob

    def forward(self, blob, mel):
        mel = _f32c(mel, "mel")
        B, F, T = mel.shape
        if F != self.cfg.n_mels:
            raise RuntimeError("expected %d mel bins, got %d" % (self.cfg.n_mels, F))
        key = (B, T, str(mel.device))
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            ws = torch.empty(int(lib().gtts_voc_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=mel.device)
            self._ws[key] = ws
        wav = torch.empty((B, 1, T * self.hop), dtype=torch.float32, device=mel.device)
        with torch.cuda.device(mel.device):
            _check(lib().gtts_voc_forward(self._h, _ptr(blob), _ptr(mel), _ptr(wav), _ptr(ws), ws.numel(), B, T, _stream()),
                   "gtts_voc_forward")
        return wav


class Encoder:
    """Grad-TTS TextEncoder (mode 'text') / DiffVC MelEncoder (mode 'mel') on the HIP kernels (csrc/enc.hip)."""

    def __init__(self, mode="text", n_vocab=149, n_feats=80, channels=192, filter_channels=768, filter_channels_dp=256,
                 n_heads=2, n_layers=6, kernel_size=3, window_size=4):
        self._kw = dict(mode=mode, n_vocab=n_vocab, n_feats=n_feats, channels=channels, filter_channels=filter_channels,
                        filter_channels_dp=filter_channels_dp, n_heads=n_heads, n_layers=n_layers, kernel_size=kernel_size,
                        window_size=window_size)
        self.mode = mode
        self.cfg = EncCfg({"text": 0, "mel": 1}[mode], int(n_vocab), int(n_feats), int(channels), int(filter_channels),
                          int(filter_channels_dp), int(n_heads), int(n_layers), int(kernel_size), int(window_size or 0))
        self._h = ctypes.c_void_p()
        _check(lib().gtts_enc_create(ctypes.byref(self.cfg), ctypes.byref(self._h)), "gtts_enc_create")
        self._ws = {}

    def __reduce__(self):
        return (_rebuild_enc, (self._kw,))

    def __deepcopy__(self, memo):
        return Encoder(**self._kw)

    def __del__(self):
        try:
            if self._h:
                lib().gtts_enc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def param_layout(self):
        L = lib()
        out = []
        for k in range(L.gtts_enc_num_params(self._h)):
            name, rank, dims = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int * 4)()
            _check(L.gtts_enc_param_info(self._h, k, ctypes.byref(name), ctypes.byref(rank), ctypes.byref(dims)),
                   "gtts_enc_param_info")
            out.append((name.value.decode(), tuple(dims[:rank.value])))
        return out

    def pack(self, state, device):
        keep = []
        for name, shape in self.param_layout():
            if name not in state:
                raise RuntimeError("state_dict is missing '%s'" % name)
            t = state[name].detach().to(device=device, dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise RuntimeError("parameter %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
            keep.append(t)
        arr = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
        blob = torch.empty(int(lib().gtts_enc_packed_bytes(self._h)), dtype=torch.uint8, device=device)
        with torch.cuda.device(blob.device):
            _check(lib().gtts_enc_pack(self._h, arr, len(keep), _ptr(blob), _stream()), "gtts_enc_pack")
            torch.cuda.current_stream().synchronize()
        return blob

    def _workspace(self, B, L, device):
        key = (B, L, str(device))
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            ws = torch.empty(int(lib().gtts_enc_workspace_bytes(self._h, B, L)), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def forward(self, blob, x, x_mask):
        """text: x = ids [B,L] int64 -> (mu [B,n_feats,L], logw [B,1,L]);  mel: x = mel [B,n_feats,L] -> [B,n_feats,L].
        x_mask [B,1,L] or [B,L] float."""
        if not x.is_cuda:
            raise RuntimeError("the encoder kernels need HIP tensors (got %s); there is no CPU fallback here" % x.device)
        m = _f32c(x_mask, "x_mask")
        if self.mode == "text":
            ids = x.to(torch.int64).contiguous()
            B, L = ids.shape
            mel = None
        else:
            mel = _f32c(x, "mel")
            B, _, L = mel.shape
            ids = None
        if m.numel() != B * L:
            raise RuntimeError("x_mask must have B*L elements")
        dev = x.device
        mu = torch.empty((B, self.cfg.n_feats, L), dtype=torch.float32, device=dev)
        logw = torch.empty((B, 1, L), dtype=torch.float32, device=dev) if self.mode == "text" else None
        ws = self._workspace(B, L, dev)
        with torch.cuda.device(dev):
            _check(lib().gtts_enc_forward(self._h, _ptr(blob), _ptr(ids), _ptr(mel), _ptr(m), _ptr(mu), _ptr(logw), _ptr(ws),
                                          ws.numel(), B, L, _stream()), "gtts_enc_forward")
        return (mu, logw) if self.mode == "text" else mu


class PostNetPlan:
    """DiffVC PostNet (DiffVC/model/postnet.py:40-53) on the HIP kernels (csrc/postnet.hip)."""

    def __init__(self, dim=128, n_feats=80, groups=8):
        self._kw = dict(dim=int(dim), n_feats=int(n_feats), groups=int(groups))
        self.dim, self.n_feats = int(dim), int(n_feats)
        self._h = ctypes.c_void_p()
        _check(lib().gtts_postnet_create(int(dim), int(n_feats), int(groups), ctypes.byref(self._h)), "gtts_postnet_create")
        self._ws = {}

    def __reduce__(self):
        return (_rebuild_postnet, (self._kw,))

    def __deepcopy__(self, memo):
        return PostNetPlan(**self._kw)

    def __del__(self):
        try:
            if self._h:
                lib().gtts_postnet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def param_layout(self):
        L = lib()
        out = []
        for k in range(L.gtts_postnet_num_params(self._h)):
            name, rank, dims = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int * 4)()
            _check(L.gtts_postnet_param_info(self._h, k, ctypes.byref(name), ctypes.byref(rank), ctypes.byref(dims)),
                   "gtts_postnet_param_info")
            out.append((name.value.decode(), tuple(dims[:rank.value])))
        return out

    def pack(self, state, device):
        keep = []
        for name, shape in self.param_layout():
            if name not in state:
                raise RuntimeError("state_dict is missing '%s'" % name)
            t = state[name].detach().to(device=device, dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise RuntimeError("parameter %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
            keep.append(t)
        arr = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
        blob = torch.empty(int(lib().gtts_postnet_packed_bytes(self._h)), dtype=torch.uint8, device=device)
        with torch.cuda.device(blob.device):
            _check(lib().gtts_postnet_pack(self._h, arr, len(keep), _ptr(blob), _stream()), "gtts_postnet_pack")
            torch.cuda.current_stream().synchronize()
        return blob

    def forward(self, blob, x, mask):
        """x [B,n_feats,T], mask [B,1,T] -> [B,n_feats,T]."""
        x, mask = _f32c(x, "x"), _f32c(mask, "mask")
        B, F, T = x.shape
        if F != self.n_feats or mask.numel() != B * T:
            raise RuntimeError("shape mismatch: x %s mask %s" % (tuple(x.shape), tuple(mask.shape)))
        key = (B, T, str(x.device))
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            ws = torch.empty(int(lib().gtts_postnet_workspace_bytes(self._h, B, T)), dtype=torch.uint8, device=x.device)
            self._ws[key] = ws
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _check(lib().gtts_postnet_forward(self._h, _ptr(blob), _ptr(x), _ptr(mask), _ptr(out), _ptr(ws), ws.numel(), B, T,
                                              _stream()), "gtts_postnet_forward")
        return out


def _rebuild_postnet(kw):
    return PostNetPlan(**kw)


def _rebuild_enc(kw):
    return Encoder(**kw)


def _rebuild_voc(kw):
    return Vocoder(**kw)


def _rebuild_plan(kw):
    return Plan(**kw)


def euler_step(xt, mu, est, mask, beta_t, h, noise=None):
    """In-place update of xt (one step of Diffusion.reverse_diffusion, diffusion.py:264-274)."""
    if not xt.is_cuda or xt.dtype != torch.float32 or not xt.is_contiguous():
        raise RuntimeError("xt must be a contiguous fp32 HIP tensor (updated in place)")
    mu, est, mask, noise = _f32c(mu, "mu"), _f32c(est, "est"), _f32c(mask, "mask"), _f32c(noise, "noise")
    B, F, T = xt.shape
    with torch.cuda.device(xt.device):
        _check(lib().gtts_euler_step(_ptr(xt), _ptr(mu), _ptr(est), _ptr(mask), _ptr(noise), float(beta_t), float(h),
                                     B, F, T, _stream()), "gtts_euler_step")
    return xt


def mas_maximum_path(value, mask):
    """monotonic_align.maximum_path(value, mask) (value, mask: [b, t_x, t_y]).

    HIP tensors run the GPU kernel.  Host tensors run the library's C++ twin gtts_mas_maximum_path_cpu -- the
    reference's wrapper accepts tensors on any device and always runs its Cython kernel on the host
    (monotonic_align/__init__.py:8-23); both are bit-identical to it."""
    if not value.is_cuda:
        v = value.detach().float().contiguous()
        m = mask.detach().to(dtype=torch.float32).contiguous()
        b, tx, ty = v.shape
        t_x = m.sum(1)[:, 0].to(torch.int32).contiguous()
        t_y = m.sum(2)[:, 0].to(torch.int32).contiguous()
        path = torch.empty((b, tx, ty), dtype=torch.int32)
        _check(lib().gtts_mas_maximum_path_cpu(_ptr(v), _ptr(m), _ptr(t_x), _ptr(t_y), _ptr(path), b, tx, ty),
               "gtts_mas_maximum_path_cpu")
        return path.to(dtype=value.dtype)
    v = value.detach().float().contiguous()
    m = mask.detach().to(device=v.device, dtype=torch.float32).contiguous()
    b, tx, ty = v.shape
    t_x = m.sum(1)[:, 0].to(torch.int32).contiguous()       # __init__.py:20-21
    t_y = m.sum(2)[:, 0].to(torch.int32).contiguous()
    path = torch.empty((b, tx, ty), dtype=torch.int32, device=v.device)
    scratch = torch.empty(int(lib().gtts_mas_scratch_bytes(b, tx, ty)), dtype=torch.uint8, device=v.device)
    with torch.cuda.device(v.device):
        _check(lib().gtts_mas_maximum_path(_ptr(v), _ptr(m), _ptr(t_x), _ptr(t_y), _ptr(path), _ptr(scratch), b, tx,
                                           ty, _stream()), "gtts_mas_maximum_path")
    return path.to(dtype=value.dtype)


# ---- training hot path (csrc/train.hip): raw kernels; the autograd wiring lives in model/_train_ops.py
_MAX_TENSOR_ELEMS = 1 << 29      # per call: B * max(cin, cout) * H * W (32-bit byte offsets inside the kernels; train_wgrad.hip)


def conv_size_ok(B, cin, cout, H, W):
    """The training kernels address one call's tensors with 32-bit byte offsets: larger shapes must take the torch path."""
    return int(B) * max(int(cin), int(cout)) * int(H) * int(W) < _MAX_TENSOR_ELEMS


def conv3x3_supported(cin, cout, need_dgrad=True, shape=None):
    """Channel counts (and, with shape = (B, H, W), tensor sizes) the training conv kernels take (forward / data gradient /
    weight gradient).  The first layer (the stacked 2- or 3-plane input, no data gradient wanted) has its own weight-gradient
    kernel."""
    def tiles(c):
        return c == 64 or (c > 64 and c % 128 == 0)
    if shape is not None and not conv_size_ok(shape[0], cin, cout, shape[1], shape[2]):
        return False
    if cin in (2, 3) and not need_dgrad:
        return tiles(cout)
    return cin % 32 == 0 and cout % 32 == 0 and tiles(cin) and tiles(cout)


_PACKED = {}       # (id(weight), transposed, kind) -> (weakref to the weight, its version, pack generation, packed blob)
_PACK_GEN = 0      # a packed copy is shared only by the forward and the backward of ONE estimator call (new_pack_generation)
_CONSTS = {}       # (device, kind, n) -> ones / zeros of the data-gradient call


def new_pack_generation():
    """Called at every entry of the training estimator (model/_train_ops.py): packed weight copies made before this point are
    not reused.  Tensor._version does not see writes through `p.data` (EMA swaps, manual SGD, `p.data = t`); with the generation
    in the validity check a stale pack cannot outlive the step that made it, at the price of one re-pack (microseconds) per
    convolution and step."""
    global _PACK_GEN
    _PACK_GEN += 1


def clear_packed_cache():
    """Drop every cached packed training weight (Diffusion.invalidate_packed and the load_state_dict hook call this)."""
    global _PACK_GEN
    _PACK_GEN += 1
    _PACKED.clear()


def _packed_weight(weight, cin, cout, transposed, kind):
    """Packed (fragment-order, bf16 hi / lo) copy of a convolution weight, cached per Parameter OBJECT and version: the entry is
    valid only while the very same tensor object is alive and unmodified (an optimizer step bumps `_version`; a data pointer
    alone can be recycled by the caching allocator for a different weight of the same shape)."""
    key = (id(weight), bool(transposed), kind)
    hit = _PACKED.get(key)
    if (hit is not None and hit[0]() is weight and hit[1] == int(weight._version) and hit[2] == _PACK_GEN and
            hit[3].device == weight.device):
        return hit[3]
    L = lib()
    if kind in ("3x3", "1x1"):
        nbytes, pack = ((L.gtts_conv3x3_packed_bytes, L.gtts_conv3x3_pack) if kind == "3x3" else (L.gtts_conv1x1_packed_bytes, L.gtts_conv1x1_pack))
        packed = torch.empty(int(nbytes(cin, cout)), dtype=torch.uint8, device=weight.device)
        _check(pack(_ptr(weight), _ptr(packed), cin, cout, 1 if transposed else 0, _stream()), "gtts_conv%s_pack" % kind)
    else:
        # "dn" Downsample forward, "up" Upsample forward, "dn_T" Downsample's data gradient: an Upsample of dy with the forward
        # weight [cout][cin][3][3] zero-padded to [.][.][4][4] (ConvTranspose2d layout: [in = cout][out = cin])
        up = 0 if kind == "dn" else 1
        src = torch.nn.functional.pad(weight, (0, 1, 0, 1)).contiguous() if kind == "dn_T" else weight
        packed = torch.empty(int(L.gtts_conv_resample_packed_bytes(cin, cout, up)), dtype=torch.uint8, device=weight.device)
        _check(L.gtts_conv_resample_pack(_ptr(src), _ptr(packed), cin, cout, up, _stream()), "gtts_conv_resample_pack")
    if len(_PACKED) >= 512:          # (two entries per convolution: ~100 for the Grad-TTS U-Net); dead entries go with the sweep
        for k in [k for k, v in _PACKED.items() if v[0]() is None]:
            del _PACKED[k]
        if len(_PACKED) >= 512:
            _PACKED.clear()
    _PACKED[key] = (weakref.ref(weight), int(weight._version), _PACK_GEN, packed)
    return packed


def _packed_conv3x3(weight, cin, cout, transposed):
    return _packed_weight(weight, cin, cout, transposed, "3x3")


_KIND_CODE = {"3x3": 0, "1x1": 1, "dn": 2, "up": 3, "dn_T": 4}


class PackSpecs(list):
    """[(weight, cin, cout, transposed, kind)] of one module tree, plus the batched-pack plan built for it (device descriptor table +
    the blobs it fills).  The plan hangs off THIS list, and the list off the estimator it was built from (_train_ops._pack_specs), so
    weights, blobs and table die with the estimator; there is no module-global table of plans."""
    plan = None


def _build_pack_plan(specs, dev, sig):
    L = lib()
    n = len(specs)
    items = (PackItem * n)()
    blobs = []
    with _on(dev):
        for k, (w, ci, co, t, kind) in enumerate(specs):
            if kind == "3x3":
                nb = L.gtts_conv3x3_packed_bytes(int(ci), int(co))
            elif kind == "1x1":
                nb = L.gtts_conv1x1_packed_bytes(int(ci), int(co))
            else:
                nb = L.gtts_conv_resample_packed_bytes(int(ci), int(co), 0 if kind == "dn" else 1)
            blob = torch.empty(int(nb), dtype=torch.uint8, device=dev)
            blobs.append(blob)
            items[k].w, items[k].packed = w.data_ptr(), blob.data_ptr()
            items[k].kind, items[k].cin, items[k].cout, items[k].transposed = _KIND_CODE[kind], int(ci), int(co), 1 if t else 0
        nbytes = int(L.gtts_pack_batch_desc_bytes(n))
        host = (ctypes.c_ubyte * nbytes)()
        grid = ctypes.c_int(0)
        _check(L.gtts_pack_batch_describe(items, n, ctypes.cast(host, ctypes.c_void_p), ctypes.byref(grid)), "gtts_pack_batch_describe")
        desc = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
    # (weights are referenced weakly here: the spec list is what keeps them alive)
    return {"desc": desc, "n": n, "grid": int(grid.value), "sig": sig,
            "entries": [((id(w), bool(t), kind), weakref.ref(w), blob) for (w, ci, co, t, kind), blob in zip(specs, blobs)]}


def prepack(specs):
    """All weight packs of one training step in ONE launch (gtts_pack_batch).  specs: [(weight, cin, cout, transposed, kind)] with the
    meanings of _packed_weight (cin / cout of the convolution being packed).  With a PackSpecs list the blobs are allocated once and
    re-filled in place at every call (a plain list gets a plan per call); the per-weight cache entries are stamped with the current pack
    generation, so the autograd Functions of this step find them and nothing older survives.  Call after new_pack_generation(), on
    the stream the step runs on."""
    if not specs:
        return
    L = lib()
    dev = specs[0][0].device
    sig = tuple([w.data_ptr() for w, *_ in specs])          # (cheap per-step check: the addresses the descriptor table holds)
    plan = getattr(specs, "plan", None)
    if plan is None or plan["sig"] != sig or plan["n"] != len(specs):
        plan = _build_pack_plan(specs, dev, sig)
        if isinstance(specs, PackSpecs):
            specs.plan = plan
    with _on(dev):
        _check(L.gtts_pack_batch(_ptr(plan["desc"]), plan["n"], plan["grid"], _stream()), "gtts_pack_batch")
    gen = _PACK_GEN
    for (key, ref, blob), spec in zip(plan["entries"], specs):
        w = spec[0]
        _PACKED[key] = (ref if ref() is w else weakref.ref(w), w._version, gen, blob)


def _const(device, kind, *shape):
    key = (str(device), kind) + shape
    v = _CONSTS.get(key)
    if v is None:
        v = (torch.ones if kind == "ones" else torch.zeros)(shape, dtype=torch.float32, device=device)
        _CONSTS[key] = v
    return v


def _conv3x3_run(x, mask_cols, weight, bias, transposed, x1=None, out_mask=None):
    B, c0, H, W = x.shape
    cin = c0 + (int(x1.shape[1]) if x1 is not None else 0)
    cout = weight.shape[1] if transposed else weight.shape[0]
    L = lib()
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    with _on(x.device):
        packed = _packed_conv3x3(weight, cin, cout, transposed)
        _check(L.gtts_conv3x3_masked3(_ptr(x), _ptr(x1), c0, _ptr(mask_cols), _ptr(out_mask), _ptr(packed), _ptr(bias), _ptr(y), B, cin,
                                      cout, H, W, _stream()), "gtts_conv3x3_masked")
    return y


def conv3x3_masked(x, mask_cols, weight, bias, x1=None):
    """Conv2d_3x3(cat(x, x1) * mask) + bias (Block.forward, diffusion.py:56-57; the concatenation of the up path, :166, is read in
    place): x [B,c0,H,W], x1 [B,c1,H,W] or None, mask_cols [B,W], weight [cout,c0+c1,3,3]."""
    x, mask_cols, weight, bias = _f32c(x, "x"), _f32c(mask_cols, "mask"), _f32c(weight, "weight"), _f32c(bias, "bias")
    return _conv3x3_run(x, mask_cols, weight, bias, False, _f32c(x1, "x1"))


def conv3x3_dgrad(dy, weight, mask_cols=None):
    """Gradient of conv3x3_masked w.r.t. (x * mask): a 3x3 convolution of dy with the transposed, flipped weights; with
    mask_cols [B,W] the gradient w.r.t. x itself (the mask is applied in the kernel's epilogue)."""
    dy, weight = _f32c(dy, "dy"), _f32c(weight, "weight")
    B, cout, H, W = dy.shape
    return _conv3x3_run(dy, _const(dy.device, "ones", B, W), weight, _const(dy.device, "zeros", int(weight.shape[1])), True,
                        out_mask=_f32c(mask_cols, "mask"))


def conv3x3_wgrad(x, mask_cols, dy, x1=None):
    """(dW [cout,cin,3,3], db [cout]) of conv3x3_masked (x1: second source of a concatenated input, c0 a multiple of 64)."""
    x, mask_cols, dy, x1 = _f32c(x, "x"), _f32c(mask_cols, "mask"), _f32c(dy, "dy"), _f32c(x1, "x1")
    B, c0, H, W = x.shape
    cin = c0 + (int(x1.shape[1]) if x1 is not None else 0)
    cout = dy.shape[1]
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device)
    with _on(x.device):
        if cin in (2, 3) and x1 is None:
            _wgrad_small(x, mask_cols, dy, dw, db, 3)
        elif cin % 64 == 0 and cout % 64 == 0:       # LDS-tiled deterministic reduction (train_wgrad.hip)
            nws = int(lib().gtts_conv3x3_wgrad_workspace_bytes(B, cin, cout, H, W))
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
            _check(lib().gtts_conv3x3_wgrad_tiled2(_ptr(x), _ptr(x1), c0, _ptr(mask_cols), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), nws,
                                                   B, cin, cout, H, W, _stream()), "gtts_conv3x3_wgrad_tiled")
        else:
            if x1 is not None:
                x = torch.cat((x, x1), 1)
            _check(lib().gtts_conv3x3_wgrad(_ptr(x), _ptr(mask_cols), _ptr(dy), _ptr(dw), _ptr(db), B, cin, cout, H, W, _stream()),
                   "gtts_conv3x3_wgrad")
    return dw, db


def conv1x1_supported(cin, cout, need_dgrad=True, shape=None):
    """Channel counts (and, with shape = (B, H, W), tensor sizes) the 1x1 training kernels take: forward cout (and, for the
    data gradient, cin) a whole number of the kernel's output tiles; the weight gradient whole 64 x 64 tiles."""
    def tiles(c):
        return c == 64 or (c > 64 and c % 128 == 0)
    if shape is not None and not conv_size_ok(shape[0], cin, cout, shape[1], shape[2]):
        return False
    if cin in (2, 3) and not need_dgrad:          # first layer: own weight-gradient kernel
        return tiles(cout)
    return tiles(cout) and cin % 64 == 0 and (tiles(cin) or not need_dgrad)


def _packed_conv1x1(weight, cin, cout, transposed):
    return _packed_weight(weight, cin, cout, transposed, "1x1")


def _conv1x1_run(x, mask_cols, weight, bias, transposed):
    B, cin, H, W = x.shape
    cout = weight.shape[1] if transposed else weight.shape[0]
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    with _on(x.device):
        packed = _packed_conv1x1(weight, cin, cout, transposed)
        _check(lib().gtts_conv1x1_masked(_ptr(x), _ptr(mask_cols), _ptr(packed), _ptr(bias), _ptr(y), B, cin, cout, H, W, _stream()),
               "gtts_conv1x1_masked")
    return y


def conv1x1_masked(x, mask_cols, weight, bias):
    """Conv2d_1x1(x * mask) + bias (res_conv / to_qkv / to_out, diffusion.py:70,87-88): x [B,cin,H,W], mask_cols [B,W] or None,
    weight [cout,cin,1,1], bias [cout] or None."""
    x, weight = _f32c(x, "x"), _f32c(weight, "weight")
    B, W = int(x.shape[0]), int(x.shape[3])
    mask_cols = _const(x.device, "ones", B, W) if mask_cols is None else _f32c(mask_cols, "mask")
    bias = _const(x.device, "zeros", int(weight.shape[0])) if bias is None else _f32c(bias, "bias")
    return _conv1x1_run(x, mask_cols, weight, bias, False)


def conv1x1_dgrad(dy, weight, mask_cols=None):
    """Gradient of conv1x1_masked w.r.t. x: the 1x1 convolution of dy * mask (columns do not mix) with the transposed weight."""
    dy, weight = _f32c(dy, "dy"), _f32c(weight, "weight")
    B, W = int(dy.shape[0]), int(dy.shape[3])
    mask_cols = _const(dy.device, "ones", B, W) if mask_cols is None else _f32c(mask_cols, "mask")
    return _conv1x1_run(dy, mask_cols, weight, _const(dy.device, "zeros", int(weight.shape[1])), True)


def _wgrad_small(x, mask_cols, dy, dw, db, ksize):
    B, cin, H, W = x.shape
    cout = int(dy.shape[1])
    scratch = torch.empty(int(lib().gtts_conv_wgrad_small_scratch_floats(B, cin, cout, ksize)), dtype=torch.float32, device=x.device)
    _check(lib().gtts_conv_wgrad_small(_ptr(x), _ptr(mask_cols), _ptr(dy), _ptr(dw), _ptr(db), _ptr(scratch), B, cin, cout, H, W, ksize,
                                       _stream()), "gtts_conv_wgrad_small")


def conv1x1_wgrad(x, mask_cols, dy, want_bias=True):
    """(dW [cout,cin,1,1], db [cout] or None) of conv1x1_masked."""
    x, dy = _f32c(x, "x"), _f32c(dy, "dy")
    B, cin, H, W = x.shape
    cout = int(dy.shape[1])
    dw = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_bias else None
    if cin in (2, 3):
        with _on(x.device):
            _wgrad_small(x, _const(x.device, "ones", B, W) if mask_cols is None else _f32c(mask_cols, "mask"), dy, dw, db, 1)
        return dw, db
    with _on(x.device):
        nws = int(lib().gtts_conv1x1_wgrad_workspace_bytes(B, cin, cout, H, W))
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        _check(lib().gtts_conv1x1_wgrad(_ptr(x), _ptr(_f32c(mask_cols, "mask")) if mask_cols is not None else None, _ptr(dy), _ptr(dw),
                                        _ptr(db) if db is not None else None, _ptr(ws), nws, B, cin, cout, H, W, _stream()),
               "gtts_conv1x1_wgrad")
    return dw, db


def add_masked(a, b, mask_cols, channels=None):
    """a + b * mask (a None: b * mask; mask_cols None: a + b) over [B,C,H,W] with mask_cols [B,W].  channels = (c_begin, c_end):
    b is that channel range of a wider contiguous tensor (read in place; the result is contiguous)."""
    a, b, mask_cols = _f32c(a, "a"), _f32c(b, "b"), _f32c(mask_cols, "mask")
    B, Cb, H, W = b.shape
    c0, c1 = (0, Cb) if channels is None else channels
    out = torch.empty((B, c1 - c0, H, W), dtype=torch.float32, device=b.device)
    with _on(b.device):
        _check(lib().gtts_add_masked(_ptr(a), ctypes.c_void_p(b.data_ptr() + 4 * c0 * H * W), _ptr(mask_cols), _ptr(out), B, c1 - c0, H, W,
                                     0 if channels is None else Cb, _stream()), "gtts_add_masked")
    return out


def zero_insert2(x):
    """[B,C,h,w] -> [B,C,2h,2w] with x at the even positions and zeros elsewhere."""
    x = _f32c(x, "x")
    B, C, h, w = x.shape
    out = torch.empty((B, C, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _check(lib().gtts_zero_insert2(_ptr(x), _ptr(out), B, C, h, w, _stream()), "gtts_zero_insert2")
    return out


def space_to_depth2(x):
    """[B,C,2h,2w] -> [B,4C,h,w]: channel block (pr * 2 + pc) holds rows 2y + 1 - pr and columns 2x + 1 - pc."""
    x = _f32c(x, "x")
    B, C, H2, W2 = x.shape
    if H2 % 2 or W2 % 2:
        raise ValueError("space_to_depth2: even height and width expected, got %d x %d" % (H2, W2))
    out = torch.empty((B, 4 * C, H2 // 2, W2 // 2), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _check(lib().gtts_space_to_depth2(_ptr(x), _ptr(out), B, C, H2 // 2, W2 // 2, _stream()), "gtts_space_to_depth2")
    return out


def final_conv_forward(x, weight, bias, mask_cols):
    """(Conv2d_1x1(x * mask) + bias) * mask for the 64 -> 1 final convolution (diffusion.py:175-176): [B,1,H,W]."""
    x, weight, bias, mask_cols = _f32c(x, "x"), _f32c(weight, "weight"), _f32c(bias, "bias"), _f32c(mask_cols, "mask")
    B, C, H, W = x.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _check(lib().gtts_final_conv_forward(_ptr(x), _ptr(weight), _ptr(bias), _ptr(mask_cols), _ptr(out), B, C, H, W, _stream()),
               "gtts_final_conv_forward")
    return out


def final_conv_backward(x, weight, mask_cols, dout):
    """(dx, dweight [1,C,1,1], dbias [1]) of final_conv_forward."""
    x, weight, mask_cols, dout = _f32c(x, "x"), _f32c(weight, "weight"), _f32c(mask_cols, "mask"), _f32c(dout, "dout")
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty((1, C, 1, 1), dtype=torch.float32, device=x.device)
    db = torch.empty((1,), dtype=torch.float32, device=x.device)
    with _on(x.device):
        scratch = torch.empty(int(lib().gtts_final_conv_scratch_floats(B, C, H, W)), dtype=torch.float32, device=x.device)
        _check(lib().gtts_final_conv_backward(_ptr(x), _ptr(weight), _ptr(mask_cols), _ptr(dout), _ptr(dx), _ptr(dw), _ptr(db),
                                              _ptr(scratch), B, C, H, W, _stream()), "gtts_final_conv_backward")
    return dx, dw, db


def resample_supported(cin, cout, H, W, up, B=1):
    def tiles(c):
        return c == 64 or (c > 64 and c % 128 == 0)
    # largest tensor of the call and of its gradients: Upsample's output (and its space_to_depth planes) is 4 cout planes of H x W
    if not conv_size_ok(B, 4 * max(cin, cout) if up else max(cin, cout), 1, H, W):
        return False
    return tiles(cin) and tiles(cout) and (up or (H % 2 == 0 and W % 2 == 0))


def conv_resample(x, mask_cols, weight, bias, up, dgrad_of_down=False):
    """Downsample (up False: Conv2d 3x3 stride 2 pad 1, weight [cout,cin,3,3]) or Upsample (up True: ConvTranspose2d 4x4 stride 2
    pad 1, weight [cin,cout,4,4]) of x * mask (diffusion.py:19-34).  dgrad_of_down: x is the gradient of a Downsample output and
    weight that Downsample's forward weight; the result is its data gradient (an Upsample with the zero-padded kernel)."""
    x, mask_cols, weight = _f32c(x, "x"), _f32c(mask_cols, "mask"), _f32c(weight, "weight")
    B, cin, H, W = x.shape
    if dgrad_of_down:
        cout, kind, up = int(weight.shape[1]), "dn_T", True
    else:
        cout, kind = (int(weight.shape[1]), "up") if up else (int(weight.shape[0]), "dn")
    bias = _const(x.device, "zeros", cout) if bias is None else _f32c(bias, "bias")
    y = torch.empty((B, cout, 2 * H, 2 * W) if up else (B, cout, H // 2, W // 2), dtype=torch.float32, device=x.device)
    with _on(x.device):
        packed = _packed_weight(weight, cin, cout, False, kind)
        _check(lib().gtts_conv_resample(_ptr(x), _ptr(mask_cols), _ptr(packed), _ptr(bias), _ptr(y), B, cin, cout, H, W, 1 if up else 0,
                                        _stream()), "gtts_conv_resample")
    return y


def attn_train_forward(qkv):
    """LinearAttention core (diffusion.py:90-100) on to_qkv's output qkv [B,384,H,W]: (out [B,128,H,W], ctx [B,4,32,32],
    stat [B,4,32,2] = softmax row maxima and reciprocal sums)."""
    qkv = _f32c(qkv, "qkv")
    B, C3, H, W = qkv.shape
    if C3 != 384:
        raise ValueError("attn_train_forward: to_qkv output must have 3 x 4 heads x 32 = 384 channels, got %d" % C3)
    N = H * W
    out = torch.empty((B, 128, H, W), dtype=torch.float32, device=qkv.device)
    ctx = torch.empty((B, 4, 32, 32), dtype=torch.float32, device=qkv.device)
    stat = torch.empty((B, 4, 32, 2), dtype=torch.float32, device=qkv.device)
    with _on(qkv.device):
        scratch = torch.empty(int(lib().gtts_attn_train_scratch_floats(B, N)), dtype=torch.float32, device=qkv.device)
        _check(lib().gtts_attn_train_forward(_ptr(qkv), _ptr(out), _ptr(ctx), _ptr(stat), _ptr(scratch), B, N, _stream()),
               "gtts_attn_train_forward")
    return out, ctx, stat


def attn_train_backward(qkv, dout, ctx, stat):
    """d qkv of attn_train_forward given d out."""
    qkv, dout = _f32c(qkv, "qkv"), _f32c(dout, "dout")
    B, C3, H, W = qkv.shape
    N = H * W
    dqkv = torch.empty_like(qkv)
    dctx = torch.empty((B, 4, 32, 32), dtype=torch.float32, device=qkv.device)
    rdot = torch.empty((B, 4, 32), dtype=torch.float32, device=qkv.device)
    with _on(qkv.device):
        scratch = torch.empty(int(lib().gtts_attn_train_scratch_floats(B, N)), dtype=torch.float32, device=qkv.device)
        _check(lib().gtts_attn_train_backward(_ptr(qkv), _ptr(dout), _ptr(ctx), _ptr(stat), _ptr(dqkv), _ptr(dctx), _ptr(rdot),
                                              _ptr(scratch), B, N, _stream()), "gtts_attn_train_backward")
    return dqkv


def rezero_forward(f, g, x):
    """f * g + x (Rezero + Residual, diffusion.py:40-46,103-108); g a 1-element device tensor."""
    f, x, g = _f32c(f, "f"), _f32c(x, "x"), _f32c(g, "g")
    y = torch.empty_like(f)
    with _on(f.device):
        _check(lib().gtts_rezero_forward(_ptr(f), _ptr(x), _ptr(g), _ptr(y), f.numel(), _stream()), "gtts_rezero_forward")
    return y


def rezero_backward(dy, f, g):
    """(d f = dy * g, d g = sum(dy * f)) of rezero_forward."""
    dy, f, g = _f32c(dy, "dy"), _f32c(f, "f"), _f32c(g, "g")
    df = torch.empty_like(f)
    dg = torch.empty_like(g)
    with _on(f.device):
        scratch = torch.empty(int(lib().gtts_rezero_scratch_bytes(f.numel())), dtype=torch.uint8, device=f.device)
        _check(lib().gtts_rezero_backward(_ptr(dy), _ptr(f), _ptr(g), _ptr(df), _ptr(dg), _ptr(scratch), f.numel(), _stream()),
               "gtts_rezero_backward")
    return df, dg


def gn_mish_forward(y, gamma, beta, mask_cols, groups, eps, tb=None):
    """(Mish(GroupNorm(y)) * mask [+ tb[:, :, None, None]], stats: (mean, rstd) pairs followed by reduction scratch) -- Block.forward
    after the convolution (diffusion.py:53-58) and, with tb [B,C], ResnetBlock's time term (diffusion.py:75-76)."""
    y, gamma, beta, mask_cols, tb = _f32c(y, "y"), _f32c(gamma, "gamma"), _f32c(beta, "beta"), _f32c(mask_cols, "mask"), _f32c(tb, "tb")
    B, C, H, W = y.shape
    out = torch.empty_like(y)
    stats = torch.empty(int(lib().gtts_gn_mish_stats_floats(B, int(groups))), dtype=torch.float32, device=y.device)
    with _on(y.device):
        _check(lib().gtts_gn_mish_forward_tb(_ptr(y), _ptr(gamma), _ptr(beta), _ptr(mask_cols), _ptr(tb), _ptr(out), _ptr(stats), B, C,
                                             H, W, int(groups), float(eps), _stream()), "gtts_gn_mish_forward")
    return out, stats


def gn_mish_backward(dout, y, gamma, beta, mask_cols, stats, groups, want_dtb=False):
    """(dy, dgamma, dbeta[, dtb [B,C]]) of gn_mish_forward."""
    dout, y = _f32c(dout, "dout"), _f32c(y, "y")
    B, C, H, W = y.shape
    dy = torch.empty_like(y)
    dg = torch.empty((C,), dtype=torch.float32, device=y.device)
    db = torch.empty((C,), dtype=torch.float32, device=y.device)
    dtb = torch.empty((B, C), dtype=torch.float32, device=y.device) if want_dtb else None
    scratch = torch.empty(int(lib().gtts_gn_mish_scratch_bytes(B, C)), dtype=torch.uint8, device=y.device)
    with _on(y.device):
        _check(lib().gtts_gn_mish_backward_tb(_ptr(dout), _ptr(y), _ptr(_f32c(gamma, "gamma")), _ptr(_f32c(beta, "beta")),
                                              _ptr(_f32c(mask_cols, "mask")), _ptr(stats), _ptr(dy), _ptr(dg), _ptr(db), _ptr(dtb),
                                              _ptr(scratch), B, C, H, W, int(groups), _stream()), "gtts_gn_mish_backward")
    return (dy, dg, db, dtb) if want_dtb else (dy, dg, db)


def in_glu_forward(y, gamma, beta, eps=1e-5):
    """(IN(y[:, :C]) * sigmoid(IN(y[:, C:])), stats) -- InstanceNorm2d(affine) + GLU(dim=1) of DiffVC's RefBlock (modules.py:128-157)."""
    y, gamma, beta = _f32c(y, "y"), _f32c(gamma, "gamma"), _f32c(beta, "beta")
    B, C2, H, W = y.shape
    C = C2 // 2
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=y.device)
    stats = torch.empty(int(lib().gtts_in_glu_stats_floats(B, C)), dtype=torch.float32, device=y.device)
    with _on(y.device):
        _check(lib().gtts_in_glu_forward(_ptr(y), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(stats), B, C, H, W, float(eps), _stream()),
               "gtts_in_glu_forward")
    return out, stats


def in_glu_backward(dout, y, gamma, beta, stats):
    """(dy, dgamma, dbeta) of in_glu_forward."""
    dout, y = _f32c(dout, "dout"), _f32c(y, "y")
    B, C2, H, W = y.shape
    C = C2 // 2
    dy = torch.empty_like(y)
    dg = torch.empty((C2,), dtype=torch.float32, device=y.device)
    db = torch.empty((C2,), dtype=torch.float32, device=y.device)
    scratch = torch.empty(int(lib().gtts_in_glu_scratch_floats(B, C)), dtype=torch.float32, device=y.device)
    with _on(y.device):
        _check(lib().gtts_in_glu_backward(_ptr(dout), _ptr(y), _ptr(_f32c(gamma, "gamma")), _ptr(_f32c(beta, "beta")), _ptr(stats),
                                          _ptr(dy), _ptr(dg), _ptr(db), _ptr(scratch), B, C, H, W, _stream()), "gtts_in_glu_backward")
    return dy, dg, db


def diffusion_noising(x0, mu, z, mask, t, beta_min, beta_max):
    """Diffusion.forward_diffusion given the N(0,1) draw z: (xt, z * mask)   (diffusion.py:244-252)."""
    x0, mu, z, mask, t = (_f32c(v, n) for v, n in ((x0, "x0"), (mu, "mu"), (z, "z"), (mask, "mask"), (t, "t")))
    B, F, T = x0.shape
    xt, zm = torch.empty_like(x0), torch.empty_like(x0)
    with _on(x0.device):
        _check(lib().gtts_diffusion_noising(_ptr(x0), _ptr(mu), _ptr(z), _ptr(mask), _ptr(t), float(beta_min), float(beta_max),
                                            _ptr(xt), _ptr(zm), B, F, T, _stream()), "gtts_diffusion_noising")
    return xt, zm


def score_loss(eps, z_masked, t, beta_min, beta_max, inv_denom, want_grad=True):
    """Diffusion.loss_t's reduction: (sum((eps s + z)^2) * inv_denom as a 0-d tensor, d loss / d eps or None).
    inv_denom: a Python float, or a 0-d device tensor (then nothing here synchronises with the host: the kernel runs with a
    unit normaliser and loss and gradient are scaled by tensor ops)."""
    if torch.is_tensor(inv_denom):
        loss, g = score_loss(eps, z_masked, t, beta_min, beta_max, 1.0, want_grad)
        return loss * inv_denom, (g * inv_denom if g is not None else None)
    eps, z_masked, t = _f32c(eps, "eps"), _f32c(z_masked, "z"), _f32c(t, "t")
    B, F, T = eps.shape
    n = int(lib().gtts_score_loss_partials(B, F, T))
    part = torch.empty(n, dtype=torch.float32, device=eps.device)
    g = torch.empty_like(eps) if want_grad else None
    with _on(eps.device):
        _check(lib().gtts_score_loss(_ptr(eps), _ptr(z_masked), _ptr(t), float(beta_min), float(beta_max), float(inv_denom),
                                     _ptr(part), _ptr(g), B, F, T, _stream()), "gtts_score_loss")
    return part.sum() * inv_denom, g


def log_prior(mu_x, y):
    """MAS score matrix of GradTTS.compute_loss (tts.py:130-139): [B,F,t_x], [B,F,T] -> [B,t_x,T], one launch."""
    if not mu_x.is_cuda:
        raise RuntimeError("log_prior needs HIP tensors; there is no CPU fallback")
    mx, yy = _f32c(mu_x.detach(), "mu_x"), _f32c(y.detach(), "y")
    B, F, tx = mx.shape
    if yy.shape[0] != B or yy.shape[1] != F:
        raise RuntimeError("mu_x [B,F,t_x] and y [B,F,T] disagree: %s vs %s" % (tuple(mx.shape), tuple(yy.shape)))
    T = yy.shape[2]
    out = torch.empty((B, tx, T), dtype=torch.float32, device=mx.device)
    with _on(mx.device):
        _check(lib().gtts_log_prior(_ptr(mx), _ptr(yy), _ptr(out), B, F, tx, T, _stream()), "gtts_log_prior")
    return out


def expand_alignment(duration, x_mask, y_lengths, mu_x, T, noise=None, temperature=1.0):
    """generate_path + mu_y = attn^T . mu_x + z = mu_y + noise / temperature in one launch (tts.py:84-94).

    duration, x_mask: [B, t_x]; y_lengths: [B] integer; mu_x: [B, F, t_x]; noise: [B, F, T] or None.
    Returns (attn [B, t_x, T], mu_y [B, F, T], z [B, F, T] or None), bit-identical to the reference's CPU path."""
    if not mu_x.is_cuda:
        raise RuntimeError("expand_alignment needs HIP tensors; there is no CPU fallback")
    d, m, mx = _f32c(duration, "duration"), _f32c(x_mask, "x_mask"), _f32c(mu_x, "mu_x")
    B, F, tx = mx.shape
    if d.shape != (B, tx) or m.shape != (B, tx):
        raise RuntimeError("duration / x_mask must be [B, t_x] = [%d, %d]" % (B, tx))
    yl = y_lengths.to(device=mx.device, dtype=torch.int32).contiguous()
    nz = _f32c(noise, "noise")
    if nz is not None and nz.shape != (B, F, T):
        raise RuntimeError("noise must be [B, F, T]")
    attn = torch.empty((B, tx, T), dtype=torch.float32, device=mx.device)
    mu_y = torch.empty((B, F, T), dtype=torch.float32, device=mx.device)
    z = torch.empty((B, F, T), dtype=torch.float32, device=mx.device) if nz is not None else None
    with _on(mx.device):
        _check(lib().gtts_expand_alignment(_ptr(d), _ptr(m), _ptr(yl), _ptr(mx), _ptr(nz), float(temperature), _ptr(attn),
                                           _ptr(mu_y), _ptr(z), B, F, tx, int(T), _stream()), "gtts_expand_alignment")
    return attn, mu_y, z


def measured_ceilings(device, seconds=2.0):
    """Measured ceilings of this chip (bench.py roofline): bf16 MFMA TFLOP/s on random / zero operands and HBM GB/s for a
    16-byte copy, triad and read-only sweep.  About `seconds` of GPU time in total.  Returns a dict."""
    L = lib()
    out = {}
    with _on(device):
        st = _stream()
        props = torch.cuda.get_device_properties(device)
        cus = int(props.multi_processor_count)
        wgs = cus * 2                                   # 2 workgroups x 4 waves per CU = 2 waves per SIMD
        g = torch.Generator(device="cpu").manual_seed(0)
        rnd = torch.randn(1 << 20, generator=g).to(torch.bfloat16).to(device)
        zero = torch.zeros(1 << 20, dtype=torch.bfloat16, device=device)
        sink = torch.empty(int(L.gtts_ubench_mfma_out_floats(wgs)), dtype=torch.float32, device=device)
        flops = ctypes.c_double(0.0)

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps

        iters = 4000                                    # 32 k MFMAs per wave: ~0.55 ms per launch at peak
        for name, src in (("random", rnd), ("zero", zero)):
            fn = lambda: _check(L.gtts_ubench_mfma(_ptr(src), ctypes.c_size_t(src.numel() * 2), _ptr(sink), wgs, iters,
                                                   ctypes.byref(flops), st), "gtts_ubench_mfma")
            t1 = timed(fn, 3)
            reps = max(3, int(seconds * 0.3 / max(t1, 1e-6)))      # long enough for the clock to settle at its power budget
            t = timed(fn, reps)
            out["mfma_bf16_tflops_%s" % name] = flops.value / t * 1e-12
        fn = lambda: _check(L.gtts_ubench_mfma(_ptr(rnd), ctypes.c_size_t(rnd.numel() * 2), _ptr(sink), wgs, -iters,
                                               ctypes.byref(flops), st), "gtts_ubench_mfma")
        t1 = timed(fn, 3)
        out["mfma_bf16_16x16x32_tflops_random"] = flops.value / timed(fn, max(3, int(seconds * 0.3 / max(t1, 1e-6)))) * 1e-12
        # what the vendor's GEMM library reaches on the same chip with the same kind of data (hipBLASLt through torch.matmul,
        # 8192^3 bf16): the practical ceiling of an LDS-fed MFMA kernel, beside the register-only stream above
        m = 8192
        ga = torch.randn(m, m, generator=g).to(torch.bfloat16).to(device)
        gb = torch.randn(m, m, generator=g).to(torch.bfloat16).to(device)
        t = timed(lambda: torch.matmul(ga, gb), 10)
        out["gemm_bf16_tflops_random_hipblaslt"] = 2.0 * m ** 3 / t * 1e-12
        del ga, gb
        n = 1 << 28                                     # 1 GiB per buffer: four times the Infinity Cache
        a = torch.empty(n, dtype=torch.float32, device=device).normal_()
        b = torch.empty(n, dtype=torch.float32, device=device).normal_()
        c = torch.empty(n, dtype=torch.float32, device=device)
        nbytes = ctypes.c_double(0.0)
        for name, mode in (("copy", 0), ("triad", 1), ("read", 2)):
            best = 0.0
            for nt in (0, 4):                           # plain and nontemporal accesses, a few grid sizes: keep the best
                for wpc in (4, 8, 16):
                    fn = lambda: _check(L.gtts_ubench_hbm(_ptr(a), _ptr(b), _ptr(c), ctypes.c_size_t(n), mode + nt, cus * wpc,
                                                          ctypes.byref(nbytes), st), "gtts_ubench_hbm")
                    t = timed(fn, 3)
                    gbs = nbytes.value / t * 1e-9
                    out.setdefault("hbm_detail", {})["%s_%s_wg%d" % (name, "nt" if nt else "plain", wpc)] = round(gbs, 1)
                    best = max(best, gbs)
            out["hbm_%s_gbs" % name] = best
        out["cus"] = cus
    return out
