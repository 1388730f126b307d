"""Host-side handle of the video UNet: the reference's `CondMargVideoModel` call surface
(improved_diffusion/unet.py:929-1026) over the HIP engine.

Nothing here computes: the object owns an engine handle (topology + packed weights on the GPU) and
marshals tensors across the C ABI.  Supported surface = what the sampling path uses
(scripts/video_sample.py:562-567,151-168): construction from the `create_video_model` keywords,
`load_state_dict`, `to`, `eval`, `parameters`, `state_dict` and `__call__` (Boundary A).
"""
import ctypes
import math
from collections import OrderedDict

import numpy as np
import torch as th

from . import _lib

_OBS_MODES = {"x_0": 0, "x_t": 1, "x_t_minus_1": 2}


class CondMargVideoModel:
    def __init__(self, *, T, image_size, model_channels, num_res_blocks, attention_resolutions, num_heads=1,
                 use_scale_shift_norm=False, use_spatial_encoding=False, use_frame_encoding=False,
                 cross_frame_attention=True, enforce_position_invariance=False, use_rpe_net=False,
                 bucket_params=None, cond_emb_type="channel", allow_interactions_between_padding=False,
                 in_channels=3, out_channels=3, dropout=0, num_classes=None, num_heads_upsample=-1,
                 use_checkpoint=False, temporal_augment_type=None, channel_mult=None, **unused):
        # unet.py:932-947: how the conditioning frames enter the stem ('-initzero' differs at initialisation only)
        cond = cond_emb_type.replace("-initzero", "")
        if cond == "channel":
            cond_mode = 0
        elif "duplicate" in cond_emb_type or "all" in cond_emb_type:
            cond_mode = 1
        elif cond_emb_type == "t=0":
            cond_mode = 2
        else:
            raise NotImplementedError(cond_emb_type)
        self.cond_emb_type = cond
        if not cross_frame_attention:
            raise NotImplementedError("cross_frame_attention=False")
        if num_classes is not None or out_channels not in (3, 6) or in_channels != 3:
            raise NotImplementedError("class conditioning / non-RGB inputs")
        if num_heads_upsample not in (-1, num_heads):
            raise NotImplementedError("num_heads_upsample != num_heads")
        self.T = T
        self.image_size = image_size
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.training = False
        self.device = th.device("cpu")
        cfg = _lib.VdConfig()
        cfg.image_size, cfg.num_channels, cfg.num_res_blocks, cfg.num_heads = image_size, model_channels, num_res_blocks, num_heads
        cfg.T = int(T)
        ads = list(attention_resolutions)
        cfg.n_attention_ds = len(ads)
        for i, d in enumerate(ads):
            cfg.attention_ds[i] = int(d)
        cfg.use_scale_shift_norm = int(bool(use_scale_shift_norm))
        cfg.use_spatial_encoding = int(bool(use_spatial_encoding))
        cfg.use_frame_encoding = int(bool(use_frame_encoding))
        cfg.enforce_position_invariance = int(bool(enforce_position_invariance))
        cfg.use_rpe_net = int(bool(use_rpe_net))
        cfg.allow_interactions_between_padding = int(bool(allow_interactions_between_padding))
        bp = bucket_params or dict(alpha=1, beta=1, gamma=1)
        assert bucket_params is not None                          # unet.py:423-427
        cfg.rp_alpha, cfg.rp_beta, cfg.rp_gamma = float(bp["alpha"]), float(bp["beta"]), float(bp["gamma"])
        cfg.time_embed_mult = 4
        cfg.cond_emb_type = cond_mode
        cfg.learn_sigma = int(out_channels == 6)
        self._cfg = cfg
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().vd_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._handle = h
        self._specs = self._read_specs()
        self._host_sd = None          # CPU copy kept until the weights are on the device
        self._wbuf = None             # packed device weights (a torch tensor: broadcastable over RCCL)
        self._wbuf_bwd = None         # backward-data image (use_gradient_method only: enable_guidance())
        self._bound_schedule = None
        self._pos_ch = _lib.lib().vd_pos_channels(h)
        self._use_frame_encoding = bool(use_frame_encoding)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.lib().vd_destroy(h)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass
            self._handle = None

    # -- parameters ----------------------------------------------------------------------------------
    def _read_specs(self):
        L = _lib.lib()
        out = []
        name = ctypes.create_string_buffer(256)
        nd = ctypes.c_int()
        shape = (ctypes.c_longlong * 4)()
        for i in range(L.vd_param_count(self._handle)):
            _lib.check(L.vd_param_info(self._handle, i, name, 256, ctypes.byref(nd), shape))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value))))
        return out

    def param_specs(self):
        """[(checkpoint key, shape)] in the reference's state_dict order."""
        return list(self._specs)

    def load_state_dict(self, state_dict, strict=True):
        """nn.Module.load_state_dict semantics: missing / unexpected keys raise RuntimeError when strict."""
        want = dict(self._specs)
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        errs = []
        if strict and missing:
            errs.append("Missing key(s) in state_dict: " + ", ".join(repr(k) for k in missing))
        if strict and unexpected:
            errs.append("Unexpected key(s) in state_dict: " + ", ".join(repr(k) for k in unexpected))
        sd = OrderedDict()
        for k, shape in self._specs:
            if k not in state_dict:
                continue
            v = state_dict[k]
            v = th.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
            if tuple(v.shape) != shape:
                errs.append(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, "
                            f"the shape in current model is {shape}.")
                continue
            sd[k] = v.detach().to(device="cpu", dtype=th.float32).contiguous()
        if errs:
            raise RuntimeError("Error(s) in loading state_dict for CondMargVideoModel:\n\t" + "\n\t".join(errs))
        self._host_sd = sd
        if self._wbuf is not None:
            self._upload()
        return self

    def _ensure_storage(self):
        if self._wbuf is None:
            nbytes = _lib.lib().vd_weights_bytes(self._handle)
            self._wbuf = th.zeros(nbytes // 4, dtype=th.float32, device=self.device)
            _lib.check(_lib.lib().vd_set_weight_storage(self._handle, _lib.ptr(self._wbuf), nbytes))
            self._upload_freqs()
        elif self._wbuf.device.type == "cpu" and self.device.type == "cuda":
            # a host-packed image (pack_on_host / a received broadcast): one H2D copy, nothing is re-packed
            nbytes = _lib.lib().vd_weights_bytes(self._handle)
            self._wbuf = self._wbuf.to(self.device)
            _lib.check(_lib.lib().vd_set_weight_storage(self._handle, _lib.ptr(self._wbuf), nbytes))
            self._upload_freqs()
            self._packed_on_device = True

    def _upload(self):
        L = _lib.lib()
        for k, v in self._host_sd.items():
            _lib.check(L.vd_load_weight(self._handle, k.encode(), _lib.ptr(v), v.numel()))
        if self._wbuf_bwd is not None:
            self._upload_bwd()

    def _upload_bwd(self):
        L = _lib.lib()
        for k, v in self._host_sd.items():
            _lib.check(L.vd_load_weight_bwd(self._handle, k.encode(), _lib.ptr(v), v.numel()))
        self._bwd_loaded = True

    # -- use_gradient_method ------------------------------------------------------------------------
    def enable_guidance(self):
        """Allocate (and, when this process holds the checkpoint, fill) the backward-data weight image that
        `p_sample(..., use_gradient_method=True)` needs: transposed linear weights and rotated transposed 3x3 kernels in
        the forward kernels' layouts.  On ranks that only received the packed forward image, call this BEFORE
        `dist.share_weights` so the second image is broadcast as well."""
        if self._wbuf_bwd is None:
            nbytes = _lib.lib().vd_bwd_weights_bytes(self._handle)
            self._wbuf_bwd = th.zeros(max(nbytes // 4, 4), dtype=th.float32, device=self.device)
            _lib.check(_lib.lib().vd_set_bwd_weight_storage(self._handle, _lib.ptr(self._wbuf_bwd), nbytes,
                                                            0 if self.device.type == "cuda" else 1))
            self._bwd_loaded = False
            if self._host_sd is not None:
                self._upload_bwd()
        return self

    def guidance_weights(self):
        """The backward-data image (None unless enable_guidance() was called): a second buffer for the start-up broadcast."""
        return self._wbuf_bwd

    def mark_guidance_received(self):
        self._bwd_loaded = True

    def _require_guidance(self):
        if self._wbuf_bwd is None:
            if self._host_sd is None:
                raise RuntimeError("use_gradient_method: this rank holds only the broadcast forward weights; call "
                                   "model.enable_guidance() before dist.share_weights")
            self.enable_guidance()
        if self._wbuf_bwd.device.type != "cuda":
            if self.device.type != "cuda":
                raise RuntimeError("model.to('cuda') first: the HIP engine has no CPU path")
            nbytes = _lib.lib().vd_bwd_weights_bytes(self._handle)
            self._wbuf_bwd = self._wbuf_bwd.to(self.device)
            _lib.check(_lib.lib().vd_set_bwd_weight_storage(self._handle, _lib.ptr(self._wbuf_bwd), nbytes, 0))
        if not getattr(self, "_bwd_loaded", False):
            raise RuntimeError("use_gradient_method: the backward-data weights were never filled")

    def _upload_freqs(self):
        # nn.py:99-101 evaluated with the reference's own float32 expression (bit-identical angles)
        half = self.model_channels // 2
        tf = th.exp(-math.log(10000) * th.arange(start=0, end=half, dtype=th.float32) / half).contiguous()
        ff, nf = None, 0
        if self._use_frame_encoding:
            nf = self._pos_ch // 2
            ff = th.exp(-math.log(self.T * 10) * th.arange(start=0, end=nf, dtype=th.float32) / nf).contiguous()
        _lib.check(_lib.lib().vd_set_freqs(self._handle, _lib.ptr(tf), half, _lib.ptr(ff), nf))

    def to(self, device):
        device = th.device(device)
        if device.type != "cuda":
            if device.type == "cpu" and (self._wbuf is None or self._wbuf.device.type == "cpu"):
                return self
            raise RuntimeError("the HIP engine runs on a GPU only (no CPU fallback in the product path)")
        if device.index is None:
            device = th.device("cuda", th.cuda.current_device())
        if self._wbuf is not None and self._wbuf.device.type == "cuda" and self.device != device:
            raise RuntimeError("engine weights are already resident on " + str(self.device))
        self.device = device
        with th.cuda.device(device):
            was_host_packed = self._wbuf is not None and self._wbuf.device.type == "cpu"
            self._ensure_storage()
            if self._host_sd is not None and not was_host_packed:
                self._upload()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the HIP engine is inference-only")
        return self

    def parameters(self):
        """Callers only use `next(model.parameters()).device` (video_sample.py:155)."""
        if self._wbuf is None or self._wbuf.device.type == "cpu":
            return iter([th.zeros(1)])
        return iter([self._wbuf])

    def state_dict(self):
        if self._host_sd is None:
            raise RuntimeError("no weights loaded")
        return OrderedDict(self._host_sd)

    def packed_weights(self):
        """The single buffer holding every parameter (for one RCCL broadcast): device memory once the
        model is on a GPU, a host image (pack_on_host) before that."""
        if self.device.type != "cuda":
            return self.pack_on_host()
        self._ensure_storage()
        return self._wbuf

    def pack_on_host(self):
        """Assemble the packed image in HOST memory (a CPU tensor; needs no GPU): what rank 0 would broadcast.  The
        model cannot compute in this state; `.to('cuda')` afterwards uploads the image in one copy."""
        if self._wbuf is not None and self._wbuf.device.type != "cpu":
            raise RuntimeError("weights already live on " + str(self._wbuf.device))
        if self._wbuf is None:
            nbytes = _lib.lib().vd_weights_bytes(self._handle)
            self._wbuf = th.zeros(nbytes // 4, dtype=th.float32)
            _lib.check(_lib.lib().vd_set_weight_storage_host(self._handle, _lib.ptr(self._wbuf), nbytes))
        if self._host_sd is not None:
            self._upload()
        return self._wbuf

    def mark_weights_received(self):
        _lib.check(_lib.lib().vd_mark_weights_loaded(self._handle))

    def weights_layout_id(self):
        """Identifies the packed layout (parameter table + arithmetic mode): ranks compare it before a broadcast."""
        return int(_lib.lib().vd_weights_layout_id(self._handle))

    def check_device_errors(self):
        """Synchronise the DEVICE (every stream: the steps may have been issued under `torch.cuda.stream(s)`) and raise what the
        reference would have raised eagerly: an out-of-range timestep index is an
        IndexError in `_extract_into_tensor` (gaussian_diffusion.py:1019-1031); the asynchronous HIP step poisons the
        output with NaN and records it in a sticky device flag instead.  A network output that is not finite (bit 1: an
        fp16-range overflow of the f16x3 arithmetic, or NaN weights / inputs) is a FloatingPointError naming the way out."""
        flags = ctypes.c_int(0)
        _lib.check(_lib.lib().vd_device_errors(self._handle, ctypes.byref(flags)))
        if flags.value & 1:
            raise IndexError("timestep index out of range for the diffusion schedule (device flag set by an earlier step)")
        if flags.value & 2:
            raise FloatingPointError(
                "a denoise step consumed a network output that is not finite (device flag set by an earlier step; the affected "
                "elements of its result are NaN).  In the default arithmetic VD_MATH=f16x3 fp32 operands travel as two fp16 pieces: "
                "an operand beyond the split's range (|x| >= 2^15; inputs of a 3x3 conv: 2^13) overflows -- set VD_MATH=bf16x6 (exact three-piece split, full "
                "fp32 exponent range) on every rank and reload the weights")

    # -- forward ---------------------------------------------------------------------------------------
    def _pack_kwargs(self, x, kw):
        """model_kwargs of video_sample.py:157-165 -> contiguous device buffers for the C ABI."""
        B, T = x.shape[:2]
        dev = x.device
        mode = kw["observed_frames"]
        if mode not in _OBS_MODES:
            raise NotImplementedError(f"observed_frames={mode!r} (training-only option)")
        if self.cond_emb_type in ("duplicate", "all"):
            src = kw["x0"]                                                           # unet.py:1014-1017: x0 * obs_mask, whatever the mode
        elif self.cond_emb_type == "t=0":
            src = x
        else:
            src = {"x_0": kw["x0"], "x_t": x, "x_t_minus_1": kw["x_t_minus_1"]}[mode]   # KeyError like unet.py:961
        f32 = lambda t: t.to(device=dev, dtype=th.float32).contiguous()  # noqa: E731
        fi = kw.get("frame_indices")
        if fi is None:
            fi = th.arange(0, T, device=dev).view(1, T).expand(B, T)                 # unet.py:900-902
        assert tuple(kw["obs_mask"].shape[:2]) == (B, T)
        return dict(obs_src=f32(src), obs_mask=f32(kw["obs_mask"]).reshape(B * T),
                    latent_mask=f32(kw["latent_mask"]).reshape(B * T),
                    kinda_marg_mask=f32(kw["kinda_marg_mask"]).reshape(B * T),
                    frame_indices=fi.to(device=dev, dtype=th.int64).contiguous(), obs_mode=_OBS_MODES[mode])

    # -- return_attn_weights ---------------------------------------------------------------------------
    def _attn_capture(self, B, T):
        """Arm the engine's attention-weight capture for the next forward and return the reference's dict
        {'temporal': [(B*HW, T, T) per block], 'spatial': [(B*T, HW, HW) per block]} (unet.py:457-466,799-836)."""
        L = _lib.lib()
        n = L.vd_attn_blocks(self._handle)
        temporal, spatial = [], []
        res, ch = ctypes.c_int(), ctypes.c_int()
        for i in range(n):
            _lib.check(L.vd_attn_block_info(self._handle, i, ctypes.byref(res), ctypes.byref(ch)))
            hw = res.value * res.value
            temporal.append(th.empty(B * hw, T, T, dtype=th.float32, device=self.device))
            spatial.append(th.empty(B * T, hw, hw, dtype=th.float32, device=self.device))
        pt = (ctypes.c_void_p * n)(*[v.data_ptr() for v in temporal])
        ps = (ctypes.c_void_p * n)(*[v.data_ptr() for v in spatial])
        _lib.check(L.vd_set_attn_capture(self._handle, pt, ps, n))
        return {"temporal": temporal, "spatial": spatial}

    def _attn_release(self):
        _lib.check(_lib.lib().vd_set_attn_capture(self._handle, None, None, 0))

    def __call__(self, x, timesteps, return_attn_weights=False, **kwargs):
        """model(x, timesteps, **model_kwargs) -> (eps, attn)   (unet.py:949-1026); attn is None unless asked for."""
        if self._wbuf is None or self._wbuf.device.type != "cuda":
            raise RuntimeError("model.to('cuda') first: the HIP engine has no CPU path")
        B, T, C, H, W = x.shape
        assert H == self.image_size and W == self.image_size and C == 3
        xs = x.to(device=self.device, dtype=th.float32).contiguous()
        kw = self._pack_kwargs(xs, kwargs)
        tm = timesteps.to(device=self.device, dtype=th.float32).reshape(B).contiguous()
        eps = th.empty(B, T, self.out_channels, H, W, dtype=th.float32, device=self.device)
        attn = self._attn_capture(B, T) if return_attn_weights else None
        try:
            _lib.check(_lib.lib().vd_unet_forward(self._handle, B, T, _lib.ptr(xs), _lib.ptr(kw["obs_src"]),
                                                  _lib.ptr(kw["obs_mask"]), _lib.ptr(kw["latent_mask"]),
                                                  _lib.ptr(kw["kinda_marg_mask"]), _lib.ptr(kw["frame_indices"]),
                                                  _lib.ptr(tm), kw["obs_mode"], _lib.ptr(eps), _lib.current_stream()))
        finally:
            if attn is not None:
                self._attn_release()
        return eps, attn

    forward = __call__
