"""Host-side mirror of the reference's `StableDiffusion` surface over libsdmi.

The reference's caller (src/bin/sample/main.rs:100-109) uses
    sd.sample_image(context, unconditional_context, scale, n_steps) -> Vec<Vec<u8>>
and the public-but-unused `sample_latent`, `latent_to_image`
(src/model/stablediffusion/mod.rs:69,102), `UNet::forward` (unet/mod.rs:109),
`Autoencoder::decode_latent` (autoencoder/mod.rs:68) and `qkv_attention`
(attention.rs:5).  This module keeps those names, argument order and meaning;
tensors are numpy float32 arrays in the reference's layouts (NCHW latents,
[n, tokens, channels] sequences).  The Rust toolchain is absent here, so this
Python layer plays the part of the Rust shim in ffi/sdmi.rs; both only marshal
arguments into the C ABI -- all arithmetic happens in the HIP library.

Errors: the reference panics on shape errors; here they raise SdmiError (from
the C status) or ValueError (caught before the call).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _capi
from ._capi import SdmiConfig, SdmiError, check, load_library

def mpk_list(path) -> list:
    """[(dump name, shape, file offset)] of a Burn .mpk record, parsed by the C++ reader (host only, no GPU)."""
    lib = load_library()
    need = C.c_size_t()
    check(lib.sdmi_mpk_list(str(path).encode(), None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    check(lib.sdmi_mpk_list(str(path).encode(), buf, need.value, C.byref(need)))
    out = []
    for line in buf.value.decode().splitlines():
        if line.startswith("#") or not line:
            continue
        name, shape, off = line.split("\t")
        out.append((name, tuple(int(v) for v in shape.split(",")) if shape else (), int(off)))
    return out


__all__ = ["ModelConfig", "StableDiffusion", "MultiStableDiffusion", "mpk_list", "UNet", "Autoencoder", "CLIP", "SimpleTokenizer", "qkv_attention", "SdmiError"]


@dataclass(frozen=True)
class ModelConfig:
    """Hyper-parameters hard-coded in the reference's *Config::init
    (unet/mod.rs:36-92, autoencoder/mod.rs:30-36, stablediffusion/mod.rs:116)."""
    model_channels: int = 320
    n_head: int = 8
    ctx_dim: int = 768
    latent_h: int = 64
    latent_w: int = 64
    vae_ch: int = 128
    precision: int = 0   # 0 = fp32 (BASELINE configs[0..1]); 1 = bf16 storage + fp32 accumulate (configs[2..3]); 2 = 1 + MXFP8 ResBlock convs (configs[4])
    # CLIP text encoder, CLIPConfig::new(49408, 768, 12, 77, 12) (stablediffusion/mod.rs:29); width = ctx_dim.
    # 0 layers (the default here: the sampling path takes embeddings) builds the context without it.
    clip_layers: int = 0
    clip_heads: int = 12
    clip_vocab: int = 49408
    clip_ctx: int = 77

    @classmethod
    def sd_v1_4(cls, precision: int = 0, clip: bool = True) -> "ModelConfig":
        """The reference's full StableDiffusionConfig::init (stablediffusion/mod.rs:22-39), text encoder included."""
        return cls(precision=precision, clip_layers=12 if clip else 0)


def _f32(a, shape=None, name="array") -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class StableDiffusion:
    """`StableDiffusion<B>` (src/model/stablediffusion/mod.rs:41-48) on one MI355X."""

    def __init__(self, config: ModelConfig = ModelConfig(), device: int = 0, _borrowed_ctx=None):
        self._lib = load_library()
        self.config = config
        self._owned = _borrowed_ctx is None
        if _borrowed_ctx is not None:      # a per-device view of a MultiStableDiffusion (sdmi_multi_ctx): not ours to destroy
            self._ctx = C.c_void_p(_borrowed_ctx)
            self.unet = UNet(self)
            self.autoencoder = Autoencoder(self)
            self.clip = CLIP(self)
            return
        cfg = SdmiConfig()
        check(self._lib.sdmi_default_config(C.byref(cfg)))
        cfg.device = device
        cfg.model_channels = config.model_channels
        cfg.n_head = config.n_head
        cfg.ctx_dim = config.ctx_dim
        cfg.latent_h = config.latent_h
        cfg.latent_w = config.latent_w
        cfg.vae_ch = config.vae_ch
        cfg.precision = config.precision
        cfg.clip_layers = config.clip_layers
        cfg.clip_heads = config.clip_heads
        cfg.clip_vocab = config.clip_vocab
        cfg.clip_ctx = config.clip_ctx
        self._ctx = C.c_void_p()
        check(self._lib.sdmi_create(C.byref(self._ctx), C.byref(cfg)))
        # every option this context was given, in order (bench.py prints the non-default ones next to its figures: `applied_options`)
        self.applied_options: list[tuple[str, str]] = []
        # SDMI_OPTS="key=value key=value": engine options applied to every context of the process (A/B runs of the test suite under another kernel setting).
        # A malformed entry is an error, not a silent no-op: a stale variable must not change a published figure unnoticed.
        for kv in os.environ.get("SDMI_OPTS", "").split():
            k, eq, v = kv.partition("=")
            if not k or not eq:
                raise ValueError(f"SDMI_OPTS: '{kv}' is not key=value")
            self.set_option(k, v)
        self.options_from_env = list(self.applied_options)
        self.unet = UNet(self)
        self.autoencoder = Autoencoder(self)
        self.clip = CLIP(self)

    # ---- lifecycle -----------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            if getattr(self, "_owned", True):
                self._lib.sdmi_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- weights ---------------------------------------------------------------
    def weight_specs(self):
        """[(name, shape)] of every tensor the hot path needs (reference dump names)."""
        n = self._lib.sdmi_weight_count(self._ctx)
        if n < 0:
            check(n)
        out = []
        name = C.c_char_p()
        ndim = C.c_int32()
        dims = (C.c_int64 * 4)()
        for i in range(n):
            check(self._lib.sdmi_weight_info(self._ctx, i, C.byref(name), C.byref(ndim), dims))
            out.append((name.value.decode(), tuple(int(dims[k]) for k in range(ndim.value))))
        return out

    def set_weight(self, name: str, array) -> None:
        a = np.ascontiguousarray(array, dtype=np.float32)
        dims = (C.c_int64 * max(1, a.ndim))(*a.shape)
        check(self._lib.sdmi_set_weight(self._ctx, name.encode(), _fp(a), a.ndim, dims))

    GROUP_HOT, GROUP_CLIP, GROUP_ENCODER = 1, 2, 4

    @staticmethod
    def _group_of(name: str) -> int:
        if name.startswith("clip/"):
            return StableDiffusion.GROUP_CLIP
        if name.startswith("autoencoder/encoder/") or name.startswith("autoencoder/quant_conv/"):
            return StableDiffusion.GROUP_ENCODER
        return StableDiffusion.GROUP_HOT

    def pack_weights(self, provider, groups: int = 1) -> np.ndarray:
        """The flat image sdmi_load_weights_packed expects: every tensor of the selected groups in weight_specs()
        order, fp32, reference layouts, back to back (SURVEY.md 8b "flat pack")."""
        from .synthetic import alphas_cumprod, named_tensor
        specs = self.weight_specs()
        shapes = dict(specs)
        n = self._lib.sdmi_packed_size(self._ctx, groups)
        if n < 0:
            check(int(n))
        flat = np.empty(int(n), dtype=np.float32)
        off = 0
        for name, shape in specs:
            if not (self._group_of(name) & groups):
                continue
            cnt = int(np.prod(shape))
            src = alphas_cumprod(shape[0]) if name == "alphas_cumprod" else named_tensor(provider, name, shape, shapes)
            flat[off:off + cnt] = np.asarray(src, dtype=np.float32).reshape(-1)
            off += cnt
        assert off == flat.size
        return flat

    def load_weights_packed(self, flat: np.ndarray, groups: int = 1) -> None:
        """One staged upload of the whole model (sdmi_load_weights_packed) + finalize."""
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        check(self._lib.sdmi_load_weights_packed(self._ctx, _fp(flat), flat.size, groups))
        check(self._lib.sdmi_finalize_weights(self._ctx))

    def load_weights_mpk(self, path) -> None:
        """The reference's `burn` model type: a NamedMpkFileRecorder<FullPrecisionSettings> record (sample/main.rs:27-34)."""
        check(self._lib.sdmi_load_weights_mpk(self._ctx, str(path).encode()))
        check(self._lib.sdmi_finalize_weights(self._ctx))

    def set_stream(self, hip_stream, enable: bool = True) -> None:
        """Name the HIP stream (integer handle, e.g. torch.cuda.current_stream().cuda_stream) the caller's device
        buffers of the *_dev calls are produced / consumed on (sdmi_set_stream)."""
        check(self._lib.sdmi_set_stream(self._ctx, C.c_void_p(int(hip_stream) if hip_stream else 0), 1 if enable else 0))

    def load_weights(self, provider, clip: bool = True, vae_encoder: bool = True) -> None:
        """Pull every tensor from `provider.get(name, shape, kind, fan_in)`
        (synthetic.SyntheticWeights) -- the counterpart of load_stable_diffusion
        (stablediffusion/load.rs:16-33) for seeded synthetic parameters.  `clip=False` leaves the
        optional clip/... group unset (context()/clip.forward then raise)."""
        specs = self.weight_specs()
        shapes = dict(specs)
        for name, shape in specs:
            if name.startswith("clip/") and not clip:
                continue
            if (name.startswith("autoencoder/encoder/") or name.startswith("autoencoder/quant_conv/")) and not vae_encoder:
                continue
            if name == "alphas_cumprod":
                from .synthetic import alphas_cumprod
                self.set_weight(name, alphas_cumprod(shape[0]))
                continue
            from .synthetic import named_tensor
            arr = named_tensor(provider, name, shape, shapes)
            self.set_weight(name, arr)
        check(self._lib.sdmi_finalize_weights(self._ctx))

    # ---- prompt -> context (stablediffusion/mod.rs:194-210) ----------------------
    def context(self, tokenizer: "SimpleTokenizer", text: str) -> np.ndarray:
        """StableDiffusion::context: CLIP embedding [1, T, ctx_dim] of "<|startoftext|>{text}<|endoftext|>", T = tokens + 2."""
        cap = self.config.clip_ctx
        out = np.empty((cap, self.config.ctx_dim), dtype=np.float32)
        T = C.c_int32()
        check(self._lib.sdmi_context(self._ctx, tokenizer._tok, text.encode("utf-8"), _fp(out), cap, C.byref(T)))
        return out[None, :T.value].copy()

    def unconditional_context(self, tokenizer: "SimpleTokenizer") -> np.ndarray:
        """StableDiffusion::unconditional_context (:194-196): context("") squeezed to [2, ctx_dim]."""
        return self.context(tokenizer, "")[0]

    def load_weights_dir(self, dump_dir: str) -> None:
        """npy-dump tree written by the reference's python/ exporters
        (load_stable_diffusion, stablediffusion/load.rs:16-33)."""
        check(self._lib.sdmi_load_weights_dir(self._ctx, str(dump_dir).encode()))
        check(self._lib.sdmi_finalize_weights(self._ctx))

    # ---- reference surface -------------------------------------------------------
    def _check_ctx(self, context, unconditional_context):
        cd = self.config.ctx_dim
        context = _f32(context, name="context")
        if context.ndim != 3 or context.shape[2] != cd:
            raise ValueError(f"context must be [n, T, {cd}], got {context.shape}")
        uncond = _f32(unconditional_context, name="unconditional_context")
        if uncond.ndim != 2 or uncond.shape[1] != cd:
            raise ValueError(f"unconditional_context must be [Tu, {cd}], got {uncond.shape}")
        return context, uncond

    def sample_latent(self, context, unconditional_context, unconditional_guidance_scale: float, n_steps: int,
                      init_latent=None, seed: int = 0) -> np.ndarray:
        """stablediffusion/mod.rs:102-160 -> latent [n,4,h,w].  `init_latent`
        is x_T (the reference draws it from an unseeded RNG)."""
        context, uncond = self._check_ctx(context, unconditional_context)
        n, T, _ = context.shape
        h, w = self.config.latent_h, self.config.latent_w
        out = np.empty((n, 4, h, w), dtype=np.float32)
        x0 = None if init_latent is None else _f32(init_latent, (n, 4, h, w), "init_latent")
        check(self._lib.sdmi_sample_latent(self._ctx, _fp(context), n, T, _fp(uncond), uncond.shape[0],
                                           float(unconditional_guidance_scale), int(n_steps),
                                           None if x0 is None else _fp(x0), int(seed), _fp(out)))
        return out

    def latent_to_image(self, latent) -> np.ndarray:
        """stablediffusion/mod.rs:69-100 -> uint8 [n, 8h, 8w, 3] (the reference's Vec<Vec<u8>>)."""
        h, w = self.config.latent_h, self.config.latent_w
        latent = _f32(latent, name="latent")
        if latent.ndim != 4 or latent.shape[1:] != (4, h, w):
            raise ValueError(f"latent must be [n,4,{h},{w}], got {latent.shape}")
        n = latent.shape[0]
        out = np.empty((n, 8 * h, 8 * w, 3), dtype=np.uint8)
        check(self._lib.sdmi_latent_to_image(self._ctx, _fp(latent), n, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def sample_image(self, context, unconditional_context, unconditional_guidance_scale: float, n_steps: int,
                     init_latent=None, seed: int = 0) -> np.ndarray:
        """stablediffusion/mod.rs:51-67 -> uint8 [n, 8h, 8w, 3]."""
        context, uncond = self._check_ctx(context, unconditional_context)
        n, T, _ = context.shape
        h, w = self.config.latent_h, self.config.latent_w
        out = np.empty((n, 8 * h, 8 * w, 3), dtype=np.uint8)
        x0 = None if init_latent is None else _f32(init_latent, (n, 4, h, w), "init_latent")
        check(self._lib.sdmi_sample_image(self._ctx, _fp(context), n, T, _fp(uncond), uncond.shape[0],
                                          float(unconditional_guidance_scale), int(n_steps),
                                          None if x0 is None else _fp(x0), int(seed),
                                          out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    # ---- device-pointer variants (zero copy; pointers are ints, e.g. torch .data_ptr()) ----
    def sample_image_dev(self, context_ptr: int, n: int, T: int, uncond_ptr: int, Tu: int, scale: float,
                         n_steps: int, init_latent_ptr: int, rgb_out_ptr: int) -> None:
        check(self._lib.sdmi_sample_image_dev(self._ctx, context_ptr, n, T, uncond_ptr, Tu, float(scale),
                                              int(n_steps), init_latent_ptr, rgb_out_ptr))

    def sample_latent_dev(self, context_ptr: int, n: int, T: int, uncond_ptr: int, Tu: int, scale: float,
                          n_steps: int, init_latent_ptr: int, latent_out_ptr: int) -> None:
        check(self._lib.sdmi_sample_latent_dev(self._ctx, context_ptr, n, T, uncond_ptr, Tu, float(scale),
                                               int(n_steps), init_latent_ptr, latent_out_ptr))

    def latent_to_image_dev(self, latent_ptr: int, n: int, rgb_out_ptr: int) -> None:
        check(self._lib.sdmi_latent_to_image_dev(self._ctx, latent_ptr, n, rgb_out_ptr))

    # ---- introspection ------------------------------------------------------------------
    def synchronize(self):
        check(self._lib.sdmi_synchronize(self._ctx))

    # options that change no kernel choice (measurement / dump switches): not listed as non-default settings
    _PASSIVE_OPTIONS = ("profile", "profile_reset", "record_shapes", "dump_shapes", "dump_choices", "dump_profile_tags", "roctx")

    def set_option(self, key: str, value) -> None:
        check(self._lib.sdmi_set_option(self._ctx, key.encode(), str(value).encode()))
        if hasattr(self, "applied_options") and key not in self._PASSIVE_OPTIONS:
            self.applied_options.append((key, str(value)))

    def last_call_stats(self) -> dict:
        ms, nk, fl = C.c_double(), C.c_int64(), C.c_double()
        check(self._lib.sdmi_last_call_stats(self._ctx, C.byref(ms), C.byref(nk), C.byref(fl)))
        return {"gpu_ms": ms.value, "kernels": nk.value, "flops": fl.value}

    PROFILE_CLASSES = ("conv_gemm", "splitk_reduce", "attention", "group_norm", "layer_norm", "conv_gemm_fp8", "conv_gemm_split", "split_rows", "other", "geglu")

    def profile_overhead_us(self) -> float:
        """what an empty HIP-event pair reads on the engine's stream (calibrated when profiling was switched on; subtracted from every sample)"""
        v = C.c_double()
        check(self._lib.sdmi_profile_overhead(self._ctx, C.byref(v)))
        return v.value * 1e3

    def profile_stats(self) -> dict:
        """Per-kernel-class HIP-event timings gathered while set_option("profile", 1)."""
        out = {}
        for i, name in enumerate(self.PROFILE_CLASSES):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            check(self._lib.sdmi_profile_stats(self._ctx, i, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out

    def bench_conv(self, n, cin, h, w, cout, k=3, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=10) -> float:
        ms = C.c_double()
        check(self._lib.sdmi_bench_conv(self._ctx, n, cin, h, w, cout, k, stride, upsample2x, tile_cfg, splitk, iters,
                                        C.byref(ms)))
        return ms.value

    def bench_attention(self, n, nq, nk, n_state, n_head, iters=10) -> float:
        ms = C.c_double()
        check(self._lib.sdmi_bench_attention(self._ctx, n, nq, nk, n_state, n_head, iters, C.byref(ms)))
        return ms.value

    # ---- operator-level entry points (parity tests) ----------------------------------------
    def op_group_norm(self, x, gamma, beta, n_group=32, eps=1e-5, silu=False):
        x = _f32(x)
        n, c, h, w = x.shape
        out = np.empty_like(x)
        check(self._lib.sdmi_op_group_norm(self._ctx, _fp(x), _fp(_f32(gamma, (c,))), _fp(_f32(beta, (c,))), n, c, h, w,
                                           n_group, eps, int(silu), _fp(out)))
        return out

    def op_group_norm_fp8(self, x, gamma, beta, n_group=32, eps=1e-5, silu=False):
        """GroupNorm(+SiLU) with MXFP8 output (precision = 2), returned dequantised."""
        x = _f32(x)
        n, c, h, w = x.shape
        out = np.empty_like(x)
        check(self._lib.sdmi_op_group_norm_fp8(self._ctx, _fp(x), _fp(_f32(gamma, (c,))), _fp(_f32(beta, (c,))), n, c, h, w,
                                               n_group, eps, int(silu), _fp(out)))
        return out

    def op_layer_norm(self, x, gamma, beta, eps=1e-5):
        x = _f32(x)
        c = x.shape[-1]
        rows = x.size // c
        out = np.empty_like(x)
        check(self._lib.sdmi_op_layer_norm(self._ctx, _fp(x), _fp(_f32(gamma, (c,))), _fp(_f32(beta, (c,))), rows, c,
                                           eps, _fp(out)))
        return out

    def op_conv2d(self, x, weight, bias=None, stride=1, pad=None, upsample2x=False):
        x = _f32(x)
        weight = _f32(weight)
        n, cin, h, w = x.shape
        cout, cin2, k, k2 = weight.shape
        if cin2 != cin or k != k2:
            raise ValueError("conv2d: weight shape does not match input")
        if pad is None:
            pad = 1 if k == 3 else 0
        ups = 1 if upsample2x else 0
        ho = ((h << ups) + 2 * pad - k) // stride + 1
        wo = ((w << ups) + 2 * pad - k) // stride + 1
        out = np.empty((n, cout, ho, wo), dtype=np.float32)
        b = None if bias is None else _f32(bias, (cout,))
        check(self._lib.sdmi_op_conv2d(self._ctx, _fp(x), _fp(weight), None if b is None else _fp(b), n, cin, h, w, cout,
                                       k, stride, pad, ups, _fp(out)))
        return out

    def op_linear(self, x, weight, bias=None):
        x = _f32(x)
        weight = _f32(weight)
        cin, cout = weight.shape
        rows = x.size // cin
        out = np.empty(x.shape[:-1] + (cout,), dtype=np.float32)
        b = None if bias is None else _f32(bias, (cout,))
        check(self._lib.sdmi_op_linear(self._ctx, _fp(x), _fp(weight), None if b is None else _fp(b), rows, cin, cout,
                                       _fp(out)))
        return out

    def op_geglu_forward(self, x, weight_in_out, bias, hidden):
        """GEGLU::forward (unet/mod.rs:579-591): x [rows, cin] -> [rows, hidden]."""
        x = _f32(x)
        rows, cin = x.shape
        w = _f32(weight_in_out, (cin, 2 * hidden))
        out = np.empty((rows, hidden), dtype=np.float32)
        b = None if bias is None else _f32(bias, (2 * hidden,))
        check(self._lib.sdmi_op_geglu_forward(self._ctx, _fp(x), _fp(w), None if b is None else _fp(b), rows, cin, hidden, _fp(out)))
        return out

    def op_geglu(self, proj):
        proj = _f32(proj)
        hidden = proj.shape[-1] // 2
        rows = proj.size // (2 * hidden)
        out = np.empty(proj.shape[:-1] + (hidden,), dtype=np.float32)
        check(self._lib.sdmi_op_geglu(self._ctx, _fp(proj), rows, hidden, _fp(out)))
        return out

    def op_timestep_embedding(self, t: int, dim: int):
        out = np.empty((1, dim), dtype=np.float32)
        check(self._lib.sdmi_op_timestep_embedding(self._ctx, int(t), dim, _fp(out)))
        return out

    def qkv_attention(self, q, k, v, mask, n_head: int):
        """attention.rs:5-45: q [n,nq,c], k,v [n,nk,c], mask [>=nq, >=nk] or None."""
        q, k, v = _f32(q), _f32(k), _f32(v)
        n, nq, c = q.shape
        nk = k.shape[1]
        if k.shape != (n, nk, c) or v.shape != (n, nk, c):
            raise ValueError("qkv_attention: q/k/v shapes disagree")
        out = np.empty_like(q)
        m = None if mask is None else _f32(mask)
        check(self._lib.sdmi_qkv_attention(self._ctx, _fp(q), _fp(k), _fp(v), None if m is None else _fp(m),
                                           0 if m is None else m.shape[1], n, nq, nk, c, n_head, _fp(out)))
        return out


def _make_cfg(lib, config: ModelConfig, device: int = 0) -> SdmiConfig:
    cfg = SdmiConfig()
    check(lib.sdmi_default_config(C.byref(cfg)))
    cfg.device = device
    for f in ("model_channels", "n_head", "ctx_dim", "latent_h", "latent_w", "vae_ch", "precision", "clip_layers", "clip_heads",
              "clip_vocab", "clip_ctx"):
        setattr(cfg, f, getattr(config, f))
    return cfg


class MultiStableDiffusion:
    """`StableDiffusion::sample_image` for n images of one prompt, sharded over the GPUs of one node behind the C ABI
    (sdmi_create_multi / sdmi_sample_image_sharded): one process, one engine + host thread per device,
    ONE RCCL broadcast of the packed prompt embedding per call, contiguous image ranges, noise keyed by the global
    image index."""

    def __init__(self, config: ModelConfig = ModelConfig(), devices=(0,)):
        self._lib = load_library()
        self.config = config
        cfg = _make_cfg(self._lib, config)
        devs = (C.c_int32 * len(devices))(*devices)
        self._m = C.c_void_p()
        check(self._lib.sdmi_create_multi(C.byref(self._m), C.byref(cfg), devs, len(devices)))
        self.devices = tuple(devices)

    def close(self):
        if getattr(self, "_m", None) is not None and self._m.value:
            self._lib.sdmi_destroy_multi(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_view(self, index: int) -> StableDiffusion:
        """The single-device surface of device `index` (weights, options); owned by this object."""
        ctx = self._lib.sdmi_multi_ctx(self._m, index)
        if not ctx:
            check(-1)
        return StableDiffusion(self.config, _borrowed_ctx=ctx)

    def load_weights(self, provider, clip: bool = False, vae_encoder: bool = False) -> None:
        for i in range(len(self.devices)):
            self.device_view(i).load_weights(provider, clip=clip, vae_encoder=vae_encoder)

    def load_weights_path(self, kind: str, path) -> None:
        """kind = "dump" (npy tree) | "burn" (.mpk record), on every device in parallel."""
        check(self._lib.sdmi_multi_load_weights(self._m, kind.encode(), str(path).encode()))

    def sample_image(self, context, unconditional_context, unconditional_guidance_scale: float, n_steps: int, n_images: int,
                     init_latents=None, seed: int = 0) -> np.ndarray:
        cd, h, w = self.config.ctx_dim, self.config.latent_h, self.config.latent_w
        context = _f32(context, name="context")
        if context.ndim == 3 and context.shape[0] == 1:
            context = context[0]
        if context.ndim != 2 or context.shape[1] != cd:
            raise ValueError(f"context must be [T, {cd}] (one prompt), got {context.shape}")
        uncond = _f32(unconditional_context, name="unconditional_context")
        if uncond.ndim != 2 or uncond.shape[1] != cd:
            raise ValueError(f"unconditional_context must be [Tu, {cd}], got {uncond.shape}")
        x0 = None if init_latents is None else _f32(init_latents, (n_images, 4, h, w), "init_latents")
        out = np.empty((n_images, 8 * h, 8 * w, 3), dtype=np.uint8)
        check(self._lib.sdmi_sample_image_sharded(self._m, _fp(context), context.shape[0], _fp(uncond), uncond.shape[0],
                                                  float(unconditional_guidance_scale), int(n_steps), int(n_images),
                                                  None if x0 is None else _fp(x0), int(seed),
                                                  out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def broadcast_count(self) -> int:
        return int(self._lib.sdmi_multi_broadcast_count(self._m))


class UNet:
    """`UNet<B>` (src/model/unet/mod.rs:96-143): forward(x, timesteps, context)."""

    def __init__(self, sd: StableDiffusion):
        self._sd = sd

    def forward(self, x, timesteps, context) -> np.ndarray:
        sd = self._sd
        h, w = sd.config.latent_h, sd.config.latent_w
        x = _f32(x, name="x")
        if x.ndim != 4 or x.shape[1:] != (4, h, w):
            raise ValueError(f"x must be [n,4,{h},{w}], got {x.shape}")
        ts = np.atleast_1d(np.asarray(timesteps)).astype(np.int64)
        if ts.size != 1:
            raise ValueError("the reference passes a single shared timestep (unet/mod.rs:112, Tensor<B,1,Int> of len 1)")
        context = _f32(context, name="context")
        n = x.shape[0]
        if context.ndim != 3 or context.shape[0] != n or context.shape[2] != sd.config.ctx_dim:
            raise ValueError(f"context must be [{n}, T, {sd.config.ctx_dim}], got {context.shape}")
        out = np.empty_like(x)
        check(sd._lib.sdmi_unet_forward(sd._ctx, _fp(x), int(ts[0]), _fp(context), n, context.shape[1], _fp(out)))
        return out


class CLIP:
    """`CLIP<B>` (src/model/clip/mod.rs:48-75): forward(tokens [n, T] int) -> [n, T, ctx_dim]."""

    def __init__(self, sd: StableDiffusion):
        self._sd = sd

    def forward(self, tokens) -> np.ndarray:
        sd = self._sd
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        if t.ndim != 2:
            raise ValueError(f"tokens must be [n, seq_len], got {t.shape}")
        out = np.empty(t.shape + (sd.config.ctx_dim,), dtype=np.float32)
        check(sd._lib.sdmi_clip_forward(sd._ctx, t.ctypes.data_as(C.POINTER(C.c_int32)), t.shape[0], t.shape[1], _fp(out)))
        return out


class SimpleTokenizer:
    """`SimpleTokenizer` (src/tokenizer.rs:74-196) -- the C++ tokenizer inside libsdmi (no GPU needed).

    The reference reads "bpe_simple_vocab_16e6.txt" from the working directory; here the merges file is an argument."""

    def __init__(self, merges_path):
        self._lib = load_library()
        self._tok = C.c_void_p()
        check(self._lib.sdmi_tokenizer_create(C.byref(self._tok), str(merges_path).encode()))

    def close(self):
        if getattr(self, "_tok", None) is not None and self._tok.value:
            self._lib.sdmi_tokenizer_destroy(self._tok)
            self._tok = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def vocab_size(self) -> int:
        return int(self._lib.sdmi_tokenizer_vocab_size(self._tok))

    def encode(self, text: str):
        data = text.encode("utf-8")
        n = C.c_int32()
        cap = 4 * len(data) + 8   # at most one token per byte
        ids = (C.c_int32 * cap)()
        check(self._lib.sdmi_tokenizer_encode(self._tok, data, ids, cap, C.byref(n)))
        return [int(ids[i]) for i in range(n.value)]

    def decode(self, tokens) -> str:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        n = C.c_int32()
        cap = 64 * max(1, t.size)
        buf = C.create_string_buffer(cap)
        check(self._lib.sdmi_tokenizer_decode(self._tok, t.ctypes.data_as(C.POINTER(C.c_int32)), t.size, buf, cap, C.byref(n)))
        return buf.raw[:n.value].decode("utf-8", "replace")


class Autoencoder:
    """`Autoencoder<B>` (src/model/autoencoder/mod.rs:47-71): decode_latent on the hot path; encode_image / forward
    with the optional encoder weights."""

    def __init__(self, sd: StableDiffusion):
        self._sd = sd

    def encode_image(self, x) -> np.ndarray:
        """autoencoder/mod.rs:60-66: image [n,3,8h,8w] -> latent [n,4,h,w] (first 4 quant_conv channels)."""
        sd = self._sd
        h, w = sd.config.latent_h, sd.config.latent_w
        x = _f32(x, name="x")
        if x.ndim != 4 or x.shape[1:] != (3, 8 * h, 8 * w):
            raise ValueError(f"x must be [n,3,{8 * h},{8 * w}], got {x.shape}")
        out = np.empty((x.shape[0], 4, h, w), dtype=np.float32)
        check(sd._lib.sdmi_encode_image(sd._ctx, _fp(x), x.shape[0], _fp(out)))
        return out

    def forward(self, x) -> np.ndarray:
        """autoencoder/mod.rs:56-58: decode_latent(encode_image(x))."""
        return self.decode_latent(self.encode_image(x))

    def decode_latent(self, latent) -> np.ndarray:
        sd = self._sd
        h, w = sd.config.latent_h, sd.config.latent_w
        latent = _f32(latent, name="latent")
        if latent.ndim != 4 or latent.shape[1:] != (4, h, w):
            raise ValueError(f"latent must be [n,4,{h},{w}], got {latent.shape}")
        n = latent.shape[0]
        out = np.empty((n, 3, 8 * h, 8 * w), dtype=np.float32)
        check(sd._lib.sdmi_decode_latent(sd._ctx, _fp(latent), n, _fp(out)))
        return out


def qkv_attention(sd: StableDiffusion, q, k, v, mask, n_head: int):
    """Free-function form, as in the reference (attention.rs:5)."""
    return sd.qkv_attention(q, k, v, mask, n_head)
