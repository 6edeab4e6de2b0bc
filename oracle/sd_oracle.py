"""CPU ORACLE for the SD v1.4 sampling hot path -- TEST INFRASTRUCTURE ONLY.

PARITY: PINNED TO THE REFERENCE'S PYTHON MODEL, NOT TO A RUN OF THE RUST BINARY (see DESIGN.md
section 3).  The reference (Gadersd/stable-diffusion-burn) is Rust on top of Burn 0.14; neither a
Rust toolchain nor the Burn crates exist in this environment, the reference ships no golden
vectors for this path (its only test is a tokenizer test, src/tokenizer.rs:205-222) and its RNG is
unseeded (src/model/stablediffusion/mod.rs:115-121).  This file is therefore a *restatement* of
the reference's arithmetic, written from the cited Rust lines (NOT from diffusers / ldm, which
differ in eps, GELU flavour, context padding and DDIM indexing).  What pins it:
  * tests/golden/gen_from_reference_python.py imports the reference's own Python model
    (python/dump.py, through a small tinygrad-API shim) and its exporters (python/save.py,
    unet.py, autoencoder.py), loads the seeded synthetic weights BY THE REFERENCE'S DUMP NAMES and
    compares: UNet forward and VAE decoder agree with this oracle to 4e-15 / 3e-15 (fp64);
    the fixtures are committed (tests/golden/refpy_*.npz, refdump/) and checked by
    tests/test_reference_python_cpu.py;
  * independent re-derivations of every op and quirk in tests/test_oracle_cpu.py.
Not pinned (unavailable here): Burn's own kernels and the Rust-only deviations from the Python
model (exact-erf GELU Q4, unpadded context Q2, DDIM step_by Q5), restated from the Rust source.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module, and only as the checker / the reported CPU baseline --
never as the thing measured or shipped.  The product path
(stable_diffusion_burn_amd/) never imports it.

Burn semantics assumed (burn 0.14, unverifiable here):
  Conv2d  = cross-correlation, weight [Cout,Cin,kh,kw], zero padding
  Linear  = x @ W[in,out] + b
  LayerNorm = (x-mean)/sqrt(var_biased+eps)*gamma+beta, eps from dump (1e-5)
  Gelu    = 0.5*x*(1+erf(x/sqrt(2)))   (exact, not tanh)
  softmax = max-subtracted

All tensors are NCHW / [n, tokens, channels] exactly as in the reference.
``dtype`` selects torch.float32 (stand-in for "the reference's CPU run") or
torch.float64 (truth for tolerance budgeting).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# model dimensions (reference values are the defaults; tests shrink them)
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class Dims:
    model_channels: int = 320   # unet/mod.rs:41 (Conv 4->320)
    n_head: int = 8             # unet/mod.rs:44 (.., 768, 8)
    ctx_dim: int = 768          # unet/mod.rs:44
    latent_h: int = 64          # stablediffusion/mod.rs:116
    latent_w: int = 64
    vae_ch: int = 128           # autoencoder/mod.rs:33-34 (.., (256,128))

    @property
    def emb_dim(self) -> int:   # 1280 = 4*320, unet/mod.rs:38-40
        return 4 * self.model_channels


# --------------------------------------------------------------------------
# parameter access: names are the reference's npy-dump tree paths
# --------------------------------------------------------------------------
class Params:
    def __init__(self, provider, dtype=torch.float32):
        self.p = provider
        self.dtype = dtype

    def _t(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.dtype)

    def conv(self, path, cin, cout, k):
        fan = cin * k * k
        w = self._t(self.p.get(f"{path}/weight", (cout, cin, k, k), "w", fan))
        b = self._t(self.p.get(f"{path}/bias", (cout,), "b", fan))
        return w, b

    def linear(self, path, cin, cout, bias=True):
        w = self._t(self.p.get(f"{path}/weight", (cin, cout), "w", cin))  # [in,out]
        b = self._t(self.p.get(f"{path}/bias", (cout,), "b", cin)) if bias else None
        return w, b

    def norm(self, path, c):
        g = self._t(self.p.get(f"{path}/weight", (c,), "gamma"))
        b = self._t(self.p.get(f"{path}/bias", (c,), "beta"))
        return g, b


# --------------------------------------------------------------------------
# L2 custom ops
# --------------------------------------------------------------------------
def layernorm_ref(x: torch.Tensor, eps: float) -> torch.Tensor:
    """src/model/groupnorm/mod.rs:75-82 (biased variance, sqrt(var+eps))."""
    u = x - x.mean(dim=-1, keepdim=True)
    return u / ((u * u).mean(dim=-1, keepdim=True) + eps).sqrt()


def group_norm(x, gamma, beta, n_group=32, eps=1e-5):
    """src/model/groupnorm/mod.rs:53-73: reshape [n, G, rest] -> layernorm -> affine."""
    shape = x.shape
    n = shape[0]
    y = layernorm_ref(x.reshape(n, n_group, -1), eps).reshape(shape)
    aff = [1] * x.dim()
    aff[1] = shape[1]
    return y * gamma.reshape(aff) + beta.reshape(aff)


def silu(x):
    """src/model/silu.rs:14-16."""
    return x * torch.sigmoid(x)


def gelu_erf(x):
    """Burn nn::Gelu (exact erf form); unet/mod.rs:566,590."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def qkv_attention(q, k, v, mask, n_head):
    """src/model/attention.rs:5-45 (== src/backend.rs:88-128)."""
    n_batch, n_qctx, n_state = q.shape
    n_ctx = k.shape[1]
    scale = (n_state / n_head) ** -0.25
    n_hstate = n_state // n_head
    q = q.reshape(n_batch, n_qctx, n_head, n_hstate).transpose(1, 2) * scale
    k = k.reshape(n_batch, n_ctx, n_head, n_hstate).transpose(1, 2).transpose(2, 3) * scale
    v = v.reshape(n_batch, n_ctx, n_head, n_hstate).transpose(1, 2)
    qk = q @ k
    if mask is not None:
        qk = qk + mask[:n_qctx, :n_ctx][None, None]
    w = torch.softmax(qk, dim=3)
    return (w @ v).transpose(1, 2).flatten(2, 3)


def layer_norm(x, gamma, beta, eps=1e-5):
    """Burn nn::LayerNorm over the last dim (unet/mod.rs:523-525; eps from dump, load.rs:95)."""
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, eps)


def linear(x, w, b):
    y = x @ w
    return y + b if b is not None else y


def conv2d(x, wb, stride=1, padding=0):
    return F.conv2d(x, wb[0], wb[1], stride=stride, padding=padding)


def upsample2x(x):
    """unet/mod.rs:392-396: reshape [n,c,h,1,w,1] -> repeat(1,1,1,2,1,2) -> [n,c,2h,2w]."""
    n, c, h, w = x.shape
    return x.reshape(n, c, h, 1, w, 1).repeat(1, 1, 1, 2, 1, 2).reshape(n, c, 2 * h, 2 * w)


def timestep_embedding(t: int, dim: int, max_period: int, dtype) -> torch.Tensor:
    """unet/mod.rs:19-30.  Computed in f32 on the device in the reference (the
    f64 scalar -(ln max_period)/half is narrowed to the tensor's element type
    when it multiplies the f32 arange)."""
    half = dim // 2
    coef = torch.tensor(-math.log(float(max_period)) / half, dtype=dtype)
    freqs = (torch.arange(0, half, dtype=dtype) * coef).exp()
    args = torch.tensor(float(t), dtype=dtype) * freqs
    return torch.cat([args.cos(), args.sin()], dim=0).unsqueeze(0)


# --------------------------------------------------------------------------
# UNet (src/model/unet/mod.rs)
# --------------------------------------------------------------------------
class UNetOracle:
    def __init__(self, provider, dims: Dims = Dims(), dtype=torch.float32, root="unet"):
        self.P = Params(provider, dtype)
        self.d = dims
        self.dtype = dtype
        self.root = root
        self.trace = None  # optional dict name -> tensor for per-block parity

    # ---- blocks -----------------------------------------------------------
    def res_block(self, path, x, emb, cin, cout):
        """ResBlock::forward unet/mod.rs:713-733."""
        P = self.P
        h = group_norm(x, *P.norm(f"{path}/norm_in", cin))
        h = silu(h)
        h = conv2d(h, P.conv(f"{path}/conv_in", cin, cout, 3), padding=1)
        e = linear(silu(emb), *P.linear(f"{path}/lin_embed", self.d.emb_dim, cout))
        h = h + e.reshape(e.shape[0], e.shape[1], 1, 1)
        h = group_norm(h, *P.norm(f"{path}/norm_out", cout))
        h = silu(h)
        h = conv2d(h, P.conv(f"{path}/conv_out", cout, cout, 3), padding=1)
        if cin != cout:
            return conv2d(x, P.conv(f"{path}/skip_connection", cin, cout, 1)) + h
        return x + h

    def mha(self, path, x, context, c, c_ctx):
        """MultiHeadAttention::forward unet/mod.rs:642-652."""
        P = self.P
        xa = x if context is None else context
        q = linear(x, *P.linear(f"{path}/query", c, c, bias=False))
        k = linear(xa, *P.linear(f"{path}/key", c_ctx, c, bias=False))
        v = linear(xa, *P.linear(f"{path}/value", c_ctx, c, bias=False))
        wv = qkv_attention(q, k, v, None, self.d.n_head)
        return linear(wv, *P.linear(f"{path}/out", c, c))

    def mlp(self, path, x, c):
        """MLP / GEGLU unet/mod.rs:552-591 (mult = 4)."""
        P = self.P
        hidden = 4 * c
        proj = linear(x, *P.linear(f"{path}/geglu/proj", c, 2 * hidden))
        a, gate = proj[..., :hidden], proj[..., hidden:]
        return linear(a * gelu_erf(gate), *P.linear(f"{path}/lin", hidden, c))

    def transformer_block(self, path, x, context, c):
        """TransformerBlock::forward unet/mod.rs:522-526."""
        P = self.P
        x = x + self.mha(f"{path}/attn1", layer_norm(x, *P.norm(f"{path}/norm1", c)), None, c, c)
        x = x + self.mha(f"{path}/attn2", layer_norm(x, *P.norm(f"{path}/norm2", c)), context, c, self.d.ctx_dim)
        return x + self.mlp(f"{path}/mlp", layer_norm(x, *P.norm(f"{path}/norm3", c)), c)

    def spatial_transformer(self, path, x, context, c):
        """SpatialTransformer::forward unet/mod.rs:462-480."""
        P = self.P
        n, _, h, w = x.shape
        x_in = x
        x = group_norm(x, *P.norm(f"{path}/norm", c))
        x = conv2d(x, P.conv(f"{path}/proj_in", c, c, 1))
        x = x.reshape(n, c, h * w).transpose(1, 2)
        x = self.transformer_block(f"{path}/transformer", x, context, c)
        x = x.transpose(1, 2).reshape(n, c, h, w)
        return x_in + conv2d(x, P.conv(f"{path}/proj_out", c, c, 1))

    def upsample(self, path, x, c):
        """Upsample::forward unet/mod.rs:391-398."""
        return conv2d(upsample2x(x), self.P.conv(f"{path}/conv", c, c, 3), padding=1)

    def downsample(self, path, x, c):
        """Downsample = Conv2d stride 2 pad 1, unet/mod.rs:408-427."""
        return conv2d(x, self.P.conv(path, c, c, 3), stride=2, padding=1)

    # ---- topology (unet/mod.rs:36-92) ---------------------------------------
    def plan(self):
        mc = self.d.model_channels
        c1, c2, c4 = mc, 2 * mc, 4 * mc
        inp = [
            ("conv", "conv", 4, c1), ("rt", "rt1", c1, c1), ("rt", "rt2", c1, c1), ("down", "d1", c1, c1),
            ("rt", "rt3", c1, c2), ("rt", "rt4", c2, c2), ("down", "d2", c2, c2),
            ("rt", "rt5", c2, c4), ("rt", "rt6", c4, c4), ("down", "d3", c4, c4),
            ("r", "r1", c4, c4), ("r", "r2", c4, c4),
        ]
        out = [
            ("r", "r1", 2 * c4, c4), ("r", "r2", 2 * c4, c4), ("ru", "ru", 2 * c4, c4),
            ("rt", "rt1", 2 * c4, c4), ("rt", "rt2", 2 * c4, c4), ("rtu", "rtu1", c4 + c2, c4),
            ("rt", "rt3", c4 + c2, c2), ("rt", "rt4", 2 * c2, c2), ("rtu", "rtu2", c2 + c1, c2),
            ("rt", "rt5", c2 + c1, c1), ("rt", "rt6", 2 * c1, c1), ("rt", "rt7", 2 * c1, c1),
        ]
        return inp, out

    def _block(self, kind, path, x, emb, ctx, cin, cout):
        if kind == "conv":
            return conv2d(x, self.P.conv(path, cin, cout, 3), padding=1)
        if kind == "down":
            return self.downsample(path, x, cin)
        if kind == "r":
            return self.res_block(path, x, emb, cin, cout)
        x = self.res_block(f"{path}/res", x, emb, cin, cout)
        if kind in ("rt", "rtu"):
            x = self.spatial_transformer(f"{path}/transformer", x, ctx, cout)
        if kind in ("ru", "rtu"):
            x = self.upsample(f"{path}/upsample", x, cout)
        return x

    def time_embed(self, t: int):
        """unet/mod.rs:115-118."""
        P, mc, ed = self.P, self.d.model_channels, self.d.emb_dim
        e = timestep_embedding(t, mc, 10000, self.dtype)
        e = linear(e, *P.linear(f"{self.root}/lin1_time_embed", mc, ed))
        e = silu(e)
        return linear(e, *P.linear(f"{self.root}/lin2_time_embed", ed, ed))

    @torch.no_grad()
    def forward(self, x, t: int, context, norm_out_eps: float = 1e-5):
        """UNet::forward unet/mod.rs:109-143.  x [n,4,h,w]; context [n,T,ctx_dim].  norm_out_eps: the eps the dump carries
        for unet/norm_out (groupnorm/load.rs:19; 1e-5 in every real dump, Q3) -- a test varies it."""
        x = x.to(self.dtype)
        context = context.to(self.dtype)
        emb = self.time_embed(t)
        inp, out = self.plan()
        mc = self.d.model_channels
        saved = []
        for i, (kind, name, cin, cout) in enumerate(inp):
            x = self._block(kind, f"{self.root}/input_blocks/{name}", x, emb, context, cin, cout)
            saved.append(x)
            if self.trace is not None:
                self.trace[f"in{i}"] = x
        # middle: ResTransformerRes unet/mod.rs:362-367
        mp = f"{self.root}/middle_block"
        x = self.res_block(f"{mp}/res1", x, emb, 4 * mc, 4 * mc)
        x = self.spatial_transformer(f"{mp}/transformer", x, context, 4 * mc)
        x = self.res_block(f"{mp}/res2", x, emb, 4 * mc, 4 * mc)
        if self.trace is not None:
            self.trace["mid"] = x
        for i, (kind, name, cin, cout) in enumerate(out):
            x = torch.cat([x, saved.pop()], dim=1)
            x = self._block(kind, f"{self.root}/output_blocks/{name}", x, emb, context, cin, cout)
            if self.trace is not None:
                self.trace[f"out{i}"] = x
        x = group_norm(x, *self.P.norm(f"{self.root}/norm_out", mc), eps=norm_out_eps)
        x = silu(x)
        return conv2d(x, self.P.conv(f"{self.root}/conv_out", mc, 4, 3), padding=1)


# --------------------------------------------------------------------------
# VAE decoder half (src/model/autoencoder/mod.rs)
# --------------------------------------------------------------------------
class DecoderOracle:
    def __init__(self, provider, dims: Dims = Dims(), dtype=torch.float32, root="autoencoder"):
        self.P = Params(provider, dtype)
        self.d = dims
        self.dtype = dtype
        self.root = root
        self.trace = None

    def resnet_block(self, path, x, cin, cout):
        """ResnetBlock::forward autoencoder/mod.rs:514-527."""
        P = self.P
        h = conv2d(silu(group_norm(x, *P.norm(f"{path}/norm1", cin))), P.conv(f"{path}/conv1", cin, cout, 3), padding=1)
        h = conv2d(silu(group_norm(h, *P.norm(f"{path}/norm2", cout))), P.conv(f"{path}/conv2", cout, cout, 3), padding=1)
        if cin != cout:
            return conv2d(x, P.conv(f"{path}/nin_shortcut", cin, cout, 1)) + h
        return x + h

    def attn_block(self, path, x, c):
        """ConvSelfAttentionBlock::forward autoencoder/mod.rs:563-607 (1 head)."""
        P = self.P
        n, _, hh, ww = x.shape
        h = group_norm(x, *P.norm(f"{path}/norm", c))
        q = conv2d(h, P.conv(f"{path}/q", c, c, 1)).reshape(n, c, hh * ww).transpose(1, 2)
        k = conv2d(h, P.conv(f"{path}/k", c, c, 1)).reshape(n, c, hh * ww).transpose(1, 2)
        v = conv2d(h, P.conv(f"{path}/v", c, c, 1)).reshape(n, c, hh * ww).transpose(1, 2)
        wv = qkv_attention(q, k, v, None, 1).transpose(1, 2).reshape(n, c, hh, ww)
        return x + conv2d(wv, P.conv(f"{path}/proj_out", c, c, 1))

    def channels(self):
        c = self.d.vae_ch  # autoencoder/mod.rs:33-34: [(512,512),(512,512),(512,256),(256,128)]
        return [(4 * c, 4 * c), (4 * c, 4 * c), (4 * c, 2 * c), (2 * c, c)]

    @torch.no_grad()
    def decode_latent(self, latent):
        """Autoencoder::decode_latent :68-71 -> Decoder::forward :205-217."""
        P, r = self.P, self.root
        chans = self.channels()
        c0 = chans[0][0]
        x = latent.to(self.dtype)
        x = conv2d(x, P.conv(f"{r}/post_quant_conv", 4, 4, 1))
        x = conv2d(x, P.conv(f"{r}/decoder/conv_in", 4, c0, 3), padding=1)
        # Mid :457-462
        x = self.resnet_block(f"{r}/decoder/mid/block_1", x, c0, c0)
        x = self.attn_block(f"{r}/decoder/mid/attn", x, c0)
        x = self.resnet_block(f"{r}/decoder/mid/block_2", x, c0, c0)
        if self.trace is not None:
            self.trace["mid"] = x
        # DecoderBlock::forward :308-323
        for i, (cin, cout) in enumerate(chans):
            bp = f"{r}/decoder/blocks/{i}"
            x = self.resnet_block(f"{bp}/res1", x, cin, cout)
            x = self.resnet_block(f"{bp}/res2", x, cout, cout)
            x = self.resnet_block(f"{bp}/res3", x, cout, cout)
            if i != len(chans) - 1:
                x = conv2d(upsample2x(x), P.conv(f"{bp}/upsampler", cout, cout, 3), padding=1)
            if self.trace is not None:
                self.trace[f"block{i}"] = x
        cl = chans[-1][1]
        x = silu(group_norm(x, *P.norm(f"{r}/decoder/norm_out", cl)))
        return conv2d(x, P.conv(f"{r}/decoder/conv_out", cl, 3, 3), padding=1)


# --------------------------------------------------------------------------
# VAE encoder half (not reachable from `sample`)
# --------------------------------------------------------------------------
def padded_conv2d(x, wb, stride, pad_left, pad_right, pad_top, pad_bottom):
    """PaddedConv2d::forward autoencoder/mod.rs:390-411: a symmetric over-padded conv followed by a slice,
    which is the conv with the asymmetric zero padding (left, right, top, bottom)."""
    return conv2d(F.pad(x, (pad_left, pad_right, pad_top, pad_bottom)), wb, stride=stride, padding=0)


class EncoderOracle(DecoderOracle):
    def enc_channels(self):
        c = self.d.vae_ch  # autoencoder/mod.rs:31-32: [(128,128),(128,256),(256,512),(512,512)]
        return [(c, c), (c, 2 * c), (2 * c, 4 * c), (4 * c, 4 * c)]

    @torch.no_grad()
    def encode_image(self, img):
        """Autoencoder::encode_image :60-66 -> Encoder::forward :133-144; returns the first 4 of the 8
        quant_conv channels (the mean of the posterior; the reference never samples it)."""
        P, r = self.P, self.root
        chans = self.enc_channels()
        x = img.to(self.dtype)
        x = conv2d(x, P.conv(f"{r}/encoder/conv_in", 3, chans[0][0], 3), padding=1)
        for i, (cin, cout) in enumerate(chans):   # EncoderBlock::forward :257-265
            bp = f"{r}/encoder/blocks/{i}"
            x = self.resnet_block(f"{bp}/res1", x, cin, cout)
            x = self.resnet_block(f"{bp}/res2", x, cout, cout)
            if i != len(chans) - 1:               # PaddingCfg::new(0, 1, 0, 1), stride 2 (:231-236)
                x = padded_conv2d(x, P.conv(f"{bp}/downsampler/conv", cout, cout, 3), 2, 0, 1, 0, 1)
        c4 = chans[-1][1]
        x = self.resnet_block(f"{r}/encoder/mid/block_1", x, c4, c4)
        x = self.attn_block(f"{r}/encoder/mid/attn", x, c4)
        x = self.resnet_block(f"{r}/encoder/mid/block_2", x, c4, c4)
        x = silu(group_norm(x, *P.norm(f"{r}/encoder/norm_out", c4)))
        x = conv2d(x, P.conv(f"{r}/encoder/conv_out", c4, 8, 3), padding=1)
        x = conv2d(x, P.conv(f"{r}/quant_conv", 8, 8, 1))
        return x[:, :4]


# --------------------------------------------------------------------------
# Pipeline (src/model/stablediffusion/mod.rs)
# --------------------------------------------------------------------------
def ddim_timesteps(n_steps: int, total: int = 1000):
    """(0..total).rev().step_by(total / n_steps), stablediffusion/mod.rs:111,123."""
    step = total // n_steps
    return list(range(total - 1, -1, -step)), step


class StableDiffusionOracle:
    def __init__(self, provider, alphas_cumprod: np.ndarray, dims: Dims = Dims(), dtype=torch.float32):
        self.unet = UNetOracle(provider, dims, dtype)
        self.decoder = DecoderOracle(provider, dims, dtype)
        self.alphas = np.asarray(alphas_cumprod, dtype=np.float32)  # f32 tensor in the reference
        self.n_steps = len(self.alphas)
        self.dtype = dtype
        self.d = dims

    @torch.no_grad()
    def forward_diffuser(self, latent, t, context, uncond, scale: float):
        """stablediffusion/mod.rs:162-192.  Two separate forwards; the uncond
        context is broadcast over the batch (intended semantics)."""
        n = latent.shape[0]
        u = self.unet.forward(latent, t, uncond.unsqueeze(0).repeat(n, 1, 1))
        c = self.unet.forward(latent, t, context)
        return u + (c - u) * scale

    @torch.no_grad()
    def sample_latent(self, context, uncond, scale: float, n_steps: int, init_latent, per_step=None):
        """stablediffusion/mod.rs:102-160 with the initial noise passed in (Q6)."""
        latent = init_latent.to(self.dtype)
        context = context.to(self.dtype)
        uncond = uncond.to(self.dtype)
        ts, step = ddim_timesteps(n_steps, self.n_steps)
        for t in ts:
            cur = float(self.alphas[t])                       # f32 -> f64 (:124-129)
            prev = float(self.alphas[t - step]) if t >= step else 1.0
            sqrt_noise = math.sqrt(1.0 - cur)
            eps = self.forward_diffuser(latent, t, context, uncond, scale)
            predx0 = (latent - eps * sqrt_noise) / math.sqrt(cur)
            dir_latent = eps * math.sqrt(1.0 - prev - 0.0)
            latent = predx0 * math.sqrt(prev) + dir_latent      # + gen_noise()*0 (:155)
            if per_step is not None:
                per_step.append(latent.clone())
        return latent

    @torch.no_grad()
    def decode_float(self, latent):
        """latent_to_image up to (and excluding) the u8 conversion: [n,3,H,W] float."""
        return self.decoder.decode_latent(latent.to(self.dtype) * (1.0 / 0.18215))

    @torch.no_grad()
    def latent_to_image(self, latent):
        """stablediffusion/mod.rs:69-100 -> uint8 [n, H, W, 3] (truncating cast, :96)."""
        img = self.decode_float(latent)
        img = (img + 1.0) / 2.0
        img = img.permute(0, 2, 3, 1) * 255.0
        return img.to(torch.float64).clamp(0.0, 255.0).to(torch.uint8).numpy(), img

    @torch.no_grad()
    def sample_image(self, context, uncond, scale, n_steps, init_latent):
        """stablediffusion/mod.rs:51-67."""
        return self.latent_to_image(self.sample_latent(context, uncond, scale, n_steps, init_latent))[0]
