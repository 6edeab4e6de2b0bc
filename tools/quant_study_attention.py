#!/usr/bin/env python3
"""What MXFP8 operands in the attention products would cost (BASELINE.json configs[4] says "fp8 conv+attn"), measured on the CPU oracle.

    python tools/quant_study_attention.py [--full]

qkv_attention (attention.rs:5-45) has two contractions: S = Q K^T over the head dimension d (40 / 80 / 160: one MX block of 32 plus a
partial block at d = 40) and O = P V over the keys (blocks of 32 keys; P in [0, 1]).  The study quantises their operands with the OCP MX
rules (oracle/mx_oracle.py) on top of the quantisation precision = 2 already applies (MxResConvs wide = True) and prints the relative RMS
error of one UNet forward against the exact fp64 network:
    base        Linear layers + convolutions in MXFP8 (what precision = 2 runs)
    +qk         ... and Q, K quantised along d before S = Q K^T
    +pv         ... and P, V quantised along the keys before O = P V
    +qk+pv      both
Default: the half-width test model (seconds); --full: SD v1.4 size, one forward per variant (about half a minute each).
The result is the written ground for keeping the attention products in bf16 ("fp8: what is and what is not quantised").
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import mx_oracle as MX  # noqa: E402
from oracle import sd_oracle as O  # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402


def make_attention(qk: bool, pv: bool):
    def attn(q, k, v, mask, n_head):
        n_batch, n_qctx, n_state = q.shape
        n_ctx = k.shape[1]
        scale = (n_state / n_head) ** -0.25
        d = n_state // n_head
        q = q.reshape(n_batch, n_qctx, n_head, d).transpose(1, 2) * scale          # [n, h, nq, d]
        k = k.reshape(n_batch, n_ctx, n_head, d).transpose(1, 2) * scale           # [n, h, nk, d]
        v = v.reshape(n_batch, n_ctx, n_head, d).transpose(1, 2)                   # [n, h, nk, d]
        if qk:
            q, k = MX.mx_quantize(q, 3), MX.mx_quantize(k, 3)
        s = q @ k.transpose(2, 3)
        if mask is not None:
            s = s + mask[:n_qctx, :n_ctx][None, None]
        w = torch.softmax(s, dim=3)
        if pv:
            w, v = MX.mx_quantize(w, 3), MX.mx_quantize(v, 2)                       # blocks along the keys for both operands
        return (w @ v).transpose(1, 2).flatten(2, 3)
    return attn


def rel_rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2) / np.mean(b ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--threads", type=int, default=6)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    d = O.Dims() if args.full else O.Dims(160, 4, 64, 16, 16, 32)
    w = syn.SyntheticWeights(cache=True)
    lat = torch.from_numpy(syn.initial_latent(0, d.latent_h, d.latent_w))[None]
    ctx = torch.from_numpy(syn.cond_context(0, 77, d.ctx_dim))[None]
    t0 = time.time()
    exact = O.UNetOracle(w, d, torch.float64).forward(lat, 999, ctx).numpy()
    print(f"{'model':12s} dims = {d}; exact forward {time.time() - t0:.1f} s")
    attn0 = O.qkv_attention
    rows = []
    for name, qk, pv, wide in (("bf16-free: convs only (fp8_linear=0)", False, False, False), ("base (precision = 2 as shipped)", False, False, True),
                               ("+qk", True, False, True), ("+pv", False, True, True), ("+qk+pv", True, True, True)):
        O.qkv_attention = make_attention(qk, pv) if (qk or pv) else attn0
        try:
            with MX.MxResConvs(wide=wide):
                got = O.UNetOracle(w, d, torch.float64).forward(lat, 999, ctx).numpy()
        finally:
            O.qkv_attention = attn0
        r = rel_rms(got, exact)
        rows.append((name, r))
        print(f"{name:40s} rel-RMS of one UNet forward vs exact fp64: {r:.3e}   ({time.time() - t0:.0f} s)", flush=True)
    base = dict(rows)["base (precision = 2 as shipped)"]
    for name, r in rows[2:]:
        print(f"{name:8s}: x{r / base:.2f} of the shipped configuration's error")


if __name__ == "__main__":
    main()
