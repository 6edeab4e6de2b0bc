"""stable_diffusion_burn_amd -- MI355X (gfx950) native SD v1.4 sampling hot path.

The package is a thin host layer over `libsdmi.so` (hand-written HIP kernels +
C++ engine, C ABI in include/sdmi.h).  It exposes the reference's
`StableDiffusion` surface (Gadersd/stable-diffusion-burn,
src/model/stablediffusion/mod.rs) and nothing else; there is no CPU or PyTorch
fallback -- importing the pipeline without a built library raises.
"""
from .pipeline import CLIP, Autoencoder, ModelConfig, MultiStableDiffusion, SdmiError, SimpleTokenizer, StableDiffusion, UNet, mpk_list, qkv_attention  # noqa: F401
from . import synthetic  # noqa: F401

__all__ = ["StableDiffusion", "MultiStableDiffusion", "UNet", "Autoencoder", "CLIP", "SimpleTokenizer", "ModelConfig", "SdmiError", "qkv_attention", "mpk_list", "synthetic"]
