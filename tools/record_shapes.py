"""Record every implicit-GEMM shape one sample_image call launches (full-size model, B=1):
    python tools/record_shapes.py gpurun_out/shapes_b1.txt
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from stable_diffusion_burn_amd import ModelConfig, StableDiffusion, synthetic as syn  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/shapes_b1.txt"
sd = StableDiffusion(ModelConfig())
sd.load_weights(syn.SyntheticWeights())
sd.set_option("record_shapes", 1)
sd.sample_image(syn.cond_context(0)[None], syn.uncond_context(), 7.5, 1, init_latent=syn.initial_latent(0)[None])
Path(out).parent.mkdir(parents=True, exist_ok=True)
sd.set_option("dump_shapes", out)
sd.set_option("dump_choices", out + ".choices")
sd.set_option("record_shapes", 0)
print(Path(out + ".choices").read_text())
