"""Which tile every GEMM of one bf16 forward chooses at batch B (CFG batch 2 B), with the time of each (M, N, K, kind) from bench_conv."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np  # noqa: E402
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion, synthetic as syn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r04w/choices_b%d.txt" % B
sd = StableDiffusion(ModelConfig(precision=1))
sd.load_weights(syn.SyntheticWeights())
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    sd.set_option(k, v)
cond = np.stack([syn.cond_context(i) for i in range(B)])
lat = np.stack([syn.initial_latent(i) for i in range(B)])
sd.sample_image(cond, syn.uncond_context(), 7.5, 1, init_latent=lat)      # warm-up: weights staged
sd.set_option("record_shapes", 1)
sd.sample_image(cond, syn.uncond_context(), 7.5, 1, init_latent=lat)
Path(out).parent.mkdir(parents=True, exist_ok=True)
sd.set_option("dump_choices", out)
sd.set_option("record_shapes", 0)
print(Path(out).read_text())
