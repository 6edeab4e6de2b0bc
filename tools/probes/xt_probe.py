import sys
sys.path.insert(0, '/root/repo')
import os
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
for s in [(32, 320, 64, 64, 320, 3, 1, 0), (32, 960, 64, 64, 320, 3, 1, 0), (32, 320, 64, 64, 320, 1, 1, 0), (16, 320, 64, 64, 320, 3, 1, 0)]:
    ms = sd.bench_conv(*s, 100, 1, 5)
    n, cin, h, w, cout, k = s[:6]
    M, N, K = n * h * w, cout, cin * k * k
    print(s, f"{ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.0f} TF", flush=True)
