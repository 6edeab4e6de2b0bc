"""tools/mpk_to_dump.py: Burn named-MessagePack record -> npy-dump tree.  UNPINNED (no Burn record available offline):
this only checks the converter against its own inverse and against the tensor names / shapes the oracle (= the
reference's dump tree, see test_reference_python_cpu.py) asks for."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import clip_oracle as CO  # noqa: E402
from oracle import sd_oracle as O  # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402
from stable_diffusion_burn_amd import weights as wio  # noqa: E402
from tools import mpk_to_dump as M  # noqa: E402


def _expected_tensors(d, cd):
    """every (dump name -> tensor) the oracles request for a small model"""
    seen = {}

    class Spy:
        def __init__(self):
            self.w = syn.SyntheticWeights()

        def get(self, name, shape, kind, fan_in=0):
            seen[name] = self.w.get(name, shape, kind, fan_in)
            return seen[name]

    spy = Spy()
    lat = torch.zeros(1, 4, d.latent_h, d.latent_w)
    O.UNetOracle(spy, d, torch.float32).forward(lat, 5, torch.zeros(1, 3, d.ctx_dim))
    enc = O.EncoderOracle(spy, d, torch.float32)
    enc.decode_latent(enc.encode_image(torch.zeros(1, 3, 8 * d.latent_h, 8 * d.latent_w)))
    CO.CLIPOracle(spy, cd, torch.float32).forward(np.array([[1, 2]]))
    seen["alphas_cumprod"] = syn.alphas_cumprod()
    return seen


def test_record_round_trip_and_dump_tree(tmp_path):
    d = O.Dims(32, 1, 32, 8, 8, 32)
    cd = CO.ClipDims(n_vocab=40, n_state=32, n_head=1, n_ctx=8, n_layer=2)
    want = _expected_tensors(d, cd)
    M.write_record(want, tmp_path / "model.mpk")
    got = M.read_record(tmp_path / "model.mpk")
    assert set(got) == set(want)
    for k in want:
        assert got[k].shape == tuple(np.shape(want[k])) and np.array_equal(got[k], want[k]), k
    # and on to the dump tree the C++ loader reads (same files as the reference's exporters write)
    M.write_dump(got, tmp_path / "params", n_head=1, clip_heads=1)
    for k, a in want.items():
        f = tmp_path / "params" / (k + ".npy")
        assert f.exists(), k
        assert np.array_equal(wio.read_tensor(f, np.ndim(a)), a), k


def test_legacy_data_serialize_flavour_and_names():
    """burn <= 0.13 stored {"value": [...], "shape": [...]}; GroupNorm / LayerNorm use gamma / beta; diffusion -> unet."""
    item = {"diffusion": {"norm_out": {"gamma": {"id": "x", "param": {"value": [1.0, 2.0], "shape": [2]}},
                                       "beta": {"id": "y", "param": {"value": [3.0, 4.0], "shape": [2]}}},
                          "input_blocks": {"d1": {"weight": {"id": "z", "param": {"value": [0.0] * 8, "shape": [2, 1, 2, 2]}}}}},
            "clip": {"position_embedding": {"id": "p", "param": {"value": [0.5] * 6, "shape": [3, 2]}},
                     "blocks": [{"attn_ln": {"gamma": {"id": "g", "param": {"value": [1.0], "shape": [1]}}}}]},
            "alpha_cumulative_products": {"id": "a", "param": {"value": [0.9, 0.8], "shape": [2]}},
            "n_steps": None}
    out = {}
    M.walk(item, [], out)
    names = {M.dump_name(k): v.shape for k, v in out.items()}
    assert names == {"unet/norm_out/weight": (2,), "unet/norm_out/bias": (2,), "unet/input_blocks/d1/weight": (2, 1, 2, 2),
                     "clip/position_embedding/weight": (3, 2), "clip/blocks/0/attn_ln/weight": (1,), "alphas_cumprod": (2,)}
