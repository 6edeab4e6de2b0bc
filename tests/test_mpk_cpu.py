"""tools/mpk_to_dump.py: Burn named-MessagePack record -> npy-dump tree.  UNPINNED (no Burn record available offline):
this only checks the converter against its own inverse and against the tensor names / shapes the oracle (= the
reference's dump tree, see test_reference_python_cpu.py) asks for."""
import sys
from pathlib import Path

import numpy as np
import pytest
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


# ---- the native C++ reader (csrc/mpk_reader.cpp) against records written by msgpack-python --------------------------
def _list(path):
    from stable_diffusion_burn_amd import mpk_list
    return mpk_list(path)


def test_cpp_reader_indexes_the_same_tensors_as_the_python_walker(tmp_path):
    """Names, shapes and the bytes at the reported file offsets of the C++ MessagePack walker == the Python converter's
    view of the same record (which itself is only self-consistent: the burn 0.14 layout is UNPINNED, see mpk_reader.hpp)."""
    d = O.Dims(32, 1, 32, 8, 8, 32)
    cd = CO.ClipDims(n_vocab=300, n_state=32, n_head=1, n_ctx=8, n_layer=2)   # 300 x 32 floats: a bin32-sized tensor
    want = _expected_tensors(d, cd)
    rec = tmp_path / "model.mpk"
    M.write_record(want, rec)
    got = _list(rec)
    assert {n for n, _, _ in got} == set(want)
    raw = rec.read_bytes()
    for name, shape, off in got:
        a = np.asarray(want[name], dtype="<f4")
        assert shape == a.shape, name
        assert off > 0 and raw[off:off + a.nbytes] == a.tobytes(), name


def test_cpp_reader_accepts_the_tolerated_variants(tmp_path):
    """legacy {"value": [...]} tensors, dtype as an externally tagged enum, bytes as a plain integer array (no serde_bytes),
    integer map keys, a record without the BurnRecord wrapper, constants / None as nil, unit modules as empty maps."""
    import msgpack
    t1 = np.arange(6, dtype="<f4").reshape(2, 3)
    item = {"diffusion": {"norm_out": {"gamma": {"id": "a", "param": {"value": [1.0, 2.5], "shape": [2]}},
                                       "beta": {"id": "b", "param": {"bytes": list(np.array([3, 4], "<f4").tobytes()), "shape": [2], "dtype": {"F32": None}}},
                                       "eps": None, "n_group": None},
                          "silu_out": {},
                          "conv_out": {"weight": {"id": "c", "param": {"bytes": t1.tobytes(), "shape": [2, 3], "dtype": "F32"}}, "bias": None,
                                       "stride": [None, None]}},
            "clip": {"blocks": [{"attn_ln": {"gamma": {"id": "g", "param": {"value": [7], "shape": [1]}}}}], "position_embedding": {"id": "p", "param": {"value": [0.5] * 4, "shape": [2, 2]}}},
            "alpha_cumulative_products": {"id": "z", "param": {"value": [0.9, 0.8, 0.7], "shape": [3]}},
            "n_steps": None, 7: "integer key"}
    p = tmp_path / "bare.mpk"
    p.write_bytes(msgpack.packb(item, use_bin_type=True))
    got = {n: s for n, s, _ in _list(p)}
    assert got == {"unet/norm_out/weight": (2,), "unet/norm_out/bias": (2,), "unet/conv_out/weight": (2, 3),
                   "clip/blocks/0/attn_ln/weight": (1,), "clip/position_embedding/weight": (2, 2), "alphas_cumprod": (3,)}


@pytest.mark.parametrize("damage", ["truncate", "half_precision", "bad_length", "garbage", "empty", "wrapping_shape", "huge_value_array", "huge_byte_array"])
def test_cpp_reader_rejects_bad_records(tmp_path, damage):
    import msgpack
    from stable_diffusion_burn_amd import SdmiError
    good = {"metadata": {"float": "f32", "int": "i32", "format": "x", "version": "0.14.0", "settings": "FullPrecisionSettings"},
            "item": {"diffusion": {"conv_out": {"weight": {"id": "c", "param": {"bytes": np.zeros(12, "<f4").tobytes(), "shape": [3, 4], "dtype": "F32"}}}}}}
    blob = msgpack.packb(good, use_bin_type=True)
    if damage == "truncate":
        blob = blob[:-9]
    elif damage == "half_precision":
        good["item"]["diffusion"]["conv_out"]["weight"]["param"]["dtype"] = "F16"
        blob = msgpack.packb(good, use_bin_type=True)
    elif damage == "bad_length":
        good["item"]["diffusion"]["conv_out"]["weight"]["param"]["shape"] = [3, 5]
        blob = msgpack.packb(good, use_bin_type=True)
    elif damage == "wrapping_shape":
        # 2^62 + 1 elements: count * 4 wraps to 4, which is exactly the payload's length -- must be rejected, not indexed
        good["item"]["diffusion"]["conv_out"]["weight"]["param"] = {"bytes": b"\x00" * 4, "shape": [2 ** 62 + 1], "dtype": "F32"}
        blob = msgpack.packb(good, use_bin_type=True)
    elif damage == "huge_value_array":
        # a `value` array header that claims 2^31 elements in a 100-byte file: no allocation of that size may happen before the elements are checked
        hdr = msgpack.packb({"metadata": {}, "item": {"diffusion": {"conv_out": {"weight": {"id": "c", "param": {"shape": [2 ** 31], "value": []}}}}}}, use_bin_type=True)
        blob = hdr[:-1] + bytes([0xdd, 0x80, 0x00, 0x00, 0x00])          # array32 with 2^31 entries, then nothing
    elif damage == "huge_byte_array":
        hdr = msgpack.packb({"metadata": {}, "item": {"diffusion": {"conv_out": {"weight": {"id": "c", "param": {"shape": [2 ** 29], "dtype": "F32", "bytes": []}}}}}}, use_bin_type=True)
        blob = hdr[:-1] + bytes([0xdd, 0x80, 0x00, 0x00, 0x00])
    elif damage == "garbage":
        blob = bytes([0xc1]) * 64          # 0xc1 is the one reserved MessagePack type byte
    elif damage == "empty":
        blob = msgpack.packb({"metadata": {}, "item": {"n_steps": None}}, use_bin_type=True)
    p = tmp_path / "bad.mpk"
    p.write_bytes(blob)
    with pytest.raises(SdmiError):
        _list(p)
    with pytest.raises(SdmiError):
        _list(tmp_path / "does_not_exist.mpk")
