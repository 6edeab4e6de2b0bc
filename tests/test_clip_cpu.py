"""CLIP text encoder + tokenizer oracles vs the reference's own Python side and its Rust test vector.

Fixtures come from tests/golden/gen_clip_from_reference_python.py (reference python/dump.py CLIP model and
python/tokenizer.py, run once in the build container).  The tokenizer tests need the reference's merges
file (not copied into this repository): $SDMI_BPE_VOCAB or /root/reference/bpe_simple_vocab_16e6.txt.
"""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import clip_oracle as CO
from oracle.tokenizer_oracle import TokenizerOracle
from stable_diffusion_burn_amd import synthetic as syn

GOLD = Path(__file__).parent / "golden"


def vocab_path():
    for p in (os.environ.get("SDMI_BPE_VOCAB"), "/root/reference/bpe_simple_vocab_16e6.txt"):
        if p and Path(p).exists():
            return p
    return None


needs_vocab = pytest.mark.skipif(vocab_path() is None, reason="reference merges file not available")


def test_attn_decoder_mask():
    """backend.rs:130-139."""
    m = CO.attn_decoder_mask(5)
    assert torch.equal(m, torch.full((5, 5), float("-inf")).triu(1))
    assert CO.attn_decoder_mask(1).tolist() == [[0.0]]


def test_clip_oracle_matches_reference_python():
    """oracle CLIP forward == python/dump.py CLIPTextTransformer on the same dump-named weights (fp64)."""
    g = np.load(GOLD / "refpy_clip.npz")
    o = CO.CLIPOracle(syn.SyntheticWeights(), CO.ClipDims(), torch.float64)
    for key in ("probe", "t2", "t17", "t77"):   # "probe": dump.py:603-611, clip(Tensor([3, 1]))
        y = o.forward(g[f"{key}_tokens"][None]).numpy()[0]
        ref = g[f"{key}_out"]
        got = y[::8] if key == "t77" else y
        err = np.abs(got - ref).max()
        print(f"{key}: max|oracle - reference python| = {err:.3e}")
        assert err < 1e-12, key


def test_clip_dump_names_are_the_reference_exporters():
    """every tensor the oracle (and the engine) asks for is a file python/clip.py writes, and vice versa."""
    g = np.load(GOLD / "refpy_clip.npz")
    ref_names = set(str(s) for s in g["dump_names"])
    asked = set()

    class Spy:
        def get(self, name, shape, kind, fan_in=0):
            asked.add(name)
            return np.zeros(shape, np.float32)

    CO.CLIPOracle(Spy(), CO.ClipDims(), torch.float32).forward(np.array([[49406, 49407]]))
    assert asked == ref_names, (sorted(asked - ref_names)[:5], sorted(ref_names - asked)[:5])


def test_causality():
    """a token's embedding does not depend on later tokens (the decoder mask)."""
    d = CO.ClipDims(n_vocab=50, n_state=64, n_head=1, n_ctx=16, n_layer=2)
    o = CO.CLIPOracle(syn.SyntheticWeights(), d, torch.float64)
    a = o.forward(np.array([[1, 2, 3, 4, 5, 6]])).numpy()[0]
    b = o.forward(np.array([[1, 2, 3, 9, 9, 9]])).numpy()[0]
    assert np.abs(a[:3] - b[:3]).max() < 1e-12 and np.abs(a[3:] - b[3:]).max() > 1e-3


@needs_vocab
def test_tokenizer_rust_kat():
    """src/tokenizer.rs:209-221 test_encode_decode."""
    doc = json.loads((GOLD / "refpy_tokens.json").read_text())["rust_kat"]
    t = TokenizerOracle(vocab_path())
    ids = t.encode(doc["text"])
    assert ids == doc["ids"]
    assert t.decode(ids) == doc["decoded"]
    assert len(t.encoder) == 49408 and t.encoder["<|startoftext|>"] == 49406 and t.encoder["<|endoftext|>"] == 49407


@needs_vocab
def test_tokenizer_matches_reference_python():
    doc = json.loads((GOLD / "refpy_tokens.json").read_text())
    t = TokenizerOracle(vocab_path())
    for row in doc["prompts"]:
        assert t.encode(row["text"]) == row["ids"], row["text"]


@needs_vocab
def test_context_tokens_unpadded():
    """StableDiffusion::context (stablediffusion/mod.rs:198-205): start + text + end, no padding (quirk Q2)."""
    t = TokenizerOracle(vocab_path())
    assert t.context_tokens("") == [49406, 49407]
    ids = t.context_tokens("a photo of a cat")
    assert ids[0] == 49406 and ids[-1] == 49407 and len(ids) == 7
