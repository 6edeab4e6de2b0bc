"""The product tokenizer (C++, inside libsdmi.so; src/tokenizer.rs twin) against the oracle tokenizer, the
reference's Rust test vector and the ids of the reference's python/tokenizer.py.  Host code: no GPU needed."""
import json
import os
import random
from pathlib import Path

import pytest

from oracle.tokenizer_oracle import TokenizerOracle

GOLD = Path(__file__).parent / "golden"
MINI = GOLD / "mini_merges.txt"


def vocab_path():
    for p in (os.environ.get("SDMI_BPE_VOCAB"), "/root/reference/bpe_simple_vocab_16e6.txt"):
        if p and Path(p).exists():
            return p
    return None


needs_vocab = pytest.mark.skipif(vocab_path() is None, reason="reference merges file not available")

ALPHABET = "abc XYZ'’ \t\n.,!?-_/0123456789éßΣσςİıǅ中文\U0001F680½²ªº  ́ſʰ·:<|>"


def _fuzz(tok, ora, n, seed):
    rnd = random.Random(seed)
    for _ in range(n):
        s = "".join(rnd.choice(ALPHABET) for _ in range(rnd.randint(0, 24)))
        a, b = tok.encode(s), ora.encode(s)
        assert a == b, repr(s)
        assert tok.decode(a) == ora.decode(b), repr(s)


def test_mini_vocab_matches_oracle():
    """always-on: a small merges file of this repository's own (tests/golden/gen_mini_merges.py)"""
    from stable_diffusion_burn_amd import SimpleTokenizer
    tok, ora = SimpleTokenizer(MINI), TokenizerOracle(MINI)
    assert tok.vocab_size == len(ora.encoder) == 512 + 264 + 2
    for text in ["", "a photo of an astronaut riding a horse", "Hello  World!!  it's   <|startoftext|>x<|endoftext|>",
                 "painting's PAINTED 123 4k", "éè 中文 \U0001F680"]:
        assert tok.encode(text) == ora.encode(text), text
    ids = tok.encode("<|startoftext|>a cat<|endoftext|>")
    assert ids[0] == tok.vocab_size - 2 and ids[-1] == tok.vocab_size - 1
    _fuzz(tok, ora, 1500, 0)


def test_missing_merges_file_is_an_error():
    from stable_diffusion_burn_amd import SdmiError, SimpleTokenizer
    with pytest.raises(SdmiError):
        SimpleTokenizer("/nonexistent/merges.txt")


@needs_vocab
def test_rust_kat():
    """src/tokenizer.rs:209-221 test_encode_decode, run on the product tokenizer."""
    from stable_diffusion_burn_amd import SimpleTokenizer
    doc = json.loads((GOLD / "refpy_tokens.json").read_text())["rust_kat"]
    tok = SimpleTokenizer(vocab_path())
    assert tok.vocab_size == 49408
    ids = tok.encode(doc["text"])
    assert ids == doc["ids"]
    assert tok.decode(ids) == doc["decoded"]


@needs_vocab
def test_reference_python_ids():
    from stable_diffusion_burn_amd import SimpleTokenizer
    doc = json.loads((GOLD / "refpy_tokens.json").read_text())
    tok = SimpleTokenizer(vocab_path())
    for row in doc["prompts"]:
        assert tok.encode(row["text"]) == row["ids"], row["text"]


@needs_vocab
def test_full_vocab_fuzz_matches_oracle():
    from stable_diffusion_burn_amd import SimpleTokenizer
    tok, ora = SimpleTokenizer(vocab_path()), TokenizerOracle(vocab_path())
    _fuzz(tok, ora, 1500, 1)
