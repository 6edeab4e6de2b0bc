"""Pin the CLIP / tokenizer oracles against the REFERENCE'S OWN Python side (run in this container only).

    python tests/golden/gen_clip_from_reference_python.py        # ~1 min

Same method as gen_from_reference_python.py: python/dump.py's CLIPTextTransformer is instantiated
through the tinygrad-API shim with every parameter a unique constant, the reference's exporter
(python/clip.py: save_clip_text_transformer) writes the dump tree -- which maps every dump file name the
Rust loader reads (src/model/clip/load.rs) to a Python attribute -- the seeded synthetic tensor of each
DUMP NAME is installed, and the reference Python forward is evaluated:

  refpy_clip.npz     token sequences (T = 2, 17, 77) and the reference-Python CLIP outputs (fp64;
                     every 8th row for T = 77)
  refpy_tokens.json  prompts and the ids python/tokenizer.py (SimpleTokenizer, ftfy stubbed to identity:
                     ftfy is not installed and the Rust tokenizer has no such step) produces with the
                     reference's merges file, plus the Rust test vector of src/tokenizer.rs:213-215

tests/test_clip_cpu.py checks oracle == these fixtures without /root/reference.
"""
import contextlib
import io
import json
import shutil
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
REF_PY = Path("/root/reference/python")

import tinygrad_shim as shim  # noqa: E402

from gen_from_reference_python import collect_params, synth_for  # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402

PROMPTS = [
    "",
    "A photo of an astronaut riding a horse on mars.",
    "Hello world! <|startoftext|>asdf<|startoftext|>",
    "  an   oil painting\tof a cat's whiskers, it's 4k; don't they'll we've I'm you'd we're  ",
    "UPPER lower MiXeD 1234567890 3.14159 #hashtag @user (parens) [brackets] {braces} ... --- !!!",
    "a cyberpunk city at night, neon lights, rain, highly detailed, 8k, trending on artstation",
    "café naïve résumé über straße 中文 日本語 \U0001F680 rocket",
    "supercalifragilisticexpialidocious antidisestablishmentarianism pneumonoultramicroscopicsilicovolcanoconiosis",
    "x", "'", "''s", "a'b", "½ cup of 100% pure a_b-c/d\\e",
]


def main():
    torch.set_num_threads(min(16, torch.get_num_threads()))
    shim.install()
    sys.path.insert(0, str(REF_PY))
    sys.modules["ftfy"] = types.SimpleNamespace(fix_text=lambda s: s)   # not installed; identity (see docstring)
    import dump           # python/dump.py
    import clip as clip_save   # python/clip.py (exporter)
    import tokenizer as ref_tok  # python/tokenizer.py

    # ---- tokenizer ---------------------------------------------------------------------------------------
    tok = ref_tok.SimpleTokenizer()
    rows = [{"text": p, "ids": [int(i) for i in tok.encode(p)]} for p in PROMPTS]
    doc = {"source": "python/tokenizer.py SimpleTokenizer.encode (ftfy stubbed to identity) on python/bpe_simple_vocab_16e6.txt.gz",
           "rust_kat": {"text": "Hello world! <|startoftext|>asdf<|startoftext|>", "ids": [3306, 1002, 256, 49406, 587, 10468, 49406],
                        "decoded": "hello world ! <|startoftext|>asdf <|startoftext|>", "where": "src/tokenizer.rs:213-215"},
           "prompts": rows}
    (HERE / "refpy_tokens.json").write_text(json.dumps(doc, indent=1, ensure_ascii=True) + "\n")
    print(f"wrote refpy_tokens.json ({len(rows)} prompts)")

    # ---- CLIP ------------------------------------------------------------------------------------------------
    model = dump.CLIPTextTransformer()
    params = []
    collect_params(model, set(), params)
    by_index = {int(p.t.flatten()[0].item()): p for p in params}
    tmp = Path(tempfile.mkdtemp(prefix="refclip_"))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            clip_save.save_clip_text_transformer(model, tmp / "clip")
        mapping, shapes = {}, {}
        extra = sorted(str(f.relative_to(tmp))[:-4] for f in tmp.rglob("*.npy") if f.stem not in ("weight", "bias"))
        for f in sorted(tmp.rglob("*.npy")):
            name = str(f.relative_to(tmp))[:-4]
            if name.rsplit("/", 1)[1] not in ("weight", "bias"):
                continue
            raw = np.load(f, mmap_mode="r")
            for d in (1, 2):
                dims = [int(v) for v in raw[:d]]
                if len(raw) == d + int(np.prod(dims)) and all(v > 0 for v in dims):
                    break
            else:
                raise RuntimeError(f"cannot parse {f}")
            p = by_index[int(raw[d])]
            is_table = "embedding" in name                       # save_embedding does not transpose (save.py:97-99)
            transposed = len(dims) == 2 and not is_table         # save_linear does (save.py:19)
            assert tuple(p.shape) == (tuple(dims[::-1]) if transposed else tuple(dims)), (name, p.shape, dims)
            mapping[name] = (p, transposed, tuple(dims))
            shapes[name] = tuple(dims)
        print(f"mapped {len(mapping)} CLIP dump tensors; non-tensor files: {len(extra)} (n_head / n_layer / eps)")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    W = syn.SyntheticWeights()
    for name, (p, transposed, dims) in mapping.items():
        t = torch.from_numpy(np.ascontiguousarray(synth_for(name, dims, shapes, W))).to(shim.DTYPE)
        p.t = t.t().contiguous() if transposed else t

    g = np.random.default_rng(7)
    seqs = {"probe": np.array([3, 1]),            # the commented probe of dump.py:603-611: clip(Tensor([3, 1]).unsqueeze(0))
            "t2": np.array([49406, 49407]),
            "t17": np.array([49406, 320, 1125, 539, 550, 18376, 6765, 320, 4558, 267, 847, 713, 14124, 272, 273, 274, 49407]),
            "t77": np.concatenate(([49406], g.integers(0, 49406, 75), [49407]))}
    out = {"dump_names": np.array(sorted(mapping))}
    for key, ids in seqs.items():
        y = model(shim.Tensor(ids[None].astype(np.int64))).numpy()[0]
        out[f"{key}_tokens"] = ids.astype(np.int32)
        out[f"{key}_out"] = (y[::8] if key == "t77" else y).astype(np.float64)
        print(f"reference-python CLIP forward T={len(ids)}: absmax {np.abs(y).max():.3f}")
    np.savez_compressed(HERE / "refpy_clip.npz", **out)
    print("wrote refpy_clip.npz")


if __name__ == "__main__":
    main()
