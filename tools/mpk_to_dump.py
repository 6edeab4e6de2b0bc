"""Convert a Burn `NamedMpkFileRecorder<FullPrecisionSettings>` record (the reference's SDv1-4.mpk,
src/bin/sample/main.rs:27-34) into the npy-dump tree that `sdmi_load_weights_dir` / `sdmi_sample dump <dir>` read.

    python tools/mpk_to_dump.py SDv1-4.mpk params/

UNPINNED: no Burn record and no Burn build exist in the environment this was written in, so the layout below is
the documented one of burn 0.14 (the version the reference pins, Cargo.toml:17), not one checked against a real
file; tests/test_mpk_cpu.py only proves the converter and its inverse (write_record) agree with each other and with
the dump names / shapes the engine expects.  The walker is deliberately tolerant of the wrapper levels:

  file    = MessagePack map {"metadata": {...}, "item": <module record>}      (rmp_serde "named" encoding)
  module  = map field-name -> module | Vec<module> (array) | Option (nil) | parameter | constant (nil / scalar)
  param   = {"id": str, "param": tensor}                                         (ParamSerde)
  tensor  = {"bytes": bin, "shape": [..], "dtype": "F32"}                        (TensorData, burn >= 0.14)
          | {"value": [f32 ..], "shape": [..]}                                    (DataSerialize, burn <= 0.13)

Field names are the Rust struct fields, which the reference's exporters also use as dump directory names
(src/model/*/load.rs); the differences are mapped here:
  StableDiffusion.diffusion -> unet/, .alpha_cumulative_products -> alphas_cumprod,
  GroupNorm / LayerNorm gamma, beta -> weight, bias;  CLIP.position_embedding (a bare Param) -> position_embedding/weight.
Linear weights are [in, out] and conv weights [Cout, Cin, kh, kw] in both formats.
"""
from __future__ import annotations

import sys
from pathlib import Path

import msgpack
import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stable_diffusion_burn_amd import weights as wio  # noqa: E402

RENAME_TOP = {"diffusion": "unet"}
RENAME_LEAF = {"gamma": "weight", "beta": "bias"}


def _tensor(node):
    """(ndarray float32) if `node` is a serialized tensor, else None."""
    if not isinstance(node, dict):
        return None
    if "bytes" in node and "shape" in node:
        dt = node.get("dtype", "F32")
        dt = dt if isinstance(dt, str) else (list(dt.keys())[0] if isinstance(dt, dict) else str(dt))
        if dt.upper() not in ("F32", "FLOAT32"):
            raise ValueError(f"tensor dtype {dt!r}: only full-precision (f32) records are supported")
        raw = node["bytes"]
        raw = bytes(raw) if not isinstance(raw, (bytes, bytearray)) else raw
        return np.frombuffer(raw, dtype="<f4").reshape([int(v) for v in node["shape"]]).copy()
    if "value" in node and "shape" in node:
        return np.asarray(node["value"], dtype=np.float32).reshape([int(v) for v in node["shape"]])
    return None


def walk(node, path, out):
    t = _tensor(node)
    if t is not None:
        out["/".join(path)] = t
        return
    if isinstance(node, dict):
        if "param" in node and "id" in node:          # ParamSerde wrapper
            walk(node["param"], path, out)
            return
        for k, v in node.items():
            k = k.decode() if isinstance(k, bytes) else str(k)
            walk(v, path + [k], out)
    elif isinstance(node, (list, tuple)):
        for i, v in enumerate(node):
            walk(v, path + [str(i)], out)


def dump_name(path: str) -> str:
    seg = path.split("/")
    if seg[0] == "alpha_cumulative_products":
        return "alphas_cumprod"
    seg[0] = RENAME_TOP.get(seg[0], seg[0])
    seg[-1] = RENAME_LEAF.get(seg[-1], seg[-1])
    if seg[-1] not in ("weight", "bias"):
        seg.append("weight")                           # a bare Param field (clip.position_embedding)
    return "/".join(seg)


def read_record(path) -> dict:
    """{dump name: float32 ndarray} of every tensor in the record."""
    with open(path, "rb") as f:
        doc = msgpack.unpackb(f.read(), raw=False, strict_map_key=False)
    item = doc["item"] if isinstance(doc, dict) and "item" in doc else doc
    found = {}
    walk(item, [], found)
    return {dump_name(k): v for k, v in found.items()}


def write_dump(tensors: dict, out_dir, n_head: int = 8, clip_heads: int = 12) -> None:
    specs = [(n, tuple(a.shape)) for n, a in tensors.items()]
    wio.write_dump_tree(out_dir, specs, lambda name, shape: tensors[name], tensors["alphas_cumprod"], n_head=n_head, clip_heads=clip_heads)


def write_record(tensors: dict, path) -> None:
    """Inverse of read_record for tests: a record with the layout described above (TensorData flavour)."""
    root = {}
    for name, a in tensors.items():
        seg = name.split("/")
        if name == "alphas_cumprod":
            seg = ["alpha_cumulative_products"]
        else:
            seg[0] = {v: k for k, v in RENAME_TOP.items()}.get(seg[0], seg[0])
            parent = seg[-2] if len(seg) > 1 else ""
            is_norm = parent.startswith("norm") or parent.endswith("_ln") or parent in ("layer_norm", "norm")
            if is_norm:
                seg[-1] = {"weight": "gamma", "bias": "beta"}[seg[-1]]
            if seg[-2:] == ["position_embedding", "weight"]:
                seg = seg[:-1]
        node = root
        for s in seg[:-1]:
            node = node.setdefault(s, {})
        a = np.ascontiguousarray(a, dtype="<f4")
        node[seg[-1]] = {"id": name, "param": {"bytes": a.tobytes(), "shape": list(a.shape), "dtype": "F32"}}

    def listify(node):   # maps whose keys are all decimal indices are Vec<Module>
        if not isinstance(node, dict) or ("id" in node and "param" in node):
            return node
        node = {k: listify(v) for k, v in node.items()}
        if node and all(k.isdigit() for k in node):
            return [node[str(i)] for i in range(len(node))]
        return node

    doc = {"metadata": {"float": "f32", "int": "i32", "format": "burn_core::record::file::NamedMpkFileRecorder<FullPrecisionSettings>",
                        "version": "0.14.0", "settings": "FullPrecisionSettings"},
           "item": listify(root)}
    with open(path, "wb") as f:
        f.write(msgpack.packb(doc, use_bin_type=True))


def main():
    if len(sys.argv) != 3:
        print(__doc__)
        sys.exit(1)
    tensors = read_record(sys.argv[1])
    print(f"{len(tensors)} tensors, {sum(a.size for a in tensors.values()) / 1e6:.1f} M parameters")
    write_dump(tensors, sys.argv[2])
    print(f"wrote {sys.argv[2]}")


if __name__ == "__main__":
    main()
