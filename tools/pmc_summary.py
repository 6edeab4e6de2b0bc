"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_summary.json.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <images_in_trace> [out.json] [config_key] [commit]

Each directory holds the `*_counter_collection.csv` of one pass (the two counters do not fit one pass on
gfx950: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).  HBM-side bytes per kernel class are
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- MI355X_MICROARCH.md: both counters are in KiB and FETCH_SIZE reports
half of a wide coalesced stream on gfx950; Infinity-Cache hits are included, so this bounds DRAM bytes from above.

config_key (round 5): "fp32_b1_s20" (the headline, default), "bf16_b16_s50", "bf16_b8_s20", "fp8_b16_s20" -- the summary of one bench.py configuration is merged
into out.json under configs[config_key]; the headline's also stays at the top level (the keys round 2-4 wrote).  `commit` is recorded so that a bench line can say
which tree the byte counters were collected on.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def classify(name: str) -> str:
    if "conv_gemm_fp8" in name:
        return "conv_gemm_fp8"
    if "conv_gemm3x" in name or "conv_gemm3p" in name:
        return "conv_gemm_split"
    if "conv_gemm" in name or "conv3_gemm" in name:
        return "conv_gemm"
    if "splitk_reduce" in name:
        return "splitk_reduce"
    if "attn" in name or "softmax_rows" in name:
        return "attention"
    if "gn_" in name or "group_norm" in name:
        return "group_norm"
    if "layer_norm" in name or "ln_" in name:
        return "layer_norm"
    return "other"


def read(dirname: str, counter: str):
    tot = defaultdict(float)
    launches = defaultdict(int)
    for path in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                c = classify(row["Kernel_Name"])
                tot[c] += float(row["Counter_Value"])
                launches[c] += 1
    return tot, launches


def main():
    fetch_dir, write_dir, images = sys.argv[1], sys.argv[2], float(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_summary.json"
    key = sys.argv[5] if len(sys.argv) > 5 else "fp32_b1_s20"
    commit = sys.argv[6] if len(sys.argv) > 6 else None
    fetch, launches = read(fetch_dir, "FETCH_SIZE")
    write, _ = read(write_dir, "WRITE_SIZE")
    classes = {}
    for c in sorted(launches, key=lambda k: -fetch[k]):
        b = (2.0 * fetch[c] + write[c]) * 1024.0
        classes[c] = {"launches_per_image": launches[c] / images, "fetch_kb": fetch[c], "write_kb": write[c],
                      "hbm_bytes_per_launch": b / launches[c], "hbm_gb_per_image": b / images / 1e9}
    doc = {}
    if os.path.exists(out):
        try:
            doc = json.load(open(out))
        except Exception:  # noqa: BLE001
            doc = {}
    doc["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py [--config N] --steps 1 --warmup 0 "
                     "--no-cpu-baseline --no-roofline --no-secondary ; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md; "
                     "Infinity-Cache hits are counted, so this is an upper bound on DRAM bytes)")
    # the tree the counters were collected on, as the build sees it (csrc/ sources + headers + tile tables + flags): bench.py compares it with the running tree's and
    # marks `traffic_stale` when they differ -- the summary must be written in the same gpurun call as the passes
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from stable_diffusion_burn_amd import build as _b
        digest = _b._digest()[:16]
    except Exception:  # noqa: BLE001
        digest = None
    doc.setdefault("configs", {})[key] = {"images_in_trace": images, "commit": commit, "kernel_source_digest": digest, "classes": classes}
    if key == "fp32_b1_s20":
        doc.update({"images_in_trace": images, "commit": commit, "classes": classes,
                    "conv_gemm_hbm_bytes_per_launch": classes.get("conv_gemm", {}).get("hbm_bytes_per_launch"),
                    "conv_gemm_split_hbm_bytes_per_launch": classes.get("conv_gemm_split", {}).get("hbm_bytes_per_launch")})
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"[{key}]")
    for c, v in classes.items():
        print(f"{c:14s} {v['launches_per_image']:8.0f} launches/img {v['hbm_gb_per_image']:8.1f} GB/img {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB/launch")


if __name__ == "__main__":
    main()
