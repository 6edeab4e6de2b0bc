"""Per-kernel summary of a `rocprofv3 --pmc ... --output-format csv` run (the *_counter_collection.csv).

usage: python tools/pmc_kernels.py <dir> [name-substring ...]

For every kernel (optionally only those whose name contains one of the substrings): dispatches, mean duration, the
mean of every collected counter, and -- when SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE were collected --
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
(GRBM_GUI_ACTIVE is summed over the 8 XCDs on gfx950) and the effective shader clock GRBM_GUI_ACTIVE / 8 / duration.
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    return name.replace("void sdmi::", "").replace("sdmi::", "")


def main():
    d = sys.argv[1]
    subs = [a for a in sys.argv[2:] if not a.startswith("--")]
    # --json=<pmc_summary.json>:<config key>: the time-weighted matrix-pipe-busy share of the GEMM and attention kernels is merged into that summary
    # (configs[key]["mfma_busy"]), next to the byte counters bench.py quotes
    jarg = next((a for a in sys.argv[2:] if a.startswith("--json=")), None)
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(float)
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if subs and not any(s in k for s in subs):
                    continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                did = row["Dispatch_Id"]
                if did not in disp[k]:
                    disp[k].add(did)
                    dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    classes = {"gemm": [0.0, 0.0], "attention": [0.0, 0.0]}
    per_kernel = {}
    for k in sorted(acc, key=lambda n: -dur[n]):
        n = len(disp[k])
        c = {name: v / n for name, v in acc[k].items()}
        us = dur[k] / n / 1e3
        line = f"{k:58s} n={n:5d} {us:9.1f} us"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            gui = c["GRBM_GUI_ACTIVE"] / 8.0
            busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * gui)
            line += f"  mfma_busy {busy * 100:5.1f} %  clock {gui / us / 1e3:4.2f} GHz"
            cls = "gemm" if "conv_gemm" in k or "conv3_gemm" in k else "attention" if "attn" in k else None
            if cls:
                classes[cls][0] += busy * dur[k]
                classes[cls][1] += dur[k]
                per_kernel[k] = {"mfma_busy": busy, "share_of_class_time": dur[k], "avg_us": us, "clock_ghz": gui / us / 1e3}
        if "SQ_LDS_BANK_CONFLICT" in c:
            line += f"  lds_conflict_cycles {c['SQ_LDS_BANK_CONFLICT']:.3g}"
        if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES"):
            line += f"  wait_inst/wave_cycles {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'] * 100:4.1f} %"
        print(line)
    if jarg:
        import json
        path, key = jarg[len("--json="):].rsplit(":", 1)
        doc = json.load(open(path))
        out = {}
        for cls, (w, t) in classes.items():
            if t > 0:
                ks = {k: dict(v, share_of_class_time=v["share_of_class_time"] / t) for k, v in per_kernel.items() if ("attn" in k) == (cls == "attention")}
                top = dict(sorted(ks.items(), key=lambda kv: -kv[1]["share_of_class_time"])[:3])
                out[cls] = {"mfma_busy_time_weighted": w / t, "top_kernels": top}
        doc.setdefault("configs", {}).setdefault(key, {})["mfma_busy"] = dict(out, source="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), "
                                                                                      "weighted by kernel time (tools/pmc_kernels.py)")
        json.dump(doc, open(path, "w"), indent=1)
        print({c: round(v["mfma_busy_time_weighted"], 3) for c, v in out.items()})


if __name__ == "__main__":
    main()
