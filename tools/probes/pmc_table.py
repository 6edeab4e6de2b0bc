"""Per-kernel means of a rocprofv3 --pmc pass (counter_collection.csv files under a directory): matrix-pipe busy and where the waves' time goes."""
import csv, glob, re, sys
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set)
for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void sdmi::", "")
        if "gemm" not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
for k in sorted(acc):
    n = len(disp[k]); c = {a: v / n for a, v in acc[k].items()}
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    if not w:
        continue
    print(f"{k:60s} n={n:3d} mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8) * 100:5.1f} %  gui_active/8 {c['GRBM_GUI_ACTIVE'] / 8:9.0f}  "
          f"of wave cycles: issuing {c['SQ_ACTIVE_INST_ANY'] / w * 100:4.1f} % (valu {c['SQ_ACTIVE_INST_VALU'] / w * 100:4.1f}, lds {c['SQ_ACTIVE_INST_LDS'] / w * 100:4.1f}) "
          f"issue-stalled {c['SQ_WAIT_INST_ANY'] / w * 100:4.1f} % (lds {c['SQ_WAIT_INST_LDS'] / w * 100:4.1f}) parked {c['SQ_WAIT_ANY'] / w * 100:4.1f} %")
