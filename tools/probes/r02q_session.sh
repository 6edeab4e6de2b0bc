#!/bin/bash
# where do the waves of the bf16 / fp8 large-tile GEMMs spend their time (parked vs issue-stalled)?
out=gpurun_out/r02q; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc1 -- python $R/tools/probes/pmc_shapes.py --bf16 > $R/$out/pmc1.log 2>&1
cd $R
python - <<'PY'
import csv, glob, re
from collections import defaultdict
d = "gpurun_out/r02q/pmc1"
acc = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set)
for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void sdmi::", "")
        if not any(s in k for s in ("gemm", "attn")): continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
for k in acc:
    n = len(disp[k]); c = {a: v / n for a, v in acc[k].items()}
    w = c["SQ_WAVE_CYCLES"]
    print(f"{k:50s} n={n} mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8) * 100:5.1f} %  of wave cycles: active {c['SQ_ACTIVE_INST_ANY'] / w * 100:4.1f} % (valu {c['SQ_ACTIVE_INST_VALU'] / w * 100:4.1f}, lds {c['SQ_ACTIVE_INST_LDS'] / w * 100:4.1f}) issue-stall {c['SQ_WAIT_INST_ANY'] / w * 100:4.1f} % (lds {c['SQ_WAIT_INST_LDS'] / w * 100:4.1f}) parked {c['SQ_WAIT_ANY'] / w * 100:4.1f} %")
PY
rm -rf $out/pmc1
