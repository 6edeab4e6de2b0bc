#!/bin/bash
# PMC pass over the split-operand kernels
out=gpurun_out/r02l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 python $R/tools/probes/pmc_split.py > $R/$out/plain.txt 2>&1; cat $R/$out/plain.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc1 -- python $R/tools/probes/pmc_split.py > $R/$out/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc2 -- python $R/tools/probes/pmc_split.py > $R/$out/pmc2.log 2>&1
cd $R
python tools/pmc_kernels.py $out/pmc1 > $out/pmc1_kernels.txt 2>&1
python tools/pmc_kernels.py $out/pmc2 > $out/pmc2_kernels.txt 2>&1
python - <<'PY'
import csv, glob, re
from collections import defaultdict
for d in ("gpurun_out/r02l/pmc1", "gpurun_out/r02l/pmc2"):
    acc = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set)
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void sdmi::", "")
            if not any(s in k for s in ("gemm3x", "attn", "gemm2x")): continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
    for k in acc:
        n = len(disp[k]); print(k, "n=%d" % n, {c: "%.4g" % (v / n) for c, v in acc[k].items()})
PY
rm -rf $out/pmc1 $out/pmc2
