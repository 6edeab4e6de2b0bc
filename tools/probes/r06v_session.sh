#!/bin/bash
out=gpurun_out/r06v; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $R/$out/p$i -- python $R/tools/probes/r06v_gemm_pmc.py > $R/$out/p$i.log 2>&1; echo "pass $i rc=$?"
  (cd $R && python tools/pmc_kernels.py $out/p$i conv_gemm_bf16x conv3_gemm attn_bf16 | python -c "
import sys
for l in sys.stdin: print(l.rstrip()[:60], end=' ')
" ; echo) > /dev/null
  (cd $R && python - $out/p$i <<'PY'
import csv, glob, sys, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void sdmi::", "")
        if not any(s in k for s in ("conv_gemm_bf16x", "conv3_gemm", "attn_bf16")): continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
for k in acc:
    print(f"{k:50s} n={len(n[k]):3d} " + "  ".join(f"{c}={v / len(n[k]):.4g}" for c, v in sorted(acc[k].items())))
PY
  ) >> $R/$out/r06v_counters.txt
  rm -rf $R/$out/p$i
done
cat $R/$out/r06v_counters.txt
