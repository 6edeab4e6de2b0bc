#!/bin/bash
# round 4, call s: rocprofv3 kernel traces of the two remaining bench configurations (bf16 batch 8 = configs[3]'s shard, precision 2 batch 16 = configs[4]'s shard)
out=gpurun_out/r04s; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in 3 4; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof$c.log 2>&1
  echo "rocprof config $c rc=$?"
  f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_config$c.csv; rm -rf $R/$out/prof
  head -6 $R/$out/kernel_stats_config$c.csv | cut -c1-150
done
