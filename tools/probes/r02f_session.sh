set -x
R=$PWD
mkdir -p gpurun_out/r02f
python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -m gpu -q -p no:cacheprovider -k "group_norm" > gpurun_out/r02f/pytest_gn.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r02f/pytest_gn.log
python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/r02f/bench_f32.json 2> gpurun_out/r02f/bench_f32.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f/prof_bf16_b16 -- python $R/bench.py --precision bf16 --batch-per-gpu 16 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/r02f/prof_bf16.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f/prof_fp8_b16 -- python $R/bench.py --precision fp8 --batch-per-gpu 16 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/r02f/prof_fp8.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02f/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/r02f/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02f/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/r02f/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/r02f/pmc_fetch gpurun_out/r02f/pmc_write 1 gpurun_out/r02f/pmc_summary.json
rm -rf gpurun_out/r02f/pmc_fetch/*/*kernel_trace* 2>/dev/null
find gpurun_out/r02f -name "*counter_collection.csv" -size +20M -delete
python -c "
import json
j=json.load(open('gpurun_out/r02f/bench_f32.json')); print(j['value'], j['kernel_classes_ms_per_image'])
"
du -sh gpurun_out/r02f
