#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 600 ./tools/gemm_probe8 c3zero > gpurun_out/r02_gemm_probe8_zero.txt 2>&1; grep -E "^==|med" gpurun_out/r02_gemm_probe8_zero.txt | cut -c1-200
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02_bench_default.err | tail -1 > gpurun_out/r02_bench_default.json
echo "bench default: $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r02_bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_default.json')); c=d['config']['c2']
print('c3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['x_passes_per_iteration'], d['cpu_baseline']['value'])
print('c3 conv', d['config'].get('fit_to_convergence'))
print('c2', c['value'], c['ms_per_step'], c['roofline']['frac'], c['get_covariance'], d['config']['get_covariance_c5_standin'])
"
