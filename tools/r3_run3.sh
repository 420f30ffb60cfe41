#!/bin/bash
# round 3, run 3: the exchange inside the engine (RCCL communicator of the handle / hook), multi-rank lcx_iterate
mkdir -p gpurun_out/r3_run3
python -m pytest tests/test_distributed_gpu.py tests/test_syn_gpu.py tests/test_integration_stub_gpu.py -m gpu -x -q > gpurun_out/r3_run3/dist.log 2>&1
echo "dist rc=$?" >> gpurun_out/r3_run3/summary.txt
python -m pytest tests/test_bench_gpu.py -m gpu -x -q -k "not default_line" > gpurun_out/r3_run3/bench_tests.log 2>&1
echo "bench_tests rc=$?" >> gpurun_out/r3_run3/summary.txt
for mode in torch engine; do
  LCX_EXCHANGE=$mode python bench.py --workload c2 --force-exchange --no-extras --steps 30 --warmup 5 > gpurun_out/r3_run3/c2_force_exchange_$mode.json 2> gpurun_out/r3_run3/c2_force_exchange_$mode.err
  echo "c2 force-exchange $mode rc=$?" >> gpurun_out/r3_run3/summary.txt
done
python bench.py --workload c2 --no-extras --steps 30 --warmup 5 > gpurun_out/r3_run3/c2_plain.json 2> gpurun_out/r3_run3/c2_plain.err
for mode in torch engine; do
  LCX_EXCHANGE=$mode python bench.py --workload c4shard --force-exchange --no-extras --steps 10 --warmup 3 --repeats 1 > gpurun_out/r3_run3/c4_force_exchange_$mode.json 2> gpurun_out/r3_run3/c4_force_exchange_$mode.err
  echo "c4shard force-exchange $mode rc=$?" >> gpurun_out/r3_run3/summary.txt
done
tail -15 gpurun_out/r3_run3/dist.log
tail -5 gpurun_out/r3_run3/bench_tests.log
cat gpurun_out/r3_run3/summary.txt
