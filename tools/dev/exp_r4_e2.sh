#!/bin/bash
# round 4: training tests + bench of all three precisions + rocprof stats of the bf16x3 step
cd "$(dirname "$0")/../.."
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r4e2_train_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/r4e2_train_pytest.log
[ $rc -ne 0 ] && exit $rc
{ for p in bf16 fp32 bf16x3; do python tools/bench_train.py --precision $p --steps 10 2>/dev/null; done; } > gpurun_out/r4e2_train_bench.txt
cat gpurun_out/r4e2_train_bench.txt
