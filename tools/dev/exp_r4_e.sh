#!/bin/bash
# round 4, step E: fp32 LDS-DMA GEMM — whole GPU suite, fp32 path timing, fp32 training step
cd "$(dirname "$0")/../.."
tools/dev/gpu_suite.sh r4e || exit 1
python bench.py --precision fp32 --steps 5 --warmup 2 --cpu-rays 0 --secondary-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 path', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
python tools/bench_train.py --precision fp32 --steps 10 2>/dev/null
python tools/bench_train.py --precision bf16x3 --steps 10 2>/dev/null
