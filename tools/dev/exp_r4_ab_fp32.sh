#!/bin/bash
# A/B of two libraries on ONE box, fp32 paths: the fp32 inference frame (bench.py --precision fp32) and the fp32 training step, interleaved
cd "$(dirname "$0")/../.."
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset PNR_LIB; else export PNR_LIB=$PWD/$v; fi
    echo "== $v round $r"
    python bench.py --precision fp32 --cpu-rays 0 --secondary-steps 0 --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 frame ms', round(d['ms_per_step'],2))"
    python tools/bench_train.py --precision fp32 --steps 10 2>/dev/null | python -c "import sys,json; print('fp32 train ms', ' / '.join(str(json.loads(l)['ms_per_step']) for l in sys.stdin))"
  done
done
