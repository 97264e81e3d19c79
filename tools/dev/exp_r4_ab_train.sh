#!/bin/bash
# A/B of two libraries on ONE box: tools/bench_train.py bf16, interleaved three times.  exp_r4_ab_train.sh LIB_A.so LIB_B.so
cd "$(dirname "$0")/../.."
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset PNR_LIB; else export PNR_LIB=$PWD/$v; fi
    echo "== $v round $r"
    python tools/bench_train.py --precision bf16 --steps 20 2>/dev/null | python -c "import sys,json; print(' / '.join(str(json.loads(l)['ms_per_step']) for l in sys.stdin))"
  done
done
