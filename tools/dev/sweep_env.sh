#!/bin/bash
# GPU box: bench.py kernel_ms / rays per second for a list of workloads under values of one environment variable
# (interleaved rounds, one device).   sweep_env.sh VAR "v1 v2 .." "workload1 workload2 .." [precision] [rounds]
cd "$(dirname "$0")/../.."
VAR=$1; VALS=$2; WLS=$3; PREC=${4:-fp16}; ROUNDS=${5:-2}
for wl in $WLS; do
  for r in $(seq $ROUNDS); do
    for v in $VALS; do
      out=$(env $VAR=$v python bench.py --workload $wl --precision $PREC --steps 5 --warmup 2 --cpu-rays 0 --secondary-steps 0 2>/dev/null | grep '^{' | tail -1)
      python - "$wl" "$VAR=$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print(f"{sys.argv[1]:44s} {sys.argv[2]:28s} kernel_ms {d['roofline']['kernel_ms']:9.3f}  ms_per_step {d['ms_per_step']:9.3f}  frac {d['roofline']['frac']:.4f}", flush=True)
PY
    done
  done
done
