#!/bin/bash
# kernel stats of a training step: prof_train.sh PRECISION
cd "$(dirname "$0")/../.."
P=${1:-fp32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_$P -- python tools/bench_train.py --precision $P --views 1 --steps 3 > gpurun_out/prof_train_$P.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof_train_$P/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:16]:
        print(f'{r["Name"][:86]:86s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
    print("total", tot / 1e6, "ms")
PY
