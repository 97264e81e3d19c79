#!/bin/bash
# round 4, step D: training path with the LDS-DMA GEMM
cd "$(dirname "$0")/../.."
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r4d_train_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/r4d_train_pytest.log
[ $rc -ne 0 ] && exit $rc
{ for p in bf16; do python tools/bench_train.py --precision $p --steps 10 2>/dev/null; done; } > gpurun_out/r4d_train_bench.txt
cat gpurun_out/r4d_train_bench.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4d_prof_train -- python tools/bench_train.py --precision bf16 --views 1 --steps 3 > gpurun_out/r4d_prof_train.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r4d_prof_train/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  {r["Percentage"]}%')
PY
