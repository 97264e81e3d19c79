#!/bin/bash
# round 4: where does k_hgemm_dma's time go?  timing-only variants (tools/dev/build_train_variant.sh hgN -DPNR_HG_DIAG=N), results garbage
cd "$(dirname "$0")/../.."
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in base "$@"; do
  if [ $v = base ]; then unset PNR_LIB; else export PNR_LIB=$PWD/tools/dev/libpnr_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hg_$v -- python tools/bench_train.py --precision bf16 --views 1 --steps 3 > gpurun_out/hg_$v.log 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/hg_$v.log; exit 1; }
  echo "== $v"
  python - $v <<'PY'
import csv, glob, sys
for f in glob.glob(f"gpurun_out/hg_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:8]:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
done
