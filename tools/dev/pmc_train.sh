#!/bin/bash
# PMC passes of the training-step bench (GPU box): per-kernel means for the GEMM kernels.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_train
ARGS="--views 1 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python3 tools/bench_train.py $ARGS > $OUT.p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD -- python3 tools/bench_train.py $ARGS > $OUT.p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "k_mgemm" in n:
                key = n[n.index("<"):n.index(">")+1]
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for key in sorted(acc):
            for k, v in sorted(acc[key].items()):
                print(p, key, k, "n=%d mean=%.6g" % (len(v), sum(v)/len(v)))
PY
