#!/bin/bash
# PMC passes of the headline bench (GPU box).  Separate runs per counter group (SQ 8 slots, TCC 4 slots; FETCH_SIZE=3, WRITE_SIZE=2).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_${1:-r01}
ARGS="--steps 3 --warmup 1 --cpu-rays 0 --secondary-steps 0 ${@:2}"
echo "# build: ${BUILD_ID:-unknown build}, $(date -u +%Y-%m-%dT%H:%MZ), python bench.py $ARGS"
echo "# sources: $(python -c 'import bench; print(bench.source_hash())')"
echo "# dtype: $(python -c 'import bench, sys; a = sys.argv[1:]; print(a[a.index("--precision") + 1] if "--precision" in a else bench.HEADLINE_DTYPE)' ${@:2})"
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python bench.py $ARGS > $OUT.p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES -- python bench.py $ARGS > $OUT.p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python bench.py $ARGS > $OUT.p3.log 2>&1
python - <<PY
import csv, glob, collections
for p in ("p1","p2","p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_point_mfma" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(p, k, "n=%d mean=%.6g" % (len(v), sum(v)/len(v)))
# duration of the same dispatches in the pass that holds GRBM_GUI_ACTIVE: effective clock = GRBM_GUI_ACTIVE / 8 / duration
# (MI355X_MICROARCH.md 'DVFS give-back'; the counter is summed over the 8 XCDs)
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_point_mfma" in r["Kernel_Name"]]
    if d:
        print("p1 KERNEL_NS n=%d mean=%.6g" % (len(d), sum(d)/len(d)))
PY
