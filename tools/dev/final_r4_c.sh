#!/bin/bash
# round 4, last pass on the final sources: headline PMC -> bench line (carries the PMC's traffic) -> 2-rank rehearsal -> training PMC
# (memory side) -> training bench -> rocprof kernel stats of the default bench command and of the bf16 training step
cd "$(dirname "$0")/../.."
export BUILD_ID="r4-$(python -c 'import bench; print(bench.source_hash())')"
bash tools/pmc_passes.sh r04 > gpurun_out/r04_pmc_k_point_mfma.txt 2> gpurun_out/r04_pmc.err
cp gpurun_out/r04_pmc_k_point_mfma.txt profiles/latest_pmc_bench_default.txt
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('pmc_pass'))"
PNR_BENCH_ONE_CARD=1 python bench.py --gpus 2 --steps 5 --warmup 1 --strong-steps 2 > gpurun_out/r04_bench_rehearsal_2ranks_one_card.json 2> gpurun_out/r04_rehearsal.err
echo "rehearsal rc $?"
bash tools/dev/pmc_train_mem.sh bf16 > gpurun_out/r04_pmc_train_mem.txt 2> gpurun_out/r04_pmc_train.err
cat gpurun_out/r04_pmc_train_mem.txt
{ for p in bf16 fp32 bf16x3; do python tools/bench_train.py --precision $p --steps 10 2>/dev/null; done; } > gpurun_out/r04_train_bench_final.txt
cat gpurun_out/r04_train_bench_final.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_default -- python bench.py --cpu-rays 0 --secondary-steps 0 > gpurun_out/r04_bench_line_under_rocprof.json 2> gpurun_out/r04_prof_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_train -- python tools/bench_train.py --precision bf16 --views 1 --steps 3 > gpurun_out/r04_prof_train.log 2>&1
echo done
