#!/bin/bash
# last measurements on the final sources: headline PMC + bench line (traffic needs the PMC of the same sources), training PMC (memory side)
cd "$(dirname "$0")/../.."
export BUILD_ID="r4-$(python -c 'import bench; print(bench.source_hash())')"
bash tools/pmc_passes.sh r04 > gpurun_out/r04_pmc_k_point_mfma.txt 2> gpurun_out/r04_pmc.err
cp gpurun_out/r04_pmc_k_point_mfma.txt profiles/latest_pmc_bench_default.txt
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('pmc_pass'))"
bash tools/dev/pmc_train_mem.sh bf16 > gpurun_out/r04_pmc_train_mem.txt 2> gpurun_out/r04_pmc_train.err
cat gpurun_out/r04_pmc_train_mem.txt
{ for p in bf16 fp32 bf16x3; do python tools/bench_train.py --precision $p --steps 10 2>/dev/null; done; } > gpurun_out/r04_train_bench_final.txt
cat gpurun_out/r04_train_bench_final.txt
