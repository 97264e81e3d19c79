#!/bin/bash
# short form of final_r4_c.sh: headline PMC -> bench line -> rocprof stats of the default command
cd "$(dirname "$0")/../.."
export BUILD_ID="r4-$(python -c 'import bench; print(bench.source_hash())')"
bash tools/pmc_passes.sh r04 > gpurun_out/r04_pmc_k_point_mfma.txt 2> gpurun_out/r04_pmc.err
cp gpurun_out/r04_pmc_k_point_mfma.txt profiles/latest_pmc_bench_default.txt
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('pmc_pass'))
for s in d['secondary']: print(s['workload'], s['dtype'], round(s['value']), round(s['roofline_frac'],4), round(s['ms_per_step'],2))"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_default -- python bench.py --cpu-rays 0 --secondary-steps 0 > gpurun_out/r04_bench_line_under_rocprof.json 2> gpurun_out/r04_prof_default.err
echo done
