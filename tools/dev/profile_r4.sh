#!/bin/bash
# round 4 profiles (GPU box): PMC passes of the default bench (fp16 headline), rocprofv3 kernel stats per workload, stamps
cd "$(dirname "$0")/../.."
export BUILD_ID="r4-$(python -c 'import bench; print(bench.source_hash())')"
bash tools/pmc_passes.sh r04 > gpurun_out/r04_pmc_k_point_mfma.txt 2> gpurun_out/r04_pmc.err
tail -25 gpurun_out/r04_pmc_k_point_mfma.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_default -- python bench.py --cpu-rays 0 --secondary-steps 0 > gpurun_out/r04_bench_line_under_rocprof.json 2> gpurun_out/r04_prof_default.err
for wl in srn_chairs_1view_128x128_k64+32 nmr_3view_64x64_k64+32 dtu_3view_400x300_k128 multiscale_cars_2view_128x128_k64+32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_$wl -- python bench.py --workload $wl --steps 5 --warmup 2 --cpu-rays 0 --secondary-steps 0 > gpurun_out/r04_prof_$wl.json 2> gpurun_out/r04_prof_$wl.err
done
for wl in dtu_3view_400x300_k128 nmr_3view_64x64_k64+32 multiscale_cars_2view_128x128_k64+32; do
  echo "== $wl"; PNR_LIB=$PWD/tools/dev/libpnr_stamps.so python tools/dev/run_stamps.py $wl 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04_stamps_multiview.txt
tail -50 gpurun_out/r04_stamps_multiview.txt
