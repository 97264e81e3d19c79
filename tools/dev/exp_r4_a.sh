#!/bin/bash
# round 4, experiment A: fp16 power levers (FP16_OVFL, no saturation, weights with fewer mantissa bits)
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out/r4_expA.txt
{
echo "== FP16_OVFL micro-test"; tools/dev/ubench/ovfl_ubench_test
echo "== fp16 A/B headline"; PNR_AB_PRECISION=fp16 python tools/dev/ab_bench.py tools/dev/libpnr_base.so tools/dev/libpnr_nosat.so tools/dev/libpnr_wmask3.so tools/dev/libpnr_wmask2.so
echo "== bf16 base for reference"; PNR_AB_PRECISION=bf16 python tools/dev/ab_bench.py tools/dev/libpnr_base.so
for v in base wmask3 wmask2; do echo "== psnr $v"; PNR_LIB=$PWD/tools/dev/libpnr_$v.so python tools/full_frame_psnr.py dtu_3view_400x300_k128 srn_chairs_1view_128x128_k128 2>/dev/null; done
} > $O 2>&1
tail -40 $O
