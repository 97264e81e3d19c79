#!/bin/bash
# round 4, experiment B: LDS compositing A/B on the headline; intra-XCD stagger on the multi-view shapes
cd "$(dirname "$0")/../.."
O=gpurun_out/r4_expB.txt
{
echo "== headline fp16: LDS compositing (base) vs memory route (nolds)"; PNR_AB_PRECISION=fp16 python tools/dev/ab_bench.py tools/dev/libpnr_base.so tools/dev/libpnr_nolds.so
echo "== multiscale fp16: image prefetch (base) vs restore (nopf)"; PNR_AB_PRECISION=fp16 python tools/dev/ab_bench.py --workload multiscale_cars_2view_128x128_k64+32 tools/dev/libpnr_base.so tools/dev/libpnr_nopf.so
for wl in dtu_3view_400x300_k128 nmr_3view_64x64_k64+32 multiscale_cars_2view_128x128_k64+32; do
echo "== $wl fp16: intra-XCD stagger"; PNR_AB_PRECISION=fp16 python tools/dev/ab_bench.py --workload $wl tools/dev/libpnr_base.so tools/dev/libpnr_stg256.so tools/dev/libpnr_stg512.so tools/dev/libpnr_stg1024.so
done
} > $O 2>&1
tail -30 $O
