#!/bin/bash
# GPU box: wall-time A/B (interleaved rounds, one device) + in-kernel section cycles / clock for a list of variant names
# built by build_variants.sh (NAME -> tools/dev/libpnr_NAME.so, tools/dev/libpnr_NAME_st.so).
#   run_experiments.sh OUT [--workload W] NAME...
set -e
cd "$(dirname "$0")/../.."
OUT=$1; shift
WL=""
if [ "$1" == "--workload" ]; then WL="$2"; shift 2; fi
LIBS=""
for n in "$@"; do LIBS="$LIBS tools/dev/libpnr_$n.so"; done
{
echo "== wall (bench.py kernel_ms, 3 interleaved rounds) ${WL:+workload $WL}"
python tools/dev/ab_bench.py ${WL:+--workload $WL} $LIBS
for n in "$@"; do
    if [ -f tools/dev/libpnr_${n}_st.so ]; then
        echo "== stamps $n"
        PNR_LIB=$PWD/tools/dev/libpnr_${n}_st.so python tools/dev/run_stamps.py $WL 2>&1 | grep -v "amdgpu.ids"
    fi
done
} 2>&1 | tee $OUT
