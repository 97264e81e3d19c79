#!/bin/bash
# Diagnostic library with extra hipcc flags on point_mfma.hip and an optional generator variant:
#   build_flagged.sh NAME "PNR_ASM_DIAG words" [hipcc flags...]   ->  tools/dev/libpnr_NAME.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; DIAG=$2; shift 2 || true
D=/tmp/pnr_variant_$NAME
rm -rf $D && mkdir -p $D/pixel_nerf_multiscale_amd && cp -r pixel_nerf_multiscale_amd/csrc $D/pixel_nerf_multiscale_amd/ && cp -r include $D/
PNR_ASM_DIAG="$DIAG" PNR_ASM_OUT=$D/pixel_nerf_multiscale_amd/csrc/resblock_asm.inc python tools/gen_resblock_asm.py > /dev/null
C=$D/pixel_nerf_multiscale_amd/csrc
L=pixel_nerf_multiscale_amd/lib
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -c $C/point_mfma.hip -o $D/point_mfma.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/libpnr_$NAME.so $L/pnr_api.o $L/stage_kernels.o $L/point_f32.o $D/point_mfma.o $L/train_f32.o
echo "built tools/dev/libpnr_$NAME.so (diag '$DIAG' flags '$*')"
