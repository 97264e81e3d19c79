#!/bin/bash
# diagnostic library from a generator variant: build_variant.sh NAME "PNR_ASM_DIAG value" [extra hipcc flags]
set -e
cd "$(dirname "$0")/../.."
NAME=$1; DIAG=$2; shift 2 || true
D=/tmp/pnr_variant_$NAME
rm -rf $D && mkdir -p $D/pixel_nerf_multiscale_amd && cp -r pixel_nerf_multiscale_amd/csrc $D/pixel_nerf_multiscale_amd/ && cp -r include $D/
PNR_ASM_DIAG="$DIAG" PNR_ASM_OUT=$D/pixel_nerf_multiscale_amd/csrc/resblock_asm.inc python tools/gen_resblock_asm.py > /dev/null
C=$D/pixel_nerf_multiscale_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -shared -o tools/dev/libpnr_$NAME.so $C/pnr_api.hip $C/stage_kernels.hip $C/point_f32.hip $C/point_mfma.hip $C/train_f32.hip 2>&1 | grep -i " error" || true
ls -la tools/dev/libpnr_$NAME.so
