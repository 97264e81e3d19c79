#!/bin/bash
# Diagnostic library with extra hipcc flags on train_f32.hip:  build_train_variant.sh NAME [hipcc flags...]  ->  tools/dev/libpnr_NAME.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
L=pixel_nerf_multiscale_amd/lib
C=pixel_nerf_multiscale_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude "$@" -c $C/train_f32.hip -o /tmp/train_f32_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/libpnr_$NAME.so $L/pnr_api.o $L/stage_kernels.o $L/point_f32.o $L/point_mfma.o /tmp/train_f32_$NAME.o
echo "built tools/dev/libpnr_$NAME.so ($*)"
