#!/bin/bash
# diagnostic library with in-kernel section stamps (never shipped): tools/dev/libpnr_stamps.so
set -e
cd "$(dirname "$0")/../.."
C=pixel_nerf_multiscale_amd/csrc
OUT=${1:-tools/dev/libpnr_stamps.so}; shift || true
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 ${STAMPS--DPNR_STAMPS} "$@" -shared -o $OUT $C/pnr_api.hip $C/stage_kernels.hip $C/point_f32.hip $C/point_mfma.hip $C/train_f32.hip
