#!/bin/bash
# Diagnostic libraries from generator variants, several at once: build_variants.sh NAME[=DIAG words][+stamps] ...
#   e.g. build_variants.sh base base+stamps "nt2=nt2" "nodma=nodma nowait"
# Only point_mfma.hip is recompiled (the other objects come from pixel_nerf_multiscale_amd/lib/, built by build_native);
# output tools/dev/libpnr_NAME.so (git-ignored; travels to the GPU box with gpurun).
set -e
cd "$(dirname "$0")/../.."
python -m pixel_nerf_multiscale_amd.build_native > /dev/null
one() {
    spec="$1"; stamps=""
    case "$spec" in *+stamps) stamps="-DPNR_STAMPS"; spec="${spec%+stamps}";; esac
    name="${spec%%=*}"; diag=""
    case "$spec" in *=*) diag="${spec#*=}";; esac
    [ -n "$stamps" ] && name="${name}_st"
    D=/tmp/pnr_variant_$name
    rm -rf $D && mkdir -p $D/pixel_nerf_multiscale_amd && cp -r pixel_nerf_multiscale_amd/csrc $D/pixel_nerf_multiscale_amd/ && cp -r include $D/
    PNR_ASM_DIAG="$diag" PNR_ASM_OUT=$D/pixel_nerf_multiscale_amd/csrc/resblock_asm.inc python tools/gen_resblock_asm.py > /dev/null
    C=$D/pixel_nerf_multiscale_amd/csrc
    L=pixel_nerf_multiscale_amd/lib
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $stamps -c $C/point_mfma.hip -o $D/point_mfma.o 2>&1 | grep -i " error" || true
    if [ -n "$stamps" ]; then   # pnr_debug_stamps lives in point_mfma.hip; the other objects are stamp-agnostic
        :
    fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/libpnr_$name.so $L/pnr_api.o $L/stage_kernels.o $L/point_f32.o $D/point_mfma.o $L/train_f32.o
    echo "built tools/dev/libpnr_$name.so  (diag: '$diag' $stamps)"
}
export -f one
printf '%s\n' "$@" | xargs -P 4 -I{} bash -c 'one "{}"'
