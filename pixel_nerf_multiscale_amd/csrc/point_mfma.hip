// placeholder until the fused MFMA kernel lands
#include "pnr_common.h"
namespace pnr {
uint64_t point_mfma_workspace_bytes(const pnr_mlp*, const pnr_views*) { return 256; }
int32_t point_mfma(const pnr_params*, const pnr_mlp*, const pnr_views*, PointSrc, int64_t, int64_t, float*, void*,
                   uint64_t, hipStream_t) { return PNR_E_UNSUPPORTED; }
}
extern "C" uint64_t pnr_packed_mlp_bytes(const pnr_mlp*) { return 0; }
extern "C" int32_t pnr_pack_mlp(const pnr_mlp*, int32_t, void*, uint64_t, void*) { return PNR_E_UNSUPPORTED; }
extern "C" uint64_t pnr_packed_latent_bytes(const pnr_views*) { return 0; }
extern "C" int32_t pnr_pack_latents(const pnr_views*, int32_t, void*, uint64_t, void*) { return PNR_E_UNSUPPORTED; }
