// Upper-bound study for the fused kernel's chunk loop: hand-scheduled asm, 4 waves/CU, persistent over all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include CHUNK_INC
__global__ void __launch_bounds__(256, 1) k(const char* stream, int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((unsigned*)smem)[i] = 0x3c003c00u;
    __syncthreads();
    unsigned lds_rd = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + lane * 16;
    unsigned lds_dma = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wv * 4096);
    unsigned goff = wv * 4096 + lane * 16;
    unsigned bias = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (lane >> 5) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        "v_mov_b32 v2, %0\n\tv_mov_b32 v3, %1\n\tv_mov_b32 v4, %2\n\t"
        "s_mov_b32 s20, %3\n\ts_mov_b64 s[22:23], %4\n\ts_mov_b32 s24, %5\n\t"
        "1:\n\t"
        CHUNK_ASM
        "s_sub_u32 s24, s24, 1\n\t"
        "s_cmp_lg_u32 s24, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\t"
        :: "v"(lds_rd), "v"(goff), "v"(bias), "s"(lds_dma), "s"(stream), "s"(iters)
        : "memory", "s20", "s22", "s23", "s24", "v2", "v3", "v4",
          "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55",
          "v60","v61","v62","v63","v64","v65","v66","v67",
          "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115",
          "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131",
          "a0","a15","a16","a31","a32","a47","a48","a63","a64","a79","a80","a95","a96","a111","a112","a127","a128","a143","a144","a159",
          "a160","a175","a176","a191","a192","a207","a208","a223","a224","a239","a240","a255");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (sink && lane == 999) sink[0] = 1.f;
}
int main() {
    const int iters = 400;
    char* stream; hipMalloc(&stream, 8 << 20); hipMemset(stream, 0, 8 << 20);
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536 + 1024, 0, stream, iters, cyc, nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= 256;
        printf("%s: %.3f ms, %.0f cycles/chunk (ideal 2048), %.1f us/chunk, clock %.2f GHz, MFMA TF %.0f\n", VARIANT, ms, mean / iters,
               ms * 1e3 / iters, mean / (ms * 1e-3) / 1e9, 256.0 * 4 * iters * 64 * 32768 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
