// What does the shape of an LDS-DMA piece cost when a GEMM k-loop streams row-major operands?
// k_hgemm_dma (csrc/train_f32.hip) pulls its 128-row A and B tiles as 64-byte row chunks (a k-step of 32 bf16): half a cache
// line per row and request.  Its k-loop runs at 10-14 TB/s of L2 -> LDS traffic (profiles/r04_hgemm_epilogue.txt), under the
// 16.8-18.8 TB/s MI355X_MICROARCH.md gives for L2-served rows moved as whole lines.  This bench runs that loop WITHOUT the
// MFMAs for row chunks of 64 / 128 / 256 bytes and several ring depths: same tile order as the GEMM (the four column tiles
// of a 128-row block on one XCD), same bytes per workgroup (2 x 128 rows x 1 KiB), one barrier per k-step, 256 threads.
//   piece = one global_load_lds_dwordx4 wave-instruction = 1 KiB = (1024 / CHUNK) rows x CHUNK bytes
// Build: hipcc -O3 --offload-arch=gfx950 -o dma_piece_ubench_test dma_piece_ubench.hip ; run once on an idle MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// CHUNK: bytes of a row per k-step; DEPTH: ring slots (DEPTH - 1 k-steps in flight); BUSY: s_sleep units per k-step (stands
// in for the MFMAs; 0 = none)
template <int CHUNK, int DEPTH, int BUSY>
__global__ void __launch_bounds__(256) k_stream(const char* __restrict__ A, const char* __restrict__ B, int M, int N, int row_bytes,
                                                unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(1024))) char ring[];
    constexpr int ROWS_PER_PIECE = 1024 / CHUNK;              // rows one wave-instruction covers
    constexpr int PIECES = 128 / ROWS_PER_PIECE;              // pieces per operand and k-step
    constexpr int PER_WAVE = 2 * PIECES / 4;                  // pieces a wave issues per k-step (both operands over 4 waves)
    constexpr int SLOT = 2 * 128 * CHUNK;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int gx = (M + 127) / 128, gy = N / 128;
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int bx = (slot_id / gy) * 8 + xcd, by = slot_id % gy;
    if (bx >= gx) return;
    // wave w issues pieces [w * PER_WAVE, (w + 1) * PER_WAVE) of the 2 * PIECES (A first, then B)
    uint32_t voff[PER_WAVE];
    const char* base[PER_WAVE];
    uint32_t dst[PER_WAVE];
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)ring;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int p = wv * PER_WAVE + j;
        const bool isb = p >= PIECES;
        const int pp = isb ? p - PIECES : p;
        const int row = pp * ROWS_PER_PIECE + lane / (CHUNK / 16);
        const int c = lane % (CHUNK / 16);
        const int x0 = isb ? by * 128 : bx * 128, X = isb ? N : M;
        const int xr = x0 + row < X ? row : X - 1 - x0;
        voff[j] = (uint32_t)xr * (uint32_t)row_bytes + 16u * c;
        base[j] = (isb ? B : A) + (size_t)x0 * row_bytes;
        dst[j] = ring_lds + (isb ? 128 * CHUNK : 0) + pp * 1024;
    }
    const int nk = row_bytes / CHUNK;
    auto issue = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const char* sb = base[j] + (size_t)ks * CHUNK;
            const uint32_t d = __builtin_amdgcn_readfirstlane(dst[j] + (uint32_t)(ks % DEPTH) * SLOT);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff[j]), "s"(sb), "s"(d) : "memory");
        }
    };
    for (int ks = 0; ks < DEPTH - 1 && ks < nk; ++ks) issue(ks);
    unsigned acc = 0;
    for (int ks = 0; ks < nk; ++ks) {
        // k-step ks has landed when at most the pieces of the younger k-steps in flight are outstanding
        const int younger = nk - 1 - ks < DEPTH - 2 ? nk - 1 - ks : DEPTH - 2;
        if (younger * PER_WAVE >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (younger * PER_WAVE == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger * PER_WAVE == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger * PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger * PER_WAVE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (younger * PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (younger * PER_WAVE == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ks + DEPTH - 1 < nk) issue(ks + DEPTH - 1);
        acc += *(const unsigned*)(ring + (ks % DEPTH) * SLOT + t * 16);          // one read per k-step: the data is used
        if (BUSY) __builtin_amdgcn_s_sleep(BUSY);
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int CHUNK, int DEPTH, int BUSY>
static int run(const char* A, const char* B, int M, int N, int row_bytes, unsigned* sink, const char* what) {
    constexpr int lds = DEPTH * 2 * 128 * CHUNK;
    CK(hipFuncSetAttribute((const void*)k_stream<CHUNK, DEPTH, BUSY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int gx = (M + 127) / 128, gy = N / 128;
    const unsigned blocks = (unsigned)((gx + 7) / 8 * 8 * gy);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<CHUNK, DEPTH, BUSY>), dim3(blocks), dim3(256), lds, 0, A, B, M, N, row_bytes, sink);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<CHUNK, DEPTH, BUSY>), dim3(blocks), dim3(256), lds, 0, A, B, M, N, row_bytes, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    const double bytes = (double)gx * gy * 2 * 128 * row_bytes;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_stream<CHUNK, DEPTH, BUSY>, 256, lds));
    printf("%-44s chunk %3d B  ring %d x %2d KiB  %d wg/CU  in flight/CU %3d KiB   %7.1f us   %5.2f TB/s L2->LDS\n", what, CHUNK, DEPTH,
           2 * 128 * CHUNK / 1024, occ, occ * (DEPTH - 1) * 2 * 128 * CHUNK / 1024, us, bytes / us * 1e-6);
    return 0;
}

int main() {
    const int N = 512, row_bytes = 1024;
    unsigned* sink;
    CK(hipMalloc(&sink, 1 << 20));
    for (int M : {32768, 49152}) {
        char *A, *B;
        CK(hipMalloc(&A, (size_t)M * row_bytes)); CK(hipMalloc(&B, (size_t)N * row_bytes));
        CK(hipMemset(A, 1, (size_t)M * row_bytes)); CK(hipMemset(B, 1, (size_t)N * row_bytes));
        printf("== A %d x 512 bf16 (%d MB), B 512 x 512; every workgroup pulls 2 x 128 rows x 1 KiB = 256 KiB; %d workgroups\n", M,
               (int)((size_t)M * row_bytes >> 20), (M / 128) * 4);
        if (run<64, 4, 0>(A, B, M, N, row_bytes, sink, "k_hgemm_dma's loop (64-B chunks, 3 in flight)")) return 1;
        if (run<64, 5, 0>(A, B, M, N, row_bytes, sink, "64-B chunks, 4 in flight")) return 1;
        if (run<128, 2, 0>(A, B, M, N, row_bytes, sink, "128-B chunks (whole lines), 1 in flight")) return 1;
        if (run<128, 3, 0>(A, B, M, N, row_bytes, sink, "128-B chunks, 2 in flight (1 wg/CU)")) return 1;
        if (run<128, 4, 0>(A, B, M, N, row_bytes, sink, "128-B chunks, 3 in flight (1 wg/CU)")) return 1;
        if (run<256, 2, 0>(A, B, M, N, row_bytes, sink, "256-B chunks, 1 in flight (1 wg/CU)")) return 1;
        if (run<64, 4, 8>(A, B, M, N, row_bytes, sink, "64-B chunks, 3 in flight, 512 busy cycles/step")) return 1;
        if (run<128, 2, 16>(A, B, M, N, row_bytes, sink, "128-B chunks, 1 in flight, 1024 busy cycles/step")) return 1;
        CK(hipFree(A)); CK(hipFree(B));
    }
    return 0;
}
