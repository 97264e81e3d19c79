// The statement-entry hazard of the fused kernel (DESIGN.md 4.1 "Statement-entry rule"), isolated.
//
// Finding (from the ISA of the failing round-2 build, not from re-running it): hipcc reloaded a spilled 16-byte value
// with `scratch_load_dwordx4 v[8:11]` right behind one asm statement; the only consumer sat on ONE of two paths, and the
// other path ran straight into the next asm statement, which writes v10 / v11 (its ring read bases) as CLOBBERED
// registers.  SIInsertWaitcnts orders a pending VMEM load against later EXPLICIT defs of its destination, but not
// against the implicit (clobber) defs of a memory-touching INLINEASM, so no s_waitcnt stood between the load and the
// statement — and the hardware has no interlock either: the load's data lands AFTER the statement's own write of the
// register and wins.
//
// The compiler half is in the ISA (profiles/r03_entry_hazard_isa_excerpt.txt); this file shows the hardware half on the
// GPU: one asm statement, `global_load_dword v10` of a cold line, then `v_mov_b32 v10, K` with and without an
// `s_waitcnt vmcnt(0)` in between; v10 is read back after a full drain.  Without the wait every lane ends up with the
// LOADED value: a pending VMEM load is not ordered against a later VALU write of its destination.
// Part C settles the second question the round-2 notes left open (point_mfma.hip, glds_stage): at which point an
// LDS-DMA instruction reads M0, i.e. how long M0 has to be held behind the last piece of a statement.
// Build: hipcc -O3 --offload-arch=gfx950 -o waw_ubench_test waw_ubench.hip ; run once, prints its findings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool WAIT>
__global__ void part_a(const unsigned* p, unsigned* out) {
    unsigned r;
    const unsigned* q = p + (size_t)threadIdx.x * 4096 + (size_t)blockIdx.x * 64 * 4096;    // one cold line per lane
    if (WAIT)
        asm volatile("global_load_dword v10, %1, off\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 v10, 0x1234\n\t"
                     "s_waitcnt vmcnt(0)\n\ts_nop 4\n\tv_mov_b32 %0, v10" : "=v"(r) : "v"(q) : "v10", "memory");
    else
        asm volatile("global_load_dword v10, %1, off\n\tv_mov_b32 v10, 0x1234\n\t"
                     "s_waitcnt vmcnt(0)\n\ts_nop 4\n\tv_mov_b32 %0, v10" : "=v"(r) : "v"(q) : "v10", "memory");
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

// ---------------------------------------------------------------------------------------------------------------
// part C: when does an LDS-DMA read M0?  `s_mov m0, A; s_nop 0; global_load_lds_dword; [K wait states]; s_mov m0, B`,
// cold source lines, optionally behind QUEUED older cold loads (a busy VMEM queue).  Reports where the bytes landed.
template <int K, int QUEUED>
__global__ void part_c(const unsigned* p, const unsigned* q, unsigned* out) {
    extern __shared__ unsigned lds[];                      // A = [0, 1024), B = [1024, 2048) dwords
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xffffffffu;
    __syncthreads();
    const unsigned a_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    const unsigned b_addr = a_addr + 4096;
    const unsigned* src = p + (size_t)threadIdx.x * 4096 + (size_t)blockIdx.x * 64 * 4096;   // one cold line per lane
    const unsigned* old = q + (size_t)threadIdx.x * 4096 + (size_t)blockIdx.x * 64 * 4096;
    unsigned keep;
#define NOPS(k) ((k) == 0 ? "" : (k) == 1 ? "s_nop 0\n\t" : (k) == 2 ? "s_nop 1\n\t" : (k) == 4 ? "s_nop 3\n\t" : "s_nop 7\n\t")
#define DMA_TEXT(older, nops)                                                                                     \
    "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\t" older "s_nop 0\n\tglobal_load_lds_dword %1, off\n\t" nops                   \
    "s_mov_b32 m0, %4\n\ts_waitcnt vmcnt(0)\n\ts_mov_b32 m0, %0"
#define OLDER "global_load_dword v20, %2, off\n\tglobal_load_dword v21, %2, off offset:256\n\tglobal_load_dword v22, %2, off offset:512\n\t" \
              "global_load_dword v23, %2, off offset:768\n\tglobal_load_dword v24, %2, off offset:1024\n\tglobal_load_dword v25, %2, off offset:1280\n\t" \
              "global_load_dword v26, %2, off offset:1536\n\tglobal_load_dword v27, %2, off offset:1792\n\t"
#define CASE(k, nops)                                                                                             \
    if (K == k) {                                                                                                 \
        if (QUEUED) asm volatile(DMA_TEXT(OLDER, nops) : "=&s"(keep) : "v"(src), "v"(old), "s"(a_addr), "s"(b_addr)  \
                                 : "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");              \
        else asm volatile(DMA_TEXT("", nops) : "=&s"(keep) : "v"(src), "v"(old), "s"(a_addr), "s"(b_addr) : "memory"); \
    }
    CASE(0, "") CASE(1, "s_nop 0\n\t") CASE(2, "s_nop 1\n\t") CASE(4, "s_nop 3\n\t") CASE(8, "s_nop 7\n\t")
    __syncthreads();
    out[(blockIdx.x * 64 + threadIdx.x) * 2 + 0] = lds[threadIdx.x];
    out[(blockIdx.x * 64 + threadIdx.x) * 2 + 1] = lds[1024 + threadIdx.x];
}

template <int K, int QUEUED>
static void run_c(const unsigned* p, const unsigned* q, unsigned* out, int blocks) {
    const int n = blocks * 64;
    hipLaunchKernelGGL((part_c<K, QUEUED>), dim3(blocks), dim3(64), 8192, 0, p, q, out);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(2 * n);
    (void)hipMemcpy(h.data(), out, 2 * n * 4, hipMemcpyDeviceToHost);
    int in_a = 0, in_b = 0, nowhere = 0;
    for (int i = 0; i < n; ++i) {
        const bool a = h[2 * i] == 0xABABABABu, b = h[2 * i + 1] == 0xABABABABu;
        in_a += a; in_b += b; nowhere += !a && !b;
    }
    printf("C  M0 rewritten %d wait state(s) behind the DMA, %d older loads queued: landed at the ORIGINAL M0 %5d, at the NEW M0 %5d, "
           "nowhere %d (of %d lanes)\n", K, QUEUED ? 8 : 0, in_a, in_b, nowhere, n);
}

int main() {
    const int blocks = 64, n = blocks * 64;
    unsigned *p, *out, *p2;
    (void)hipMalloc(&p, (size_t)n * 4096 * 4 + 4096);
    (void)hipMalloc(&out, n * 8);
    (void)hipMalloc(&p2, (size_t)n * 4096 * 4 + 4096);
    (void)hipMemset(p2, 0x11, (size_t)n * 4096 * 4);
    (void)hipMemset(p, 0xAB, (size_t)n * 4096 * 4);
    std::vector<unsigned> h(n);
    auto report = [&](const char* name, unsigned expect) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        int bad = 0, first = 0;
        for (int i = n - 1; i >= 0; --i)
            if (h[i] != expect) { ++bad; first = i; }
        printf("%-72s lanes that lost the statement's own value: %5d of %d (example 0x%08x, expected 0x%08x)\n", name, bad, n,
               h[first], expect);
    };
    hipLaunchKernelGGL(part_a<false>, dim3(blocks), dim3(64), 0, 0, p, out);
    report("A  load v10; v_mov v10 (no wait): the load lands later and wins", 0x1234);
    hipLaunchKernelGGL(part_a<true>, dim3(blocks), dim3(64), 0, 0, p, out);
    report("A' load v10; vmcnt(0); v_mov v10", 0x1234);
    // fresh cold lines for every variant: offset the base by one line each time
    run_c<0, 0>(p + 32, p2, out, blocks);
    run_c<1, 0>(p + 64, p2, out, blocks);
    run_c<2, 0>(p + 96, p2, out, blocks);
    run_c<4, 0>(p + 128, p2, out, blocks);
    run_c<8, 0>(p + 160, p2, out, blocks);
    run_c<0, 1>(p + 192, p2 + 32, out, blocks);
    run_c<1, 1>(p + 224, p2 + 64, out, blocks);
    run_c<4, 1>(p + 256, p2 + 96, out, blocks);
    run_c<8, 1>(p + 288, p2 + 128, out, blocks);
    return 0;
}
