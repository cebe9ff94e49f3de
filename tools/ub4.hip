// Select idioms (diagnostic only; same harness as ub3.hip): what the issue ports of
// MI355X sustain for the instruction kinds these kernels are made of.  One workgroup per CU (LDS-sized so), 4 k waves in it
// (k per SIMD), every wave runs the same loop of 8 independent chains x 32 instructions; rate = instructions of all waves /
// a wave's mean s_memtime span (all waves of a CU run the same loop side by side).
//   hipcc -O3 --offload-arch=gfx950 -o ub3 tools/ub3.hip && ./ub3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define TT(t) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define R8(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7)
#define B4(b) b b b b
#define ITER 256
// each body: 8 independent chains, one instruction each; B4(B4(..)) of it... 32 per asm, x ITER
#define VADD(i)  "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define P0(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t" "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t" "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n\t"
#define P1(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t" "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t" "s_nop 0\n\t" "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n\t"
#define P2(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t" "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n\t" "v_cndmask_b32_e64 %" #i ", %8, %" #i ", vcc\n\t"
#define P3(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t" "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t" "v_cndmask_b32_e64 %" #i ", %8, %" #i ", vcc\n\t"
#define P4(i) "v_cmp_lt_u32_e64 s[22:23], %" #i ", %8\n\t" "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[22:23]\n\t" "v_cndmask_b32_e64 %" #i ", %8, %" #i ", s[22:23]\n\t"
#define P5(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t" "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t" "v_mov_b32 v40, %8\n\t" "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n\t"
#define P6(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t" "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t" "s_mov_b32 s24, 0\n\t" "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n\t"
#define P7(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t" "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t" "s_nop 1\n\t" "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n\t"
#define P8(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t" "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define P9(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"

template <int KIND>
__device__ void body(uint32_t (&v)[8], uint32_t w, uint32_t (&s)[8], uint32_t sb)
{
    for (int it = 0; it < ITER; it++) {
#define VB(X) asm volatile(B4(R8(X)) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(w) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "memory")
#define SB(X) asm volatile(B4(R8(X)) : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7]) : "s"(sb) : "scc")
        if (KIND == 0) VB(P0);
        if (KIND == 1) VB(P1);
        if (KIND == 2) VB(P2);
        if (KIND == 3) VB(P3);
        if (KIND == 4) VB(P4);
        if (KIND == 5) VB(P5);
        if (KIND == 6) VB(P6);
        if (KIND == 7) VB(P7);
        if (KIND == 8) VB(P8);
        if (KIND == 9) VB(P9);
    }
}
#define NKIND 10
extern __shared__ uint32_t lds[];
__global__ void __launch_bounds__(1024) k(uint64_t *out, uint32_t seed)
{
    uint32_t v[8], s[8];
    for (int i = 0; i < 8; i++) { v[i] = threadIdx.x * (i + 3) + seed; s[i] = __builtin_amdgcn_readfirstlane(seed * (i + 7)); }
    const uint32_t w = threadIdx.x * 5 + 1, sb = __builtin_amdgcn_readfirstlane(seed | 1);
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint64_t t0, t1;
#define RUN(K) __syncthreads(); TT(t0); body<K>(v, w, s, sb); TT(t1); __syncthreads(); if ((threadIdx.x & 63) == 0) { out[(size_t)wave * 2 * NKIND + 2 * K] = t0; out[(size_t)wave * 2 * NKIND + 2 * K + 1] = t1; }
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_mov_b64 s[20:21], vcc" :: "v"(v[0]), "v"(w) : "vcc", "s20", "s21");
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc += v[i] + s[i];
    if (acc == 0x12345u) out[0] = acc;
}
int main()
{
    const char *names[NKIND] = {"cmp; cnd; cnd", "cmp; cnd; s_nop 0; cnd", "cmp; cnd_e64 vcc; cnd_e64 vcc", "cmp; cnd; cnd_e64 vcc", "cmp_e64 s22; cnd_e64 s22; cnd_e64 s22", "cmp; cnd; v_mov v40; cnd", "cmp; cnd; s_mov s24; cnd", "cmp; cnd; s_nop 1; cnd", "cmp; cnd (other chain's reg between)", "v_add_u32"};
    const int ncu = 256;
    uint64_t *d; size_t bytes = (size_t)ncu * 32 * 2 * NKIND * 8; (void)hipMalloc(&d, bytes);
    uint64_t *h = (uint64_t *)malloc(bytes);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("instructions per cycle, whole wavefront (64 lanes) instructions; per SIMD for vector kinds, per CU for scalar kinds\n");
    printf("%-24s", "waves per SIMD:");
    for (int kw = 1; kw <= 4; kw *= 2) printf("  %8d", kw);
    printf("\n");
    double res[NKIND][4];
    int col = 0;
    for (int kw = 1; kw <= 4; kw *= 2, col++) {
        // kw <= 4: one workgroup of 4 kw waves per CU (100 KB of LDS: a second does not fit); kw = 8: two of 16 waves (64 KB each)
        const int waves = kw == 8 ? 16 : 4 * kw, nblk = kw == 8 ? 2 * ncu : ncu, ldsb = kw == 8 ? 64 * 1024 : 100 * 1024;
        for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(nblk), dim3(64 * waves), ldsb, 0, d, 12345u + r); if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; } }
        (void)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
        for (int K = 0; K < NKIND; K++) {
            double span = 0;                                   // every wave runs the same work beside the same neighbours: mean span
            for (int wv = 0; wv < nblk * waves; wv++) span += (double)(h[(size_t)wv * 2 * NKIND + 2 * K + 1] - h[(size_t)wv * 2 * NKIND + 2 * K]);
            span /= (double)nblk * waves;
            static const double ninst[NKIND] = {3, 4, 3, 3, 3, 4, 4, 4, 2, 1}; const double per_wave = ninst[K] * 32.0 * ITER;
            res[K][col] = per_wave * 4 * kw / span;            // per CU
        }
    }
    for (int K = 0; K < NKIND; K++) {
        printf("%-36s", names[K]);
        for (int c = 0; c < 3; c++) printf("  %8.3f", res[K][c] / 4.0);
        printf("   (per SIMD)\n");
    }
    return 0;
}
