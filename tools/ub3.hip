// Aggregate issue rate of one SIMD / one CU with 1..8 wavefronts per SIMD (diagnostic only): what the issue ports of
// MI355X sustain for the instruction kinds these kernels are made of.  One workgroup per CU (LDS-sized so), 4 k waves in it
// (k per SIMD), every wave runs the same loop of 8 independent chains x 32 instructions; rate = instructions of all waves /
// a wave's mean s_memtime span (all waves of a CU run the same loop side by side), and per cycle of a 2.4 GHz clock from the
// launch's hipEvent time (one launch per instruction kind and residency).
//   hipcc -O3 --offload-arch=gfx950 -o ub3 tools/ub3.hip && ./ub3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define TT(t) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define R8(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7)
#define B4(b) b b b b
#define ITER 256
#define REPS 8
// each body: 8 independent chains, one instruction each; B4(B4(..)) of it... 32 per asm, x ITER
#define VADD(i)  "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define VAND(i)  "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define VXOR(i)  "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define VSHL(i)  "v_lshlrev_b32 %" #i ", 1, %" #i "\n\t"
#define VMUL(i)  "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define VCND(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define VCMP(i)  "v_cmp_lt_u32 vcc, %" #i ", %8\n\t"
#define VMBC(i)  "v_mbcnt_lo_u32_b32 %" #i ", -1, %" #i "\n\t"
#define VDPP(i)  "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define VRDL(i)  "v_readlane_b32 s20, %" #i ", 3\n\t"
#define VBPM(i)  "ds_bpermute_b32 %" #i ", %8, %" #i "\n\t"
#define VADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n\t"
#define VBFE(i)  "v_bfe_u32 %" #i ", %" #i ", 1, 31\n\t"
#define SADD(i)  "s_add_u32 %" #i ", %" #i ", %8\n\t"
#define SAND(i)  "s_and_b32 %" #i ", %" #i ", %8\n\t"
#define SMIX(i)  "s_add_u32 %" #i ", %" #i ", %8\n\tv_add_u32 %9, %9, %10\n\t"

template <int KIND>
__device__ void body(uint32_t (&v)[8], uint32_t w, uint32_t (&s)[8], uint32_t sb)
{
    for (int it = 0; it < ITER; it++) {
#define VB(X) asm volatile(B4(R8(X)) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(w) : "vcc", "s20", "memory")
#define SB(X) asm volatile(B4(R8(X)) : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7]) : "s"(sb) : "scc")
        if (KIND == 0) VB(VADD);
        if (KIND == 1) VB(VAND);
        if (KIND == 2) VB(VSHL);
        if (KIND == 3) VB(VMUL);
        if (KIND == 4) VB(VCND);
        if (KIND == 5) VB(VCMP);
        if (KIND == 6) VB(VMBC);
        if (KIND == 7) VB(VDPP);
        if (KIND == 8) VB(VRDL);
        if (KIND == 9) { VB(VBPM); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (KIND == 10) VB(VADD3);
        if (KIND == 11) VB(VBFE);
        if (KIND == 12) SB(SADD);
        if (KIND == 13) SB(SAND);
        if (KIND == 14) asm volatile(B4(R8(SMIX)) : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7]) : "s"(sb), "v"(v[0]), "v"(w) : "scc");
        if (KIND == 15) VB(VXOR);
    }
}
#define NKIND 16
extern __shared__ uint32_t lds[];
template <int K>
__global__ void __launch_bounds__(1024) k(uint64_t *out, uint32_t seed)
{
    uint32_t v[8], s[8];
    for (int i = 0; i < 8; i++) { v[i] = threadIdx.x * (i + 3) + seed; s[i] = __builtin_amdgcn_readfirstlane(seed * (i + 7)); }
    const uint32_t w = threadIdx.x * 5 + 1, sb = __builtin_amdgcn_readfirstlane(seed | 1);
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint64_t t0, t1;
    __syncthreads(); TT(t0);
    for (int rep = 0; rep < REPS; rep++) body<K>(v, w, s, sb);
    TT(t1);
    if ((threadIdx.x & 63) == 0) { out[(size_t)wave * 2] = t0; out[(size_t)wave * 2 + 1] = t1; }
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc += v[i] + s[i];
    if (acc == 0x12345u) out[0] = acc;
}
template <int K>
static void run(uint64_t *d, uint64_t *h, size_t bytes, double (&res)[NKIND][4][2])
{
    const int ncu = 256;
    (void)hipFuncSetAttribute((const void *)k<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int col = 0;
    for (int kw = 1; kw <= 8; kw *= 2, col++) {
        // kw <= 4: one workgroup of 4 kw waves per CU (100 KB of LDS: a second does not fit); kw = 8: two of 16 waves (64 KB each)
        const int waves = kw == 8 ? 16 : 4 * kw, nblk = kw == 8 ? 2 * ncu : ncu, ldsb = kw == 8 ? 64 * 1024 : 100 * 1024;
        float ms = 0;
        for (int r = 0; r < 2; r++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<K>, dim3(nblk), dim3(64 * waves), ldsb, 0, d, 12345u + r);
            (void)hipEventRecord(e1, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
        double span = 0;
        for (int wv = 0; wv < nblk * waves; wv++) span += (double)(h[(size_t)wv * 2 + 1] - h[(size_t)wv * 2]);
        span /= (double)nblk * waves;
        const double per_wave = (K == 14 ? 2.0 : 1.0) * 32.0 * ITER * REPS;
        res[K][col][0] = per_wave * 4 * kw / span;                               // per CU per s_memtime tick
        res[K][col][1] = per_wave * 4 * kw / ((double)ms * 1e-3 * 2.4e9);        // per CU per cycle of a 2.4 GHz clock, launch included
    }
}
int main()
{
    const char *names[NKIND] = {"v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_mul_lo_u32", "v_cndmask_b32 (back to back)", "v_cmp_lt_u32", "v_mbcnt_lo", "v_add_u32 DPP row_shr",
                                "v_readlane_b32", "ds_bpermute_b32", "v_add3_u32", "v_bfe_u32", "s_add_u32", "s_and_b32", "s_add + v_add pairs", "v_xor_b32"};
    uint64_t *d; size_t bytes = (size_t)256 * 32 * 2 * 8; (void)hipMalloc(&d, bytes);
    uint64_t *h = (uint64_t *)malloc(bytes);
    static double res[NKIND][4][2];
    run<0>(d, h, bytes, res); run<1>(d, h, bytes, res); run<2>(d, h, bytes, res); run<3>(d, h, bytes, res); run<4>(d, h, bytes, res); run<5>(d, h, bytes, res);
    run<6>(d, h, bytes, res); run<7>(d, h, bytes, res); run<8>(d, h, bytes, res); run<9>(d, h, bytes, res); run<10>(d, h, bytes, res); run<11>(d, h, bytes, res);
    run<12>(d, h, bytes, res); run<13>(d, h, bytes, res); run<14>(d, h, bytes, res); run<15>(d, h, bytes, res);
    for (int m = 0; m < 2; m++) {
        printf(m == 0 ? "whole-wavefront instructions per s_memtime tick (mean span of a wave):\n" : "the same per cycle of a 2.4 GHz clock, from the launch's hipEvent time (%d x %d x 32 instructions per wave):\n", REPS, ITER);
        printf("%-32s", "waves per SIMD:");
        for (int kw = 1; kw <= 8; kw *= 2) printf("  %8d", kw);
        printf("\n");
        for (int K = 0; K < NKIND; K++) {
            const bool percu = K == 12 || K == 13 || K == 14;
            printf("%-32s", names[K]);
            for (int c = 0; c < 4; c++) printf("  %8.3f", percu ? res[K][c][m] : res[K][c][m] / 4.0);
            printf("%s\n", K == 14 ? "   (per CU, both kinds counted)" : percu ? "   (per CU)" : "   (per SIMD)");
        }
    }
    return 0;
}
