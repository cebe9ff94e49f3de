// lone-wave issue rate vs loop body size (diagnostic only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory")
#define A1 "s_add_u32 %0, %0, %1\n\t"
#define A4 A1 A1 A1 A1
#define A8 A4 A4
#define A16 A8 A8
#define A32 A16 A16
#define A64 A32 A32
#define V1 "v_add_u32 %0, %0, %1\n\t"
#define V4 V1 V1 V1 V1
#define V8 V4 V4
#define V16 V8 V8
#define V32_ V16 V16
#define V64 V32_ V32_
#define LOOP(body, iters) T0(); for (int i = 0; i < iters; i++) { asm volatile(body : "+s"(a) : "s"(b) : "scc"); } T1(); out[n++] = t1 - t0;
#define VLOOP(body, iters) T0(); for (int i = 0; i < iters; i++) { asm volatile(body : "+v"(v) : "v"(w)); } T1(); out[n++] = t1 - t0;
__global__ void k(uint64_t *out, uint32_t seed)
{
    uint64_t t0, t1; int n = 0;
    uint32_t a = __builtin_amdgcn_readfirstlane(seed), b = a * 3 + 1;
    uint32_t v = threadIdx.x + seed, w = v * 5;
    LOOP(A4, 4096) LOOP(A8, 2048) LOOP(A8 A4, 1024) LOOP(A16, 1024) LOOP(A16 A8, 1024) LOOP(A32, 512) LOOP(A32 A16, 512) LOOP(A64, 256) LOOP(A64 A64, 128) LOOP(A64 A64 A64 A64, 64)
    VLOOP(V4, 4096) VLOOP(V8, 2048) VLOOP(V16, 1024) VLOOP(V32_, 512) VLOOP(V64, 256) VLOOP(V64 V64 V64 V64, 64)
    out[n++] = a; out[n++] = v;
}
int main() {
    uint64_t *d; (void)hipMalloc(&d, 512); (void)hipMemset(d, 0, 512);
    for (int r = 0; r < 2; r++) { k<<<1, 64>>>(d, 12345u); (void)hipDeviceSynchronize(); }
    uint64_t h[64]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int sz[] = {4, 8, 12, 16, 24, 32, 48, 64, 128, 256, 4, 8, 16, 32, 64, 256};
    int it[] = {4096, 2048, 1024, 1024, 1024, 512, 512, 256, 128, 64, 4096, 2048, 1024, 512, 256, 64};
    for (int i = 0; i < 16; i++) printf("%s loop body %3d instrs: %6.2f cycles/instr (incl. loop control)\n", i < 10 ? "SALU" : "VALU", sz[i], (double)h[i] / ((double)sz[i] * it[i]));
    return 0;
}
