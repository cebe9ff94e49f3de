// single-wave instruction-latency microbenchmarks (diagnostic only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory")
__global__ void k(uint64_t *out, uint32_t seed)
{
    uint64_t t0, t1; int n = 0;
    uint32_t a = __builtin_amdgcn_readfirstlane(seed), b = a * 3 + 1, c = 7;
    uint32_t v = threadIdx.x + seed, w = v * 5;
    __shared__ uint32_t lds[256];
    lds[threadIdx.x] = v; __syncthreads();
    // 0: dependent s_add chain
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("s_add_u32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc");) } T1(); out[n++] = t1 - t0;
    // 1: dependent s_mul_i32 chain
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("s_mul_i32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc");) } T1(); out[n++] = t1 - t0;
    // 2: dependent s_mul_hi_u32 chain
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("s_mul_hi_u32 %0, %0, %1\n\ts_or_b32 %0, %0, 0x10000" : "+s"(a) : "s"(b) : "scc");) } T1(); out[n++] = t1 - t0;
    // 3: independent s_add (two chains)
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("s_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %2" : "+s"(a), "+s"(c) : "s"(b) : "scc");) } T1(); out[n++] = t1 - t0;
    // 4: v_readlane -> s_add dependent -> (lane idx from sgpr)
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("s_and_b32 %1, %1, 63\n\tv_readlane_b32 %0, %2, %1\n\ts_add_u32 %1, %1, %0" : "=&s"(a), "+s"(c) : "v"(v) : "scc");) } T1(); out[n++] = t1 - t0;
    // 5: dependent v_add chain
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w));) } T1(); out[n++] = t1 - t0;
    // 6: sgpr -> v_mov -> v_readfirstlane -> sgpr round trip
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("v_mov_b32 %1, %0\n\tv_readfirstlane_b32 %0, %1" : "+s"(a), "+v"(v));) } T1(); out[n++] = t1 - t0;
    // 7: s_cmp + s_cselect dependent
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("s_cmp_ge_u32 %0, %1\n\ts_cselect_b32 %0, %1, %0\n\ts_add_u32 %0, %0, 1" : "+s"(a) : "s"(b) : "scc");) } T1(); out[n++] = t1 - t0;
    // 8: taken branch each iteration (short loop of 4 instrs)
    T0(); for (int i = 0; i < 1024; i++) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc"); } T1(); out[n++] = t1 - t0;
    // 9: LDS read uniform + readfirstlane round trip
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("v_and_b32 %1, 0x3fc, %1\n\tds_read_b32 %1, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %0, %1" : "+s"(a), "+v"(v) :: "memory");) } T1(); out[n++] = t1 - t0;
    // 10: v_cmp + v_cndmask (set_lane pattern) with sgpr operands
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("v_cmp_eq_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(v) : "s"(a), "v"(w), "v"(w) : "vcc");) } T1(); out[n++] = t1 - t0;
    // 11: DPP reduce ladder (6 dpp adds + readlane)
    T0(); for (int i = 0; i < 64; i++) {
        int x = (int)v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
        a += (uint32_t)__builtin_amdgcn_readlane(x, 63); asm volatile("" : "+s"(a)); v += a;
    } T1(); out[n++] = t1 - t0;
    // 12: ballot (v_cmp to sgpr pair) + s_ff1 + use
    T0(); for (int i = 0; i < 64; i++) { REP16(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n\ts_ff1_i32_b64 %0, vcc\n\ts_and_b32 %0, %0, 63" : "+s"(a) : "v"(w) : "vcc", "scc");) } T1(); out[n++] = t1 - t0;
    out[n++] = a + c; out[n++] = v;
}
int main() {
    fprintf(stderr, "start\n"); uint64_t *d; hipError_t e = hipMalloc(&d, 256); fprintf(stderr, "malloc %d\n", (int)e); hipMemset(d, 0, 256);
    for (int r = 0; r < 2; r++) { k<<<1, 64>>>(d, 12345u); e = hipDeviceSynchronize(); fprintf(stderr, "run %d: %d\n", r, (int)e); }
    uint64_t h[32]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    const char *nm[] = {"dep s_add", "dep s_mul_i32", "dep s_mul_hi+s_or (2)", "2 indep s_add chains (2)", "s_and+v_readlane+s_add (3)", "dep v_add", "v_mov+v_readfirstlane (2)", "s_cmp+s_cselect+s_add (3)", "loop: 1 add + loop ctl, taken branch", "v_and+ds_read+wait+readfirstlane (4)", "v_cmp+v_cndmask (2)", "DPP reduce ladder + readlane (per reduce)", "v_cmp->vcc + s_ff1 + s_and (3)"};
    int cnt[] = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 64, 1024};
    for (int i = 0; i < 13; i++) printf("%-45s %8.1f cycles per group\n", nm[i], (double)h[i] / cnt[i]);
    return 0;
}
