/*
 * cbc_wave_gpu.h -- the 64-lane wavefront primitives the codec body is written against, gfx950.
 *
 * The codec body (cbc_encode_body.h) is a template over a "wave" policy W so that the exact same
 * control flow can be single-stepped on a CPU lock-step emulation in tests/emu (debugging aid
 * only: the product has no CPU path).  On the GPU a per-lane value is a plain uint32_t held in a
 * VGPR, wave-uniform values are plain scalars the compiler keeps in SGPRs, and every branch in
 * the body is wave-uniform.
 */
#ifndef CBC_WAVE_GPU_H
#define CBC_WAVE_GPU_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CBC_FN __device__ __forceinline__
#define CBC_MFN __device__ __forceinline__

struct WaveGPU {
    typedef uint32_t V32;   /* one 32-bit value per lane  */
    typedef bool     Mask;  /* one predicate per lane     */

    static CBC_FN V32 lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
    static CBC_FN V32 splat(uint32_t x) { return x; }
    static CBC_FN Mask all() { return true; }
    static CBC_FN void barrier() { __syncthreads(); }          /* the two wavefronts of a block's workgroup */
    static CBC_FN V32 select(Mask m, V32 a, V32 b) { return m ? a : b; }
    static CBC_FN uint64_t ballot(Mask m) { return __ballot(m); }
    /* make a value the compiler cannot prove uniform a scalar (it IS uniform by construction) */
    static CBC_FN uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    /* same, for a value just produced by a VALU-only instruction (f64 math): the empty asm makes the
     * VGPR value opaque so the optimiser cannot commute the readfirstlane back in front of the
     * producing instruction (it otherwise rewrites readfirstlane(cvt(x)) as cvt(readfirstlane(x)),
     * which lands in a VGPR again and drags every dependent scalar op onto the VALU) */
    static CBC_FN uint32_t to_scalar(uint32_t x)
    {
        asm volatile("" : "+v"(x));
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    }
    static CBC_FN uint32_t readlane(V32 v, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)uni(k)); }

    /* sum over the 64 lanes: inclusive DPP scan inside each row of 16, then row_bcast15 /
     * row_bcast31 carry the row totals up; lane 63 ends with the wave total. */
    static CBC_FN uint32_t reduce_add(V32 v)
    {
        int x = (int)v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   /* row_shr:1 */
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   /* row_shr:2 */
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   /* row_shr:4 */
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   /* row_shr:8 */
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   /* row_bcast:15 -> rows 1,3 */
        x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   /* row_bcast:31 -> rows 2,3 */
        return (uint32_t)__builtin_amdgcn_readlane(x, 63);
    }

    /* inclusive prefix sum over the lanes (the same DPP ladder as reduce_add, kept in every lane) */
    static CBC_FN V32 scan_incl_add(V32 v)
    {
        int x = (int)v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
        return (uint32_t)x;
    }

    /* number of set bits of `m` below this lane (v_mbcnt) */
    static CBC_FN V32 prefix_popc(uint64_t m)
    {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    }
    /* lane i takes lane i-1's value, lane 0 takes `fill` (ds_bpermute: LDS crossbar, no memory) */
    static CBC_FN V32 shift_up1(V32 v, uint32_t fill)
    {
        uint32_t l = lane();
        uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((l - 1u) << 2), (int)v);
        return l == 0u ? fill : t;
    }
    /* lane i takes the value of lane idx[i] (ds_bpermute); ({hi, lo} >> sh) low word, sh < 32 (v_alignbit) */
    static CBC_FN V32 lane_gather(V32 v, V32 idx) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)v); }
    static CBC_FN V32 funnel_shr(V32 hi, V32 lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
    /* bit `lane` of a wave-uniform 64-bit mask */
    static CBC_FN Mask lane_bit(uint64_t m)
    {
        uint32_t l = lane();
        uint32_t w = l < 32u ? (uint32_t)m : (uint32_t)(m >> 32);
        return ((w >> (l & 31u)) & 1u) != 0u;
    }
    /* floor(c * 2^32 / n), clamped to 2^32 - 1, per lane (c <= n < 2^21, n != 0).  The f64 quotient is
     * correctly rounded (IEEE division) with 53 bits for a value <= 2^32, i.e. an error below 2^-21,
     * while a non-integer c * 2^32 / n is at least 1/n > 2^-21 away from the next integer: the floor of
     * the rounded quotient is the floor of the exact one. */
    static CBC_FN V32 frac32(V32 c, V32 n)
    {
        double q = (double)c * 4294967296.0 / (double)(n ? n : 1u);
        return q >= 4294967295.0 ? 0xffffffffu : (uint32_t)q;
    }
    /* OR into LDS words: lanes may name the same word (ds_or_b32) */
    static CBC_FN void lds_add(uint32_t *p, V32 idx, V32 val, Mask m) { if (m) __hip_atomic_fetch_add(p + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    static CBC_FN void lds_or(uint32_t *p, V32 idx, V32 val, Mask m) { if (m) __hip_atomic_fetch_or(p + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    static CBC_FN void set_lane(V32 &v, uint32_t k, uint32_t val) { v = lane() == k ? val : v; }
    static CBC_FN V32 bswap_v(V32 x) { return __builtin_bswap32(x); }
    /* A wave-uniform value deliberately kept in a VECTOR register: every lane computes the same thing.
     * The coder recurrence uses it so that its ~50 operations per symbol go to the SIMD's vector unit
     * (four per CU) instead of the CU's single scalar unit, which all resident wavefronts share.  uv()
     * passes the value through an opaque v_mov so the compiler cannot prove it uniform and move the
     * arithmetic back to the scalar unit. */
    typedef uint32_t Uv;
    static CBC_FN Uv uv(uint32_t x) { uint32_t r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
    static CBC_FN Uv uv_opaque(Uv x) { asm volatile("" : "+v"(x)); return x; }     /* keeps the compiler from re-associating across it */
    static CBC_FN uint32_t uv_scalar(Uv x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    /* x >= c for a uniform x without the trip through a scalar register: v_cmp -> vcc -> branch */
    static CBC_FN bool uv_gt(Uv x, Uv y) { return __builtin_amdgcn_uicmp(x, y, 34) != 0ull; }               /* 34 = ICMP_UGT */
    static CBC_FN bool uv_ge(Uv x, uint32_t c) { return __builtin_amdgcn_uicmp(x, c, 35) != 0ull; }      /* 35 = ICMP_UGE */
    static CBC_FN void mul64(Uv a, uint32_t b, Uv &hi, Uv &lo) { uint64_t p = (uint64_t)a * b; hi = (uint32_t)(p >> 32); lo = (uint32_t)p; }
    static CBC_FN Uv clz_uv(Uv x) { return (uint32_t)__builtin_clz(x); }                  /* x != 0 */
    static CBC_FN void set_lane_uv(V32 &v, uint32_t k, Uv val) { v = lane() == k ? val : v; }
    /* The decoder's coder state (l, range, tag - l), wave-uniform like the encoder's: in vector registers by default -- the
     * decode kernel issues 845 scalar against 502 vector instructions per record and the scalar unit is the port its
     * wavefronts queue on -- or in scalar registers with -DCBC_DEC_STATE_SCALAR (A/B).  Conditions on it go through
     * dv_ge / dv_gt so that they stay uniform branches (v_cmp -> vcc) instead of divergent control flow. */
#ifndef CBC_DEC_STATE_SCALAR
    static CBC_FN Uv dv(uint32_t x) { return uv(x); }
    static CBC_FN bool dv_ge(Uv x, Uv y) { return __builtin_amdgcn_uicmp(x, y, 35) != 0ull; }              /* 35 = ICMP_UGE */
    static CBC_FN bool dv_gt(Uv x, Uv y) { return __builtin_amdgcn_uicmp(x, y, 34) != 0ull; }
    static CBC_FN bool dv_nz(Uv x) { return __builtin_amdgcn_uicmp(x, 0u, 33) != 0ull; }                   /* 33 = ICMP_NE */
    static CBC_FN uint32_t dv_scalar(Uv x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    static CBC_FN V32 dvv(Uv x) { return x; }
#else
    static CBC_FN Uv dv(uint32_t x) { return x; }
    static CBC_FN bool dv_ge(Uv x, Uv y) { return x >= y; }
    static CBC_FN bool dv_gt(Uv x, Uv y) { return x > y; }
    static CBC_FN bool dv_nz(Uv x) { return x != 0u; }
    static CBC_FN uint32_t dv_scalar(Uv x) { return x; }
    static CBC_FN V32 dvv(Uv x) { return x; }
#endif
    /* the hand-off counters in LDS: a real ds_read per poll (acquire), a store ordered after the wave's
     * earlier LDS traffic (release), and a short sleep between polls so a waiting wave leaves the issue
     * slots to the others */
    static CBC_FN uint32_t ctl_load(const uint32_t *p)
    {
        return uni(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    static CBC_FN void ctl_store(uint32_t *p, uint32_t v)
    {
        if (lane() == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    static CBC_FN void nap() { __builtin_amdgcn_s_sleep(2); }
    static CBC_FN void prio(int p) { if (p == 0) __builtin_amdgcn_s_setprio(0); else if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
    /* emulation-only cross-check hook */
    static CBC_FN void expect_eq(uint32_t, uint32_t, const char *) {}

    /* inclusive prefix maximum over the lanes (unsigned; the same DPP ladder, 0 is the identity) */
    static CBC_FN V32 scan_incl_max(V32 v)
    {
        uint32_t x = v;
#define CBC_DPP_MAX(ctrl, rmask) { uint32_t y_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, false); x = x > y_ ? x : y_; }
        CBC_DPP_MAX(0x111, 0xf) CBC_DPP_MAX(0x112, 0xf) CBC_DPP_MAX(0x114, 0xf) CBC_DPP_MAX(0x118, 0xf)
        CBC_DPP_MAX(0x142, 0xa) CBC_DPP_MAX(0x143, 0xc)
#undef CBC_DPP_MAX
        return x;
    }
    /* per-lane gathers / scatters; `m` false = lane does not touch memory */
    static CBC_FN V32 load32(const uint32_t *p, V32 idx, Mask m, uint32_t other) { return m ? p[idx] : other; }
    static CBC_FN void store32(uint32_t *p, V32 idx, V32 val, Mask m) { if (m) p[idx] = val; }
    /* 4 bytes at an arbitrary byte offset (global memory handles unaligned dwords) */
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    static CBC_FN V32 load32_bytes(const uint8_t *p, V32 byteoff, Mask m)
    {
        return m ? *(const u32_unaligned *)(p + byteoff) : 0u;
    }
    static CBC_FN V32 load8(const uint8_t *p, V32 byteoff, Mask m) { return m ? (uint32_t)p[byteoff] : 0u; }
    static CBC_FN void store8(uint8_t *p, V32 byteoff, V32 val, Mask m) { if (m) p[byteoff] = (uint8_t)val; }
    static CBC_FN void store32_bytes(uint8_t *p, V32 byteoff, V32 val, Mask m) { if (m) *(u32_unaligned *)(p + byteoff) = val; }
    static CBC_FN void store_rec(uint4 *p, V32 idx, Mask m, V32 a, V32 b, V32 c, V32 d) { if (m) p[idx] = make_uint4(a, b, c, d); }
    /* 16-byte record gather (cbc_read_rec) */
    static CBC_FN void load_rec(const uint4 *p, V32 idx, Mask m, V32 &a, V32 &b, V32 &c, V32 &d)
    {
        uint4 r = m ? p[idx] : make_uint4(0, 0, 0, 0);
        a = r.x; b = r.y; c = r.z; d = r.w;
    }
    /* A list in GLOBAL memory that the same wavefront appends to and re-reads (var events): reads
     * bypass the CU's vector L1 (agent-scope relaxed atomic load = glc/sc1 load) and list_fence()
     * drains the appending lane's stores before a scan's first read, so a lane never sees a stale line. */
    static CBC_FN V32 load32_list(const uint32_t *p, V32 idx, Mask m, uint32_t other)
    {
        return m ? __hip_atomic_load(p + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : other;
    }
    static CBC_FN void store32_list(uint32_t *p, V32 idx, V32 val, Mask m)
    {
        if (m) __hip_atomic_store(p + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    /* += into such a list, lanes may name the same word */
    static CBC_FN void list_add(uint32_t *p, V32 idx, V32 val, Mask m)
    {
        if (m) __hip_atomic_fetch_add(p + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static CBC_FN void append_list(uint32_t *p, uint32_t idx, uint32_t val)
    {
        if (lane() == 0) __hip_atomic_store(p + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    /* before re-reading the list: every append of this wavefront has reached L2 (the appends themselves
     * do not wait -- a scan is the rare case, an append happens for every var symbol) */
    static CBC_FN void list_fence() { __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0) */ }
    /* wave-uniform reads (same address in every lane) */
    static CBC_FN uint32_t read_uni(const uint32_t *p, uint32_t idx) { return uni(p[idx]); }
    static CBC_FN uint32_t read_uni8(const uint8_t *p, uint32_t idx) { return uni((uint32_t)p[idx]); }
    /* wave-uniform write: one lane stores */
    static CBC_FN void write_uni(uint32_t *p, uint32_t idx, uint32_t val) { if (lane() == 0) p[idx] = val; }

    /* qh = floor(range * hi / n) and ql = floor(range * lo / n) for range <= 2^26, lo < hi <= n < 2^21
     * (quotients < 2^27), exact.  The two divisions run side by side in lanes 0 and 1 of one VALU
     * instruction stream; `inv` = rcp((float)n), relative error <= 2^-22 (v_rcp_f32 is 1 ulp, (float)n
     * is exact).  Per lane, with p = range * c:
     *   q0 = trunc((float)(p >> 16) * 65536 * inv):   |q0 - p/n| <= 2^16/n + 33
     *   r0 = p - q0*n   (|r0| < 2^18 + 34 n < 2^27: the low 32 bits are the whole value)
     *   q1 = floor((float)r0 * inv): within 1 of floor(r0 / n)
     *   q  = q0 + q1, then one step of fix-up in either direction on r1 = r0 - q1*n.
     * The results go back to SGPRs through readlane so the coder recurrence stays on the scalar unit. */
    static CBC_FN void muldiv2(uint32_t range, uint32_t lo, uint32_t hi, uint32_t n, float inv,
                               uint32_t &ql, uint32_t &qh)
    {
        uint32_t c = lane() == 0u ? hi : lo;
        uint32_t plo = range * c, phi = __umulhi(range, c);
        float pf = (float)((phi << 16) | (plo >> 16)) * 65536.0f;
        uint32_t q0 = (uint32_t)(pf * inv);
        int32_t r0 = (int32_t)(plo - q0 * n);
        int32_t q1 = (int32_t)__builtin_floorf((float)r0 * inv);
        int32_t r1 = r0 - q1 * (int32_t)n;
        uint32_t q = q0 + (uint32_t)q1 - (uint32_t)(r1 < 0) + (uint32_t)(r1 >= (int32_t)n);
        qh = (uint32_t)__builtin_amdgcn_readlane((int)q, 0);
        ql = (uint32_t)__builtin_amdgcn_readlane((int)q, 1);
    }
    /* the same division in every lane: floor(range * c / n) for a count per lane, c <= n < 2^21, range <= 2^26
     * (CbcDec::scaled(): the decoder searches by scaled bounds, the quotients stay in vector registers) */
    static CBC_FN V32 muldiv_v(uint32_t range, V32 c, uint32_t n)
    {
        const float inv = __builtin_amdgcn_rcpf((float)n);
        uint32_t plo = range * c, phi = __umulhi(range, c);
        float pf = (float)((phi << 16) | (plo >> 16)) * 65536.0f;
        uint32_t q0 = (uint32_t)(pf * inv);
        int32_t r0 = (int32_t)(plo - q0 * n);
        int32_t q1 = (int32_t)__builtin_floorf((float)r0 * inv);
        int32_t r1 = r0 - q1 * (int32_t)n;
        return q0 + (uint32_t)q1 - (uint32_t)(r1 < 0) + (uint32_t)(r1 >= (int32_t)n);
    }
    /* floor(p / d) for p < 2^47, 2^24 < d <= 2^26 and a quotient < 2^21 (the decoder's target,
     * Arithmetic_stream.c:373-381), exact: same two-estimate scheme as muldiv2 with the divisor d.
     * (float)d is rounded (d has up to 27 bits), which only adds 2^-24 to the relative error. */
    static CBC_FN uint32_t divq(uint64_t p, uint32_t d)
    {
        float inv = __builtin_amdgcn_rcpf((float)d);
        uint32_t plo = (uint32_t)p, phi = (uint32_t)(p >> 32);
        float pf = (float)((phi << 16) | (plo >> 16)) * 65536.0f;
        uint32_t q0 = to_scalar((uint32_t)(pf * inv));
        int32_t r0 = (int32_t)(plo - q0 * d);
        int32_t q1 = (int32_t)to_scalar((uint32_t)(int32_t)__builtin_floorf((float)r0 * inv));
        int32_t r1 = r0 - q1 * (int32_t)d;
        return q0 + (uint32_t)q1 - (uint32_t)(r1 < 0) + (uint32_t)(r1 >= (int32_t)d);
    }
    /* reciprocals of all queued totals at once (lane k = k-th pending symbol) */
    static CBC_FN V32 recip_v(V32 n) { float r = __builtin_amdgcn_rcpf((float)n); return __builtin_bit_cast(uint32_t, r); }
    static CBC_FN float lane_float(V32 v, uint32_t k) { return __builtin_bit_cast(float, readlane(v, k)); }

    /* per lane: set bits; index of the lowest set bit (32 for 0) */
    static CBC_FN V32 popc_v(V32 x) { return (uint32_t)__builtin_popcount(x); }
    static CBC_FN V32 ctz_v(V32 x) { return x ? (uint32_t)__builtin_ctz(x) : 32u; }
    static CBC_FN uint32_t clz32(uint32_t x) { return (uint32_t)__builtin_clz(x); }       /* x != 0 */
    static CBC_FN uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }     /* x != 0 */
    static CBC_FN uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
    static CBC_FN uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
};

#endif
