/*
 * wave_emu.h -- 64-lane lock-step emulation of the wave policy used by cbc_amd/csrc (the _body.h files).
 *
 * DEBUGGING / TEST AID ONLY.  It lets the kernel body be single-stepped, bounds-checked
 * (ASan/UBSan) and compared with the oracle on a machine without a GPU, so that a kernel is not
 * launched on real hardware before its indexing is known to be in range.  It is never built into
 * the product library and never stands in for the HIP path: cbc_amd fails loudly without a GPU.
 */
#ifndef CBC_WAVE_EMU_H
#define CBC_WAVE_EMU_H

#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define CBC_FN static inline
#define CBC_MFN inline

struct uint4 { uint32_t x, y, z, w; };

struct EmuMask {
    bool b[64];
    EmuMask operator&(const EmuMask &o) const { EmuMask r; for (int i = 0; i < 64; i++) r.b[i] = b[i] && o.b[i]; return r; }
    EmuMask operator|(const EmuMask &o) const { EmuMask r; for (int i = 0; i < 64; i++) r.b[i] = b[i] || o.b[i]; return r; }
    EmuMask operator!() const { EmuMask r; for (int i = 0; i < 64; i++) r.b[i] = !b[i]; return r; }
    EmuMask operator&(bool o) const { EmuMask r; for (int i = 0; i < 64; i++) r.b[i] = b[i] && o; return r; }
};

struct EmuV32 {
    uint32_t v[64];
};

#define EMU_BINOP(op)                                                                                   \
    static inline EmuV32 operator op(const EmuV32 &a, const EmuV32 &b) { EmuV32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b.v[i]; return r; } \
    static inline EmuV32 operator op(const EmuV32 &a, uint32_t b) { EmuV32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b; return r; }          \
    static inline EmuV32 operator op(uint32_t a, const EmuV32 &b) { EmuV32 r; for (int i = 0; i < 64; i++) r.v[i] = a op b.v[i]; return r; }
EMU_BINOP(+) EMU_BINOP(-) EMU_BINOP(*) EMU_BINOP(&) EMU_BINOP(|) EMU_BINOP(^)
#undef EMU_BINOP
/* shifts use the low 5 bits of the count, like v_lshlrev_b32 / v_lshrrev_b32 */
static inline EmuV32 operator<<(const EmuV32 &a, const EmuV32 &b) { EmuV32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] << (b.v[i] & 31u); return r; }
static inline EmuV32 operator<<(const EmuV32 &a, uint32_t b) { EmuV32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] << (b & 31u); return r; }
static inline EmuV32 operator>>(const EmuV32 &a, const EmuV32 &b) { EmuV32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] >> (b.v[i] & 31u); return r; }
static inline EmuV32 operator>>(const EmuV32 &a, uint32_t b) { EmuV32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] >> (b & 31u); return r; }

#define EMU_CMP(op)                                                                                     \
    static inline EmuMask operator op(const EmuV32 &a, const EmuV32 &b) { EmuMask r; for (int i = 0; i < 64; i++) r.b[i] = a.v[i] op b.v[i]; return r; } \
    static inline EmuMask operator op(const EmuV32 &a, uint32_t b) { EmuMask r; for (int i = 0; i < 64; i++) r.b[i] = a.v[i] op b; return r; } \
    static inline EmuMask operator op(uint32_t a, const EmuV32 &b) { EmuMask r; for (int i = 0; i < 64; i++) r.b[i] = a op b.v[i]; return r; }
EMU_CMP(==) EMU_CMP(!=) EMU_CMP(<) EMU_CMP(<=) EMU_CMP(>) EMU_CMP(>=)
#undef EMU_CMP

extern "C" void emu_oob(const char *what);

struct WaveEmu {
    typedef EmuV32 V32;
    typedef EmuMask Mask;

    static V32 lane() { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)i; return r; }
    static V32 splat(uint32_t x) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = x; return r; }
    static void barrier() { emu_oob("barrier reached in the fused emulation"); }
    static Mask all() { Mask m; for (int i = 0; i < 64; i++) m.b[i] = true; return m; }
    static V32 select(const Mask &m, const V32 &a, const V32 &b) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = m.b[i] ? a.v[i] : b.v[i]; return r; }
    static uint64_t ballot(const Mask &m) { uint64_t r = 0; for (int i = 0; i < 64; i++) if (m.b[i]) r |= 1ull << i; return r; }
    static uint32_t uni(uint32_t x) { return x; }
    static uint32_t readlane(const V32 &v, uint32_t k) { if (k >= 64) { emu_oob("readlane index"); return 0; } return v.v[k]; }
    static uint32_t reduce_add(const V32 &v) { uint32_t s = 0; for (int i = 0; i < 64; i++) s += v.v[i]; return s; }

    static V32 prefix_popc(uint64_t m) { V32 r; uint32_t c = 0; for (int i = 0; i < 64; i++) { r.v[i] = c; c += (uint32_t)((m >> i) & 1u); } return r; }
    static V32 lane_gather(const V32 &v, const V32 &idx) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = v.v[idx.v[i] & 63u]; return r; }
    static V32 funnel_shr(const V32 &hi, const V32 &lo, uint32_t sh)
    { V32 r; if (sh >= 32u) emu_oob("funnel_shr shift"); for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)((((uint64_t)hi.v[i] << 32) | lo.v[i]) >> sh); return r; }
    static V32 shift_up1(const V32 &v, uint32_t fill) { V32 r; r.v[0] = fill; for (int i = 1; i < 64; i++) r.v[i] = v.v[i - 1]; return r; }
    static Mask lane_bit(uint64_t m) { Mask r; for (int i = 0; i < 64; i++) r.b[i] = ((m >> i) & 1u) != 0; return r; }
    static V32 frac32(const V32 &c, const V32 &n)
    {
        V32 r;
        for (int i = 0; i < 64; i++) {
            double q = (double)c.v[i] * 4294967296.0 / (double)(n.v[i] ? n.v[i] : 1u);
            r.v[i] = q >= 4294967295.0 ? 0xffffffffu : (uint32_t)q;
            if (n.v[i] && c.v[i] <= n.v[i]) {                       /* the f64 route must give the exact floor */
                unsigned __int128 e = ((unsigned __int128)c.v[i] << 32) / n.v[i];
                uint32_t want = e > 0xffffffffu ? 0xffffffffu : (uint32_t)e;
                if (r.v[i] != want) emu_oob("frac32 is not the exact floor");
            }
        }
        return r;
    }
    static void lds_or(uint32_t *p, const V32 &idx, const V32 &val, const Mask &m) { for (int i = 0; i < 64; i++) if (m.b[i]) p[idx.v[i]] |= val.v[i]; }
    static void lds_add(uint32_t *p, const V32 &idx, const V32 &val, const Mask &m) { for (int i = 0; i < 64; i++) if (m.b[i]) p[idx.v[i]] += val.v[i]; }
    static void set_lane(V32 &v, uint32_t k, uint32_t val) { if (k >= 64) { emu_oob("set_lane index"); return; } v.v[k] = val; }
    static V32 bswap_v(const V32 &x) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = __builtin_bswap32(x.v[i]); return r; }
    static uint32_t ctl_load(const uint32_t *) { emu_oob("hand-off counter read in the fused emulation"); return 0; }
    static void ctl_store(uint32_t *, uint32_t) { emu_oob("hand-off counter write in the fused emulation"); }
    static void nap() {}
    static void prio(int) {}
    typedef uint32_t Uv;
    static Uv uv(uint32_t x) { return x; }
    static uint32_t uv_scalar(Uv x) { return x; }
    static Uv uv_opaque(Uv x) { return x; }
    static bool uv_ge(Uv x, uint32_t c) { return x >= c; }
    static Uv dv(uint32_t x) { return x; }
    static bool dv_ge(Uv x, Uv y) { return x >= y; }
    static bool dv_gt(Uv x, Uv y) { return x > y; }
    static bool dv_nz(Uv x) { return x != 0u; }
    static uint32_t dv_scalar(Uv x) { return x; }
    static V32 dvv(Uv x) { return splat(x); }
    static bool uv_gt(Uv x, Uv y) { return x > y; }
    static void mul64(Uv a, uint32_t b, Uv &hi, Uv &lo) { uint64_t p = (uint64_t)a * b; hi = (uint32_t)(p >> 32); lo = (uint32_t)p; }
    static Uv clz_uv(Uv x) { return clz32(x); }
    static void set_lane_uv(V32 &v, uint32_t k, Uv val) { set_lane(v, k, val); }
    static void expect_eq(uint32_t a, uint32_t b, const char *what) { if (a != b) emu_oob(what); }
    static V32 scan_incl_max(const V32 &v) { V32 r; uint32_t m = 0; for (int i = 0; i < 64; i++) { if (v.v[i] > m) m = v.v[i]; r.v[i] = m; } return r; }
    static V32 scan_incl_add(const V32 &v) { V32 r; uint32_t s = 0; for (int i = 0; i < 64; i++) { s += v.v[i]; r.v[i] = s; } return r; }
    static V32 load8(const uint8_t *p, const V32 &off, const Mask &m) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = m.b[i] ? p[off.v[i]] : 0u; return r; }
    static void store8(uint8_t *p, const V32 &off, const V32 &val, const Mask &m) { for (int i = 0; i < 64; i++) if (m.b[i]) p[off.v[i]] = (uint8_t)val.v[i]; }
    static void store32_bytes(uint8_t *p, const V32 &off, const V32 &val, const Mask &m) { for (int i = 0; i < 64; i++) if (m.b[i]) memcpy(p + off.v[i], &val.v[i], 4); }
    static void store_rec(uint4 *p, const V32 &idx, const Mask &m, const V32 &a, const V32 &b, const V32 &c, const V32 &d)
    { for (int i = 0; i < 64; i++) if (m.b[i]) { uint4 r = {a.v[i], b.v[i], c.v[i], d.v[i]}; p[idx.v[i]] = r; } }
    static uint32_t divq(uint64_t p, uint32_t d)
    {
        float inv = 1.0f / (float)d;
        static thread_local unsigned tick = 0;
        uint32_t bits; memcpy(&bits, &inv, 4); bits += (uint32_t)((int)(tick++ % 5u) - 2); memcpy(&inv, &bits, 4);
        uint32_t plo = (uint32_t)p, phi = (uint32_t)(p >> 32);
        float pf = (float)((phi << 16) | (plo >> 16)) * 65536.0f;
        uint32_t q0 = (uint32_t)(pf * inv);
        int32_t r0 = (int32_t)(plo - q0 * d);
        int32_t q1 = (int32_t)__builtin_floorf((float)r0 * inv);
        int32_t r1 = r0 - q1 * (int32_t)d;
        uint32_t q = q0 + (uint32_t)q1 - (uint32_t)(r1 < 0) + (uint32_t)(r1 >= (int32_t)d);
        if (q != (uint32_t)(p / d)) emu_oob("divq two-estimate mismatch");
        return q;
    }
    static V32 load32(const uint32_t *p, const V32 &idx, const Mask &m, uint32_t other)
    { V32 r; for (int i = 0; i < 64; i++) r.v[i] = m.b[i] ? p[idx.v[i]] : other; return r; }
    static void store32(uint32_t *p, const V32 &idx, const V32 &val, const Mask &m)
    { for (int i = 0; i < 64; i++) if (m.b[i]) p[idx.v[i]] = val.v[i]; }
    static V32 load32_bytes(const uint8_t *p, const V32 &off, const Mask &m)
    { V32 r; for (int i = 0; i < 64; i++) { uint32_t t = 0; if (m.b[i]) memcpy(&t, p + off.v[i], 4); r.v[i] = t; } return r; }
    static void load_rec(const uint4 *p, const V32 &idx, const Mask &m, V32 &a, V32 &b, V32 &c, V32 &d)
    {
        for (int i = 0; i < 64; i++) {
            uint4 r = {0, 0, 0, 0};
            if (m.b[i]) r = p[idx.v[i]];
            a.v[i] = r.x; b.v[i] = r.y; c.v[i] = r.z; d.v[i] = r.w;
        }
    }
    static V32 load32_list(const uint32_t *p, const V32 &idx, const Mask &m, uint32_t other) { return load32(p, idx, m, other); }
    static void append_list(uint32_t *p, uint32_t idx, uint32_t val) { p[idx] = val; }
    static void list_add(uint32_t *p, const V32 &idx, const V32 &val, const Mask &m) { for (int i = 0; i < 64; i++) if (m.b[i]) p[idx.v[i]] += val.v[i]; }
    static void store32_list(uint32_t *p, const V32 &idx, const V32 &val, const Mask &m) { store32(p, idx, val, m); }
    static void list_fence() {}
    static uint32_t read_uni(const uint32_t *p, uint32_t idx) { return p[idx]; }
    static uint32_t read_uni8(const uint8_t *p, uint32_t idx) { return p[idx]; }
    static void write_uni(uint32_t *p, uint32_t idx, uint32_t val) { p[idx] = val; }

    /* same f32 two-estimate formula as the GPU policy, cross-checked against exact integer division
     * on every call.  The reciprocal is deliberately perturbed by up to +-2 ulp (v_rcp_f32 is a 1-ulp
     * approximation, the CPU's 1.0f/x is correctly rounded) so the fix-up logic is exercised. */
    static uint32_t muldiv1(uint32_t range, uint32_t c, uint32_t n, float inv)
    {
        uint64_t p = (uint64_t)range * c;
        uint32_t plo = (uint32_t)p, phi = (uint32_t)(p >> 32);
        float pf = (float)((phi << 16) | (plo >> 16)) * 65536.0f;
        uint32_t q0 = (uint32_t)(pf * inv);
        int32_t r0 = (int32_t)(plo - q0 * n);
        int32_t q1 = (int32_t)__builtin_floorf((float)r0 * inv);
        int32_t r1 = r0 - q1 * (int32_t)n;
        uint32_t q = q0 + (uint32_t)q1 - (uint32_t)(r1 < 0) + (uint32_t)(r1 >= (int32_t)n);
        if (q != (uint32_t)(p / n)) emu_oob("muldiv f32 two-estimate mismatch");
        return q;
    }
    static void muldiv2(uint32_t range, uint32_t lo, uint32_t hi, uint32_t n, float inv, uint32_t &ql, uint32_t &qh)
    { qh = muldiv1(range, hi, n, inv); ql = muldiv1(range, lo, n, inv); }
    static V32 recip_v(const V32 &n)
    {
        V32 r;
        for (int i = 0; i < 64; i++) {
            float f = n.v[i] ? 1.0f / (float)n.v[i] : 0.0f;
            uint32_t bits; memcpy(&bits, &f, 4);
            static thread_local unsigned tick = 0;
            if (n.v[i]) bits += (uint32_t)((int)(tick++ % 5u) - 2);          /* -2..+2 ulp */
            r.v[i] = bits;
        }
        return r;
    }
    static V32 muldiv_v(uint32_t range, const V32 &c, uint32_t n)
    {
        V32 r; const V32 iv = recip_v(splat(n));
        for (int i = 0; i < 64; i++) {
            if (c.v[i] > n || n >= (1u << 21) || range > (1u << 26)) emu_oob("muldiv_v operand range");
            float f; memcpy(&f, &iv.v[i], 4); r.v[i] = muldiv1(range, c.v[i], n, f);
        }
        return r;
    }
    static float lane_float(const V32 &v, uint32_t k) { float f; uint32_t b = readlane(v, k); memcpy(&f, &b, 4); return f; }
    static uint32_t clz32(uint32_t x) { if (!x) emu_oob("clz32(0)"); return (uint32_t)__builtin_clz(x); }
    static uint32_t ctz64(uint64_t x) { if (!x) emu_oob("ctz64(0)"); return (uint32_t)__builtin_ctzll(x); }
    static uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
    static V32 popc_v(const V32 &x) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)__builtin_popcount(x.v[i]); return r; }
    static V32 ctz_v(const V32 &x) { V32 r; for (int i = 0; i < 64; i++) r.v[i] = x.v[i] ? (uint32_t)__builtin_ctz(x.v[i]) : 32u; return r; }
    static uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
};

#endif
