/*
 * cbc_encode_body.h -- one arithmetic stream per workgroup: the per-read encode loop of the
 * reference (compress_line .. encoder_last_step), split over two wavefronts by model ownership
 * (see "hand-off" below; CbcEnc::publish / pull); the CPU emulation runs it fused in one.
 *
 * Template parameter W supplies the 64-lane primitives (cbc_wave_gpu.h on the GPU).  Every
 * branch below is wave-uniform; per-lane data only flows through W::select / ballot / reduce_add /
 * readlane and masked loads and stores.
 *
 * Where the 64 lanes are used:
 *   - one lane per RECORD for the per-record models (rlength, pos, flag, match): inside a block
 *     they are counting models, so (cum, count) of 64 records come from ballots / v_mbcnt
 *     (fixed_group); one lane per SYMBOL for the scaled fractions the coder step divides by
 *     (frac32) and for placing the output bits (pack)
 *   - read-vs-reference compare: 4 bases per lane, one ballot            (read_compression.c:291-296)
 *   - cumulative-frequency lookup of the edit models: masked gather of the sparse/dense table +
 *     wave sum                                                           (stream_model.c:64-69)
 *   - var-context statistics: ballot + popcount over the block's var events
 *   - snpInRef window: a 256-bit sliding bitmap, first-set search         (read_compression.c:703-718)
 * What is serial: the range-coder recurrence (l, u, pending E3 count) -- kept in VECTOR registers on
 * purpose (W::Uv) -- and the edit models' totals (wave-uniform scalars).
 *
 * Model tables are kept SPARSE and exact (SURVEY.md section 7 hard part 2): every symbol of the
 * big models starts at count 1 and rescaling maps 1 -> (1>>1)+1 = 1, so with e[s] = count[s]-1:
 *     cum(x) = x + sum_{s<x} e[s],   n = card + sum e[s],   rescale: e' = (1+e)>>1.
 *
 * CBC_LOOP note (uniformity).  A chunk loop `for (b = 0; b < N; b += 64) { m = lane + b < N; ... }`
 * must take its bound from W::uni(N), an opaque copy: hipcc otherwise proves "m false => loop done"
 * (lane id < 64), threads the per-lane mask into the loop exit, the exit becomes divergent, and
 * everything live across the loop is classified divergent and moved from the scalar unit to VGPRs +
 * exec-mask branches.
 */
#ifndef CBC_ENCODE_BODY_H
#define CBC_ENCODE_BODY_H

#include <stdint.h>
#include <type_traits>
#include "../../include/cbc_gpu.h"
#include "cbc_plan.h"

#ifndef CBC_PRIO_CODER
#define CBC_PRIO_CODER 3                  /* s_setprio of the two wavefronts of a block (A/B: profiles/r02_ab_kernels.log) */
#define CBC_PRIO_MODEL 0
#endif
#define CBC_REC_PACK_AT 56u    /* CbcEnc::step packs once this many steps are recorded (see room()) */
#define CBC_AWORD     26u
#define CBC_M26       ((1u << 26) - 1u)
#define CBC_M25       ((1u << 25) - 1u)
#define CBC_RESCALE   (1u << 20)
#define CBC_NVARCTX   0xffffu
#define CBC_NOMEMO    0xffffffffu
#define CBC_BLOOM_LOG2 13u       /* log2(32 * CBC_BLOOM_WORDS) */
#define CBC_ROLE_FUSED 0u      /* one wavefront does models and coder (CPU emulation)        */
#define CBC_ROLE_MODEL 1u      /* wavefront 0 of the workgroup: models, produces symbol batches */
#define CBC_ROLE_CODER 2u      /* wavefront 1: range coder, consumes them                        */

/* lane map of the register-resident small models (one VGPR, `small`) */
#define CBC_LT_MATCH   0u     /* 4 ctx x 2                 sam_models.c:204-241 */
#define CBC_LT_SAMEREF 8u     /* 1 x 2                     sam_models.c:617     */
#define CBC_LT_CHARS   16u    /* 6 rows x 5, stride 8      sam_models.c:350-411 */

/* LDS words in front of the two variable-size tables */
#define CBC_LDS_RLEN    0u                         /* 256: rlength[0] excess            */
#define CBC_LDS_SNPS    256u                       /* 256: snps excess                  */
#define CBC_LDS_INDELS  512u                       /* 256: indels excess                */
#define CBC_LDS_RNKEY   768u                       /* CBC_CAP_NAME: (ctx<<8)|char       */
#define CBC_LDS_RNEXC   (768u + CBC_CAP_NAME)      /* CBC_CAP_NAME                      */
#define CBC_LDS_BLOOM   (768u + 2u * CBC_CAP_NAME) /* CBC_BLOOM_WORDS: Bloom filter on var ctx (two hashes) */
#define CBC_LDS_P0      (768u + 2u * CBC_CAP_NAME + CBC_BLOOM_WORDS) /* 2 x CBC_P0_WORDS: var events of the p = 0 contexts, per strand */
#define CBC_LDS_BATCH   CBC_PLAN_TABLE_WORDS       /* CBC_BATCH_SLOTS x CBC_BATCH_WORDS: model wave -> coder wave */
#ifndef CBC_BATCH_MIN
#define CBC_BATCH_MIN   40u    /* <= 64 - 12 (a record's fixed symbols) - 4 (edit counts) - slack: see the 56 checks */
#endif
#define CBC_LDS_CTL     (CBC_LDS_BATCH + CBC_BATCH_SLOTS * CBC_BATCH_WORDS)   /* 8: produced, consumed */
#define CBC_LDS_RING    (CBC_LDS_CTL + 8u)
#define CBC_LDS_FIXED   CBC_PLAN_LDS_FIXED_WORDS
/* then pos_val[cap_pos], pos_occ[cap_pos], pos_pre[cap_pos]; the var-event list lives in global memory (see var_code) */

#ifdef CBC_EMU_TRACE
extern "C" void cbc_emu_trace(uint32_t read, uint32_t lo, uint32_t cnt, uint32_t n);
#endif

struct cbc_enc_args {
    const cbc_read_rec   *recs;
    const uint8_t        *seq;
    const uint32_t       *tok;
    const uint8_t        *names;
    const cbc_block_desc *blocks;
    const uint8_t        *ref;
    uint8_t              *out;
    cbc_block_result     *results;
    uint64_t ref_bytes, out_bytes, seq_bytes, n_tok, n_recs;
    uint32_t n_blocks, cap_pos, cap_var, names_bytes;
};

/* CBC_STAMP: diagnostic build only (never shipped): per-section s_memtime sums, written over the
 * start of the block's payload area at the end -- outputs of such a build are garbage by design. */
#if defined(CBC_STAMP) && defined(__HIP_DEVICE_COMPILE__)
#define CBC_T0() do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); E.t_last = t_; } while (0)
#define CBC_TS(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); E.t_sum[k] += t_ - E.t_last; E.t_last = t_; } while (0)
#define CBC_TSM(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); t_sum[k] += t_ - t_last; t_last = t_; } while (0)
#elif defined(CBC_MARKS) && defined(__HIP_DEVICE_COMPILE__)   /* diagnostic: section markers in the ISA listing */
#define CBC_TSM(k) asm volatile("; ==== CBC_MARK M" #k)
#define CBC_T0() asm volatile("; ==== CBC_MARK start")
#define CBC_TS(k) asm volatile("; ==== CBC_MARK " #k)
#else
#define CBC_T0() do {} while (0)
#define CBC_TS(k) do {} while (0)
#define CBC_TSM(k) do {} while (0)
#endif

/* bytes readable from offset `off` of a buffer with `total` bytes, clamped to 32 bits */
/* a <= b for 64-bit values using 32-bit compares only: gfx950's scalar ALU has no ordered 64-bit
 * compare, and one VALU compare on a uniform value drags everything downstream onto the VALU */
CBC_FN bool cbc_le64(uint64_t a, uint64_t b)
{
    uint32_t ah = (uint32_t)(a >> 32), al = (uint32_t)a, bh = (uint32_t)(b >> 32), bl = (uint32_t)b;
    return (ah < bh) | ((ah == bh) & (al <= bl));
}
/* [off, off + len) lies inside a buffer of `total` bytes; written without the sum off + len, which a
 * crafted descriptor could make wrap */
CBC_FN bool cbc_fits64(uint64_t off, uint64_t len, uint64_t total)
{
    return cbc_le64(off, total) && cbc_le64(len, total - off);
}
CBC_FN uint32_t cbc_avail32(uint64_t total, uint32_t off)
{
    uint64_t a = total - off;                       /* callers guarantee off <= total or get 0 below */
    if (!cbc_le64((uint64_t)off, total)) return 0u;
    return (uint32_t)(a >> 32) ? 0xffffffffu : (uint32_t)a;
}
CBC_FN uint32_t cbc_basepair(uint32_t c)            /* char2basepair sam_models.c:11-21 */
{
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

/* GEN = false: the block kernels (LDS-resident sparse tables, counting-model closed forms for the per-record
 * models).  GEN = true: the whole-file stream kernel of cbc_stream_body.h -- the same coder, bit packer and edit
 * walk, with the general (rescaling) form of every model and the var table dense in global memory. */
/* ---- snpInRef[cumsumP - 1 .. + 255] (read_compression.c:589, 703-718): a 256-bit window ---------------------------------
 * Kept in ONE vector register, 32 bits in each of lanes 0..7 (the other lanes hold 0): sliding it is two lane gathers and a
 * funnel shift, the first mark at or after p is one ballot and two count-trailing-zeros, setting a mark is a compare and an
 * OR.  (Round 1 kept it in four 64-bit scalars: ~30 scalar instructions per slide or search, on the unit that is these
 * kernels' busiest port; -DCBC_WIN_SCALAR restores that form for A/B.) */
template <class W>
struct CbcWin {
    typedef typename W::V32 V32;
#ifndef CBC_WIN_SCALAR
    V32 w;
    CBC_MFN void clear() { w = W::splat(0u); }
    CBC_MFN void shift(uint32_t d)                      /* drop the d lowest positions */
    {
        if (d == 0u) return;
        if (d >= 256u) { clear(); return; }
        const V32 ln = W::lane();
        const V32 src = ln + (d >> 5);
        const V32 lo = W::lane_gather(w, src), hi = W::lane_gather(w, src + 1u);      /* lanes 8.. hold 0 */
        w = W::select(ln < 8u, W::funnel_shr(hi, lo, d & 31u), W::splat(0u));
    }
    /* compute_delta_to_first_snp, read_compression.c:703-718 */
    CBC_MFN uint32_t first(uint32_t p, uint32_t rl)
    {
        uint32_t out = rl + 2u;
        if (p >= rl) return out;
        const V32 ln = W::lane();
        const uint32_t pw = p >> 5;
        const V32 a = W::select(ln == pw, w & (0xffffffffu << (p & 31u)), W::select(ln > pw, w, W::splat(0u)));
        const uint64_t hb = W::ballot(a != 0u);
        if (hb) {
            const uint32_t hl = W::ctz64(hb);
            const uint32_t pos = hl * 32u + W::ctz64((uint64_t)W::readlane(a, hl));
            if (pos < rl) out = pos - p;
        }
        return out;
    }
    CBC_MFN void set(uint32_t k)                        /* k >= 256: outside the window, ignored */
    {
        if (k < 256u) w = w | W::select(W::lane() == (k >> 5), W::splat(1u << (k & 31u)), W::splat(0u));
    }
#else
    uint64_t w0, w1, w2, w3;
    CBC_MFN void clear() { w0 = w1 = w2 = w3 = 0; }
    CBC_MFN void shift(uint32_t d)
    {
        if (d == 0u) return;
        if (d >= 256u) { clear(); return; }
        uint32_t wsh = d >> 6, bsh = d & 63u;
        if (wsh == 1u) { w0 = w1; w1 = w2; w2 = w3; w3 = 0; }
        else if (wsh == 2u) { w0 = w2; w1 = w3; w2 = 0; w3 = 0; }
        else if (wsh == 3u) { w0 = w3; w1 = 0; w2 = 0; w3 = 0; }
        if (bsh) {
            uint32_t inv = 64u - bsh;
            w0 = (w0 >> bsh) | (w1 << inv);
            w1 = (w1 >> bsh) | (w2 << inv);
            w2 = (w2 >> bsh) | (w3 << inv);
            w3 = w3 >> bsh;
        }
    }
    CBC_MFN uint32_t first(uint32_t p, uint32_t rl)
    {
        uint32_t out = rl + 2u;
        if (p >= rl) return out;
        uint32_t pw = p >> 6; uint64_t pm = ~0ull << (p & 63u);
        uint64_t a0 = pw == 0u ? (w0 & pm) : 0ull;
        uint64_t a1 = pw == 1u ? (w1 & pm) : (pw < 1u ? w1 : 0ull);
        uint64_t a2 = pw == 2u ? (w2 & pm) : (pw < 2u ? w2 : 0ull);
        uint64_t a3 = pw == 3u ? (w3 & pm) : w3;
        uint32_t pos = 0xffffffffu;
        if (a0) pos = W::ctz64(a0);
        else if (a1) pos = 64u + W::ctz64(a1);
        else if (a2) pos = 128u + W::ctz64(a2);
        else if (a3) pos = 192u + W::ctz64(a3);
        if (pos < rl) out = pos - p;
        return out;
    }
    CBC_MFN void set(uint32_t k)
    {
        /* selects, not an if-chain: an if-chain over adjacent members is turned back into a
         * runtime-indexed access by the optimiser, which forces the window into scratch memory */
        uint64_t bit = 1ull << (k & 63u); uint32_t kw = k >> 6;
        w0 |= (kw == 0u) ? bit : 0ull;
        w1 |= (kw == 1u) ? bit : 0ull;
        w2 |= (kw == 2u) ? bit : 0ull;
        w3 |= (kw == 3u) ? bit : 0ull;
    }
#endif
};

template <class W, bool GEN = false>
struct CbcEnc {
    typedef typename W::V32 V32;
    typedef typename W::Mask Mask;

    /* ---- range coder + bit writer (Arithmetic_stream.c:155-194, 274-371) ---- */
    typedef typename W::Uv Uv;               /* a wave-uniform value kept in a VECTOR register (see cbc_wave_gpu.h) */
    Uv l, rng;                               /* lower bound, range = u - l + 1 (vector registers)   */
    Uv lm1;                                  /* l - 1, formed at the end of a step (beside the next step's multiplies) */
    Uv thr0;                                 /* step_known0(): the new range above which nothing shifts (a function of l) */
    CBC_MFN void set_l_forms()
    {
        lm1 = W::uv_opaque(l - 1u);
        thr0 = ((l & (1u << 24)) | (1u << 25)) - l;
    }
    uint32_t scale3;                         /* E3 count pending after the steps packed so far      */
    uint32_t bitpos, flushed;               /* bits produced; words already stored (multiple of 64)  */
    uint32_t *ring;                         /* CBC_RING_WORDS of LDS, zero except for the pending bits */
    uint32_t *out32; uint32_t cap_words;
    uint32_t status, nsym, fail_read, cur_read;
    V32 q_lo, q_cnt, q_n; uint32_t q_len;   /* pending symbols: lane k = k-th queued (lo, cnt, n)      */
    uint32_t role, batch_i; uint32_t *batch, *ctl;  /* CBC_ROLE_*; hand-off ring and its two counters in LDS */
    /* coder wave: the batch taken from the model wave (lane k = its k-th symbol) and the read cursor */
    V32 b_lo, b_hi, b_n, b_fl, b_fh; uint32_t b_len, b_pos, b_stop, b_flags, seen_last; uint64_t b_neq;
    V32 rec_a, rec_s; uint32_t rec_n;       /* output of the coder steps not packed yet (see pack())   */
#ifdef CBC_STAMP
    unsigned long long t_last, t_sum[16];
#endif

    /* ---- models ---- */
    V32 small;                              /* match / same_ref / chars lane table            */
    V32 fkey, fexc; uint32_t fcount;        /* flag: sparse, one entry per lane                */
    V32 hkey, hexc; uint32_t hc0, hc1, hc2, hc3, hn0, hn1, hn2, hn3;   /* codebook ctx 0..3: sparse, 8 lanes each */
    uint32_t *rlen_exc, *snps_exc, *indels_exc, *rname_key, *rname_exc, *pos_val, *pos_occ, *pos_pre, *pos_idx, *var_ev, *bloom, *p0ev;
    uint32_t snps_n, indels_n;
    uint32_t rn_count, rn_cap;               /* contig-name pairs in use / capacity (CBC_CAP_NAME in the block kernels) */
    uint32_t *vtab;                          /* GEN: var excess table in global memory, row = context, L0 words per row */
    /* GEN (whole-file stream): what no LDS budget bounds lives in global memory, as the var table does --
     *   flag: values beyond the CBC_CAP_FLAG register pairs, unsorted (value, excess) pairs (sam_models.c:96-130 has 65536);
     *   pos:  alphabet entries beyond the pos_lds_cap held in LDS (MAX_CARDINALITY, sam_block.h:55);
     *   pos_alpha: the four byte models as dense excess tables with their rescale (the derived form of pos_alpha() stops
     *   at ~104 k registered values, where those models first rescale).  nullptr / cap_pos in the other kernels. */
    uint32_t *fsp_key, *fsp_exc; uint32_t fsp_count;
    uint32_t *pos_ov_val, *pos_ov_occ; uint32_t pos_lds_cap;      /* pos_lds_cap: a multiple of 64 (chunks never straddle) */
    uint32_t *palpha; uint32_t pa_n0, pa_n1, pa_n2, pa_n3;
    uint32_t pos_card, cap_pos;              /* pos alphabet: value / occurrences / prefix by index, in LDS */
    uint32_t nev, nev1, cap_var;             /* var events of strand 0 (from the bottom of the area) / strand 1 (from the top) */
    V32 p0cnt; uint32_t p0over;              /* p = 0 contexts: events held per (strand, d & 7) bucket, lane = strand * 8 + bucket; bit = that bucket spilled to the global list */
    uint32_t L0;

    /* ---- cross-read state (T7/T8 of SURVEY.md) ---- */
    uint32_t prevPos, prevM, prevChar, win_pos;
    CbcWin<W> win;                           /* snpInRef[cumsumP-1 .. +255]                       */

    /* ======================================================================================= */
    CBC_MFN void fail(uint32_t st) { if (status == CBC_ST_OK) { status = st; fail_read = cur_read; } }

    /* ---- bit writer (stream_write_bits / stream_write_bit, io_functions.c; MSB first).  The stream is
     * assembled in a ring of LDS words: a piece of <= 32 bits at bit offset `off` is OR-ed into one or two
     * words, so the pieces of all symbols of a batch are placed at once, one lane each, at offsets from
     * a prefix sum of their lengths.  Complete runs of 64 words leave as one coalesced 256-byte store. ---- */
    CBC_MFN void place(const V32 &val, const V32 &len, const V32 &off, const Mask &m)
    {
        const V32 aligned = val << (W::splat(32u) - len);             /* len 1..32: shift 31..0 */
        const V32 sh = off & 31u, w = (off >> 5) & (CBC_RING_WORDS - 1u);
        const Mask on = m & (len != 0u);
        W::lds_or(ring, w, aligned >> sh, on);
        W::lds_or(ring, (w + 1u) & (CBC_RING_WORDS - 1u), aligned << (W::splat(32u) - sh), on & (sh != 0u));
    }
    CBC_MFN void flush_full()
    {
        const V32 ln = W::lane();
        while ((bitpos >> 5) - flushed >= 64u) {
            const V32 idx = (ln + flushed) & (CBC_RING_WORDS - 1u);
            const V32 wv = W::load32(ring, idx, W::all(), 0u);
            if (flushed + 64u <= cap_words) W::store32(out32, ln + flushed, W::bswap_v(wv), W::all());
            else fail(CBC_ST_OUT_FULL);
            W::store32(ring, idx, W::splat(0u), W::all());
            flushed += 64u;
        }
    }
    CBC_MFN void put(uint32_t v, uint32_t n)          /* n <= 32 bits, one piece, lane 0 places it */
    {
        if (n == 0) return;
        place(W::splat(v), W::splat(n), W::splat(bitpos), W::lane() == 0u);
        bitpos += n;
        flush_full();
    }
    CBC_MFN void put_run(uint32_t bit, uint32_t n)    /* n copies of bit */
    {
        uint32_t pat = bit ? 0xffffffffu : 0u;
        while (n >= 32u) { put(pat, 32u); n -= 32u; }
        if (n) put(pat >> (32u - n), n);
    }
    /* the output of up to 64 coder steps: lane k holds step k's lower bound before its shifts with k1 on top
     * (l | k1 << 26) and its E3 count k3.  The k1 leading bits of l leave through E1/E2 as: the first
     * bit, then as many inverted copies as E3 shifts were pending, then the other k1 - 1 bits
     * (Arithmetic_stream.c:296-341).  "Pending" for step i = the k3 of all steps since the last one that
     * emitted (that one included): a segmented sum, here S_ex(i) - S_ex(last emitter before i) with the
     * last emitter found by a prefix maximum (S_ex is non-decreasing), plus the count carried in from
     * the previous batch when no step of this batch has emitted yet. */
    CBC_MFN void pack(const V32 &rec_a, const V32 &rec_k3, uint32_t m)
    {
#ifdef CBC_ABLATE_PACK            /* timing experiments only (tools/README.md): the stream is wrong */
        bitpos += m; return;
#endif
        const V32 ln = W::lane();
        const Mask in = ln < m;
        const V32 k1 = rec_a >> 26, bits = (rec_a & CBC_M26) >> ((W::splat(26u) - k1) & 31u);   /* k1 = 0: l >> 26 = 0 */
        const Mask act = in & (k1 != 0u);
        const V32 k3 = W::select(in, rec_k3, W::splat(0u));
        const V32 s_in = W::scan_incl_add(k3), s_ex = s_in - k3;
        const V32 mark = W::select(act, s_ex + 1u, W::splat(0u));           /* emitters carry S_ex + 1, others 0 */
        const V32 m_in = W::scan_incl_max(mark);
        const V32 m_ex = W::shift_up1(m_in, 0u);                             /* last emitter strictly before this lane */
        const V32 rec_s = W::select(m_ex == 0u, s_ex + scale3, s_ex - (m_ex - 1u));
        {
            const uint32_t total = W::readlane(s_in, 63u), last = W::readlane(m_in, 63u);
            scale3 = last ? total - (last - 1u) : scale3 + total;           /* pending after this batch */
        }
        const V32 len = W::select(act, k1 + rec_s, W::splat(0u));
#ifndef CBC_PACK_SERIAL_AT
#define CBC_PACK_SERIAL_AT 32u                           /* tests lower it to exercise the piece-by-piece path */
#endif
        if (W::ballot(act & (len > CBC_PACK_SERIAL_AT))) {    /* a long E3 run: piece by piece */
            for (uint32_t k = 0; k < m; k++) {
                const uint32_t kk = W::readlane(k1, k), bb = W::readlane(bits, k), sc = W::readlane(rec_s, k);
                if (kk == 0u) continue;
                if (sc == 0u) put(bb, kk);
                else {
                    const uint32_t b0 = bb >> (kk - 1u);
                    put(b0, 1u); put_run(b0 ^ 1u, sc);
                    put(bb & ((1u << (kk - 1u)) - 1u), kk - 1u);
                }
            }
            return;
        }
        const V32 incl = W::scan_incl_add(len);
        const V32 km1 = (k1 - 1u) & 31u;                      /* inactive lanes: any in-range shift */
        const V32 b0 = bits >> km1;
        const V32 rest = bits & ((W::splat(1u) << km1) - 1u);
        const V32 run = W::select(b0 != 0u, W::splat(0u), (W::splat(1u) << rec_s) - 1u);
        const V32 val = (b0 << ((rec_s + km1) & 31u)) | (run << km1) | rest;
        place(val, len, incl - len + bitpos, act);
        bitpos += W::readlane(incl, 63u);
        flush_full();
    }

    /* arithmetic_encoder_step, Arithmetic_stream.c:274-345, with the E1/E2 and E3 loops in
     * closed form: within one step all E1/E2 iterations come first (they strip the common
     * leading bits of l and u), then all E3 iterations (they strip the run of positions below
     * the MSB where l has 1 and u has 0); E3 leaves msb(l)=0, msb(u)=1, so E1/E2 cannot recur. */
    /* Symbol queue.  The models do not depend on the coder, so they push their (lo, cnt, n) triples
     * into three VGPRs (lane k = k-th pending symbol) and the coder step is instantiated only at the
     * few drain() sites instead of at every model call site: a ~6x smaller kernel (it has to live in
     * the instruction cache next to other waves' working sets) and the coder state is not live across
     * the model code. */
    CBC_MFN void encode(uint32_t lo, uint32_t cnt, uint32_t n)
    {
#ifdef CBC_EMU_TRACE                     /* tests/emu debugging aid: log every symbol as the models emit it */
        cbc_emu_trace(cur_read, lo, cnt, n);
#endif
        V32 ln = W::lane();
        Mask here = ln == q_len;
        q_lo = W::select(here, W::splat(lo), q_lo);
        q_cnt = W::select(here, W::splat(cnt), q_cnt);
        q_n = W::select(here, W::splat(n), q_n);
        q_len++;
    }
    /* ---- hand-off between the two wavefronts ------------------------------------------------------
     * The coder wave owns the per-record ("fixed") models and codes their symbols straight from
     * fixed_group()'s lanes; the model wave owns the edit models and produces everything else as
     * SEGMENTS of the symbol stream, each closed by an END entry: the stream header, the contig name of
     * record 0, the edits of one imperfect record, the end-of-stream sentinel.  Segments travel in
     * batches of <= 64 entries through a ring of CBC_BATCH_SLOTS LDS buffers guarded by two counters in
     * LDS, `produced` (written by the model wave only) and `consumed` (written by the coder wave only):
     * the model wave fills slot produced % SLOTS once produced - consumed < SLOTS and then bumps
     * `produced` with release order; the coder waits for produced > consumed, copies the slot into
     * registers and bumps `consumed`.  The ring lets the model wave run several batches ahead, which
     * absorbs the burstiness of the two sides (a run of perfect records costs the coder time and the
     * model wave none; an imperfect record the reverse).  Per group of 64 records the model wave also
     * sends one empty batch flagged GROUP that carries the group's match-test mask.  The model wave
     * always ends with a batch flagged LAST (which carries its status) and never waits after it;
     * whatever happens, the coder takes batches until it has seen LAST, so both waves run to their end. */
#define CBC_FRAC(c, n) W::frac32(c, n)
#define CBC_BF_LAST  1u
#define CBC_BF_GROUP 2u
#define CBC_END_N    0xffffffffu                     /* n of an END entry */
    CBC_MFN void drain() { if (role == CBC_ROLE_MODEL) publish(0u, 0ull); else drain_q(); }
    CBC_MFN void seg_end()
    {
        if (role != CBC_ROLE_MODEL) { drain_q(); return; }
        if (q_len >= 64u) publish(0u, 0ull);
        V32 ln = W::lane();
        Mask here = ln == q_len;
        q_lo = W::select(here, W::splat(0u), q_lo); q_cnt = W::select(here, W::splat(1u), q_cnt);
        q_n = W::select(here, W::splat(CBC_END_N), q_n);
        q_len++;
    }
    /* seg_end() of the model wavefront where the caller knows the END entry fits (edits() drains at 56 entries and adds at
     * most two per check): no hand-off code at this call site */
    CBC_MFN void seg_end_fits()
    {
        W::expect_eq(q_len < 64u ? 1u : 0u, 1u, "seg_end_fits: queue full");
        V32 ln = W::lane();
        Mask here = ln == q_len;
        q_lo = W::select(here, W::splat(0u), q_lo); q_cnt = W::select(here, W::splat(1u), q_cnt);
        q_n = W::select(here, W::splat(CBC_END_N), q_n);
        q_len++;
    }
    CBC_MFN void publish(uint32_t flags, uint64_t neq)
    {
        V32 ln = W::lane();
        CBC_TSM(5);
        while (batch_i - W::ctl_load(ctl + 1) >= CBC_BATCH_SLOTS) W::nap();      /* a free slot */
        CBC_TSM(6);                                          /* model wave: waiting for the coder */
        uint32_t *buf = batch + (batch_i & (CBC_BATCH_SLOTS - 1u)) * CBC_BATCH_WORDS;
        Mask m = ln < q_len;
        W::store32(buf, ln, q_lo, m); W::store32(buf + 64u, ln, q_cnt, m); W::store32(buf + 128u, ln, q_n, m);
        V32 hdr = W::select(ln == 0u, W::splat(q_len), W::select(ln == 1u, W::splat(flags),
                  W::select(ln == 2u, W::splat(status), W::select(ln == 3u, W::splat(status == CBC_ST_OK ? cur_read : fail_read),
                  W::select(ln == 4u, W::splat((uint32_t)neq), W::splat((uint32_t)(neq >> 32)))))));
        W::store32(buf + 192u, ln, hdr, ln < 6u);
        batch_i++;
        W::ctl_store(ctl, batch_i);                          /* release: the slot's words are in LDS before the count */
        q_len = 0;
    }
    /* one symbol through the coder; the step's output waits in rec_a / rec_s for pack() */
    CBC_MFN void step(uint32_t lo, uint32_t hi, uint32_t n, uint32_t flo, uint32_t fhi)
    {
        Uv rec, k3;
        code1(lo, hi, n, flo, fhi, rec, k3);
        W::set_lane_uv(rec_a, rec_n, rec);
        W::set_lane_uv(rec_s, rec_n, k3);
        nsym++;
        if (++rec_n >= CBC_REC_PACK_AT) { pack(rec_a, rec_s, rec_n); rec_n = 0; }    /* leaves room for a record's 8 fixed symbols */
    }
    CBC_MFN void flush_recs() { if (rec_n) { pack(rec_a, rec_s, rec_n); rec_n = 0; } }
    /* The per-record fixed symbols go through step_fixed(): the same step without the "lanes full -> pack" test, so that
     * pack() is instantiated at a handful of sites and not at every fixed symbol -- the kernel's code shares an
     * instruction cache with the other wavefronts of two CUs.  Invariant: step() packs once CBC_REC_PACK_AT = 56 steps
     * are recorded, room(8) at the top of a record packs unless more than 8 lanes are free, so the at most 8 step_fixed()
     * calls of a record (with or without drained symbols in between) always find a lane. */
    CBC_MFN void room(uint32_t k) { if (rec_n + k >= 64u) flush_recs(); }      /* a later step() must still find lane <= 63 free */
    template <class FLO, class FHI, class FN>
    CBC_MFN void step_fixed(FLO lo_of, FHI hi_of, FN n_of, uint32_t flo, uint32_t fhi)
    {
        Uv rec, k3;
        code1_lazy(flo, fhi, lo_of, hi_of, n_of, rec, k3);
#ifdef CBC_ABLATE_SETLANE         /* timing experiments only */
        rec_a ^= rec; rec_s ^= k3;
#else
        W::set_lane_uv(rec_a, rec_n, rec);
        W::set_lane_uv(rec_s, rec_n, k3);
#endif
        nsym++; rec_n++;
    }
    /* the symbols queued by this wavefront itself (fused form; in the coder wave: its own few queued ones).
     * The two divisions of a step, floor(range * c / n) for c = cum and c = cum + count, need no divide on
     * the serial path: f = floor(c * 2^32 / n) is computed for all queued symbols at once (one lane
     * each), and code1() finishes with one multiply-high and a remainder test. */
    CBC_MFN void drain_q()
    {
        V32 ln = W::lane();
        uint32_t m = W::uni(q_len);
        /* assert(cumCountX_1 < cumCountX) of every pending symbol at once (stream_model.c:71) */
        const uint64_t bad = W::ballot((ln < m) & ((q_cnt == 0u) | (q_n == 0u)));
        if (bad) m = W::ctz64(bad);
        const V32 q_hi = q_lo + q_cnt;
        const V32 f_lo = CBC_FRAC(q_lo, q_n), f_hi = CBC_FRAC(q_hi, q_n);
        for (uint32_t k = 0; k < m; k++)
            step(W::readlane(q_lo, k), W::readlane(q_hi, k), W::readlane(q_n, k), W::readlane(f_lo, k), W::readlane(f_hi, k));
        if (bad) fail(CBC_ST_ASSERT);
        q_len = 0;
    }
    /* coder wave: take the next batch */
    CBC_MFN void pull()
    {
        V32 ln = W::lane();
        CBC_TSM(9);                                           /* coder wave: coding */
        while (W::ctl_load(ctl) == batch_i) W::nap();         /* acquire: the next batch is in its slot */
        CBC_TSM(10);                                          /* coder wave: waiting for a batch */
        const uint32_t *buf = batch + (batch_i & (CBC_BATCH_SLOTS - 1u)) * CBC_BATCH_WORDS;
        V32 hdr = W::load32(buf + 192u, ln, ln < 6u, 0u);
        const uint32_t len = W::readlane(hdr, 0u);
        const uint32_t pst = W::readlane(hdr, 2u), prec = W::readlane(hdr, 3u);
        b_flags = W::readlane(hdr, 1u);
        b_neq = (uint64_t)W::readlane(hdr, 4u) | ((uint64_t)W::readlane(hdr, 5u) << 32);
        Mask m = ln < len;
        b_lo = W::load32(buf, ln, m, 0u); b_n = W::load32(buf + 128u, ln, m, 1u);
        const V32 cnt = W::load32(buf + 64u, ln, m, 1u);
        b_hi = b_lo + cnt;
        b_fl = CBC_FRAC(b_lo, b_n); b_fh = CBC_FRAC(b_hi, b_n);
        b_len = len > 64u ? 64u : len; b_pos = 0;
        const uint64_t bad = W::ballot(m & ((cnt == 0u) | (b_n == 0u)));
        b_stop = bad ? W::ctz64(bad) : 64u;
        batch_i++;
        W::ctl_store(ctl + 1, batch_i);                       /* release: the slot has been copied to registers */
        if (b_flags & CBC_BF_LAST) seen_last = 1u;
        if (pst != CBC_ST_OK && status == CBC_ST_OK) { status = pst; fail_read = prec; }
    }
    /* coder wave: code the model wave's symbols up to the next END */
    CBC_MFN void seg_consume()
    {
        /* the entries of the current batch up to its END (found by one ballot) are a counted loop with no way out of
         * its middle; what ends the segment, asks for the next batch or fails is decided between such runs */
        const V32 ln = W::lane();
        uint32_t done = 0u;                                  /* a word, not a bool: see var_code */
        while ((done | status) == 0u) {
            if (b_pos == b_len) {
                if (seen_last) fail(CBC_ST_ASSERT);                            /* stream ended inside a segment */
                else { pull(); if (b_flags & CBC_BF_GROUP) fail(CBC_ST_ASSERT); }   /* protocol: group mark inside a segment */
            } else {
                const uint64_t em = W::ballot((b_n == CBC_END_N) & (ln >= b_pos) & (ln < b_len));
                uint32_t kend = em ? W::ctz64(em) : b_len;
                const uint32_t bad = b_stop < kend ? 1u : 0u;                  /* zero count / total: stream_model.c:71 */
                if (bad) kend = b_stop;
                for (uint32_t k = b_pos; k < kend; k++) {            /* cum / cum + count / total only in the rare remainder branch */
                    Uv rec, k3;
                    code1_lazy(W::readlane(b_fl, k), W::readlane(b_fh, k), [&]() -> uint32_t { return W::readlane(b_lo, k); },
                               [&]() -> uint32_t { return W::readlane(b_hi, k); }, [&]() -> uint32_t { return W::readlane(b_n, k); }, rec, k3);
                    W::set_lane_uv(rec_a, rec_n, rec);
                    W::set_lane_uv(rec_s, rec_n, k3);
                    nsym++;
                    if (++rec_n >= CBC_REC_PACK_AT) { pack(rec_a, rec_s, rec_n); rec_n = 0; }
                }
                if (bad) fail(CBC_ST_ASSERT);
                else if (em) { b_pos = kend + 1u; done = 1u; }
                else b_pos = kend;
            }
        }
    }
    /* coder wave as a pure consumer (the long-read encoder): every batch, every entry, until the batch flagged LAST.  A
     * failure on this side (full output area) or the model's (it travels in the batch header) ends the coding, not the
     * pulling: the model wavefront must always find a free slot for its LAST batch. */
    CBC_MFN void consume_all()
    {
        while (!seen_last) {
            pull();
            if (status != CBC_ST_OK) continue;
            const uint32_t kend = b_stop < b_len ? b_stop : b_len;
            for (uint32_t k = 0; k < kend; k++) {
                Uv rec, k3;
                code1_lazy(W::readlane(b_fl, k), W::readlane(b_fh, k), [&]() -> uint32_t { return W::readlane(b_lo, k); },
                           [&]() -> uint32_t { return W::readlane(b_hi, k); }, [&]() -> uint32_t { return W::readlane(b_n, k); }, rec, k3);
                W::set_lane_uv(rec_a, rec_n, rec);
                W::set_lane_uv(rec_s, rec_n, k3);
                nsym++;
                if (++rec_n >= CBC_REC_PACK_AT) { pack(rec_a, rec_s, rec_n); rec_n = 0; }
            }
            if (b_stop < b_len) fail(CBC_ST_ASSERT);                          /* zero count / total: stream_model.c:71 */
        }
    }
    /* ---- the match test runs in the CODER wavefront (it has the idle time at cfg2: stamps in profiles/r02_final_stamps.log)
     * one group ahead of its coding, and reaches the model wavefront through a mailbox in LDS: ctl[2] = groups posted
     * (release / acquire like the batch counters), ctl[3] = 1 once the coder refused a group, ctl[4..7] = two slots of
     * (mask lo, mask hi).  Two slots are enough: the mask of group g + 2 is posted when the coder begins group g + 1, i.e.
     * after it has coded everything the model wavefront produced for group g, which the model wavefront did after reading
     * mask g.  No wait cycle: the model wavefront hands over all pending symbols before it waits for a mask, and the coder
     * posts mask g + 1 before it waits for anything of group g. */
    CBC_MFN void post_group(uint32_t g, uint64_t neq, uint32_t refused)
    {
        W::write_uni(ctl, 4u + 2u * (g & 1u), (uint32_t)neq);
        W::write_uni(ctl, 5u + 2u * (g & 1u), (uint32_t)(neq >> 32));
        if (refused) W::write_uni(ctl, 3u, 1u);
        W::ctl_store(ctl + 2, g + 1u);
    }
    /* the coder wavefront stops coding (a cap, a full output area, a refused record ...): no more masks will come, and the
     * model wavefront must not wait for one -- it has to reach its LAST batch, which this wavefront waits for in pull_rest() */
    CBC_MFN void abort_groups()
    {
        W::write_uni(ctl, 3u, 1u);
        W::ctl_store(ctl + 2, 0xffffffffu);
    }
    CBC_MFN bool wait_group(uint32_t g, uint64_t &neq)       /* model wavefront; false: the coder refused the group */
    {
        CBC_TSM(5);
        while (W::ctl_load(ctl + 2) < g + 1u) W::nap();
        CBC_TSM(6);
        neq = (uint64_t)W::read_uni(ctl, 4u + 2u * (g & 1u)) | ((uint64_t)W::read_uni(ctl, 5u + 2u * (g & 1u)) << 32);
        return W::read_uni(ctl, 3u) == 0u;
    }
    /* coder wave: the next group's match mask (an empty batch flagged GROUP) */
    CBC_MFN uint64_t pull_group()
    {
        if (status != CBC_ST_OK) return 0ull;
        if (b_pos != b_len || seen_last) { fail(CBC_ST_ASSERT); return 0ull; }
        pull();
        if (status == CBC_ST_OK && !(b_flags & CBC_BF_GROUP)) fail(CBC_ST_ASSERT);
        return b_neq;
    }
    /* coder wave: before leaving, take every batch the model wave still sends */
    CBC_MFN void pull_rest() { while (!seen_last) pull(); }
    /* one coder step without its output: the range update (Arithmetic_stream.c:274-295) and the E1/E2
     * and E3 loops (:296-341) in closed form -- within one step all E1/E2 iterations come first (they
     * strip the common leading bits of l and u), then all E3 iterations (the run of positions below the
     * MSB where l has 1 and u has 0); E3 leaves msb(l) = 0, msb(u) = 1, so E1/E2 cannot recur.
     * The step records (l | k1 << 26, k3); what leaves the coder is assembled by pack().
     * The recurrence runs on the VECTOR unit although every lane holds the same value: a CU has one
     * scalar unit for all its wavefronts and four vector units, and with ten blocks resident per CU the
     * scalar unit is what the kernel queues on. */
    CBC_MFN void code1(uint32_t lo, uint32_t hi, uint32_t n, uint32_t flo, uint32_t fhi, Uv &rec, Uv &k3)
    {
        code1_lazy(flo, fhi, [&]() { return lo; }, [&]() { return hi; }, [&]() { return n; }, rec, k3);
    }
    /* the same step with cum / cum + count / total fetched only where they are needed: the remainder test that one step
     * in 64 takes (and the emulation's cross-check).  The fixed symbols keep their operands in lanes of vector registers
     * (one lane per record), so every operand costs a v_readlane: two per step on the usual path instead of five. */
    template <class FLO, class FHI, class FN>
    CBC_MFN void code1_lazy(uint32_t flo, uint32_t fhi, FLO lo_of, FHI hi_of, FN n_of, Uv &rec, Uv &k3)
    {
#ifdef CBC_ABLATE_CODER          /* timing experiments only: keeps the operands live, skips the coder */
        l ^= flo; rng ^= fhi; rec = W::uv(0u); k3 = rec; return;
#endif
        const Uv range = rng;
        /* floor(range * c / n) for c = lo and c = hi, given f = floor(c * 2^32 / n) (clamped to 2^32 - 1
         * when c == n).  range * f = q * 2^32 + t: the true quotient is q + (t + range * g / n) / 2^32
         * with g < n the remainder of f's own division, and range <= 2^26, so q is final unless
         * t >= 2^32 - 2^26 (one step in 64 for a random t, and always when cum + count = n).  Only then
         * is the remainder formed (it is < 2n < 2^22, so its low 32 bits are all of it): one 32x32->64
         * multiply per division on the usual path instead of three quarter-rate ones. */
        Uv ql, qh, tl, th;
        W::mul64(range, flo, ql, tl); W::mul64(range, fhi, qh, th);
        if (W::uv_ge(tl | th, 0xfc000000u)) {
            const uint32_t lo = lo_of(), hi = hi_of(), n = n_of();
            ql += (range * lo - ql * n >= n) ? 1u : 0u;
            qh += (range * hi - qh * n >= n) ? 1u : 0u;
        }
#ifndef __HIP_DEVICE_COMPILE__     /* emulation: every quotient against integer division */
        W::expect_eq(W::uv_scalar(ql), (uint32_t)((uint64_t)W::uv_scalar(range) * lo_of() / n_of()), "scaled_div(cum)");
        W::expect_eq(W::uv_scalar(qh), (uint32_t)((uint64_t)W::uv_scalar(range) * hi_of() / n_of()), "scaled_div(cum + count)");
#endif
        code_tail(ql, qh, rec, k3);
    }
    /* the step once its two quotients are known */
    CBC_MFN void code_tail(Uv ql, Uv qh, Uv &rec, Uv &k3)
    {
        const Uv u = lm1 + qh;                                   /* the state is (l, range): every E1/E2/E3 shift doubles the range */
        l = l + ql;
        /* Branch-free, in 32-bit arithmetic, and with the two shifts merged.  E1/E2 shift k1 = number of common
         * leading bits (of 26); E3 then shifts out the run, below the new MSB, where l has 1 and u has 0.  After E1/E2
         * the low k1 bits of l are 0 and of u are 1, so that run can be read from the unshifted values:
         * y = ((l & ~u) << 7) << k1 (32-bit shifts drop exactly the bits the masks would), k3 = its leading ones,
         * k1 + k3 <= 26, and
         *     l' = (l << (k1 + k3)) & M25,   u' = ((u << (k1 + k3)) & M25) | 2^25 | (2^(k1 + k3) - 1)
         * equal the two updates of Arithmetic_stream.c:296-341 applied one after the other (k1 = 0 or
         * k3 = 0 make the respective part the identity: bit 25 of l is then 0 and of u is 1).  Each of
         * those shifts maps [l, u] with slope 2 (u' - l' + 1 = 2 (u - l + 1)), so instead of u' the
         * state keeps range' = (qh - ql) << (k1 + k3), which is also what the next step starts from.
         * The pending-E3 bookkeeping is not done here: pack() derives it from the recorded k3 of all steps.
         * What does not need k1 is formed beside the k1 chain (the wavefront issues in order and a dependent
         * instruction waits ~8 cycles for its operand): l - 1 at the end of the previous step, and the complement of
         * (l & ~u) << 7 -- its low 7 bits are set, so shifted by k1 <= 26 it stays non-zero and its leading zeros
         * are y's leading ones.  (Shifting l and the range in two stages, by k1 and then by k3, shortens the chain by
         * one more instruction but adds one, and measures slower: profiles/r02_ab_kernels.log run 7.) */
        const Uv x = l ^ u;
        const Uv nya = ((~l | u) << 7) | 127u;
        const Uv k1 = W::clz_uv((x << 6) | 32u);                 /* leading zeros of the 26-bit x; 26 when x = 0 */
        rec = l | (k1 << 26);                                    /* pack() takes the k1 leading bits of l from here */
        k3 = W::clz_uv(nya << k1);
        const Uv sh = k1 + k3;
        l = (l << sh) & CBC_M25;
        rng = (qh - ql) << sh;
        set_l_forms();
#ifndef __HIP_DEVICE_COMPILE__     /* emulation: the explicit upper bound against the range form */
        W::expect_eq(W::uv_scalar(l + rng - 1u), W::uv_scalar(((u << sh) & CBC_M25) | (1u << 25) | ((1u << sh) - 1u)), "range form of the upper bound");
#endif
    }
    /* A fixed symbol that is symbol 0 of its model (same_ref after record 0, rlength[1..3]): cum = 0, so the lower bound
     * stays and range' = floor(range * count0 / n).  These counts lie within 254 of their totals, the range shrinks by a
     * fraction of a percent and mostly nothing shifts: with l < 2^25 (true between steps: bit 25 of l is clear after any
     * step), E1/E2 need u' < 2^25 and E3 needs l >= 2^24 and u' < 3 * 2^24, so neither applies when
     *     range' > thr0 = (2^25 | (l & 2^24)) - l.
     * Such a step records nothing (pack() would find k1 = k3 = 0) and costs one multiply and two tests instead of the
     * whole recurrence; any other goes through the usual tail. */
    template <class FHI, class FN>
    CBC_MFN void step_known0(FHI hi_of, FN n_of, uint32_t fhi)
    {
#ifdef CBC_ABLATE_CODER
        rng ^= fhi; nsym++; return;
#endif
        Uv qh, th;
        W::mul64(rng, fhi, qh, th);
        if (W::uv_ge(th, 0xfc000000u)) {
            const uint32_t hi = hi_of(), n = n_of();
            qh += (rng * hi - qh * n >= n) ? 1u : 0u;
        }
#ifndef __HIP_DEVICE_COMPILE__
        W::expect_eq(W::uv_scalar(qh), (uint32_t)((uint64_t)W::uv_scalar(rng) * hi_of() / n_of()), "scaled_div(count0)");
#endif
        nsym++;
        if (W::uv_gt(qh, thr0)) {
#ifndef __HIP_DEVICE_COMPILE__     /* emulation: the recurrence agrees that nothing shifts */
            const uint32_t lf = W::uv_scalar(l), uf = lf + W::uv_scalar(qh) - 1u;
            W::expect_eq(((lf ^ uf) >> 25) & 1u, 1u, "step_known0: E1/E2 would shift");
            W::expect_eq((lf >> 24) & (~uf >> 24) & 1u, 0u, "step_known0: E3 would shift");
#endif
            rng = qh;
            return;
        }
        Uv rec, k3;
        code_tail(W::uv(0u), qh, rec, k3);
        W::set_lane_uv(rec_a, rec_n, rec);
        W::set_lane_uv(rec_s, rec_n, k3);
        rec_n++;
    }
    CBC_MFN uint32_t finish()                        /* encoder_last_step :348-363 + stream_finish_byte */
    {
        const uint32_t lf = W::uv_scalar(l);
        uint32_t msb = lf >> 25;
        put(msb, 1u); put_run(msb ^ 1u, scale3); scale3 = 0u;
        put(lf & CBC_M25, 25u);
        const uint32_t nbytes = (bitpos >> 3) + 1u;          /* +1: the partial byte, or the extra 0x00 */
        const uint32_t nw = (nbytes + 3u) >> 2;              /* the ring is zero past the last bit */
        const V32 ln = W::lane();
        if (nw > cap_words) { fail(CBC_ST_OUT_FULL); return nbytes; }
        while (flushed < nw) {
            const V32 idx = (ln + flushed) & (CBC_RING_WORDS - 1u);
            const V32 wv = W::load32(ring, idx, W::all(), 0u);
            W::store32(out32, ln + flushed, W::bswap_v(wv), (ln + flushed) < nw);
            flushed += 64u;
        }
        return nbytes;
    }

    /* ---- register-resident sparse model (flag, codebook): entries at lanes [base, base+count) ---- */
    CBC_MFN void regsparse_code(V32 &key, V32 &exc, uint32_t base, uint32_t cap, uint32_t &count, uint32_t &n,
                               uint32_t card, uint32_t step, uint32_t x, uint32_t cap_status)
    {
        V32 ln = W::lane();
        Mask live = (ln >= base) & (ln < base + count);
        uint32_t lo = x + W::reduce_add(W::select(live & (key < x), exc, W::splat(0u)));
        uint64_t eq = W::ballot(live & (key == x));
        uint32_t idx = 0, cnt = 1u;
        if (eq) { idx = W::ctz64(eq); cnt = 1u + W::readlane(exc, idx); }
        if (x >= card) { fail(CBC_ST_ASSERT); return; }
        encode(lo, cnt, n);
        if (eq) exc = W::select(ln == idx, exc + step, exc);
        else {
            if (count >= cap) { fail(cap_status); return; }
            idx = base + count;
            key = W::select(ln == idx, W::splat(x), key);
            exc = W::select(ln == idx, W::splat(step), exc);
            count++;
        }
        n += step;
        if (n >= CBC_RESCALE) {                               /* update_model stream_model.c:41-48 */
            live = (ln >= base) & (ln < base + count);
            exc = W::select(live, (exc + 1u) >> 1, exc);
            n = card + W::reduce_add(W::select(live, exc, W::splat(0u)));
        }
    }

    /* ---- LDS dense-excess model (rlength[0], snps, indels) ---- */
    CBC_MFN void dense_lookup(const uint32_t *exc, uint32_t x, uint32_t &lo, uint32_t &cnt)
    {
        V32 ln = W::lane();
        if (x < 64u) {                                       /* one load holds everything; the usual symbols (0..3: edit counts) need no wave sum */
            const V32 e = W::load32(exc, ln, ln <= x, 0u);
            cnt = 1u + W::readlane(e, x);
            if (x < 4u) { lo = x; for (uint32_t s = 0; s < x; s++) lo += W::readlane(e, s); }
            else lo = x + W::reduce_add(W::select(ln < x, e, W::splat(0u)));
            return;
        }
        V32 a = W::splat(0u);
        const uint32_t xb = W::uni(x);                       /* opaque loop bound: see CBC_LOOP note */
        for (uint32_t b = 0; b < xb; b += 64u) { V32 i = ln + b; a = a + W::load32(exc, i, i < x, 0u); }
        lo = x + W::reduce_add(a);
        cnt = 1u + W::read_uni(exc, x);
    }
    CBC_MFN void dense_rescale(uint32_t *exc, uint32_t card, uint32_t &n)   /* stream_model.c:41-48 on e = count-1 */
    {
        V32 ln = W::lane();
        V32 a = W::splat(0u);
        const uint32_t cb = W::uni(card);
        for (uint32_t b = 0; b < cb; b += 64u) {
            V32 i = ln + b; Mask m = i < card;
            V32 e = (W::load32(exc, i, m, 0u) + 1u) >> 1;
            W::store32(exc, i, e, m);
            a = a + W::select(m, e, W::splat(0u));
        }
        n = card + W::reduce_add(a);
    }
    /* dense_code() for a caller that has checked x < 64 and n + step < CBC_RESCALE (the SNP count of an ordinary read): no
     * table sweep, no rescale at this call site */
    CBC_MFN void dense_code_low(uint32_t *exc, uint32_t step, uint32_t x, uint32_t &n)
    {
        const V32 ln = W::lane();
        const V32 e = W::load32(exc, ln, ln <= x, 0u);
        const uint32_t cnt = 1u + W::readlane(e, x);
        uint32_t lo = x;
        if (x < 4u) { for (uint32_t s = 0; s < x; s++) lo += W::readlane(e, s); }
        else lo += W::reduce_add(W::select(ln < x, e, W::splat(0u)));
        encode(lo, cnt, n);
        W::write_uni(exc, x, cnt - 1u + step);
        n += step;
    }
    CBC_MFN void dense_code(uint32_t *exc, uint32_t card, uint32_t step, uint32_t x, uint32_t &n)
    {
        if (x >= card) { fail(CBC_ST_ASSERT); return; }       /* assert(x < alphabetCard) stream_model.c:62 */
        uint32_t lo, cnt;
        dense_lookup(exc, x, lo, cnt);
        encode(lo, cnt, n);
        W::write_uni(exc, x, cnt - 1u + step);                /* update_model with the count already at hand */
        n += step;
        if (n >= CBC_RESCALE) dense_rescale(exc, card, n);
    }

    /* ---- lane-table literal-count models (match, same_ref, chars); step/rescale literal ---- */
    CBC_MFN void small_code(uint32_t base, uint32_t card, uint32_t step, uint32_t x)
    {
        uint32_t lo = 0, n = 0, cnt = 0;
        for (uint32_t j = 0; j < card; j++) {
            uint32_t c = W::readlane(small, base + j);
            if (j < x) lo += c;
            if (j == x) cnt = c;
            n += c;
        }
        encode(lo, cnt, n);
        V32 ln = W::lane();
        small = W::select(ln == base + x, small + step, small);
        if (n + step >= CBC_RESCALE) {
            Mask m = (ln >= base) & (ln < base + card);
            small = W::select(m, (small >> 1) + 1u, small);
        }
    }

    /* ---- rname: 256 contexts x 256, sparse (ctx,char)->excess list in LDS (id_compression.c:39-65) ---- */
    CBC_MFN void rname_code(uint32_t ctx, uint32_t sym)
    {
        V32 ln = W::lane();
        V32 an = W::splat(0u), alo = W::splat(0u);
        uint32_t key = (ctx << 8) | sym, found = CBC_NOMEMO;
        const uint32_t rb = W::uni(rn_count);
        for (uint32_t b = 0; b < rb; b += 64u) {
            V32 i = ln + b; Mask m = i < rn_count;
            V32 k = W::load32(rname_key, i, m, 0xffffffffu);
            V32 e = W::load32(rname_exc, i, m, 0u);
            Mask inctx = m & ((k >> 8) == ctx);
            an = an + W::select(inctx, e, W::splat(0u));
            alo = alo + W::select(inctx & ((k & 0xffu) < sym), e, W::splat(0u));
            uint64_t eq = W::ballot(m & (k == key));
            if (eq) found = b + W::ctz64(eq);
        }
        uint32_t n = 256u + W::reduce_add(an);
        uint32_t lo = sym + W::reduce_add(alo);
        uint32_t cnt = 1u + (found != CBC_NOMEMO ? W::read_uni(rname_exc, found) : 0u);
        if (n + 10u >= CBC_RESCALE) { fail(CBC_ST_ASSERT); return; }   /* unreachable within the name cap */
        encode(lo, cnt, n);
        if (found != CBC_NOMEMO) W::write_uni(rname_exc, found, cnt - 1u + 10u);
        else {
            if (rn_count >= rn_cap) { fail(CBC_ST_CAP_NAME); return; }
            W::write_uni(rname_key, rn_count, key);
            W::write_uni(rname_exc, rn_count, 10u);
            rn_count++;
        }
    }

    /* ---- pos tables: entry i < pos_lds_cap in LDS, the rest in the global overflow arrays (one wavefront appends to and
     * re-reads them: the list accessors of the var events).  `b` = the 64-aligned base of the chunk `i` belongs to. ---- */
    CBC_MFN V32 ptab_ld(const uint32_t *t_lds, const uint32_t *t_ov, uint32_t b, V32 i, Mask m, uint32_t other)
    {
        if (b < pos_lds_cap) return W::load32(t_lds, i, m, other);
        return W::load32_list(t_ov, i - pos_lds_cap, m, other);
    }
    CBC_MFN void ptab_st(uint32_t *t_lds, uint32_t *t_ov, uint32_t b, V32 i, V32 v, Mask m)
    {
        if (b < pos_lds_cap) W::store32(t_lds, i, v, m);
        else W::store32_list(t_ov, i - pos_lds_cap, v, m);
    }
    CBC_MFN uint32_t ptab_rd(const uint32_t *t_lds, const uint32_t *t_ov, uint32_t idx)
    {
        if (idx < pos_lds_cap) return W::read_uni(t_lds, idx);
        W::list_fence();
        return W::readlane(W::load32_list(t_ov, W::splat(idx - pos_lds_cap), W::all(), 0u), 0u);
    }
    CBC_MFN void ptab_wr(uint32_t *t_lds, uint32_t *t_ov, uint32_t idx, uint32_t v)
    {
        if (idx < pos_lds_cap) W::write_uni(t_lds, idx, v);
        else W::append_list(t_ov, idx - pos_lds_cap, v);
    }
    /* compress_pos_alpha in its general form: four dense 256-symbol models, step 10 (sam_models.c:164-202) */
    CBC_MFN void pos_alpha_dense(uint32_t x)
    {
        dense_code(palpha, 256u, 10u, x >> 24, pa_n0);
        dense_code(palpha + 256u, 256u, 10u, (x >> 16) & 0xffu, pa_n1);
        dense_code(palpha + 512u, 256u, 10u, (x >> 8) & 0xffu, pa_n2);
        dense_code(palpha + 768u, 256u, 10u, x & 0xffu, pa_n3);
    }

    /* ---- flag in its general form (compress_flag read_compression.c:50-70; the whole-file stream): the first CBC_CAP_FLAG
     * distinct values as (value, excess) register pairs, any further ones as pairs in global memory; cum(x) = x + the excess
     * of every smaller value seen, whichever side holds it ---- */
    CBC_MFN void flag_gen_code(uint32_t x, uint32_t &n)
    {
        const V32 ln = W::lane();
        Mask live = ln < fcount;
        V32 acc = W::select(live & (fkey < x), fexc, W::splat(0u));
        const uint64_t eq = W::ballot(live & (fkey == x));
        uint32_t idx = 0, cnt = 1u, sp_idx = CBC_NOMEMO;
        if (eq) { idx = W::ctz64(eq); cnt = 1u + W::readlane(fexc, idx); }
        if (fsp_count) {
            W::list_fence();
            const uint32_t nb = W::uni(fsp_count);
            for (uint32_t b = 0; b < nb; b += 64u) {
                const V32 i = ln + b; const Mask m = i < fsp_count;
                const V32 k = W::load32_list(fsp_key, i, m, 0xffffffffu), e = W::load32_list(fsp_exc, i, m, 0u);
                acc = acc + W::select(m & (k < x), e, W::splat(0u));
                const uint64_t hit = W::ballot(m & (k == x));
                if (hit) { const uint32_t hl = W::ctz64(hit); sp_idx = b + hl; cnt = 1u + W::readlane(e, hl); }
            }
        }
        if (x >= 65536u) { fail(CBC_ST_ASSERT); return; }
        encode(x + W::reduce_add(acc), cnt, n);
        if (eq) fexc = W::select(ln == idx, fexc + 8u, fexc);
        else if (sp_idx != CBC_NOMEMO) W::append_list(fsp_exc, sp_idx, cnt - 1u + 8u);
        else if (fcount < CBC_CAP_FLAG) {
            fkey = W::select(ln == fcount, W::splat(x), fkey);
            fexc = W::select(ln == fcount, W::splat(8u), fexc);
            fcount++;
        } else {
            if (fsp_key == nullptr || fsp_count >= 65536u) { fail(CBC_ST_CAP_FLAG); return; }
            W::append_list(fsp_key, fsp_count, x); W::append_list(fsp_exc, fsp_count, 8u);
            fsp_count++;
        }
        n += 8u;
        if (n >= CBC_RESCALE) {                               /* update_model stream_model.c:41-48 on e = count - 1 */
            live = ln < fcount;
            fexc = W::select(live, (fexc + 1u) >> 1, fexc);
            V32 a = W::select(live, fexc, W::splat(0u));
            if (fsp_count) {
                W::list_fence();
                const uint32_t nb = W::uni(fsp_count);
                for (uint32_t b = 0; b < nb; b += 64u) {
                    const V32 i = ln + b; const Mask m = i < fsp_count;
                    const V32 e = (W::load32_list(fsp_exc, i, m, 0u) + 1u) >> 1;
                    W::store32_list(fsp_exc, i, e, m);
                    a = a + W::select(m, e, W::splat(0u));
                }
            }
            n = 65536u + W::reduce_add(a);
        }
    }

    /* ---- pos alphabet bytes: compress_pos_alpha (read_compression.c:75-108).  The four 256-symbol
     * models only ever see the bytes of the values already registered in the alphabet, so their state
     * is recomputed from it: count(b) = 1 + 10 * #{registered v : byte_k(v) == b}.  `card` is the
     * alphabet size when the escape was coded (entries 1 .. card-1 are registered). ---- */
    CBC_MFN void pos_alpha(uint32_t x, uint32_t card)
    {
        V32 ln = W::lane();
        uint32_t lt3 = 0, eq3 = 0, lt2 = 0, eq2 = 0, lt1 = 0, eq1 = 0, lt0 = 0, eq0 = 0;
        const uint32_t b3 = x >> 24, b2 = (x >> 16) & 0xffu, b1 = (x >> 8) & 0xffu, b0 = x & 0xffu;
        const uint32_t pc = W::uni(card);
        for (uint32_t b = 0; b < pc; b += 64u) {
            V32 i = ln + b; Mask m = (i != 0u) & (i < card);
            V32 v = W::load32(pos_val, i, m, 0u);
            V32 v3 = v >> 24, v2 = (v >> 16) & 0xffu, v1 = (v >> 8) & 0xffu, v0 = v & 0xffu;
            lt3 += W::popc64(W::ballot(m & (v3 < b3))); eq3 += W::popc64(W::ballot(m & (v3 == b3)));
            lt2 += W::popc64(W::ballot(m & (v2 < b2))); eq2 += W::popc64(W::ballot(m & (v2 == b2)));
            lt1 += W::popc64(W::ballot(m & (v1 < b1))); eq1 += W::popc64(W::ballot(m & (v1 == b1)));
            lt0 += W::popc64(W::ballot(m & (v0 < b0))); eq0 += W::popc64(W::ballot(m & (v0 == b0)));
        }
        const uint32_t n = 256u + 10u * (card - 1u);
        if (n + 10u >= CBC_RESCALE) { fail(CBC_ST_ASSERT); return; }
        encode(b3 + 10u * lt3, 1u + 10u * eq3, n);
        encode(b2 + 10u * lt2, 1u + 10u * eq2, n);
        encode(b1 + 10u * lt1, 1u + 10u * eq1, n);
        encode(b0 + 10u * lt0, 1u + 10u * eq0, n);
    }


    /* ---- pos model in its general form (compress_pos read_compression.c:113-159; used by the whole-file stream and
     * the long-read format): the alphabet in order of appearance in pos_val[], LITERAL counts in pos_occ[] (entry 0
     * = the escape, count 1), step 10, rescale at 2^20; the four pos_alpha byte models are derived from the
     * registered values (pos_alpha()).  pos_n = the model's total. ---- */
    CBC_MFN void pos_lit_update(uint32_t idx, uint32_t &pos_n)
    {
        const V32 ln = W::lane();
        ptab_wr(pos_occ, pos_ov_occ, idx, ptab_rd(pos_occ, pos_ov_occ, idx) + 10u);
        pos_n += 10u;
        if (pos_n >= CBC_RESCALE) {
            V32 a = W::splat(0u);
            const uint32_t cb = W::uni(pos_card);
            if (pos_card > pos_lds_cap) W::list_fence();
            for (uint32_t b = 0; b < cb; b += 64u) {
                V32 i = ln + b; Mask m = i < pos_card;
                V32 c = (ptab_ld(pos_occ, pos_ov_occ, b, i, m, 0u) >> 1) + 1u;
                ptab_st(pos_occ, pos_ov_occ, b, i, c, m);
                a = a + W::select(m, c, W::splat(0u));
            }
            pos_n = W::reduce_add(a);
        }
    }
    CBC_MFN void pos_lit_code(uint32_t x, uint32_t &pos_n)
    {
        const V32 ln = W::lane();
        uint32_t idx = 0;
        const uint32_t cb = W::uni(pos_card);
        if (pos_card > pos_lds_cap) W::list_fence();
        for (uint32_t b = 0; b < cb; b += 64u) {
            V32 i = ln + b;
            V32 v = ptab_ld(pos_val, pos_ov_val, b, i, (i != 0u) & (i < pos_card), 0xffffffffu);
            uint64_t hit = W::ballot(v == x);
            if (hit) { idx = b + W::ctz64(hit); break; }
        }
        if (idx) {
            V32 a = W::splat(0u);
            const uint32_t ib = W::uni(idx);
            for (uint32_t b = 0; b < ib; b += 64u) { V32 i = ln + b; a = a + ptab_ld(pos_occ, pos_ov_occ, b, i, i < idx, 0u); }
            encode(W::reduce_add(a), ptab_rd(pos_occ, pos_ov_occ, idx), pos_n);
            pos_lit_update(idx, pos_n);
            return;
        }
        if (pos_card >= cap_pos) { fail(CBC_ST_CAP_POS); return; }
        encode(0u, W::read_uni(pos_occ, 0u), pos_n);
        pos_lit_update(0u, pos_n);
        if (palpha) pos_alpha_dense(x);                         /* the four byte models: dense with their rescale (whole-file stream) */
        else pos_alpha(x, pos_card);                           /* ... or derived from the registered values */
        ptab_wr(pos_val, pos_ov_val, pos_card, x); ptab_wr(pos_occ, pos_ov_occ, pos_card, 0u);
        pos_card++;
        pos_lit_update(pos_card - 1u, pos_n);                  /* update_model(P, alphabetCard++) without a send (:153) */
    }

    /* ---- the fixed symbols of 64 records at once, one lane per record ------------------------------
     * A block never holds more than CBC_MAX_BLOCK_READS records, so none of the per-record models
     * (rlength, pos, flag, match, same_ref) reaches its 2^20 rescale point inside a block and each
     * is a pure COUNTING model: with occ(s) = how often s was coded before this record,
     *     count(s) = init(s) + step * occ(s),  cum(s) = sum over s' < s,  n = n0 + step * records.
     * "Before this record" splits into the carried tables (records of earlier groups) and the lower
     * lanes of this group, which ballots / v_mbcnt count without any serial step.  What stays
     * serial: one iteration per DISTINCT rlength / FLAG value and per NEW pos delta in the group.
     *   rlength[0]  compress_rlength  read_compression.c:29-33   (quirk Q1: low byte only)
     *   pos         compress_pos      read_compression.c:113-159 (alphabet in order of appearance)
     *   flag        compress_flag     read_compression.c:50-70
     *   match       compress_match    read_compression.c:204-228 (context: dx == 1, previous match)
     * Results: per-lane (cum, count[, n]); `st` = the status the serial code would have stopped with
     * at that record (CBC_ST_OK if none), `esc` = records whose pos delta is new (escape + 4 bytes). */
    struct Fixed {
        V32 rl_lo, rl_cnt, p_lo, p_cnt, p_card, fl_lo, fl_cnt, m_lo, m_cnt, m_n, st;
        uint64_t esc, bad;
    };
    CBC_MFN void fixed_group(Fixed &F, uint32_t c0, uint32_t cn, const V32 &r_pos, const V32 &r_fl, uint64_t neq)
    {
        const V32 ln = W::lane();
        const Mask live = ln < cn;
        const V32 zero = W::splat(0u);
        V32 st = W::splat((uint32_t)CBC_ST_OK);

        /* -- pos delta: dx = pos - prevPos + 1, 1 <= dx < MAX_ALPHA (sam_block.h:54) -- */
        const V32 prevp = W::shift_up1(r_pos, prevPos);
        const V32 dxv = r_pos - prevp + 1u;
        const Mask dxbad = live & ((((r_pos - prevp) >> 31) != 0u) | (dxv >= 5000000u));
        prevPos = W::readlane(r_pos, cn - 1u);

        /* -- flag (value-ordered sparse table, one entry per lane of fkey/fexc; step 8) -- */
        const V32 flagv = r_fl & 0xffffu;
        V32 fl_base = zero, fl_extra = zero, fl_cnt = zero;
        Mask capflag = W::lane_bit(0ull);
        V32 fexc_new = fexc;                                 /* lookups see the table as it was before the group */
        {
            uint64_t rem = W::ballot(live);
            while (rem) {
                const uint32_t v = W::readlane(flagv, W::ctz64(rem));
                const Mask mv = live & (flagv == v);
                const uint64_t m = W::ballot(mv);
                rem &= ~m;
                const Mask tl = ln < fcount;
                const uint32_t lo0 = v + W::reduce_add(W::select(tl & (fkey < v), fexc, zero));
                const uint64_t eq = W::ballot(tl & (fkey == v));
                uint32_t idx, e0 = 0;
                if (eq) { idx = W::ctz64(eq); e0 = W::readlane(fexc, idx); }
                else {
                    if (fcount >= CBC_CAP_FLAG) { capflag = capflag | mv; continue; }
                    idx = fcount++;
                    fkey = W::select(ln == idx, W::splat(v), fkey);
                }
                const V32 before = W::prefix_popc(m) * 8u;
                fl_cnt = W::select(mv, before + (1u + e0), fl_cnt);
                fl_base = W::select(mv, W::splat(lo0), fl_base);
                fl_extra = fl_extra + W::select(live & (flagv > v), before, zero);
                fexc_new = W::select(ln == idx, W::splat(e0 + 8u * W::popc64(m)), fexc_new);
            }
        }
        fexc = fexc_new;
        F.fl_lo = fl_base + fl_extra; F.fl_cnt = fl_cnt;

        /* -- rlength[0] (dense excess table in LDS; step 10) -- */
        const V32 xv = (r_fl >> 16) & 0xffu;
        const Mask rlbad = live & (xv >= 255u);            /* assert(x < alphabetCard) stream_model.c:62 */
        V32 rl_base = zero, rl_extra = zero, rl_cnt = zero, rl_new = zero;
        {
            uint64_t rem = W::ballot(live & (xv < 255u));
            while (rem) {
                const uint32_t v = W::readlane(xv, W::ctz64(rem));
                const Mask mv = live & (xv == v);
                const uint64_t m = W::ballot(mv);
                rem &= ~m;
                uint32_t lo0, cnt0;
                dense_lookup(rlen_exc, v, lo0, cnt0);
                const V32 before = W::prefix_popc(m) * 10u;
                rl_cnt = W::select(mv, before + cnt0, rl_cnt);
                rl_base = W::select(mv, W::splat(lo0), rl_base);
                rl_extra = rl_extra + W::select(live & (xv > v), before, zero);
                rl_new = W::select(mv, W::splat(cnt0 - 1u + 10u * W::popc64(m)), rl_new);
            }
            /* the table is written after all lookups (they must see it as it was before the group);
             * lanes with the same value store the same word */
            W::store32(rlen_exc, xv, rl_new, live & (xv < 255u));
        }
        F.rl_lo = rl_base + rl_extra; F.rl_cnt = rl_cnt;

        /* -- match: context = (dx == 1) * 2 + previous record's match; literal counts, step 1 -- */
        {
            const uint64_t mt = ~neq;
            const Mask sym = W::lane_bit(mt), pm = W::lane_bit((mt << 1) | (uint64_t)(prevM & 1u));
            const V32 ctx = W::select(dxv == 1u, W::splat(2u), zero) + W::select(pm, W::splat(1u), zero);
            V32 m_lo = zero, m_cnt = zero, m_n = zero;
            for (uint32_t c = 0; c < 4u; c++) {
                const Mask inc = live & (ctx == c);
                const uint64_t m0 = W::ballot(inc & !sym), m1 = W::ballot(inc & sym);
                const uint32_t k0 = W::readlane(small, CBC_LT_MATCH + 2u * c), k1 = W::readlane(small, CBC_LT_MATCH + 2u * c + 1u);
                const V32 b0 = W::prefix_popc(m0) + k0, b1 = W::prefix_popc(m1) + k1;
                m_lo = W::select(inc, W::select(sym, b0, zero), m_lo);
                m_cnt = W::select(inc, W::select(sym, b1, b0), m_cnt);
                m_n = W::select(inc, b0 + b1, m_n);
                small = W::select(ln == CBC_LT_MATCH + 2u * c, small + W::popc64(m0),
                        W::select(ln == CBC_LT_MATCH + 2u * c + 1u, small + W::popc64(m1), small));
            }
            F.m_lo = m_lo; F.m_cnt = m_cnt; F.m_n = m_n;
            prevM = (uint32_t)((mt >> (cn - 1u)) & 1ull);
        }

        /* -- pos: alphabet index of every delta (entry 0 = escape; pos_val[0] never matches) -- */
        const uint32_t card0 = pos_card;
        V32 kv = zero;
        /* deltas below CBC_POS_IDX_WORDS (nearly all: they are gaps between neighbouring reads) find their alphabet index
         * with one LDS load per lane; 0 there = not registered yet.  Only a group that holds a larger delta walks the alphabet. */
        const Mask small_dx = live & (dxv < CBC_POS_IDX_WORDS);
        if (CBC_POS_IDX_WORDS) kv = W::load32(pos_idx, dxv, small_dx, 0u);
        if (!CBC_POS_IDX_WORDS || W::ballot(live & !small_dx) != 0ull) {
            const uint32_t cb = W::uni(card0);
            for (uint32_t b = 0; b < cb; b += 64u) {
                const V32 av = W::load32(pos_val, ln + b, (ln + b) < card0, 0xffffffffu);
                const uint32_t lim = card0 - b < 64u ? card0 - b : 64u;
                for (uint32_t a = 0; a < lim; a++) kv = W::select(dxv == W::readlane(av, a), W::splat(b + a), kv);
                if (W::ballot(live & !small_dx & (kv == 0u)) == 0ull) break;
            }
        }
        /* deltas not in the alphabet yet: the first record with each becomes an escape and registers it */
        Mask cappos = W::lane_bit(0ull);
        uint64_t esc = 0;
        {
            uint64_t nf = W::ballot(live & !dxbad & (kv == 0u));
            uint32_t t = 0;
            while (nf) {
                const uint32_t f = W::ctz64(nf);
                const uint32_t v = W::readlane(dxv, f);
                const Mask mv = live & (dxv == v);
                nf &= ~W::ballot(mv);
                if (card0 + t >= cap_pos) { cappos = cappos | mv; continue; }
                kv = W::select(mv, W::splat(card0 + t), kv);
                esc |= 1ull << f;
                W::write_uni(pos_val, card0 + t, v);
                if (v < CBC_POS_IDX_WORDS) W::write_uni(pos_idx, v, card0 + t);
                t++;
            }
            pos_card = card0 + t;
        }
        /* occurrences of the same / of lower indices among the lower lanes */
        V32 lt = zero, eqb = zero, eqa = zero;
        for (uint32_t j = 0; j < cn; j++) {
            const uint32_t kj = W::readlane(kv, j);
            const Mask same = kv == kj;
            eqa = eqa + W::select(same, W::splat(1u), zero);
            eqb = eqb + W::select(same & (ln > j), W::splat(1u), zero);
            lt = lt + W::select((kv > kj) & (ln > j), W::splat(1u), zero);
        }
        {
            const Mask old = live & (kv != 0u) & (kv < card0);
            const V32 occ0 = W::load32(pos_occ, kv, old, 0u);
            const V32 pre0 = W::load32(pos_pre, kv, old, c0);          /* new entries: every earlier record lies below */
            const V32 cardv = W::prefix_popc(esc) + card0;             /* alphabet size when this record is coded */
            const V32 c_esc = (cardv - 1u) * 10u + 1u;                 /* count of the escape symbol then */
            const Mask isesc = W::lane_bit(esc);
            F.p_card = cardv;
            F.p_lo = W::select(isesc, zero, c_esc + (pre0 + lt) * 10u);
            F.p_cnt = W::select(isesc, c_esc, (occ0 + eqb) * 10u);
            W::store32(pos_occ, kv, occ0 + eqa, live & (kv != 0u) & !dxbad);
        }
        {   /* prefix of the occurrence counts by alphabet index, for the next group */
            uint32_t carry = 0;
            const uint32_t cb = W::uni(pos_card);
            for (uint32_t b = 0; b < cb; b += 64u) {
                const V32 i = ln + b; const Mask m = (i != 0u) & (i < pos_card);
                const V32 o = W::load32(pos_occ, i, m, 0u);
                const V32 inc = W::scan_incl_add(o);
                W::store32(pos_pre, i, inc - o + carry, i < pos_card);
                carry += W::readlane(inc, 63u);
            }
        }

        /* the status the record-by-record code stops with, in its order of checks */
        st = W::select(capflag, W::splat((uint32_t)CBC_ST_CAP_FLAG), st);
        st = W::select(cappos, W::splat((uint32_t)CBC_ST_CAP_POS), st);
        st = W::select(dxbad, W::splat((uint32_t)CBC_ST_ASSERT), st);
        st = W::select(rlbad, W::splat((uint32_t)CBC_ST_ASSERT), st);
        F.st = st; F.esc = esc;
        F.bad = W::ballot(live & (st != (uint32_t)CBC_ST_OK));
    }

    /* ---- var (read_compression.c:230-245): 65535 contexts x L0, kept as the list of events.
     * Most contexts are seen once per block, so an 8192-bit Bloom filter on the context answers
     * "never seen" (n = L0, cum = sym) without touching the list; only on a filter hit is the list
     * scanned, 512 events per trip (eight coalesced loads in flight). ---- */
    /* GEN: the reference's dense table (sam_models.c:311-348), kept as e = count - 1 in global memory (65535 rows
     * of L0 words, zero = untouched).  One round trip per symbol: the row's L0 <= 256 words are fetched as four
     * chunks of one word per lane, all in flight together; the same wavefront wrote them, through the L1-bypassing
     * list accessors, and waits for its own stores first. */
    CBC_MFN void var_code_dense(uint32_t ctx, uint32_t sym)
    {
        V32 ln = W::lane();
        uint32_t *row = vtab + (uint64_t)ctx * L0;
        W::list_fence();
        V32 ev[4];
        for (uint32_t q = 0; q < 4u; q++) { V32 i = ln + 64u * q; ev[q] = W::load32_list(row, i, i < L0, 0u); }
        V32 alo = W::splat(0u), aall = W::splat(0u);
        uint32_t esym = 0;
        for (uint32_t q = 0; q < 4u; q++) {
            V32 i = ln + 64u * q;
            alo = alo + W::select(i < sym, ev[q], W::splat(0u));
            aall = aall + ev[q];
            if ((sym >> 6) == q) esym = W::readlane(ev[q], sym & 63u);
        }
        const uint32_t lo = sym + W::reduce_add(alo), n = L0 + W::reduce_add(aall);
        encode(lo, 1u + esym, n);
        W::append_list(row, sym, esym + 10u);
        if (n + 10u >= CBC_RESCALE) {                       /* update_model stream_model.c:41-48 on e = count - 1 */
            W::list_fence();
            for (uint32_t q = 0; q < 4u; q++) {
                V32 i = ln + 64u * q;
                V32 e = W::load32_list(row, i, i < L0, 0u);
                W::store32_list(row, i, (e + 1u) >> 1, i < L0);
            }
        }
    }
    CBC_MFN void var_code(uint32_t ctx, uint32_t sym)
    {
        if (ctx >= CBC_NVARCTX || sym >= L0) { fail(CBC_ST_ASSERT); return; }
        if (GEN) { var_code_dense(ctx, sym); return; }
        /* Two classes of contexts (the class is a function of the context number alone):
         *  - "p = 0": (ctx >> 1) & 127 == 0 and ctx >> 8 < 255, i.e. ctx = d << 8 | strand -- the FIRST edit of a read, with the
         *    distance d to the first known SNP ahead.  With a few percent of the reference positions marked by
         *    earlier reads these are the contexts that repeat (a few hundred of them, each used several times per
         *    block), and they carry most of a block's var symbols.  Their events are 16 bits (d << 8 | symbol)
         *    in LDS, bucketed by (strand, d & 7), 128 events per bucket: a lookup is ONE LDS load per lane over the
         *    context's bucket, the three tallies in one register, one wave sum, and never leaves the CU.
         *  - all other contexts are mostly seen once per block: Bloom filter on the context, and only on a
         *    filter hit a scan of the strand's event list in global memory.
         * A bucket that is full spills its further events to the global list (p0over). */
        V32 ln = W::lane();
        uint32_t cn = 0, clo = 0, ceq = 0;
        const uint32_t key = (ctx << 8) | sym, strand1 = ctx & 1u;
        /* flags as 0 / 1 words, not bool: a bool that lives across blocks is kept as a 64-bit lane mask */
        const uint32_t p0class = (((ctx >> 1) & 0x7fu) == 0u && (ctx >> 8) != 255u) ? 1u : 0u;   /* 255 is the unused-half marker */
        uint32_t to_global = p0class ^ 1u;
        const uint32_t bkt = strand1 * 8u + ((ctx >> 8) & 7u);   /* the context's bucket: strand, d & 7 */
        uint32_t have = 0, half_word = 0;                      /* the bucket's last word when its upper half is free */
        if (p0class) {
            const uint32_t d = ctx >> 8, key16 = (d << 8) | sym;
            const uint32_t *arr = p0ev + bkt * CBC_P0_BUCKET_WORDS;
            have = W::readlane(p0cnt, bkt);
            const uint32_t nw = (have + 1u) >> 1;
            /* the whole bucket in one load; an unused upper half holds 0xffff (d = 255 is not in the class) */
            const V32 w = W::load32(arr, ln, ln < nw, 0xffffffffu);
            if (have & 1u) half_word = W::readlane(w, have >> 1);  /* saves the read of the read-modify-write below */
            const V32 e0 = w & 0xffffu, e1 = w >> 16;
            const V32 acc = W::select((e0 >> 8) == d, W::select((e0 & 0xffu) < sym, W::splat(1u + (1u << 10)), W::splat(1u)) +
                                                       W::select(e0 == key16, W::splat(1u << 20), W::splat(0u)), W::splat(0u)) +
                            W::select((e1 >> 8) == d, W::select((e1 & 0xffu) < sym, W::splat(1u + (1u << 10)), W::splat(1u)) +
                                                       W::select(e1 == key16, W::splat(1u << 20), W::splat(0u)), W::splat(0u));
            const uint32_t tot = W::reduce_add(acc);
            cn = tot & 1023u; clo = (tot >> 10) & 1023u; ceq = tot >> 20;
            if (have >= CBC_P0_CAP) to_global = 1u;
        }
        uint32_t h1 = 0, h2 = 0, bw1 = 0, bw2 = 0, bb1 = 0, bb2 = 0;
        if ((to_global | ((p0over >> bkt) & 1u)) != 0u) {
            /* two hash functions, both words fetched by one LDS instruction (lanes 0 and 1) */
            h1 = (ctx * 0x9E3779B1u) >> (32u - CBC_BLOOM_LOG2); h2 = (ctx * 0x85EBCA6Bu + 0x27D4EB2Fu) >> (32u - CBC_BLOOM_LOG2);
            const V32 bwv = W::load32(bloom, W::select(ln == 0u, W::splat(h1 >> 5), W::splat(h2 >> 5)), ln < 2u, 0u);
            bw1 = W::readlane(bwv, 0u); bw2 = W::readlane(bwv, 1u);
            bb1 = 1u << (h1 & 31u); bb2 = 1u << (h2 & 31u);
            if ((bw1 & bb1) && (bw2 & bb2)) {
                W::list_fence();
                /* the context's strand bit picks the list: strand 0 grows up from the bottom of the event area,
                 * strand 1 down from its top, so a scan reads half of the block's events and the two share the
                 * capacity */
                const uint32_t cnt_s = strand1 ? nev1 : nev, base_s = strand1 ? cap_var - nev1 : 0u;
                const uint32_t nb = W::uni(cnt_s);
                for (uint32_t b = 0; b < nb; b += 512u) {           /* eight coalesced loads in flight per trip */
                    V32 ev[8];
                    for (uint32_t q = 0; q < 8u; q++) { V32 i = ln + (b + 64u * q); ev[q] = W::load32_list(var_ev, i + base_s, i < cnt_s, 0xffffffffu); }
                    for (uint32_t q = 0; q < 8u; q++) {
                        const V32 e = ev[q];
                        const uint64_t bc = W::ballot((e >> 8) == ctx);    /* lanes past nev hold 0xffffffff: never a context */
                        if (bc) {
                            cn += W::popc64(bc);
                            clo += W::popc64(W::ballot(((e >> 8) == ctx) & ((e & 0xffu) < sym)));
                            ceq += W::popc64(W::ballot(e == key));
                        }
                    }
                }
            }
        }
        encode(sym + 10u * clo, 1u + 10u * ceq, L0 + 10u * cn);
        if (!to_global) {                                        /* p = 0 context with room in its bucket */
            uint32_t *arr = p0ev + bkt * CBC_P0_BUCKET_WORDS;
            const uint32_t k16 = ((ctx >> 8) << 8) | sym;
            if (have & 1u) W::write_uni(arr, have >> 1, (half_word & 0xffffu) | (k16 << 16));
            else W::write_uni(arr, have >> 1, 0xffff0000u | k16);
            p0cnt = W::select(ln == bkt, p0cnt + 1u, p0cnt);
            return;
        }
        if (nev + nev1 >= cap_var) { fail(CBC_ST_CAP_VAR); return; }
        if (p0class) p0over |= 1u << bkt;
        if (!((bw1 & bb1) && (bw2 & bb2))) {
            if ((h1 >> 5) == (h2 >> 5)) W::write_uni(bloom, h1 >> 5, bw1 | bb1 | bb2);
            else { W::write_uni(bloom, h1 >> 5, bw1 | bb1); W::write_uni(bloom, h2 >> 5, bw2 | bb2); }
        }
        if (strand1) { nev1++; W::append_list(var_ev, cap_var - nev1, key); }
        else { W::append_list(var_ev, nev, key); nev++; }
    }

    CBC_MFN void win_clear() { win.clear(); }
    CBC_MFN void win_shift(uint32_t d) { win.shift(d); }
    CBC_MFN uint32_t win_first(uint32_t p, uint32_t rl) { return win.first(p, rl); }
    CBC_MFN void win_set(uint32_t k) { win.set(k); }

    /* compress_edits for an imperfect read (read_compression.c:308-600).
     * The packer has already counted the edits (token word 1) and checked that the MD string is
     * consistent with the read, so every MD token becomes exactly one SNP: numSnps = n_md. */
    /* MODE 0: any record.  MODE 1: the caller has looked at the token header -- no indels, fewer than 64 SNPs, no rescale of
     * the SNP-count model due; the model wavefront codes runs of such records in a loop that holds neither the CIGAR walks nor
     * a table sweep.  (MODE 2: indels only.) */
    template <int MODE = 0>
    CBC_MFN void edits(uint32_t pos, uint32_t flw, uint32_t tok_off, const V32 &seqv, const V32 &tokv,
                       const uint32_t *tokb, uint32_t n_tok_blk)
    {
            CbcEnc &E = *this;
            const uint32_t rl = flw >> 16, strand = (flw >> 4) & 1u;
            {   /* snpInRef window: slide it to this record's POS (chr_change clears it, compression.c:62-63;
                 * it is empty at the start of the block, so the first slide may be anything) */
                const uint32_t d = pos - E.win_pos;
                E.win_shift(d > 256u ? 256u : d);
                E.win_pos = pos;
            }
            const uint32_t hdr = W::readlane(tokv, 0u), hdr1 = W::readlane(tokv, 1u);
            const uint32_t n_cig = hdr & 0xffffu, n_md = hdr >> 16;
            const uint32_t nSnp = n_md, nDel = hdr1 & 0xffffu, nIns = hdr1 >> 16;
            if (tok_off + 2u + n_cig + n_md > n_tok_blk || nSnp >= 1024u || nDel >= 1024u || nIns >= 1024u) {
                E.fail(CBC_ST_ASSERT); return;
            }
            CBC_TSM(11);                                      /* window slide, token header (waits for the prefetched loads) */
#define CBC_TOK(i) ((i) < 64u ? W::readlane(tokv, (i)) : W::read_uni(tokb + tok_off, (i)))
#define CBC_READ_BYTE(i) ((i) < rl ? ((W::readlane(seqv, (i) >> 2) >> (((i) & 3u) * 8u)) & 0xffu) : 0u)
#define CBC_SNP(gap_, letter_, cum_, p_) do {                                                              \
                CBC_TS(3);                                                                                  \
                uint32_t d_ = E.win_first(p_, rl);                                                          \
                CBC_TS(4);                                                                                  \
                E.var_code(((((d_ << 7) + (p_)) << 1) | strand), (gap_));                                   \
                CBC_TS(7);                                                                                  \
                (p_) += (gap_) + 1u;                                                                        \
                E.win_set((p_) - 1u);                     /* snpInRef[cumsumP+prev_pos-2] = 1  (:589) */     \
                E.small_code(CBC_LT_CHARS + cbc_basepair(letter_) * 8u, 5u, 8u, cbc_basepair(CBC_READ_BYTE(cum_))); \
                CBC_TS(8);                                                                                  \
            } while (0)
            const bool snp_only = MODE == 1 ? true : MODE == 2 ? false : (nDel | nIns) == 0u;
            if (MODE != 2 && snp_only) {
                /* SNP-only read (:557-558, :573-593): no insertion can interleave, so the MD tokens are
                 * the SNP list in order -- one loop, no CIGAR walk */
                if (MODE == 1) E.dense_code_low(E.snps_exc, 10u, nSnp, E.snps_n);     /* the caller checked: nSnp < 64 <= L0, no rescale */
                else E.dense_code(E.snps_exc, L0, 10u, nSnp & 0xffu, E.snps_n);
                CBC_TSM(12);                                  /* edit counts */
                uint32_t cum = 0, p = 0, k = 0;
                while (k < n_md && E.status == CBC_ST_OK) {   /* a full symbol queue (a read with dozens of SNPs) ends the run at its header */
                    for (; k < n_md && E.status == CBC_ST_OK && E.q_len < 56u; k++) {
                        const uint32_t t = CBC_TOK(2u + n_cig + k);
                        const uint32_t g = t >> 8;
                        cum += g;
                        CBC_SNP(g, t & 0xffu, cum, p);
                        cum++;
                    }
                    if (k < n_md && E.status == CBC_ST_OK) E.drain();
                }
                /* a leading soft clip / '*' is rejected by the packer; refuse it here as well */
                { const uint32_t t0 = CBC_TOK(2u); const uint32_t op0 = t0 & 15u;
                  if (n_cig && (op0 == CBC_OP_STAR || op0 == CBC_OP_S)) E.fail(CBC_ST_UNSUPPORTED); }
            } else if (MODE != 1) {
                E.dense_code(E.snps_exc, L0, 10u, 0u, E.snps_n);                 /* :561-564 */
                E.dense_code(E.indels_exc, L0, 16u, nSnp & 0xffu, E.indels_n);
                E.dense_code(E.indels_exc, L0, 16u, nDel & 0xffu, E.indels_n);
                E.dense_code(E.indels_exc, L0, 16u, nIns & 0xffu, E.indels_n);
                CBC_TSM(12);
                /* three walks over the CIGAR, in the emission order of :568-600: deletions, SNPs
                 * (interleaved with the insertions through add_snps_to_array's early return), insertions */
                for (int pass = 1; pass < 4 && E.status == CBC_ST_OK; pass++) {
                    if ((pass == 1 && nDel == 0u) || (pass == 2 && nSnp == 0u) || (pass == 3 && nIns == 0u)) continue;
                    uint32_t Mc = 0, ins = 0, prevI = 0, prevD = 0, p = 0;
                    uint32_t k = 0, cum = 0; bool more = true;       /* add_snps_to_array statics */
                    for (uint32_t o = 0; o <= n_cig && E.status == CBC_ST_OK; o++) {
                        uint32_t op, len;
                        if (o < n_cig) { uint32_t t = CBC_TOK(2u + o); op = t & 15u; len = t >> 4; }
                        else { op = 99u; len = 1u; }                  /* final pull with limit rl+1 :551 */
                        if (op == CBC_OP_M) { Mc += len; continue; }
                        if (op == CBC_OP_STAR || (op == CBC_OP_S && o == 0u)) { E.fail(CBC_ST_UNSUPPORTED); break; }
                        if (op == CBC_OP_D) {
                            if (pass == 1)
                                for (uint32_t c = 0; c < len && E.status == CBC_ST_OK; c++) {
                                    uint32_t g = Mc - prevD;
                                    if (E.q_len >= 56u) E.drain();
                                    E.var_code((p << 1) | strand, g);
                                    p += g; prevD = Mc;
                                }
                            continue;
                        }
                        /* I, trailing S, or the final pull */
                        for (uint32_t c = 0; c < len && E.status == CBC_ST_OK; c++) {
                            if (E.q_len >= 56u) E.drain();
                            if ((op == CBC_OP_I || op == 99u) && more && pass == 2) {
                                uint32_t limit = (op == 99u) ? rl + 1u : Mc + ins;
                                more = false;
                                while (k < n_md && E.status == CBC_ST_OK) {
                                    uint32_t t = CBC_TOK(2u + n_cig + k);
                                    uint32_t g = t >> 8;
                                    if (cum + g >= limit) { cum++; more = true; break; }
                                    if (E.q_len >= 56u) E.drain();
                                    cum += g;
                                    CBC_SNP(g, t & 0xffu, cum, p);
                                    cum++; k++;
                                }
                            }
                            if (op == 99u) break;
                            if (pass == 3) {
                                uint32_t g = Mc - prevI;
                                uint32_t base = cbc_basepair(CBC_READ_BYTE(Mc + ins));
                                E.var_code((p << 1) | strand, g);
                                p += g;
                                E.small_code(CBC_LT_CHARS + 5u * 8u, 5u, 8u, base);
                                prevI = Mc;
                            }
                            ins++;
                        }
                    }
                }
            }
            CBC_TSM(13);                                      /* rest of edits(): loop ends, the indel walks */
#undef CBC_SNP
#undef CBC_TOK
#undef CBC_READ_BYTE
    }
};

/* ===========================================================================================
 * cbc_encode_stream: code block `blk` completely.  `lds` = this wavefront's table memory
 * (cbc_gpu_lds_bytes() bytes).
 * =========================================================================================== */
template <class W, uint32_t ROLE>
CBC_FN void cbc_encode_stream(const cbc_enc_args &A, uint32_t blk, uint32_t *lds)
{
    const uint32_t role = ROLE;
    typedef typename W::V32 V32;
    typedef typename W::Mask Mask;
    const V32 ln = W::lane();
    const cbc_block_desc *bd = A.blocks + blk;
    CbcEnc<W> E;

    /* ---- block descriptor (uniform) ---- */
    const uint64_t rec_base = bd->rec_base, seq_base = bd->seq_base, tok_base = bd->tok_base;
    const uint64_t ref_off = bd->ref_off, out_off = bd->out_off;
    const uint32_t out_cap = bd->out_cap, n_reads = bd->n_reads, name_off = bd->name_off;
    const uint32_t L0 = bd->read_length, n_tok_blk = bd->n_tok;

    E.status = CBC_ST_OK; E.nsym = 0; E.fail_read = 0; E.cur_read = 0;
    E.l = W::uv(0u); E.set_l_forms(); E.rng = W::uv(CBC_M26 + 1u); E.scale3 = 0u; E.bitpos = 0; E.flushed = 0;
    E.ring = lds + CBC_LDS_RING;
    E.q_lo = W::splat(0u); E.q_cnt = W::splat(0u); E.q_n = W::splat(0u); E.q_len = 0;
    E.b_lo = W::splat(0u); E.b_hi = W::splat(0u); E.b_n = W::splat(1u); E.b_fl = W::splat(0u); E.b_fh = W::splat(0u);
    E.b_len = 0; E.b_pos = 0; E.b_stop = 64u; E.b_flags = 0; E.seen_last = 0; E.b_neq = 0;
    E.rec_a = W::splat(0u); E.rec_s = W::splat(0u); E.rec_n = 0;
    E.role = role; E.batch_i = 0; E.batch = lds + CBC_LDS_BATCH; E.ctl = lds + CBC_LDS_CTL;
    if (ROLE != CBC_ROLE_FUSED) {                            /* the only barrier: the counters start at zero for both waves */
        if (ROLE == CBC_ROLE_MODEL) { W::write_uni(E.ctl, 0u, 0u); W::write_uni(E.ctl, 1u, 0u); W::write_uni(E.ctl, 2u, 0u); W::write_uni(E.ctl, 3u, 0u); }
        W::barrier();
    }
    /* the block's out area: [0, payload_cap) payload, [payload_cap, out_cap) its var-event list */
    const uint32_t payload_cap = bd->reserved;
    E.out32 = (uint32_t *)(A.out + out_off);
    E.cap_words = payload_cap >> 2;
    E.var_ev = (uint32_t *)(A.out + out_off + payload_cap);
    E.cap_var = payload_cap <= out_cap ? (out_cap - payload_cap) >> 2 : 0u;
    bool args_ok = cbc_fits64(out_off, out_cap, A.out_bytes) && ((out_off & 3u) == 0) && ((payload_cap & 3u) == 0) &&
                   (payload_cap <= out_cap) &&
                   cbc_fits64(rec_base, n_reads, A.n_recs) && cbc_fits64(tok_base, n_tok_blk, A.n_tok) &&
                   (L0 >= 1u && L0 <= 256u) && (name_off < A.names_bytes);
    if (!args_ok) { E.cap_words = 0; E.fail(CBC_ST_ASSERT); }
    /* the closed forms of fixed_group() hold while no per-record model can reach its rescale point */
    if (n_reads > CBC_MAX_BLOCK_READS) E.fail(CBC_ST_UNSUPPORTED);

    /* ---- model tables.  LDS words [0, 256) rlength, the output ring and the pos arrays belong to the
     * wavefront that codes the fixed symbols (coder / fused); snps, indels, the name list, the Bloom
     * filter and the hot var slots to the one that produces the segments (model / fused).
     * (alloc_read_models_t sam_models.c:562-586 etc.) ---- */
    E.L0 = L0;
    E.rlen_exc = lds + CBC_LDS_RLEN; E.snps_exc = lds + CBC_LDS_SNPS; E.indels_exc = lds + CBC_LDS_INDELS;
    E.rname_key = lds + CBC_LDS_RNKEY; E.rname_exc = lds + CBC_LDS_RNEXC;
    E.bloom = lds + CBC_LDS_BLOOM; E.p0ev = lds + CBC_LDS_P0;
    E.pos_val = lds + CBC_LDS_FIXED; E.pos_occ = E.pos_val + A.cap_pos; E.pos_pre = E.pos_occ + A.cap_pos;
    E.fsp_key = E.fsp_exc = nullptr; E.fsp_count = 0; E.pos_ov_val = E.pos_ov_occ = nullptr; E.pos_lds_cap = 0xffffffc0u; E.palpha = nullptr;
    E.pos_idx = E.pos_pre + A.cap_pos;
    E.cap_pos = A.cap_pos;
    if (ROLE != CBC_ROLE_MODEL) {
        for (uint32_t b = 0; b < CBC_POS_IDX_WORDS; b += 64u) W::store32(E.pos_idx, ln + b, W::splat(0u), W::all());
        for (uint32_t b = 0; b < CBC_RING_WORDS; b += 64u) W::store32(E.ring, ln + b, W::splat(0u), W::all());
        for (uint32_t b = 0; b < 256u; b += 64u) W::store32(E.rlen_exc, ln + b, W::splat(0u), W::all());
        W::write_uni(E.pos_val, 0u, 0xffffffffu); W::write_uni(E.pos_occ, 0u, 0u); W::write_uni(E.pos_pre, 0u, 0u);
    }
    if (ROLE != CBC_ROLE_CODER) {
        for (uint32_t b = 0; b < 512u; b += 64u) W::store32(E.snps_exc, ln + b, W::splat(0u), W::all());   /* snps + indels */
        for (uint32_t b = 0; b < CBC_BLOOM_WORDS; b += 64u) W::store32(E.bloom, ln + b, W::splat(0u), W::all());   /* the p = 0 arrays need no clearing */
    }
    E.p0cnt = W::splat(0u); E.p0over = 0;
    E.snps_n = L0; E.indels_n = L0;
    E.rn_count = 0; E.rn_cap = CBC_CAP_NAME; E.vtab = nullptr;
    E.pos_card = 1u;                                         /* initialize_stream_model_pos :132-162: the escape */
    E.nev = 0; E.nev1 = 0;
    E.fkey = W::splat(0u); E.fexc = W::splat(0u); E.fcount = 0;
    E.hkey = W::splat(0u); E.hexc = W::splat(0u);
    E.hc0 = E.hc1 = E.hc2 = E.hc3 = 0; E.hn0 = E.hn1 = E.hn2 = E.hn3 = 256u;
    {   /* lane table: match 1,1 (n=2); same_ref 1,1; chars rows (sam_models.c:372-401) */
        V32 sm = W::splat(0u);
        sm = W::select(ln < 10u, W::splat(1u), sm);
        V32 r = (ln - CBC_LT_CHARS) >> 3, c = (ln - CBC_LT_CHARS) & 7u;
        Mask inch = (ln >= CBC_LT_CHARS) & (r < 6u) & (c < 5u);
        V32 cv = W::select(c == 4u, W::splat(1u), W::select(c == r, W::splat(0u), W::splat(8u)));
        /* +8 on two entries per row: A:{C,G} C:{A,T} G:{A,T} T:{C,G} */
        Mask bump = ((r == 0u) & ((c == 1u) | (c == 2u))) | ((r == 1u) & ((c == 0u) | (c == 3u))) |
                    ((r == 2u) & ((c == 0u) | (c == 3u))) | ((r == 3u) & ((c == 1u) | (c == 2u)));
        cv = W::select(bump, cv + 8u, cv);
        sm = W::select(inch, cv, sm);
        E.small = sm;
    }
    E.prevPos = 0; E.prevM = 0; E.prevChar = 0; E.win_pos = 0;
    E.win_clear();

    const uint4 *recs4 = (const uint4 *)(A.recs + rec_base);
    const uint8_t *seqb = A.seq + seq_base;
    const uint32_t *tokb = A.tok + tok_base;
    const uint8_t *refb = A.ref + ref_off;
    const uint64_t seq_avail = cbc_le64(seq_base, A.seq_bytes) ? A.seq_bytes - seq_base : 0;
    const uint64_t ref_avail = cbc_le64(ref_off, A.ref_bytes) ? A.ref_bytes - ref_off : 0;
    const uint32_t seq_lim = cbc_avail32(seq_avail, 0u), ref_lim = cbc_avail32(ref_avail, 0u);
#ifdef CBC_STAMP
    for (int i = 0; i < 16; i++) E.t_sum[i] = 0;
    CBC_T0();
#endif

    /* ================================ segment generators ==================================== */
    /* stream header: int(L0), 32 x int(WELL), int(LOSSLESS=8)  (sam_file_allocation.c:371, 392-403;
     * compression.c:139; compress_int qv_codebook.c:14-50) */
    auto gen_header = [&]() {
        for (uint32_t k = 0; k < 34u && E.status == CBC_ST_OK; k++) {
            uint32_t v = (k == 0u) ? L0 : (k == 33u) ? 8u : CBC_WELL_SEED;
            E.regsparse_code(E.hkey, E.hexc, 0u, 8u, E.hc0, E.hn0, 256u, 1u, v >> 24, CBC_ST_ASSERT);
            E.regsparse_code(E.hkey, E.hexc, 8u, 8u, E.hc1, E.hn1, 256u, 1u, (v >> 16) & 0xffu, CBC_ST_ASSERT);
            E.regsparse_code(E.hkey, E.hexc, 16u, 8u, E.hc2, E.hn2, 256u, 1u, (v >> 8) & 0xffu, CBC_ST_ASSERT);
            E.regsparse_code(E.hkey, E.hexc, 24u, 8u, E.hc3, E.hn3, 256u, 1u, v & 0xffu, CBC_ST_ASSERT);
            if ((k & 7u) == 7u) E.drain();
        }
    };
    /* the contig name of record 0, compress_rname (id_compression.c:39-65); a block holds one contig */
    auto gen_rname = [&]() {
        for (uint32_t q = 0; E.status == CBC_ST_OK; q++) {
            uint32_t ch = (name_off + q < A.names_bytes) ? W::read_uni8(A.names, name_off + q) : 0u;
            E.rname_code(E.prevChar, ch);
            if ((q & 31u) == 31u) E.drain();                  /* long contig names: keep the queue short */
            if (ch == 0u) break;
            E.prevChar = ch;
        }
    };
    /* end-of-stream sentinel compress_rname("\n") (compression.c:152), after its same_ref symbol */
    auto gen_sentinel = [&]() {
        if (E.q_len >= 56u) E.drain();
        E.cur_read = n_reads;
        E.rname_code(E.prevChar, (uint32_t)'\n');
        E.rname_code((uint32_t)'\n', 0u);
    };
    /* match test of a group's records (read_compression.c:291-296): lane l compares bases 4l..4l+3;
     * the loads of 8 records are in flight together */
    auto match_group = [&](uint32_t cn, const V32 &r_pos, const V32 &r_fl, const V32 &r_seq) -> uint64_t {
        uint64_t neq = 0;
        for (uint32_t j0 = 0; j0 < cn; j0 += 8u) {
            V32 sv[8], rv[8]; uint32_t rls[8];
            const V32 bo = ln * 4u;
            for (uint32_t q = 0; q < 8u; q++) {                 /* lanes past cn hold zero records: nothing is loaded */
                const uint32_t jj = (j0 + q) & 63u;
                const uint32_t pos = W::readlane(r_pos, jj), so = W::readlane(r_seq, jj);
                rls[q] = (j0 + q < cn) ? W::readlane(r_fl, jj) >> 16 : 0u;
                sv[q] = W::load32_bytes(seqb + so, bo, bo < rls[q]);
                rv[q] = W::load32_bytes(refb + (pos - 1u), bo, bo < rls[q]);
            }
            for (uint32_t q = 0; q < 8u; q++) {
                const uint32_t rl = rls[q];
                V32 bmask = W::select(bo + 4u <= rl, W::splat(0xffffffffu),
                                      W::select(bo < rl, (W::splat(1u) << ((W::splat(rl) - bo) * 8u)) - 1u, W::splat(0u)));
                if (W::ballot(((sv[q] ^ rv[q]) & bmask) != 0u)) neq |= 1ull << ((j0 + q) & 63u);
            }
        }
        return neq;
    };
    /* the 64 records of a group, validated one lane each so that the per-record loads need no clamping:
     * read length 1..252, POS >= 1, bases and reference window inside the buffers */
    auto load_group = [&](uint32_t c0, V32 &r_pos, V32 &r_fl, V32 &r_seq, V32 &r_tok, bool check) -> bool {
        W::load_rec(recs4, ln + c0, (ln + c0) < n_reads, r_pos, r_fl, r_seq, r_tok);
        if (!check) return true;
        V32 vrl = r_fl >> 16;
        Mask live = (ln + c0) < n_reads;
        Mask bad = live & ((vrl == 0u) | (vrl > CBC_MAX_READ_LEN) | (r_pos == 0u) |
                           ((r_seq + vrl + 4u) > seq_lim) | ((r_pos + vrl + 3u) > ref_lim) | (r_tok >= n_tok_blk));
        uint64_t bb = W::ballot(bad);
        if (bb) { E.cur_read = c0 + W::ctz64(bb); E.fail(CBC_ST_ASSERT); return false; }
        return true;
    };

    /* ================================ model wavefront ======================================== */
    if (ROLE == CBC_ROLE_MODEL) {
        if (CBC_PRIO_MODEL) W::prio(CBC_PRIO_MODEL);
        if (E.status == CBC_ST_OK) { gen_header(); E.seg_end(); }
        for (uint32_t c0 = 0; c0 < n_reads && E.status == CBC_ST_OK; c0 += 64u) {
            V32 r_pos, r_fl, r_seq, r_tok;
            const uint32_t cn = n_reads - c0 < 64u ? n_reads - c0 : 64u;
            E.cur_read = c0;
#ifndef CBC_MATCH_IN_MODEL
            /* the coder wavefront has validated the group and run the match test (post_group); everything this
             * wavefront still holds goes over first -- the coder may need it before it can post the next mask */
            if (E.q_len) E.publish(0u, 0ull);
            uint64_t neq;
            if (!E.wait_group(c0 >> 6, neq)) { E.fail(CBC_ST_ASSERT); break; }   /* the coder reports the record */
            load_group(c0, r_pos, r_fl, r_seq, r_tok, false);
            CBC_TS(0);                                        /* group loads (+ match test) */
#else
            if (!load_group(c0, r_pos, r_fl, r_seq, r_tok, true)) break;
            const uint64_t neq = match_group(cn, r_pos, r_fl, r_seq);
            CBC_TS(0);                                        /* group loads + match test */
            if (E.q_len) E.publish(0u, 0ull);                 /* a GROUP batch carries no symbols */
            E.publish(CBC_BF_GROUP, neq);
#endif
            if (c0 == 0u) { gen_rname(); E.seg_end(); }
            /* software prefetch: bases and tokens of the next imperfect record */
            uint64_t todo = neq;
            V32 nx_seq = W::splat(0u), nx_tok = W::splat(0u);
            if (todo) {
                const uint32_t jn = W::ctz64(todo);
                const uint32_t so = W::readlane(r_seq, jn), to = W::readlane(r_tok, jn), nrl = W::readlane(r_fl, jn) >> 16;
                V32 bo = ln * 4u;
                nx_seq = W::load32_bytes(seqb + so, bo, bo < nrl);
                nx_tok = W::load32(tokb + to, ln, (ln + to) < n_tok_blk, 0u);
            }
            /* the record at the head of `todo` (its bases and tokens are in nx_seq / nx_tok), and the next one's loads */
            auto one = [&](auto mode) {
                const uint32_t j = W::ctz64(todo);
                todo &= todo - 1ull;
                E.cur_read = c0 + j;
                const V32 seqv = nx_seq, tokv = nx_tok;
                if (todo) {
                    const uint32_t jn = W::ctz64(todo);
                    const uint32_t so = W::readlane(r_seq, jn), to = W::readlane(r_tok, jn), nrl = W::readlane(r_fl, jn) >> 16;
                    V32 bo = ln * 4u;
                    nx_seq = W::load32_bytes(seqb + so, bo, bo < nrl);
                    nx_tok = W::load32(tokb + to, ln, (ln + to) < n_tok_blk, 0u);
                }
                E.template edits<decltype(mode)::value>(W::readlane(r_pos, j), W::readlane(r_fl, j), W::readlane(r_tok, j), seqv, tokv, tokb, n_tok_blk);
                E.seg_end_fits();
                CBC_TS(1);                                    /* edits of one record */
                if (E.q_len >= CBC_BATCH_MIN) E.drain();      /* hand over once a few records' symbols are pending */
            };
            /* runs of ordinary records -- SNPs only (token word 1 = deletions | insertions << 16 is zero), fewer than 64 of
             * them, the SNP-count model not about to rescale -- go through a loop whose body holds no CIGAR walk and no table
             * sweep; any other record is coded between two such runs by the general form */
            auto ordinary = [&]() { return W::readlane(nx_tok, 1u) == 0u && (W::readlane(nx_tok, 0u) >> 16) < 64u && L0 >= 64u && E.snps_n + 10u < CBC_RESCALE; };
            while (todo && E.status == CBC_ST_OK) {
                while (todo && E.status == CBC_ST_OK && ordinary()) one(std::integral_constant<int, 1>());
                if (todo && E.status == CBC_ST_OK) one(std::integral_constant<int, 0>());
            }
        }
        if (E.status == CBC_ST_OK) { gen_sentinel(); E.seg_end(); }
#if defined(CBC_STAMP) && defined(__HIP_DEVICE_COMPILE__)
        if (E.cap_var >= 64u) for (int i = 0; i < 16; i++) {   /* diagnostic build: tail of the event area */
            W::write_uni(E.var_ev + E.cap_var - 32u, 2 * i, (uint32_t)E.t_sum[i]); W::write_uni(E.var_ev + E.cap_var - 32u, 2 * i + 1, (uint32_t)(E.t_sum[i] >> 32)); }
#endif
        E.publish(CBC_BF_LAST, 0ull);                         /* carries the status if a check failed */
        return;
    }

    /* ========================== coder wavefront / fused emulation =========================== */
    const bool fused = ROLE == CBC_ROLE_FUSED;
    /* the coder wavefront carries the block's serial chain: it goes first at the SIMD's issue port (s_setprio), the
     * model wavefronts that share the SIMD fill the gaps (cfg2 10.26 -> 9.91 ms, profiles/r02_ab_kernels.log run 9) */
    if (!fused) W::prio(CBC_PRIO_CODER);
#ifndef CBC_MATCH_IN_MODEL
    /* validate a group and run its match test for both wavefronts; a refused group is posted as such (the model
     * wavefront then winds up) and fails this wavefront through load_group() */
    uint64_t neq_ahead = 0;
    auto look_ahead = [&](uint32_t g0) {
        V32 a_pos, a_fl, a_seq, a_tok;
        const uint32_t an = n_reads - g0 < 64u ? n_reads - g0 : 64u;
        const uint32_t saved = E.cur_read;
        const bool okg = load_group(g0, a_pos, a_fl, a_seq, a_tok, true);
        neq_ahead = okg ? match_group(an, a_pos, a_fl, a_seq) : 0ull;
        E.post_group(g0 >> 6, neq_ahead, okg ? 0u : 1u);
        if (okg) E.cur_read = saved;
    };
    if (!fused && n_reads != 0u && E.status == CBC_ST_OK) look_ahead(0u);
#endif
    if (E.status == CBC_ST_OK) { if (fused) { gen_header(); E.seg_end(); } else E.seg_consume(); }
    for (uint32_t c0 = 0; c0 < n_reads && E.status == CBC_ST_OK; c0 += 64u) {
        V32 r_pos, r_fl, r_seq, r_tok;
        const uint32_t cn = n_reads - c0 < 64u ? n_reads - c0 : 64u;
        E.cur_read = c0;
        uint64_t neq;
        if (fused) {
            if (!load_group(c0, r_pos, r_fl, r_seq, r_tok, true)) break;
            neq = match_group(cn, r_pos, r_fl, r_seq);
        } else {
#ifndef CBC_MATCH_IN_MODEL
            neq = neq_ahead;                                  /* posted one group ago */
            if (c0 + 64u < n_reads) look_ahead(c0 + 64u);     /* before anything of this group is waited for */
            if (E.status != CBC_ST_OK) break;
#else
            neq = E.pull_group();                             /* the model wave has validated the group */
            if (E.status != CBC_ST_OK) break;
#endif
            load_group(c0, r_pos, r_fl, r_seq, r_tok, false);
        }
        typename CbcEnc<W>::Fixed F;
        E.fixed_group(F, c0, cn, r_pos, r_fl, neq);
        /* totals and the scaled fractions of every fixed symbol of the group, one lane per record */
        const V32 rdv = ln + c0;
        const V32 sr_hi = rdv * 10u - 9u, sr_n = rdv * 10u + 2u;          /* same_ref symbol 0 of record r >= 1 */
        const V32 sr_fh = CBC_FRAC(sr_hi, sr_n);
        const V32 t_hi = rdv * 10u + 1u, t_n = rdv * 10u + 255u;          /* rlength[1..3] symbol 0; rlength[0]'s total */
        const V32 t_fh = CBC_FRAC(t_hi, t_n);
        const V32 rl_hi = F.rl_lo + F.rl_cnt;
        const V32 rl_fl = CBC_FRAC(F.rl_lo, t_n), rl_fh = CBC_FRAC(rl_hi, t_n);
        const V32 p_n = (F.p_card - 1u) * 10u + rdv * 10u + 1u, p_hi = F.p_lo + F.p_cnt;
        const V32 p_fl = CBC_FRAC(F.p_lo, p_n), p_fh = CBC_FRAC(p_hi, p_n);
        const V32 fl_n = rdv * 8u + 65536u, fl_hi = F.fl_lo + F.fl_cnt;
        const V32 fl_fl = CBC_FRAC(F.fl_lo, fl_n), fl_fh = CBC_FRAC(fl_hi, fl_n);
        const V32 m_hi = F.m_lo + F.m_cnt;
        const V32 m_fl = CBC_FRAC(F.m_lo, F.m_n), m_fh = CBC_FRAC(m_hi, F.m_n);

#ifndef CBC_ENC_NO_PEEL      /* record 0 is its own instantiation of the body: its rare, branchy code stays out of the loop
                             * (cfg2 encode 10.15 -> 9.06 ms, decode 30.4 -> 29.85 ms: profiles/r02_ab_kernels.log run 13) */
#define CBC_ENC_FIRST(first, r) (decltype(first)::value)
#else
#define CBC_ENC_FIRST(first, r) ((r) == 0u)
#endif
        /* one record's symbols.  `first` = record 0 of the block (a compile-time flag: the loop body proper has no
         * special case in it, and nothing leaves the loop from inside -- every early exit costs the structurised
         * control flow a flag that is then tested at each join) */
#define CBC_LZ(expr) [&]() -> uint32_t { return (expr); }
        auto code_record = [&](uint32_t j, auto first, auto esc) {          /* esc: 0 = no escape, 1 = escape, 2 = look */
            const uint32_t r = c0 + j;
            E.cur_read = r;
            /* -- compress_rname (id_compression.c:39-65): same_ref is (1,1) until record 0 codes symbol 1,
             *    after which only symbol 0 is coded; the name itself is the model wave's segment -- */
            E.room(8u);                                       /* same_ref, rlength x 4, pos, flag, match */
            if (!CBC_ENC_FIRST(first, r)) E.step_known0(CBC_LZ(W::readlane(sr_hi, j)), CBC_LZ(10u * r + 2u), W::readlane(sr_fh, j));
            else {
                E.encode(1u, 1u, 2u); E.drain_q();
                if (fused) { gen_rname(); E.seg_end(); } else E.seg_consume();
            }
            /* -- read length, 4 "bytes" (read_compression.c:29-33, quirk Q1): rlength[0] from the group
             *    pass; contexts 1..3 only ever code symbol 0, each once per record -- */
            {
                const uint32_t tn = 255u + 10u * r, tf = W::readlane(t_fh, j);
                E.step_fixed(CBC_LZ(W::readlane(F.rl_lo, j)), CBC_LZ(W::readlane(rl_hi, j)), CBC_LZ(tn), W::readlane(rl_fl, j), W::readlane(rl_fh, j));
                E.step_known0(CBC_LZ(W::readlane(t_hi, j)), CBC_LZ(tn), tf);
                E.step_known0(CBC_LZ(W::readlane(t_hi, j)), CBC_LZ(tn), tf);
                E.step_known0(CBC_LZ(W::readlane(t_hi, j)), CBC_LZ(tn), tf);
            }
            /* -- compress_pos: hit, or escape + the four bytes of the new delta -- */
            E.step_fixed(CBC_LZ(W::readlane(F.p_lo, j)), CBC_LZ(W::readlane(p_hi, j)), CBC_LZ(W::readlane(p_n, j)), W::readlane(p_fl, j), W::readlane(p_fh, j));
            if (decltype(esc)::value == 1 || (decltype(esc)::value == 2 && ((F.esc >> j) & 1ull))) {
                const uint32_t card = W::readlane(F.p_card, j);
                E.pos_alpha(W::read_uni(E.pos_val, card), card);
                E.drain_q();
            }
            /* -- compress_flag (read_compression.c:50-70), then the match flag -- */
            E.step_fixed(CBC_LZ(W::readlane(F.fl_lo, j)), CBC_LZ(W::readlane(fl_hi, j)), CBC_LZ(65536u + 8u * r), W::readlane(fl_fl, j), W::readlane(fl_fh, j));
            E.step_fixed(CBC_LZ(W::readlane(F.m_lo, j)), CBC_LZ(W::readlane(m_hi, j)), CBC_LZ(W::readlane(F.m_n, j)), W::readlane(m_fl, j), W::readlane(m_fh, j));
            if ((neq >> j) & 1ull) {
                if (fused) {
                    const uint32_t so = W::readlane(r_seq, j), to = W::readlane(r_tok, j), flw = W::readlane(r_fl, j);
                    const V32 bo = ln * 4u;
                    const V32 seqv = W::load32_bytes(seqb + so, bo, bo < (flw >> 16));
                    const V32 tokv = W::load32(tokb + to, ln, (ln + to) < n_tok_blk, 0u);
                    E.edits(W::readlane(r_pos, j), flw, to, seqv, tokv, tokb, n_tok_blk);
                    E.seg_end();
                } else E.seg_consume();
            }
        };
#undef CBC_LZ
        /* records up to the first one fixed_group() refused; that one reports its status after the loop */
        uint32_t jn = cn;
        if (F.bad) { const uint32_t fb = W::ctz64(F.bad); if (fb < cn) jn = fb; }
#ifndef CBC_ENC_NO_PEEL
        /* ... and so is a record whose POS delta is new (escape + four byte symbols + a drain of the symbol queue, one
         * record in ~16): the records between two of them run through a body without that branch */
        typedef std::integral_constant<int, 0> Plain; typedef std::integral_constant<int, 1> Esc; typedef std::integral_constant<int, 2> Look;
        uint32_t j = 0;
        if (c0 == 0u && jn != 0u) { code_record(0u, std::true_type(), Look()); j = 1u; }
        while (j < jn && E.status == CBC_ST_OK) {
            const uint64_t ahead = F.esc >> j;                            /* j < 64 */
            uint32_t run_end = ahead ? j + W::ctz64(ahead) : jn;
            if (run_end > jn) run_end = jn;
            for (; j < run_end && E.status == CBC_ST_OK; j++) code_record(j, std::false_type(), Plain());
            if (j < jn && E.status == CBC_ST_OK) { code_record(j, std::false_type(), Esc()); j++; }
        }
#else
        for (uint32_t j = 0; j < jn && E.status == CBC_ST_OK; j++) code_record(j, 0, std::integral_constant<int, 2>());
#endif
#undef CBC_ENC_FIRST
        if (jn < cn && E.status == CBC_ST_OK) { E.cur_read = c0 + jn; E.fail(W::readlane(F.st, jn)); }
    }
    /* ---- end-of-stream sentinel (compression.c:152): same_ref symbol 1 at counts (1 + 10 (n - 1), 11),
     *      then the model wave's two name symbols; flush ---- */
    uint32_t nbytes = 0;
    if (E.status == CBC_ST_OK) {
        E.cur_read = n_reads;
        if (n_reads) E.encode(10u * n_reads - 9u, 11u, 10u * n_reads + 2u);
        else E.encode(1u, 1u, 2u);
        E.drain_q();
        if (fused) { gen_sentinel(); E.seg_end(); } else E.seg_consume();
    }
    if (E.status == CBC_ST_OK) { E.flush_recs(); nbytes = E.finish(); }
    if (E.status != CBC_ST_OK) nbytes = 0;
#ifndef CBC_MATCH_IN_MODEL
#ifndef CBC_MATCH_NO_ABORT_FOR_TEST
    if (!fused && E.status != CBC_ST_OK) E.abort_groups();
#endif
#endif
    if (!fused) E.pull_rest();
#if defined(CBC_STAMP) && defined(__HIP_DEVICE_COMPILE__)
    if (payload_cap >= 128u) for (int i = 0; i < 16; i++) {   /* diagnostic build: over the payload start */
        W::write_uni(E.out32, 2 * i, (uint32_t)E.t_sum[i]); W::write_uni(E.out32, 2 * i + 1, (uint32_t)(E.t_sum[i] >> 32)); }
#endif
    V32 resv = W::select(ln == 0u, W::splat(nbytes), W::select(ln == 1u, W::splat(E.status),
               W::select(ln == 2u, W::splat(E.nsym), W::splat(E.fail_read))));
    W::store32((uint32_t *)(A.results + blk), ln, resv, ln < 4u);
}

#endif /* CBC_ENCODE_BODY_H */
