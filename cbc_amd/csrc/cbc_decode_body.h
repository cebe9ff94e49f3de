/*
 * cbc_decode_body.h -- one arithmetic stream per wavefront, decode direction.
 *
 * Mirror of cbc_encode_body.h: the same sparse/exact model tables, evolved by the same updates, but
 * every symbol is FOUND from the coder's target value instead of looked up:
 *   read_value_from_as()        src/stream_model.c:78-117      (linear search -> wave-parallel search)
 *   arithmetic_get_symbol_range src/Arithmetic_stream.c:373-381
 *   arithmetic_decoder_step     src/Arithmetic_stream.c:389-454 (E1/E2 + E3 loops in closed form)
 *   decompress_rname / _read / _pos / _flag / reconstruct_read
 *                               src/id_compression.c:67-94, src/read_decompression.c:59-529
 * It is the exact inverse of this repository's encoder (same contexts, same snpInRef marks) and
 * writes one reconstructed SEQ per record (what print_line, src/compression.c:16-40, ends up printing:
 * the reference reverse-complements reverse-strand reads twice).  It does not reproduce the
 * reference decoder's own quirks (header-length reads, the deltaPos rule of :365-369).
 *
 * Search for a symbol given the target t in [0, n): with e = count-1 and cum(x) = x + sum_{s<x} e[s],
 * a SEEN symbol s owns [cum(s), cum(s)+1+e[s]) and an unseen symbol x owns the single value cum(x);
 * so either t falls into a seen symbol's interval, or x = t - (sum of e over seen symbols that lie
 * entirely below t).  Both tests are one compare per table entry + a ballot / wave sum.
 */
#ifndef CBC_DECODE_BODY_H
#define CBC_DECODE_BODY_H

#include <stdint.h>
#include <type_traits>
#include "../../include/cbc_gpu.h"
#include "cbc_encode_body.h"      /* constants, lane-table map, cbc_basepair, cbc_le64 */

/* LDS words of the decoder: the encoder's tables (its batch area is reused here for the scratch read
 * and the deletion list), then the 4 x 256 pos_alpha byte histograms as u16 pairs and the insertion list */
#define CBC_DLDS_TMP     CBC_PLAN_TABLE_WORDS          /* 80 words: insertion-free read, bytes      */
#define CBC_DLDS_DELS    (CBC_PLAN_TABLE_WORDS + 80u)  /* 256: deletion positions (matched coords)  */
/* The decoder's Bloom filter is its own (a filter only decides whether the list is scanned, never what is decoded): it uses
 * half of the table's words, the other half holds var_dec's symbol histogram -- the decoder's LDS must stay at 15 KB, a tenth
 * wavefront does not fit on a CU above that (profiles/r02_ab_kernels.log runs 28-29) */
#define CBC_DBLOOM_LOG2  (CBC_BLOOM_LOG2 - 1u)
#define CBC_DLDS_VHIST   (CBC_LDS_BLOOM + CBC_BLOOM_WORDS / 2u)   /* 128: histogram of a bucket's matching events by symbol, u16 x 2 per word */
#define CBC_DLDS_HIST    (CBC_PLAN_TABLE_WORDS + CBC_PLAN_DLDS_SCRATCH_WORDS)   /* 512: registered POS deltas per byte value, u16 x 2 per word */
#define CBC_DLDS_INS     (CBC_DLDS_HIST + 512u)        /* 256: (output index << 8) | base char      */
#define CBC_DLDS_FIXED   CBC_PLAN_DLDS_FIXED_WORDS

/* CBC_DSTAMP: diagnostic build only (cf. CBC_STAMP in cbc_encode_body.h): per-section s_memtime sums of the decode loop,
 * written over the start of the block's SEQ output -- outputs of such a build are garbage by design. */
#if defined(CBC_DSTAMP) && defined(__HIP_DEVICE_COMPILE__)
#define CBC_DT0() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(D.dt_last) :: "memory"); } while (0)
#define CBC_DT(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); D.dt_sum[k] += t_ - D.dt_last; D.dt_last = t_; } while (0)
#else
#define CBC_DT0() do {} while (0)
#define CBC_DT(k) do {} while (0)
#endif

struct cbc_dec_args {
    const uint8_t            *in;         /* payload bytes of all blocks                    */
    const cbc_dec_block_desc *blocks;
    const uint8_t            *ref;
    cbc_read_rec             *recs;       /* out: pos (block-local), flag, rlen, seq_off    */
    uint8_t                  *seq;        /* out: bases, seq_stride bytes reserved per read */
    cbc_block_result         *results;
    uint32_t                 *var_scratch; /* n_blocks * cap_var words                        */
    uint64_t in_bytes, ref_bytes, n_recs, seq_bytes, var_scratch_words;
    uint32_t n_blocks, cap_pos, cap_var;
};

CBC_FN uint32_t cbc_basechar(uint32_t b)             /* basepair2char sam_models.c:23-33 */
{
    return b == 0u ? 'A' : b == 1u ? 'C' : b == 2u ? 'G' : b == 3u ? 'T' : 'N';
}

/* GEN = false: the block decoder.  GEN = true: the whole-file stream decoder of cbc_stream_body.h (dense var table in
 * global memory, table pointers and capacities set by the caller, no closed forms). */
template <class W, bool GEN = false>
struct CbcDec {
    typedef typename W::V32 V32;
    typedef typename W::Mask Mask;

    /* ---- range decoder + bit reader ---- */
    typedef typename W::Uv Uv;
    Uv l, rng, d;                             /* lower bound, range = u - l + 1, tag - l: wave-uniform, see W::dv */
    uint64_t acc; uint32_t navail;
    V32 wordv; uint32_t widx, nwords_in, tail_valid; const uint8_t *inb;
    uint32_t status, nsym, fail_read, cur_read;

    /* ---- models (same representations as CbcEnc) ---- */
    V32 small, fkey, fexc, hkey, hexc, pval, pcnt;
    uint32_t fcount, fn, hc0, hc1, hc2, hc3, hn0, hn1, hn2, hn3;
    uint32_t *lds, *evp;
    uint32_t *rname_key, *rname_exc, *histp, *pos_valp, *pos_cntp, *vtab; uint32_t rn_cap;   /* set by the stream function */
    /* GEN (whole-file stream): the global-memory sides of flag, pos and pos_alpha -- see CbcEnc */
    uint32_t *fsp_key, *fsp_exc; uint32_t fsp_count;
    uint32_t *pos_ov_valp, *pos_ov_cntp; uint32_t pos_lds_cap;
    uint32_t *palpha; uint32_t pa_n0, pa_n1, pa_n2, pa_n3;
    uint32_t rlen_n, rl123_c0, rl123_n, snps_n, indels_n, rn_count, pos_card, pos_n, cap_pos, nev, nev1, cap_var, L0;
    uint32_t prevPos, prevM, prevChar;
    uint32_t rl_memo_x, rl_memo_lo, rl_memo_cnt, rl_last_x;
    V32 p0cnt; uint32_t p0over;
    CbcWin<W> win;
#if defined(CBC_DSTAMP) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long dt_last, dt_sum[16];
#endif

    CBC_MFN void fail(uint32_t st) { if (status == CBC_ST_OK) { status = st; fail_read = cur_read; } }
    CBC_MFN uint32_t *tab(uint32_t off) { return lds + off; }
    CBC_MFN uint32_t *pos_val_p() { return pos_valp; }
    CBC_MFN uint32_t *pos_cnt_p() { return pos_cntp; }
    CBC_MFN uint32_t *var_ev_p() { return evp; }

    /* ---- bit input: 64 big-endian words staged in a VGPR, refilled with one coalesced load ---- */
    CBC_MFN void load_chunk()
    {
        V32 ln = W::lane();
        V32 wi = ln + widx;
        V32 w = W::load32_bytes(inb, wi * 4u, wi < nwords_in);       /* past the end: zeros */
        /* the last word may be partial: bytes past the payload read as zero, like the reference's
         * zero-filled buffer past EOF (the bytes there belong to the next block's payload) */
        w = W::select((wi + 1u == nwords_in) & (W::splat(tail_valid) != 0u), w & ((1u << (tail_valid * 8u)) - 1u), w);
        wordv = w;
    }
    CBC_MFN uint32_t take(uint32_t k)                   /* next k <= 26 bits, MSB first */
    {
        if (k == 0u) return 0u;
        if (navail < k) {
            if ((widx & 63u) == 0u) load_chunk();
            uint32_t w = W::bswap32(W::readlane(wordv, widx & 63u));
            widx++;
            acc = (acc << 32) | (uint64_t)w; navail += 32u;
        }
        navail -= k;
        return (uint32_t)(acc >> navail) & ((1u << k) - 1u);
    }
    /* arithmetic_get_symbol_range, Arithmetic_stream.c:373-381 */
    CBC_MFN uint32_t target(uint32_t n)
    {
        if (n == 0u || W::dv_ge(d, rng)) { fail(CBC_ST_ASSERT); return 0u; }   /* gap = t - l + 1: 0 or beyond the range */
        uint64_t p = ((uint64_t)d + 1u) * n - 1u;
        /* back to a scalar register: the state lives in vector registers (W::dv), and a target left there would make the
         * symbol, the next context and every branch on them divergent in the compiler's eyes */
        return W::dv_scalar(W::divq(p, rng));
    }
    /* the E1/E2 and E3 shifts of arithmetic_decoder_step (Arithmetic_stream.c:401-454) in closed form and merged into one
     * shift, as in CbcEnc::code_tail.  The state is (l, range, d = tag - l) instead of the reference's (l, u, t): each of
     * the three scalings maps the interval with slope 2 and appends the next stream bit to the tag (E1: x -> 2x, E2:
     * x -> 2(x - 2^25), E3: x -> 2(x - 2^24), for l, u and t alike), so range' = range << sh and d' = d << sh | the next
     * sh bits -- one bit-reader call per step, and the models below read d and range as they are. */
    CBC_MFN void renorm()
    {
        const Uv uu = l + rng - 1u;
        const Uv x = l ^ uu;
        const Uv k1 = W::clz_uv((x << 6) | 32u);                    /* leading zeros of the 26-bit x; 26 when x = 0 */
        const Uv k3 = W::clz_uv((((~l | uu) << 7) | 127u) << k1);   /* leading ones of ((l & ~u) << 7) << k1 */
        const Uv sh = k1 + k3;
        if (W::dv_nz(sh)) {
            const uint32_t bits = take(W::dv_scalar(sh));
            l = (l << sh) & CBC_M25;
            rng = rng << sh;
            d = (d << sh) | bits;
        }
    }
    /* A symbol that can only be symbol 0 of its model (same_ref after record 0, rlength[1..3]; their
     * counts are closed forms of the record index because no rescale can happen below
     * CBC_MAX_BLOCK_READS): no target division and no search -- the symbol is 0 exactly when the tag lies
     * below the new upper bound l + floor(range * count0 / n) - 1, and `f` = floor(count0 * 2^32 / n) was
     * computed for 64 records at once (cf. CbcEnc::code1 / scaled_div). */
    CBC_MFN void step_known0(uint32_t cnt0, uint32_t n, uint32_t f)
    {
        nsym++;
        Uv q, plo;
        W::mul64(rng, f, q, plo);
        if (W::uv_ge(plo, 0xfc000000u)) q += (rng * cnt0 - q * n >= n) ? 1u : 0u;
        W::expect_eq(W::dv_scalar(q), (uint32_t)((uint64_t)W::dv_scalar(rng) * cnt0 / n), "step_known0 quotient");
        if (!W::dv_nz(q) || W::dv_ge(d, q)) { fail(CBC_ST_ASSERT); return; }   /* another symbol was coded here */
        rng = q;
        /* mostly nothing shifts after a symbol this probable: l < 2^25 between steps, so E1/E2 need u < 2^25 and E3
         * needs l >= 2^24 and u < 3 * 2^24 -- neither when q exceeds the bound below (cf. CbcEnc::step_known0) */
        if (W::dv_gt(q, ((l & (1u << 24)) | (1u << 25)) - l)) {
            W::expect_eq(((W::dv_scalar(l) ^ (W::dv_scalar(l) + W::dv_scalar(q) - 1u)) >> 25) & 1u, 1u, "step_known0: E1/E2 would shift");
            W::expect_eq((W::dv_scalar(l) >> 24) & (~(W::dv_scalar(l) + W::dv_scalar(q) - 1u) >> 24) & 1u, 0u, "step_known0: E3 would shift");
            return;
        }
        renorm();
    }
    /* arithmetic_decoder_step, Arithmetic_stream.c:389-454, loops in closed form (cf. CbcEnc::code1) */
    CBC_MFN void step(uint32_t lo, uint32_t cnt, uint32_t n)
    {
        if (cnt == 0u || n == 0u || lo + cnt > n) { fail(CBC_ST_ASSERT); return; }
        nsym++;
        uint32_t qh, ql;
        float inv = W::lane_float(W::recip_v(W::splat(n)), 0u);
        W::muldiv2(rng, lo, lo + cnt, n, inv, ql, qh);
        step_q0(ql, qh);
    }

    /* ---- search by scaled bounds ------------------------------------------------------------------------------
     * The decoder's symbol is the s with cum(s) <= target < cum(s + 1), target = floor(((t - l + 1) n - 1) / range)
     * (Arithmetic_stream.c:373-381).  For an integer C:  target < C  <=>  (t - l + 1) n <= C range  <=>
     * t - l < floor(range C / n).  So with a model's cumulative counts laid out one per lane, the symbol is the lane
     * whose scaled bounds floor(range cum / n) enclose t - l -- and those two quotients are what the coder step needs
     * next (:389-400): no target division, no search over counts, no second pair of divisions, and the 64 divisions
     * are one vector instruction stream (W::muldiv_v) with no trip through scalar registers.  The models below fall
     * back to target() + search when the tag lies outside the lanes they hold. */
    CBC_MFN bool tag_ok(uint32_t n)                       /* the checks of target() */
    {
        if (n == 0u || W::dv_ge(d, rng)) { fail(CBC_ST_ASSERT); return false; }
        return true;
    }
    CBC_MFN void step_q0(uint32_t ql, uint32_t qh) { l += ql; d -= ql; rng = W::dv(qh - ql); renorm(); }
    CBC_MFN void step_q(uint32_t ql, uint32_t qh) { nsym++; step_q0(ql, qh); }
    /* cum_incl: inclusive cumulative count per lane, non-decreasing over the live lanes [first, first + m), 0 elsewhere */
    CBC_MFN bool prefix_find(V32 cum_incl, Mask live, uint32_t first, uint32_t n, uint32_t &idx, uint32_t &ql, uint32_t &qh)
    {
        const V32 qv = W::muldiv_v(rng, cum_incl, n);
        const uint64_t hb = W::ballot(live & (W::dvv(d) < qv));
        if (!hb) return false;
        const uint32_t hl = W::ctz64(hb);
        qh = W::readlane(qv, hl); ql = hl != first ? W::readlane(qv, hl - 1u) : 0u;
        idx = hl - first;
        return true;
    }

    /* ---- lane-table literal models (match, same_ref, chars) ---- */
    /* a two-symbol literal-count model (match, same_ref): no target division and no search.  The tag lies in
     * symbol 0's interval exactly when t - l < floor(range * count0 / n)  (the same test the target of
     * Arithmetic_stream.c:373-381 would give: floor(((t-l+1) n - 1) / range) < count0), and symbol 1's
     * upper bound is the old one, so one division decides and performs the step. */
    CBC_MFN uint32_t bin_dec(uint32_t base, uint32_t stp)
    {
        const uint32_t c0 = W::readlane(small, base), c1 = W::readlane(small, base + 1u), n = c0 + c1;
        if (c0 == 0u || c1 == 0u) { fail(CBC_ST_ASSERT); return 0u; }
        nsym++;
        uint32_t q0, qn;
        float inv = W::lane_float(W::recip_v(W::splat(n)), 0u);
        W::muldiv2(rng, c0, n, n, inv, q0, qn);
        const uint32_t x = W::dv_ge(d, q0) ? 1u : 0u;
        if (x == 0u) rng = W::dv(q0); else { l += q0; d -= q0; rng -= q0; }      /* symbol 1 keeps the old upper bound */
        renorm();
        V32 ln = W::lane();
        small = W::select(ln == base + x, small + stp, small);
        if (n + stp >= CBC_RESCALE) {
            Mask m = (ln >= base) & (ln < base + 2u);
            small = W::select(m, (small >> 1) + 1u, small);
        }
        return x;
    }
    CBC_MFN uint32_t small_dec(uint32_t base, uint32_t card, uint32_t stp)
    {
        if (card == 2u) return bin_dec(base, stp);
        V32 ln = W::lane();
        const Mask live = (ln >= base) & (ln < base + card);
        const V32 inc = W::scan_incl_add(W::select(live, small, W::splat(0u)));
        const uint32_t n = W::readlane(inc, base + card - 1u);
        uint32_t x, ql, qh;
        if (!tag_ok(n)) return 0u;
        if (!prefix_find(inc, live, base, n, x, ql, qh)) { fail(CBC_ST_ASSERT); return 0u; }
        step_q(ql, qh);
        small = W::select(ln == base + x, small + stp, small);
        if (n + stp >= CBC_RESCALE) small = W::select(live, (small >> 1) + 1u, small);
        return x;
    }

    /* ---- generic search over (key, excess) pairs held one per lane ---- */
    CBC_MFN uint32_t pairs_search(V32 key, V32 exc, Mask live, uint32_t first, uint32_t count, uint32_t tg,
                                  uint32_t &lo, uint32_t &cnt, uint32_t &hit_lane, bool &hit)
    {
        /* A = cum(key) = key + sum of the excess of all smaller keys */
        V32 A = key;
        for (uint32_t j = 0; j < count; j++) {
            uint32_t kj = W::readlane(key, first + j), ej = W::readlane(exc, first + j);
            A = A + W::select(key > kj, W::splat(ej), W::splat(0u));
        }
        uint64_t hb = W::ballot(live & (A <= tg) & (W::splat(tg) - A <= exc));
        if (hb) {
            hit = true; hit_lane = W::ctz64(hb);
            lo = W::readlane(A, hit_lane); cnt = 1u + W::readlane(exc, hit_lane);
            return W::readlane(key, hit_lane);
        }
        hit = false; hit_lane = 0;
        uint32_t S = W::reduce_add(W::select(live & (A < tg), exc, W::splat(0u)));   /* A + e < tg here */
        lo = tg; cnt = 1u;
        return tg - S;
    }

    /* ---- register-resident sparse model (flag, codebook) ---- */
    /* a key seen before, alone: true and the symbol in `x`; on false nothing has changed and regsparse_dec() decides */
    CBC_MFN bool regsparse_fast(const V32 &key, V32 &exc, uint32_t base, uint32_t count, uint32_t &n, uint32_t card, uint32_t stp, uint32_t &x)
    {
        if (n == 0u || W::dv_ge(d, rng) || n + stp >= CBC_RESCALE) return false;
        V32 ln = W::lane();
        const Mask live = (ln >= base) & (ln < base + count);
        V32 A = key;
        for (uint32_t j = 0; j < count; j++) {
            uint32_t kj = W::readlane(key, base + j), ej = W::readlane(exc, base + j);
            A = A + W::select(key > kj, W::splat(ej), W::splat(0u));
        }
        const V32 qlv = W::muldiv_v(rng, W::select(live, A, W::splat(0u)), n);
        const V32 qhv = W::muldiv_v(rng, W::select(live, A + 1u + exc, W::splat(0u)), n);
        const uint64_t hb = W::ballot(live & (qlv <= W::dvv(d)) & (W::dvv(d) < qhv));
        if (!hb) return false;
        const uint32_t hl = W::ctz64(hb);
        x = W::readlane(key, hl);
        if (x >= card) return false;
        step_q(W::readlane(qlv, hl), W::readlane(qhv, hl));
        exc = W::select(ln == hl, exc + stp, exc);
        n += stp;
        return true;
    }
    CBC_MFN uint32_t regsparse_dec(V32 &key, V32 &exc, uint32_t base, uint32_t cap, uint32_t &count, uint32_t &n,
                                   uint32_t card, uint32_t stp, uint32_t cap_status)
    {
        V32 ln = W::lane();
        Mask live = (ln >= base) & (ln < base + count);
        if (!tag_ok(n)) return 0u;
        /* A = cum(key) = key + the excess of all smaller keys; a seen key owns [A, A + 1 + excess) */
        V32 A = key;
        for (uint32_t j = 0; j < count; j++) {
            uint32_t kj = W::readlane(key, base + j), ej = W::readlane(exc, base + j);
            A = A + W::select(key > kj, W::splat(ej), W::splat(0u));
        }
        const uint32_t range = rng;
        const V32 qlv = W::muldiv_v(range, W::select(live, A, W::splat(0u)), n);
        const V32 qhv = W::muldiv_v(range, W::select(live, A + 1u + exc, W::splat(0u)), n);
        const uint64_t hb = W::ballot(live & (qlv <= W::dvv(d)) & (W::dvv(d) < qhv));
        uint32_t x, hl = 0; const bool hit = hb != 0ull;
        if (hit) {
            hl = W::ctz64(hb); x = W::readlane(key, hl);
            if (x >= card) { fail(CBC_ST_ASSERT); return 0u; }
            step_q(W::readlane(qlv, hl), W::readlane(qhv, hl));
        } else {                                             /* an unseen symbol: count 1 at cum = target */
            const uint32_t tg = target(n);
            const uint32_t S = W::reduce_add(W::select(live & (A < tg), exc, W::splat(0u)));   /* A + e < tg here */
            x = tg - S;
            if (x >= card) { fail(CBC_ST_ASSERT); return 0u; }
            step(tg, 1u, n);
        }
        if (hit) exc = W::select(ln == hl, exc + stp, exc);
        else {
            if (count >= cap) { fail(cap_status); return 0u; }
            uint32_t idx = base + count;
            key = W::select(ln == idx, W::splat(x), key);
            exc = W::select(ln == idx, W::splat(stp), exc);
            count++;
        }
        n += stp;
        if (n >= CBC_RESCALE) {
            live = (ln >= base) & (ln < base + count);
            exc = W::select(live, (exc + 1u) >> 1, exc);
            n = card + W::reduce_add(W::select(live, exc, W::splat(0u)));
        }
        return x;
    }

    /* ---- dense-excess tables in LDS: 4 consecutive symbols per lane, one wave scan ---- */
    /* symbols 4*lane .. 4*lane+3 with excess e0..e3: find the one whose interval holds tg */
    /* ---- flag in its general form (decompress_flag read_decompression.c:120-139; CbcEnc::flag_gen_code): register pairs
     * + pairs in global memory.  A value seen before, held in registers, with nothing spilled: the search by scaled
     * bounds (regsparse_fast).  Anything else: cum(s) = s + the excess of all smaller values seen is increasing in s, so the
     * symbol is found by bisection over s on the target (17 evaluations of cum, each one pass over the pairs). ---- */
    CBC_MFN uint32_t flag_cum(uint32_t sv)
    {
        const V32 ln = W::lane();
        V32 acc = W::select((ln < fcount) & (fkey < sv), fexc, W::splat(0u));
        const uint32_t nb = W::uni(fsp_count);
        for (uint32_t b = 0; b < nb; b += 64u) {
            const V32 i = ln + b; const Mask m = i < fsp_count;
            const V32 k = W::load32_list(fsp_key, i, m, 0xffffffffu), e = W::load32_list(fsp_exc, i, m, 0u);
            acc = acc + W::select(m & (k < sv), e, W::splat(0u));
        }
        return sv + W::reduce_add(acc);
    }
    CBC_MFN uint32_t flag_gen_dec()
    {
        uint32_t x = 0;
        if (fsp_count == 0u && regsparse_fast(fkey, fexc, 0u, fcount, fn, 65536u, 8u, x)) return x;
        const V32 ln = W::lane();
        const uint32_t tg = target(fn);
        if (status != CBC_ST_OK) return 0u;
        if (fsp_count) W::list_fence();
        uint32_t lo_s = 0u, hi_s = 65535u;
        while (lo_s < hi_s) {                                 /* the largest s with cum(s) <= target */
            const uint32_t mid = (lo_s + hi_s + 1u) >> 1;
            if (flag_cum(mid) <= tg) lo_s = mid; else hi_s = mid - 1u;
        }
        x = lo_s;
        const uint32_t cum = flag_cum(x);
        uint32_t cnt = 1u, idx = 0, sp_idx = CBC_NOMEMO;
        const uint64_t eq = W::ballot((ln < fcount) & (fkey == x));
        if (eq) { idx = W::ctz64(eq); cnt = 1u + W::readlane(fexc, idx); }
        else {
            const uint32_t nb = W::uni(fsp_count);
            for (uint32_t b = 0; b < nb; b += 64u) {
                const V32 i = ln + b; const Mask m = i < fsp_count;
                const V32 k = W::load32_list(fsp_key, i, m, 0xffffffffu);
                const uint64_t hit = W::ballot(m & (k == x));
                if (hit) { sp_idx = b + W::ctz64(hit); cnt = 1u + W::readlane(W::load32_list(fsp_exc, W::splat(sp_idx), W::all(), 0u), 0u); }
            }
        }
        if (tg < cum || tg - cum >= cnt) { fail(CBC_ST_ASSERT); return 0u; }
        step(cum, cnt, fn);
        if (eq) fexc = W::select(ln == idx, fexc + 8u, fexc);
        else if (sp_idx != CBC_NOMEMO) W::append_list(fsp_exc, sp_idx, cnt - 1u + 8u);
        else if (fcount < CBC_CAP_FLAG) {
            fkey = W::select(ln == fcount, W::splat(x), fkey);
            fexc = W::select(ln == fcount, W::splat(8u), fexc);
            fcount++;
        } else {
            if (fsp_key == nullptr || fsp_count >= 65536u) { fail(CBC_ST_CAP_FLAG); return 0u; }
            W::append_list(fsp_key, fsp_count, x); W::append_list(fsp_exc, fsp_count, 8u);
            fsp_count++;
        }
        fn += 8u;
        if (fn >= CBC_RESCALE) {
            const Mask live = ln < fcount;
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
            fn = 65536u + W::reduce_add(a);
        }
        return x;
    }

    CBC_MFN uint32_t search4(V32 e0, V32 e1, V32 e2, V32 e3, uint32_t card, uint32_t tg, uint32_t &lo, uint32_t &cnt)
    {
        V32 s0 = W::lane() * 4u;
        V32 tot = e0 + e1 + e2 + e3;
        V32 c0 = s0 + (W::scan_incl_add(tot) - tot);          /* cum(4*lane) */
        V32 c1 = c0 + 1u + e0, c2 = c1 + 1u + e1, c3 = c2 + 1u + e2, c4 = c3 + 1u + e3;
        uint64_t h0 = W::ballot((s0 < card) & (c0 <= tg) & (tg < c1));
        uint64_t h1 = W::ballot(((s0 + 1u) < card) & (c1 <= tg) & (tg < c2));
        uint64_t h2 = W::ballot(((s0 + 2u) < card) & (c2 <= tg) & (tg < c3));
        uint64_t h3 = W::ballot(((s0 + 3u) < card) & (c3 <= tg) & (tg < c4));
        uint32_t x;
        if (h0) { uint32_t hl = W::ctz64(h0); x = hl * 4u; lo = W::readlane(c0, hl); cnt = 1u + W::readlane(e0, hl); }
        else if (h1) { uint32_t hl = W::ctz64(h1); x = hl * 4u + 1u; lo = W::readlane(c1, hl); cnt = 1u + W::readlane(e1, hl); }
        else if (h2) { uint32_t hl = W::ctz64(h2); x = hl * 4u + 2u; lo = W::readlane(c2, hl); cnt = 1u + W::readlane(e2, hl); }
        else if (h3) { uint32_t hl = W::ctz64(h3); x = hl * 4u + 3u; lo = W::readlane(c3, hl); cnt = 1u + W::readlane(e3, hl); }
        else { fail(CBC_ST_ASSERT); x = 0; lo = 0; cnt = 1; }
        return x;
    }
    CBC_MFN uint32_t dense_search(const uint32_t *exc, uint32_t card, uint32_t mul, uint32_t tg, uint32_t &lo, uint32_t &cnt)
    {
        V32 s0 = W::lane() * 4u;
        V32 e0 = W::load32(exc, s0, s0 < card, 0u) * mul, e1 = W::load32(exc, s0 + 1u, (s0 + 1u) < card, 0u) * mul;
        V32 e2 = W::load32(exc, s0 + 2u, (s0 + 2u) < card, 0u) * mul, e3 = W::load32(exc, s0 + 3u, (s0 + 3u) < card, 0u) * mul;
        return search4(e0, e1, e2, e3, card, tg, lo, cnt);
    }
    CBC_MFN void dense_rescale(uint32_t *exc, uint32_t card, uint32_t &n)
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
    /* rlength[0]: the same length nearly always repeats.  The previous symbol's (cum, count) are kept in
     * scalars; the guess is verified with the two divisions the step needs anyway (the tag must lie in
     * [l + floor(range cum / n), l + floor(range (cum + count) / n))), which replaces the target division
     * and the table search.  On a miss the count is written back and the generic decode runs. */
    /* the verified guess alone: true and the length in `x` when the tag lies in the remembered symbol's interval; the
     * state is untouched otherwise and the caller goes through rlen_dec() (outside the record loop) */
    CBC_MFN bool rlen_fast(uint32_t &x)
    {
        if (rl_memo_x == CBC_NOMEMO || rlen_n + 10u >= CBC_RESCALE) return false;
        uint32_t ql, qh;
        float inv = W::lane_float(W::recip_v(W::splat(rlen_n)), 0u);
        W::muldiv2(rng, rl_memo_lo, rl_memo_lo + rl_memo_cnt, rlen_n, inv, ql, qh);
        if (!(W::dv_ge(d, ql) && W::dv_gt(qh, d))) return false;
        step_q(ql, qh);
        rl_memo_cnt += 10u; rlen_n += 10u;
        x = rl_memo_x;
        return true;
    }
    CBC_MFN uint32_t rlen_dec()
    {
        uint32_t *exc = tab(CBC_LDS_RLEN);
        if (rl_memo_x != CBC_NOMEMO) {
            uint32_t ql, qh;
            float inv = W::lane_float(W::recip_v(W::splat(rlen_n)), 0u);
            W::muldiv2(rng, rl_memo_lo, rl_memo_lo + rl_memo_cnt, rlen_n, inv, ql, qh);
            if (W::dv_ge(d, ql) && W::dv_gt(qh, d)) {
                step_q(ql, qh);
                rl_memo_cnt += 10u; rlen_n += 10u;
                if (rlen_n >= CBC_RESCALE) {                   /* through the table: unreachable below CBC_MAX_BLOCK_READS */
                    W::write_uni(exc, rl_memo_x, rl_memo_cnt - 1u);
                    dense_rescale(exc, 255u, rlen_n);
                    rl_memo_x = CBC_NOMEMO;
                }
                return rl_memo_x != CBC_NOMEMO ? rl_memo_x : rl_last_x;
            }
            W::write_uni(exc, rl_memo_x, rl_memo_cnt - 1u);
            rl_memo_x = CBC_NOMEMO;
        }
        uint32_t tg = target(rlen_n), lo, cnt;
        uint32_t x = dense_search(exc, 255u, 1u, tg, lo, cnt);
        if (status != CBC_ST_OK) return 0u;
        step(lo, cnt, rlen_n);
        rlen_n += 10u;
        if (rlen_n >= CBC_RESCALE) { W::write_uni(exc, x, cnt - 1u + 10u); dense_rescale(exc, 255u, rlen_n); }
        else { rl_memo_x = x; rl_memo_lo = lo; rl_memo_cnt = cnt + 10u; }
        rl_last_x = x;
        return x;
    }
    /* the same for a table whose symbols are small in practice (the SNP / indel counts of a read): symbols 0..63 one
     * per lane and the search by scaled bounds; any other symbol through dense_dec() */
    CBC_MFN uint32_t dense_dec_low(uint32_t *exc, uint32_t card, uint32_t stp, uint32_t &n)
    {
        if (!tag_ok(n)) return 0u;
        V32 ln = W::lane();
        const Mask live = ln < card;
        const V32 e = W::load32(exc, ln, live, 0u);
        const V32 cum = W::select(live, W::scan_incl_add(e) + ln + 1u, W::splat(0u));
        uint32_t x, ql, qh;
        if (!prefix_find(cum, live, 0u, n, x, ql, qh)) return dense_dec(exc, card, stp, n);
        step_q(ql, qh);
        W::write_uni(exc, x, W::readlane(e, x) + stp);
        n += stp;
        if (n >= CBC_RESCALE) dense_rescale(exc, card, n);
        return x;
    }
    CBC_MFN uint32_t dense_dec(uint32_t *exc, uint32_t card, uint32_t stp, uint32_t &n)
    {
        uint32_t tg = target(n), lo, cnt;
        uint32_t x = dense_search(exc, card, 1u, tg, lo, cnt);
        if (status != CBC_ST_OK) return 0u;
        step(lo, cnt, n);
        W::write_uni(exc, x, cnt - 1u + stp);
        n += stp;
        if (n >= CBC_RESCALE) dense_rescale(exc, card, n);
        return x;
    }

    /* ---- rname: gather the context's pairs into lanes, then the generic search ---- */
    CBC_MFN uint32_t rname_dec(uint32_t ctx)
    {
        V32 ln = W::lane();
        uint32_t *rkey = rname_key, *rexc = rname_exc;
        V32 gk = W::splat(0u), ge = W::splat(0u), gi = W::splat(0u);
        uint32_t m = 0, nsum = 0;
        const uint32_t rb = W::uni(rn_count);
        for (uint32_t b = 0; b < rb; b += 64u) {
            V32 i = ln + b; Mask mm = i < rn_count;
            V32 k = W::load32(rkey, i, mm, 0xffffffffu);
            V32 e = W::load32(rexc, i, mm, 0u);
            uint64_t bb = W::ballot(mm & ((k >> 8) == ctx));
            while (bb) {
                uint32_t src = W::ctz64(bb); bb &= bb - 1u;
                uint32_t kk = W::readlane(k, src) & 0xffu, ee = W::readlane(e, src);
                if (m < 64u) {
                    gk = W::select(ln == m, W::splat(kk), gk); ge = W::select(ln == m, W::splat(ee), ge);
                    gi = W::select(ln == m, W::splat(b + src), gi);
                }
                m++; nsum += ee;
            }
        }
        if (m > 64u) { fail(CBC_ST_CAP_NAME); return 0u; }
        uint32_t n = 256u + nsum;
        if (n + 10u >= CBC_RESCALE) { fail(CBC_ST_ASSERT); return 0u; }
        uint32_t tg = target(n), lo, cnt, hl; bool hit;
        uint32_t x = pairs_search(gk, ge, ln < m, 0u, m, tg, lo, cnt, hl, hit);
        if (x >= 256u) { fail(CBC_ST_ASSERT); return 0u; }
        step(lo, cnt, n);
        if (hit) W::write_uni(rexc, W::readlane(gi, hl), cnt - 1u + 10u);
        else {
            if (rn_count >= rn_cap) { fail(CBC_ST_CAP_NAME); return 0u; }
            W::write_uni(rkey, rn_count, (ctx << 8) | x);
            W::write_uni(rexc, rn_count, 10u);
            rn_count++;
        }
        return x;
    }

    /* ---- pos: literal counts, two tiers (cf. CbcEnc::pos_code); entries from pos_lds_cap on live in the global overflow
     * arrays (CbcEnc::ptab_*) ---- */
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
    CBC_MFN void pos_rescale()
    {
        V32 ln = W::lane();
        Mask m0 = ln < (pos_card < 64u ? pos_card : 64u);
        pcnt = W::select(m0, (pcnt >> 1) + 1u, pcnt);
        V32 a = W::select(m0, pcnt, W::splat(0u));
        const uint32_t pc = W::uni(pos_card);
        if (pos_card > pos_lds_cap) W::list_fence();
        for (uint32_t b = 64u; b < pc; b += 64u) {
            V32 i = ln + b; Mask m = i < pos_card;
            V32 c = (ptab_ld(pos_cnt_p(), pos_ov_cntp, b, i, m, 0u) >> 1) + 1u;
            ptab_st(pos_cnt_p(), pos_ov_cntp, b, i, c, m);
            a = a + W::select(m, c, W::splat(0u));
        }
        pos_n = W::reduce_add(a);
    }
    CBC_MFN void pos_update(uint32_t idx)
    {
        if (idx < 64u) pcnt = W::select(W::lane() == idx, pcnt + 10u, pcnt);
        else ptab_wr(pos_cnt_p(), pos_ov_cntp, idx, ptab_rd(pos_cnt_p(), pos_ov_cntp, idx) + 10u);
        pos_n += 10u;
        if (pos_n >= CBC_RESCALE) pos_rescale();
    }
    CBC_MFN uint32_t pos_alpha_dec(uint32_t k)           /* one byte of decompress_pos_alpha :144-181 */
    {
        if (palpha) {                                        /* the general form: four dense models with their rescale */
            uint32_t *t = palpha + 256u * k;
            return k == 0u ? dense_dec(t, 256u, 10u, pa_n0) : k == 1u ? dense_dec(t, 256u, 10u, pa_n1)
                 : k == 2u ? dense_dec(t, 256u, 10u, pa_n2) : dense_dec(t, 256u, 10u, pa_n3);
        }
        uint32_t n = 256u + 10u * (pos_card - 1u);
        if (n + 10u >= CBC_RESCALE) { fail(CBC_ST_ASSERT); return 0u; }
        uint32_t tg = target(n), lo, cnt;
        if (histp == nullptr) {
            /* no histograms (the long-read decoder: at most 64 reads, hence at most 64 registered deltas, per block): the
             * registered values one per lane, count(b) = 1 + 10 * #{values whose byte k is b}, cum(b) = b + 10 * #{... below b};
             * cum is increasing, so the symbol comes out of eight bisection steps of one ballot each */
            if (status != CBC_ST_OK) return 0u;
            if (pos_card > 65u) { fail(CBC_ST_CAP_POS); return 0u; }
            V32 ln = W::lane();
            const Mask have = (ln + 1u) < pos_card;                          /* lane i: alphabet entry i + 1 */
            V32 v = W::lane_gather(pval, ln + 1u);
            if (pos_card == 65u) v = W::select(ln == 63u, W::splat(W::read_uni(pos_val_p(), 64u)), v);
            const V32 by = (v >> (8u * (3u - k))) & 0xffu;
            uint32_t lo_b = 0u, hi_b = 255u;
            while (lo_b < hi_b) {
                const uint32_t mid = (lo_b + hi_b + 1u) >> 1;
                if (mid + 10u * W::popc64(W::ballot(have & (by < mid))) <= tg) lo_b = mid; else hi_b = mid - 1u;
            }
            lo = lo_b + 10u * W::popc64(W::ballot(have & (by < lo_b)));
            cnt = 1u + 10u * W::popc64(W::ballot(have & (by == lo_b)));
            if (tg < lo || tg - lo >= cnt) { fail(CBC_ST_ASSERT); return 0u; }
            step(lo, cnt, n);
            return lo_b;
        }
        /* symbols 4*lane..4*lane+3 of context k = two words of u16 counts; excess = 10 per registered delta */
        const uint32_t *h = histp + 128u * k;
        V32 ln = W::lane();
        V32 wa = W::load32(h, ln * 2u, W::all(), 0u), wb = W::load32(h, ln * 2u + 1u, W::all(), 0u);
        uint32_t x = search4((wa & 0xffffu) * 10u, (wa >> 16) * 10u, (wb & 0xffffu) * 10u, (wb >> 16) * 10u, 256u, tg, lo, cnt);
        if (status != CBC_ST_OK) return 0u;
        step(lo, cnt, n);
        return x;
    }
    CBC_MFN void hist_inc(uint32_t k, uint32_t b)
    {
        if (palpha || histp == nullptr) return;               /* dense tables carry their own counts; the lane form derives them */
        uint32_t *h = histp + 128u * k;
        W::write_uni(h, b >> 1, W::read_uni(h, b >> 1) + (1u << ((b & 1u) * 16u)));
    }
    /* the POS symbol: its alphabet index (0 = escape), model updated; CBC_NOMEMO when the decode failed */
    CBC_MFN uint32_t pos_sym()
    {
        V32 ln = W::lane();
        if (!tag_ok(pos_n)) return CBC_NOMEMO;
        Mask m0 = ln < (pos_card < 64u ? pos_card : 64u);
        V32 c = W::select(m0, pcnt, W::splat(0u));
        V32 inc = W::scan_incl_add(c);
        uint32_t idx = 0, ql, qh;
        if (prefix_find(inc, m0, 0u, pos_n, idx, ql, qh)) step_q(ql, qh);       /* the first 64 entries: lanes */
        else {
            const uint32_t tg = target(pos_n);
            uint32_t lo = 0, cnt = 0, found = 0;
            uint32_t base_sum = W::readlane(inc, 63u);
            const uint32_t pc = W::uni(pos_card);
            if (pos_card > pos_lds_cap) W::list_fence();
            for (uint32_t b = 64u; b < pc && !found; b += 64u) {
                V32 i = ln + b; Mask m = i < pos_card;
                V32 cc = ptab_ld(pos_cnt_p(), pos_ov_cntp, b, i, m, 0u);
                V32 ic = W::scan_incl_add(cc) + base_sum;
                uint64_t h2 = W::ballot(m & ((ic - cc) <= tg) & (tg < ic));
                if (h2) { uint32_t hl = W::ctz64(h2); idx = b + hl; lo = W::readlane(ic, hl) - W::readlane(cc, hl); cnt = W::readlane(cc, hl); found = 1; }
                base_sum = W::readlane(ic, 63u);
            }
            if (!found) { fail(CBC_ST_ASSERT); return CBC_NOMEMO; }
            step(lo, cnt, pos_n);
        }
        pos_update(idx);
        return idx;
    }
    CBC_MFN uint32_t pos_value(uint32_t idx) { return idx < 64u ? W::readlane(pval, idx) : ptab_rd(pos_val_p(), pos_ov_valp, idx); }
    /* after an escape: the four bytes of the new delta, which joins the alphabet (decompress_pos :126-141) */
    CBC_MFN uint32_t pos_escape()
    {
        V32 ln = W::lane();
        uint32_t b3 = pos_alpha_dec(0u), b2 = pos_alpha_dec(1u), b1 = pos_alpha_dec(2u), b0 = pos_alpha_dec(3u);
        uint32_t x = (b3 << 24) | (b2 << 16) | (b1 << 8) | b0;
        if (status != CBC_ST_OK) return 0u;
        if (pos_card >= cap_pos) { fail(CBC_ST_CAP_POS); return 0u; }
        hist_inc(0u, b3); hist_inc(1u, b2); hist_inc(2u, b1); hist_inc(3u, b0);
        if (pos_card < 64u) {
            pval = W::select(ln == pos_card, W::splat(x), pval);
            pcnt = W::select(ln == pos_card, W::splat(0u), pcnt);
        } else {
            ptab_wr(pos_val_p(), pos_ov_valp, pos_card, x);
            ptab_wr(pos_cnt_p(), pos_ov_cntp, pos_card, 0u);
        }
        pos_card++;
        pos_update(pos_card - 1u);
        return x;
    }
    CBC_MFN uint32_t pos_dec()                           /* returns x = delta + 1 */
    {
        const uint32_t idx = pos_sym();
        if (idx == CBC_NOMEMO) return 0u;
        return idx != 0u ? pos_value(idx) : pos_escape();
    }

    /* ---- var: Bloom filter, then gather the context's events (each worth 10) into lanes ---- */
    /* GEN: the dense table of CbcEnc::var_code_dense, searched: lane l takes symbols 4l .. 4l+3 of the row */
    CBC_MFN uint32_t var_dec_dense(uint32_t ctx)
    {
        V32 ln = W::lane();
        uint32_t *row = vtab + (uint64_t)ctx * L0;
        W::list_fence();
        V32 s0 = ln * 4u;
        V32 e0 = W::load32_list(row, s0, s0 < L0, 0u), e1 = W::load32_list(row, s0 + 1u, (s0 + 1u) < L0, 0u);
        V32 e2 = W::load32_list(row, s0 + 2u, (s0 + 2u) < L0, 0u), e3 = W::load32_list(row, s0 + 3u, (s0 + 3u) < L0, 0u);
        const uint32_t n = L0 + W::reduce_add(e0 + e1 + e2 + e3);
        uint32_t tg = target(n), lo, cnt;
        if (status != CBC_ST_OK) return 0u;
        uint32_t x = search4(e0, e1, e2, e3, L0, tg, lo, cnt);
        if (status != CBC_ST_OK) return 0u;
        step(lo, cnt, n);
        W::append_list(row, x, cnt - 1u + 10u);
        if (n + 10u >= CBC_RESCALE) {
            W::list_fence();
            for (uint32_t q = 0; q < 4u; q++) {
                V32 i = ln + 64u * q;
                V32 e = W::load32_list(row, i, i < L0, 0u);
                W::store32_list(row, i, (e + 1u) >> 1, i < L0);
            }
        }
        return x;
    }
    CBC_MFN uint32_t var_dec(uint32_t ctx)
    {
        CbcDec &D = *this; (void)D;                          /* for the diagnostic stamps */
        if (ctx >= CBC_NVARCTX) { fail(CBC_ST_ASSERT); return 0u; }
        if (GEN) return var_dec_dense(ctx);
        /* the two context classes of CbcEnc::var_code: "p = 0" contexts (ctx = d << 8 | strand) keep their 16-bit
         * events in an LDS array per strand, the others sit behind the Bloom filter in the global list; every
         * event of the context adds 10 to its symbol's excess (4 symbols per lane), then one search */
        V32 ln = W::lane();
        uint32_t *bloom = tab(CBC_LDS_BLOOM), *ev = var_ev_p();
        const uint32_t strand1 = ctx & 1u;
        /* flags as 0 / 1 words, not bool: a bool that lives across blocks is kept as a 64-bit lane mask */
        const uint32_t p0class = (((ctx >> 1) & 0x7fu) == 0u && (ctx >> 8) != 255u) ? 1u : 0u;   /* 255 is the unused-half marker */
        uint32_t to_global = p0class ^ 1u;
        V32 e0 = W::splat(0u), e1 = W::splat(0u), e2 = W::splat(0u), e3 = W::splat(0u);
        uint32_t m = 0;
        const uint32_t bkt = strand1 * 8u + ((ctx >> 8) & 7u);
        uint32_t have = 0, half_word = 0;                      /* the bucket's last word when its upper half is free */
        if (p0class) {
            const uint32_t d = ctx >> 8;
            const uint32_t *arr = tab(CBC_LDS_P0) + bkt * CBC_P0_BUCKET_WORDS;
            have = W::readlane(p0cnt, bkt);
            const uint32_t nw = (have + 1u) >> 1;
            const V32 w = W::load32(arr, ln, ln < nw, 0xffffffffu);       /* the whole bucket in one load */
            if (have & 1u) half_word = W::readlane(w, have >> 1);         /* saves the read of the read-modify-write below */
            /* the context's events, by symbol: every matching lane adds its event to a 256-bin u16 histogram in LDS (DS
             * atomics; the wavefront's DS operations complete in order), read back four bins per lane.  The serial form of
             * this -- one trip per matching event -- was the decoder's largest single item (profiles/r02_ab_kernels.log). */
            const V32 ea = w & 0xffffu, eb = w >> 16;
            const Mask ha = (ea >> 8) == d, hb = (eb >> 8) == d;        /* an unused half holds 0xffff: d = 255 is not in the class */
            m = W::popc64(W::ballot(ha)) + W::popc64(W::ballot(hb));
            if (m) {
                uint32_t *vh = tab(CBC_DLDS_VHIST);
                W::store32(vh, ln * 2u, W::splat(0u), W::all()); W::store32(vh, ln * 2u + 1u, W::splat(0u), W::all());
                W::lds_add(vh, (ea & 0xffu) >> 1, W::splat(1u) << ((ea & 1u) * 16u), ha);
                W::lds_add(vh, (eb & 0xffu) >> 1, W::splat(1u) << ((eb & 1u) * 16u), hb);
                const V32 c01 = W::load32(vh, ln * 2u, W::all(), 0u), c23 = W::load32(vh, ln * 2u + 1u, W::all(), 0u);
                e0 = (c01 & 0xffffu) * 10u; e1 = (c01 >> 16) * 10u; e2 = (c23 & 0xffffu) * 10u; e3 = (c23 >> 16) * 10u;
            }
            if (have >= CBC_P0_CAP) to_global = 1u;
        }
        CBC_DT(11);                                           /* var_dec: class, bucket load, its events */
        uint32_t h1 = 0, h2 = 0, bw1 = 0, bw2 = 0, bb1 = 0, bb2 = 0;
        if ((to_global | ((p0over >> bkt) & 1u)) != 0u) {
            /* the encoder's filter: two hash functions, both words fetched by one LDS instruction */
            h1 = (ctx * 0x9E3779B1u) >> (32u - CBC_DBLOOM_LOG2); h2 = (ctx * 0x85EBCA6Bu + 0x27D4EB2Fu) >> (32u - CBC_DBLOOM_LOG2);
            const V32 bwv = W::load32(bloom, W::select(ln == 0u, W::splat(h1 >> 5), W::splat(h2 >> 5)), ln < 2u, 0u);
            bw1 = W::readlane(bwv, 0u); bw2 = W::readlane(bwv, 1u);
            bb1 = 1u << (h1 & 31u); bb2 = 1u << (h2 & 31u);
            if ((bw1 & bb1) && (bw2 & bb2)) {
                W::list_fence();
                /* two lists by the context's strand bit, one from each end of the area (cf. CbcEnc::var_code) */
                const uint32_t cnt_s = strand1 ? nev1 : nev, base_s = strand1 ? cap_var - nev1 : 0u;
                const uint32_t nb = W::uni(cnt_s);
                for (uint32_t b = 0; b < nb; b += 512u) {             /* eight coalesced loads in flight per trip */
                    V32 ev4[8];
                    for (uint32_t q = 0; q < 8u; q++) { V32 i = ln + (b + 64u * q); ev4[q] = W::load32_list(ev, i + base_s, i < cnt_s, 0xffffffffu); }
                    for (uint32_t q = 0; q < 8u; q++) {
                        const V32 e = ev4[q];
                        uint64_t bb = W::ballot((e >> 8) == ctx);     /* lanes past nev hold 0xffffffff: ctx 0xffffff never matches */
                        while (bb) {
                            uint32_t src = W::ctz64(bb); bb &= bb - 1u;
                            uint32_t kk = W::readlane(e, src) & 0xffu;
                            Mask hitl = ln == (kk >> 2);
                            uint32_t sub = kk & 3u;
                            e0 = W::select(hitl & (sub == 0u), e0 + 10u, e0);
                            e1 = W::select(hitl & (sub == 1u), e1 + 10u, e1);
                            e2 = W::select(hitl & (sub == 2u), e2 + 10u, e2);
                            e3 = W::select(hitl & (sub == 3u), e3 + 10u, e3);
                            m++;
                        }
                    }
                }
            }
        }
        CBC_DT(12);                                           /* var_dec: filter, global list */
        uint32_t n = L0 + 10u * m;
        uint32_t tg = target(n), x, lo, cnt;
        if (m == 0u) { x = tg; lo = tg; cnt = 1u; }
        else x = search4(e0, e1, e2, e3, L0, tg, lo, cnt);
        if (status != CBC_ST_OK) return 0u;
        if (x >= L0) { fail(CBC_ST_ASSERT); return 0u; }
        CBC_DT(13);                                           /* var_dec: target, search */
        step(lo, cnt, n);
        CBC_DT(14);                                           /* var_dec: step */
        if (!to_global) {
            uint32_t *arr = tab(CBC_LDS_P0) + bkt * CBC_P0_BUCKET_WORDS;
            const uint32_t k16 = ((ctx >> 8) << 8) | x;
            if (have & 1u) W::write_uni(arr, have >> 1, (half_word & 0xffffu) | (k16 << 16));
            else W::write_uni(arr, have >> 1, 0xffff0000u | k16);
            p0cnt = W::select(ln == bkt, p0cnt + 1u, p0cnt);
            return x;
        }
        if (nev + nev1 >= cap_var) { fail(CBC_ST_CAP_VAR); return 0u; }
        if (p0class) p0over |= 1u << bkt;
        if (!((bw1 & bb1) && (bw2 & bb2))) {
            if ((h1 >> 5) == (h2 >> 5)) W::write_uni(bloom, h1 >> 5, bw1 | bb1 | bb2);
            else { W::write_uni(bloom, h1 >> 5, bw1 | bb1); W::write_uni(bloom, h2 >> 5, bw2 | bb2); }
        }
        if (strand1) { nev1++; W::append_list(ev, cap_var - nev1, (ctx << 8) | x); }
        else { W::append_list(ev, nev, (ctx << 8) | x); nev++; }
        return x;
    }

    /* ---- snpInRef window (same as the encoder's) ---- */
    CBC_MFN void win_clear() { win.clear(); }
    CBC_MFN void win_shift(uint32_t d) { win.shift(d); }
    CBC_MFN uint32_t win_first(uint32_t p, uint32_t rl) { return win.first(p, rl); }
    CBC_MFN void win_set(uint32_t k) { win.set(k); }

    /* the edit counts of an imperfect read (read_decompression.c:404-438) */
    CBC_MFN void edit_counts(uint32_t rl, uint32_t &nSnp, uint32_t &nDel, uint32_t &nIns)
    {
            nSnp = dense_dec_low(tab(CBC_LDS_SNPS), L0, 10u, snps_n); nDel = 0; nIns = 0;
            if (status == CBC_ST_OK && nSnp == 0u) indel_counts(rl, nSnp, nDel, nIns);
    }
    /* after an SNP count of 0: the three counts of a read with indels (:420-438) */
    CBC_MFN void indel_counts(uint32_t rl, uint32_t &nSnp, uint32_t &nDel, uint32_t &nIns)
    {
            uint32_t c0 = 0, c1 = 0, c2 = 0;
            for (int k = 0; k < 3 && status == CBC_ST_OK; k++) {
                const uint32_t c = dense_dec_low(tab(CBC_LDS_INDELS), L0, 16u, indels_n);
                c0 = k == 0 ? c : c0; c1 = k == 1 ? c : c1; c2 = k == 2 ? c : c2;
            }
            if (status != CBC_ST_OK) return;
            nSnp = c0; nDel = c1; nIns = c2;
            if (nIns > rl) fail(CBC_ST_ASSERT);
    }
    /* a read with SNPs only */
    CBC_MFN void edits_snp(uint32_t rl, uint32_t strand, uint32_t nSnp, const V32 &refw, uint8_t *dst)
    {
            CbcDec &D = *this;
            const V32 ln = W::lane();
            const V32 bo = ln * 4u;
                /* SNPs only (read_decompression.c:440-458): the read is the reference window with a few
                 * bytes replaced -- patched in the register that holds 4 bases per lane, no LDS scratch read */
                V32 w = refw;
                uint32_t p = 0;
                for (uint32_t sidx = 0; sidx < nSnp && D.status == CBC_ST_OK; sidx++) {
                    CBC_DT(7);                                     /* to the next SNP (first: the SNP count, loop entry) */
                    uint32_t dl = D.win_first(p, rl);
                    CBC_DT(8);                                     /* win_first */
                    uint32_t g = D.var_dec(((((dl << 7) + p) << 1) | strand));        /* failed: 0, and the loop header ends it */
                    CBC_DT(9);                                     /* var_dec */
                    uint32_t at = p + g;
                    p += g + 1u;
                    D.win_set(p - 1u);
                    const uint32_t shf = (at & 3u) * 8u;
                    uint32_t refch = at < rl ? ((W::readlane(w, (at >> 2) & 63u) >> shf) & 0xffu) : 0u;
                    uint32_t alt = D.small_dec(CBC_LT_CHARS + cbc_basepair(refch) * 8u, 5u, 8u);
                    if (at < rl)
                        w = W::select(ln == (at >> 2), (w & ~(0xffu << shf)) | (cbc_basechar(alt) << shf), w);
                    CBC_DT(10);                                    /* chars + patch */
                }
                if (D.status != CBC_ST_OK) return;
                W::store32_bytes(dst, bo, w, bo < rl);
    }
    /* a read with deletions and / or insertions (and SNPs): tmpb / tmpw = 320 bytes of LDS scratch, dels / insl = 256 words each */
    CBC_MFN void edits_indel(uint32_t pos, uint32_t rl, uint32_t strand, uint32_t nSnp, uint32_t nDel, uint32_t nIns, uint8_t *dst,
                             const uint8_t *refb, uint8_t *tmpb, uint32_t *tmpw, uint32_t *dels, uint32_t *insl)
    {
            CbcDec &D = *this;
            const V32 ln = W::lane();
            const uint32_t T = rl - nIns;                          /* insertion-free length */
            /* deletions: cumulative matched coordinate of each deleted base */
            uint32_t p = 0;
            for (uint32_t d = 0; d < nDel && D.status == CBC_ST_OK; d++) {
                uint32_t g = D.var_dec((p << 1) | strand);
                p += g;
                W::write_uni(dels, d, p);
            }
            if (D.status != CBC_ST_OK) return;
            /* insertion-free read from the reference: base m comes from ref[pos-1 + m + #{dels at <= m}] */
            for (uint32_t b = 0; b < T; b += 64u) {
                V32 m = ln + b;
                V32 sh = W::splat(0u);
                for (uint32_t d = 0; d < nDel; d++) { uint32_t dc = W::read_uni(dels, d); sh = sh + W::select(m >= dc, W::splat(1u), W::splat(0u)); }
                V32 ch = W::load8(refb + (pos - 1u), m + sh, (m < T) & ((m + sh) < 512u));
                W::store8(tmpb, m, ch, m < T);
            }
            /* SNPs (read_decompression.c:440-458) */
            p = 0;
            for (uint32_t s = 0; s < nSnp && D.status == CBC_ST_OK; s++) {
                uint32_t dl = D.win_first(p, rl);
                uint32_t g = D.var_dec(((((dl << 7) + p) << 1) | strand));
                uint32_t at = p + g;
                p += g + 1u;
                D.win_set(p - 1u);
                uint32_t refch = at < T ? ((W::read_uni(tmpw, at >> 2) >> ((at & 3u) * 8u)) & 0xffu) : 0u;
                uint32_t alt = D.small_dec(CBC_LT_CHARS + cbc_basepair(refch) * 8u, 5u, 8u);
                if (at < T) {
                    uint32_t wv = W::read_uni(tmpw, at >> 2), shf = (at & 3u) * 8u;
                    W::write_uni(tmpw, at >> 2, (wv & ~(0xffu << shf)) | (cbc_basechar(alt) << shf));
                }
            }
            if (D.status != CBC_ST_OK) return;
            /* insertions: output index = matched coordinate + number of earlier insertions */
            p = 0;
            for (uint32_t i = 0; i < nIns && D.status == CBC_ST_OK; i++) {
                uint32_t g = D.var_dec((p << 1) | strand);
                p += g;
                uint32_t base = D.small_dec(CBC_LT_CHARS + 5u * 8u, 5u, 8u);
                W::write_uni(insl, i, ((p + i) << 8) | cbc_basechar(base));
            }
            if (D.status != CBC_ST_OK) return;
            for (uint32_t b = 0; b < rl; b += 64u) {
                V32 q = ln + b;
                V32 nb = W::splat(0u), isins = W::splat(0u), ich = W::splat(0u);
                for (uint32_t i = 0; i < nIns; i++) {
                    uint32_t e = W::read_uni(insl, i), oi = e >> 8;
                    nb = nb + W::select(q > oi, W::splat(1u), W::splat(0u));
                    Mask here = q == oi;
                    isins = W::select(here, W::splat(1u), isins);
                    ich = W::select(here, W::splat(e & 0xffu), ich);
                }
                V32 m = q - nb;
                V32 ch = W::load8(tmpb, m, (q < rl) & (m < 320u) & (isins == 0u));
                ch = W::select(isins != 0u, ich, ch);
                W::store8(dst, q, ch, q < rl);
            }
    }
    /* the edits of an imperfect read and the read itself (read_decompression.c:404-529): counts, deletions, SNPs
     * (the reference-derived base is the chars context, :454-455), insertions.  `refw` = the read's reference window,
     * 4 bases per lane.  false = failed. */
    CBC_MFN bool edits_dec(uint32_t pos, uint32_t rl, uint32_t strand, const V32 &refw, uint8_t *dst, const uint8_t *refb,
                           uint8_t *tmpb, uint32_t *tmpw, uint32_t *dels, uint32_t *insl)
    {
        uint32_t nSnp, nDel, nIns;
        edit_counts(rl, nSnp, nDel, nIns);
        if (status != CBC_ST_OK) return false;
        if ((nDel | nIns) == 0u) edits_snp(rl, strand, nSnp, refw, dst);
        else edits_indel(pos, rl, strand, nSnp, nDel, nIns, dst, refb, tmpb, tmpw, dels, insl);
        return status == CBC_ST_OK;
    }
};


/* ===========================================================================================
 * cbc_decode_stream: decode block `blk` completely.
 * =========================================================================================== */
template <class W>
CBC_FN void cbc_decode_stream(const cbc_dec_args &A, uint32_t blk, uint32_t *lds)
{
    typedef typename W::V32 V32;
    typedef typename W::Mask Mask;
    const V32 ln = W::lane();
    const cbc_dec_block_desc *bd = A.blocks + blk;
    CbcDec<W> D;

    const uint64_t in_off = bd->in_off, ref_off = bd->ref_off, rec_base = bd->rec_base, seq_base = bd->seq_base;
    const uint32_t in_bytes = bd->in_bytes, n_reads = bd->n_reads, L0 = bd->read_length, stride = bd->seq_stride;

    D.status = CBC_ST_OK; D.nsym = 0; D.fail_read = 0; D.cur_read = 0;
    D.l = W::dv(0u); D.rng = W::dv(CBC_M26 + 1u); D.d = W::dv(0u); D.acc = 0; D.navail = 0; D.widx = 0; D.wordv = W::splat(0u);
    D.inb = A.in + in_off;
    D.lds = lds; D.cap_pos = A.cap_pos; D.cap_var = A.cap_var; D.L0 = L0;
    D.evp = A.var_scratch + (uint64_t)blk * A.cap_var;
    D.rname_key = lds + CBC_LDS_RNKEY; D.rname_exc = lds + CBC_LDS_RNEXC; D.rn_cap = CBC_CAP_NAME; D.histp = lds + CBC_DLDS_HIST;
    D.pos_valp = lds + CBC_DLDS_FIXED; D.pos_cntp = lds + CBC_DLDS_FIXED + A.cap_pos; D.vtab = nullptr;
    D.fsp_key = D.fsp_exc = nullptr; D.fsp_count = 0; D.pos_ov_valp = D.pos_ov_cntp = nullptr; D.pos_lds_cap = 0xffffffc0u; D.palpha = nullptr;
    /* the payload buffer must leave 3 spare bytes after the last payload (whole-dword reads) */
    /* every range test is written without the sum base + length, which a crafted descriptor could make wrap */
    bool args_ok = cbc_fits64(in_off, ((uint64_t)in_bytes + 3u) & ~3ull, A.in_bytes) && cbc_fits64(rec_base, n_reads, A.n_recs) &&
                   (stride >= 4u && stride <= 256u && (stride & 3u) == 0u) &&
                   cbc_fits64(seq_base, (uint64_t)n_reads * stride + 8u, A.seq_bytes) && (L0 >= 1u && L0 <= 256u) &&
                   cbc_le64(ref_off, A.ref_bytes) &&
                   cbc_le64(((uint64_t)blk + 1u) * A.cap_var, A.var_scratch_words);
    D.nwords_in = (in_bytes + 3u) >> 2;
    D.tail_valid = in_bytes & 3u;
    if (!args_ok) { D.nwords_in = 0; D.fail(CBC_ST_ASSERT); }
    if (n_reads > CBC_MAX_BLOCK_READS) D.fail(CBC_ST_UNSUPPORTED);   /* the closed forms below assume no rescale */

    for (uint32_t b = 0; b < CBC_DLDS_FIXED; b += 64u) W::store32(lds, ln + b, W::splat(0u), (ln + b) < CBC_DLDS_FIXED);
    D.rlen_n = 255u; D.rl123_c0 = 1u; D.rl123_n = 255u; D.snps_n = L0; D.indels_n = L0; D.rn_count = 0;
    D.pos_card = 1u; D.pos_n = 1u; D.nev = 0; D.nev1 = 0;
    D.pval = W::splat(0xffffffffu); D.pcnt = W::select(ln == 0u, W::splat(1u), W::splat(0u));
    D.fkey = W::splat(0u); D.fexc = W::splat(0u); D.fcount = 0; D.fn = 65536u;
    D.hkey = W::splat(0u); D.hexc = W::splat(0u);
    D.hc0 = D.hc1 = D.hc2 = D.hc3 = 0; D.hn0 = D.hn1 = D.hn2 = D.hn3 = 256u;
    {
        V32 s = W::select(ln < 10u, W::splat(1u), W::splat(0u));
        V32 r = (ln - CBC_LT_CHARS) >> 3, c = (ln - CBC_LT_CHARS) & 7u;
        Mask inch = (ln >= CBC_LT_CHARS) & (r < 6u) & (c < 5u);
        V32 cv = W::select(c == 4u, W::splat(1u), W::select(c == r, W::splat(0u), W::splat(8u)));
        Mask bump = ((r == 0u) & ((c == 1u) | (c == 2u))) | ((r == 1u) & ((c == 0u) | (c == 3u))) |
                    ((r == 2u) & ((c == 0u) | (c == 3u))) | ((r == 3u) & ((c == 1u) | (c == 2u)));
        cv = W::select(bump, cv + 8u, cv);
        D.small = W::select(inch, cv, s);
    }
    D.prevPos = 0; D.prevM = 0; D.prevChar = 0; D.win_clear();
    D.rl_memo_x = CBC_NOMEMO; D.rl_memo_lo = 0; D.rl_memo_cnt = 0; D.rl_last_x = 0;
    D.p0cnt = W::splat(0u); D.p0over = 0;

    /* the tag: first 26 bits (alloc_arithmetic_stream, Arithmetic_stream.c:260-263) */
    if (D.status == CBC_ST_OK) D.d = W::dv(D.take(26u));

    /* stream header: int(L0), 32 x int(WELL), int(8) */
    for (uint32_t k = 0; k < 34u && D.status == CBC_ST_OK; k++) {
        uint32_t v = D.regsparse_dec(D.hkey, D.hexc, 0u, 8u, D.hc0, D.hn0, 256u, 1u, CBC_ST_ASSERT) << 24;
        v |= D.regsparse_dec(D.hkey, D.hexc, 8u, 8u, D.hc1, D.hn1, 256u, 1u, CBC_ST_ASSERT) << 16;
        v |= D.regsparse_dec(D.hkey, D.hexc, 16u, 8u, D.hc2, D.hn2, 256u, 1u, CBC_ST_ASSERT) << 8;
        v |= D.regsparse_dec(D.hkey, D.hexc, 24u, 8u, D.hc3, D.hn3, 256u, 1u, CBC_ST_ASSERT);
        if (D.status != CBC_ST_OK) break;
        if (k == 0u && v != L0) D.fail(CBC_ST_ASSERT);            /* container and stream disagree */
        if (k == 33u && v != 8u) D.fail(CBC_ST_UNSUPPORTED);       /* LOSSY streams are out of scope */
    }

    uint4 *recs4 = (uint4 *)(A.recs + rec_base);
    uint8_t *seqo = A.seq + seq_base;
    const uint8_t *refb = A.ref + ref_off;
    const uint64_t ref_avail = cbc_le64(ref_off, A.ref_bytes) ? A.ref_bytes - ref_off : 0;
    const uint32_t ref_lim = cbc_avail32(ref_avail, 0u);
    uint8_t *tmpb = (uint8_t *)(lds + CBC_DLDS_TMP);
    uint32_t *dels = lds + CBC_DLDS_DELS, *insl = lds + CBC_DLDS_INS;

    /* a perfect read is its reference window: the window register of record r is stored at the top of record
     * r + 1 (or after the loop), by when its load has long completed -- no copy, no wait */
    V32 refw = W::splat(0u); uint8_t *pend_dst = seqo; uint32_t pend_rl = 0;
    V32 sr_fh = W::splat(0u), t_fh = W::splat(0u);
#if defined(CBC_DSTAMP) && defined(__HIP_DEVICE_COMPILE__)
    D.dt_last = 0;
    for (int i = 0; i < 16; i++) D.dt_sum[i] = 0;
    CBC_DT0();
#endif
    /* ---- the record loop ------------------------------------------------------------------------------------------
     * Control flow is paid for in scalar instructions, so the loop that runs for every record holds only
     * what nearly every record needs, and leaves only through its header:
     *   - record 0 (contig name) is coded before the loop;
     *   - a record that turns out to need rare, bulky code -- a POS escape (four byte symbols, a new alphabet entry), a
     *     read with indels (LDS scratch read, three edit lists) -- notes where it stands in `defer`, the loop ends at
     *     its header, the record is finished by the general code below, and the loop is entered again;
     *   - a failed check sets D.status, the rest of the record is skipped, and the header ends the loop.
     * The phases of a record are lambdas over the variables just below (one set per wavefront, in scalar registers). */
    uint32_t rl = 0, x = 0, pos = 0, flag = 0, strand = 0, match = 0, nSnp = 0, nDel = 0, nIns = 0;
    /* where the deferred record stands: 3 = rlength[0] is next (the verified guess missed), 1 = the POS escape symbol is
     * taken, 4 = FLAG is next (not a value seen before), 2 = the SNP count is taken and is 0 (the counts of a read with indels follow) */
    uint32_t defer = 0;
    uint8_t *dst = seqo;
    auto ok = [&]() { return D.status == CBC_ST_OK; };
    /* S1: same_ref, or for record 0 the contig name */
    auto ph_same_ref = [&](uint32_t r, auto first) {
        D.cur_read = r;
        if (pend_rl) { W::store32_bytes(pend_dst, ln * 4u, refw, (ln * 4u) < pend_rl); pend_rl = 0; }   /* record r - 1 was perfect */
        if ((r & 63u) == 0u) {                               /* scaled fractions of the closed-form symbols of 64 records */
            const V32 rdv = ln + r;
            sr_fh = W::frac32(rdv * 10u - 9u, rdv * 10u + 2u);     /* same_ref symbol 0 of record r >= 1 */
            t_fh = W::frac32(rdv * 10u + 1u, rdv * 10u + 255u);    /* rlength[1..3] symbol 0 */
        }
        /* -- decompress_rname (id_compression.c:67-94): same_ref is (1,1) until record 0 takes symbol 1,
         *    after which every record of the block must take symbol 0 (one contig per block) -- */
        if constexpr (decltype(first)::value) {
            uint32_t sr = D.small_dec(CBC_LT_SAMEREF, 2u, 10u);
            if (ok() && sr != 1u) D.fail(CBC_ST_ASSERT);
            for (uint32_t q = 0; q < CBC_CAP_NAME && ok(); q++) {
                uint32_t ch = D.rname_dec(D.prevChar);
                if (ch == 0u) break;
                if (ch == (uint32_t)'\n' && q == 0u) { D.fail(CBC_ST_ASSERT); break; }   /* empty block */
                D.prevChar = ch;
            }
            D.prevPos = 0; D.win_clear();
        } else D.step_known0(10u * r - 9u, 10u * r + 2u, W::readlane(sr_fh, r & 63u));
        CBC_DT(0);                                            /* same_ref (+ name) */
    };
    /* S3: rlength[1..3] (read_decompression.c:68-74: only the low byte carries information, quirk Q1) and the length check */
    auto ph_rl_tail = [&](uint32_t r) {
        const uint32_t tf = W::readlane(t_fh, r & 63u);
        for (int k = 1; k < 4; k++) D.step_known0(10u * r + 1u, 10u * r + 255u, tf);      /* a failed one leaves the state alone */
        if (ok() && (rl == 0u || rl > CBC_MAX_READ_LEN || rl > stride)) D.fail(CBC_ST_ASSERT);
        CBC_DT(1);                                            /* rlength x 4 */
    };
    /* S5: from the POS delta x to the position; the snpInRef window slides */
    auto ph_pos = [&](auto first) {
        if (x < 1u || x >= 5000000u) { D.fail(CBC_ST_ASSERT); return; }
        pos = D.prevPos + x - 1u;
        if (pos < D.prevPos) { D.fail(CBC_ST_ASSERT); return; }     /* the 32-bit sum wrapped: not a position of this window */
        D.win_shift(decltype(first)::value ? 256u : x - 1u);
        D.prevPos = pos;
        CBC_DT(2);                                            /* pos */
    };
    /* S7: after FLAG: strand, the reference window load, the match flag */
    auto ph_flag_tail = [&](uint32_t r) {
        strand = (flag >> 4) & 1u;
        if (pos == 0u || pos > ref_lim || ref_lim - pos < rl + 3u + 256u) { D.fail(CBC_ST_ASSERT); return; }
        /* the reference window of the read, 4 bases per lane: issued now, needed after the match flag (perfect read: it
         * IS the read) or after the edits (SNP-only read: patched in place).  Issued one model earlier, right after POS,
         * it measures 0.6 % slower (run 23): FLAG's own waits then include it. */
        refw = W::load32_bytes(refb + (pos - 1u), ln * 4u, (ln * 4u) < rl);
        CBC_DT(3);                                            /* flag */
        match = D.small_dec(CBC_LT_MATCH + (((x == 1u) ? 2u : 0u) | D.prevM) * 2u, 2u, 1u);
        if (!ok()) return;
        D.prevM = match;
        dst = seqo + (uint64_t)r * stride;
        CBC_DT(4);                                            /* match */
    };
    auto ph_store = [&](uint32_t r) {
        CBC_DT(5);                                            /* copy / edits + reconstruction */
        V32 rv0 = W::splat(pos), rv1 = W::splat(flag | (rl << 16)), rv2 = W::splat(r * stride), rv3 = W::splat(0u);
        W::store_rec(recs4, W::splat(r), ln == 0u, rv0, rv1, rv2, rv3);
        CBC_DT(6);                                            /* record store */
    };
    /* a record from step `from` on with every model in its complete form (record 0, and what the loop deferred) */
    auto general = [&](uint32_t r, auto first, uint32_t from) {
        D.cur_read = r;
        if (from == 3u) {
            rl = D.rlen_dec();
            if (ok()) ph_rl_tail(r);
            if (ok()) x = D.pos_dec();
        }
        if (from == 1u) x = D.pos_escape();
        if ((from == 1u || from == 3u) && ok()) ph_pos(first);
        if (from != 2u && ok()) {
            flag = D.regsparse_dec(D.fkey, D.fexc, 0u, CBC_CAP_FLAG, D.fcount, D.fn, 65536u, 8u, CBC_ST_CAP_FLAG);
            if (ok()) ph_flag_tail(r);
            if (ok()) {
                if (match) { pend_dst = dst; pend_rl = rl; }   /* stored at the top of the next record; stride >= rl rounded to 4 */
                else D.edit_counts(rl, nSnp, nDel, nIns);
            }
        }
        if (from == 2u) { nDel = 0; nIns = 0; D.indel_counts(rl, nSnp, nDel, nIns); }
        if (ok() && !match) {
            if ((nDel | nIns) == 0u) D.edits_snp(rl, strand, nSnp, refw, dst);
            else D.edits_indel(pos, rl, strand, nSnp, nDel, nIns, dst, refb, tmpb, lds + CBC_DLDS_TMP, dels, insl);
        }
        if (ok()) ph_store(r);
    };
    uint32_t r = 0;
    if (n_reads && ok()) {                                   /* record 0 */
        ph_same_ref(0u, std::true_type());
        if (ok()) general(0u, std::true_type(), 3u);
        r = 1u;
    }
    while (r < n_reads && ok()) {
        for (; r < n_reads && ok() && defer == 0u; r++) {     /* the loop proper: every model in its usual-case form */
            ph_same_ref(r, std::false_type());
            if (!ok()) continue;
            if (!D.rlen_fast(rl)) { defer = 3u; continue; }
            ph_rl_tail(r);
            if (!ok()) continue;
            const uint32_t idx = D.pos_sym();
            if (idx == CBC_NOMEMO) continue;                 /* failed */
            if (idx == 0u) { defer = 1u; continue; }
            x = D.pos_value(idx);
            ph_pos(std::false_type());
            if (!ok()) continue;
            if (!D.regsparse_fast(D.fkey, D.fexc, 0u, D.fcount, D.fn, 65536u, 8u, flag)) { defer = 4u; continue; }
            ph_flag_tail(r);
            if (!ok()) continue;
            if (match) { pend_dst = dst; pend_rl = rl; }
            else {
                nSnp = D.dense_dec_low(D.tab(CBC_LDS_SNPS), L0, 10u, D.snps_n);
                if (!ok()) continue;
                if (nSnp == 0u) { defer = 2u; continue; }        /* three more counts follow: a read with indels */
                D.edits_snp(rl, strand, nSnp, refw, dst);
                if (!ok()) continue;
            }
            ph_store(r);
        }
        if (defer != 0u && ok()) general(r - 1u, std::false_type(), defer);   /* finish the record the loop left */
        defer = 0u;
    }

    if (pend_rl) W::store32_bytes(pend_dst, ln * 4u, refw, (ln * 4u) < pend_rl);

    /* sentinel: same_ref(1), '\n', NUL (compression.c:152; decompress_rname returns -1 on it) */
    if (D.status == CBC_ST_OK) {
        D.cur_read = n_reads;
        if (n_reads > 1u)                                    /* same_ref counts after n records: (1 + 10 (n - 1), 11) */
            D.small = W::select(ln == CBC_LT_SAMEREF, W::splat(10u * n_reads - 9u), D.small);
        uint32_t sr = D.small_dec(CBC_LT_SAMEREF, 2u, 10u);
        if (D.status == CBC_ST_OK && sr != 1u) D.fail(CBC_ST_ASSERT);
        if (D.status == CBC_ST_OK) {
            uint32_t ch = D.rname_dec(D.prevChar);
            if (D.status == CBC_ST_OK && ch != (uint32_t)'\n') D.fail(CBC_ST_ASSERT);
            if (D.status == CBC_ST_OK && D.rname_dec((uint32_t)'\n') != 0u) D.fail(CBC_ST_ASSERT);
        }
    }
#if defined(CBC_DSTAMP) && defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < 16; i++) { W::write_uni((uint32_t *)seqo, 2 * i, (uint32_t)D.dt_sum[i]); W::write_uni((uint32_t *)seqo, 2 * i + 1, (uint32_t)(D.dt_sum[i] >> 32)); }
#endif
    V32 resv = W::select(ln == 0u, W::splat(D.status == CBC_ST_OK ? n_reads : D.cur_read), W::select(ln == 1u, W::splat(D.status),
               W::select(ln == 2u, W::splat(D.nsym), W::splat(D.fail_read))));
    W::store32((uint32_t *)(A.results + blk), ln, resv, ln < 4u);
}

#endif /* CBC_DECODE_BODY_H */
