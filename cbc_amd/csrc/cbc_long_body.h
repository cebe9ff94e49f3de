/*
 * cbc_long_body.h -- the LONG-READ FORMAT EXTENSION (stream version 3).
 *
 * The reference cannot code a read longer than 252 bases (var contexts < 65535: sam_models.c:317; MAX_READ_LENGTH
 * 1024: sam_block.h:38; pos steps < 5e6: :54), so this is a format of its own under its own version number -- no
 * reference parity exists for it; oracle/cbc_long.c is its independent CPU statement and the tests hold GPU ==
 * that file byte for byte, and decode(encode(x)) == x.  What is kept from the reference: the range coder, the
 * adaptive model rule (stream_model.c:31-51), compress_int, the contig-name, pos, FLAG and chars models.
 *
 *   stream  := int(0x43424C03) int(8)  record*  same_ref(1) '\n' NUL  flush
 *   record  := same_ref | name ...        len x 4 (real bytes, MSB first)    pos_sym [esc b3 b2 b1 b0]    flag
 *              ne_hi ne_lo                { gap [gx_hi gx_lo] kind [base] } x ne
 *   gap: matched bases since the previous edit, min(g, 255) in context 2 * prev_kind + strand (prev_kind 3 = first
 *   edit), g >= 255 followed by g - 255 as two bytes; kind: 0 sub, 1 ins / soft clip, 2 del, context prev_kind;
 *   base: chars[reference base] for a sub, chars[O] for an ins.  Edits are found HERE, from read vs reference along
 *   the CIGAR (64 bases per compare + ballot); MD is not used.
 *
 * One block = one stream = one wavefront (blocks are cut at 64 reads / 1 Mbase, so a 1 M x 10 kb input is ~15 600
 * streams).  Every model is in its general (rescaling) form: a 10 kb read carries ~500 edits, a block ~100 k
 * symbols per model.  Dense excess tables in LDS (16 x 256 words), pos alphabet with literal counts.
 */
#ifndef CBC_LONG_BODY_H
#define CBC_LONG_BODY_H

#include "cbc_encode_body.h"
#include "cbc_decode_body.h"

#define CBC_LONG_MAGIC 0x43424C03u
/* LDS words.  Both directions: of each of the eight gap tables only the symbols 0..63 (where a 5 % edit rate puts 96 %
 * of the gaps), the six tables that see at most ONE symbol per read (len x 4, ne x 2) as sparse (value << 24 | excess) lists
 * of 64 words each -- a block holds at most 64 reads, so such a list cannot overflow and its model cannot reach the rescale
 * point (n <= 256 + 10 * 64) --, the contig-name pairs.  Then, encode only: the output bit ring, the hand-off ring of the
 * two wavefronts with its counters, the triples of a batch.  Then pos value / count (the decoder derives pos_alpha from them).
 * The gap symbols 64..255 and the two gx tables live in GLOBAL memory, CBC_LONG_SCRATCH_WORDS per block (the kernel zeroes
 * them): these kernels are bound by how many blocks a CU holds -- 20 KB of LDS per block in round 2 (2 wavefronts per SIMD),
 * 16.5 KB with the sparse lists, 7.5 KB now: the register file, not LDS, sets the residency. */
#define CBC_LLDS_GAPLO  0u         /* 8 x 64: excess of gap symbols 0..63 */
#define CBC_LLDS_SP     512u       /* 6 x 64: len[0..3], ne[0..1] */
#define CBC_LLDS_RNKEY  896u       /* CBC_CAP_NAME */
#define CBC_LLDS_RNEXC  (896u + CBC_CAP_NAME)
#define CBC_LLDS_ROLE   (896u + 2u * CBC_CAP_NAME)
#define CBC_LLDS_RING   CBC_LLDS_ROLE                               /* encode: CBC_RING_WORDS */
#define CBC_LLDS_BATCH  (CBC_LLDS_ROLE + CBC_RING_WORDS)            /* encode: CBC_BATCH_SLOTS x CBC_BATCH_WORDS */
#define CBC_LLDS_CTL    (CBC_LLDS_BATCH + CBC_BATCH_SLOTS * CBC_BATCH_WORDS)   /* encode: 8 */

#define CBC_LLDS_TRIP   (CBC_LLDS_CTL + 8u)                         /* encode: 3 x 192: the (cum, count, total) triples of a batch of 64 edits */
#define CBC_LLDS_FIXED  (CBC_LLDS_TRIP + 576u)
static inline uint32_t cbc_long_lds_bytes(uint32_t cap_pos) { return 4u * (CBC_LLDS_FIXED + 2u * cap_pos); }       /* encode */
static inline uint32_t cbc_long_dec_lds_bytes(uint32_t cap_pos) { return 4u * (CBC_LLDS_ROLE + 2u * cap_pos); }   /* decode: no rings */
/* global scratch of a block, in words: [8 x 192 gap symbols 64..255][2 x 256 gx] */
#define CBC_LSCR_GAPHI  0u
#define CBC_LSCR_GX     1536u
#define CBC_LSCR_EDITS  2048u      /* encode: the edits of the read being coded (their count precedes them in the stream) */
#ifndef CBC_LONG_EDIT_CAP
#define CBC_LONG_EDIT_CAP 8192u    /* a read with more edits than HALF this is walked twice (count, then code) */
#endif
#define CBC_LONG_EDIT_HALF (CBC_LONG_EDIT_CAP / 2u)                  /* the buffer is two halves: the walk of read r + 1 fills one while read r is coded from the other */
#define CBC_LONG_TABLE_WORDS 2048u                                   /* what the decoder needs per block */
#define CBC_ROLE_WALKER 3u         /* third wavefront of the long-read encoder: finds the edits of the reads, one read ahead */
#define CBC_LONG_SCRATCH_WORDS (2048u + CBC_LONG_EDIT_CAP)           /* ... and the encoder */

/* gap tables: index 0..7 = 2 * prev_kind + strand; gx: 0..1; sparse lists: 0..3 len, 4..5 ne */
#define CBC_LS_LEN 0u
#define CBC_LS_NE  4u
/* kind model (4 contexts x 3, init 1, step 8) in the free lanes of the small lane table: 0-2, 3-5, 10-12, 13-15 */
CBC_FN uint32_t cbc_long_kind_base(uint32_t pk) { return pk < 2u ? pk * 3u : 10u + (pk - 2u) * 3u; }

struct cbc_long_args {
    const cbc_read_rec   *recs;
    const uint8_t        *seq;
    const uint32_t       *tok;
    const uint8_t        *names;
    const cbc_block_desc *blocks;
    const uint8_t        *ref;
    uint8_t              *out;
    cbc_block_result     *results;
    uint64_t ref_bytes, out_bytes, seq_bytes, n_tok, n_recs;
    uint32_t n_blocks, cap_pos, names_bytes;
    uint32_t *scratch;                  /* n_blocks x CBC_LONG_SCRATCH_WORDS (no initial content needed) */
};

/* 512 consecutive bytes of a byte array in two vector registers (4 bytes per lane: [base, base + 256) and the 256 after
 * them) with a third register loaded one window further ahead; unaligned dwords per lane come out of the two by lane
 * gathers and a funnel shift -- no memory access.  The encoder walks a read and its reference through such windows: a 10 kb
 * read with 5 % edits has ~340 CIGAR runs of ~30 bases, and a global-memory round trip per run was nine tenths of the model
 * wavefront's time once the models themselves ran a batch at a time (profiles/r03_i_ab_long.log: 2800 cycles per run). */
template <class W>
struct CbcLWin {
    typedef typename W::V32 V32;
    V32 cur, nx1, nx2; uint32_t base, lim; const uint8_t *p;
    CBC_MFN V32 ld(uint32_t b)
    {
        const V32 bo = W::lane() * 4u;
        return W::load32_bytes(p + b, bo, (bo + b < lim) & ((lim - b - bo) >= 4u));      /* bytes past `lim` read as 0 */
    }
    CBC_MFN void reset(const uint8_t *p_, uint32_t lim_, uint32_t pos)
    {
        p = p_; lim = lim_; base = pos & ~3u; cur = ld(base); nx1 = ld(base + 256u); nx2 = ld(base + 512u);
    }
    CBC_MFN void to(uint32_t pos)                              /* afterwards base <= pos < base + 256 */
    {
        while (pos - base >= 256u) {
            if (pos - base < 768u) { cur = nx1; nx1 = nx2; base += 256u; nx2 = ld(base + 512u); }
            else { reset(p, lim, pos); }
        }
    }
    /* lane l: bytes [pos + 4l, pos + 4l + 4); wants base <= pos < base + 256; bytes from base + 512 on are garbage */
    CBC_MFN V32 dwords(uint32_t pos)
    {
        const uint32_t off = pos - base;
        const V32 wi = W::lane() + (off >> 2);
        const V32 lo = W::select(wi < 64u, W::lane_gather(cur, wi), W::lane_gather(nx1, wi));
        const V32 hi = W::select(wi + 1u < 64u, W::lane_gather(cur, wi + 1u), W::lane_gather(nx1, wi + 1u));
        return W::funnel_shr(hi, lo, (off & 3u) * 8u);
    }
};

/* One block = one stream, coded by THREE wavefronts (round 3; CBC_ROLE_FUSED = all in one, the CPU emulation):
 *   WALKER  finds the edits of every read (read vs reference along the CIGAR, one lane per CIGAR run) and leaves them in the
 *           block's edit buffer in global memory, one read ahead of the model wavefront (the buffer has two halves);
 *   MODEL   codes the record headers and turns the edits, 64 at a time, into (cum, count, total) triples of the adaptive models;
 *   CODER   runs the range coder and the bit packer over the triples.
 * WALKER -> MODEL: ctl[2] = reads walked (release / acquire like the batch counters), ctl[3] = reads the model wavefront is
 * done with, ctl[4 + (r & 1)] = edit count of read r | "did not fit" << 31, ctl[6] = the walker's failure (status | read << 16),
 * ctl[7] = the model wavefront has left.  The walker waits only for a free half (read r - 2 done), the model wavefront only
 * for read r's walk: no cycle; either side's exit releases the other.  MODEL -> CODER: the LDS ring of the block encoder
 * (CbcEnc::publish / pull); the model wavefront ends with a batch flagged LAST that carries the status, the coder takes
 * batches until it has seen it, whatever happens on either side. */
template <class W, uint32_t ROLE>
CBC_FN void cbc_long_encode(const cbc_long_args &A, uint32_t blk, uint32_t *lds)
{
    typedef typename W::V32 V32;
    typedef typename W::Mask Mask;
    typedef CbcEnc<W, true> Enc;
    const V32 ln = W::lane();
    const cbc_block_desc *bd = A.blocks + blk;
    Enc E;
    const uint64_t rec_base = bd->rec_base, seq_base = bd->seq_base, tok_base = bd->tok_base, ref_off = bd->ref_off, out_off = bd->out_off;
    const uint32_t out_cap = bd->out_cap, n_reads = bd->n_reads, name_off = bd->name_off, n_tok_blk = bd->n_tok;

    E.status = CBC_ST_OK; E.nsym = 0; E.fail_read = 0; E.cur_read = 0;
    E.l = W::uv(0u); E.set_l_forms(); E.rng = W::uv(CBC_M26 + 1u); E.scale3 = 0u; E.bitpos = 0; E.flushed = 0;
    E.ring = lds + CBC_LLDS_RING;
    E.q_lo = W::splat(0u); E.q_cnt = W::splat(0u); E.q_n = W::splat(0u); E.q_len = 0;
    E.b_lo = W::splat(0u); E.b_hi = W::splat(0u); E.b_n = W::splat(1u); E.b_fl = W::splat(0u); E.b_fh = W::splat(0u);
    E.rec_a = W::splat(0u); E.rec_s = W::splat(0u); E.rec_n = 0;
    E.role = ROLE; E.batch_i = 0; E.batch = lds + CBC_LLDS_BATCH; E.ctl = lds + CBC_LLDS_CTL;
    E.b_len = 0; E.b_pos = 0; E.b_stop = 64u; E.b_flags = 0; E.seen_last = 0; E.b_neq = 0;
    if (ROLE != CBC_ROLE_FUSED) {                            /* the only barrier: the hand-off counters start at zero for both */
        if (ROLE == CBC_ROLE_MODEL) for (uint32_t k = 0; k < 8u; k++) W::write_uni(E.ctl, k, 0u);
        W::barrier();
    }
    E.out32 = (uint32_t *)(A.out + out_off);
    E.cap_words = out_cap >> 2;
    bool args_ok = cbc_fits64(out_off, out_cap, A.out_bytes) && ((out_off & 3u) == 0u) && cbc_fits64(rec_base, n_reads, A.n_recs) &&
                   cbc_fits64(tok_base, n_tok_blk, A.n_tok) && (name_off < A.names_bytes) && A.cap_pos >= 2u && n_reads <= 64u &&
                   A.scratch != nullptr;
    if (!args_ok) { E.cap_words = 0; E.fail(CBC_ST_ASSERT); }
#ifdef CBC_STAMP
    for (int i = 0; i < 16; i++) E.t_sum[i] = 0;
    CBC_T0();
#endif

    /* ================================ coder wavefront ======================================= */
    if (ROLE == CBC_ROLE_CODER) {
        W::prio(CBC_PRIO_CODER);
        for (uint32_t b = 0; b < CBC_RING_WORDS; b += 64u) W::store32(E.ring, ln + b, W::splat(0u), W::all());
        E.consume_all();
        uint32_t nbytes = 0;
        if (E.status == CBC_ST_OK) { E.flush_recs(); nbytes = E.finish(); }
        if (E.status != CBC_ST_OK) nbytes = 0;
#if defined(CBC_STAMP) && defined(__HIP_DEVICE_COMPILE__)
        CBC_TS(9);
        if (out_cap >= 512u) for (int i = 0; i < 16; i++) {      /* diagnostic build: over the payload start */
            W::write_uni(E.out32, 2 * i, (uint32_t)E.t_sum[i]); W::write_uni(E.out32, 2 * i + 1, (uint32_t)(E.t_sum[i] >> 32)); }
#endif
        V32 resv = W::select(ln == 0u, W::splat(nbytes), W::select(ln == 1u, W::splat(E.status),
                   W::select(ln == 2u, W::splat(E.nsym), W::splat(E.fail_read))));
        W::store32((uint32_t *)(A.results + blk), ln, resv, ln < 4u);
        return;
    }

    /* ================================ model wavefront (or both, fused) ====================== */
    E.L0 = 256u;                                              /* alphabet of the dense tables (var is not used) */
    E.rlen_exc = nullptr; E.snps_exc = nullptr; E.indels_exc = nullptr;
    E.rname_key = lds + CBC_LLDS_RNKEY; E.rname_exc = lds + CBC_LLDS_RNEXC; E.rn_cap = CBC_CAP_NAME; E.rn_count = 0;
    E.pos_val = lds + CBC_LLDS_FIXED; E.pos_occ = E.pos_val + A.cap_pos; E.pos_pre = nullptr; E.cap_pos = A.cap_pos;
    E.fsp_key = E.fsp_exc = nullptr; E.fsp_count = 0; E.pos_ov_val = E.pos_ov_occ = nullptr; E.pos_lds_cap = 0xffffffc0u; E.palpha = nullptr;
    E.bloom = nullptr; E.var_ev = nullptr; E.nev = E.nev1 = 0; E.cap_var = 0; E.vtab = nullptr; E.p0ev = nullptr;
    E.p0cnt = W::splat(0u); E.p0over = 0;
    uint32_t *scr = A.scratch + (uint64_t)blk * CBC_LONG_SCRATCH_WORDS;
    if (ROLE != CBC_ROLE_WALKER) {                            /* the tables are the model wavefront's */
        for (uint32_t b = 0; b < CBC_LLDS_SP; b += 64u) W::store32(lds, ln + b, W::splat(0u), W::all());        /* gap symbols 0..63 */
        if (E.status == CBC_ST_OK) for (uint32_t b = 0; b < CBC_LONG_TABLE_WORDS; b += 64u) W::store32_list(scr, ln + b, W::splat(0u), W::all());   /* the rest, gx */
        if (ROLE == CBC_ROLE_FUSED) for (uint32_t b = 0; b < CBC_RING_WORDS; b += 64u) W::store32(E.ring, ln + b, W::splat(0u), W::all());
        W::write_uni(E.pos_val, 0u, 0xffffffffu); W::write_uni(E.pos_occ, 0u, 1u);
    }
    E.snps_n = 0; E.indels_n = 0; E.pos_card = 1u;
    E.fkey = W::splat(0u); E.fexc = W::splat(0u); E.fcount = 0;
    E.hkey = W::splat(0u); E.hexc = W::splat(0u);
    E.hc0 = E.hc1 = E.hc2 = E.hc3 = 0; E.hn0 = E.hn1 = E.hn2 = E.hn3 = 256u;
    {   /* lane table: kind contexts (lanes 0-5, 10-15) 1,1,1; same_ref 1,1 (lanes 8, 9); chars rows (16..63) */
        V32 sm = W::select(ln < 16u, W::splat(1u), W::splat(0u));
        V32 r = (ln - CBC_LT_CHARS) >> 3, c = (ln - CBC_LT_CHARS) & 7u;
        Mask inch = (ln >= CBC_LT_CHARS) & (r < 6u) & (c < 5u);
        V32 cv = W::select(c == 4u, W::splat(1u), W::select(c == r, W::splat(0u), W::splat(8u)));
        Mask bump = ((r == 0u) & ((c == 1u) | (c == 2u))) | ((r == 1u) & ((c == 0u) | (c == 3u))) |
                    ((r == 2u) & ((c == 0u) | (c == 3u))) | ((r == 3u) & ((c == 1u) | (c == 2u)));
        cv = W::select(bump, cv + 8u, cv);
        E.small = W::select(inch, cv, sm);
    }
    E.prevPos = 0; E.prevM = 0; E.prevChar = 0; E.win_pos = 0; E.win_clear();
    V32 ntab = W::splat(256u), lsum = W::splat(0u);            /* lane t < 8: total of gap table t, excess held by its symbols 0..63; lanes 8, 9: totals of gx */
    uint32_t flag_n = 65536u, pos_n = 1u;
    V32 spc = W::splat(0u);                                    /* entries of the six sparse lists, lane = list */

    /* a 256-symbol dense table in global memory (the list accessors: this wavefront alone writes and re-reads it) */
    auto gsum_below = [&](const uint32_t *row, uint32_t k) -> uint32_t {     /* excess of the row's entries [0, k), k <= 256 */
        V32 a = W::splat(0u);
        const uint32_t kb = W::uni(k);
        for (uint32_t b = 0; b < kb; b += 64u) a = a + W::load32_list(row, ln + b, (ln + b) < k, 0u);
        return W::reduce_add(a);
    };
    auto grescale = [&](uint32_t *row, uint32_t card) -> uint32_t {          /* stream_model.c:41-48 on e = count - 1; returns the new sum */
        V32 a = W::splat(0u);
        for (uint32_t b = 0; b < card; b += 64u) {
            const Mask m = (ln + b) < card;
            const V32 e = (W::load32_list(row, ln + b, m, 0u) + 1u) >> 1;
            W::store32_list(row, ln + b, e, m);
            a = a + W::select(m, e, W::splat(0u));
        }
        return W::reduce_add(a);
    };
    /* gap in context t (0..7): symbols 0..63 from LDS, the others from the block's global scratch */
    auto gap_code = [&](uint32_t t, uint32_t x) {
        uint32_t n = W::readlane(ntab, t), low = W::readlane(lsum, t);
        uint32_t *lo_tab = lds + CBC_LLDS_GAPLO + 64u * t, *hi_row = scr + CBC_LSCR_GAPHI + 192u * t;
        if (x < 64u) {
            uint32_t lo, cnt;
            E.dense_lookup(lo_tab, x, lo, cnt);
            E.encode(lo, cnt, n);
            W::write_uni(lo_tab, x, cnt - 1u + 10u);
            low += 10u;
        } else {
            W::list_fence();
            const uint32_t k = x - 64u;
            const uint32_t cnt = 1u + W::readlane(W::load32_list(hi_row, W::splat(k), W::all(), 0u), 0u);
            E.encode(x + low + gsum_below(hi_row, k), cnt, n);
            W::append_list(hi_row, k, cnt - 1u + 10u);
        }
        n += 10u;
        if (n >= CBC_RESCALE) {
            uint32_t dummy = 0;
            E.dense_rescale(lo_tab, 64u, dummy);                                  /* dummy = 64 + the new low sum */
            low = dummy - 64u;
            W::list_fence();
            n = 256u + low + grescale(hi_row, 192u);
        }
        ntab = W::select(ln == t, W::splat(n), ntab);
        lsum = W::select(ln == t, W::splat(low), lsum);
    };
    auto gx_code = [&](uint32_t which, uint32_t x) {            /* the two bytes of a long gap: rare, all of it in global memory */
        uint32_t n = W::readlane(ntab, 8u + which);
        uint32_t *row = scr + CBC_LSCR_GX + 256u * which;
        W::list_fence();
        const uint32_t cnt = 1u + W::readlane(W::load32_list(row, W::splat(x), W::all(), 0u), 0u);
        E.encode(x + gsum_below(row, x), cnt, n);
        W::append_list(row, x, cnt - 1u + 10u);
        n += 10u;
        if (n >= CBC_RESCALE) { W::list_fence(); n = 256u + grescale(row, 256u); }
        ntab = W::select(ln == 8u + which, W::splat(n), ntab);
    };
    /* the six models that see one symbol per read: all six have coded r symbols when record r comes, n = 256 + 10 r */
    auto sp_code = [&](uint32_t list, uint32_t x, uint32_t r) {
        uint32_t *tab = lds + CBC_LLDS_SP + 64u * list;
        const uint32_t count = W::readlane(spc, list);
        const Mask live = ln < count;
        const V32 w = W::load32(tab, ln, live, 0u);
        const V32 k = w >> 24, e = w & 0xffffffu;
        const uint32_t lo = x + W::reduce_add(W::select(live & (k < x), e, W::splat(0u)));
        const uint64_t eq = W::ballot(live & (k == x));
        uint32_t cnt = 1u, idx = count;
        if (eq) { idx = W::ctz64(eq); cnt = 1u + W::readlane(e, idx); }
        E.encode(lo, cnt, 256u + 10u * r);
        if (idx >= 64u) { E.fail(CBC_ST_ASSERT); return; }     /* more than 64 reads: refused above */
        W::write_uni(tab, idx, (x << 24) | (cnt - 1u + 10u));
        if (!eq) spc = W::select(ln == list, W::splat(count + 1u), spc);
    };
    auto code_int = [&](uint32_t v) {
        E.regsparse_code(E.hkey, E.hexc, 0u, 8u, E.hc0, E.hn0, 256u, 1u, v >> 24, CBC_ST_ASSERT);
        E.regsparse_code(E.hkey, E.hexc, 8u, 8u, E.hc1, E.hn1, 256u, 1u, (v >> 16) & 0xffu, CBC_ST_ASSERT);
        E.regsparse_code(E.hkey, E.hexc, 16u, 8u, E.hc2, E.hn2, 256u, 1u, (v >> 8) & 0xffu, CBC_ST_ASSERT);
        E.regsparse_code(E.hkey, E.hexc, 24u, 8u, E.hc3, E.hn3, 256u, 1u, v & 0xffu, CBC_ST_ASSERT);
    };
    /* ---- the edits of a read reach the models in batches of up to 64, one lane per edit.  Within a batch every model is a
     * COUNTING model (no total can reach the rescale point inside it -- checked, else the batch goes the serial way): the
     * (cum, count, total) a symbol sees = the tables before the batch + what the lower lanes of the batch add, which
     * gathers, ballots and one 64-step compare loop give to all lanes at once.  The serial form of this -- three model
     * calls per edit on wave-uniform values -- kept the CU's one scalar unit 55 % busy (profiles/r03_ab_kernels.log:
     * 106 scalar + 80 vector instructions per coded symbol).  An edit word: M coordinate | kind << 16 | read base << 18 |
     * chars row << 21; carry_end / carry_pk: M coordinate after, and kind of, the previous edit of the read. ---- */
    V32 eb = W::splat(0u); uint32_t ecount = 0, carry_end = 0, carry_pk = 3u;
    uint32_t *trip = lds + CBC_LLDS_TRIP;
    auto serial_edit = [&](uint32_t g, uint32_t pk, uint32_t strand, uint32_t kind, uint32_t row, uint32_t base) {
        if (E.q_len >= 56u) E.drain();
        gap_code(2u * pk + strand, g < 255u ? g : 255u);
        if (g >= 255u) { gx_code(0u, ((g - 255u) >> 8) & 0xffu); gx_code(1u, (g - 255u) & 0xffu); }
        E.small_code(cbc_long_kind_base(pk), 3u, 8u, kind);
        if (kind != 2u) E.small_code(CBC_LT_CHARS + row * 8u, 5u, 8u, base);
    };
    auto flush_edits = [&](uint32_t strand) {
        const uint32_t m = ecount;
        ecount = 0;
        if (m == 0u || E.status != CBC_ST_OK) return;
        CBC_TS(0);                                                /* the walk that collected the batch */
        const Mask live = ln < m;
        const V32 mc = eb & 0xffffu, kind = (eb >> 16) & 3u, base = (eb >> 18) & 7u, row = (eb >> 21) & 7u;
        const V32 mend = mc + W::select(kind == 0u, W::splat(1u), W::splat(0u));
        const V32 pk = W::shift_up1(kind, carry_pk);
        const V32 g = mc - W::shift_up1(mend, carry_end);
        const V32 t = pk * 2u + strand;
        const uint32_t next_end = W::readlane(mend, m - 1u), next_pk = W::readlane(kind, m - 1u);
        /* lane-table models: prefix sums over the lanes of `small` give every context's (cum, total) by three gathers */
        const V32 P = W::scan_incl_add(E.small);
        const V32 kb = W::select(pk < 2u, pk * 3u, (pk - 2u) * 3u + 10u);             /* cbc_long_kind_base per lane */
        const V32 cb = row * 8u + CBC_LT_CHARS;
        auto below = [&](const V32 &first) -> V32 { return W::select(first == 0u, W::splat(0u), W::lane_gather(P, first - 1u)); };
        const V32 k_lo0 = W::select(kind == 0u, W::splat(0u), W::lane_gather(P, kb + kind - 1u) - below(kb)), k_n0 = W::lane_gather(P, kb + 2u) - below(kb);
        const V32 k_c0 = W::lane_gather(E.small, kb + kind);
        const V32 c_lo0 = W::select(base == 0u, W::splat(0u), W::lane_gather(P, cb + base - 1u) - below(cb)), c_n0 = W::lane_gather(P, cb + 4u) - below(cb);
        const V32 c_c0 = W::lane_gather(E.small, cb + base);
        const Mask has_base = live & (kind != 2u);
        const V32 g_n0 = W::lane_gather(ntab, t);
        /* the serial way: a gap that needs its two gx bytes, or a total within a batch of its rescale point (once in ~10^5 uses) */
        const uint64_t odd = W::ballot(live & ((g >= 255u) | (g_n0 + 640u >= CBC_RESCALE) | (k_n0 + 512u >= CBC_RESCALE))) |
                             W::ballot(has_base & (c_n0 + 512u >= CBC_RESCALE));
        if (odd) {
            for (uint32_t j = 0; j < m && E.status == CBC_ST_OK; j++)
                serial_edit(W::readlane(g, j), W::readlane(pk, j), strand, W::readlane(kind, j), W::readlane(row, j), W::readlane(base, j));
            carry_end = next_end; carry_pk = next_pk;
            return;
        }
        /* -- gap: the table rows before the batch -- */
        const Mask glo = live & (g < 64u), ghi = live & (g >= 64u);
        const V32 gk = g - 64u;                                   /* index in the global row (ghi lanes) */
        V32 g_e0 = W::load32(lds + CBC_LLDS_GAPLO, t * 64u + g, glo, 0u), g_pre = W::splat(0u);
        if (W::ballot(ghi)) {
            W::list_fence();
            g_e0 = W::select(ghi, W::load32_list(scr + CBC_LSCR_GAPHI, t * 192u + gk, ghi, 0u), g_e0);
            g_pre = W::select(ghi, W::lane_gather(lsum, t), g_pre);      /* everything the row holds below symbol 64 */
        }
        for (uint32_t tt = 0; tt < 8u; tt++) {
            const Mask mine = live & (t == tt);
            if (!W::ballot(mine)) continue;
            const V32 Pl = W::scan_incl_add(W::load32(lds + CBC_LLDS_GAPLO + 64u * tt, ln, W::all(), 0u));
            g_pre = W::select(mine & glo & (g != 0u), W::lane_gather(Pl, g - 1u), g_pre);
            if (W::ballot(mine & ghi)) {
                uint32_t run = 0;
                for (uint32_t c = 0; c < 3u; c++) {
                    const V32 Pc = W::scan_incl_add(W::load32_list(scr + CBC_LSCR_GAPHI + 192u * tt, ln + 64u * c, W::all(), 0u)) + run;
                    g_pre = g_pre + W::select(mine & ghi & (gk != 0u) & (((gk - 1u) >> 6) == c), W::lane_gather(Pc, (gk - 1u) & 63u), W::splat(0u));
                    run = W::readlane(Pc, 63u);
                }
            }
        }
        /* -- what the lower lanes of the batch add: same table and a smaller / the same gap; same table at all.  Bit-sliced: every
         *    lane keeps, as a 64-bit mask, the live lanes that agree with it on the table and on the bits of the gap looked at so far
         *    (one ballot per bit, the gap's from the top); a lane whose gap has the bit set counts the lower lanes of that set whose
         *    gap has it clear.  11 ballots and ~180 vector instructions for what a loop over the lanes did in ~750. -- */
        V32 ga = W::splat(0u), gb, gc;
        {
            const uint64_t lm = W::ballot(live);
            const V32 bl_lo = W::select(ln < 32u, (W::splat(1u) << (ln & 31u)) - 1u, W::splat(0xffffffffu));      /* the lanes below me */
            const V32 bl_hi = W::select(ln < 32u, W::splat(0u), (W::splat(1u) << (ln & 31u)) - 1u);
            V32 e_lo = W::splat((uint32_t)lm), e_hi = W::splat((uint32_t)(lm >> 32));
            auto narrow = [&](const V32 &key, uint32_t b, const bool count) {
                const V32 mine = (key >> b) & 1u;
                const uint64_t B = W::ballot(live & (mine != 0u));
                const V32 m1 = W::splat(0u) - mine;                                              /* all ones where my bit is set */
                const V32 nx_lo = W::splat((uint32_t)B) ^ m1, nx_hi = W::splat((uint32_t)(B >> 32)) ^ m1;   /* the lanes whose bit is not mine */
                if (count) ga = ga + W::popc_v(e_lo & nx_lo & bl_lo & m1) + W::popc_v(e_hi & nx_hi & bl_hi & m1);
                e_lo = e_lo & (nx_lo ^ 0xffffffffu); e_hi = e_hi & (nx_hi ^ 0xffffffffu);
            };
            for (uint32_t b = 0; b < 3u; b++) narrow(t, b, false);
            gc = (W::popc_v(e_lo & bl_lo) + W::popc_v(e_hi & bl_hi)) * 10u;
            for (uint32_t b = 8u; b-- > 0u; ) narrow(g, b, true);                                /* g < 255 here */
            gb = (W::popc_v(e_lo & bl_lo) + W::popc_v(e_hi & bl_hi)) * 10u;
            ga = ga * 10u;
        }
        const V32 g_lo = g + g_pre + ga, g_cnt = g_e0 + 1u + gb, g_n = g_n0 + gc;
        /* -- kind (4 contexts x 3) and chars (6 rows x 5): the batch's uses per (context, symbol), lower lanes by v_mbcnt -- */
        V32 ka = W::splat(0u), kbq = W::splat(0u), kc = W::splat(0u), ca = W::splat(0u), cbq = W::splat(0u), cc = W::splat(0u), upd = W::splat(0u);
        for (uint32_t ctx = 0; ctx < 4u; ctx++) for (uint32_t sy = 0; sy < 3u; sy++) {
            const uint64_t um = W::ballot(live & (pk == ctx) & (kind == sy));
            if (!um) continue;
            const V32 pp = W::prefix_popc(um) * 8u;
            const Mask in = live & (pk == ctx);
            kc = kc + W::select(in, pp, W::splat(0u));
            ka = ka + W::select(in & (kind > sy), pp, W::splat(0u));
            kbq = kbq + W::select(in & (kind == sy), pp, W::splat(0u));
            upd = upd + W::select(ln == cbc_long_kind_base(ctx) + sy, W::splat(8u * W::popc64(um)), W::splat(0u));
        }
        for (uint32_t rw = 0; rw < 6u; rw++) {
            if (!W::ballot(has_base & (row == rw))) continue;
            for (uint32_t sy = 0; sy < 5u; sy++) {
                const uint64_t um = W::ballot(has_base & (row == rw) & (base == sy));
                if (!um) continue;
                const V32 pp = W::prefix_popc(um) * 8u;
                const Mask in = has_base & (row == rw);
                cc = cc + W::select(in, pp, W::splat(0u));
                ca = ca + W::select(in & (base > sy), pp, W::splat(0u));
                cbq = cbq + W::select(in & (base == sy), pp, W::splat(0u));
                upd = upd + W::select(ln == CBC_LT_CHARS + rw * 8u + sy, W::splat(8u * W::popc64(um)), W::splat(0u));
            }
        }
        /* -- the tables after the batch -- */
        E.small = E.small + upd;
        W::lds_add(lds + CBC_LLDS_GAPLO, t * 64u + g, W::splat(10u), glo);
        W::list_add(scr + CBC_LSCR_GAPHI, t * 192u + gk, W::splat(10u), ghi);
        for (uint32_t tt = 0; tt < 8u; tt++) {
            const uint64_t um = W::ballot(live & (t == tt));
            if (!um) continue;
            const uint32_t nlo = W::popc64(W::ballot(glo & (t == tt)));
            ntab = ntab + W::select(ln == tt, W::splat(10u * W::popc64(um)), W::splat(0u));
            lsum = lsum + W::select(ln == tt, W::splat(10u * nlo), W::splat(0u));
        }
        /* -- the triples, in stream order: gap, kind, [base] per edit -- */
        const uint64_t delm = W::ballot(live & (kind == 2u));
        const V32 o = ln * 3u - W::prefix_popc(delm);
        const uint32_t total = 3u * m - W::popc64(delm);
        W::store32(trip, o, g_lo, live); W::store32(trip + 192u, o, g_cnt, live); W::store32(trip + 384u, o, g_n, live);
        W::store32(trip, o + 1u, k_lo0 + ka, live); W::store32(trip + 192u, o + 1u, k_c0 + kbq, live); W::store32(trip + 384u, o + 1u, k_n0 + kc, live);
        W::store32(trip, o + 2u, c_lo0 + ca, has_base); W::store32(trip + 192u, o + 2u, c_c0 + cbq, has_base); W::store32(trip + 384u, o + 2u, c_n0 + cc, has_base);
        CBC_TS(1);                                                /* the batch's model arithmetic */
        if (E.q_len) E.drain();                                   /* what the record's header left queued goes first */
        for (uint32_t c0 = 0; c0 < total; c0 += 64u) {
            const uint32_t cnt = total - c0 < 64u ? total - c0 : 64u;
            E.q_lo = W::load32(trip, ln + c0, ln < cnt, 0u); E.q_cnt = W::load32(trip + 192u, ln + c0, ln < cnt, 1u);
            E.q_n = W::load32(trip + 384u, ln + c0, ln < cnt, 1u);
            E.q_len = cnt;
            E.drain();
        }
        CBC_TS(7);                                                /* hand-over of the triples (waiting for a slot: 6) */
        carry_end = next_end; carry_pk = next_pk;
    };

    if (ROLE != CBC_ROLE_WALKER && E.status == CBC_ST_OK) { code_int(CBC_LONG_MAGIC); code_int(8u); E.drain(); }

    const uint4 *recs4 = (const uint4 *)(A.recs + rec_base);
    const uint8_t *seqb = A.seq + seq_base;
    const uint32_t *tokb = A.tok + tok_base;
    const uint8_t *refb = A.ref + ref_off;
    const uint64_t seq_avail = cbc_le64(seq_base, A.seq_bytes) ? A.seq_bytes - seq_base : 0;
    const uint64_t ref_avail = cbc_le64(ref_off, A.ref_bytes) ? A.ref_bytes - ref_off : 0;
    const uint32_t seq_lim = cbc_avail32(seq_avail, 0u), ref_lim = cbc_avail32(ref_avail, 0u);
    const V32 bo = ln * 4u;
    /* bytes [0, c) of a 256-byte chunk, 4 per lane: which of a lane's four are inside */
    auto chunk_mask = [&](uint32_t c) -> V32 {
        return W::select(bo + 4u <= c, W::splat(0xffffffffu), W::select(bo < c, (W::splat(1u) << ((W::splat(c) - bo) * 8u)) - 1u, W::splat(0u)));
    };

    V32 r_pos = W::splat(0u), r_fl = W::splat(0u), r_seq = W::splat(0u), r_tok = W::splat(0u);
    if (E.status == CBC_ST_OK) {
        W::load_rec(recs4, ln, ln < n_reads, r_pos, r_fl, r_seq, r_tok);
        V32 vrl = r_fl >> 16;
        Mask live = ln < n_reads;
        /* the compares read dwords: 3 readable bytes past a read's last base (the batch's 8 pad bytes) and past the reference run */
        Mask bad = live & ((vrl == 0u) | (r_pos == 0u) | (r_seq > seq_lim) | ((seq_lim - r_seq) < (vrl + 3u)) | (r_pos > ref_lim) | (r_tok >= n_tok_blk));
        uint64_t bb = W::ballot(bad);
        if (bb) { E.cur_read = W::ctz64(bb); E.fail(CBC_ST_ASSERT); }
    }
    for (uint32_t r = 0; r < n_reads && E.status == CBC_ST_OK; r++) {
        E.cur_read = r;
        const uint32_t pos = W::readlane(r_pos, r), flw = W::readlane(r_fl, r), rl = flw >> 16, strand = (flw >> 4) & 1u;
        const uint32_t so = W::readlane(r_seq, r), to = W::readlane(r_tok, r);
        const uint8_t *rdb = seqb + so;
        if (ROLE == CBC_ROLE_WALKER) {
            /* the half read r's edits go to is free once the model wavefront is done with read r - 2 */
            uint32_t gone = 0;
            while (r >= 2u && W::ctl_load(E.ctl + 3) + 2u <= r && !(gone = W::ctl_load(E.ctl + 7))) W::nap();
            if (gone) break;
        } else {
            if (E.q_len >= 32u) E.drain();
            if (r == 0u) {
                E.small_code(CBC_LT_SAMEREF, 2u, 10u, 1u);
                for (uint32_t q = 0; E.status == CBC_ST_OK; q++) {
                    uint32_t ch = (name_off + q < A.names_bytes) ? W::read_uni8(A.names, name_off + q) : 0u;
                    E.rname_code(E.prevChar, ch);
                    if ((q & 31u) == 31u) E.drain();
                    if (ch == 0u) break;
                    E.prevChar = ch;
                }
                E.drain();
            } else E.small_code(CBC_LT_SAMEREF, 2u, 10u, 0u);
            for (uint32_t k = 0; k < 4u; k++) sp_code(CBC_LS_LEN + k, (rl >> (8u * (3u - k))) & 0xffu, r);
            /* -- pos: the reference's model (compress_pos read_compression.c:113-159), any 31-bit step -- */
            if (pos < E.prevPos) { E.fail(CBC_ST_ASSERT); break; }
            E.pos_lit_code(pos - E.prevPos + 1u, pos_n);
            E.prevPos = pos;
            E.regsparse_code(E.fkey, E.fexc, 0u, CBC_CAP_FLAG, E.fcount, flag_n, 65536u, 8u, flw & 0xffffu, CBC_ST_CAP_FLAG);
            if (E.status != CBC_ST_OK) break;
        }

        const uint32_t hdr = W::read_uni(tokb, to), n_cig = hdr & 0xffffu;
        if (to + 2u + n_cig > n_tok_blk) { E.fail(CBC_ST_ASSERT); break; }

        /* -- the edits, in read order.  ONE walk along the CIGAR (the walker wavefront's) collects them into a half of the
         *    block's edit buffer in global memory; their number, which the stream carries first, is then known, and the
         *    model wavefront reads the half back a batch at a time into flush_edits().  A read with more edits than a half
         *    holds is walked a second time by the model wavefront itself, coding as it goes. -- */
        uint32_t *ebuf = scr + CBC_LSCR_EDITS + (r & 1u) * CBC_LONG_EDIT_HALF;
        CbcLWin<W> rw, fw;
        uint32_t over = 0;                                        /* first walk: the buffer is full, it only counts from there */
        auto walk = [&](const bool second) -> uint32_t {
            uint32_t i = 0, mcoord = 0, n = 0, stored = 0; uint32_t jr = pos - 1u;    /* read index, M bases consumed, edits, reference index */
            over = 0; ecount = 0;
            rw.reset(rdb, rl + 3u, 0u); fw.reset(refb, ref_lim, jr);
            auto emit = [&](uint32_t mc, uint32_t kind, uint32_t row, uint32_t base) {
                n++;
                if (over) return;
                eb = W::select(ln == ecount, W::splat(mc | (kind << 16) | (base << 18) | (row << 21)), eb);
                if (++ecount < 64u) return;
                if (second) { if (ROLE != CBC_ROLE_WALKER) flush_edits(strand); }
                else if (stored + 64u <= CBC_LONG_EDIT_HALF) { W::store32_list(ebuf, ln + stored, eb, W::all()); stored += 64u; ecount = 0; }
                else { over = 1u; ecount = 0; }
            };
            V32 tokv = W::splat(0u);
            for (uint32_t o = 0; o < n_cig && E.status == CBC_ST_OK; o++) {
                if ((o & 63u) == 0u) tokv = W::load32(tokb + to + 2u + o, ln, (ln + o) < n_cig, 0u);
                const uint32_t t = W::readlane(tokv, o & 63u), op = t & 15u, len = t >> 4;
                if (op == CBC_OP_M) {
                    if (len > rl - i || jr > ref_lim || len > ref_lim - jr || ref_lim - jr - len < 3u) { E.fail(CBC_ST_ASSERT); break; }
                    for (uint32_t b = 0; b < len && E.status == CBC_ST_OK; ) {
                        /* a piece of the run: at most 256 bases, all inside both 512-byte windows (+ 3 bytes of dword slack) */
                        rw.to(i + b); fw.to(jr + b);
                        uint32_t c = len - b < 256u ? len - b : 256u;
                        const uint32_t room_r = 509u - (i + b - rw.base), room_f = 509u - (jr + b - fw.base);
                        if (c > room_r) c = room_r;
                        if (c > room_f) c = room_f;
                        const V32 rd = rw.dwords(i + b), rf = fw.dwords(jr + b);
                        const V32 x = (rd ^ rf) & chunk_mask(c);
                        uint64_t mm = W::ballot(x != 0u);
                        while (mm && E.status == CBC_ST_OK) {
                            const uint32_t k = W::ctz64(mm); mm &= mm - 1ull;
                            const uint32_t xk = W::readlane(x, k), rdk = W::readlane(rd, k), rfk = W::readlane(rf, k);
                            for (uint32_t q = 0; q < 4u; q++) if ((xk >> (8u * q)) & 0xffu)
                                emit(mcoord + b + 4u * k + q, 0u, cbc_basepair((rfk >> (8u * q)) & 0xffu), cbc_basepair((rdk >> (8u * q)) & 0xffu));
                        }
                        b += c;
                    }
                    i += len; jr += len; mcoord += len;
                } else if (op == CBC_OP_I || op == CBC_OP_S) {
                    if (len > rl - i) { E.fail(CBC_ST_ASSERT); break; }
                    for (uint32_t b = 0; b < len && E.status == CBC_ST_OK; ) {          /* the inserted bases, through the read window */
                        rw.to(i + b);
                        uint32_t c = len - b < 256u ? len - b : 256u;
                        const uint32_t room = 509u - (i + b - rw.base);
                        if (c > room) c = room;
                        const V32 ib = rw.dwords(i + b);
                        for (uint32_t q = 0; q < c && E.status == CBC_ST_OK; q++)
                            emit(mcoord, 1u, 5u, cbc_basepair((W::readlane(ib, q >> 2) >> (8u * (q & 3u))) & 0xffu));
                        b += c;
                    }
                    i += len;
                } else if (op == CBC_OP_D) {
                    if (jr > ref_lim || len > ref_lim - jr) { E.fail(CBC_ST_ASSERT); break; }
                    for (uint32_t c = 0; c < len && E.status == CBC_ST_OK; c++) emit(mcoord, 2u, 0u, 0u);
                    jr += len;
                } else { E.fail(CBC_ST_UNSUPPORTED); break; }
                if (n > 0xffffu) { E.fail(CBC_ST_ASSERT); break; }                    /* the edit count is a u16 in the stream */
            }
            if (E.status == CBC_ST_OK && i != rl) E.fail(CBC_ST_ASSERT);              /* the CIGAR must consume the read exactly */
            if (second) { if (ROLE != CBC_ROLE_WALKER) flush_edits(strand); return n; }
            if (!over && ecount) { W::store32_list(ebuf, ln + stored, eb, ln < ecount); stored += ecount; }
            ecount = 0;
            return n;
        };
        /* The walk, one lane per CIGAR run, 64 runs at a time (fast_walk): where every run starts in the read, in the reference
         * and in M coordinates comes from three scans over the token lengths; every M lane compares ITS run's bases (a 64-bit
         * mismatch mask per 64 bases of the run: the loop counter is wave-uniform, so building the mask is two shifts), I / S / D
         * lanes count their length; a scan of the counts says where each run's edits go in the buffer, and the lanes put them
         * there -- mismatches bit by bit out of the mask, inserted bases by their index.  The run-by-run walk above cost ~2900
         * cycles per run (a chain of scalar <-> vector dependencies) and was 72 % of the model wavefront (profiles/r03_ab_kernels.log);
         * it stays as the fallback for what the lane form does not take: an M run longer than 256 bases, an I / S / D run
         * longer than 64, more edits than the buffer holds.  0xffffffff = fall back. */
        auto basepair_v = [&](const V32 &c) -> V32 {
            return W::select(c == (uint32_t)'A', W::splat(0u), W::select(c == (uint32_t)'C', W::splat(1u),
                   W::select(c == (uint32_t)'G', W::splat(2u), W::select(c == (uint32_t)'T', W::splat(3u), W::splat(4u)))));
        };
        auto fast_walk = [&]() -> uint32_t {
            uint32_t i0 = 0, j0 = pos - 1u, m0 = 0, n = 0;
            for (uint32_t ob = 0; ob < n_cig && E.status == CBC_ST_OK; ob += 64u) {
                const Mask live = (ln + ob) < n_cig;
                const V32 tk = W::load32(tokb + to + 2u + ob, ln, live, 0u);
                const V32 op = tk & 15u, len = tk >> 4;
                const Mask isM = live & (op == CBC_OP_M), isI = live & ((op == CBC_OP_I) | (op == CBC_OP_S)), isD = live & (op == CBC_OP_D);
                if (W::ballot(live & !(isM | isI | isD))) { E.fail(CBC_ST_UNSUPPORTED); return 0u; }
                if (W::ballot((isM & (len > 256u)) | ((isI | isD) & (len > 64u)))) return 0xffffffffu;
                const V32 di = W::select(isM | isI, len, W::splat(0u)), dj = W::select(isM | isD, len, W::splat(0u)), dm = W::select(isM, len, W::splat(0u));
                const V32 Si = W::scan_incl_add(di), Sj = W::scan_incl_add(dj), Sm = W::scan_incl_add(dm);
                const V32 io = Si - di + i0, jo = Sj - dj + j0, mo = Sm - dm + m0;       /* no overflow: 64 lengths of <= 256 */
                const Mask bad = (isM & ((io > rl) | (len > rl - io) | (jo > ref_lim) | (len > ref_lim - jo) | ((ref_lim - jo - len) < 3u))) |
                                 (isI & ((io > rl) | (len > rl - io))) | (isD & ((jo > ref_lim) | (len > ref_lim - jo)));
                if (W::ballot(bad)) { E.fail(CBC_ST_ASSERT); return 0u; }
                /* the 64-bit mismatch mask of segment `sg` (bases 64 sg .. 64 sg + 63) of every M lane's run.  Four dwords per
                 * step, their eight loads issued before the first is looked at: a step then waits for memory once, not four times
                 * (the loop counter is wave-uniform, so placing a step's bits in the mask is a shift by a constant) */
                auto seg_mask = [&](uint32_t sg, V32 &lo, V32 &hi) {
                    lo = W::splat(0u); hi = W::splat(0u);
                    for (uint32_t st = 0; st < 4u; st++) {
                        const uint32_t p0 = 64u * sg + 16u * st;
                        if (!W::ballot(isM & (len > p0))) break;
                        V32 rd[4], rf[4];
                        for (uint32_t q = 0; q < 4u; q++) {
                            const Mask m = isM & (len > p0 + 4u * q);
                            rd[q] = W::load32_bytes(rdb, io + (p0 + 4u * q), m); rf[q] = W::load32_bytes(refb, jo + (p0 + 4u * q), m);
                        }
                        V32 bits = W::splat(0u);
                        for (uint32_t q = 0; q < 4u; q++) {
                            const uint32_t p = p0 + 4u * q;
                            const Mask m = isM & (len > p);
                            const V32 left = len - p;                           /* >= 1 where m */
                            V32 x = (rd[q] ^ rf[q]) & W::select(left >= 4u, W::splat(0xffffffffu), (W::splat(1u) << (left * 8u)) - 1u);
                            x = W::select(m, x, W::splat(0u));
                            V32 y = x | (x >> 1); y = y | (y >> 2); y = y | (y >> 4);             /* bit 0 of every byte: the byte is non-zero */
                            bits = bits | (((((y & 0x01010101u) * 0x01020408u) >> 24) & 0xfu) << (4u * q));
                        }
                        if (st < 2u) lo = lo | (bits << (16u * st)); else hi = hi | (bits << (16u * (st - 2u)));
                    }
                };
                /* -- counts (the masks of the runs' first 64 bases are kept: most runs end there) -- */
                V32 cnt = W::select(isI | isD, len, W::splat(0u));
                V32 lo0 = W::splat(0u), hi0 = W::splat(0u);
                for (uint32_t sg = 0; sg < 4u; sg++) {
                    if (!W::ballot(isM & (len > 64u * sg))) break;
                    V32 lo, hi; seg_mask(sg, lo, hi);
                    if (sg == 0u) { lo0 = lo; hi0 = hi; }
                    cnt = cnt + W::popc_v(lo) + W::popc_v(hi);
                }
                const V32 Sc = W::scan_incl_add(cnt);
                const uint32_t total = W::readlane(Sc, 63u);
                if (n + total > CBC_LONG_EDIT_HALF) return 0xffffffffu;
                V32 w = Sc - cnt + n;                                        /* where this run's next edit goes */
                /* -- mismatches: out of the masks, lowest position first, four per step (positions, then their eight byte loads,
                 *    then the stores) -- */
                for (uint32_t sg = 0; sg < 4u; sg++) {
                    if (!W::ballot(isM & (len > 64u * sg))) break;
                    V32 lo, hi;
                    if (sg == 0u) { lo = lo0; hi = hi0; } else seg_mask(sg, lo, hi);
                    while (W::ballot((lo | hi) != 0u)) {
                        V32 pp[4], rb[4], fb[4]; Mask hs[4];
                        for (uint32_t q = 0; q < 4u; q++) {
                            const Mask has = (lo | hi) != 0u, inlo = lo != 0u;
                            pp[q] = W::select(inlo, W::ctz_v(lo), W::ctz_v(hi) + 32u) + 64u * sg;
                            lo = W::select(inlo, lo & (lo - 1u), lo); hi = W::select(has & !inlo, hi & (hi - 1u), hi);
                            hs[q] = has;
                            rb[q] = W::load8(rdb, io + pp[q], has); fb[q] = W::load8(refb, jo + pp[q], has);
                        }
                        for (uint32_t q = 0; q < 4u; q++) {
                            W::store32_list(ebuf, w, (mo + pp[q]) | (basepair_v(rb[q]) << 18) | (basepair_v(fb[q]) << 21), hs[q]);
                            w = w + W::select(hs[q], W::splat(1u), W::splat(0u));
                        }
                    }
                }
                /* -- inserted / clipped bases and deleted positions, four per step -- */
                for (uint32_t k0 = 0; k0 < 64u; k0 += 4u) {
                    if (!W::ballot((isI | isD) & (len > k0))) break;
                    V32 rb[4];
                    for (uint32_t q = 0; q < 4u; q++) rb[q] = W::load8(rdb, io + (k0 + q), isI & (len > k0 + q));
                    for (uint32_t q = 0; q < 4u; q++) {
                        const Mask mi = isI & (len > k0 + q), md = isD & (len > k0 + q);
                        W::store32_list(ebuf, w, W::select(mi, mo | (1u << 16) | (basepair_v(rb[q]) << 18) | (5u << 21), mo | (2u << 16)), mi | md);
                        w = w + W::select(mi | md, W::splat(1u), W::splat(0u));
                    }
                }
                n += total;
                i0 += W::readlane(Si, 63u); j0 += W::readlane(Sj, 63u); m0 += W::readlane(Sm, 63u);
                if (n > 0xffffu) { E.fail(CBC_ST_ASSERT); return 0u; }                 /* the edit count is a u16 in the stream */
            }
            if (E.status == CBC_ST_OK && i0 != rl) E.fail(CBC_ST_ASSERT);             /* the CIGAR must consume the read exactly */
            return n;
        };
        uint32_t ne = 0, walk_again = 0;
        if (ROLE == CBC_ROLE_MODEL) {
            CBC_TS(2);                                            /* record header */
            while (W::ctl_load(E.ctl + 2) <= r) W::nap();         /* acquire: read r has been walked (or the walker has failed) */
            CBC_TS(4);                                            /* waiting for the walker */
            const uint32_t wst = W::read_uni(E.ctl, 6u);
            if (wst) { if (E.status == CBC_ST_OK) { E.status = wst & 0xffffu; E.fail_read = wst >> 16; } break; }
            const uint32_t w = W::read_uni(E.ctl, 4u + (r & 1u));
            ne = w & 0x7fffffffu; walk_again = w >> 31;
        } else {
            ne = fast_walk();
            over = 0;
            if (ne == 0xffffffffu) ne = walk(false);
            walk_again = over;
            if (E.status != CBC_ST_OK) break;
        }
        if (ROLE == CBC_ROLE_WALKER) {
            W::list_fence();                                      /* release: the edits are in memory before the count says so */
            W::write_uni(E.ctl, 4u + (r & 1u), ne | (walk_again << 31));
            W::ctl_store(E.ctl + 2, r + 1u);
            continue;
        }
        if (ROLE == CBC_ROLE_FUSED) CBC_TS(2);                    /* record header + the walk */
        sp_code(CBC_LS_NE, ne >> 8, r); sp_code(CBC_LS_NE + 1u, ne & 0xffu, r);
        carry_end = 0; carry_pk = 3u;
        if (walk_again) (void)walk(true);                         /* more edits than a half holds: collect and code in one go */
        else {
            W::list_fence();
            for (uint32_t k = 0; k < ne && E.status == CBC_ST_OK; k += 64u) {
                eb = W::load32_list(ebuf, ln + k, (ln + k) < ne, 0u);
                ecount = ne - k < 64u ? ne - k : 64u;
                flush_edits(strand);
            }
        }
        if (ROLE == CBC_ROLE_MODEL) W::ctl_store(E.ctl + 3, r + 1u);     /* the half is free again */
    }
    if (ROLE == CBC_ROLE_WALKER) {
        /* a failure here reaches the model wavefront through the mailbox; it waits for nothing else of this wavefront */
        if (E.status != CBC_ST_OK) { W::write_uni(E.ctl, 6u, E.status | (E.fail_read << 16)); W::ctl_store(E.ctl + 2, 0x7fffffffu); }
        return;
    }
    if (ROLE == CBC_ROLE_MODEL) W::ctl_store(E.ctl + 7, 1u);     /* the walker must not wait for a half any more */
    if (E.status == CBC_ST_OK) {
        E.cur_read = n_reads;
        if (E.q_len >= 32u) E.drain();
        E.small_code(CBC_LT_SAMEREF, 2u, 10u, 1u);
        E.rname_code(E.prevChar, (uint32_t)'\n');
        E.rname_code((uint32_t)'\n', 0u);
    }
#if defined(CBC_STAMP) && defined(__HIP_DEVICE_COMPILE__)
    if (ROLE == CBC_ROLE_MODEL && out_cap >= 512u) { CBC_TS(3); for (int i = 0; i < 16; i++) {      /* diagnostic build: payload area, second 128 bytes */
        W::write_uni(E.out32 + 32, 2 * i, (uint32_t)E.t_sum[i]); W::write_uni(E.out32 + 32, 2 * i + 1, (uint32_t)(E.t_sum[i] >> 32)); } }
#endif
    if (ROLE == CBC_ROLE_MODEL) { E.publish(CBC_BF_LAST, 0ull); return; }   /* with whatever is still queued, and this wavefront's status */
    uint32_t nbytes = 0;
    if (E.status == CBC_ST_OK) E.drain_q();
    if (E.status == CBC_ST_OK) { E.flush_recs(); nbytes = E.finish(); }
    if (E.status != CBC_ST_OK) nbytes = 0;
    V32 resv = W::select(ln == 0u, W::splat(nbytes), W::select(ln == 1u, W::splat(E.status),
               W::select(ln == 2u, W::splat(E.nsym), W::splat(E.fail_read))));
    W::store32((uint32_t *)(A.results + blk), ln, resv, ln < 4u);
}

/* ---------------------------------------------------------------------------------------------------------------
 * decode: records into recs[] (pos block-local, flag, length, offset of the bases inside the block), bases compact
 * at seq + seq_base.  cbc_dec_block_desc.reserved[0] = bases the block holds (from the container index).
 * ------------------------------------------------------------------------------------------------------------- */
template <class W>
CBC_FN void cbc_long_decode(const cbc_dec_args &A, uint32_t blk, uint32_t *lds)
{
    typedef typename W::V32 V32;
    typedef typename W::Mask Mask;
    const V32 ln = W::lane();
    const cbc_dec_block_desc *bd = A.blocks + blk;
    CbcDec<W, true> D;
    const uint64_t in_off = bd->in_off, ref_off = bd->ref_off, rec_base = bd->rec_base, seq_base = bd->seq_base;
    const uint32_t in_bytes = bd->in_bytes, n_reads = bd->n_reads, blk_bases = bd->reserved[0];

    D.status = CBC_ST_OK; D.nsym = 0; D.fail_read = 0; D.cur_read = 0;
    D.l = W::dv(0u); D.rng = W::dv(CBC_M26 + 1u); D.d = W::dv(0u); D.acc = 0; D.navail = 0; D.widx = 0; D.wordv = W::splat(0u);
    D.inb = A.in + in_off;
    D.lds = lds; D.cap_pos = A.cap_pos; D.cap_var = 0; D.L0 = 256u; D.evp = nullptr; D.vtab = nullptr;
    D.rname_key = lds + CBC_LLDS_RNKEY; D.rname_exc = lds + CBC_LLDS_RNEXC; D.rn_cap = CBC_CAP_NAME; D.histp = nullptr;
    D.pos_valp = lds + CBC_LLDS_ROLE; D.pos_cntp = D.pos_valp + A.cap_pos;
    D.fsp_key = D.fsp_exc = nullptr; D.fsp_count = 0; D.pos_ov_valp = D.pos_ov_cntp = nullptr; D.pos_lds_cap = 0xffffffc0u; D.palpha = nullptr;
    bool args_ok = cbc_fits64(in_off, ((uint64_t)in_bytes + 3u) & ~3ull, A.in_bytes) && cbc_fits64(rec_base, n_reads, A.n_recs) &&
                   cbc_fits64(seq_base, (uint64_t)blk_bases + 8u, A.seq_bytes) && cbc_le64(ref_off, A.ref_bytes) && A.cap_pos >= 2u &&
                   n_reads <= 64u && A.var_scratch != nullptr && cbc_le64(((uint64_t)blk + 1u) * CBC_LONG_TABLE_WORDS, A.var_scratch_words);
    D.nwords_in = (in_bytes + 3u) >> 2;
    D.tail_valid = in_bytes & 3u;
    if (!args_ok) { D.nwords_in = 0; D.fail(CBC_ST_ASSERT); }
    for (uint32_t b = 0; b < CBC_LLDS_ROLE; b += 64u) W::store32(lds, ln + b, W::splat(0u), (ln + b) < CBC_LLDS_ROLE);
    if (args_ok) for (uint32_t b = 0; b < CBC_LONG_TABLE_WORDS; b += 64u)
        W::store32_list(A.var_scratch + (uint64_t)blk * CBC_LONG_TABLE_WORDS, ln + b, W::splat(0u), W::all());
    D.rlen_n = 0; D.rl123_c0 = 0; D.rl123_n = 0; D.snps_n = 0; D.indels_n = 0; D.rn_count = 0;
    D.pos_card = 1u; D.pos_n = 1u; D.nev = 0; D.nev1 = 0;
    D.pval = W::splat(0xffffffffu); D.pcnt = W::select(ln == 0u, W::splat(1u), W::splat(0u));
    D.fkey = W::splat(0u); D.fexc = W::splat(0u); D.fcount = 0; D.fn = 65536u;
    D.hkey = W::splat(0u); D.hexc = W::splat(0u);
    D.hc0 = D.hc1 = D.hc2 = D.hc3 = 0; D.hn0 = D.hn1 = D.hn2 = D.hn3 = 256u;
    {
        V32 s = W::select(ln < 16u, W::splat(1u), W::splat(0u));
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
    V32 ntab = W::splat(256u), lsum = W::splat(0u), spc = W::splat(0u);
    uint32_t *scr = A.var_scratch + (uint64_t)blk * CBC_LONG_TABLE_WORDS;
    /* the symbol of a dense table row in global memory whose cumulative interval holds tg; `first` = cum of the row's entry 0 */
    auto gsearch = [&](const uint32_t *row, uint32_t card, uint32_t first, uint32_t tg, uint32_t &lo, uint32_t &cnt) -> uint32_t {
        uint32_t run = first, found = CBC_NOMEMO;
        for (uint32_t b = 0; b < card && found == CBC_NOMEMO; b += 64u) {
            const Mask m = (ln + b) < card;
            const V32 c = W::select(m, W::load32_list(row, ln + b, m, 0u) + 1u, W::splat(0u));     /* counts */
            const V32 inc = W::scan_incl_add(c) + run;
            const uint64_t h = W::ballot(m & ((inc - c) <= tg) & (tg < inc));
            if (h) { const uint32_t hl = W::ctz64(h); found = b + hl; cnt = W::readlane(c, hl); lo = W::readlane(inc, hl) - cnt; }
            run = W::readlane(inc, 63u);
        }
        if (found == CBC_NOMEMO) D.fail(CBC_ST_ASSERT);
        return found;
    };
    auto grescale = [&](uint32_t *row, uint32_t card) -> uint32_t {
        V32 a = W::splat(0u);
        for (uint32_t b = 0; b < card; b += 64u) {
            const Mask m = (ln + b) < card;
            const V32 e = (W::load32_list(row, ln + b, m, 0u) + 1u) >> 1;
            W::store32_list(row, ln + b, e, m);
            a = a + W::select(m, e, W::splat(0u));
        }
        return W::reduce_add(a);
    };
    auto gap_dec = [&](uint32_t t) -> uint32_t {              /* cbc_long_encode: gap_code */
        uint32_t n = W::readlane(ntab, t), low = W::readlane(lsum, t), x = 0;
        uint32_t *lo_tab = lds + CBC_LLDS_GAPLO + 64u * t, *hi_row = scr + CBC_LSCR_GAPHI + 192u * t;
        if (!D.tag_ok(n)) return 0u;
        const V32 e = W::load32(lo_tab, ln, W::all(), 0u);
        const V32 cum = W::scan_incl_add(e) + ln + 1u;        /* lane s: cum of symbol s + 1 */
        uint32_t ql, qh;
        if (D.prefix_find(cum, W::all(), 0u, n, x, ql, qh)) {  /* symbols 0..63: one per lane, the search by scaled bounds */
            D.step_q(ql, qh);
            W::write_uni(lo_tab, x, W::readlane(e, x) + 10u);
            low += 10u;
        } else {
            const uint32_t tg = D.target(n);
            if (D.status != CBC_ST_OK) return 0u;
            W::list_fence();
            uint32_t lo = 0, cnt = 0;
            const uint32_t k = gsearch(hi_row, 192u, 64u + low, tg, lo, cnt);
            if (D.status != CBC_ST_OK) return 0u;
            D.step(lo, cnt, n);
            W::append_list(hi_row, k, cnt - 1u + 10u);
            x = 64u + k;
        }
        n += 10u;
        if (n >= CBC_RESCALE) {
            uint32_t dummy = 0;
            D.dense_rescale(lo_tab, 64u, dummy);
            low = dummy - 64u;
            W::list_fence();
            n = 256u + low + grescale(hi_row, 192u);
        }
        ntab = W::select(ln == t, W::splat(n), ntab);
        lsum = W::select(ln == t, W::splat(low), lsum);
        return x;
    };
    auto gx_dec = [&](uint32_t which) -> uint32_t {
        uint32_t n = W::readlane(ntab, 8u + which);
        uint32_t *row = scr + CBC_LSCR_GX + 256u * which;
        const uint32_t tg = D.target(n);
        if (D.status != CBC_ST_OK) return 0u;
        W::list_fence();
        uint32_t lo = 0, cnt = 0;
        const uint32_t x = gsearch(row, 256u, 0u, tg, lo, cnt);
        if (D.status != CBC_ST_OK) return 0u;
        D.step(lo, cnt, n);
        W::append_list(row, x, cnt - 1u + 10u);
        n += 10u;
        if (n >= CBC_RESCALE) { W::list_fence(); n = 256u + grescale(row, 256u); }
        ntab = W::select(ln == 8u + which, W::splat(n), ntab);
        return x;
    };
    /* the six models that see one symbol per read (cbc_long_encode: sp_code): sparse (value << 24 | excess) lists */
    auto sp_dec = [&](uint32_t list, uint32_t r) -> uint32_t {
        uint32_t *tab = lds + CBC_LLDS_SP + 64u * list;
        const uint32_t count = W::readlane(spc, list), n = 256u + 10u * r;
        const Mask live = ln < count;
        const V32 w = W::load32(tab, ln, live, 0u);
        const uint32_t tg = D.target(n);
        if (D.status != CBC_ST_OK) return 0u;
        uint32_t lo, cnt, hl; bool hit;
        const uint32_t x = D.pairs_search(w >> 24, w & 0xffffffu, live, 0u, count, tg, lo, cnt, hl, hit);
        if (x >= 256u || (!hit && count >= 64u)) { D.fail(CBC_ST_ASSERT); return 0u; }
        D.step(lo, cnt, n);
        W::write_uni(tab, hit ? hl : count, (x << 24) | (cnt - 1u + 10u));
        if (!hit) spc = W::select(ln == list, W::splat(count + 1u), spc);
        return x;
    };
    auto dec_int = [&]() -> uint32_t {
        uint32_t v = D.regsparse_dec(D.hkey, D.hexc, 0u, 8u, D.hc0, D.hn0, 256u, 1u, CBC_ST_ASSERT) << 24;
        v |= D.regsparse_dec(D.hkey, D.hexc, 8u, 8u, D.hc1, D.hn1, 256u, 1u, CBC_ST_ASSERT) << 16;
        v |= D.regsparse_dec(D.hkey, D.hexc, 16u, 8u, D.hc2, D.hn2, 256u, 1u, CBC_ST_ASSERT) << 8;
        v |= D.regsparse_dec(D.hkey, D.hexc, 24u, 8u, D.hc3, D.hn3, 256u, 1u, CBC_ST_ASSERT);
        return v;
    };
    if (D.status == CBC_ST_OK) D.d = W::dv(D.take(26u));
    if (D.status == CBC_ST_OK && dec_int() != CBC_LONG_MAGIC) D.fail(CBC_ST_UNSUPPORTED);
    if (D.status == CBC_ST_OK && dec_int() != 8u) D.fail(CBC_ST_UNSUPPORTED);

    uint4 *recs4 = (uint4 *)(A.recs + rec_base);
    uint8_t *seqo = A.seq + seq_base;
    const uint8_t *refb = A.ref + ref_off;
    const uint64_t ref_avail = cbc_le64(ref_off, A.ref_bytes) ? A.ref_bytes - ref_off : 0;
    const uint32_t ref_lim = cbc_avail32(ref_avail, 0u);
    uint32_t so = 0;                                             /* bases written so far in this block */
    /* a run of matched bases is the reference itself: g bytes from refb + jr to seqo + at, 64 per step */
    auto copy_run = [&](uint32_t at, uint32_t jr, uint32_t g) {
        const uint32_t gb = W::uni(g);
        for (uint32_t b = 0; b < gb; b += 64u) {
            const V32 v = W::load8(refb, ln + (jr + b), (ln + b) < g);
            W::store8(seqo, ln + (at + b), v, (ln + b) < g);
        }
    };
    for (uint32_t r = 0; r < n_reads && D.status == CBC_ST_OK; r++) {
        D.cur_read = r;
        uint32_t sr = D.small_dec(CBC_LT_SAMEREF, 2u, 10u);
        if (D.status != CBC_ST_OK) break;
        if (sr != (r == 0u ? 1u : 0u)) { D.fail(CBC_ST_ASSERT); break; }
        if (r == 0u) {
            for (uint32_t q = 0; q < CBC_CAP_NAME && D.status == CBC_ST_OK; q++) {
                uint32_t ch = D.rname_dec(D.prevChar);
                if (ch == 0u) break;
                if (ch == (uint32_t)'\n' && q == 0u) { D.fail(CBC_ST_ASSERT); break; }
                D.prevChar = ch;
            }
            if (D.status != CBC_ST_OK) break;
        }
        uint32_t rl = 0;
        for (uint32_t k = 0; k < 4u && D.status == CBC_ST_OK; k++) rl = (rl << 8) | sp_dec(CBC_LS_LEN + k, r);
        if (D.status != CBC_ST_OK) break;
        if (rl == 0u || rl > 65535u || rl > blk_bases - so) { D.fail(CBC_ST_ASSERT); break; }
        uint32_t x = D.pos_dec();
        if (D.status != CBC_ST_OK) break;
        if (x < 1u) { D.fail(CBC_ST_ASSERT); break; }
        uint32_t pos = D.prevPos + x - 1u;
        if (pos < D.prevPos || pos == 0u || pos > ref_lim) { D.fail(CBC_ST_ASSERT); break; }
        D.prevPos = pos;
        uint32_t flag = D.regsparse_dec(D.fkey, D.fexc, 0u, CBC_CAP_FLAG, D.fcount, D.fn, 65536u, 8u, CBC_ST_CAP_FLAG);
        const uint32_t strand = (flag >> 4) & 1u;
        uint32_t ne = sp_dec(CBC_LS_NE, r) << 8; ne |= sp_dec(CBC_LS_NE + 1u, r);
        if (D.status != CBC_ST_OK) break;
        uint32_t i = 0, jr = pos - 1u, pk = 3u;
        for (uint32_t k = 0; k < ne && D.status == CBC_ST_OK; k++) {
            uint32_t g = gap_dec(2u * pk + strand);
            if (g == 255u) { uint32_t hi = gx_dec(0u); g = 255u + ((hi << 8) | gx_dec(1u)); }
            uint32_t kind = D.small_dec(cbc_long_kind_base(pk), 3u, 8u);
            if (D.status != CBC_ST_OK) break;
            if (g > rl - i || jr > ref_lim || g > ref_lim - jr) { D.fail(CBC_ST_ASSERT); break; }
            copy_run(so + i, jr, g);
            i += g; jr += g;
            if (kind == 0u) {
                if (i >= rl || jr >= ref_lim) { D.fail(CBC_ST_ASSERT); break; }
                uint32_t alt = D.small_dec(CBC_LT_CHARS + cbc_basepair(W::read_uni8(refb, jr)) * 8u, 5u, 8u);
                W::store8(seqo, W::splat(so + i), W::splat(cbc_basechar(alt)), ln == 0u);
                i++; jr++;
            } else if (kind == 1u) {
                if (i >= rl) { D.fail(CBC_ST_ASSERT); break; }
                uint32_t alt = D.small_dec(CBC_LT_CHARS + 5u * 8u, 5u, 8u);
                W::store8(seqo, W::splat(so + i), W::splat(cbc_basechar(alt)), ln == 0u);
                i++;
            } else jr++;
            pk = kind;
        }
        if (D.status != CBC_ST_OK) break;
        if (jr > ref_lim || rl - i > ref_lim - jr) { D.fail(CBC_ST_ASSERT); break; }
        copy_run(so + i, jr, rl - i);
        V32 rv0 = W::splat(pos), rv1 = W::splat(flag | (rl << 16)), rv2 = W::splat(so), rv3 = W::splat(0u);
        W::store_rec(recs4, W::splat(r), ln == 0u, rv0, rv1, rv2, rv3);
        so += rl;
    }
    if (D.status == CBC_ST_OK) {                                  /* sentinel */
        D.cur_read = n_reads;
        uint32_t sr = D.small_dec(CBC_LT_SAMEREF, 2u, 10u);
        if (D.status == CBC_ST_OK && sr != 1u) D.fail(CBC_ST_ASSERT);
        if (D.status == CBC_ST_OK) {
            uint32_t ch = D.rname_dec(D.prevChar);
            if (D.status == CBC_ST_OK && ch != (uint32_t)'\n') D.fail(CBC_ST_ASSERT);
            if (D.status == CBC_ST_OK && D.rname_dec((uint32_t)'\n') != 0u) D.fail(CBC_ST_ASSERT);
        }
        if (D.status == CBC_ST_OK && so != blk_bases) D.fail(CBC_ST_ASSERT);      /* container index and stream disagree */
    }
    V32 resv = W::select(ln == 0u, W::splat(D.status == CBC_ST_OK ? n_reads : D.cur_read), W::select(ln == 1u, W::splat(D.status),
               W::select(ln == 2u, W::splat(D.nsym), W::splat(D.fail_read))));
    W::store32((uint32_t *)(A.results + blk), ln, resv, ln < 4u);
}

#endif /* CBC_LONG_BODY_H */
