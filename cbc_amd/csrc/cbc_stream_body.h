/*
 * cbc_stream_body.h -- the WHOLE-FILE stream: the reference's own output format ("compat" mode).
 *
 *   compress()    src/compression.c:112-170   one arithmetic stream per file; the models are never reset, not
 *                                             even at a contig change (:58-64 resets only cumsumP / snpInRef,
 *                                             compress_pos :123-124 resets prevPos)
 *   decompress()  src/compression.c:173-216
 *
 * The block kernels (cbc_encode_body.h / cbc_decode_body.h) cut the record stream into independent blocks and
 * use, inside a block, what holds while no adaptive total can reach the 2^20 rescale point: counting-model
 * closed forms, LDS-sized sparse tables.  None of that holds for a file of millions of records, so this body
 * codes ONE stream with ONE wavefront and the GENERAL form of every model (stream_model.c:31-51: add the step,
 * and once the total reaches 2^20 halve every count and add one):
 *   - same_ref, match, chars: literal counts in a lane table (CbcEnc::small_code / CbcDec::small_dec)
 *   - rlength[0], snps, indels: dense excess tables in LDS with their rescale sweep (dense_code / dense_dec);
 *     rlength[1..3] only ever see symbol 0 (quirk Q1): two scalars (count of 0, total) with the same rescale
 *   - pos: the alphabet in order of appearance with LITERAL counts in LDS (cap_pos entries), linear lookup;
 *     pos_alpha's four byte models are derived from the registered values (valid while 256 + 10 * cap_pos < 2^20)
 *   - flag, codebook: value/excess pairs in registers (<= CBC_CAP_FLAG distinct FLAG values per file)
 *   - rname: (context, char) pairs in LDS (cap_name of them)
 *   - var: the reference's dense 65535 x L0 table as excess words in GLOBAL memory (39 MB at L0 = 150), one
 *     round trip per var symbol
 * The record stream arrives as SEGMENTS (cbc_block_desc): consecutive records of one contig with their POS as in
 * the SAM (not rebased); a segment whose name_off differs from its predecessor's starts a new contig: same_ref 1,
 * the name, prevPos = 0, an empty snpInRef window, the next reference.  Speed is secondary here (the serial
 * chain of one stream on one wavefront); the output is byte-identical to the reference encoder's file, which
 * the reference's own `-x` reads.
 *
 * The same body also runs one stream PER segment (cbc_stream_args.per_segment): the rescale-capable fallback
 * for block descriptors the block kernels refuse (more than CBC_MAX_BLOCK_READS records).
 */
#ifndef CBC_STREAM_BODY_H
#define CBC_STREAM_BODY_H

#include "cbc_encode_body.h"
#include "cbc_decode_body.h"

/* LDS words of the stream kernels; then rname key/excess (cap_name each), pos value/count (cap_pos each) */
#define CBC_SLDS_RLEN    0u        /* 256: rlength[0] excess  */
#define CBC_SLDS_SNPS    256u
#define CBC_SLDS_INDELS  512u
#define CBC_SLDS_RING    768u      /* CBC_RING_WORDS: output bit ring (encode) / scratch read + lists (decode) */
#define CBC_SLDS_DEC_TMP  768u     /* decode: 80 words insertion-free read, 256 deletions, 256 insertions, 512 pos_alpha histograms */
#define CBC_SLDS_DEC_DELS (768u + 80u)
#define CBC_SLDS_DEC_INS  (768u + 336u)
#define CBC_SLDS_DEC_HIST (768u + 592u)
#define CBC_SLDS_PALPHA  (768u + 1104u)   /* 1024: the four pos_alpha byte models as dense excess tables */
#define CBC_SLDS_FIXED   (768u + 1104u + 1024u)

/* What the reference's tables hold and no LDS budget does lives in global memory ("aux", one area per stream slot):
 *   flag   65536 (value, excess) pairs beyond the register pairs           (sam_models.c:96-130: a 65536-entry table)
 *   pos    alphabet entries beyond the CBC_STREAM_POS_LDS held in LDS      (sam_block.h:54-55: MAX_ALPHA / MAX_CARDINALITY)
 * aux layout, in words: [flag value 65536][flag excess 65536][pos value pos_ov][pos count pos_ov] */
#define CBC_STREAM_POS_LDS 8192u          /* pos alphabet entries in LDS (a multiple of 64); the packer's cap_pos may exceed it */
#define CBC_STREAM_POS_MAX 5000000u       /* MAX_ALPHA: no POS step, hence no alphabet entry, beyond it */
struct cbc_stream_caps { uint32_t cap_pos, cap_name; };
#ifdef __HIPCC__
#define CBC_SHD __host__ __device__ static inline
#else
#define CBC_SHD static inline
#endif
CBC_SHD uint32_t cbc_stream_pos_lds(uint32_t cap_pos) { return cap_pos < CBC_STREAM_POS_LDS ? ((cap_pos + 63u) & ~63u) : CBC_STREAM_POS_LDS; }
CBC_SHD uint32_t cbc_stream_pos_ov(uint32_t cap_pos) { return cap_pos > CBC_STREAM_POS_LDS ? cap_pos - CBC_STREAM_POS_LDS : 0u; }
CBC_SHD uint64_t cbc_stream_aux_words(uint32_t cap_pos) { return 2ull * 65536ull + 2ull * cbc_stream_pos_ov(cap_pos) + 64ull; }
static inline uint32_t cbc_stream_lds_bytes(const cbc_stream_caps *c) { return 4u * (CBC_SLDS_FIXED + 2u * c->cap_name + 2u * cbc_stream_pos_lds(c->cap_pos)); }

struct cbc_stream_args {
    const cbc_read_rec   *recs;
    const uint8_t        *seq;
    const uint32_t       *tok;
    const uint8_t        *names;
    const cbc_block_desc *segs;
    const uint8_t        *ref;
    uint8_t              *out;
    cbc_block_result     *results;      /* one per stream */
    uint32_t             *vtab;         /* n_vtab tables of 65535 * 256 words, zero-filled by the caller */
    uint32_t             *aux;          /* n_vtab areas of cbc_stream_aux_words(cap_pos) words (no initial content needed) */
    uint64_t ref_bytes, out_bytes, seq_bytes, n_tok, n_recs;
    uint32_t n_segs, cap_pos, cap_name, names_bytes, per_segment, n_vtab;
};
#define CBC_VTAB_WORDS (65535ull * 256ull)

template <class W>
CBC_FN void cbc_encode_whole(const cbc_stream_args &A, uint32_t stream, uint32_t slot, uint32_t *lds)
{
    typedef typename W::V32 V32;
    typedef typename W::Mask Mask;
    typedef CbcEnc<W, true> Enc;
    const V32 ln = W::lane();
    Enc E;
    const uint32_t seg0 = A.per_segment ? stream : 0u, seg1 = A.per_segment ? stream + 1u : A.n_segs;
    const cbc_block_desc *first = A.segs + seg0;
    const uint64_t out_off = first->out_off;
    const uint32_t out_cap = first->out_cap, L0 = first->read_length;

    E.status = CBC_ST_OK; E.nsym = 0; E.fail_read = 0; E.cur_read = 0;
    E.l = W::uv(0u); E.set_l_forms(); E.rng = W::uv(CBC_M26 + 1u); E.scale3 = 0u; E.bitpos = 0; E.flushed = 0;
    E.ring = lds + CBC_SLDS_RING;
    E.q_lo = W::splat(0u); E.q_cnt = W::splat(0u); E.q_n = W::splat(0u); E.q_len = 0;
    E.rec_a = W::splat(0u); E.rec_s = W::splat(0u); E.rec_n = 0;
    E.role = CBC_ROLE_FUSED; E.batch_i = 0; E.batch = nullptr; E.ctl = nullptr;
    E.b_len = 0; E.b_pos = 0; E.b_stop = 64u; E.b_flags = 0; E.seen_last = 0; E.b_neq = 0;
    E.out32 = (uint32_t *)(A.out + out_off);
    E.cap_words = out_cap >> 2;
    bool args_ok = cbc_fits64(out_off, out_cap, A.out_bytes) && ((out_off & 3u) == 0u) && (L0 >= 1u && L0 <= 256u) &&
                   (slot < A.n_vtab) && (A.cap_pos >= 2u) && (A.cap_pos <= CBC_STREAM_POS_MAX) && (A.cap_name >= 4u) && (seg0 < A.n_segs) &&
                   (A.aux != nullptr);
    if (!args_ok) { E.cap_words = 0; E.fail(CBC_ST_ASSERT); }

    /* ---- model tables ---- */
    E.L0 = L0;
    E.rlen_exc = lds + CBC_SLDS_RLEN; E.snps_exc = lds + CBC_SLDS_SNPS; E.indels_exc = lds + CBC_SLDS_INDELS;
    E.rname_key = lds + CBC_SLDS_FIXED; E.rname_exc = E.rname_key + A.cap_name; E.rn_cap = A.cap_name; E.rn_count = 0;
    const uint32_t pos_lds = cbc_stream_pos_lds(A.cap_pos);
    E.pos_val = E.rname_exc + A.cap_name; E.pos_occ = E.pos_val + pos_lds; E.pos_pre = nullptr; E.cap_pos = A.cap_pos;
    {   /* the global-memory sides of flag and pos (this stream's aux area), the dense pos_alpha tables */
        uint32_t *aux = A.aux + (uint64_t)slot * cbc_stream_aux_words(A.cap_pos);
        E.fsp_key = aux; E.fsp_exc = aux + 65536u; E.fsp_count = 0;
        E.pos_ov_val = aux + 131072u; E.pos_ov_occ = E.pos_ov_val + cbc_stream_pos_ov(A.cap_pos); E.pos_lds_cap = pos_lds;
        E.palpha = lds + CBC_SLDS_PALPHA; E.pa_n0 = E.pa_n1 = E.pa_n2 = E.pa_n3 = 256u;
    }
    E.bloom = nullptr; E.var_ev = nullptr; E.nev = E.nev1 = 0; E.cap_var = 0;
    E.vtab = A.vtab + (uint64_t)slot * CBC_VTAB_WORDS;
    for (uint32_t b = 0; b < CBC_SLDS_RING + CBC_RING_WORDS; b += 64u) W::store32(lds, ln + b, W::splat(0u), W::all());
    for (uint32_t b = 0; b < 1024u; b += 64u) W::store32(lds + CBC_SLDS_PALPHA, ln + b, W::splat(0u), W::all());
    W::write_uni(E.pos_val, 0u, 0xffffffffu); W::write_uni(E.pos_occ, 0u, 1u);      /* the escape: counts[0] = 1 (sam_models.c:132-162) */
    E.p0cnt = W::splat(0u); E.p0over = 0; E.p0ev = nullptr;
    E.snps_n = L0; E.indels_n = L0;
    E.pos_card = 1u;
    E.fkey = W::splat(0u); E.fexc = W::splat(0u); E.fcount = 0;
    E.hkey = W::splat(0u); E.hexc = W::splat(0u);
    E.hc0 = E.hc1 = E.hc2 = E.hc3 = 0; E.hn0 = E.hn1 = E.hn2 = E.hn3 = 256u;
    {   /* lane table: match 1,1; same_ref 1,1; chars rows (sam_models.c:372-401) -- as in cbc_encode_stream */
        V32 sm = W::select(ln < 10u, W::splat(1u), W::splat(0u));
        V32 r = (ln - CBC_LT_CHARS) >> 3, c = (ln - CBC_LT_CHARS) & 7u;
        Mask inch = (ln >= CBC_LT_CHARS) & (r < 6u) & (c < 5u);
        V32 cv = W::select(c == 4u, W::splat(1u), W::select(c == r, W::splat(0u), W::splat(8u)));
        Mask bump = ((r == 0u) & ((c == 1u) | (c == 2u))) | ((r == 1u) & ((c == 0u) | (c == 3u))) |
                    ((r == 2u) & ((c == 0u) | (c == 3u))) | ((r == 3u) & ((c == 1u) | (c == 2u)));
        cv = W::select(bump, cv + 8u, cv);
        E.small = W::select(inch, cv, sm);
    }
    E.prevPos = 0; E.prevM = 0; E.prevChar = 0; E.win_pos = 0;
    E.win_clear();
    uint32_t rlen_n = 255u, rl123_c0 = 1u, rl123_n = 255u, flag_n = 65536u, pos_n = 1u;

    /* stream header: int(L0), 32 x int(WELL), int(LOSSLESS) */
    for (uint32_t k = 0; k < 34u && E.status == CBC_ST_OK; k++) {
        uint32_t v = (k == 0u) ? L0 : (k == 33u) ? 8u : CBC_WELL_SEED;
        E.regsparse_code(E.hkey, E.hexc, 0u, 8u, E.hc0, E.hn0, 256u, 1u, v >> 24, CBC_ST_ASSERT);
        E.regsparse_code(E.hkey, E.hexc, 8u, 8u, E.hc1, E.hn1, 256u, 1u, (v >> 16) & 0xffu, CBC_ST_ASSERT);
        E.regsparse_code(E.hkey, E.hexc, 16u, 8u, E.hc2, E.hn2, 256u, 1u, (v >> 8) & 0xffu, CBC_ST_ASSERT);
        E.regsparse_code(E.hkey, E.hexc, 24u, 8u, E.hc3, E.hn3, 256u, 1u, v & 0xffu, CBC_ST_ASSERT);
        if ((k & 7u) == 7u) E.drain_q();
    }
    E.drain_q();

    uint32_t prev_name = 0xffffffffu;
    uint64_t rec_index = 0;
    for (uint32_t sg = seg0; sg < seg1 && E.status == CBC_ST_OK; sg++) {
        const cbc_block_desc *bd = A.segs + sg;
        const uint64_t rec_base = bd->rec_base, seq_base = bd->seq_base, tok_base = bd->tok_base, ref_off = bd->ref_off;
        const uint32_t n_reads = bd->n_reads, name_off = bd->name_off, n_tok_blk = bd->n_tok;
        if (!(cbc_fits64(rec_base, n_reads, A.n_recs) && cbc_fits64(tok_base, n_tok_blk, A.n_tok) && name_off < A.names_bytes &&
              bd->read_length == L0)) { E.fail(CBC_ST_ASSERT); break; }
        const uint4 *recs4 = (const uint4 *)(A.recs + rec_base);
        const uint8_t *seqb = A.seq + seq_base;
        const uint32_t *tokb = A.tok + tok_base;
        const uint8_t *refb = A.ref + ref_off;
        const uint64_t seq_avail = cbc_le64(seq_base, A.seq_bytes) ? A.seq_bytes - seq_base : 0;
        const uint64_t ref_avail = cbc_le64(ref_off, A.ref_bytes) ? A.ref_bytes - ref_off : 0;
        const uint32_t seq_lim = cbc_avail32(seq_avail, 0u), ref_lim = cbc_avail32(ref_avail, 0u);
        const bool new_contig = name_off != prev_name;
        prev_name = name_off;

        for (uint32_t c0 = 0; c0 < n_reads && E.status == CBC_ST_OK; c0 += 64u) {
            V32 r_pos, r_fl, r_seq, r_tok;
            const uint32_t cn = n_reads - c0 < 64u ? n_reads - c0 : 64u;
            E.cur_read = (uint32_t)rec_index;
            W::load_rec(recs4, ln + c0, (ln + c0) < n_reads, r_pos, r_fl, r_seq, r_tok);
            {   /* validated one lane each so that the per-record loads need no clamping */
                V32 vrl = r_fl >> 16;
                Mask live = (ln + c0) < n_reads;
                Mask bad = live & ((vrl == 0u) | (vrl > CBC_MAX_READ_LEN) | (r_pos == 0u) | (r_seq > seq_lim) | ((seq_lim - r_seq) < (vrl + 4u)) |
                                   (r_pos > ref_lim) | ((ref_lim - r_pos) < (vrl + 3u)) | (r_tok >= n_tok_blk));
                uint64_t bb = W::ballot(bad);
                if (bb) { E.cur_read = (uint32_t)rec_index + W::ctz64(bb); E.fail(CBC_ST_ASSERT); break; }
            }
            /* match test of the group (read_compression.c:291-296): lane l compares bases 4l..4l+3, 8 records in flight */
            uint64_t neq = 0;
            for (uint32_t j0 = 0; j0 < cn; j0 += 8u) {
                V32 sv[8], rv[8]; uint32_t rls[8];
                const V32 bo = ln * 4u;
                for (uint32_t q = 0; q < 8u; q++) {
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
            for (uint32_t j = 0; j < cn && E.status == CBC_ST_OK; j++) {
                const uint32_t r = c0 + j;
                E.cur_read = (uint32_t)(rec_index + j);
                if (E.q_len >= 32u) E.drain_q();
                /* -- compress_rname (id_compression.c:39-65) -- */
                const bool chr_change = new_contig && r == 0u;
                if (chr_change) {
                    E.small_code(CBC_LT_SAMEREF, 2u, 10u, 1u);
                    for (uint32_t q = 0; E.status == CBC_ST_OK; q++) {
                        uint32_t ch = (name_off + q < A.names_bytes) ? W::read_uni8(A.names, name_off + q) : 0u;
                        E.rname_code(E.prevChar, ch);
                        if ((q & 31u) == 31u) E.drain_q();
                        if (ch == 0u) break;
                        E.prevChar = ch;
                    }
                    E.drain_q();
                    E.prevPos = 0; E.win_clear(); E.win_pos = 0;      /* compress_pos :123-124; compression.c:62-63 */
                } else E.small_code(CBC_LT_SAMEREF, 2u, 10u, 0u);
                const uint32_t pos = W::readlane(r_pos, j), flw = W::readlane(r_fl, j), rl = flw >> 16;
                /* -- read length, 4 "bytes" (read_compression.c:29-33, quirk Q1): the low byte, then three zeros -- */
                E.dense_code(E.rlen_exc, 255u, 10u, rl & 0xffu, rlen_n);
                for (int k = 1; k < 4; k++) {
                    E.encode(0u, rl123_c0, rl123_n);
                    if (k == 3) {                                    /* the three contexts evolve in lock step */
                        rl123_c0 += 10u; rl123_n += 10u;
                        if (rl123_n >= CBC_RESCALE) { rl123_c0 = (rl123_c0 >> 1) + 1u; rl123_n = 254u + rl123_c0; }
                    }
                }
                /* -- compress_pos -- */
                if (pos < E.prevPos) { E.fail(CBC_ST_ASSERT); break; }                  /* unsorted: x <= 0 aborts there */
                const uint32_t x = pos - E.prevPos + 1u;
                if (x >= 5000000u) { E.fail(CBC_ST_ASSERT); break; }                     /* MAX_ALPHA, sam_block.h:54 */
                E.pos_lit_code(x, pos_n);
                E.prevPos = pos;
                /* -- compress_flag, compress_match -- */
                E.flag_gen_code(flw & 0xffffu, flag_n);
                const uint32_t imperfect = (uint32_t)((neq >> j) & 1ull);
                E.small_code(CBC_LT_MATCH + (((x == 1u) ? 2u : 0u) | E.prevM) * 2u, 2u, 1u, imperfect ^ 1u);
                E.prevM = imperfect ^ 1u;
                if (imperfect && E.status == CBC_ST_OK) {
                    if (E.q_len >= 32u) E.drain_q();
                    const uint32_t so = W::readlane(r_seq, j), to = W::readlane(r_tok, j);
                    const V32 bo = ln * 4u;
                    const V32 seqv = W::load32_bytes(seqb + so, bo, bo < rl);
                    const V32 tokv = W::load32(tokb + to, ln, (ln + to) < n_tok_blk, 0u);
                    E.edits(pos, flw, to, seqv, tokv, tokb, n_tok_blk);
                }
            }
            rec_index += cn;
        }
    }
    /* ---- end-of-stream sentinel (compression.c:152): same_ref 1, '\n', NUL; flush ---- */
    uint32_t nbytes = 0;
    if (E.status == CBC_ST_OK) {
        E.cur_read = (uint32_t)rec_index;
        if (E.q_len >= 32u) E.drain_q();
        E.small_code(CBC_LT_SAMEREF, 2u, 10u, 1u);
        E.rname_code(E.prevChar, (uint32_t)'\n');
        E.rname_code((uint32_t)'\n', 0u);
        E.drain_q();
    }
    if (E.status == CBC_ST_OK) { E.flush_recs(); nbytes = E.finish(); }
    if (E.status != CBC_ST_OK) nbytes = 0;
    V32 resv = W::select(ln == 0u, W::splat(nbytes), W::select(ln == 1u, W::splat(E.status),
               W::select(ln == 2u, W::splat(E.nsym), W::splat(E.fail_read))));
    W::store32((uint32_t *)(A.results + stream), ln, resv, ln < 4u);
}

/* ---------------------------------------------------------------------------------------------------------------
 * decode direction: decompress() src/compression.c:173-216, decompress_line :71-108 -- the exact inverse of
 * cbc_encode_whole (same model forms), one wavefront, records until the end-of-stream sentinel.
 * ------------------------------------------------------------------------------------------------------------- */
struct cbc_dstream_args {
    const uint8_t  *in;
    const uint8_t  *ref;
    const uint64_t *contig_off, *contig_len;   /* FASTA order: the stream only ever says "next contig" */
    cbc_read_rec   *recs;
    uint8_t        *seq;
    cbc_block_result *results;
    uint32_t       *vtab;
    uint32_t       *aux;                       /* cbc_stream_aux_words(cap_pos) words */
    uint64_t in_bytes, ref_bytes, rec_cap, seq_bytes;
    uint32_t n_contigs, cap_pos, cap_name, seq_stride, read_length;
};

template <class W>
CBC_FN void cbc_decode_whole(const cbc_dstream_args &A, uint32_t *lds)
{
    typedef typename W::V32 V32;
    typedef typename W::Mask Mask;
    const V32 ln = W::lane();
    CbcDec<W, true> D;
    const uint32_t L0 = A.read_length, stride = A.seq_stride;

    D.status = CBC_ST_OK; D.nsym = 0; D.fail_read = 0; D.cur_read = 0;
    D.l = W::dv(0u); D.rng = W::dv(CBC_M26 + 1u); D.d = W::dv(0u); D.acc = 0; D.navail = 0; D.widx = 0; D.wordv = W::splat(0u);
    D.inb = A.in;
    D.lds = lds; D.cap_pos = A.cap_pos; D.cap_var = 0; D.L0 = L0; D.evp = nullptr;
    D.rname_key = lds + CBC_SLDS_FIXED; D.rname_exc = D.rname_key + A.cap_name; D.rn_cap = A.cap_name;
    const uint32_t pos_lds = cbc_stream_pos_lds(A.cap_pos);
    D.pos_valp = D.rname_exc + A.cap_name; D.pos_cntp = D.pos_valp + pos_lds; D.histp = lds + CBC_SLDS_DEC_HIST;
    D.fsp_key = A.aux; D.fsp_exc = A.aux + 65536u; D.fsp_count = 0;
    D.pos_ov_valp = A.aux + 131072u; D.pos_ov_cntp = D.pos_ov_valp + cbc_stream_pos_ov(A.cap_pos); D.pos_lds_cap = pos_lds;
    D.palpha = lds + CBC_SLDS_PALPHA; D.pa_n0 = D.pa_n1 = D.pa_n2 = D.pa_n3 = 256u;
    D.vtab = A.vtab;
    bool args_ok = (L0 >= 1u && L0 <= 256u) && (stride >= 4u && stride <= 256u && (stride & 3u) == 0u) && A.n_contigs >= 1u &&
                   A.cap_pos >= 2u && A.cap_pos <= CBC_STREAM_POS_MAX && A.aux != nullptr && A.cap_name >= 4u && cbc_le64(A.in_bytes, 0x3fffffff0ull) &&
                   cbc_le64(A.rec_cap, 0xffffffffull) && cbc_le64(A.rec_cap * (uint64_t)stride + 8u, A.seq_bytes);
    D.nwords_in = (uint32_t)((A.in_bytes + 3u) >> 2);
    D.tail_valid = (uint32_t)A.in_bytes & 3u;
    if (!args_ok) { D.nwords_in = 0; D.fail(CBC_ST_ASSERT); }
    for (uint32_t b = 0; b < CBC_SLDS_FIXED; b += 64u) W::store32(lds, ln + b, W::splat(0u), (ln + b) < CBC_SLDS_FIXED);
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

    if (D.status == CBC_ST_OK) D.d = W::dv(D.take(26u));                 /* the tag (alloc_arithmetic_stream :260-263) */
    for (uint32_t k = 0; k < 34u && D.status == CBC_ST_OK; k++) {  /* header: int(L0), 32 x int(WELL), int(8) */
        uint32_t v = D.regsparse_dec(D.hkey, D.hexc, 0u, 8u, D.hc0, D.hn0, 256u, 1u, CBC_ST_ASSERT) << 24;
        v |= D.regsparse_dec(D.hkey, D.hexc, 8u, 8u, D.hc1, D.hn1, 256u, 1u, CBC_ST_ASSERT) << 16;
        v |= D.regsparse_dec(D.hkey, D.hexc, 16u, 8u, D.hc2, D.hn2, 256u, 1u, CBC_ST_ASSERT) << 8;
        v |= D.regsparse_dec(D.hkey, D.hexc, 24u, 8u, D.hc3, D.hn3, 256u, 1u, CBC_ST_ASSERT);
        if (D.status != CBC_ST_OK) break;
        if (k == 0u && v != L0) D.fail(CBC_ST_ASSERT);
        if (k == 33u && v != 8u) D.fail(CBC_ST_UNSUPPORTED);       /* LOSSY streams are out of scope */
    }

    uint4 *recs4 = (uint4 *)A.recs;
    uint8_t *tmpb = (uint8_t *)(lds + CBC_SLDS_DEC_TMP);
    uint32_t *dels = lds + CBC_SLDS_DEC_DELS, *insl = lds + CBC_SLDS_DEC_INS;
    const uint8_t *refb = A.ref; uint32_t ref_lim = 0, contig = 0xffffffffu;
    V32 refw = W::splat(0u); uint8_t *pend_dst = A.seq; uint32_t pend_rl = 0;
    uint32_t r = 0;
    bool done = false;
    while (D.status == CBC_ST_OK && !done) {
        D.cur_read = r;
        if (pend_rl) { W::store32_bytes(pend_dst, ln * 4u, refw, (ln * 4u) < pend_rl); pend_rl = 0; }
        /* -- decompress_rname (id_compression.c:67-94) -- */
        uint32_t sr = D.small_dec(CBC_LT_SAMEREF, 2u, 10u);
        if (D.status != CBC_ST_OK) break;
        if (sr) {
            for (uint32_t q = 0; D.status == CBC_ST_OK; q++) {
                uint32_t ch = D.rname_dec(D.prevChar);
                if (D.status != CBC_ST_OK || ch == 0u) break;
                if (ch == (uint32_t)'\n') {                        /* the end-of-stream sentinel (compression.c:152) */
                    if (q != 0u || D.rname_dec((uint32_t)'\n') != 0u) D.fail(CBC_ST_ASSERT);
                    done = true; break;
                }
                if (q >= CBC_CAP_NAME) { D.fail(CBC_ST_CAP_NAME); break; }
                D.prevChar = ch;
            }
            if (done || D.status != CBC_ST_OK) break;
            contig++;                                              /* the next FASTA record (read_decompression.c:28-42) */
            if (contig >= A.n_contigs) { D.fail(CBC_ST_ASSERT); break; }
            const uint64_t co = W::read_uni((const uint32_t *)(A.contig_off + contig), 0u) | ((uint64_t)W::read_uni((const uint32_t *)(A.contig_off + contig), 1u) << 32);
            const uint64_t cl = W::read_uni((const uint32_t *)(A.contig_len + contig), 0u) | ((uint64_t)W::read_uni((const uint32_t *)(A.contig_len + contig), 1u) << 32);
            if (!cbc_fits64(co, cl + CBC_REF_PAD, A.ref_bytes)) { D.fail(CBC_ST_ASSERT); break; }
            refb = A.ref + co;
            ref_lim = cbc_avail32(cl + CBC_REF_PAD, 0u);
            D.prevPos = 0; D.win_clear();
        } else if (contig == 0xffffffffu) { D.fail(CBC_ST_ASSERT); break; }
        if (!cbc_le64((uint64_t)r + 1u, A.rec_cap)) { D.fail(CBC_ST_OUT_FULL); break; }

        /* -- read length (read_decompression.c:68-74, quirk Q1): the low byte, then three symbols that can only be 0 -- */
        uint32_t rl = D.rlen_dec();
        for (int k = 1; k < 4 && D.status == CBC_ST_OK; k++) {
            D.nsym++;
            uint32_t q0, qn;
            float inv = W::lane_float(W::recip_v(W::splat(D.rl123_n)), 0u);
            W::muldiv2(D.rng, D.rl123_c0, D.rl123_n, D.rl123_n, inv, q0, qn);
            if (q0 == 0u || W::dv_ge(D.d, q0)) { D.fail(CBC_ST_ASSERT); break; }   /* another symbol was coded here */
            D.rng = W::dv(q0);
            D.renorm();
        }
        D.rl123_c0 += 10u; D.rl123_n += 10u;
        if (D.rl123_n >= CBC_RESCALE) { D.rl123_c0 = (D.rl123_c0 >> 1) + 1u; D.rl123_n = 254u + D.rl123_c0; }
        if (D.status != CBC_ST_OK) break;
        if (rl == 0u || rl > CBC_MAX_READ_LEN || rl > stride) { D.fail(CBC_ST_ASSERT); break; }

        /* -- pos, flag -- */
        uint32_t x = D.pos_dec();
        if (D.status != CBC_ST_OK) break;
        if (x < 1u || x >= 5000000u) { D.fail(CBC_ST_ASSERT); break; }
        uint32_t pos = D.prevPos + x - 1u;
        if (pos < D.prevPos) { D.fail(CBC_ST_ASSERT); break; }
        D.win_shift(x - 1u > 256u ? 256u : x - 1u);
        D.prevPos = pos;
        uint32_t flag = D.flag_gen_dec();
        if (D.status != CBC_ST_OK) break;
        const uint32_t strand = (flag >> 4) & 1u;
        if (pos == 0u || pos > ref_lim || ref_lim - pos < rl + 3u + 256u) { D.fail(CBC_ST_ASSERT); break; }
        refw = W::load32_bytes(refb + (pos - 1u), ln * 4u, (ln * 4u) < rl);

        uint32_t match = D.small_dec(CBC_LT_MATCH + (((x == 1u) ? 2u : 0u) | D.prevM) * 2u, 2u, 1u);
        if (D.status != CBC_ST_OK) break;
        D.prevM = match;
        uint8_t *dst = A.seq + (uint64_t)r * stride;
        if (match) { pend_dst = dst; pend_rl = rl; }
        else if (!D.edits_dec(pos, rl, strand, refw, dst, refb, tmpb, lds + CBC_SLDS_DEC_TMP, dels, insl)) break;
        V32 rv0 = W::splat(pos), rv1 = W::splat(flag | (rl << 16)), rv2 = W::splat(r * stride), rv3 = W::splat(contig);
        W::store_rec(recs4, W::splat(r), ln == 0u, rv0, rv1, rv2, rv3);
        r++;
    }
    if (pend_rl) W::store32_bytes(pend_dst, ln * 4u, refw, (ln * 4u) < pend_rl);
    V32 resv = W::select(ln == 0u, W::splat(r), W::select(ln == 1u, W::splat(D.status),
               W::select(ln == 2u, W::splat(D.nsym), W::splat(D.fail_read))));
    W::store32((uint32_t *)A.results, ln, resv, ln < 4u);
}

#endif /* CBC_STREAM_BODY_H */
