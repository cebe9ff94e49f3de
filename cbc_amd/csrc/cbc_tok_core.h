/*
 * cbc_tok_core.h -- the SAM record tokeniser as plain functions over byte ranges, so that the same code runs one GPU
 * thread per line (cbc_tokenise.hip) and, compiled by gcc, under the CPU tests (tests/emu) next to the host packer.
 *
 * It restates, for one line / one record, exactly what the host packer (cbc_pack.c) restates of the reference:
 *   load_sam_line()         src/sam_file_allocation.c:437-529   strtok("\t") columns of a line that still carries its
 *                                                                '\n' (quirk Q2), FLAG / POS via atoi, MD|XD aux field
 *   CIGAR scan              src/read_compression.c:308-352       atoi() of the UNCONSUMED segment at every M I D S *
 *   MD scan                 src/read_compression.c:613-701       (gap << 8) | letter per mismatch, '^' runs add to the gap
 *   consistency replay      src/read_compression.c:551-552, 656-659  every MD mismatch must be consumed by the read
 * A record without an MD / XD field inherits the text of the nearest earlier line that has one (read_line_t.edits persists
 * across load_sam_line calls: src/sam_file_allocation.c:505-511) -- here: a look-back over the lines already split, at most
 * CBC_TOK_MD_LOOKBACK of them (cbc_tok_md_source()).
 * What it does NOT do, and reports as CBC_TOK_NEEDS_HOST so that the caller falls back to the host packer: a record
 * whose CIGAR starts with a soft clip -- quirk Q6: the reference rewrites that record's MD text IN the buffer that persists
 * across records and does not terminate what it wrote (read_compression.c:357-468, :460), so the text depends on the
 * buffer's whole history, which is serial state -- and an MD-less record with no MD within the look-back.
 */
#ifndef CBC_TOK_CORE_H
#define CBC_TOK_CORE_H

#include <stdint.h>
#include "../../include/cbc_gpu.h"

#if defined(__HIPCC__)
#define CBC_TOK_FN __host__ __device__ static inline
#else
#define CBC_TOK_FN static inline
#endif

#define CBC_TOK_MD_LOOKBACK 4096u         /* lines an MD-less record looks back for its text (beyond: the host packer) */
#define CBC_TOK_LINE_MAX 1023u            /* fgets(buffer, 1024, ...): longer lines are refused, as by the host packer */

enum { CBC_TOK_OK = 0, CBC_TOK_SKIP = 1 /* no token on the line */, CBC_TOK_UNMAPPED = 2,
       CBC_TOK_NEEDS_HOST = 3, CBC_TOK_E_COLUMNS = 4, CBC_TOK_E_LINE_LONG = 5, CBC_TOK_E_CIGAR = 6, CBC_TOK_E_STAR = 7,
       CBC_TOK_E_MD = 8, CBC_TOK_E_INCONSISTENT = 9, CBC_TOK_E_READ_LEN = 10, CBC_TOK_E_POS = 11, CBC_TOK_E_TOO_MANY = 12 };

/* one line after the column split: offsets into the text */
typedef struct cbc_tok_line {
    uint64_t rname, cigar, seq, md;       /* byte offsets of the tokens in the SAM text */
    uint32_t rname_len, cigar_len, seq_len, md_len;   /* md_len includes a trailing '\n' when MD is the last column (Q2) */
    int32_t  pos; uint32_t flag;
    uint32_t status, has_md;
} cbc_tok_line;

/* what the host needs of a mapped record to cut blocks (cbc_pack.c place_record) */
typedef cbc_tok_record_summary cbc_tok_summary;         /* the one public 16-byte struct (include/cbc_gpu.h): nt | ev << 16, line index */

CBC_TOK_FN int cbc_tok_isdigit(uint8_t c) { return c >= '0' && c <= '9'; }
CBC_TOK_FN int cbc_tok_isspace(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }
/* atoi() over [p, e): leading white space, one sign, digits; saturates instead of overflowing */
CBC_TOK_FN long long cbc_tok_atoi(const uint8_t *p, const uint8_t *e)
{
    while (p < e && cbc_tok_isspace(*p)) p++;
    int neg = 0;
    if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
    long long v = 0;
    while (p < e && cbc_tok_isdigit(*p)) { if (v < (1ll << 40)) v = v * 10 + (*p - '0'); p++; }
    return neg ? -v : v;
}
CBC_TOK_FN uint32_t cbc_tok_num_digits(uint32_t x)      /* compute_num_digits read_compression.c:720-743 */
{
    return x < 10 ? 1 : x < 100 ? 2 : x < 1000 ? 3 : x < 10000 ? 4 : x < 100000 ? 5 : x < 1000000 ? 6 : x < 10000000 ? 7 : x < 100000000 ? 8 : 9;
}

/* strtok(line, "\t") over [b, e) (e = one past the line's '\n', or the end of the text): the 11 compulsory columns and
 * the MD / XD aux field (the LAST one seen wins; at most 20 other aux fields are looked at: MAX_AUX_FIELDS) */
CBC_TOK_FN void cbc_tok_split(const uint8_t *sam, uint64_t b, uint64_t e, cbc_tok_line *L)
{
    L->status = CBC_TOK_OK; L->has_md = 0; L->md = 0; L->md_len = 0;
    L->rname = L->cigar = L->seq = 0; L->rname_len = L->cigar_len = L->seq_len = 0; L->pos = 0; L->flag = 0;
    if (e - b > CBC_TOK_LINE_MAX) { L->status = CBC_TOK_E_LINE_LONG; return; }
    uint64_t p = b; uint32_t nf = 0, aux = 0;
    for (;;) {
        while (p < e && sam[p] == '\t') p++;
        if (p >= e) break;
        uint64_t q = p;
        while (q < e && sam[q] != '\t') q++;
        if (nf < 11) {
            if (nf == 1) L->flag = (uint32_t)(uint16_t)cbc_tok_atoi(sam + p, sam + q);
            else if (nf == 2) { L->rname = p; L->rname_len = (uint32_t)(q - p); }
            else if (nf == 3) L->pos = (int32_t)cbc_tok_atoi(sam + p, sam + q);
            else if (nf == 5) { L->cigar = p; L->cigar_len = (uint32_t)(q - p); }
            else if (nf == 9) { L->seq = p; L->seq_len = (uint32_t)(q - p); }
            nf++;
        } else {
            if (q - p >= 2 && (sam[p] == 'M' || sam[p] == 'X') && sam[p + 1] == 'D') {
                L->has_md = 1;
                if (q - p >= 5) { L->md = p + 5; L->md_len = (uint32_t)(q - p - 5); } else { L->md = p; L->md_len = 0; }
            } else if (++aux == 20) break;
        }
        p = q;
    }
    if (nf == 0) { L->status = CBC_TOK_SKIP; return; }
    if (nf < 11) { L->status = CBC_TOK_E_COLUMNS; return; }
    if ((L->flag & 4u) == 4u) { L->status = CBC_TOK_UNMAPPED; return; }
    /* no MD / XD field: the caller gives the record the text it inherits (cbc_tok_md_source) before cbc_tok_record() */
}
/* Where an MD-less record at line k takes its MD text from: get(j) returns line j's split (NULL for a line that is not part
 * of the body: '@' headers).  An unmapped line's MD counts too -- load_sam_line() has copied it before compress_line() looks
 * at the FLAG.  *md / *md_len: the text (empty when no earlier body line has one: the buffer starts zeroed). */
template <class GET>
CBC_TOK_FN uint32_t cbc_tok_md_source(uint64_t k, GET get, uint64_t *md, uint32_t *md_len)
{
    *md = 0; *md_len = 0;
    for (uint64_t back = 1; back <= k; back++) {
        const cbc_tok_line *P = get(k - back);
        if (!P) return CBC_TOK_OK;                           /* reached the header: nothing to inherit */
        if (P->has_md) { *md = P->md; *md_len = P->md_len; return CBC_TOK_OK; }
        if (back >= CBC_TOK_MD_LOOKBACK) return CBC_TOK_NEEDS_HOST;
    }
    return CBC_TOK_OK;
}

/* CIGAR + MD of one mapped record -> token words.  `tk` may be NULL (count only).  Returns a CBC_TOK_* status;
 * *nt_out = words, *ev_out = the record's bound on var symbols. */
CBC_TOK_FN uint32_t cbc_tok_record(const uint8_t *sam, const cbc_tok_line *L, uint32_t *tk, uint32_t *nt_out, uint32_t *ev_out)
{
    const uint8_t *cig = sam + L->cigar, *ce = cig + L->cigar_len;
    const uint8_t *md = sam + L->md, *me = md + L->md_len;
    const uint32_t rl = L->seq_len;
    if (rl == 0 || rl > CBC_MAX_READ_LEN) return CBC_TOK_E_READ_LEN;
    if (L->pos < 1) return CBC_TOK_E_POS;
    uint32_t nt = 2, n_cig = 0, n_md = 0, ev = 0;
    /* ---- CIGAR (read_compression.c:308-549 scanning rule): a segment runs from the end of the previous recognised op */
    {
        const uint8_t *seg = cig;
        for (const uint8_t *c = cig; c < ce; c++) {
            const uint8_t ch = *c;
            if (cbc_tok_isdigit(ch)) continue;
            uint32_t op = ch == 'M' ? CBC_OP_M : ch == 'I' ? CBC_OP_I : ch == 'D' ? CBC_OP_D : ch == 'S' ? CBC_OP_S : ch == '*' ? CBC_OP_STAR : 0xffu;
            if (op == 0xffu) continue;                      /* not an op: it stays inside the segment */
            long long v = cbc_tok_atoi(seg, c + 1);          /* atoi() of the unconsumed segment */
            if (v < 0 || v > 0x0fffffff) return CBC_TOK_E_CIGAR;
            if (nt >= 2048u) return CBC_TOK_E_TOO_MANY;
            if (op == CBC_OP_STAR) return CBC_TOK_E_STAR;
            if (op == CBC_OP_S && n_cig == 0) return CBC_TOK_NEEDS_HOST;      /* leading soft clip: quirk Q6 */
            if (tk) tk[nt] = ((uint32_t)v << 4) | op;
            nt++; n_cig++;
            if (op != CBC_OP_M) ev += (uint32_t)v;
            seg = c + 1;
        }
    }
    /* ---- MD (add_snps_to_array scanning rule) ---- */
    const uint32_t md0 = nt;
    {
        const uint8_t *p = md;
        while (p < me) {
            uint32_t gap = (uint32_t)cbc_tok_atoi(p, me);
            p += cbc_tok_num_digits(gap);
            if (p > me) p = me;
            uint8_t ch = p < me ? *p : 0; if (ch) p++;
            int ended = 0;
            while (ch == '^') {
                while (p < me && !cbc_tok_isdigit(*p)) p++;
                if (p >= me) { ended = 1; break; }
                uint32_t v = (uint32_t)cbc_tok_atoi(p, me);
                gap += v; p += cbc_tok_num_digits(v);
                if (p > me) p = me;
                ch = p < me ? *p : 0; if (ch) p++;
            }
            if (ended || ch == 0) break;
            if (gap > 0x00ffffffu) return CBC_TOK_E_MD;
            if (nt >= 4095u) return CBC_TOK_E_TOO_MANY;
            if (tk) tk[nt] = (gap << 8) | ch;
            nt++; n_md++;
            if (p >= me) break;
        }
    }
    if (n_cig > 0xffffu || n_md > 0xffffu) return CBC_TOK_E_TOO_MANY;
    /* ---- consistency replay: every MD mismatch must be consumed (compress_edits' walk with the early-return rule).
     * The MD gaps are re-read from the text when tk is NULL: redo the MD scan inline with a second cursor. ---- */
    uint32_t nDel = 0, nIns = 0, nS = 0;
    {
        const uint8_t *p = md; uint32_t k = 0, cum = 0, Mc = 0, ins = 0; int more = 1;
        uint32_t g_next = 0; int have_next = 0;
        /* next_gap(): the gap of mismatch token k, scanning the text once, in order */
#define CBC_TOK_NEXT_GAP() do { if (!have_next && k < n_md) {                                                           \
            uint32_t gap_ = (uint32_t)cbc_tok_atoi(p, me); p += cbc_tok_num_digits(gap_); if (p > me) p = me;           \
            uint8_t ch_ = p < me ? *p : 0; if (ch_) p++;                                                                \
            while (ch_ == '^') { while (p < me && !cbc_tok_isdigit(*p)) p++; if (p >= me) break;                          \
                uint32_t v_ = (uint32_t)cbc_tok_atoi(p, me); gap_ += v_; p += cbc_tok_num_digits(v_); if (p > me) p = me; \
                ch_ = p < me ? *p : 0; if (ch_) p++; }                                                                  \
            g_next = gap_; have_next = 1; } } while (0)
        const uint8_t *seg = cig; uint32_t o = 0;
        for (const uint8_t *c = cig; ; c++) {
            uint32_t op, len;
            if (c < ce) {
                const uint8_t ch = *c;
                if (cbc_tok_isdigit(ch)) continue;
                op = ch == 'M' ? CBC_OP_M : ch == 'I' ? CBC_OP_I : ch == 'D' ? CBC_OP_D : ch == 'S' ? CBC_OP_S : 0xffu;
                if (op == 0xffu) continue;
                len = (uint32_t)cbc_tok_atoi(seg, c + 1); seg = c + 1; o++;
            } else { op = 99u; len = 1u; }
            if (op == CBC_OP_M) { Mc += len; continue; }
            if (op == CBC_OP_D) { nDel += len; continue; }
            for (uint32_t i = 0; i < len; i++) {
                if ((op == CBC_OP_I || op == 99u) && more) {
                    const uint32_t limit = op == 99u ? rl + 1u : Mc + ins;
                    more = 0;
                    while (k < n_md) {
                        CBC_TOK_NEXT_GAP();
                        if (cum + g_next >= limit) { cum++; more = 1; break; }
                        cum += g_next + 1u; nS++; k++; have_next = 0;
                    }
                }
                if (op == 99u) break;
                nIns++; ins++;
            }
            if (op == 99u) break;
        }
#undef CBC_TOK_NEXT_GAP
        (void)o;
    }
    if (nS != n_md) return CBC_TOK_E_INCONSISTENT;
    if (nDel > 0xffffu || nIns > 0xffffu) return CBC_TOK_E_TOO_MANY;
    if (tk) { tk[0] = n_cig | (n_md << 16); tk[1] = nDel | (nIns << 16); }
    (void)md0;
    *nt_out = nt; *ev_out = ev + n_md;
    return CBC_TOK_OK;
}

#endif /* CBC_TOK_CORE_H */
