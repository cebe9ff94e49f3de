/*
 * cbc_long.c -- CPU statement of the LONG-READ FORMAT EXTENSION (stream version 3).
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  NO REFERENCE PARITY EXISTS OR CAN EXIST for this
 * format: the reference cannot code reads longer than 252 bases at all (var contexts < 65535, sam_models.c:317;
 * MAX_READ_LENGTH 1024, sam_block.h:38; MAX_ALPHA 5e6, :54).  This file is the independent CPU statement of the
 * extension's specification, written from the specification and not from the HIP kernel, so
 * that "GPU == this file, byte for byte" and "decode(encode(x)) == x" are the two checks the format has.
 *
 * What is kept from the reference: the 26-bit range coder and bit I/O (Arithmetic_stream.c), the adaptive model
 * (stream_model.c: add the step, halve-and-increment at 2^20), compress_int, the contig-name model, the pos model
 * with its escape alphabet, the FLAG model, the chars model (6 x 5 with the reference's initial counts).
 * What is new (one stream per BLOCK of reads, as in block mode; `x` = value, all via send + update):
 *   stream  := int(0x43424C03 "CBL\3") int(8)  record*  same_ref(1) '\n' NUL  flush
 *   record  := same_ref(0) | same_ref(1) name NUL
 *              len[0](b3) len[1](b2) len[2](b1) len[3](b0)        read length, 4 real bytes MSB first, 4 x 256, step 10
 *              pos_sym [esc: b3 b2 b1 b0]                         x = POS - prevPOS + 1, any x < 2^31 (no MAX_ALPHA)
 *              flag                                               1 x 65536, step 8
 *              ne[0](hi) ne[1](lo)                                number of edits (u16), 2 x 256, step 10
 *              { gap kind [base] } x ne
 *   gap     := gapm[2 * prev_kind + strand]( min(g, 255) ) [ g >= 255: gx[0]((g-255) >> 8) gx[1]((g-255) & 255) ]
 *              g = matched bases since the previous edit; prev_kind: 0 sub, 1 ins, 2 del, 3 = first edit; 8 x 256, step 10
 *   kind    := kindm[prev_kind](k)                                k: 0 sub, 1 ins (and soft clip), 2 del; 4 x 3, init 1, step 8
 *   base    := chars[row](basepair(read base))                    sub: row = basepair(reference base); ins: row 5
 * Edits are in read order: walk the CIGAR; inside an M run every read base that differs from the reference base is
 * a sub; I / S bases are ins; D bases are del.  MD is not used.
 */
#include "../include/cbc_gpu.h"

#define LONG_MAGIC 0x43424C03u

typedef struct {
    models_t M;                         /* codebook, same_ref, rname, pos (+alpha), flag, chars are used */
    model_t len[4], ne[2], gapm[8], gx[2], kindm[4];
} long_models;

static int long_models_init(long_models *L)
{
    int rc = models_init(&L->M, 1);
    for (int i = 0; i < 4; i++) rc |= model_alloc(&L->len[i], 256, 1, 10);
    for (int i = 0; i < 2; i++) rc |= model_alloc(&L->ne[i], 256, 1, 10);
    for (int i = 0; i < 8; i++) rc |= model_alloc(&L->gapm[i], 256, 1, 10);
    for (int i = 0; i < 2; i++) rc |= model_alloc(&L->gx[i], 256, 1, 10);
    for (int i = 0; i < 4; i++) rc |= model_alloc(&L->kindm[i], 3, 1, 8);
    return rc;
}
static void long_models_free(long_models *L)
{
    models_free(&L->M);
    for (int i = 0; i < 4; i++) { free(L->len[i].counts); free(L->kindm[i].counts); }
    for (int i = 0; i < 2; i++) { free(L->ne[i].counts); free(L->gx[i].counts); }
    for (int i = 0; i < 8; i++) free(L->gapm[i].counts);
}

/* the pos model without the reference's 5e6 bound: the escape carries any 31-bit value */
static int long_alpha_reserve(models_t *M, uint32_t x)
{
    if (x >= 0x80000000u) return ERR_ASSERT;
    if (x < M->alpha_cap) return 0;
    uint64_t nc = M->alpha_cap; while (nc <= x) nc <<= 1;
    int32_t *am = (int32_t *)realloc(M->alphaMap, sizeof(int32_t) * nc);
    if (!am) return ERR_NOMEM;
    M->alphaMap = am;
    uint8_t *ae = (uint8_t *)realloc(M->alphaExist, nc);
    if (!ae) return ERR_NOMEM;
    M->alphaExist = ae;
    memset(am + M->alpha_cap, 0, sizeof(int32_t) * (nc - M->alpha_cap));
    memset(ae + M->alpha_cap, 0, nc - M->alpha_cap);
    M->alpha_cap = (uint32_t)nc;
    return 0;
}

typedef struct { uint32_t gap; uint8_t kind, base_row, base; } long_edit;

/* edits of one record from its CIGAR tokens, the read and the reference window (ref points at POS) */
static int long_edits(const uint32_t *t, const uint8_t *read, uint32_t rl, const uint8_t *ref, uint64_t ref_avail,
                      long_edit *E, uint32_t cap, uint32_t *n_out)
{
    const uint32_t n_cig = t[0] & 0xffffu;
    uint32_t i = 0, g = 0, n = 0; uint64_t j = 0;
    for (uint32_t o = 0; o < n_cig; o++) {
        uint32_t op = t[2 + o] & 15u, len = t[2 + o] >> 4;
        for (uint32_t c = 0; c < len; c++) {
            if (op == CBC_OP_M) {
                if (i >= rl || j >= ref_avail) return ERR_INPUT;
                if (read[i] == ref[j]) g++;
                else {
                    if (n >= cap) return ERR_INPUT;
                    E[n].gap = g; E[n].kind = 0; E[n].base_row = (uint8_t)char2basepair((char)ref[j]); E[n].base = (uint8_t)char2basepair((char)read[i]); n++; g = 0;
                }
                i++; j++;
            } else if (op == CBC_OP_I || op == CBC_OP_S) {
                if (i >= rl || n >= cap) return ERR_INPUT;
                E[n].gap = g; E[n].kind = 1; E[n].base_row = BP_O; E[n].base = (uint8_t)char2basepair((char)read[i]); n++; g = 0;
                i++;
            } else if (op == CBC_OP_D) {
                if (j >= ref_avail || n >= cap) return ERR_INPUT;
                E[n].gap = g; E[n].kind = 2; E[n].base_row = 0; E[n].base = 0; n++; g = 0;
                j++;
            } else return ERR_INPUT;
        }
    }
    if (i != rl) return ERR_INPUT;                   /* the CIGAR must consume the read exactly */
    *n_out = n;
    return 0;
}

static int64_t long_encode_block(const cbc_cpu_ctx *C, const cbc_host_batch *hb, const cbc_block_desc *bd,
                                 uint8_t *out, size_t cap, cbc_block_result *res)
{
    enc_t *E = (enc_t *)calloc(1, sizeof(enc_t));
    long_models *L = (long_models *)calloc(1, sizeof(long_models));
    long_edit *ed = (long_edit *)malloc(sizeof(long_edit) * 65536);
    if (!E || !L || !ed) { free(E); free(L); free(ed); return ERR_NOMEM; }
    int64_t ret; uint32_t cur = 0;
    ac_init(&E->ac);
    memset(out, 0, cap);
    E->ac.io.buf = out; E->ac.io.cap = cap;
    const cbc_read_rec *recs = hb->recs + bd->rec_base;
    const uint8_t *seq = hb->seq + bd->seq_base; const uint32_t *tok = hb->tok + bd->tok_base;
    if (bd->ref_off > C->ref_bytes || long_models_init(L)) { ret = ERR_NOMEM; goto done; }
    E->M = L->M;                                         /* compress_int / compress_rname / compress_flag work on E->M */
    const uint8_t *ref = C->ref + bd->ref_off; const uint64_t ref_avail = C->ref_bytes - bd->ref_off;
    compress_int(E, LONG_MAGIC);
    compress_int(E, LOSSLESS_CODE);
    uint32_t prevPos = 0;
    for (uint32_t r = 0; r < bd->n_reads && !E->ac.err; r++) {
        const cbc_read_rec *rr = &recs[r];
        cur = r;
        compress_rname(E, (const char *)hb->names + bd->name_off);
        const uint32_t rl = rr->rlen;
        for (int k = 0; k < 4; k++) send_upd(&E->ac, &L->len[k], (int32_t)((rl >> (8 * (3 - k))) & 0xffu));
        {   /* pos: the reference's model, any 31-bit step */
            models_t *M = &E->M;
            if (rr->pos < prevPos) { E->ac.err = ERR_ASSERT; break; }
            uint32_t x = rr->pos - prevPos + 1;
            if (long_alpha_reserve(M, x)) { E->ac.err = ERR_ASSERT; break; }
            if (M->alphaExist[x]) send_upd(&E->ac, &M->pos, M->alphaMap[x]);
            else {
                send_upd(&E->ac, &M->pos, 0);
                send_upd(&E->ac, &M->pos_alpha[0], (int32_t)(x >> 24));
                send_upd(&E->ac, &M->pos_alpha[1], (int32_t)((x >> 16) & 0xffu));
                send_upd(&E->ac, &M->pos_alpha[2], (int32_t)((x >> 8) & 0xffu));
                send_upd(&E->ac, &M->pos_alpha[3], (int32_t)(x & 0xffu));
                if (pos_reserve(M)) { E->ac.err = ERR_NOMEM; break; }
                M->alphaExist[x] = 1; M->alphaMap[x] = (int32_t)M->pos.card; M->pos.alphabet[M->pos.card] = (int32_t)x;
                uint32_t idx = M->pos.card++;
                model_update(&M->pos, idx);
            }
            prevPos = rr->pos;
        }
        uint32_t strand = compress_flag(E, rr->flag);
        if (E->ac.err) break;
        if (rr->pos == 0 || (uint64_t)rr->pos - 1 > ref_avail) { E->ac.err = ERR_INPUT; break; }
        uint32_t ne = 0;
        int rc = long_edits(tok + rr->tok_off, seq + rr->seq_off, rl, ref + (rr->pos - 1), ref_avail - (rr->pos - 1), ed, 65535, &ne);
        if (rc) { E->ac.err = rc; break; }
        send_upd(&E->ac, &L->ne[0], (int32_t)(ne >> 8));
        send_upd(&E->ac, &L->ne[1], (int32_t)(ne & 0xffu));
        uint32_t pk = 3;
        for (uint32_t k = 0; k < ne && !E->ac.err; k++) {
            const uint32_t g = ed[k].gap;
            send_upd(&E->ac, &L->gapm[2 * pk + strand], (int32_t)(g < 255 ? g : 255));
            if (g >= 255) {
                if (g - 255 > 0xffffu) { E->ac.err = ERR_ASSERT; break; }
                send_upd(&E->ac, &L->gx[0], (int32_t)((g - 255) >> 8));
                send_upd(&E->ac, &L->gx[1], (int32_t)((g - 255) & 0xffu));
            }
            send_upd(&E->ac, &L->kindm[pk], ed[k].kind);
            if (ed[k].kind != 2) send_upd(&E->ac, &E->M.chars[ed[k].base_row], ed[k].base);
            pk = ed[k].kind;
        }
    }
    if (!E->ac.err) { compress_rname(E, "\n"); ac_finish(&E->ac); }
    if (E->ac.err) ret = E->ac.err;
    else if (E->ac.io.overflow) ret = ERR_OUTCAP;
    else ret = (int64_t)E->ac.io.pos;
    L->M = E->M;                                        /* the tables may have been reallocated */
done:
    if (res) {
        res->nbytes = ret > 0 ? (uint32_t)ret : 0; res->n_symbols = (uint32_t)E->ac.nsym; res->fail_read = ret < 0 ? cur : 0;
        res->status = ret >= 0 ? CBC_ST_OK : ret == ERR_OUTCAP ? CBC_ST_OUT_FULL : CBC_ST_ASSERT;
    }
    long_models_free(L); free(L); free(E); free(ed);
    return ret;
}

/* same contract as cbc_cpu_encode_blocks, long-read format */
ORACLE_API int cbc_cpu_long_encode_blocks(cbc_cpu_ctx *C, const cbc_host_batch *hb, uint8_t *out, uint64_t out_cap,
                                          uint64_t *out_offsets, cbc_block_result *results)
{
    if (!C || !hb || !out || !out_offsets || !C->ref) return CBC_E_ARG;
    uint64_t off = 0; int rc = CBC_OK;
    out_offsets[0] = 0;
    for (uint32_t b = 0; b < hb->n_blocks; b++) {
        cbc_block_result r; memset(&r, 0, sizeof r);
        const cbc_block_desc *bd = &hb->blocks[b];
        uint64_t bases = 0;
        for (uint32_t k = 0; k < bd->n_reads; k++) bases += hb->recs[bd->rec_base + k].rlen;
        uint64_t need = 4096 + 64ull * bd->n_reads + 9ull * bases;      /* <= 3 symbols of < 20 bits per base, + slack */
        if (off + need > out_cap) need = out_cap - off;
        int64_t n = long_encode_block(C, hb, bd, out + off, (size_t)need, &r);
        if (n < 0) { if (rc == CBC_OK) { rc = CBC_E_BLOCK; snprintf(C->err, sizeof C->err, "block %u failed (%lld)", b, (long long)n); } n = 0; }
        if (results) results[b] = r;
        off += (uint64_t)n;
        out_offsets[b + 1] = off;
    }
    return rc;
}

/* decode one long-format block: recs (pos, flag, rlen, seq_off) and bases; returns records decoded or < 0 */
ORACLE_API int64_t cbc_cpu_long_decode_block(cbc_cpu_ctx *C, const uint8_t *in, uint64_t in_bytes, uint64_t ref_off,
                                             cbc_read_rec *recs, uint64_t rec_cap, uint8_t *seq, uint64_t seq_cap)
{
    if (!C || !C->ref || ref_off > C->ref_bytes) return CBC_E_ARG;
    dec_t *D = (dec_t *)calloc(1, sizeof(dec_t));
    long_models *L = (long_models *)calloc(1, sizeof(long_models));
    if (!D || !L) { free(D); free(L); return ERR_NOMEM; }
    int64_t ret = 0; uint64_t nr = 0, so = 0;
    ac_init(&D->ac);
    D->ac.io.in = in; D->ac.io.in_len = in_bytes;
    for (unsigned b = 0; b < AWORD; b++) D->ac.t = (D->ac.t << 1) | get_bit(&D->ac.io);
    if (long_models_init(L)) { ret = ERR_NOMEM; goto done; }
    D->M = L->M;
    const uint8_t *ref = C->ref + ref_off; const uint64_t ref_avail = C->ref_bytes - ref_off;
    if (decompress_int(D) != LONG_MAGIC || decompress_int(D) != LOSSLESS_CODE) { ret = ERR_INPUT; goto done; }
    uint32_t prevPos = 0;
    while (!D->ac.err) {
        int chr = decompress_rname(D);
        if (D->ac.err || chr == -1) break;
        uint32_t rl = 0;
        for (int k = 0; k < 4; k++) rl = (rl << 8) | (uint32_t)read_upd(&D->ac, &L->len[k]);
        models_t *M = &D->M;
        int am = read_upd(&D->ac, &M->pos);
        if (D->ac.err) break;
        int32_t x = M->pos.alphabet[am];
        if (x == -1) {
            uint32_t ux = (uint32_t)read_upd(&D->ac, &M->pos_alpha[0]) << 24;
            ux |= (uint32_t)read_upd(&D->ac, &M->pos_alpha[1]) << 16;
            ux |= (uint32_t)read_upd(&D->ac, &M->pos_alpha[2]) << 8;
            ux |= (uint32_t)read_upd(&D->ac, &M->pos_alpha[3]);
            x = (int32_t)ux;
            if (D->ac.err || long_alpha_reserve(M, ux) || pos_reserve(M)) { D->ac.err = ERR_ASSERT; break; }
            M->alphaExist[x] = 1; M->alphaMap[x] = (int32_t)M->pos.card; M->pos.alphabet[M->pos.card] = x;
            uint32_t idx = M->pos.card++;
            model_update(&M->pos, idx);
        }
        uint32_t pos = prevPos + (uint32_t)x - 1; prevPos = pos;
        uint32_t flag = (uint32_t)read_upd(&D->ac, &M->flag);
        uint32_t strand = (flag >> 4) & 1u;
        uint32_t ne = (uint32_t)read_upd(&D->ac, &L->ne[0]) << 8; ne |= (uint32_t)read_upd(&D->ac, &L->ne[1]);
        if (D->ac.err) break;
        if (nr >= rec_cap || so + rl + 8 > seq_cap || pos == 0 || rl > 65535) { ret = ERR_OUTCAP; goto done; }
        uint32_t i = 0, pk = 3; uint64_t j = pos - 1;
        uint8_t *dst = seq + so;
        for (uint32_t k = 0; k < ne && !D->ac.err; k++) {
            uint32_t g = (uint32_t)read_upd(&D->ac, &L->gapm[2 * pk + strand]);
            if (g == 255) { uint32_t hi = (uint32_t)read_upd(&D->ac, &L->gx[0]); g = 255 + ((hi << 8) | (uint32_t)read_upd(&D->ac, &L->gx[1])); }
            uint32_t kind = (uint32_t)read_upd(&D->ac, &L->kindm[pk]);
            if (D->ac.err) break;
            if ((uint64_t)i + g > rl || j + g > ref_avail) { D->ac.err = ERR_ASSERT; break; }
            memcpy(dst + i, ref + j, g); i += g; j += g;
            if (kind == 0) { if (i >= rl || j >= ref_avail) { D->ac.err = ERR_ASSERT; break; }
                             dst[i++] = (uint8_t)basepair2char(read_upd(&D->ac, &D->M.chars[char2basepair((char)ref[j])])); j++; }
            else if (kind == 1) { if (i >= rl) { D->ac.err = ERR_ASSERT; break; } dst[i++] = (uint8_t)basepair2char(read_upd(&D->ac, &D->M.chars[BP_O])); }
            else j++;
            pk = kind;
        }
        if (D->ac.err) break;
        if (j + (rl - i) > ref_avail) { D->ac.err = ERR_ASSERT; break; }
        memcpy(dst + i, ref + j, rl - i);
        recs[nr].pos = pos; recs[nr].flag = (uint16_t)flag; recs[nr].rlen = (uint16_t)rl; recs[nr].seq_off = (uint32_t)so; recs[nr].tok_off = 0;
        nr++; so += rl;
    }
    ret = D->ac.err ? D->ac.err : (int64_t)nr;
    L->M = D->M;
done:
    long_models_free(L); free(L); free(D);
    return ret;
}
