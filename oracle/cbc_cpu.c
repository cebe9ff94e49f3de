/*
 * cbc_cpu.c -- the `cbc_cpu_*` entry points SURVEY.md section 8(b) asks for: the SAME packed inputs and
 * the SAME signatures as the HIP library's host-buffer entry points (include/cbc_gpu.h:
 * cbc_gpu_init / cbc_gpu_upload_reference / cbc_gpu_encode_blocks / cbc_gpu_decode_blocks), coded on one
 * CPU core with the reference's dense model tables (cbc_oracle.c, which this file includes).
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: it is the block-by-block checker for inputs too
 * large to be written out as SAM text (cfg2 / cfg3 at full size) and the "parse excluded" leg of bench.py's
 * cpu_baseline.  Nothing under cbc_amd/ may include, link, load or call it.  Parity status: the same as
 * cbc_oracle.c ("parity unpinned", see its header); tests/test_oracle.py pins this file to cbc_oracle.c's text
 * path (same bytes for every block of every fixture), nothing pins either to the upstream binary.
 *
 * What is restated here on top of cbc_oracle.c: compress_edits() (src/read_compression.c:265-606) reading the
 * packer's CIGAR / MD TOKENS instead of text -- the walk of SURVEY.md section 8a "Edit-list construction,
 * restated" -- and the block framing (POS rebased by the packer, one contig per block, the block's reference
 * window instead of a FASTA file).
 */
#include "cbc_oracle.c"
#include "../include/cbc_gpu.h"

typedef struct cbc_cpu_ctx { const uint8_t *ref; uint64_t ref_bytes; char err[256]; } cbc_cpu_ctx;

ORACLE_API int cbc_cpu_init(int ordinal, cbc_cpu_ctx **out)
{
    (void)ordinal;
    if (!out) return CBC_E_ARG;
    *out = (cbc_cpu_ctx *)calloc(1, sizeof(cbc_cpu_ctx));
    return *out ? CBC_OK : CBC_E_NOMEM;
}
ORACLE_API int cbc_cpu_shutdown(cbc_cpu_ctx *c) { free(c); return CBC_OK; }
ORACLE_API const char *cbc_cpu_last_error(cbc_cpu_ctx *c) { return c ? c->err : "no context"; }
/* the caller's buffer is used in place (it must outlive the calls) */
ORACLE_API int cbc_cpu_upload_reference(cbc_cpu_ctx *c, const uint8_t *bases, uint64_t nbytes)
{
    if (!c || !bases) return CBC_E_ARG;
    c->ref = bases; c->ref_bytes = nbytes;
    return CBC_OK;
}

/* the edits of one imperfect record from its tokens (cbc_gpu.h "Token stream of one record") */
typedef struct { const uint32_t *md; uint32_t n_md, k, cum; const uint8_t *read; uint32_t rl; snp_t *snps; unsigned n_snps; } md_cursor;

static int md_pull(md_cursor *c, uint32_t limit)            /* add_snps_to_array read_compression.c:613-701 */
{
    while (c->k < c->n_md) {
        uint32_t g = c->md[c->k] >> 8;
        if (c->cum + g >= limit) { c->cum++; return 1; }    /* :656-659: stop before this mismatch */
        c->cum += g;
        if (c->n_snps >= MAX_READ_LEN) return -1;
        c->snps[c->n_snps].pos = g;
        c->snps[c->n_snps].refChar = char2basepair((char)(c->md[c->k] & 0xffu));
        c->snps[c->n_snps].targetChar = char2basepair(c->cum < c->rl ? (char)c->read[c->cum] : 0);
        c->n_snps++; c->cum++; c->k++;
    }
    return 0;
}

static void cpu_edits(enc_t *E, const uint32_t *t, const uint8_t *read, uint32_t rl, uint8_t strand)
{
    static __thread uint32_t Dels[MAX_READ_LEN]; static __thread ins_t Insers[MAX_READ_LEN];
    static __thread snp_t SNPs[MAX_READ_LEN];
    const uint32_t n_cig = t[0] & 0xffffu, n_md = t[0] >> 16;
    const uint32_t *cig = t + 2;
    md_cursor mc = { t + 2 + n_cig, n_md, 0, 0, read, rl, SNPs, 0 };
    unsigned nIns = 0, nDel = 0; uint32_t Mc = 0, prevI = 0, prevD = 0; int more = 1;
    for (uint32_t o = 0; o < n_cig; o++) {
        uint32_t op = cig[o] & 15u, len = cig[o] >> 4;
        if (op == CBC_OP_M) { Mc += len; continue; }
        if (op == CBC_OP_D) {
            for (uint32_t c = 0; c < len; c++) {
                if (nDel >= MAX_READ_LEN) { E->ac.err = ERR_ASSERT; return; }
                Dels[nDel++] = Mc - prevD; prevD = Mc;
            }
            continue;
        }
        if (op != CBC_OP_I && !(op == CBC_OP_S && o != 0)) { E->ac.err = ERR_ASSERT; return; }   /* '*', raw leading S */
        for (uint32_t c = 0; c < len; c++) {
            if (op == CBC_OP_I && more) { more = md_pull(&mc, Mc + nIns); if (more < 0) { E->ac.err = ERR_ASSERT; return; } }
            if (nIns >= MAX_READ_LEN) { E->ac.err = ERR_ASSERT; return; }
            Insers[nIns].pos = Mc - prevI;
            Insers[nIns].targetChar = char2basepair(Mc + nIns < rl ? (char)read[Mc + nIns] : 0);
            prevI = Mc; nIns++;
        }
    }
    if (more) { if (md_pull(&mc, rl + 1) < 0) { E->ac.err = ERR_ASSERT; return; } }
    const unsigned nSnp = mc.n_snps;
    if ((nDel | nIns) == 0) send_upd(&E->ac, &E->M.snps, (uint8_t)nSnp);                  /* :557-565 */
    else {
        send_upd(&E->ac, &E->M.snps, 0);
        send_upd(&E->ac, &E->M.indels, (uint8_t)nSnp);
        send_upd(&E->ac, &E->M.indels, (uint8_t)nDel);
        send_upd(&E->ac, &E->M.indels, (uint8_t)nIns);
    }
    uint32_t p = 0;
    for (unsigned k = 0; k < nDel && !E->ac.err; k++) { compress_var(E, Dels[k], p, strand); p += Dels[k]; }
    p = 0;
    for (unsigned k = 0; k < nSnp && !E->ac.err; k++) {                                     /* :573-593 */
        uint32_t d = delta_to_first_snp(E->ctg.snpInRef, E->cumsumP, p, rl) << BITS_DELTA;
        compress_var(E, SNPs[k].pos, d + p, strand);
        p += SNPs[k].pos + 1;
        E->ctg.snpInRef[E->cumsumP + p - 2] = 1;
        compress_chars(E, SNPs[k].refChar, SNPs[k].targetChar);
    }
    p = 0;
    for (unsigned k = 0; k < nIns && !E->ac.err; k++) {
        compress_var(E, Insers[k].pos, p, strand); p += Insers[k].pos;
        compress_chars(E, BP_O, Insers[k].targetChar);
    }
}

/* one block = one stream: what the reference writes when it is run on the block alone */
static int64_t cpu_encode_block(const cbc_cpu_ctx *C, const cbc_host_batch *hb, const cbc_block_desc *bd,
                                uint8_t *out, size_t cap, cbc_block_result *res)
{
    enc_t *E = (enc_t *)calloc(1, sizeof(enc_t));
    if (!E) return ERR_NOMEM;
    int64_t ret; uint32_t r = 0, cur = 0;
    ac_init(&E->ac);
    memset(out, 0, cap);
    E->ac.io.buf = out; E->ac.io.cap = cap;
    const uint32_t L0 = bd->read_length;
    const cbc_read_rec *recs = hb->recs + bd->rec_base;
    const uint8_t *seq = hb->seq + bd->seq_base; const uint32_t *tok = hb->tok + bd->tok_base;
    uint32_t last_pos = bd->n_reads ? recs[bd->n_reads - 1].pos : 1;
    if (bd->ref_off > C->ref_bytes) { free(E); return ERR_INPUT; }
    E->ctg.ref = (char *)(C->ref + bd->ref_off);                       /* the block's window; not owned */
    E->ctg.len = (uint32_t)((C->ref_bytes - bd->ref_off) > 0xfffffff0ull ? 0xfffffff0u : (C->ref_bytes - bd->ref_off));
    E->ctg.snpInRef = (uint8_t *)calloc((size_t)last_pos + 2 * MAX_READ_LEN + 16, 1);
    int rc = models_init(&E->M, L0);
    if (rc || !E->ctg.snpInRef) { ret = ERR_NOMEM; goto done; }
    compress_int(E, L0);
    for (int i = 0; i < 32; i++) compress_int(E, WELL_DEBUG);
    compress_int(E, LOSSLESS_CODE);
    for (r = 0; r < bd->n_reads && !E->ac.err; r++) {
        const cbc_read_rec *rr = &recs[r];
        cur = r;
        int chr_change = compress_rname(E, (const char *)hb->names + bd->name_off);
        if (chr_change) E->cumsumP = 0;
        const uint32_t rl = rr->rlen;
        E->read_length = rl;
        for (int k = 0; k < 4; k++) {                                      /* quirk Q1 */
            uint32_t mask = 0xFFu << (k * 8);
            send_upd(&E->ac, &E->M.rlength[k], (uint8_t)((uint8_t)(rl & mask) >> (k * 8)));
        }
        uint32_t dP = compress_pos(E, rr->pos, chr_change);
        if (E->ac.err) break;
        uint8_t strand = (uint8_t)compress_flag(E, rr->flag);
        if (E->ac.err) break;
        if ((uint64_t)rr->pos + rl > (uint64_t)E->ctg.len || rr->pos == 0) { E->ac.err = ERR_INPUT; break; }
        E->cumsumP = E->cumsumP + dP - 1;                                  /* :281 */
        const uint8_t *read = seq + rr->seq_off;
        if (memcmp(read, E->ctg.ref + rr->pos - 1, rl) == 0) { compress_match(E, 1, dP); continue; }
        compress_match(E, 0, dP);
        cpu_edits(E, tok + rr->tok_off, read, rl, strand);
        if (!E->ac.err && E->cumsumP != rr->pos) E->ac.err = ERR_ASSERT;   /* assert :41 */
    }
    if (!E->ac.err) { compress_rname(E, "\n"); ac_finish(&E->ac); }
    if (E->ac.err) ret = E->ac.err;
    else if (E->ac.io.overflow) ret = ERR_OUTCAP;
    else ret = (int64_t)E->ac.io.pos;
done:
    if (res) {
        res->nbytes = ret > 0 ? (uint32_t)ret : 0; res->n_symbols = (uint32_t)E->ac.nsym; res->fail_read = ret < 0 ? cur : 0;
        res->status = ret >= 0 ? CBC_ST_OK : ret == ERR_OUTCAP ? CBC_ST_OUT_FULL : CBC_ST_ASSERT;
    }
    models_free(&E->M); free(E->ctg.snpInRef); free(E);
    return ret;
}

/* cbc_gpu_encode_blocks' contract on the CPU: block b's payload is out[out_offsets[b] .. out_offsets[b+1]) */
ORACLE_API int cbc_cpu_encode_blocks(cbc_cpu_ctx *C, const cbc_host_batch *hb, uint8_t *out, uint64_t out_cap,
                                     uint64_t *out_offsets, cbc_block_result *results)
{
    if (!C || !hb || !out || !out_offsets || !C->ref) return CBC_E_ARG;
    uint64_t off = 0; int rc = CBC_OK;
    out_offsets[0] = 0;
    for (uint32_t b = 0; b < hb->n_blocks; b++) {
        cbc_block_result r; memset(&r, 0, sizeof r);
        const cbc_block_desc *bd = &hb->blocks[b];
        /* worst case of a block: 3 bytes per coded symbol (every total < 2^20), cf. cbc_plan_output() */
        uint64_t need = 4096 + 48ull * bd->n_reads + 8ull * bd->n_tok;
        if (off + need > out_cap) need = out_cap - off;
        int64_t n = cpu_encode_block(C, hb, bd, out + off, (size_t)need, &r);
        if (n < 0) { if (rc == CBC_OK) { rc = CBC_E_BLOCK; snprintf(C->err, sizeof C->err, "block %u failed (%lld)", b, (long long)n); } n = 0; }
        if (results) results[b] = r;
        off += (uint64_t)n;
        out_offsets[b + 1] = off;
    }
    return rc;
}

/* the long-read format extension (stream version 3): its CPU statement lives in its own file */
#include "cbc_long.c"
