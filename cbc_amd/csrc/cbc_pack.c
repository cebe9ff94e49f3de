/*
 * cbc_pack.c -- host packer: SAM text + FASTA text -> packed record blocks for the HIP encoder.
 *
 * Restates, for the product, the parsing rules of the reference's record loader and FASTA loader
 * (they decide which bytes the models see, so they are load-bearing for bit-exactness):
 *   load_sam_line()              src/sam_file_allocation.c:437-529
 *   get_read_length()            src/sam_file_allocation.c:26-79
 *   store_reference_in_memory()  src/read_decompression.c:17-53
 *   CIGAR scan of compress_edits()      src/read_compression.c:308-352,469-484
 *   MD scan of add_snps_to_array()      src/read_compression.c:613-701
 * and cuts the record stream into independent blocks (hard part 1): one
 * contig per block, POS rebased so that the block's first record has POS 1, at most
 * opts.block_reads records, and never more distinct POS deltas / var symbols / FLAG values than
 * the kernel's LDS tables hold.
 *
 * No arithmetic coding happens here.
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>
#include "../../include/cbc_host.h"

#define LINE_BUF 1024              /* fgets(buffer, 1024, ...) in the reference */
#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
static int grow(void **p, uint64_t *cap, uint64_t need, size_t elt)
{
    if (need <= *cap) return 0;
    uint64_t nc = *cap ? *cap : 1024;
    while (nc < need) nc += nc / 2 + 1024;
    void *q = realloc(*p, (size_t)(nc * elt));
    if (!q) return CBC_E_NOMEM;
    *p = q; *cap = nc;
    return 0;
}
static int grow32(void **p, uint32_t *cap, uint64_t need, size_t elt)
{
    uint64_t c = *cap; int rc = grow(p, &c, need, elt);
    *cap = (uint32_t)c; return rc;
}
static uint32_t num_digits(uint32_t x)             /* compute_num_digits read_compression.c:720-743 */
{
    if (x < 10) return 1; if (x < 100) return 2; if (x < 1000) return 3; if (x < 10000) return 4;
    if (x < 100000) return 5; if (x < 1000000) return 6; if (x < 10000000) return 7;
    if (x < 100000000) return 8; return 9;
}

typedef struct {
    cbc_packed *P;
    cbc_pack_opts o;
    char *err; size_t errlen;
    /* current contig / block */
    char prev_name[LINE_BUF];
    int have_contig;
    uint32_t contig;               /* index of current contig */
    uint32_t n_fasta;              /* contigs available in ref */
    int blk_open;
    uint64_t blk_first_pos;        /* absolute POS of the block's first record */
    uint32_t blk_prev_pos;         /* absolute POS of the previous record in the block */
    uint32_t blk_reads, blk_var, blk_nflags, blk_ndelta;
    int seg_continues;             /* whole-file mode: the segment being opened continues the previous one's contig */
    uint64_t blk_bases;
    uint16_t blk_flags[CBC_CAP_FLAG];
    uint8_t *wf_seen;                                  /* whole-file mode: one bit per POS step below MAX_ALPHA seen so far */
    uint32_t *dset; uint32_t *dstamp; uint32_t dmask, depoch;   /* distinct-delta hash set */
    char edits[2 * LINE_BUF];      /* persists across records like read_line_t.edits */
    uint32_t tokbuf[4 * LINE_BUF];
} packer_t;

static int fail(packer_t *S, int code, const char *fmt, const char *a, long long b)
{
    if (S->err && S->errlen) snprintf(S->err, S->errlen, fmt, a ? a : "", b);
    return code;
}

/* FASTA: every record, in file order (the reference consumes the NEXT record at each RNAME change
 * and ignores the FASTA header text, read_decompression.c:28-42). */
static int load_fasta(packer_t *S, const char *fa, size_t len)
{
    cbc_packed *P = S->P;
    size_t off = 0;
    int first_line = 1, in_contig = 0;
    uint64_t start = 0;
    S->n_fasta = 0;
    while (off < len) {
        size_t e; { const char *nl = (const char *)memchr(fa + off, '\n', len - off); e = nl ? (size_t)(nl - fa) : len; }
        size_t ll = e - off;
        if (ll >= LINE_BUF - 1) return fail(S, CBC_E_INPUT, "FASTA line longer than %s1022 bytes at offset %lld (reference loader limit)", "", (long long)off);
        int is_hdr = (ll > 0 && fa[off] == '>') || (first_line && off == 0);
        if (is_hdr) {
            if (in_contig) {                            /* a '>' line ends the record being read */
                if (grow((void **)&P->ref, &P->cap_ref, P->ref_bytes + CBC_REF_PAD, 1)) return CBC_E_NOMEM;
                memset(P->ref + P->ref_bytes, 0, CBC_REF_PAD);
                if (grow32((void **)&P->contigs, &P->cap_contigs, (uint64_t)S->n_fasta + 1, sizeof(cbc_contig_info))) return CBC_E_NOMEM;
                P->contigs[S->n_fasta].ref_off = start; P->contigs[S->n_fasta].length = P->ref_bytes - start;
                P->contigs[S->n_fasta].name_off = 0; P->contigs[S->n_fasta].reserved = 0;
                P->ref_bytes += CBC_REF_PAD; S->n_fasta++;
            }
            in_contig = 1; start = P->ref_bytes;
        } else {
            if (!in_contig) { in_contig = 1; start = P->ref_bytes; }
            if (grow((void **)&P->ref, &P->cap_ref, P->ref_bytes + ll, 1)) return CBC_E_NOMEM;
            {   /* toupper() in the C locale, written so that it vectorises */
                uint8_t *d = P->ref + P->ref_bytes; const uint8_t *q = (const uint8_t *)fa + off;
                for (size_t i = 0; i < ll; i++) { uint8_t ch = q[i]; d[i] = (uint8_t)(ch - (((uint8_t)(ch - 'a') < 26u) << 5)); }
            }
            P->ref_bytes += ll;
        }
        first_line = 0;
        off = e + 1;
    }
    if (in_contig) {
        if (grow((void **)&P->ref, &P->cap_ref, P->ref_bytes + CBC_REF_PAD, 1)) return CBC_E_NOMEM;
        memset(P->ref + P->ref_bytes, 0, CBC_REF_PAD);
        if (grow32((void **)&P->contigs, &P->cap_contigs, (uint64_t)S->n_fasta + 1, sizeof(cbc_contig_info))) return CBC_E_NOMEM;
        P->contigs[S->n_fasta].ref_off = start; P->contigs[S->n_fasta].length = P->ref_bytes - start;
        P->contigs[S->n_fasta].name_off = 0; P->contigs[S->n_fasta].reserved = 0;
        P->ref_bytes += CBC_REF_PAD; S->n_fasta++;
    }
    return 0;
}

/* ---- the same loader on several threads: the text is cut at line starts; pass 1 counts,
 * per chunk, the header lines and the sequence bytes before / after each of them; a serial prefix turns the
 * counts into contig records and output offsets; pass 2 copies and upper-cases.  Byte-identical to
 * load_fasta() (tests).  Used for texts of 8 MiB and more. */
typedef struct {
    const char *fa; size_t beg, end, len;
    int first_chunk;
    /* pass 1 */
    uint64_t n_hdr, seq_before_first_hdr, seq_total; int too_long; size_t bad_off;
    uint64_t *seg; uint64_t cap_seg;             /* sequence bytes after each header line of the chunk */
    /* pass 2 */
    uint8_t *out; uint64_t out_pos;              /* where this chunk's first sequence byte goes */
    int phase;
} fa_chunk_t;

static inline int fa_is_hdr(const fa_chunk_t *c, size_t off, size_t ll) { return (ll > 0 && c->fa[off] == '>') || off == 0; }

static void *fa_run(void *arg)
{
    fa_chunk_t *c = (fa_chunk_t *)arg;
    const char *fa = c->fa;
    size_t off = c->beg;
    uint64_t cur = 0; int seen_hdr = 0;
    uint8_t *d = c->out ? c->out + c->out_pos : NULL;
    while (off < c->end) {
        const char *nl = (const char *)memchr(fa + off, '\n', c->len - off);
        size_t e = nl ? (size_t)(nl - fa) : c->len;
        size_t ll = e - off;
        if (c->phase == 1 && ll >= LINE_BUF - 1 && !c->too_long) { c->too_long = 1; c->bad_off = off; }
        if (fa_is_hdr(c, off, ll)) {
            if (c->phase == 1) {
                if (!seen_hdr) c->seq_before_first_hdr = cur; else c->seg[c->n_hdr - 1] = cur;
                if (grow((void **)&c->seg, &c->cap_seg, c->n_hdr + 1, sizeof(uint64_t))) { c->too_long = 2; return NULL; }
                c->n_hdr++; seen_hdr = 1; cur = 0;
            } else if (off != 0) {                 /* a header line closes the contig before it: its pad */
                memset(d, 0, CBC_REF_PAD); d += CBC_REF_PAD;
            }
        } else if (c->phase == 1) { cur += ll; c->seq_total += ll; }
        else {
            const uint8_t *q = (const uint8_t *)fa + off;
            for (size_t i = 0; i < ll; i++) { uint8_t ch = q[i]; d[i] = (uint8_t)(ch - (((uint8_t)(ch - 'a') < 26u) << 5)); }
            d += ll;
        }
        off = e + 1;
    }
    if (c->phase == 1) { if (!seen_hdr) c->seq_before_first_hdr = cur; else c->seg[c->n_hdr - 1] = cur; }
    return NULL;
}

static int load_fasta_mt(packer_t *S, const char *fa, size_t len, int nthreads)
{
    cbc_packed *P = S->P;
    fa_chunk_t *ch = (fa_chunk_t *)calloc((size_t)nthreads, sizeof(fa_chunk_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    int rc = 0;
    if (!ch || !th) { free(ch); free(th); return CBC_E_NOMEM; }
    size_t b = 0;
    for (int t = 0; t < nthreads; t++) {
        size_t e = (t == nthreads - 1) ? len : len / (size_t)nthreads * (size_t)(t + 1);
        if (e < b) e = b;
        if (e < len && t != nthreads - 1) { const char *nl = (const char *)memchr(fa + e, '\n', len - e); e = nl ? (size_t)(nl - fa) + 1 : len; }
        ch[t].fa = fa; ch[t].beg = b; ch[t].end = e; ch[t].len = len; ch[t].phase = 1; b = e;
    }
    for (int pass = 1; pass <= 2 && !rc; pass++) {
        int started = 0;
        for (int t = 0; t < nthreads; t++) { ch[t].phase = pass; if (pthread_create(&th[t], NULL, fa_run, &ch[t]) != 0) break; started++; }
        for (int t = started; t < nthreads; t++) fa_run(&ch[t]);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
        if (pass == 2) break;
        /* serial prefix: contig records and output offsets */
        for (int t = 0; t < nthreads && !rc; t++) {
            if (ch[t].too_long == 2) rc = CBC_E_NOMEM;
            else if (ch[t].too_long) rc = fail(S, CBC_E_INPUT, "FASTA line longer than %s1022 bytes at offset %lld (reference loader limit)", "", (long long)ch[t].bad_off);
        }
        if (rc) break;
        uint64_t n_hdr = 0, seq = 0;
        for (int t = 0; t < nthreads; t++) { n_hdr += ch[t].n_hdr; seq += ch[t].seq_total; }
        S->n_fasta = 0;
        if (n_hdr == 0) break;                                    /* empty text: no contig (len == 0) */
        uint64_t total = seq + n_hdr * CBC_REF_PAD;               /* one pad after every contig */
        if (grow((void **)&P->ref, &P->cap_ref, total, 1) ||
            grow32((void **)&P->contigs, &P->cap_contigs, n_hdr, sizeof(cbc_contig_info))) { rc = CBC_E_NOMEM; break; }
        uint64_t pos = 0; uint64_t cur_start = 0, cur_len = 0; int open = 0; uint32_t nc = 0;
        for (int t = 0; t < nthreads; t++) {
            ch[t].out = P->ref; ch[t].out_pos = pos;
            cur_len += ch[t].seq_before_first_hdr; pos += ch[t].seq_before_first_hdr;
            for (uint64_t k = 0; k < ch[t].n_hdr; k++) {
                if (open) {                                       /* this header closes the contig before it */
                    P->contigs[nc].ref_off = cur_start; P->contigs[nc].length = cur_len; P->contigs[nc].name_off = 0; P->contigs[nc].reserved = 0;
                    nc++; pos += CBC_REF_PAD;
                }
                open = 1; cur_start = pos; cur_len = ch[t].seg[k]; pos += ch[t].seg[k];
            }
        }
        P->contigs[nc].ref_off = cur_start; P->contigs[nc].length = cur_len; P->contigs[nc].name_off = 0; P->contigs[nc].reserved = 0;
        nc++;
        memset(P->ref + pos, 0, CBC_REF_PAD); pos += CBC_REF_PAD;
        P->ref_bytes = pos; S->n_fasta = nc;
    }
    for (int t = 0; t < nthreads; t++) free(ch[t].seg);
    free(ch); free(th);
    return rc;
}

static int online_cpus(void)
{
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    return nc < 1 ? 1 : nc > 64 ? 64 : (int)nc;
}
/* the FASTA loader behind both the packer and the unpack plan: threaded for big texts */
static int load_reference(packer_t *S, const char *fa, size_t len, uint32_t n_threads)
{
    int nt = n_threads ? (int)n_threads : online_cpus();
    if (nt > 64) nt = 64;
    if (nt > 1 && (n_threads > 1 || len >= ((size_t)8 << 20))) return load_fasta_mt(S, fa, len, nt);
    return load_fasta(S, fa, len);
}

static void close_block(packer_t *S)
{
    cbc_packed *P = S->P;
    if (!S->blk_open) return;
    cbc_block_desc *bd = &P->blocks[P->n_blocks];
    cbc_block_info *bi = &P->info[P->n_blocks];
    bd->n_reads = S->blk_reads;
    bd->n_tok = (uint32_t)(P->n_tok - bd->tok_base);
    bi->n_reads = S->blk_reads; bi->n_bases = S->blk_bases;
    uint32_t need_pos = S->blk_ndelta + 2, need_var = S->blk_var + 1;
    if (need_pos > P->caps.cap_pos) P->caps.cap_pos = need_pos;
    if (need_var > P->caps.cap_var) P->caps.cap_var = need_var;
    P->n_blocks++;
    S->blk_open = 0;
}

static int open_block(packer_t *S, uint32_t pos)
{
    cbc_packed *P = S->P;
    if (grow32((void **)&P->blocks, &P->cap_blocks, (uint64_t)P->n_blocks + 1, sizeof(cbc_block_desc))) return CBC_E_NOMEM;
    {   /* info array shares the block capacity */
        cbc_block_info *ni = (cbc_block_info *)realloc(P->info, sizeof(cbc_block_info) * (size_t)P->cap_blocks);
        if (!ni) return CBC_E_NOMEM;
        P->info = ni;
    }
    cbc_block_desc *bd = &P->blocks[P->n_blocks];
    cbc_block_info *bi = &P->info[P->n_blocks];
    memset(bd, 0, sizeof *bd); memset(bi, 0, sizeof *bi);
    const cbc_contig_info *c = &P->contigs[S->contig];
    bd->rec_base = P->n_recs; bd->seq_base = P->seq_bytes; bd->tok_base = P->n_tok;
    if (S->o.whole_file) pos = 1;                       /* a segment of the whole-file stream keeps the SAM's POS */
    bd->ref_off = c->ref_off + (uint64_t)(pos - 1);
    bd->name_off = c->name_off; bd->read_length = P->read_length;
    bi->contig = S->contig; bi->window_start = (uint64_t)(pos - 1);
    S->blk_open = 1; S->blk_first_pos = pos;
    S->blk_reads = 0; S->blk_bases = 0;
    if (S->o.whole_file && S->seg_continues) { S->seg_continues = 0; return 0; }   /* same contig, same stream state */
    S->blk_prev_pos = 0;
    if (S->o.whole_file) return 0;                      /* the models run on: flags and POS steps count per file */
    S->blk_var = 0; S->blk_nflags = 0; S->blk_ndelta = 0;
    S->depoch++;
    if (S->depoch == 0) { memset(S->dstamp, 0, sizeof(uint32_t) * (S->dmask + 1)); S->depoch = 1; }
    return 0;
}

static int delta_seen(packer_t *S, uint32_t x, int insert)
{
    uint32_t h = (x * 2654435761u) & S->dmask;
    for (;;) {
        if (S->dstamp[h] != S->depoch) {
            if (insert) { S->dstamp[h] = S->depoch; S->dset[h] = x; }
            return 0;
        }
        if (S->dset[h] == x) return 1;
        h = (h + 1) & S->dmask;
    }
}

/* RNAME change (compress_rname strcmp, id_compression.c:46): the next FASTA record becomes current. */
static int contig_open(packer_t *S, const char *rname, size_t nl)
{
    cbc_packed *P = S->P;
    close_block(S);
    uint32_t ci = S->have_contig ? S->contig + 1 : 0;
    char nm[64]; snprintf(nm, sizeof nm, "%.*s", (int)(nl < 60 ? nl : 60), rname);
    if (ci >= S->n_fasta) return fail(S, CBC_E_INPUT, "RNAME %s is contig #%lld of the SAM but the FASTA has fewer records (contigs are consumed in FASTA order)", nm, (long long)ci + 1);
    if (nl + 3 > CBC_CAP_NAME) return fail(S, CBC_E_INPUT, "RNAME %s longer than %lld characters", nm, (long long)CBC_CAP_NAME - 3);
    if (grow32((void **)&P->names, &P->cap_names, (uint64_t)P->names_bytes + nl + 1, 1)) return CBC_E_NOMEM;
    memcpy(P->names + P->names_bytes, rname, nl); P->names[P->names_bytes + nl] = 0;
    P->contigs[ci].name_off = P->names_bytes;
    P->names_bytes += (uint32_t)nl + 1;
    S->contig = ci; S->have_contig = 1;
    if (P->n_contigs < ci + 1) P->n_contigs = ci + 1;
    return 0;
}


/* ---- MD text derived from an alignment (quirk Q6: what the reference's compress_edits leaves in the record's
 * MD buffer when the CIGAR starts with a soft clip and the read is imperfect, read_compression.c:357-468).
 * Walk the ops after the clip with a cursor in the read (starts behind the clip) and one in the contig (starts
 * at POS): an M op compares base by base and writes "<run><contig letter>" for every mismatch (run 0 included);
 * I skips read bases; D closes the current run (only if > 0), writes '^' and the deleted contig letters -- taken
 * one base FURTHER than the cursor (no -1 there, :425) -- and moves the contig cursor; a trailing S only closes
 * the run; at the end the run is written if > 0.  Nothing terminates the text except the NUL that comes with a
 * number, so when the last thing written is a letter the buffer's older content follows it (:460 clears a
 * pointer, not the string) -- which is why this writes into the persistent buffer and not into a fresh one.
 * Returns 0, -1 = an op runs past SEQ or past the contig + pad, -2 = no room in the buffer. */
typedef struct { char *at, *end; } md_text;
static int md_number(md_text *w, uint32_t v)
{
    if (w->end - w->at < 12) return -2;
    w->at += sprintf(w->at, "%u", v);            /* leaves its NUL at *w->at */
    return 0;
}
static int md_letter(md_text *w, uint8_t c)
{
    if (w->end - w->at < 2) return -2;
    *w->at++ = (char)c;                          /* no terminator: see above */
    return 0;
}
static int md_from_alignment(char *buf, size_t buf_len, const uint32_t *ops, uint32_t n_ops, uint32_t clip_len,
                             const uint8_t *read, uint32_t read_len, const uint8_t *contig, uint32_t pos, uint64_t contig_avail)
{
    md_text w = { buf, buf + buf_len };
    uint64_t in_read = clip_len, in_contig = pos;         /* next read index (0-based); next contig base (1-based) */
    uint32_t run = 0;
    int rc = 0;
    for (uint32_t k = 0; k < n_ops && !rc; k++) {
        const uint32_t kind = ops[k] & 15u; const uint64_t n = ops[k] >> 4;
        if (kind == CBC_OP_M) {
            if (in_read + n > read_len || in_contig - 1 + n > contig_avail) return -1;
            for (uint64_t j = 0; j < n && !rc; j++) {
                const uint8_t have = read[in_read + j], want = contig[in_contig - 1 + j];
                if (have == want) run++;
                else { rc = md_number(&w, run); if (!rc) rc = md_letter(&w, want); run = 0; }
            }
            in_read += n; in_contig += n;
        } else if (kind == CBC_OP_I) in_read += n;
        else if (kind == CBC_OP_D) {
            if (in_contig + n > contig_avail) return -1;
            if (run) { rc = md_number(&w, run); run = 0; }
            if (!rc) rc = md_letter(&w, '^');
            for (uint64_t j = 0; j < n && !rc; j++) rc = md_letter(&w, (uint8_t)toupper(contig[in_contig + j]));
            in_contig += n;
        } else if (kind == CBC_OP_S) { if (run) { rc = md_number(&w, run); run = 0; } }
    }
    if (!rc && run) rc = md_number(&w, run);
    return rc;
}

/* CIGAR + MD of one mapped record -> token words tk[0..*nt_out) and the record's var-symbol bound.
 * Reads only shared, already final state (P->ref, the contig table), so worker threads may call it. */
static int tokenise_record(packer_t *S, const cbc_contig_info *ctg, const char *rname, int32_t pos_i, const char *cigar,
                           const char *seq, size_t rl, const char *edits, uint32_t *tk, uint32_t *nt_out, uint32_t *ev_out)
{
    cbc_packed *P = S->P;
    if (rl == 0 || rl > CBC_MAX_READ_LEN)
        return fail(S, CBC_E_INPUT, "read length %s%lld outside 1..252 (var-context limit of the reference, sam_models.c:317)", "", (long long)rl);
    if (pos_i < 1) return fail(S, CBC_E_INPUT, "POS %s%lld < 1", "", pos_i);
    uint32_t pos = (uint32_t)pos_i;
    if ((uint64_t)pos - 1 + rl > ctg->length + CBC_REF_PAD - 8)
        return fail(S, CBC_E_INPUT, "record at %s POS %lld runs past the contig end + pad", rname, pos);

    /* ---- tokens: CIGAR ---- */
    uint32_t nt = 2, n_cig = 0, n_md = 0;
    uint32_t ev = 0;                                    /* upper bound on var symbols of this record */
    int lead_clip = 0;                                  /* the first CIGAR op is a soft clip (quirk Q6) */
    {
        const char *seg = cigar; int i = 0;
        while (*seg != 0) {                             /* read_compression.c:308-549 scanning rule */
            char ch = seg[i];
            if (ch == 0) break;
            if (!isdigit((unsigned char)ch)) {
                uint32_t op = 0xff;
                if (ch == 'M') op = CBC_OP_M; else if (ch == 'I') op = CBC_OP_I; else if (ch == 'D') op = CBC_OP_D;
                else if (ch == 'S') op = CBC_OP_S; else if (ch == '*') op = CBC_OP_STAR;
                if (op != 0xff) {
                    long v = atoi(seg);                 /* atoi(cigar) of the unconsumed segment */
                    if (v < 0 || v > 0x0fffffff) return fail(S, CBC_E_INPUT, "CIGAR %s: bad length %lld", cigar, v);
                    if (nt >= 2 * LINE_BUF) return fail(S, CBC_E_INPUT, "CIGAR %s too long%lld", cigar, 0);
                    tk[nt++] = ((uint32_t)v << 4) | op; n_cig++;
                    if (op == CBC_OP_STAR) return fail(S, CBC_E_INPUT, "CIGAR '*' on a mapped record at %s:%lld (the reference aborts on it)", rname, pos);
                    if (op == CBC_OP_S && n_cig == 1) { lead_clip = 1; tk[nt - 1] = ((uint32_t)v << 4) | CBC_OP_I; }
                    if (op != CBC_OP_M) ev += (uint32_t)v;
                    seg = seg + i + 1; i = -1;
                }
            }
            i++;
        }
    }
    if (lead_clip) {
        /* Leading soft clip (quirk Q6, read_compression.c:357-468).  The clipped bases are coded exactly like
         * an insertion at matched coordinate 0 (the op was stored as CBC_OP_I above, which the kernels
         * already code that way), and for an IMPERFECT read the reference first replaces the record's MD
         * text by one it derives from the alignment itself (over the ops after the clip).  Only then. */
        const uint8_t *contig_bases = P->ref + ctg->ref_off;
        if (memcmp(seq, contig_bases + (pos - 1), rl) != 0) {
            int rc = md_from_alignment((char *)edits, 2 * LINE_BUF, tk + 3, n_cig - 1, tk[2] >> 4,
                                       (const uint8_t *)seq, (uint32_t)rl, contig_bases, pos,
                                       ctg->length + CBC_REF_PAD);
            if (rc == -1) return fail(S, CBC_E_INPUT, "CIGAR of the soft-clipped record at %s:%lld runs past SEQ or past the contig", rname, pos);
            if (rc == -2) return fail(S, CBC_E_INPUT, "derived MD too long at %s:%lld", rname, pos);
        }
    }
    /* ---- tokens: MD (add_snps_to_array scanning rule, consumption branch :661-695) ---- */
    {
        const char *p = edits;
        while (*p != 0) {
            uint32_t gap = (uint32_t)atoi(p);
            p += num_digits(gap);
            char ch = *p; if (ch) p++;
            int ended = 0;
            while (ch == '^') {
                while (*p && !isdigit((unsigned char)*p)) p++;
                if (!*p) { ended = 1; break; }
                uint32_t v = (uint32_t)atoi(p);
                gap += v; p += num_digits(v);
                ch = *p; if (ch) p++;
            }
            if (ended || ch == 0) break;
            if (gap > 0x00ffffff) return fail(S, CBC_E_INPUT, "MD %s: gap %lld too large", edits, gap);
            if (nt >= 4 * LINE_BUF - 1) return fail(S, CBC_E_INPUT, "MD %s too long%lld", edits, 0);
            tk[nt++] = (gap << 8) | (uint8_t)ch; n_md++;
            if (*p == 0) break;
        }
    }
    if (n_cig > 0xffff || n_md > 0xffff) return fail(S, CBC_E_INPUT, "too many CIGAR/MD tokens%s%lld", "", 0);
    tk[0] = n_cig | (n_md << 16);
    {
        /* Edit counts for the kernel (token word 1) and a consistency check: replay compress_edits'
         * CIGAR walk with add_snps_to_array's early-return rule (read_compression.c:308-352, 551-552,
         * 656-659) and require that every MD mismatch token is consumed.  An MD string that names
         * more bases than the read has makes the reference leave its parser statics dirty for the
         * NEXT record (undefined behaviour there); such input is rejected here. */
        uint32_t Mc = 0, ins = 0, nDel = 0, nIns = 0, k = 0, cum = 0, nS = 0; int more = 1;
        for (uint32_t o = 0; o <= n_cig; o++) {
            uint32_t op = 99, len = 1;
            if (o < n_cig) { op = tk[2 + o] & 15u; len = tk[2 + o] >> 4; }
            if (op == CBC_OP_M) { Mc += len; continue; }
            if (op == CBC_OP_D) { nDel += len; continue; }
            for (uint32_t c = 0; c < len; c++) {
                if ((op == CBC_OP_I || op == 99) && more) {
                    uint32_t limit = (op == 99) ? (uint32_t)rl + 1 : Mc + ins;
                    more = 0;
                    while (k < n_md) {
                        uint32_t g = tk[2 + n_cig + k] >> 8;
                        if (cum + g >= limit) { cum++; more = 1; break; }
                        cum += g + 1; nS++; k++;
                    }
                }
                if (op == 99) break;
                nIns++; ins++;
            }
        }
        if (nS != n_md)
            return fail(S, CBC_E_INPUT, "MD:Z:%s is inconsistent with CIGAR/SEQ at POS %lld (it names more bases than the read has)", edits, pos);
        if (nDel > 0xffff || nIns > 0xffff) return fail(S, CBC_E_INPUT, "too many inserted/deleted bases%s%lld", "", 0);
        tk[1] = nDel | (nIns << 16);
    }
    ev += n_md;
    if (ev + 1 > S->o.max_cap_var) return fail(S, CBC_E_INPUT, "record at %s:%lld has more edits than max_cap_var", rname, pos);
    *nt_out = nt; *ev_out = ev;
    return 0;
}

/* Block cut decision + the record's cbc_read_rec.  Advances P->seq_bytes / P->n_tok as counters; the
 * caller stores the SEQ bytes and token words at the offsets they had on entry. */
static int place_record(packer_t *S, uint32_t pos, uint32_t flag, size_t rl, uint32_t ev, uint32_t nt)
{
    cbc_packed *P = S->P;
    const char *rname = (const char *)P->names + P->contigs[S->contig].name_off;
    int need_new = !S->blk_open;
    uint32_t x = 0;
    if (S->o.whole_file) {
        /* one stream per file: cut only where a 32-bit offset of cbc_read_rec would overflow */
        if (S->blk_open) {
            const cbc_block_desc *cur = &P->blocks[P->n_blocks];
            if (pos < S->blk_prev_pos) return fail(S, CBC_E_INPUT, "SAM is not sorted by position at %s:%lld", rname, pos);
            if (P->seq_bytes - cur->seq_base + rl > 0xfff00000ull || P->n_tok - cur->tok_base + nt > 0xfff00000ull ||
                S->blk_reads >= 0xfff00000u) { need_new = 1; S->seg_continues = 1; }
        }
        x = pos - (S->blk_open ? S->blk_prev_pos : 0u) + 1;   /* prevPos is 0 at a contig's first record (compress_pos :123-124) */
        if (x >= 5000000u)
            return fail(S, CBC_E_INPUT, "POS step of 5 000 000 or more at %s:%lld: outside the reference's pos alphabet (MAX_ALPHA); use block mode", rname, pos);
        if (need_new) { close_block(S); int rc = open_block(S, pos); if (rc) return rc; }
        /* every POS step below MAX_ALPHA and every 16-bit FLAG is in the reference's tables (sam_block.h:54-55,
         * sam_models.c:96-130), so the whole-file stream takes them all: the stream kernels keep what does not fit their
         * registers / LDS in global memory.  The distinct steps are counted exactly (one bit per possible step): the
         * count sizes the kernels' pos tables. */
        if (!(S->wf_seen[x >> 3] & (1u << (x & 7u)))) { S->wf_seen[x >> 3] |= (uint8_t)(1u << (x & 7u)); S->blk_ndelta++; }
        goto record;
    }
    if (S->blk_open) {
        if (pos < S->blk_prev_pos) return fail(S, CBC_E_INPUT, "SAM is not sorted by position at %s:%lld", rname, pos);
        x = pos - S->blk_prev_pos + 1;
        int newflag = 1;
        for (uint32_t i = 0; i < S->blk_nflags; i++) if (S->blk_flags[i] == (uint16_t)flag) { newflag = 0; break; }
        int newdelta = !delta_seen(S, x, 0);
        if (S->blk_reads >= S->o.block_reads) need_new = 1;
        else if (S->o.long_reads && S->blk_bases + rl > CBC_LONG_BLOCK_BASES) need_new = 1;
        else if (x >= 5000000u && !S->o.long_reads) need_new = 1;              /* MAX_ALPHA sam_block.h:54 */
        else if (newdelta && S->blk_ndelta + 3 > S->o.max_cap_pos) need_new = 1;
        else if (S->blk_var + ev + 1 > S->o.max_cap_var) need_new = 1;
        else if (newflag && S->blk_nflags >= CBC_CAP_FLAG) need_new = 1;
        else if ((uint64_t)pos - S->blk_first_pos > 0xfff00000ull) need_new = 1;
    }
    if (need_new) {
        close_block(S);
        int rc = open_block(S, pos);
        if (rc) return rc;
        x = 2;                                          /* local POS 1, prevPos 0 */
    }
    if (!delta_seen(S, x, 1)) S->blk_ndelta++;
    {
        int newflag = 1;
        for (uint32_t i = 0; i < S->blk_nflags; i++) if (S->blk_flags[i] == (uint16_t)flag) { newflag = 0; break; }
        if (newflag) S->blk_flags[S->blk_nflags++] = (uint16_t)flag;
    }
    S->blk_var += ev;

record:;
    /* ---- record ---- */
    cbc_block_desc *bd = &P->blocks[P->n_blocks];
    if (grow((void **)&P->recs, &P->cap_recs, P->n_recs + 1, sizeof(cbc_read_rec))) return CBC_E_NOMEM;
    cbc_read_rec *r = &P->recs[P->n_recs++];
    r->pos = (uint32_t)((uint64_t)pos - S->blk_first_pos + 1);
    r->flag = (uint16_t)flag; r->rlen = (uint16_t)rl;
    r->seq_off = (uint32_t)(P->seq_bytes - bd->seq_base);
    r->tok_off = (uint32_t)(P->n_tok - bd->tok_base);
    P->seq_bytes += rl; P->n_tok += nt;
    S->blk_reads++; S->blk_bases += rl; S->blk_prev_pos = pos;
    P->n_bases += rl;
    if (rl > P->max_read_len) P->max_read_len = (uint32_t)rl;
    return 0;
}

/* One mapped record, fields exactly as load_sam_line leaves them in read_line_t (single-thread path). */
static int add_record(packer_t *S, const char *rname, uint32_t flag, int32_t pos_i, const char *cigar,
                      const char *seq, const char *edits)
{
    cbc_packed *P = S->P;
    size_t rl = strlen(seq);
    if (!S->have_contig || strcmp(rname, S->prev_name) != 0) {
        int rc = contig_open(S, rname, strlen(rname));
        if (rc) return rc;
        snprintf(S->prev_name, sizeof S->prev_name, "%s", rname);
    }
    uint32_t nt = 0, ev = 0;
    int rc = tokenise_record(S, &P->contigs[S->contig], rname, pos_i, cigar, seq, rl, edits, S->tokbuf, &nt, &ev);
    if (rc) return rc;
    uint64_t so = P->seq_bytes, to = P->n_tok;
    if (grow((void **)&P->seq, &P->cap_seq, so + rl + 8, 1)) return CBC_E_NOMEM;
    if (grow((void **)&P->tok, &P->cap_tok, to + nt, sizeof(uint32_t))) return CBC_E_NOMEM;
    rc = place_record(S, (uint32_t)pos_i, flag, rl, ev, nt);
    if (rc) return rc;
    memcpy(P->seq + so, seq, rl);
    memcpy(P->tok + to, S->tokbuf, sizeof(uint32_t) * nt);
    return 0;
}


/* ================================= long-read format ====================== */
/* One mapped record of a long-read SAM: fields by pointer and length (lines are not copied: a 10 kb read's line
 * is ~20 kB).  Tokens: word 0 = n_cigar, word 1 = 0, then (len << 4) | op per CIGAR op; MD is not used -- the
 * kernels derive the edits from read vs reference.  SEQ must be over {A,C,G,T,N} so that the round trip is exact
 * (the base models code basepair classes, sam_models.c:11-33). */
static int add_record_long(packer_t *S, const char *rname, size_t nl, uint32_t flag, int64_t pos_i, const char *cigar, size_t cl,
                           const char *seq, size_t rl)
{
    cbc_packed *P = S->P;
    if (!S->have_contig || nl != strlen(S->prev_name) || memcmp(rname, S->prev_name, nl) != 0) {
        if (nl >= LINE_BUF) return fail(S, CBC_E_INPUT, "RNAME longer than %s%lld characters", "", (long long)LINE_BUF - 1);
        int rc = contig_open(S, rname, nl);
        if (rc) return rc;
        memcpy(S->prev_name, rname, nl); S->prev_name[nl] = 0;
    }
    const cbc_contig_info *ctg = &P->contigs[S->contig];
    const char *nm = (const char *)P->names + ctg->name_off;
    if (rl == 0 || rl > CBC_LONG_MAX_READ_LEN) return fail(S, CBC_E_INPUT, "read length %s%lld outside 1..65535", "", (long long)rl);
    if (pos_i < 1 || pos_i > 0x7fffffff) return fail(S, CBC_E_INPUT, "POS %s%lld outside 1..2^31-1", "", (long long)pos_i);
    for (size_t i = 0; i < rl; i++) {
        char c = seq[i];
        if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N'))
            return fail(S, CBC_E_INPUT, "SEQ at %s:%lld holds a character outside ACGTN (long-read format)", nm, (long long)pos_i);
    }
    /* CIGAR -> tokens, and what it consumes */
    uint64_t to = P->n_tok, nt = 2, in_read = 0, in_ref = 0; uint32_t n_cig = 0;
    {
        size_t i = 0;
        while (i < cl) {
            uint64_t v = 0; size_t d0 = i;
            while (i < cl && cigar[i] >= '0' && cigar[i] <= '9') { v = v * 10 + (uint64_t)(cigar[i] - '0'); if (v > 0x0fffffff) break; i++; }
            if (i >= cl || i == d0 || v == 0 || v > 0x0fffffff) return fail(S, CBC_E_INPUT, "malformed CIGAR at %s:%lld", nm, (long long)pos_i);
            char c = cigar[i++];
            uint32_t op;
            if (c == 'M' || c == '=' || c == 'X') { op = CBC_OP_M; in_read += v; in_ref += v; }
            else if (c == 'I' || c == 'S') { op = (c == 'I') ? CBC_OP_I : CBC_OP_S; in_read += v; }
            else if (c == 'D' || c == 'N') { op = CBC_OP_D; in_ref += v; }
            else if (c == 'H' || c == 'P') continue;                      /* consume nothing, code nothing */
            else return fail(S, CBC_E_INPUT, "CIGAR operation not understood at %s:%lld", nm, (long long)pos_i);
            if (grow((void **)&P->tok, &P->cap_tok, to + nt + 1, sizeof(uint32_t))) return CBC_E_NOMEM;
            P->tok[to + nt++] = ((uint32_t)v << 4) | op; n_cig++;
            if (n_cig > 0xffff) return fail(S, CBC_E_INPUT, "more than 65535 CIGAR operations at %s:%lld", nm, (long long)pos_i);
        }
    }
    if (in_read != rl) return fail(S, CBC_E_INPUT, "CIGAR and SEQ lengths differ at %s:%lld", nm, (long long)pos_i);
    if ((uint64_t)pos_i - 1 + in_ref > ctg->length + CBC_REF_PAD - 8) return fail(S, CBC_E_INPUT, "record at %s POS %lld runs past the contig end + pad", nm, (long long)pos_i);
    if (grow((void **)&P->tok, &P->cap_tok, to + 2, sizeof(uint32_t))) return CBC_E_NOMEM;
    P->tok[to] = n_cig; P->tok[to + 1] = 0;
    uint64_t so = P->seq_bytes;
    if (grow((void **)&P->seq, &P->cap_seq, so + rl + 8, 1)) return CBC_E_NOMEM;
    int rc = place_record(S, (uint32_t)pos_i, flag, rl, 0, (uint32_t)nt);
    if (rc) return rc;
    memcpy(P->seq + so, seq, rl);
    return 0;
}

/* the body of a long-read SAM: lines of any length, the 11 compulsory columns by tab */
static int pack_body_long(packer_t *S, const char *sam, size_t off, size_t sam_len)
{
    while (off < sam_len) {
        const char *nlp = (const char *)memchr(sam + off, '\n', sam_len - off);
        size_t e = nlp ? (size_t)(nlp - sam) : sam_len;
        const char *f[11]; size_t fl[11]; int nf = 0;
        size_t p = off;
        while (nf < 11 && p <= e) {
            const char *t = (const char *)memchr(sam + p, '\t', e - p);
            size_t q = t ? (size_t)(t - sam) : e;
            f[nf] = sam + p; fl[nf] = q - p; nf++;
            if (!t) break;
            p = q + 1;
        }
        size_t line_off = off;
        off = e + 1;
        if (nf == 0 || (nf == 1 && fl[0] == 0)) continue;
        if (nf < 11) return fail(S, CBC_E_INPUT, "SAM record with fewer than 11 columns near offset %s%lld", "", (long long)line_off);
        char num[24]; size_t l = fl[1] < 23 ? fl[1] : 23; memcpy(num, f[1], l); num[l] = 0;
        uint32_t flag = (uint16_t)atoi(num);
        if ((flag & 4) == 4) { S->P->n_skipped_unmapped++; continue; }
        l = fl[3] < 23 ? fl[3] : 23; memcpy(num, f[3], l); num[l] = 0;
        int rc = add_record_long(S, f[2], fl[2], flag, atoll(num), f[5], fl[5], f[9], fl[9]);
        if (rc) return rc;
    }
    return 0;
}

static int finish_pack(packer_t *S)
{
    cbc_packed *P = S->P;
    close_block(S);
    if (grow((void **)&P->seq, &P->cap_seq, P->seq_bytes + 8, 1)) return CBC_E_NOMEM;
    memset(P->seq + P->seq_bytes, 0, 8); P->seq_bytes += 8;          /* the kernel reads whole dwords */
    if (P->caps.cap_pos < 64) P->caps.cap_pos = 64;
    if (P->caps.cap_var < 64) P->caps.cap_var = 64;
    P->caps.cap_pos = (P->caps.cap_pos + 63u) & ~63u;
    P->caps.cap_var = (P->caps.cap_var + 63u) & ~63u;
    if (P->n_tok == 0) { if (grow((void **)&P->tok, &P->cap_tok, 1, sizeof(uint32_t))) return CBC_E_NOMEM; P->tok[0] = 0; }
    P->whole_file = S->o.long_reads ? 2u : S->o.whole_file ? 1u : 0u;
    return 0;
}

static int packer_init(packer_t *S, const cbc_pack_opts *opts, char *errbuf, size_t errlen)
{
    memset(S, 0, sizeof *S);
    if (opts) S->o = *opts; else cbc_pack_default_opts(&S->o);
    if (S->o.long_reads) {                                  /* blocks are cut by bases: 64 reads of 10 kb fill one */
        S->o.whole_file = 0;
        if (S->o.block_reads == 0 || S->o.block_reads > 64) S->o.block_reads = 64;   /* format limit: the per-read models keep <= 64 entries */
    }
    if (S->o.block_reads == 0) S->o.block_reads = 4096;
    if (S->o.block_reads > CBC_MAX_BLOCK_READS) S->o.block_reads = CBC_MAX_BLOCK_READS;
    if (S->o.max_cap_pos < 64) S->o.max_cap_pos = 2048;
    if (S->o.max_cap_var < 64) S->o.max_cap_var = 8192;
    if (S->o.whole_file) S->o.max_cap_pos = 8192;                  /* (sizes the hash set below, which whole-file mode does not use) */
    if (S->o.max_cap_pos > 4096 && !S->o.whole_file) S->o.max_cap_pos = 4096;
    if (S->o.max_cap_var > 32768) S->o.max_cap_var = 32768;       /* keeps L0 + 10*uses < 2^20 */
    S->err = errbuf; S->errlen = errlen;
    if (errbuf && errlen) errbuf[0] = 0;
    S->P = (cbc_packed *)calloc(1, sizeof(cbc_packed));
    if (!S->P) return CBC_E_NOMEM;
    uint32_t sz = 1; while (sz < 4 * S->o.max_cap_pos) sz <<= 1;
    S->dmask = sz - 1;
    S->dset = (uint32_t *)calloc(sz, sizeof(uint32_t));
    S->dstamp = (uint32_t *)calloc(sz, sizeof(uint32_t));
    if (!S->dset || !S->dstamp) return CBC_E_NOMEM;
    S->depoch = 1;                                  /* stamps start at 0 = "never used" */
    if (S->o.whole_file) {
        S->wf_seen = (uint8_t *)calloc(5000000u / 8u + 1u, 1);
        if (!S->wf_seen) return CBC_E_NOMEM;
    }
    return 0;
}
static void packer_release(packer_t *S) { free(S->dset); free(S->dstamp); free(S->wf_seen); }

API void cbc_pack_default_opts(cbc_pack_opts *o)
{
    o->block_reads = 4096; o->max_cap_pos = 2048; o->max_cap_var = 8192; o->var_length = 0; o->n_threads = 0; o->whole_file = 0; o->long_reads = 0;
}

API void cbc_packed_free(cbc_packed *p)
{
    if (!p) return;
    free(p->recs); free(p->seq); free(p->tok); free(p->names); free(p->blocks); free(p->info);
    free(p->contigs); free(p->ref); free(p);
}
API void cbc_free(void *p) { free(p); }

/* get_read_length (sam_file_allocation.c:26-79): skip '@' lines, skip the first record line, take
 * the 10th whitespace-separated field of what follows; a file with a single record makes the
 * reference's fscanf fail and it returns strlen() of that first line. */
static uint32_t header_read_length(const char *sam, size_t len, size_t *body_off, int var_length)
{
    size_t off = 0;
    while (off < len && sam[off] == '@') { while (off < len && sam[off] != '\n') off++; if (off < len) off++; }
    *body_off = off;
    size_t e = off; while (e < len && sam[e] != '\n') e++;
    size_t first_len = (e < len ? e + 1 : e) - off; if (first_len > 4095) first_len = 4095;
    size_t q = e < len ? e + 1 : e;
    uint32_t result = 0; int got = 0;
    for (;;) {
        int field = 0; size_t s = q, t = q; int ok = 0;
        while (s < len) {
            while (s < len && isspace((unsigned char)sam[s])) s++;
            if (s >= len) break;
            t = s; while (t < len && !isspace((unsigned char)sam[t])) t++;
            if (++field == 10) { ok = 1; break; }
            s = t;
        }
        if (!ok) break;
        uint32_t l = (uint32_t)(t - s);
        if (!var_length) { result = l; got = 1; break; }
        if (l > result) result = l;
        got = 1;
        while (t < len && sam[t] != '\n') t++;
        if (t >= len) break;
        q = t + 1;
    }
    if (!got && !var_length) return (uint32_t)first_len;
    return result;
}

/* One text line (NUL-terminated, with its '\n' as fgets leaves it) -> the 11 compulsory columns by
 * strtok("\t") and the MD/XD aux field copied into S->edits, which otherwise keeps the previous
 * line's value (read_line_t.edits persists, sam_file_allocation.c:437-529).
 * Returns 1 = record, 0 = nothing on the line, < 0 = error. */
static int split_line(packer_t *S, char *buffer, size_t off_after, char **f, uint32_t *flag, int32_t *pos, int *md_seen)
{
    char *save = NULL; int nf = 0;
    *md_seen = 0;
    while (nf < 11) {
        char *t = strtok_r(nf ? NULL : buffer, "\t", &save);
        if (!t) break;
        f[nf++] = t;
    }
    if (nf == 0) return 0;
    if (nf < 11) return fail(S, CBC_E_INPUT, "SAM record with fewer than 11 columns near offset %s%lld", "", (long long)off_after);
    *flag = (uint16_t)atoi(f[1]);
    *pos = atoi(f[3]);
    int auxCnt = 0;
    for (char *t = strtok_r(NULL, "\t", &save); t; t = strtok_r(NULL, "\t", &save)) {
        if ((t[0] == 'M' || t[0] == 'X') && t[1] == 'D') {
            size_t tl = strlen(t);
            strcpy(S->edits, tl >= 5 ? t + 5 : "");
            *md_seen = 1;
        } else { auxCnt++; if (auxCnt == 20) break; }
    }
    return 1;
}

/* =============================== multi-threaded text path ================================= */
/* The SAM body is cut into one chunk per thread at line boundaries.
 *   phase 1 (parallel)   RNAME changes inside each chunk (mapped records only)
 *   phase 1b (serial)    contig numbering across chunks, names table
 *   phase 2 (parallel)   split + tokenise every line into chunk-local record / SEQ / token pools
 *   phase 3 (serial)     block cutting over the 16-byte record summaries  } run
 *   phase 4 (parallel)   pools -> the final SEQ / token arrays             } concurrently
 * The result is identical, array for array, to the serial path.  A chunk whose first lines carry no
 * MD/XD field would need the previous chunk's (possibly rewritten) MD text: that case, which real
 * aligner output does not produce, returns MT_FALLBACK and the serial path runs instead. */
#define MT_FALLBACK   1000
#define MT_MAX_THREADS 64
#define MT_MIN_BYTES  (1u << 20)

typedef struct { uint32_t pos; uint16_t flag, rl; uint32_t nt_ev; uint32_t contig; } lrec_t;   /* nt | ev << 16 */
typedef struct { uint64_t ord; const char *name; uint32_t len, contig; } chg_t;

typedef struct chunk {
    packer_t *S;                       /* thread-local parse state; S->P is the shared result */
    const char *sam; size_t beg, end;
    /* phase 1 */
    uint64_t n_lines, n_mapped, seq_pred;
    chg_t *chg; uint64_t n_chg, cap_chg;
    const char *first_name, *last_name; uint32_t first_len, last_len;
    /* phase 1b */
    uint32_t contig0;                  /* contig of the chunk's first mapped record */
    /* phase 2 */
    lrec_t *lr; uint64_t n_lr;
    uint8_t *seq; uint64_t n_seq;      /* written in place: P->seq + seq_dst, seq_pred bytes */
    uint32_t *tok; uint64_t n_tok, cap_tok;
    uint64_t n_unmapped;
    int rc; char err[512];
    /* phase 4 */
    uint64_t seq_dst, tok_dst;
    int phase;
} chunk_t;

static inline const char *next_field(const char *p, const char *e, const char **tok_end)
{   /* strtok("\t"): skip tabs, the token runs to the next tab or the line end */
    while (p < e && *p == '\t') p++;
    if (p >= e) return NULL;
    const char *q = (const char *)memchr(p, '\t', (size_t)(e - p));
    *tok_end = q ? q : e;
    return p;
}

static void chunk_phase1(chunk_t *c)
{
    const char *p = c->sam + c->beg, *end = c->sam + c->end;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *e = nl ? nl + 1 : end;
        c->n_lines++;
        const char *t1e, *t2e, *t3e;
        const char *t1 = next_field(p, e, &t1e);
        const char *t2 = t1 ? next_field(t1e, e, &t2e) : NULL;
        const char *t3 = t2 ? next_field(t2e, e, &t3e) : NULL;
        if (t3 && t3e < e && *t3e == '\t') {
            char num[16]; size_t l = (size_t)(t2e - t2); if (l > 15) l = 15;
            memcpy(num, t2, l); num[l] = 0;
            uint32_t flag = (uint16_t)atoi(num);
            if ((flag & 4) != 4) {
                const char *q = t3e, *qe = t3e, *t10 = NULL;
                for (int k = 4; k <= 10 && q; k++) { t10 = next_field(qe, e, &qe); q = t10; }
                if (t10 && qe < e && *qe == '\t') c->seq_pred += (uint64_t)(qe - t10);   /* an 11th column follows */
                uint32_t len = (uint32_t)(t3e - t3);
                if (c->n_mapped == 0) { c->first_name = t3; c->first_len = len; }
                else if (len != c->last_len || memcmp(t3, c->last_name, len) != 0) {
                    if (grow((void **)&c->chg, &c->cap_chg, c->n_chg + 1, sizeof(chg_t))) { c->rc = CBC_E_NOMEM; return; }
                    c->chg[c->n_chg].ord = c->n_mapped; c->chg[c->n_chg].name = t3; c->chg[c->n_chg].len = len; c->chg[c->n_chg].contig = 0;
                    c->n_chg++;
                }
                c->last_name = t3; c->last_len = len;
                c->n_mapped++;
            }
        }
        p = e;
    }
}

static void chunk_phase2(chunk_t *c)
{
    packer_t *S = c->S; cbc_packed *P = S->P;
    const char *sam = c->sam;
    size_t off = c->beg;
    char buffer[LINE_BUF];
    int have_md = 0;
    uint64_t ord = 0, k = 0; uint32_t contig = c->contig0;
    c->lr = (lrec_t *)malloc(sizeof(lrec_t) * (size_t)(c->n_mapped ? c->n_mapped : 1));
    if (!c->lr) { c->rc = CBC_E_NOMEM; return; }
    while (off < c->end) {
        size_t e = off; { const char *nl = (const char *)memchr(sam + off, '\n', c->end - off); e = nl ? (size_t)(nl - sam) : c->end; }
        size_t ll = (e < c->end ? e + 1 : e) - off;
        if (ll > LINE_BUF - 1) { c->rc = fail(S, CBC_E_INPUT, "SAM line longer than %s1023 bytes at offset %lld (reference fgets limit)", "", (long long)off); return; }
        memcpy(buffer, sam + off, ll); buffer[ll] = 0;
        off += ll;
        char *f[11]; uint32_t flag; int32_t pos; int md_seen;
        int r = split_line(S, buffer, off, f, &flag, &pos, &md_seen);
        if (r < 0) { c->rc = r; return; }
        if (r == 0) continue;
        if (md_seen) have_md = 1;
        else if (!have_md) { c->rc = MT_FALLBACK; return; }       /* would inherit MD text from before the chunk */
        if ((flag & 4) == 4) { c->n_unmapped++; continue; }
        if (ord >= c->n_mapped) { c->rc = fail(S, CBC_E_INPUT, "internal: chunk record count changed%s%lld", "", 0); return; }
        if (k < c->n_chg && c->chg[k].ord == ord) contig = c->chg[k++].contig;
        size_t rl = strlen(f[9]);
        uint32_t nt = 0, ev = 0;
        if (grow((void **)&c->tok, &c->cap_tok, c->n_tok + 4 * LINE_BUF, sizeof(uint32_t))) { c->rc = CBC_E_NOMEM; return; }
        int rc = tokenise_record(S, &P->contigs[contig], f[2], pos, f[5], f[9], rl, S->edits, c->tok + c->n_tok, &nt, &ev);
        if (rc) { c->rc = rc; return; }
        if (c->n_seq + rl > c->seq_pred) { c->rc = fail(S, CBC_E_INPUT, "internal: chunk SEQ bytes changed%s%lld", "", 0); return; }
        memcpy(c->seq + c->n_seq, f[9], rl);
        c->n_seq += rl; c->n_tok += nt;
        lrec_t *l = &c->lr[ord++];
        l->pos = (uint32_t)pos; l->flag = (uint16_t)flag; l->rl = (uint16_t)rl; l->nt_ev = nt | (ev << 16); l->contig = contig;
    }
    c->n_lr = ord;
}

static void chunk_phase4(chunk_t *c)
{
    cbc_packed *P = c->S->P;
    if (c->n_tok) memcpy(P->tok + c->tok_dst, c->tok, sizeof(uint32_t) * (size_t)c->n_tok);
    free(c->tok); c->tok = NULL;
}

static void *chunk_run(void *arg)
{
    chunk_t *c = (chunk_t *)arg;
    if (c->phase == 1) chunk_phase1(c); else if (c->phase == 2) chunk_phase2(c); else chunk_phase4(c);
    return NULL;
}

static int run_phase(chunk_t *ch, pthread_t *th, int n, int phase, int wait)
{
    int started = 0;
    for (int t = 0; t < n; t++) {
        ch[t].phase = phase;
        if (pthread_create(&th[t], NULL, chunk_run, &ch[t]) != 0) break;
        started++;
    }
    for (int t = started; t < n; t++) chunk_run(&ch[t]);           /* could not spawn: do it here */
    if (wait) for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    return started;
}

static double mt_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static int pack_body_mt(packer_t *S, const char *sam, size_t off, size_t sam_len, int nthreads)
{
    cbc_packed *P = S->P;
    chunk_t *ch = (chunk_t *)calloc((size_t)nthreads, sizeof(chunk_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    packer_t *ps = (packer_t *)calloc((size_t)nthreads, sizeof(packer_t));
    int rc = 0, committed = 0;
    if (!ch || !th || !ps) { rc = CBC_E_NOMEM; goto out; }
    {   /* chunk boundaries at line starts */
        size_t body = sam_len - off, b = off;
        for (int t = 0; t < nthreads; t++) {
            size_t e = (t == nthreads - 1) ? sam_len : off + body / (size_t)nthreads * (size_t)(t + 1);
            if (e < b) e = b;
            if (e < sam_len && t != nthreads - 1) {
                const char *nl = (const char *)memchr(sam + e, '\n', sam_len - e);
                e = nl ? (size_t)(nl - sam) + 1 : sam_len;
            }
            ch[t].sam = sam; ch[t].beg = b; ch[t].end = e; b = e;
            ps[t].P = P; ps[t].o = S->o; ps[t].err = ch[t].err; ps[t].errlen = sizeof ch[t].err; ps[t].n_fasta = S->n_fasta;
            ch[t].S = &ps[t];
        }
    }
    double T0 = mt_now();
    run_phase(ch, th, nthreads, 1, 1);
    double T1 = mt_now();
    for (int t = 0; t < nthreads; t++) if (ch[t].rc) { rc = ch[t].rc; goto out; }
    /* phase 1b: contig numbering in file order */
    {
        const char *prev = NULL; uint32_t prev_len = 0;
        uint32_t names0 = P->names_bytes;
        for (int t = 0; t < nthreads && !rc; t++) {
            chunk_t *c = &ch[t];
            if (c->n_mapped == 0) { c->contig0 = S->contig; continue; }
            if (!S->have_contig || c->first_len != prev_len || memcmp(c->first_name, prev, prev_len) != 0)
                rc = contig_open(S, c->first_name, c->first_len);
            c->contig0 = S->contig;
            for (uint64_t k = 0; k < c->n_chg && !rc; k++) {
                rc = contig_open(S, c->chg[k].name, c->chg[k].len);
                c->chg[k].contig = S->contig;
            }
            prev = c->last_name; prev_len = c->last_len;
        }
        if (rc) goto out;           /* contig errors are input errors, not fallbacks */
        (void)names0;
    }
    {   /* SEQ bytes go straight to their final place: phase 1 measured them */
        uint64_t nseq = P->seq_bytes;
        for (int t = 0; t < nthreads; t++) { ch[t].seq_dst = nseq; nseq += ch[t].seq_pred; }
        if (grow((void **)&P->seq, &P->cap_seq, nseq + 8, 1)) { rc = CBC_E_NOMEM; goto out; }
        for (int t = 0; t < nthreads; t++) ch[t].seq = P->seq + ch[t].seq_dst;
    }
    double T2 = mt_now();
    run_phase(ch, th, nthreads, 2, 1);
    double T3 = mt_now();
    for (int t = 0; t < nthreads; t++) if (ch[t].rc == MT_FALLBACK) { rc = MT_FALLBACK; goto out; }
    for (int t = 0; t < nthreads; t++) if (ch[t].rc) {
        rc = ch[t].rc;
        if (S->err && S->errlen) snprintf(S->err, S->errlen, "%s", ch[t].err);
        goto out;
    }
    /* sizes -> final arrays */
    {
        uint64_t nrec = 0, nseq = 0, ntok = 0;
        for (int t = 0; t < nthreads; t++) {
            if (ch[t].n_seq != ch[t].seq_pred) { rc = fail(S, CBC_E_INPUT, "internal: chunk SEQ bytes changed%s%lld", "", 0); goto out; }
            ch[t].tok_dst = ntok; nrec += ch[t].n_lr; nseq += ch[t].n_seq; ntok += ch[t].n_tok; P->n_skipped_unmapped += ch[t].n_unmapped;
        }
        if (grow((void **)&P->recs, &P->cap_recs, nrec + 1, sizeof(cbc_read_rec)) ||
            grow((void **)&P->tok, &P->cap_tok, ntok + 1, sizeof(uint32_t))) { rc = CBC_E_NOMEM; goto out; }
    }
    committed = 1;
    double T3b = mt_now();
    {
        int started = run_phase(ch, th, nthreads, 4, 0);
        /* phase 3 on this thread while the pools are copied */
        S->have_contig = 0;                              /* replay the contig sequence for block cutting */
        uint32_t cur = 0;
        for (int t = 0; t < nthreads && !rc; t++) {
            const lrec_t *lr = ch[t].lr;
            for (uint64_t i = 0; i < ch[t].n_lr; i++) {
                const lrec_t *l = &lr[i];
                if (!S->have_contig || l->contig != cur) { close_block(S); cur = l->contig; S->contig = cur; S->have_contig = 1; }
                rc = place_record(S, l->pos, l->flag, l->rl, l->nt_ev >> 16, l->nt_ev & 0xffffu);
                if (rc) break;
            }
        }
        double T4 = mt_now();
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
        if (getenv("CBC_PACK_TIMES")) fprintf(stderr, "pack_mt: scan %.3f  contigs %.3f  tokenise %.3f  alloc %.3f  place %.3f  copy-wait %.3f s\n", T1 - T0, T2 - T1, T3 - T2, T3b - T3, T4 - T3b, mt_now() - T4);
    }
out:
    if (ch) for (int t = 0; t < nthreads; t++) { free(ch[t].chg); free(ch[t].lr); free(ch[t].tok); }
    free(ch); free(th); free(ps);
    if (rc == MT_FALLBACK && !committed) {
        /* undo phase 1b so that the serial path starts clean */
        P->names_bytes = 0; P->n_contigs = 0; S->have_contig = 0; S->contig = 0; P->n_skipped_unmapped = 0;
    }
    return rc;
}

API int cbc_pack_sam(const char *sam, size_t sam_len, const char *fasta, size_t fasta_len,
                     const cbc_pack_opts *opts, cbc_packed **out, char *errbuf, size_t errlen)
{
    if (!sam || !fasta || !out) return CBC_E_ARG;
    packer_t *S = (packer_t *)malloc(sizeof(packer_t));
    if (!S) return CBC_E_NOMEM;
    int rc = packer_init(S, opts, errbuf, errlen);
    if (rc) goto done;
    rc = load_reference(S, fasta, fasta_len, S->o.n_threads);
    if (rc) goto done;
    size_t off = 0;
    S->P->read_length = header_read_length(sam, sam_len, &off, (int)S->o.var_length);
    if (S->o.long_reads) {
        if (S->P->read_length < 1) S->P->read_length = 1;       /* not part of the long-read stream */
        if (S->P->read_length > 256) S->P->read_length = 256;
        rc = pack_body_long(S, sam, off, sam_len);
        if (!rc) rc = finish_pack(S);
        goto done;
    }
    if (S->P->read_length < 1 || S->P->read_length > 256) {
        rc = fail(S, CBC_E_INPUT, "header read length %s%lld outside 1..256 (the reference's limit is 252 bases per read; longer reads need the long-read format: cbc --long / long_reads)", "", S->P->read_length); goto done;
    }
    if (S->o.n_threads != 1) {
        int nt = (int)S->o.n_threads;
        if (nt == 0) {                                  /* auto: one per online CPU, serial for small inputs */
            long nc = sysconf(_SC_NPROCESSORS_ONLN);
            nt = nc < 1 ? 1 : nc > MT_MAX_THREADS ? MT_MAX_THREADS : (int)nc;
            if (sam_len - off < MT_MIN_BYTES) nt = 1;
        }
        if (nt > MT_MAX_THREADS) nt = MT_MAX_THREADS;
        if (nt > 1) {
            rc = pack_body_mt(S, sam, off, sam_len, nt);
            if (rc != MT_FALLBACK) { if (!rc) rc = finish_pack(S); goto done; }
            rc = 0;                                     /* nothing was committed: run the serial path */
        }
    }
    char buffer[LINE_BUF];
    while (off < sam_len) {
        size_t e = off; while (e < sam_len && sam[e] != '\n') e++;
        size_t ll = (e < sam_len ? e + 1 : e) - off;           /* includes the '\n' like fgets */
        if (ll > LINE_BUF - 1) { rc = fail(S, CBC_E_INPUT, "SAM line longer than %s1023 bytes at offset %lld (reference fgets limit)", "", (long long)off); goto done; }
        memcpy(buffer, sam + off, ll); buffer[ll] = 0;
        off += ll;
        char *f[11]; uint32_t flag; int32_t pos; int md_seen;
        int r = split_line(S, buffer, off, f, &flag, &pos, &md_seen);
        if (r < 0) { rc = r; goto done; }
        if (r == 0) continue;
        if ((flag & 4) == 4) { S->P->n_skipped_unmapped++; continue; }      /* compression.c:50-52 */
        rc = add_record(S, f[2], flag, pos, f[5], f[9], S->edits);
        if (rc) goto done;
    }
    rc = finish_pack(S);
done:
    packer_release(S);
    if (rc) { cbc_packed_free(S->P); *out = NULL; } else *out = S->P;
    free(S);
    return rc;
}

/* ================================= synthetic workload ===================================== */
typedef struct { uint64_t s[4]; } rng_t;
static uint64_t splitmix64(uint64_t *x) { uint64_t z = (*x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t *r)          /* xoshiro256** */
{
    uint64_t *s = r->s, res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}
static inline double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t rng_below(rng_t *r, uint64_t n) { return (uint64_t)(rng_unit(r) * (double)n); }
static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }

API int cbc_synth_packed(const cbc_synth_opts *so, const cbc_pack_opts *po, cbc_packed **out,
                         char **sam_out, size_t *sam_len, char **fasta_out, size_t *fasta_len,
                         char *errbuf, size_t errlen)
{
    if (!so || !out || so->read_len < 30 || so->read_len > CBC_MAX_READ_LEN || so->contig_len < (uint64_t)so->read_len + 64 ||
        so->contig_len > 0xf0000000ull) return CBC_E_ARG;
    static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
    const char *name = so->name ? so->name : "chr1";
    const uint32_t L = so->read_len;
    packer_t *S = (packer_t *)malloc(sizeof(packer_t));
    if (!S) return CBC_E_NOMEM;
    uint32_t *starts = NULL; char *sam = NULL; size_t samcap = 0, samn = 0;
    int rc = packer_init(S, po, errbuf, errlen);
    if (rc) goto done;
    cbc_packed *P = S->P;
    rng_t R; uint64_t sd = so->seed;
    for (int i = 0; i < 4; i++) R.s[i] = splitmix64(&sd);
    /* contig */
    if (grow((void **)&P->ref, &P->cap_ref, so->contig_len + CBC_REF_PAD, 1)) { rc = CBC_E_NOMEM; goto done; }
    for (uint64_t i = 0; i < so->contig_len; i += 32) {
        uint64_t w = rng_next(&R);
        for (int k = 0; k < 32 && i + k < so->contig_len; k++) { P->ref[i + k] = (uint8_t)ACGT[w & 3]; w >>= 2; }
    }
    memset(P->ref + so->contig_len, 0, CBC_REF_PAD);
    P->ref_bytes = so->contig_len + CBC_REF_PAD;
    if (grow32((void **)&P->contigs, &P->cap_contigs, 1, sizeof(cbc_contig_info))) { rc = CBC_E_NOMEM; goto done; }
    P->contigs[0].ref_off = 0; P->contigs[0].length = so->contig_len; P->contigs[0].name_off = 0; P->contigs[0].reserved = 0;
    S->n_fasta = 1;
    P->read_length = L;
    /* sorted start positions */
    starts = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(so->n_reads ? so->n_reads : 1));
    if (!starts) { rc = CBC_E_NOMEM; goto done; }
    uint64_t span = so->contig_len - L - 8;
    for (uint64_t i = 0; i < so->n_reads; i++) starts[i] = (uint32_t)rng_below(&R, span);
    qsort(starts, (size_t)so->n_reads, sizeof(uint32_t), cmp_u32);
    char seq[LINE_BUF], cigar[64], md[4 * LINE_BUF], qual[LINE_BUF];
    memset(qual, 'I', L); qual[L] = 0;
    for (uint64_t i = 0; i < so->n_reads; i++) {
        const uint8_t *ref = P->ref + starts[i];
        uint32_t flag = (rng_next(&R) & 1) ? 16u : 0u;
        int has_indel = rng_unit(&R) < so->indel_frac;
        uint32_t o = 0, k = 0; int is_ins = 0;
        if (has_indel) { k = 1 + (uint32_t)rng_below(&R, 3); o = 10 + (uint32_t)rng_below(&R, L - 20 - k); is_ins = (int)(rng_next(&R) & 1); }
        /* assemble read and the per-base reference it aligns to (0 = inserted base) */
        uint8_t rb[LINE_BUF]; uint32_t q = 0, rp = 0, nm = 0;
        if (!has_indel) { memcpy(seq, ref, L); memcpy(rb, ref, L); }
        else if (is_ins) {
            for (q = 0; q < o; q++) { seq[q] = (char)ref[rp]; rb[q] = ref[rp++]; }
            for (uint32_t c = 0; c < k; c++, q++) { seq[q] = ACGT[rng_next(&R) & 3]; rb[q] = 0; }
            for (; q < L; q++) { seq[q] = (char)ref[rp]; rb[q] = ref[rp++]; }
        } else {
            for (q = 0; q < o; q++) { seq[q] = (char)ref[rp]; rb[q] = ref[rp++]; }
            rp += k;
            for (; q < L; q++) { seq[q] = (char)ref[rp]; rb[q] = ref[rp++]; }
        }
        seq[L] = 0;
        /* substitutions on aligned bases */
        int nsub = 0;
        if (so->sub_rate > 0) {
            for (uint32_t b = 0; b < L; b++) {
                if (rb[b] && rng_unit(&R) < so->sub_rate) {
                    char old = seq[b]; char alt;
                    do { alt = ACGT[rng_next(&R) & 3]; } while (alt == old);
                    seq[b] = alt; nsub++;
                }
            }
        }
        /* CIGAR + MD */
        if (!has_indel) snprintf(cigar, sizeof cigar, "%uM", L);
        else if (is_ins) snprintf(cigar, sizeof cigar, "%uM%uI%uM", o, k, L - o - k);
        else snprintf(cigar, sizeof cigar, "%uM%uD%uM", o, k, L - o);
        {
            char *m = md; uint32_t run = 0;
            for (uint32_t b = 0; b < L; b++) {
                if (has_indel && !is_ins && b == o) {
                    m += sprintf(m, "%u^", run); run = 0;
                    for (uint32_t c = 0; c < k; c++) *m++ = (char)ref[o + c];
                    nm += k;
                }
                if (!rb[b]) { nm++; continue; }
                if ((uint8_t)seq[b] == rb[b]) run++;
                else { m += sprintf(m, "%u%c", run, (char)rb[b]); run = 0; nm++; }
            }
            m += sprintf(m, "%u", run);
            *m = 0;
        }
        rc = add_record(S, name, flag, (int32_t)starts[i] + 1, cigar, seq, md);
        if (rc) goto done;
        if (sam_out) {
            if (samn + 2 * LINE_BUF > samcap) {
                size_t nc = samcap ? samcap * 2 : (1u << 20);
                char *ns = (char *)realloc(sam, nc); if (!ns) { rc = CBC_E_NOMEM; goto done; }
                sam = ns; samcap = nc;
            }
            samn += (size_t)sprintf(sam + samn, "r%llu\t%u\t%s\t%u\t60\t%s\t*\t0\t0\t%s\t%s\tMD:Z:%s\tNM:i:%u\n",
                                    (unsigned long long)i, flag, name, starts[i] + 1, cigar, seq, qual, md, nm);
        }
    }
    rc = finish_pack(S);
    if (rc) goto done;
    if (sam_out) {
        if (!sam) { sam = (char *)malloc(1); if (!sam) { rc = CBC_E_NOMEM; goto done; } }
        *sam_out = sam; *sam_len = samn; sam = NULL;
    }
    if (fasta_out) {
        uint64_t nlines = (so->contig_len + 59) / 60;
        size_t cap = (size_t)(so->contig_len + nlines + strlen(name) + 8);
        char *fa = (char *)malloc(cap); if (!fa) { rc = CBC_E_NOMEM; goto done; }
        size_t n = (size_t)sprintf(fa, ">%s\n", name);
        for (uint64_t i = 0; i < so->contig_len; i += 60) {
            uint64_t c = so->contig_len - i < 60 ? so->contig_len - i : 60;
            memcpy(fa + n, P->ref + i, (size_t)c); n += (size_t)c; fa[n++] = '\n';
        }
        *fasta_out = fa; *fasta_len = n;
    }
done:
    free(starts); free(sam);
    packer_release(S);
    if (rc) { cbc_packed_free(S->P); *out = NULL; } else *out = S->P;
    free(S);
    return rc;
}


/* ================================= cfg5: synthetic long reads ============================== */
typedef struct {
    const cbc_synth_opts *so; const uint8_t *ref; const uint32_t *starts;
    uint64_t r0, r1;                       /* reads [r0, r1) */
    uint8_t *seq;                          /* final array: read r at r * read_len */
    uint32_t *tok; uint64_t n_tok, cap_tok; uint32_t *tok_of; uint16_t *flags;
    char *sam; size_t samn, samcap; int want_sam; const char *name;
    int rc;
} long_job;

static void *long_gen(void *arg)
{
    long_job *J = (long_job *)arg;
    static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
    const uint32_t L = J->so->read_len;
    char *cig = (char *)malloc(16 * (size_t)L + 64);
    if (!cig) { J->rc = CBC_E_NOMEM; return NULL; }
    for (uint64_t r = J->r0; r < J->r1 && !J->rc; r++) {
        rng_t R; uint64_t sd = J->so->seed ^ (0x9e3779b97f4a7c15ull * (r + 1));
        for (int i = 0; i < 4; i++) R.s[i] = splitmix64(&sd);
        const uint8_t *ref = J->ref + J->starts[r];
        uint8_t *out = J->seq + r * (uint64_t)L;
        if (grow((void **)&J->tok, &J->cap_tok, J->n_tok + 2 + 2 * (uint64_t)L + 8, sizeof(uint32_t))) { J->rc = CBC_E_NOMEM; break; }
        uint32_t *t = J->tok + J->n_tok; uint32_t nt = 2, n_cig = 0;
        J->tok_of[r - J->r0] = (uint32_t)J->n_tok;          /* relative to the job's pool; rebased when placed */
        J->flags[r - J->r0] = (rng_next(&R) & 1) ? 16 : 0;
        uint32_t i = 0, j = 0, run = 0;
#define EMIT(op_, n_) do { if ((n_)) { if (n_cig && (t[nt - 1] & 15u) == (op_)) t[nt - 1] += (uint32_t)(n_) << 4; else { t[nt++] = ((uint32_t)(n_) << 4) | (op_); n_cig++; } } } while (0)
        while (i < L) {
            double u = rng_unit(&R);
            if (u >= J->so->sub_rate || i == 0 || i + 1 >= L) { out[i++] = ref[j++]; run++; continue; }
            uint32_t kind = (uint32_t)(rng_next(&R) % 3);
            if (kind == 0) { char alt; do { alt = ACGT[rng_next(&R) & 3]; } while ((uint8_t)alt == ref[j]); out[i++] = (uint8_t)alt; j++; run++; }
            else if (kind == 1) { EMIT(CBC_OP_M, run); run = 0; out[i++] = (uint8_t)ACGT[rng_next(&R) & 3]; EMIT(CBC_OP_I, 1u); }
            else { EMIT(CBC_OP_M, run); run = 0; j++; EMIT(CBC_OP_D, 1u); }
        }
        EMIT(CBC_OP_M, run);
#undef EMIT
        t[0] = n_cig; t[1] = 0;
        J->n_tok += nt;
        if (J->want_sam) {
            size_t cl = 0;
            for (uint32_t k = 0; k < n_cig; k++) cl += (size_t)sprintf(cig + cl, "%u%c", t[2 + k] >> 4, "MIDS"[t[2 + k] & 15u]);
            size_t need = J->samn + 2 * (size_t)L + cl + 256;
            if (need > J->samcap) { size_t nc = J->samcap ? J->samcap * 2 : (1u << 20); while (nc < need) nc *= 2;
                                    char *ns = (char *)realloc(J->sam, nc); if (!ns) { J->rc = CBC_E_NOMEM; break; } J->sam = ns; J->samcap = nc; }
            J->samn += (size_t)sprintf(J->sam + J->samn, "r%llu\t%u\t%s\t%u\t60\t%s\t*\t0\t0\t", (unsigned long long)r, J->flags[r - J->r0], J->name, J->starts[r] + 1, cig);
            memcpy(J->sam + J->samn, out, L); J->samn += L; J->sam[J->samn++] = '\t';
            memset(J->sam + J->samn, 'I', L); J->samn += L; J->sam[J->samn++] = '\n';
        }
    }
    free(cig);
    return NULL;
}

API int cbc_synth_long(const cbc_synth_opts *so, const cbc_pack_opts *po, cbc_packed **out,
                       char **sam_out, size_t *sam_len, char **fasta_out, size_t *fasta_len, char *errbuf, size_t errlen)
{
    if (!so || !out || so->read_len < 64 || so->read_len > CBC_LONG_MAX_READ_LEN || so->contig_len < 2ull * so->read_len + 64 ||
        so->contig_len > 0xf0000000ull || so->sub_rate < 0 || so->sub_rate > 0.5) return CBC_E_ARG;
    static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
    const char *name = so->name ? so->name : "chrL";
    const uint32_t L = so->read_len;
    cbc_pack_opts o; if (po) o = *po; else cbc_pack_default_opts(&o);
    o.long_reads = 1;
    packer_t *S = (packer_t *)malloc(sizeof(packer_t));
    if (!S) return CBC_E_NOMEM;
    uint32_t *starts = NULL; long_job *jobs = NULL; pthread_t *th = NULL;
    int nt = o.n_threads ? (int)o.n_threads : online_cpus();
    if (nt > 64) nt = 64;
    if ((uint64_t)nt > so->n_reads) nt = so->n_reads ? (int)so->n_reads : 1;
    int rc = packer_init(S, &o, errbuf, errlen);
    if (rc) goto done;
    cbc_packed *P = S->P;
    rng_t R; uint64_t sd = so->seed;
    for (int i = 0; i < 4; i++) R.s[i] = splitmix64(&sd);
    if (grow((void **)&P->ref, &P->cap_ref, so->contig_len + CBC_REF_PAD, 1)) { rc = CBC_E_NOMEM; goto done; }
    for (uint64_t i = 0; i < so->contig_len; i += 32) {
        uint64_t w = rng_next(&R);
        for (int k = 0; k < 32 && i + k < so->contig_len; k++) { P->ref[i + k] = (uint8_t)ACGT[w & 3]; w >>= 2; }
    }
    memset(P->ref + so->contig_len, 0, CBC_REF_PAD);
    P->ref_bytes = so->contig_len + CBC_REF_PAD;
    if (grow32((void **)&P->contigs, &P->cap_contigs, 1, sizeof(cbc_contig_info))) { rc = CBC_E_NOMEM; goto done; }
    P->contigs[0].ref_off = 0; P->contigs[0].length = so->contig_len; P->contigs[0].name_off = 0; P->contigs[0].reserved = 0;
    S->n_fasta = 1; P->read_length = L > 256 ? 256 : L;
    starts = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(so->n_reads ? so->n_reads : 1));
    jobs = (long_job *)calloc((size_t)nt, sizeof(long_job)); th = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
    if (!starts || !jobs || !th) { rc = CBC_E_NOMEM; goto done; }
    {   /* every read may consume up to 2 L reference bases (deletions) */
        uint64_t span = so->contig_len - 2ull * L - 8;
        for (uint64_t i = 0; i < so->n_reads; i++) starts[i] = (uint32_t)rng_below(&R, span);
        qsort(starts, (size_t)so->n_reads, sizeof(uint32_t), cmp_u32);
    }
    if (grow((void **)&P->seq, &P->cap_seq, so->n_reads * (uint64_t)L + 8, 1)) { rc = CBC_E_NOMEM; goto done; }
    for (int t = 0; t < nt; t++) {
        long_job *J = &jobs[t];
        J->so = so; J->ref = P->ref; J->starts = starts; J->seq = P->seq; J->name = name; J->want_sam = sam_out != NULL;
        J->r0 = so->n_reads * (uint64_t)t / (uint64_t)nt; J->r1 = so->n_reads * (uint64_t)(t + 1) / (uint64_t)nt;
        J->tok_of = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(J->r1 - J->r0 + 1));
        J->flags = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(J->r1 - J->r0 + 1));
        if (!J->tok_of || !J->flags) { rc = CBC_E_NOMEM; goto done; }
    }
    {
        int started = 0;
        for (int t = 0; t < nt; t++) { if (pthread_create(&th[t], NULL, long_gen, &jobs[t]) != 0) break; started++; }
        for (int t = started; t < nt; t++) long_gen(&jobs[t]);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    }
    for (int t = 0; t < nt; t++) if (jobs[t].rc) { rc = jobs[t].rc; goto done; }
    {   /* block cutting over the reads in order; the SEQ bytes are already in place, the tokens are copied per read */
        rc = contig_open(S, name, strlen(name));
        if (rc) goto done;
        uint64_t ntok = 0;
        for (int t = 0; t < nt; t++) ntok += jobs[t].n_tok;
        if (grow((void **)&P->tok, &P->cap_tok, ntok + 1, sizeof(uint32_t))) { rc = CBC_E_NOMEM; goto done; }
        for (int t = 0; t < nt && !rc; t++) {
            long_job *J = &jobs[t];
            for (uint64_t r = J->r0; r < J->r1; r++) {
                const uint32_t *tk = J->tok + J->tok_of[r - J->r0];
                uint32_t n = 2 + (tk[0] & 0xffffu);
                uint64_t to = P->n_tok;
                rc = place_record(S, starts[r] + 1, J->flags[r - J->r0], L, 0, n);
                if (rc) break;
                memcpy(P->tok + to, tk, sizeof(uint32_t) * n);
            }
        }
        if (rc) goto done;
    }
    rc = finish_pack(S);
    if (rc) goto done;
    if (sam_out) {
        size_t total = 0;
        for (int t = 0; t < nt; t++) total += jobs[t].samn;
        char *sam = (char *)malloc(total + 1);
        if (!sam) { rc = CBC_E_NOMEM; goto done; }
        size_t n = 0;
        for (int t = 0; t < nt; t++) { memcpy(sam + n, jobs[t].sam, jobs[t].samn); n += jobs[t].samn; }
        *sam_out = sam; *sam_len = n;
    }
    if (fasta_out) {
        uint64_t nlines = (so->contig_len + 59) / 60;
        size_t cap = (size_t)(so->contig_len + nlines + strlen(name) + 8);
        char *fa = (char *)malloc(cap); if (!fa) { rc = CBC_E_NOMEM; goto done; }
        size_t n = (size_t)sprintf(fa, ">%s\n", name);
        for (uint64_t i = 0; i < so->contig_len; i += 60) {
            uint64_t c = so->contig_len - i < 60 ? so->contig_len - i : 60;
            memcpy(fa + n, P->ref + i, (size_t)c); n += (size_t)c; fa[n++] = '\n';
        }
        *fasta_out = fa; *fasta_len = n;
    }
done:
    if (jobs) for (int t = 0; t < nt; t++) { free(jobs[t].tok); free(jobs[t].tok_of); free(jobs[t].flags); free(jobs[t].sam); }
    free(jobs); free(th); free(starts);
    packer_release(S);
    if (rc) { cbc_packed_free(S->P); *out = NULL; } else *out = S->P;
    free(S);
    return rc;
}

/* ================================= block container ======================================== */
/* layout (little-endian), version 2:
 *   u32 magic "CBCB", u32 version, u32 read_length, u32 n_contigs, u32 n_blocks, u32 names_bytes,
 *   u32 cap_pos, u32 cap_var, u32 max_read_len                                   (36 bytes)
 *   names blob, padded to 4
 *   n_contigs x { u32 name_off, u32 reserved, u64 length }
 *   n_blocks  x { u32 contig, u32 n_reads, u64 window_start, u64 payload_off, u32 payload_bytes, u32 reserved }
 *   payloads, concatenated in block order
 */
#define CBC_CONTAINER_HDR 36ull
static uint64_t container_header_bytes(const cbc_packed *p)
{
    return CBC_CONTAINER_HDR + ((p->names_bytes + 3u) & ~3u) + 16ull * p->n_contigs + 32ull * p->n_blocks;
}
API int64_t cbc_container_size(const cbc_packed *p, const uint64_t *out_offsets)
{
    if (!p || !out_offsets) return CBC_E_ARG;
    return (int64_t)(container_header_bytes(p) + out_offsets[p->n_blocks]);
}
static void w32(uint8_t **d, uint32_t v) { memcpy(*d, &v, 4); *d += 4; }
static void w64(uint8_t **d, uint64_t v) { memcpy(*d, &v, 8); *d += 8; }
API int64_t cbc_container_write(const cbc_packed *p, const uint8_t *payloads, const uint64_t *out_offsets,
                                uint8_t *dst, uint64_t dst_cap)
{
    if (!p || !payloads || !out_offsets || !dst) return CBC_E_ARG;
    uint64_t hb = container_header_bytes(p), total = hb + out_offsets[p->n_blocks];
    if (dst_cap < total) return CBC_E_ARG;
    uint8_t *d = dst;
    w32(&d, CBC_CONTAINER_MAGIC); w32(&d, p->whole_file == 2 ? CBC_CONTAINER_VERSION_LONG : CBC_CONTAINER_VERSION); w32(&d, p->read_length);
    w32(&d, p->n_contigs); w32(&d, p->n_blocks); w32(&d, p->names_bytes);
    w32(&d, p->caps.cap_pos); w32(&d, p->caps.cap_var); w32(&d, p->max_read_len);
    uint32_t nb = (p->names_bytes + 3u) & ~3u;
    memset(d, 0, nb); memcpy(d, p->names, p->names_bytes); d += nb;
    for (uint32_t i = 0; i < p->n_contigs; i++) { w32(&d, p->contigs[i].name_off); w32(&d, 0); w64(&d, p->contigs[i].length); }
    for (uint32_t b = 0; b < p->n_blocks; b++) {
        w32(&d, p->info[b].contig); w32(&d, p->info[b].n_reads); w64(&d, p->info[b].window_start);
        w64(&d, out_offsets[b]); w32(&d, (uint32_t)(out_offsets[b + 1] - out_offsets[b]));
        w32(&d, p->whole_file == 2 ? (uint32_t)p->info[b].n_bases : 0u);     /* version 3: the decoder writes the bases compactly */
    }
    memcpy(d, payloads, (size_t)out_offsets[p->n_blocks]);
    return (int64_t)total;
}



/* ================================= after the device tokeniser ============ */
API uint64_t cbc_sam_body_offset(const char *sam, size_t len)
{
    size_t off = 0;
    if (!sam) return 0;
    (void)header_read_length(sam, len, &off, 0);
    return off;
}

typedef cbc_tok_record_summary dev_summary;             /* the one public struct (include/cbc_gpu.h) */
_Static_assert(sizeof(cbc_tok_record_summary) == 16, "record summary crosses the device/host seam as 16 bytes");

API int cbc_pack_from_device_tokens(const char *sam, size_t sam_len, const char *fasta, size_t fasta_len, const cbc_pack_opts *opts,
                                    const void *summaries, const uint8_t *rname_change, const uint64_t *change_name_off,
                                    const uint32_t *change_name_len, uint64_t n_recs, uint64_t n_unmapped,
                                    uint8_t *seq, uint64_t seq_bytes, uint32_t *tok, uint64_t n_tok,
                                    cbc_packed **out, char *errbuf, size_t errlen)
{
    if (!sam || !fasta || !out || (n_recs && (!summaries || !rname_change || !change_name_off || !change_name_len))) return CBC_E_ARG;
    packer_t *S = (packer_t *)malloc(sizeof(packer_t));
    if (!S) return CBC_E_NOMEM;
    int rc = packer_init(S, opts, errbuf, errlen);
    if (rc) goto done;
    if (S->o.whole_file || S->o.long_reads) { rc = fail(S, CBC_E_ARG, "the device tokeniser feeds block mode only%s%lld", "", 0); goto done; }
    rc = load_reference(S, fasta, fasta_len, S->o.n_threads);
    if (rc) goto done;
    cbc_packed *P = S->P;
    {
        size_t off = 0;
        P->read_length = header_read_length(sam, sam_len, &off, (int)S->o.var_length);
        if (P->read_length < 1 || P->read_length > 256) { rc = fail(S, CBC_E_INPUT, "header read length %s%lld outside 1..256", "", P->read_length); goto done; }
    }
    if (grow((void **)&P->recs, &P->cap_recs, n_recs + 1, sizeof(cbc_read_rec))) { rc = CBC_E_NOMEM; goto done; }
    {
        const dev_summary *sm = (const dev_summary *)summaries;
        uint64_t k = 0;
        for (uint64_t r = 0; r < n_recs && !rc; r++) {
            if (rname_change[r]) {
                if (change_name_off[k] + change_name_len[k] > sam_len) { rc = fail(S, CBC_E_ARG, "contig name outside the text%s%lld", "", 0); break; }
                rc = contig_open(S, sam + change_name_off[k], change_name_len[k]); k++;
                if (rc) break;
            }
            const cbc_contig_info *ctg = &P->contigs[S->contig];
            const uint32_t nt = sm[r].nt_ev & 0xffffu, ev = sm[r].nt_ev >> 16;
            if ((uint64_t)sm[r].pos - 1 + sm[r].rl > ctg->length + CBC_REF_PAD - 8) {
                rc = fail(S, CBC_E_INPUT, "record at %s POS %lld runs past the contig end + pad", (const char *)P->names + ctg->name_off, sm[r].pos); break; }
            if (ev + 1 > S->o.max_cap_var) { rc = fail(S, CBC_E_INPUT, "record at %s:%lld has more edits than max_cap_var", (const char *)P->names + ctg->name_off, sm[r].pos); break; }
            rc = place_record(S, sm[r].pos, sm[r].flag, sm[r].rl, ev, nt);
        }
        if (rc) goto done;
    }
    if (P->seq_bytes != seq_bytes || P->n_tok != n_tok) { rc = fail(S, CBC_E_ARG, "summaries and array sizes disagree%s%lld", "", 0); goto done; }
    P->n_skipped_unmapped = n_unmapped;
    close_block(S);
    if (seq) { free(P->seq); P->seq = seq; P->cap_seq = seq_bytes + 8; memset(P->seq + seq_bytes, 0, 8); }
    if (tok) { free(P->tok); P->tok = tok; P->cap_tok = n_tok; }
    P->seq_bytes += 8;                                  /* the 8 readable pad bytes (the device array has them too) */
    if (P->caps.cap_pos < 64) P->caps.cap_pos = 64;
    if (P->caps.cap_var < 64) P->caps.cap_var = 64;
    P->caps.cap_pos = (P->caps.cap_pos + 63u) & ~63u;
    P->caps.cap_var = (P->caps.cap_var + 63u) & ~63u;
    if (P->n_tok == 0 && !P->tok) { if (grow((void **)&P->tok, &P->cap_tok, 1, sizeof(uint32_t))) { rc = CBC_E_NOMEM; goto done; } P->tok[0] = 0; }
    P->whole_file = 0;
done:
    packer_release(S);
    if (rc) { if (seq && S->P && S->P->seq == seq) S->P->seq = NULL; if (tok && S->P && S->P->tok == tok) S->P->tok = NULL; cbc_packed_free(S->P); *out = NULL; }
    else *out = S->P;
    free(S);
    return rc;
}

/* ================================= 2-bit transport ======================= */
typedef struct { const uint8_t *b; uint32_t *codes; uint64_t i0, i1; cbc_2bit_run *runs; uint64_t n_runs, cap_runs; int rc; } twobit_job;

static inline uint32_t base_code(uint8_t c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; }

static void *twobit_run(void *arg)
{
    twobit_job *J = (twobit_job *)arg;                     /* [i0, i1) is a multiple of 16 bases except at the very end */
    const uint8_t *b = J->b;
    for (uint64_t i = J->i0; i < J->i1; i += 16) {
        uint32_t w = 0;
        const uint64_t e = i + 16 < J->i1 ? i + 16 : J->i1;
        for (uint64_t k = i; k < e; k++) {
            uint32_t c = base_code(b[k]);
            if (c < 4u) w |= c << (2u * (uint32_t)(k & 15u));
            else {                                          /* an exception: extend the current run or open one */
                cbc_2bit_run *last = J->n_runs ? &J->runs[J->n_runs - 1] : NULL;
                if (last && last->start + last->length == k && last->byte == b[k] && last->length < 0xffffffffu) last->length++;
                else {
                    if (grow((void **)&J->runs, &J->cap_runs, J->n_runs + 1, sizeof(cbc_2bit_run))) { J->rc = CBC_E_NOMEM; return NULL; }
                    J->runs[J->n_runs].start = k; J->runs[J->n_runs].length = 1; J->runs[J->n_runs].byte = b[k]; J->n_runs++;
                }
            }
        }
        J->codes[i >> 4] = w;
    }
    return NULL;
}

API void cbc_2bit_free(cbc_2bit *p) { if (p) { free(p->codes); free(p->runs); free(p); } }

API int cbc_2bit_pack(const uint8_t *bases, uint64_t n, uint32_t n_threads, cbc_2bit **out)
{
    if (!bases || !out) return CBC_E_ARG;
    *out = NULL;
    int nt = n_threads ? (int)n_threads : online_cpus();
    if (nt > 64) nt = 64;
    if (n < ((uint64_t)1 << 22)) nt = 1;
    cbc_2bit *P = (cbc_2bit *)calloc(1, sizeof(cbc_2bit));
    twobit_job *jobs = (twobit_job *)calloc((size_t)nt, sizeof(twobit_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
    int rc = 0;
    if (!P || !jobs || !th) { rc = CBC_E_NOMEM; goto done; }
    P->n_bases = n;
    P->codes = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)((n + 15) / 16 + 1));
    if (!P->codes) { rc = CBC_E_NOMEM; goto done; }
    {
        const uint64_t words = (n + 15) / 16;
        int started = 0;
        for (int t = 0; t < nt; t++) {
            jobs[t].b = bases; jobs[t].codes = P->codes;
            jobs[t].i0 = words * (uint64_t)t / (uint64_t)nt * 16; jobs[t].i1 = (t == nt - 1) ? n : words * (uint64_t)(t + 1) / (uint64_t)nt * 16;
            if (jobs[t].i1 > n) jobs[t].i1 = n;
        }
        for (int t = 0; t < nt; t++) { if (pthread_create(&th[t], NULL, twobit_run, &jobs[t]) != 0) break; started++; }
        for (int t = started; t < nt; t++) twobit_run(&jobs[t]);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    }
    {   /* concatenate the threads' runs, merging a run that continues across a boundary */
        uint64_t total = 0;
        for (int t = 0; t < nt; t++) { if (jobs[t].rc) rc = jobs[t].rc; total += jobs[t].n_runs; }
        if (rc) goto done;
        P->runs = (cbc_2bit_run *)malloc(sizeof(cbc_2bit_run) * (size_t)(total ? total : 1));
        if (!P->runs) { rc = CBC_E_NOMEM; goto done; }
        for (int t = 0; t < nt; t++)
            for (uint64_t k = 0; k < jobs[t].n_runs; k++) {
                cbc_2bit_run *r = &jobs[t].runs[k], *last = P->n_runs ? &P->runs[P->n_runs - 1] : NULL;
                if (last && last->start + last->length == r->start && last->byte == r->byte && (uint64_t)last->length + r->length <= 0xffffffffull) last->length += r->length;
                else P->runs[P->n_runs++] = *r;
            }
    }
    *out = P; P = NULL;
done:
    if (jobs) for (int t = 0; t < nt; t++) free(jobs[t].runs);
    free(jobs); free(th); cbc_2bit_free(P);
    return rc;
}

API int cbc_2bit_unpack(const cbc_2bit *p, uint8_t *bases)
{
    if (!p || !bases || (p->n_bases && !p->codes) || (p->n_runs && !p->runs)) return CBC_E_ARG;
    static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
    for (uint64_t i = 0; i < p->n_bases; i++) bases[i] = (uint8_t)ACGT[(p->codes[i >> 4] >> (2u * (uint32_t)(i & 15u))) & 3u];
    for (uint64_t k = 0; k < p->n_runs; k++) {
        /* no `start + length`: that sum wraps in u64 for a crafted run (round-2 advisor finding) */
        if (p->runs[k].start > p->n_bases || p->runs[k].length > p->n_bases - p->runs[k].start) return CBC_E_INPUT;
        memset(bases + p->runs[k].start, (int)p->runs[k].byte, p->runs[k].length);
    }
    return 0;
}

/* ================================= sharding =============================================== */
API uint64_t cbc_checksum64(const uint8_t *bytes, uint64_t n)
{
    uint64_t acc = 0;
    if (!bytes) return 0;
    for (uint64_t i = 0; i < n; i++) acc += CBC_CHECKSUM_TERM(i, bytes[i]);
    return acc;
}

API int cbc_assign_contigs(const cbc_packed *p, uint32_t n_parts, uint32_t *part_of_contig)
{
    if (!p || !part_of_contig || n_parts == 0) return CBC_E_ARG;
    uint32_t nc = p->n_contigs;
    uint64_t *reads = (uint64_t *)calloc(nc ? nc : 1, sizeof(uint64_t)), *load = (uint64_t *)calloc(n_parts, sizeof(uint64_t));
    uint8_t *done = (uint8_t *)calloc(nc ? nc : 1, 1);
    if (!reads || !load || !done) { free(reads); free(load); free(done); return CBC_E_NOMEM; }
    for (uint32_t b = 0; b < p->n_blocks; b++) reads[p->info[b].contig] += p->info[b].n_reads;
    for (uint32_t k = 0; k < nc; k++) {                     /* largest remaining contig -> least loaded part */
        uint32_t best = 0; int have = 0;
        for (uint32_t c = 0; c < nc; c++) if (!done[c] && (!have || reads[c] > reads[best])) { best = c; have = 1; }
        uint32_t part = 0;
        for (uint32_t q = 1; q < n_parts; q++) if (load[q] < load[part]) part = q;
        part_of_contig[best] = part; load[part] += reads[best]; done[best] = 1;
    }
    free(reads); free(load); free(done);
    return 0;
}

/* ================================= reference alone ======================================== */
API void cbc_reference_free(cbc_reference *r)
{
    if (!r) return;
    free(r->bases); free(r->contig_off); free(r->contig_len); free(r);
}
API int cbc_reference_load(const char *fasta, size_t fasta_len, uint32_t n_threads, cbc_reference **out, char *errbuf, size_t errlen)
{
    if (!fasta || !out) return CBC_E_ARG;
    *out = NULL;
    packer_t *S = (packer_t *)calloc(1, sizeof(packer_t));
    cbc_reference *r = (cbc_reference *)calloc(1, sizeof(cbc_reference));
    if (!S || !r) { free(S); free(r); return CBC_E_NOMEM; }
    S->err = errbuf; S->errlen = errlen;
    if (errbuf && errlen) errbuf[0] = 0;
    int rc = CBC_E_NOMEM;
    S->P = (cbc_packed *)calloc(1, sizeof(cbc_packed));
    if (S->P) rc = load_reference(S, fasta, fasta_len, n_threads);
    if (!rc && S->n_fasta == 0) rc = fail(S, CBC_E_INPUT, "the FASTA holds no sequence%s%lld", "", 0);
    if (!rc) {
        r->contig_off = (uint64_t *)calloc(S->n_fasta, sizeof(uint64_t));
        r->contig_len = (uint64_t *)calloc(S->n_fasta, sizeof(uint64_t));
        if (!r->contig_off || !r->contig_len) rc = CBC_E_NOMEM;
    }
    if (!rc) {
        for (uint32_t i = 0; i < S->n_fasta; i++) { r->contig_off[i] = S->P->contigs[i].ref_off; r->contig_len[i] = S->P->contigs[i].length; }
        r->n_contigs = S->n_fasta; r->bases = S->P->ref; r->n_bytes = S->P->ref_bytes; S->P->ref = NULL;
        *out = r;
    } else cbc_reference_free(r);
    if (S->P) cbc_packed_free(S->P);
    free(S);
    return rc;
}

/* ================================= unpack side ============================================ */
static uint32_t r32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t r64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

API void cbc_unpack_plan_free(cbc_unpack_plan *u)
{
    if (!u) return;
    free(u->blocks); free(u->window_start); free(u->ref); free(u);
}

/* Parse a container, load the FASTA the way the packer does (contigs in file order, upper-cased,
 * padded) and lay out the decode launch: one cbc_dec_block_desc per block. */
API int cbc_unpack_plan_create(const uint8_t *blob, uint64_t len, const char *fasta, size_t fasta_len,
                               cbc_unpack_plan **out, char *errbuf, size_t errlen)
{
    if (!blob || !fasta || !out) return CBC_E_ARG;
    *out = NULL;
    packer_t *S = (packer_t *)calloc(1, sizeof(packer_t));
    cbc_unpack_plan *u = (cbc_unpack_plan *)calloc(1, sizeof(cbc_unpack_plan));
    if (!S || !u) { free(S); free(u); return CBC_E_NOMEM; }
    S->err = errbuf; S->errlen = errlen;
    if (errbuf && errlen) errbuf[0] = 0;
    int rc = 0;
    S->P = (cbc_packed *)calloc(1, sizeof(cbc_packed));
    if (!S->P) { rc = CBC_E_NOMEM; goto fail; }
    if (len < CBC_CONTAINER_HDR || r32(blob) != CBC_CONTAINER_MAGIC) { rc = fail(S, CBC_E_INPUT, "not a cbc block container%s%lld", "", 0); goto fail; }
    const int is_long = r32(blob + 4) == CBC_CONTAINER_VERSION_LONG;
    if (r32(blob + 4) != CBC_CONTAINER_VERSION && !is_long) { rc = fail(S, CBC_E_INPUT, "unsupported container version %s%lld", "", r32(blob + 4)); goto fail; }
    {
        uint32_t L0 = r32(blob + 8), nc = r32(blob + 12), nb = r32(blob + 16), nbytes = r32(blob + 20);
        uint32_t cap_pos = r32(blob + 24), cap_var = r32(blob + 28), max_rl = r32(blob + 32);
        uint64_t names_pad = ((uint64_t)nbytes + 3u) & ~3ull;
        uint64_t hdr = CBC_CONTAINER_HDR + names_pad + 16ull * nc + 32ull * nb;
        if (hdr > len || L0 < 1 || (L0 > 256 && !is_long) || max_rl < 1 || max_rl > (is_long ? CBC_LONG_MAX_READ_LEN : CBC_MAX_READ_LEN) || cap_pos < 2 || cap_pos > 8192 ||
            cap_var < 1 || cap_var > 32768) { rc = fail(S, CBC_E_INPUT, "corrupt container header%s%lld", "", 0); goto fail; }
        const uint8_t *ctab = blob + CBC_CONTAINER_HDR + names_pad, *btab = ctab + 16ull * nc, *pay = btab + 32ull * nb;
        uint64_t pay_bytes = len - hdr;
        rc = load_reference(S, fasta, fasta_len, 0u);
        if (rc) goto fail;
        if (S->n_fasta < nc) { rc = fail(S, CBC_E_INPUT, "the FASTA has fewer records than the container has contigs%s%lld", "", 0); goto fail; }
        for (uint32_t i = 0; i < nc; i++)
            if (r64(ctab + 16ull * i + 8) != S->P->contigs[i].length) {
                rc = fail(S, CBC_E_INPUT, "contig %s#%lld has a different length in the FASTA than at compression time", "", (long long)i + 1); goto fail; }
        u->blocks = (cbc_dec_block_desc *)calloc(nb ? nb : 1, sizeof(cbc_dec_block_desc));
        u->window_start = (uint64_t *)calloc(nb ? nb : 1, sizeof(uint64_t));
        if (!u->blocks || !u->window_start) { rc = CBC_E_NOMEM; goto fail; }
        uint32_t stride = is_long ? 0u : ((max_rl + 3u) & ~3u);
        uint64_t nrec = 0, nbases = 0;
        for (uint32_t b = 0; b < nb; b++) {
            const uint8_t *e = btab + 32ull * b;
            uint32_t contig = r32(e), nreads = r32(e + 4), pbytes = r32(e + 24), blk_bases = r32(e + 28);
            uint64_t w0 = r64(e + 8), poff = r64(e + 16);
            /* no sums of file-supplied values: poff + pbytes could wrap */
            if (contig >= nc || poff > pay_bytes || pbytes > pay_bytes - poff || w0 >= S->P->contigs[contig].length + 1 ||
                nreads > CBC_MAX_BLOCK_READS || (is_long && (uint64_t)blk_bases > (uint64_t)nreads * CBC_LONG_MAX_READ_LEN)) {
                rc = fail(S, CBC_E_INPUT, "corrupt block index entry %s%lld", "", b); goto fail; }
            cbc_dec_block_desc *d = &u->blocks[b];
            d->in_off = poff; d->in_bytes = pbytes; d->ref_off = S->P->contigs[contig].ref_off + w0;
            d->rec_base = nrec; d->seq_base = is_long ? nbases : nrec * stride; d->n_reads = nreads; d->read_length = L0; d->seq_stride = stride;
            d->reserved[0] = is_long ? blk_bases : 0u;
            u->window_start[b] = w0;
            nrec += nreads; nbases += ((uint64_t)blk_bases + 7u) & ~7ull;
        }
        u->n_blocks = nb; u->payloads = pay; u->payload_bytes = pay_bytes; u->caps.cap_pos = cap_pos; u->caps.cap_var = cap_var;
        u->read_length = L0; u->seq_stride = stride; u->n_recs = nrec;
        u->long_reads = is_long ? 1u : 0u; u->max_read_len = max_rl; u->seq_total = is_long ? nbases + 8 : nrec * stride + 8;
        u->ref = S->P->ref; u->ref_bytes = S->P->ref_bytes; S->P->ref = NULL;
    }
    cbc_packed_free(S->P); free(S);
    *out = u;
    return 0;
fail:
    if (S->P) cbc_packed_free(S->P);
    free(S); cbc_unpack_plan_free(u);
    return rc;
}

/* One reconstructed read per line (print_line, src/compression.c:16-40). */
API int64_t cbc_unpack_write_text(const cbc_unpack_plan *u, const cbc_read_rec *recs, const uint8_t *seq,
                                  char *dst, uint64_t cap)
{
    if (!u || !recs || !seq || !dst) return CBC_E_ARG;
    uint64_t n = 0;
    if (u->long_reads) {                                 /* bases are compact: block base + the record's offset */
        for (uint32_t b = 0; b < u->n_blocks; b++) {
            const cbc_dec_block_desc *d = &u->blocks[b];
            for (uint32_t k = 0; k < d->n_reads; k++) {
                const cbc_read_rec *r = &recs[d->rec_base + k];
                if ((uint64_t)r->seq_off + r->rlen > d->reserved[0] || n + r->rlen + 1 > cap) return CBC_E_ARG;
                memcpy(dst + n, seq + d->seq_base + r->seq_off, r->rlen); n += r->rlen;
                dst[n++] = '\n';
            }
        }
        return (int64_t)n;
    }
    for (uint64_t r = 0; r < u->n_recs; r++) {
        uint32_t rl = recs[r].rlen;
        if (rl > u->seq_stride || n + rl + 1 > cap) return CBC_E_ARG;
        memcpy(dst + n, seq + r * u->seq_stride, rl); n += rl;
        dst[n++] = '\n';
    }
    return (int64_t)n;
}
