/*
 * cbc_tokenise.h -- SAM text -> packed records ON THE DEVICE, included by cbc_gpu.hip.
 *
 * The text goes up once (2.4 bytes per base), the bases and tokens are produced where the encoder reads them, and
 * what comes back to the host is 20 bytes per record for the serial part of packing (block cutting, contig
 * numbering: cbc_pack_from_device_tokens in libcbc_host).  Kernels, all HBM-bound byte streaming:
 *   cbc_tok_count_kernel     newlines per 4096-byte tile (1 byte per lane, ballot + popcount)
 *   cbc_tok_lines_kernel     line starts in file order (tile base + ballot prefix)
 *   cbc_tok_parse_kernel     one thread per line: columns, FLAG / POS, CIGAR + MD token COUNT (cbc_tok_core.h)
 *   cbc_tok_emit_kernel      one thread per mapped record: token words, record summary
 *   cbc_tok_seq_kernel       one wavefront per 16 records: the SEQ bytes, coalesced 64 bytes per step
 *   cbc_tok_names_kernel     RNAME of record r differs from record r - 1 (contig change flags)
 * plus a three-kernel exclusive scan (u32 values -> u64 offsets).  Rules and limits: cbc_tok_core.h.
 */
#ifndef CBC_TOKENISE_H
#define CBC_TOKENISE_H

#include "cbc_tok_core.h"

#define CBC_TOK_TILE 4096u

__global__ void __launch_bounds__(64)
cbc_tok_count_kernel(const uint8_t *__restrict__ sam, uint64_t len, uint32_t *__restrict__ tile_count)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * CBC_TOK_TILE;
    uint32_t n = 0;
    for (uint32_t k = 0; k < CBC_TOK_TILE; k += 64) {
        const uint64_t i = t0 + k + threadIdx.x;
        n += (uint32_t)__popcll(__ballot(i < len && sam[i] == '\n'));
    }
    if (threadIdx.x == 0) tile_count[blockIdx.x] = n;
}

/* line_start[k] = offset of line k; line_start[n_lines] = len (a last line without '\n' is a line too) */
__global__ void __launch_bounds__(64)
cbc_tok_lines_kernel(const uint8_t *__restrict__ sam, uint64_t len, const uint64_t *__restrict__ tile_base, uint64_t *__restrict__ line_start)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * CBC_TOK_TILE;
    uint64_t run = tile_base[blockIdx.x];
    for (uint32_t k = 0; k < CBC_TOK_TILE; k += 64) {
        const uint64_t i = t0 + k + threadIdx.x;
        const bool nl = i < len && sam[i] == '\n';
        const uint64_t m = __ballot(nl);
        if (nl && i + 1 < len) line_start[run + 1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = i + 1;
        run += (uint64_t)__popcll(m);
    }
}

struct cbc_tok_perline {            /* what pass 1 leaves per line: the column split is done once */
    uint64_t rname; uint32_t rname_len; uint32_t status;
    uint32_t rl, nt;                /* of a mapped, well-formed record; 0 otherwise */
    cbc_tok_line L;                 /* offsets of the columns the later passes read (CIGAR, SEQ, MD) */
};

/* pass 1a: the column split of every line (the parse below may look at earlier lines' splits) */
__global__ void __launch_bounds__(256)
cbc_tok_split_kernel(const uint8_t *__restrict__ sam, const uint64_t *__restrict__ line_start, uint64_t n_lines, uint64_t body_off,
                     cbc_tok_perline *__restrict__ pl)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_lines) return;
    cbc_tok_perline o;
    memset(&o, 0, sizeof o);
    const uint64_t b = line_start[k], e = line_start[k + 1];
    if (b < body_off) o.status = CBC_TOK_SKIP;                        /* '@' header lines lie before the body */
    else { cbc_tok_split(sam, b, e, &o.L); o.status = o.L.status; }
    pl[k] = o;
}

__global__ void __launch_bounds__(256)
cbc_tok_parse_kernel(const uint8_t *__restrict__ sam, const uint64_t *__restrict__ line_start, uint64_t n_lines, uint64_t body_off,
                     cbc_tok_perline *pl /* in: the splits; out: + status, counts, the inherited MD */, uint32_t *__restrict__ is_rec,
                     uint32_t *__restrict__ v_rl, uint32_t *__restrict__ v_nt)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_lines) return;
    cbc_tok_line L = pl[k].L;                                         /* cbc_tok_split_kernel's */
    const uint64_t b = line_start[k];
    uint32_t status = pl[k].status, nt = 0, ev = 0;
    if (status == CBC_TOK_OK && !L.has_md) {                          /* inherits the text of the nearest earlier line that has one */
        uint64_t md = 0; uint32_t md_len = 0;
        status = cbc_tok_md_source(k, [&](uint64_t j) -> const cbc_tok_line * { return line_start[j] < body_off ? (const cbc_tok_line *)0 : &pl[j].L; }, &md, &md_len);
        L.md = md; L.md_len = md_len;
    }
    if (status == CBC_TOK_OK) status = cbc_tok_record(sam, &L, NULL, &nt, &ev);
    cbc_tok_perline o;
    o.rname = status == CBC_TOK_OK ? L.rname : 0; o.rname_len = status == CBC_TOK_OK ? L.rname_len : 0; o.status = status;
    o.rl = status == CBC_TOK_OK ? L.seq_len : 0; o.nt = status == CBC_TOK_OK ? nt : 0;
    if (b >= body_off) o.L = L; else memset(&o.L, 0, sizeof o.L);
    pl[k] = o;
    is_rec[k] = status == CBC_TOK_OK ? 1u : 0u; v_rl[k] = o.rl; v_nt[k] = o.nt;
}

__global__ void __launch_bounds__(256)
cbc_tok_emit_kernel(const uint8_t *__restrict__ sam, const uint64_t *__restrict__ line_start, uint64_t n_lines,
                    const cbc_tok_perline *__restrict__ pl, const uint64_t *__restrict__ rec_of, const uint64_t *__restrict__ tok_of,
                    uint32_t *__restrict__ tok, cbc_tok_summary *__restrict__ sum)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_lines || pl[k].status != CBC_TOK_OK) return;
    const cbc_tok_line L = pl[k].L;
    uint32_t nt = 0, ev = 0;
    (void)cbc_tok_record(sam, &L, tok + tok_of[k], &nt, &ev);
    cbc_tok_summary s;
    s.pos = (uint32_t)L.pos; s.flag = (uint16_t)L.flag; s.rl = (uint16_t)L.seq_len; s.nt_ev = nt | (ev << 16); s.line = (uint32_t)k;
    sum[rec_of[k]] = s;
}

/* SEQ bytes of record r to seq[seq_of[line]]: one wavefront per 16 records, 64 bytes per step */
__global__ void __launch_bounds__(64)
cbc_tok_seq_kernel(const uint8_t *__restrict__ sam, const cbc_tok_perline *__restrict__ pl, const cbc_tok_summary *__restrict__ sum,
                   uint64_t n_recs, const uint64_t *__restrict__ seq_of, uint8_t *__restrict__ seq)
{
    for (uint32_t q = 0; q < 16; q++) {
        const uint64_t r = (uint64_t)blockIdx.x * 16 + q;
        if (r >= n_recs) return;
        const uint32_t line = sum[r].line, rl = sum[r].rl;
        const uint64_t src = pl[line].L.seq, dst = seq_of[line];
        for (uint32_t i = threadIdx.x; i < rl; i += 64) seq[dst + i] = sam[src + i];
    }
}

/* chg[r] = 1 when record r's RNAME differs from record r - 1's (strcmp in compress_rname, id_compression.c:46) */
__global__ void __launch_bounds__(256)
cbc_tok_names_kernel(const uint8_t *__restrict__ sam, const cbc_tok_perline *__restrict__ pl, const cbc_tok_summary *__restrict__ sum,
                     uint64_t n_recs, uint8_t *__restrict__ chg)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_recs) return;
    uint8_t c = 1;
    if (r > 0) {
        const cbc_tok_perline a = pl[sum[r].line], b = pl[sum[r - 1].line];
        if (a.rname_len == b.rname_len) {
            c = 0;
            for (uint32_t i = 0; i < a.rname_len; i++) if (sam[a.rname + i] != sam[b.rname + i]) { c = 1; break; }
        }
    }
    chg[r] = c;
}

/* ---- exclusive scan, u32 values -> u64 offsets: per-block scan of 1024 values, scan of the block totals, add ---- */
__global__ void __launch_bounds__(256)
cbc_scan_block_kernel(const uint32_t *__restrict__ v, uint64_t n, uint64_t *__restrict__ out, uint64_t *__restrict__ block_total)
{
    __shared__ uint32_t part[256];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t x[4], s = 0;
    for (int k = 0; k < 4; k++) { x[k] = base + k < n ? v[base + k] : 0u; s += x[k]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t t = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t run = part[threadIdx.x] - s;
    for (int k = 0; k < 4; k++) { if (base + k < n) out[base + k] = run; run += x[k]; }
    if (threadIdx.x == 255) block_total[blockIdx.x] = part[255];
}
__global__ void __launch_bounds__(1024)
cbc_scan_totals_kernel(uint64_t *__restrict__ block_total, uint64_t n_blocks, uint64_t *__restrict__ grand)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint64_t per = (n_blocks + 1023) / 1024, b0 = t * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    uint64_t s = 0;
    for (uint64_t b = b0; b < b1; b++) s += block_total[b];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = part[t] - s;
    for (uint64_t b = b0; b < b1; b++) { uint64_t x = block_total[b]; block_total[b] = run; run += x; }
    if (t == 1023) *grand = part[1023];
}
__global__ void __launch_bounds__(256)
cbc_scan_add_kernel(uint64_t *__restrict__ out, uint64_t n, const uint64_t *__restrict__ block_base)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += block_base[i >> 10];
}

#endif /* CBC_TOKENISE_H */
