/*
 * cbc_cli_unpack.c -- `cbc -d / -x`: container + FASTA -> one reconstructed read per line, decoded
 * on the GPU (decompress(), src/compression.c:173-216; print_line :16-40).
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "../../include/cbc_host.h"

static char *slurp2(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cbc: cannot open %s: %s\n", path, strerror(errno)); return NULL; }
    fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
    if (n < 0) { fclose(f); return NULL; }
    char *buf = (char *)malloc((size_t)n + 1);
    if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); fprintf(stderr, "cbc: cannot read %s\n", path); return NULL; }
    fclose(f); buf[n] = 0; *len = (size_t)n;
    return buf;
}

/* a file that does not start with the container magic is the reference's own format: one stream (decompress(),
 * src/compression.c:173-216).  The record count is not known in advance: the buffers grow until the kernel stops
 * reporting OUT_FULL. */
static int decompress_stream(const uint8_t *blob, size_t blob_len, const char *fa, size_t fa_len, const char *out, int device)
{
    char err[512];
    cbc_reference *R = NULL;
    if (cbc_reference_load(fa, fa_len, 0, &R, err, sizeof err)) { fprintf(stderr, "cbc: %s\n", err); return 1; }
    uint32_t L0 = cbc_stream_read_length(blob, blob_len);
    if (L0 < 1 || L0 > 256) { fprintf(stderr, "cbc: not a cbc file (neither a block container nor a stream with a sane read length)\n"); return 1; }
    cbc_gpu_ctx *ctx = NULL;
    int rc = cbc_gpu_init(device, &ctx);
    if (rc) { fprintf(stderr, "cbc: no usable MI355X (cbc_gpu_init = %d); there is no CPU fallback\n", rc); return 1; }
    if (cbc_gpu_upload_reference(ctx, R->bases, R->n_bytes)) { fprintf(stderr, "cbc: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    /* rows of the header read length rounded up to 4 (the decoder refuses a longer read: quirk Q7 makes fixed-length input
     * the only kind that decodes), room for one record per stream byte to begin with (files run at 1.4 - 2 bytes per read),
     * doubled while the kernel reports OUT_FULL; the record count is capped where rec_cap stops fitting 32 bits */
    const uint32_t stride = ((L0 < 4 ? 4 : L0) + 3u) & ~3u;
    uint64_t cap = (uint64_t)blob_len + 65536;
    for (;;) {
        if (cap > 0xffffffffull) cap = 0xffffffffull;
        cbc_read_rec *recs = (cbc_read_rec *)malloc((size_t)cap * sizeof(cbc_read_rec));
        uint8_t *seq = (uint8_t *)malloc((size_t)(cap * stride + 8));
        if (!recs || !seq) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
        cbc_stream_result sr; memset(&sr, 0, sizeof sr);
        rc = cbc_gpu_decode_stream(ctx, blob, blob_len, R->contig_off, R->contig_len, R->n_contigs, recs, cap, seq, cap * stride + 8, stride, &sr);
        if (rc == CBC_E_BLOCK && sr.status == CBC_ST_OUT_FULL && cap < 0xffffffffull) { free(recs); free(seq); cap *= 2; continue; }
        if (rc) { fprintf(stderr, "cbc: decode failed: %s\n", cbc_gpu_last_error(ctx)); return 1; }
        FILE *fo = fopen(out, "wb");
        if (!fo) { fprintf(stderr, "cbc: cannot write %s\n", out); return 1; }
        for (uint64_t r = 0; r < sr.nbytes; r++) {                 /* print_line, src/compression.c:16-40 */
            fwrite(seq + r * stride, 1, recs[r].rlen, fo); fputc('\n', fo);
        }
        if (fclose(fo) != 0) { fprintf(stderr, "cbc: cannot write %s\n", out); return 1; }
        printf("%llu reads decompressed from one stream\n", (unsigned long long)sr.nbytes);
        free(recs); free(seq);
        break;
    }
    cbc_gpu_shutdown(ctx);
    cbc_reference_free(R);
    return 0;
}

/* one device's share of the blocks: a contiguous range, decoded into its slice of the output arrays */
typedef struct { const cbc_unpack_plan *u; int device; uint32_t b0, b1; cbc_read_rec *recs; uint8_t *seq; int rc; char err[512]; } dec_job;

static void *dev_decode(void *arg)
{
    dec_job *J = (dec_job *)arg;
    const cbc_unpack_plan *u = J->u;
    if (J->b1 <= J->b0) return NULL;
    cbc_gpu_ctx *ctx = NULL;
    J->rc = cbc_gpu_init(J->device, &ctx);
    if (J->rc) { snprintf(J->err, sizeof J->err, "no usable MI355X at ordinal %d (cbc_gpu_init = %d)", J->device, J->rc); return NULL; }
    J->rc = cbc_gpu_upload_reference(ctx, u->ref, u->ref_bytes);
    if (!J->rc) {
        const uint32_t nb = J->b1 - J->b0;
        cbc_dec_block_desc *bl = (cbc_dec_block_desc *)malloc((size_t)nb * sizeof(cbc_dec_block_desc));
        if (!bl) J->rc = CBC_E_NOMEM;
        else {
            memcpy(bl, u->blocks + J->b0, (size_t)nb * sizeof(cbc_dec_block_desc));
            const uint64_t in0 = bl[0].in_off, r0 = bl[0].rec_base;
            uint64_t in1 = 0, nrec = 0;
            for (uint32_t k = 0; k < nb; k++) {
                if (bl[k].in_off + bl[k].in_bytes > in1) in1 = bl[k].in_off + bl[k].in_bytes;
                bl[k].in_off -= in0; bl[k].rec_base -= r0; bl[k].seq_base = bl[k].rec_base * u->seq_stride; nrec += bl[k].n_reads;
            }
            J->rc = cbc_gpu_decode_blocks(ctx, u->payloads + in0, in1 - in0, bl, nb, &u->caps, J->recs + r0, nrec,
                                          J->seq + r0 * u->seq_stride, nrec * u->seq_stride, NULL);   /* exactly this range's bytes: the next range belongs to another thread */
            free(bl);
        }
    }
    if (J->rc) snprintf(J->err, sizeof J->err, "%s", cbc_gpu_last_error(ctx));
    cbc_gpu_shutdown(ctx);
    return NULL;
}

int cbc_cli_decompress(const char *in, const char *out, const char *ref, const int *devs, int ndev)
{
    const int device = devs[0];
    size_t blob_len = 0, fa_len = 0;
    char *blob = slurp2(in, &blob_len), *fa = slurp2(ref, &fa_len);
    if (!blob || !fa) return 1;
    if (blob_len < 4 || memcmp(blob, "CBCB", 4) != 0) {
        int rc = decompress_stream((const uint8_t *)blob, blob_len, fa, fa_len, out, device);
        free(blob); free(fa);
        return rc;
    }
    char err[512];
    cbc_unpack_plan *u = NULL;
    int rc = cbc_unpack_plan_create((const uint8_t *)blob, blob_len, fa, fa_len, &u, err, sizeof err);
    free(fa);
    if (rc) { fprintf(stderr, "cbc: %s\n", err); return 1; }
    cbc_gpu_ctx *ctx = NULL;
    rc = cbc_gpu_init(device, &ctx);
    if (rc) { fprintf(stderr, "cbc: no usable MI355X (cbc_gpu_init = %d); there is no CPU fallback\n", rc); return 1; }
    if (cbc_gpu_upload_reference(ctx, u->ref, u->ref_bytes)) { fprintf(stderr, "cbc: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    uint64_t seq_bytes = u->long_reads ? u->seq_total : u->n_recs * u->seq_stride + 8;
    const uint64_t text_cap = (u->long_reads ? u->seq_total : u->n_recs * u->seq_stride) + u->n_recs + 16;
    cbc_read_rec *recs = (cbc_read_rec *)calloc((size_t)(u->n_recs ? u->n_recs : 1), sizeof(cbc_read_rec));
    uint8_t *seq = (uint8_t *)calloc((size_t)seq_bytes, 1);
    char *text = (char *)malloc((size_t)text_cap);
    if (!recs || !seq || !text) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
    if (u->long_reads) {
        rc = cbc_gpu_long_decode_blocks(ctx, u->payloads, u->payload_bytes, u->blocks, u->n_blocks, &u->caps, recs, u->n_recs, seq, seq_bytes, NULL);
        if (rc) { fprintf(stderr, "cbc: decode failed: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    } else if (ndev > 1) {
        /* contiguous block ranges balanced by record count, one host thread and one context per device */
        dec_job jobs[16]; pthread_t th[16];
        if (ndev > 16) ndev = 16;
        uint32_t b = 0; uint64_t done = 0;
        for (int d = 0; d < ndev; d++) {
            memset(&jobs[d], 0, sizeof jobs[d]);
            jobs[d].u = u; jobs[d].device = devs[d]; jobs[d].recs = recs; jobs[d].seq = seq; jobs[d].b0 = b;
            uint64_t target = u->n_recs * (uint64_t)(d + 1) / (uint64_t)ndev;
            while (b < u->n_blocks && (done < target || d == ndev - 1)) { done += u->blocks[b].n_reads; b++; }
            jobs[d].b1 = b;
            if (pthread_create(&th[d], NULL, dev_decode, &jobs[d]) != 0) { fprintf(stderr, "cbc: cannot start a device thread\n"); return 1; }
        }
        for (int d = 0; d < ndev; d++) { pthread_join(th[d], NULL); if (jobs[d].rc) { fprintf(stderr, "cbc: device %d: decode failed: %s\n", devs[d], jobs[d].err); rc = 1; } }
        if (rc) return 1;
    } else {
    rc = cbc_gpu_decode_blocks(ctx, u->payloads, u->payload_bytes, u->blocks, u->n_blocks, &u->caps, recs, u->n_recs, seq, seq_bytes, NULL);
    if (rc) { fprintf(stderr, "cbc: decode failed: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    }
    int64_t n = cbc_unpack_write_text(u, recs, seq, text, text_cap);
    if (n < 0) { fprintf(stderr, "cbc: text assembly failed\n"); return 1; }
    FILE *fo = fopen(out, "wb");
    if (!fo || fwrite(text, 1, (size_t)n, fo) != (size_t)n || fclose(fo) != 0) { fprintf(stderr, "cbc: cannot write %s\n", out); return 1; }
    printf("%llu reads decompressed from %u blocks\n", (unsigned long long)u->n_recs, u->n_blocks);
    free(text); free(seq); free(recs); free(blob);
    cbc_gpu_shutdown(ctx);
    cbc_unpack_plan_free(u);
    return 0;
}
