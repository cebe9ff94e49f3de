/*
 * cbc_cli_unpack.c -- `cbc -d / -x`: container + FASTA -> one reconstructed read per line, decoded
 * on the GPU (decompress(), src/compression.c:173-216; print_line :16-40).
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
    const uint32_t stride = 256;
    uint64_t cap = blob_len * 2 + 65536;
    for (;;) {
        cbc_read_rec *recs = (cbc_read_rec *)malloc((size_t)cap * sizeof(cbc_read_rec));
        uint8_t *seq = (uint8_t *)malloc((size_t)(cap * stride + 8));
        if (!recs || !seq) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
        cbc_stream_result sr; memset(&sr, 0, sizeof sr);
        rc = cbc_gpu_decode_stream(ctx, blob, blob_len, R->contig_off, R->contig_len, R->n_contigs, recs, cap, seq, cap * stride + 8, stride, &sr);
        if (rc == CBC_E_BLOCK && sr.status == CBC_ST_OUT_FULL && cap < 0x7fffffffull) { free(recs); free(seq); cap *= 4; continue; }
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

int cbc_cli_decompress(const char *in, const char *out, const char *ref, int device)
{
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
    uint64_t seq_bytes = u->n_recs * u->seq_stride + 8;
    cbc_read_rec *recs = (cbc_read_rec *)calloc((size_t)(u->n_recs ? u->n_recs : 1), sizeof(cbc_read_rec));
    uint8_t *seq = (uint8_t *)calloc((size_t)seq_bytes, 1);
    char *text = (char *)malloc((size_t)(u->n_recs * (u->seq_stride + 1) + 16));
    if (!recs || !seq || !text) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
    rc = cbc_gpu_decode_blocks(ctx, u->payloads, u->payload_bytes, u->blocks, u->n_blocks, &u->caps, recs, u->n_recs, seq, seq_bytes, NULL);
    if (rc) { fprintf(stderr, "cbc: decode failed: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    int64_t n = cbc_unpack_write_text(u, recs, seq, text, u->n_recs * (u->seq_stride + 1) + 16);
    if (n < 0) { fprintf(stderr, "cbc: text assembly failed\n"); return 1; }
    FILE *fo = fopen(out, "wb");
    if (!fo || fwrite(text, 1, (size_t)n, fo) != (size_t)n || fclose(fo) != 0) { fprintf(stderr, "cbc: cannot write %s\n", out); return 1; }
    printf("%llu reads decompressed from %u blocks\n", (unsigned long long)u->n_recs, u->n_blocks);
    free(text); free(seq); free(recs); free(blob);
    cbc_gpu_shutdown(ctx);
    cbc_unpack_plan_free(u);
    return 0;
}
