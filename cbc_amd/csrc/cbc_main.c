/*
 * cbc_main.c -- the `cbc` command line (C host program): keeps the reference's CLI surface for the
 * hot path and drives the HIP library through the C ABI.
 *
 *   cbc -c <in.sam> <out.cbc> <ref.fa>          README.md:58-62 form
 *   cbc -c 1 <in.sam> <out.cbc> <ref.fa>        src/main.c:114-123 form (-c eats a ratio; 1 = lossless)
 *   cbc -d <in.cbc> <out.txt> <ref.fa>          README.md:66-70 form   (local decompress)
 *   cbc -x <in.cbc> <out.txt> <ref.fa>          src/main.c:136-139 form
 *
 * Exactly three positional file names, in the reference's order (src/main.c:88-108, 200-204).
 * Network / QV modes of the reference (-u -s -r -D -w -t, and -d with user@host:file) are outside
 * the hot path and are refused with a message.  Extra options: --block-reads N, --device N.
 * Exit status: 0 on success (the reference returns 1 on success, src/main.c:370 -- not reproduced).
 *
 * There is no CPU encoder or decoder in this program: without an MI355X it exits with an error.
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/cbc_host.h"

static void usage(const char *p)
{
    fprintf(stderr,
            "usage: %s -c [1] <in.sam> <out.cbc> <ref.fa>   compress the reads of a position-sorted SAM\n"
            "       %s -d|-x <in.cbc> <out.txt> <ref.fa>    reconstruct the reads, one per line\n"
            "options: -l (header read length = longest read)  --block-reads N (default 4096)  --device N (default 0)\n", p, p);
}

static char *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cbc: cannot open %s: %s\n", path, strerror(errno)); return NULL; }
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return NULL; }
    long n = ftell(f);
    if (n < 0) { fclose(f); return NULL; }
    rewind(f);
    char *buf = (char *)malloc((size_t)n + 1);
    if (!buf) { fclose(f); fprintf(stderr, "cbc: out of memory reading %s\n", path); return NULL; }
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); fprintf(stderr, "cbc: short read on %s\n", path); return NULL; }
    fclose(f);
    buf[n] = 0;
    *len = (size_t)n;
    return buf;
}

static int is_number(const char *s)
{
    if (!s || !*s) return 0;
    char *e = NULL;
    (void)strtod(s, &e);
    return e && *e == 0;
}

static int do_compress(const char *in, const char *out, const char *ref, uint32_t block_reads, int device, int var_length)
{
    size_t sam_len = 0, fa_len = 0;
    char *sam = slurp(in, &sam_len), *fa = slurp(ref, &fa_len);
    if (!sam || !fa) return 1;
    printf("Compressing...\n");                                   /* src/compression.c:120 */
    char err[512];
    cbc_pack_opts po; cbc_pack_default_opts(&po);
    if (block_reads) po.block_reads = block_reads;
    po.var_length = (uint32_t)var_length;
    cbc_packed *p = NULL;
    int rc = cbc_pack_sam(sam, sam_len, fa, fa_len, &po, &p, err, sizeof err);
    free(sam); free(fa);
    if (rc) { fprintf(stderr, "cbc: %s\n", err); return 1; }
    cbc_gpu_ctx *ctx = NULL;
    rc = cbc_gpu_init(device, &ctx);
    if (rc) { fprintf(stderr, "cbc: no usable MI355X (cbc_gpu_init = %d); there is no CPU fallback\n", rc); cbc_packed_free(p); return 1; }
    rc = cbc_gpu_upload_reference(ctx, p->ref, p->ref_bytes);
    if (rc) { fprintf(stderr, "cbc: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    cbc_host_batch hb;
    memset(&hb, 0, sizeof hb);
    hb.recs = p->recs; hb.n_recs = p->n_recs; hb.seq = p->seq; hb.seq_bytes = p->seq_bytes;
    hb.tok = p->tok; hb.n_tok = p->n_tok; hb.names = p->names; hb.names_bytes = p->names_bytes;
    hb.blocks = p->blocks; hb.n_blocks = p->n_blocks; hb.caps = p->caps;
    uint64_t cap = cbc_gpu_plan_output(p->blocks, p->n_blocks, p->recs, p->tok);
    uint8_t *payloads = (uint8_t *)malloc(cap ? cap : 1);
    uint64_t *offs = (uint64_t *)calloc((size_t)p->n_blocks + 1, sizeof(uint64_t));
    if (!payloads || !offs) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
    rc = cbc_gpu_encode_blocks(ctx, &hb, payloads, cap, offs, NULL);
    if (rc) { fprintf(stderr, "cbc: encode failed: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    int64_t n = cbc_container_size(p, offs);
    uint8_t *blob = (uint8_t *)malloc((size_t)n);
    if (!blob || cbc_container_write(p, payloads, offs, blob, (uint64_t)n) != n) { fprintf(stderr, "cbc: container write failed\n"); return 1; }
    FILE *fo = fopen(out, "wb");
    if (!fo || fwrite(blob, 1, (size_t)n, fo) != (size_t)n || fclose(fo) != 0) { fprintf(stderr, "cbc: cannot write %s\n", out); return 1; }
    printf("Final Size: %lld\n", (long long)n);                   /* src/compression.c:157 */
    printf("%llu reads in %u blocks, %llu bases\n", (unsigned long long)p->n_recs, p->n_blocks, (unsigned long long)p->n_bases);
    free(blob); free(payloads); free(offs);
    cbc_gpu_shutdown(ctx);
    cbc_packed_free(p);
    return 0;
}

int cbc_cli_decompress(const char *in, const char *out, const char *ref, int device);   /* cbc_unpack.c */

int main(int argc, char **argv)
{
    const char *files[3] = { 0, 0, 0 };
    int nfiles = 0, mode = 0 /* 0 none, 1 compress, 2 decompress */, device = 0, var_length = 0;
    uint32_t block_reads = 0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-') {                                        /* src/main.c:88-108 */
            if (nfiles >= 3) { fprintf(stderr, "Garbage argument \"%s\" detected.\n", a); usage(argv[0]); return 1; }
            files[nfiles++] = a;
            continue;
        }
        if (!strcmp(a, "--block-reads") && i + 1 < argc) { block_reads = (uint32_t)strtoul(argv[++i], NULL, 10); continue; }
        if (!strcmp(a, "--device") && i + 1 < argc) { device = atoi(argv[++i]); continue; }
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) { usage(argv[0]); return 0; }
        switch (a[1]) {
        case 'c':
            mode = 1;
            /* src/main.c:114-123: -c consumes a ratio; README.md:58 has none.  Take it only when the
             * next argument is a number and three file names still follow. */
            if (i + 1 < argc && is_number(argv[i + 1])) {
                int remaining_files = 0;
                for (int k = i + 2; k < argc; k++) if (argv[k][0] != '-') remaining_files++;
                if (remaining_files + nfiles >= 3) {
                    if (strtod(argv[i + 1], NULL) != 1.0) { fprintf(stderr, "cbc: lossy quality-value mode (-c ratio != 1) is outside this build's scope\n"); return 1; }
                    i++;
                }
            }
            break;
        case 'x': mode = 2; break;
        case 'd': mode = 2; break;                                /* README form: local decompress */
        case 'l': var_length = 1; break;                         /* src/main.c: header read length = longest SEQ */
        case 'u': case 's': case 'r': case 'D': case 'w': case 't': case 'v':
            fprintf(stderr, "cbc: option %s belongs to the reference's network / quality-value modes, which are out of scope\n", a);
            return 1;
        default:
            fprintf(stderr, "cbc: unknown option %s\n", a); usage(argv[0]); return 1;
        }
    }
    if (mode == 0 || nfiles != 3) {                               /* src/main.c:200-204 */
        fprintf(stderr, "Missing required filenames (%d)\n", nfiles);
        usage(argv[0]);
        return 1;
    }
    if (mode == 2 && strchr(files[0], '@') && strchr(files[0], ':')) {
        fprintf(stderr, "cbc: user@host:file download mode (src/main.c:306-326) is out of scope\n");
        return 1;
    }
    return mode == 1 ? do_compress(files[0], files[1], files[2], block_reads, device, var_length)
                     : cbc_cli_decompress(files[0], files[1], files[2], device);
}
