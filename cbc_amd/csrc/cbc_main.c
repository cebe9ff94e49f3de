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
 * the hot path and are refused with a message.  Extra options: --block-reads N, --device N,
 * --threads N, --verbose, and --compat: write the reference's OWN file format (one arithmetic stream for the whole
 * file, compress() src/compression.c:112-170) instead of the block container -- byte-identical to the in-repo
 * oracle of compress() (-DDEBUG seed; parity with the upstream binary itself is unpinned);
 * one wavefront codes it, so it is slow.
 * `cbc -d` recognises either format.  --devices 0,1,...: one context and one host thread per listed device; whole
 * contigs are dealt to them largest first (encode) / contiguous block ranges (decode); the output does not depend
 * on the device count.  --long: the long-read format extension (stream version 3) -- reads up
 * to 65535 bases, which the reference cannot code at all; `cbc -d` recognises it by the container version.
 * Exit status: 0 on success (the reference returns 1 on success, src/main.c:370 -- not reproduced).
 *
 * There is no CPU encoder or decoder in this program: without an MI355X it exits with an error.
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <pthread.h>
#include "../../include/cbc_host.h"

static void usage(const char *p)
{
    fprintf(stderr,
            "usage: %s -c [1] <in.sam> <out.cbc> <ref.fa>   compress the reads of a position-sorted SAM\n"
            "       %s -d|-x <in.cbc> <out.txt> <ref.fa>    reconstruct the reads, one per line\n"
            "options: -l (header read length = longest read)  --block-reads N (default 4096)  --device N (default 0)\n"
            "         --threads N (SAM parser threads, default one per CPU)  --verbose (stage times)\n"
            "         --compat (write the reference's own single-stream format; slow: one stream = one wavefront)\n"
            "         --devices 0,1,... (shard the contigs / block ranges over several MI355X, one host thread each; the bitstreams\n"
            "                           reach device 0 over RCCL / xGMI when every ordinal is a device of its own)   --rccl (RCCL or fail)\n"
            "         --device-parse (tokenise the SAM text on the GPU; falls back to the host parser for leading soft clips / records without MD)\n"
            "         --long (long-read format extension: reads up to 65535 bases, any SAM line length; not a reference format)\n", p, p);
}

/* Input files are mapped, not copied: the packer only ever reads [0, len). */
static const char *map_file(const char *path, size_t *len)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) { fprintf(stderr, "cbc: cannot open %s: %s\n", path, strerror(errno)); return NULL; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { fprintf(stderr, "cbc: %s is not a regular file\n", path); close(fd); return NULL; }
    *len = (size_t)st.st_size;
    if (st.st_size == 0) { close(fd); return ""; }
    void *p = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { fprintf(stderr, "cbc: cannot map %s: %s\n", path, strerror(errno)); return NULL; }
    (void)madvise(p, (size_t)st.st_size, MADV_SEQUENTIAL);
    return (const char *)p;
}
static void unmap_file(const char *p, size_t len) { if (p && len) munmap((void *)p, len); }

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static int is_number(const char *s)
{
    if (!s || !*s) return 0;
    char *e = NULL;
    (void)strtod(s, &e);
    return e && *e == 0;
}

/* ---- several devices: whole contigs are dealt to the devices (cbc_assign_contigs), one
 * host thread and one context per device, no data exchanged between them; the container is assembled on the host in
 * global block order, so it is byte-identical to the one-device container. ---- */
#define CBC_MAX_DEVICES 16
typedef struct {
    const cbc_packed *p; int device; uint32_t part; const uint32_t *part_of_contig;
    uint8_t **blk_payload; uint32_t *blk_bytes;        /* per block, filled by the owning thread (malloc'ed runs) */
    uint8_t **runs; uint32_t n_runs;
    int rc; char err[512]; double seconds; float kernel_ms; uint64_t n_reads, ref_bytes;
    int keep_on_device;                /* the bitstreams stay in the context's stash (blk_stash_off) for the RCCL gather */
    uint64_t *blk_stash_off; cbc_gpu_ctx *ctx;
} dev_job;

static void *dev_encode(void *arg)
{
    dev_job *J = (dev_job *)arg;
    const cbc_packed *p = J->p;
    double t0 = now_s();
    cbc_gpu_ctx *ctx = NULL;
    J->rc = cbc_gpu_init(J->device, &ctx);
    if (J->rc) { snprintf(J->err, sizeof J->err, "no usable MI355X at ordinal %d (cbc_gpu_init = %d)", J->device, J->rc); return NULL; }
    /* this device's reference = the contigs it was dealt, end to end (not the whole genome: at GRCh38 size that would be
     * 3.1 GB over PCIe per device); new_base[c] = where contig c starts in it */
    uint64_t *new_base = (uint64_t *)calloc(p->n_contigs ? p->n_contigs : 1, sizeof(uint64_t));
    {
        const uint8_t **parts = (const uint8_t **)calloc(p->n_contigs ? p->n_contigs : 1, sizeof(uint8_t *));
        uint64_t *bytes = (uint64_t *)calloc(p->n_contigs ? p->n_contigs : 1, sizeof(uint64_t));
        uint32_t np = 0; uint64_t at = 0;
        if (!new_base || !parts || !bytes) { J->rc = CBC_E_NOMEM; snprintf(J->err, sizeof J->err, "out of memory"); free(parts); free(bytes); free(new_base); cbc_gpu_shutdown(ctx); return NULL; }
        for (uint32_t c = 0; c < p->n_contigs; c++) {
            if (J->part_of_contig[c] != J->part) continue;
            const uint64_t end = c + 1 < p->n_contigs ? p->contigs[c + 1].ref_off : p->ref_bytes;
            parts[np] = p->ref + p->contigs[c].ref_off; bytes[np] = end - p->contigs[c].ref_off;
            new_base[c] = at; at += bytes[np]; np++;
        }
        J->ref_bytes = at;
        J->rc = np ? cbc_gpu_upload_reference_parts(ctx, parts, bytes, np) : CBC_OK;
        free(parts); free(bytes);
    }
    if (J->rc) { snprintf(J->err, sizeof J->err, "%s", cbc_gpu_last_error(ctx)); free(new_base); cbc_gpu_shutdown(ctx); return NULL; }
    J->runs = (uint8_t **)calloc((size_t)p->n_blocks + 1, sizeof(uint8_t *));
    for (uint32_t b0 = 0; b0 < p->n_blocks && !J->rc; ) {
        if (J->part_of_contig[p->info[b0].contig] != J->part) { b0++; continue; }
        uint32_t b1 = b0;                                    /* a maximal run of consecutive blocks of this part */
        while (b1 < p->n_blocks && J->part_of_contig[p->info[b1].contig] == J->part) b1++;
        const uint32_t nb = b1 - b0;
        cbc_block_desc *bl = (cbc_block_desc *)malloc((size_t)nb * sizeof(cbc_block_desc));
        uint64_t *offs = (uint64_t *)calloc((size_t)nb + 1, sizeof(uint64_t));
        if (!bl || !offs) { J->rc = CBC_E_NOMEM; free(bl); free(offs); break; }
        memcpy(bl, p->blocks + b0, (size_t)nb * sizeof(cbc_block_desc));
        /* the run's records, bases and tokens are contiguous: hand over just that slice, bases rebased */
        const uint64_t r0 = bl[0].rec_base, s0 = bl[0].seq_base, t0k = bl[0].tok_base;
        const uint64_t r1 = bl[nb - 1].rec_base + bl[nb - 1].n_reads;
        const uint64_t s1 = (b1 < p->n_blocks) ? p->blocks[b1].seq_base : p->seq_bytes - 8, t1k = (b1 < p->n_blocks) ? p->blocks[b1].tok_base : p->n_tok;
        for (uint32_t k = 0; k < nb; k++) {
            const uint32_t c = p->info[b0 + k].contig;
            bl[k].rec_base -= r0; bl[k].seq_base -= s0; bl[k].tok_base -= t0k; J->n_reads += bl[k].n_reads;
            bl[k].ref_off = new_base[c] + (bl[k].ref_off - p->contigs[c].ref_off);      /* into this device's reference */
        }
        cbc_host_batch hb;
        memset(&hb, 0, sizeof hb);
        hb.recs = p->recs + r0; hb.n_recs = r1 - r0; hb.seq = p->seq + s0; hb.seq_bytes = s1 - s0 + 8;   /* 8 readable pad bytes follow */
        hb.tok = p->tok + t0k; hb.n_tok = t1k - t0k; hb.names = p->names; hb.names_bytes = p->names_bytes;
        hb.blocks = bl; hb.n_blocks = nb; hb.caps = p->caps;
        uint64_t cap = cbc_gpu_plan_output_caps(bl, nb, &hb.caps);
        uint8_t *pay = J->keep_on_device ? NULL : (uint8_t *)malloc(cap ? cap : 1);
        if (!pay && !J->keep_on_device) { J->rc = CBC_E_NOMEM; free(bl); free(offs); break; }
        const uint64_t stash0 = cbc_gpu_stash_bytes(ctx);
        J->rc = cbc_gpu_encode_blocks(ctx, &hb, pay, cap, offs, NULL);     /* pay == NULL: the bitstreams stay on the device */
        if (J->rc) snprintf(J->err, sizeof J->err, "%s", cbc_gpu_last_error(ctx));
        else {
            float ms = 0; if (!cbc_gpu_last_kernel_ms(ctx, &ms)) J->kernel_ms += ms;
            for (uint32_t k = 0; k < nb; k++) {
                J->blk_bytes[b0 + k] = (uint32_t)(offs[k + 1] - offs[k]);
                if (J->keep_on_device) J->blk_stash_off[b0 + k] = stash0 + offs[k];
                else J->blk_payload[b0 + k] = pay + offs[k];
            }
            if (pay) { J->runs[J->n_runs++] = pay; pay = NULL; }
        }
        free(pay); free(bl); free(offs);
        b0 = b1;
    }
    free(new_base);
    if (J->keep_on_device && !J->rc) J->ctx = ctx;               /* the gather needs the context; compress_on_devices shuts it down */
    else cbc_gpu_shutdown(ctx);
    J->seconds = now_s() - t0;
    return NULL;
}

static int parse_devices(const char *s, int *devs)
{
    int n = 0;
    while (*s && n < CBC_MAX_DEVICES) {
        char *e = NULL; long v = strtol(s, &e, 10);
        if (e == s || v < 0) return -1;
        devs[n++] = (int)v;
        s = (*e == ',') ? e + 1 : e;
        if (*e && *e != ',') return -1;
    }
    return n;
}

static int compress_on_devices(const cbc_packed *p, const int *devs, int ndev, const char *out, int verbose, int want_rccl)
{
    /* the exchange step: over RCCL (every device's bitstreams to device 0 over xGMI, then one D2H) when every listed ordinal
     * is a device of its own; two contexts on one device (rehearsals on a one-GPU box) bring theirs back over PCIe */
    int distinct = 1;
    for (int a = 0; a < ndev; a++) for (int b = 0; b < a; b++) if (devs[a] == devs[b]) distinct = 0;
    const int use_rccl = distinct && (ndev > 1 || want_rccl);
    if (want_rccl && !distinct) { fprintf(stderr, "cbc: --rccl wants every --devices ordinal once\n"); return 1; }
    uint64_t *blk_stash_off = (uint64_t *)calloc((size_t)p->n_blocks + 1, sizeof(uint64_t));
    uint8_t *gathered = NULL;
    uint32_t *part = (uint32_t *)calloc(p->n_contigs ? p->n_contigs : 1, sizeof(uint32_t));
    uint8_t **blk_payload = (uint8_t **)calloc((size_t)p->n_blocks + 1, sizeof(uint8_t *));
    uint32_t *blk_bytes = (uint32_t *)calloc((size_t)p->n_blocks + 1, sizeof(uint32_t));
    uint64_t *offs = (uint64_t *)calloc((size_t)p->n_blocks + 1, sizeof(uint64_t));
    dev_job jobs[CBC_MAX_DEVICES]; pthread_t th[CBC_MAX_DEVICES];
    if (!part || !blk_payload || !blk_bytes || !offs || cbc_assign_contigs(p, (uint32_t)ndev, part)) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
    double t0 = now_s();
    for (int d = 0; d < ndev; d++) {
        memset(&jobs[d], 0, sizeof jobs[d]);
        jobs[d].p = p; jobs[d].device = devs[d]; jobs[d].part = (uint32_t)d; jobs[d].part_of_contig = part;
        jobs[d].blk_payload = blk_payload; jobs[d].blk_bytes = blk_bytes;
        jobs[d].keep_on_device = use_rccl; jobs[d].blk_stash_off = blk_stash_off;
        if (pthread_create(&th[d], NULL, dev_encode, &jobs[d]) != 0) { fprintf(stderr, "cbc: cannot start a device thread\n"); return 1; }
    }
    int bad = 0;
    for (int d = 0; d < ndev; d++) { pthread_join(th[d], NULL); if (jobs[d].rc) { fprintf(stderr, "cbc: device %d: %s\n", devs[d], jobs[d].err); bad = 1; } }
    if (bad) return 1;
    const char *exchange = "one D2H per device (PCIe)";
    if (use_rccl) {
        cbc_gpu_ctx *ctxs[CBC_MAX_DEVICES]; uint64_t nbytes[CBC_MAX_DEVICES], sums[CBC_MAX_DEVICES], base[CBC_MAX_DEVICES], total = 0;
        for (int d = 0; d < ndev; d++) { ctxs[d] = jobs[d].ctx; nbytes[d] = cbc_gpu_stash_bytes(ctxs[d]); base[d] = total; total += nbytes[d]; }
        gathered = (uint8_t *)malloc(total ? total : 1);
        if (!gathered) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
        cbc_gpu_group *grp = NULL;
        int rc = cbc_gpu_group_create(ctxs, ndev, &grp);
        if (!rc) {
            rc = cbc_gpu_group_gather(grp, gathered, total, nbytes, sums);
            if (rc) fprintf(stderr, "cbc: RCCL gather failed: %s\n", cbc_gpu_group_last_error(grp));
            else {
                exchange = "RCCL: grouped ncclSend / ncclRecv to device 0 over xGMI, checksums verified, one D2H";
                if (verbose) for (int d = 0; d < ndev; d++) printf("rccl: member %d (device %d) %llu bytes, checksum %016llx sent == received\n", d, devs[d], (unsigned long long)nbytes[d], (unsigned long long)sums[d]);
            }
            cbc_gpu_group_destroy(grp);
        } else fprintf(stderr, "cbc: no RCCL group (%s)%s\n", cbc_gpu_last_error(ctxs[0]), want_rccl ? "" : "; falling back to one D2H per device");
        if (rc && want_rccl) return 1;
        if (rc) for (int d = 0; d < ndev; d++) if (cbc_gpu_stash_fetch(ctxs[d], gathered + base[d], nbytes[d])) { fprintf(stderr, "cbc: device %d: %s\n", devs[d], cbc_gpu_last_error(ctxs[d])); return 1; }
        for (uint32_t b = 0; b < p->n_blocks; b++) blk_payload[b] = gathered + base[part[p->info[b].contig]] + blk_stash_off[b];
        for (int d = 0; d < ndev; d++) cbc_gpu_shutdown(ctxs[d]);
    }
    double t1 = now_s();
    for (uint32_t b = 0; b < p->n_blocks; b++) offs[b + 1] = offs[b] + blk_bytes[b];
    uint8_t *flat = (uint8_t *)malloc(offs[p->n_blocks] ? offs[p->n_blocks] : 1);
    if (!flat) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
    for (uint32_t b = 0; b < p->n_blocks; b++) memcpy(flat + offs[b], blk_payload[b], blk_bytes[b]);   /* global block order */
    int64_t n = cbc_container_size(p, offs);
    uint8_t *blob = (uint8_t *)malloc((size_t)n);
    if (!blob || cbc_container_write(p, flat, offs, blob, (uint64_t)n) != n) { fprintf(stderr, "cbc: container write failed\n"); return 1; }
    FILE *fo = fopen(out, "wb");
    if (!fo || fwrite(blob, 1, (size_t)n, fo) != (size_t)n || fclose(fo) != 0) { fprintf(stderr, "cbc: cannot write %s\n", out); return 1; }
    printf("Final Size: %lld\n", (long long)n);
    printf("%llu reads in %u blocks, %llu bases, %d devices\n", (unsigned long long)p->n_recs, p->n_blocks, (unsigned long long)p->n_bases, ndev);
    if (verbose) for (int d = 0; d < ndev; d++)
        printf("device %d: %llu reads, %llu reference bytes uploaded (its contigs only), %.3f s (init + reference upload + encode), kernels %.3f ms\n", devs[d],
               (unsigned long long)jobs[d].n_reads, (unsigned long long)jobs[d].ref_bytes, jobs[d].seconds, (double)jobs[d].kernel_ms);
    if (verbose) printf("exchange: %s\n", exchange);
    if (verbose) printf("time: all devices %.3f s, assemble + write %.3f s\n", t1 - t0, now_s() - t1);
    free(gathered); free(blk_stash_off);
    for (int d = 0; d < ndev; d++) { for (uint32_t k = 0; k < jobs[d].n_runs; k++) free(jobs[d].runs[k]); free(jobs[d].runs); }
    free(flat); free(blob); free(part); free(blk_payload); free(blk_bytes); free(offs);
    return 0;
}

/* cbc_gpu_init (HIP runtime start-up, ~0.1 s) on its own thread while the host parses the text */
typedef struct { int device; cbc_gpu_ctx *ctx; int rc; double seconds; uint64_t est_recs, est_seq, est_tok, est_scratch; uint32_t est_blocks; } init_job;
static void *init_thread(void *arg)
{
    init_job *J = (init_job *)arg;
    double t = now_s();
    J->rc = cbc_gpu_init(J->device, &J->ctx);
    /* the device buffers of the encode call too, sized from an estimate of the file (they grow if it was short) */
    if (!J->rc && J->est_recs) (void)cbc_gpu_reserve_encode(J->ctx, J->est_recs, J->est_seq, J->est_tok, J->est_blocks, J->est_scratch);
    J->seconds = now_s() - t;
    return NULL;
}
/* records, bases and tokens a SAM text will pack into, from its first record line (uniform line lengths assumed) */
static void estimate_batch(const char *sam, size_t sam_len, uint32_t block_reads, init_job *J)
{
    size_t off = (size_t)cbc_sam_body_offset(sam, sam_len);
    if (off >= sam_len) return;
    const char *nl = (const char *)memchr(sam + off, '\n', sam_len - off);
    const size_t ll = nl ? (size_t)(nl - (sam + off)) + 1 : sam_len - off;
    if (ll < 20) return;
    size_t col = 0, seq_len = 0;
    for (size_t i = off; i < off + ll; i++) { if (sam[i] == '\t') { col++; continue; } if (col == 9) seq_len++; }
    if (seq_len == 0 || seq_len > 70000) return;
    const uint32_t br = block_reads ? block_reads : 4096u;
    J->est_recs = (uint64_t)((double)(sam_len - off) / (double)ll * 1.02) + 1024;
    J->est_seq = J->est_recs * seq_len + 64;
    J->est_tok = J->est_recs * 5;
    J->est_blocks = (uint32_t)(J->est_recs / br + J->est_recs / br / 8 + 16);
    J->est_scratch = (uint64_t)J->est_blocks * (3ull * (392ull + 16ull * br + 2ull * 8192ull) + 512ull + 4ull * (8192ull + 64ull) + 256ull);
}

static double g_main_t0;

static int do_compress(const char *in, const char *out, const char *ref, uint32_t block_reads, int device, int var_length, int threads, int verbose, int compat,
                       const int *devs, int ndev, int long_reads, int device_parse, int want_rccl)
{
    size_t sam_len = 0, fa_len = 0;
    double t0 = now_s();
    init_job IJ; pthread_t init_th; int init_started = 0;
    memset(&IJ, 0, sizeof IJ); IJ.device = device;
    const char *sam = map_file(in, &sam_len), *fa = map_file(ref, &fa_len);
    if (!sam || !fa) return 1;
    if (!(device_parse && !compat && !long_reads && ndev <= 1) && ndev <= 1 && !want_rccl) {
        if (!compat && !long_reads) estimate_batch(sam, sam_len, block_reads, &IJ);
        init_started = pthread_create(&init_th, NULL, init_thread, &IJ) == 0;
    }
    printf("Compressing...\n");                                   /* src/compression.c:120 */
    char err[512];
    cbc_pack_opts po; cbc_pack_default_opts(&po);
    if (block_reads) po.block_reads = block_reads;
    po.var_length = (uint32_t)var_length;
    po.n_threads = (uint32_t)threads;
    po.whole_file = compat ? 1u : 0u;
    po.long_reads = long_reads ? 1u : 0u;
    if (long_reads && !block_reads) po.block_reads = 64;
    cbc_packed *p = NULL;
    int rc = 0;
    cbc_gpu_ctx *tctx = NULL; cbc_tok_result tr; int tokenised = 0;
    memset(&tr, 0, sizeof tr);
    if (device_parse && !compat && !long_reads && ndev <= 1) {
        /* SAM text -> bases + tokens on the device (cbc_gpu_tokenise_sam); the host only cuts the blocks */
        rc = cbc_gpu_init(device, &tctx);
        if (rc) { fprintf(stderr, "cbc: no usable MI355X (cbc_gpu_init = %d); there is no CPU fallback\n", rc); return 1; }
        rc = cbc_gpu_tokenise_sam(tctx, sam, sam_len, cbc_sam_body_offset(sam, sam_len), &tr);
        if (rc) { fprintf(stderr, "cbc: %s\n", cbc_gpu_last_error(tctx)); return 1; }
        if (tr.status == 3) { if (verbose) printf("device tokeniser: line %llu needs the host packer (leading soft clip or no MD field); parsing on the host\n", (unsigned long long)tr.bad_line + 1); }
        else if (tr.status) { fprintf(stderr, "cbc: malformed SAM at line %llu (tokeniser status %u)\n", (unsigned long long)tr.bad_line + 1, tr.status); return 1; }
        else {
            rc = cbc_pack_from_device_tokens(sam, sam_len, fa, fa_len, &po, tr.summaries, tr.rname_change, tr.change_name_off, tr.change_name_len,
                                             tr.n_recs, tr.n_unmapped, NULL, tr.seq_bytes, NULL, tr.n_tok, &p, err, sizeof err);
            tokenised = !rc;
        }
    }
    if (!tokenised && !rc) rc = cbc_pack_sam(sam, sam_len, fa, fa_len, &po, &p, err, sizeof err);
    unmap_file(sam, sam_len); unmap_file(fa, fa_len);
    double t1 = now_s();
    if (rc) { if (init_started) pthread_join(init_th, NULL); fprintf(stderr, "cbc: %s\n", err); return 1; }
    if ((ndev > 1 || want_rccl) && !compat && !long_reads) {
        rc = compress_on_devices(p, devs, ndev, out, verbose, want_rccl);
        cbc_packed_free(p);
        return rc;
    }
    cbc_gpu_ctx *ctx = tctx;
    if (init_started) { pthread_join(init_th, NULL); init_started = 0; ctx = IJ.ctx; rc = IJ.rc; }
    else if (!ctx) rc = cbc_gpu_init(device, &ctx);
    if (rc) { fprintf(stderr, "cbc: no usable MI355X (cbc_gpu_init = %d); there is no CPU fallback\n", rc); cbc_packed_free(p); return 1; }
    double t1b = now_s();
    rc = cbc_gpu_upload_reference(ctx, p->ref, p->ref_bytes);
    if (rc) { fprintf(stderr, "cbc: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    cbc_host_batch hb;
    memset(&hb, 0, sizeof hb);
    hb.recs = p->recs; hb.n_recs = p->n_recs; hb.seq = p->seq; hb.seq_bytes = p->seq_bytes;
    hb.tok = p->tok; hb.n_tok = p->n_tok; hb.names = p->names; hb.names_bytes = p->names_bytes;
    hb.blocks = p->blocks; hb.n_blocks = p->n_blocks; hb.caps = p->caps;
    if (compat) {
        /* the reference's format: the stream IS the file (no container, no index) */
        uint64_t scap = 4096 + 48ull * p->n_recs + 8ull * p->n_tok;
        uint8_t *stream = (uint8_t *)malloc(scap);
        cbc_stream_result sr;
        if (!stream) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
        double t2 = now_s();
        rc = cbc_gpu_encode_stream(ctx, &hb, stream, scap, &sr);
        if (rc) { fprintf(stderr, "cbc: encode failed: %s\n", cbc_gpu_last_error(ctx)); return 1; }
        FILE *fo = fopen(out, "wb");
        if (!fo || fwrite(stream, 1, (size_t)sr.nbytes, fo) != (size_t)sr.nbytes || fclose(fo) != 0) { fprintf(stderr, "cbc: cannot write %s\n", out); return 1; }
        printf("Final Size: %llu\n", (unsigned long long)sr.nbytes);
        printf("%llu reads in one stream, %llu bases, %llu coded symbols\n", (unsigned long long)p->n_recs, (unsigned long long)p->n_bases, (unsigned long long)sr.n_symbols);
        if (verbose) printf("time: pack %.3f s, device init + reference upload %.3f s, encode %.3f s\n", t1 - t0, t2 - t1, now_s() - t2);
        free(stream); cbc_gpu_shutdown(ctx); cbc_packed_free(p);
        return 0;
    }
    uint64_t cap = tokenised ? 4096ull * p->n_blocks + 48ull * p->n_recs + 16ull * p->n_tok
                             : cbc_gpu_plan_output_caps(p->blocks, p->n_blocks, &p->caps);
    uint8_t *payloads = (uint8_t *)malloc(cap ? cap : 1);
    uint64_t *offs = (uint64_t *)calloc((size_t)p->n_blocks + 1, sizeof(uint64_t));
    if (!payloads || !offs) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
    double t2 = now_s();
    if (long_reads) {                                             /* stream version 3: reads up to 65535 bases */
        free(payloads);
        cap = 8192ull * p->n_blocks + 64ull * p->n_recs + 13ull * p->n_bases;
        if (p->n_bases > (1ull << 28)) cap = 8192ull * p->n_blocks + 64ull * p->n_recs + 2ull * p->n_bases;
        payloads = (uint8_t *)malloc(cap);
        if (!payloads) { fprintf(stderr, "cbc: out of memory\n"); return 1; }
        rc = cbc_gpu_long_encode_blocks(ctx, &hb, payloads, cap, offs, NULL);
    } else if (tokenised) {
        rc = cbc_gpu_encode_blocks_tokenised(ctx, &tr, &hb, payloads, cap, offs, NULL);
        cbc_gpu_tokenise_free(ctx, &tr);
    } else
    rc = cbc_gpu_encode_blocks(ctx, &hb, payloads, cap, offs, NULL);
    if (rc) { fprintf(stderr, "cbc: encode failed: %s\n", cbc_gpu_last_error(ctx)); return 1; }
    double t3 = now_s();
    int64_t n = cbc_container_size(p, offs);
    uint8_t *blob = (uint8_t *)malloc((size_t)n);
    if (!blob || cbc_container_write(p, payloads, offs, blob, (uint64_t)n) != n) { fprintf(stderr, "cbc: container write failed\n"); return 1; }
    FILE *fo = fopen(out, "wb");
    if (!fo || fwrite(blob, 1, (size_t)n, fo) != (size_t)n || fclose(fo) != 0) { fprintf(stderr, "cbc: cannot write %s\n", out); return 1; }
    printf("Final Size: %lld\n", (long long)n);                   /* src/compression.c:157 */
    printf("%llu reads in %u blocks, %llu bases\n", (unsigned long long)p->n_recs, p->n_blocks, (unsigned long long)p->n_bases);
    float kms = 0; (void)cbc_gpu_last_kernel_ms(ctx, &kms);
    double t4 = now_s();
    if (verbose)
        printf("time: pack %.3f s, device init + reference upload %.3f s, encode (H2D + kernel %.3f ms + D2H) %.3f s, write %.3f s\n",
               t1 - t0, t2 - t1, (double)kms, t3 - t2, t4 - t3);
    /* The output file is complete and closed.  Like the reference, which frees nothing (src/compression.c:112-170), the
     * process leaves its gigabytes of host arrays and its device context to the operating system: freeing them one by one
     * and running the HIP runtime's exit handlers took a third of a 4 M-read run (0.14 s + 0.14 s of 0.54 s). */
    const int keep = getenv("CBC_FULL_TEARDOWN") != NULL;          /* leak checkers and the like */
    if (keep) { free(blob); free(payloads); free(offs); cbc_gpu_shutdown(ctx); cbc_packed_free(p); }
    if (verbose) {
        /* every second between main() and here, by stage (SURVEY.md 8d region iii); what is left of the process's wall time
         * is program loading before main() */
        double t5 = now_s();
        printf("stage argv + map files: %.3f s\n", t0 - g_main_t0);
        printf("stage parse + pack (host%s): %.3f s\n", tokenised ? ", tokens from the device" : "", t1 - t0);
        printf("stage device init: %.3f s on its own thread, %.3f s not hidden behind the packer\n", IJ.seconds, t1b - t1);
        printf("stage reference upload: %.3f s\n", t2 - t1b);
        printf("stage encode (H2D + kernels + D2H): %.3f s\n", t3 - t2);
        printf("stage container + write: %.3f s\n", t4 - t3);
        printf("stage teardown (free, context shutdown): %.3f s\n", t5 - t4);
        printf("stage total inside main: %.3f s\n", t5 - g_main_t0);
    }
    if (!keep) { fflush(NULL); _exit(0); }
    return 0;
}

int cbc_cli_decompress(const char *in, const char *out, const char *ref, const int *devs, int ndev);   /* cbc_cli_unpack.c */

int main(int argc, char **argv)
{
    const char *files[3] = { 0, 0, 0 };
    int nfiles = 0, mode = 0 /* 0 none, 1 compress, 2 decompress */, device = 0, var_length = 0, threads = 0, verbose = 0, compat = 0, long_reads = 0, device_parse = 0, want_rccl = 0;
    int devs[CBC_MAX_DEVICES] = { 0 }, ndev = 0;
    uint32_t block_reads = 0;
    g_main_t0 = now_s();
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-') {                                        /* src/main.c:88-108 */
            if (nfiles >= 3) { fprintf(stderr, "Garbage argument \"%s\" detected.\n", a); usage(argv[0]); return 1; }
            files[nfiles++] = a;
            continue;
        }
        if (!strcmp(a, "--block-reads") && i + 1 < argc) { block_reads = (uint32_t)strtoul(argv[++i], NULL, 10); continue; }
        if (!strcmp(a, "--device") && i + 1 < argc) { device = atoi(argv[++i]); continue; }
        if (!strcmp(a, "--devices") && i + 1 < argc) {            /* e.g. 0,1,2,3: contigs are sharded over them */
            ndev = parse_devices(argv[++i], devs);
            if (ndev < 1) { fprintf(stderr, "cbc: --devices wants a comma-separated list of ordinals\n"); return 1; }
            device = devs[0];
            continue;
        }
        if (!strcmp(a, "--threads") && i + 1 < argc) { threads = atoi(argv[++i]); if (threads < 0) threads = 0; continue; }
        if (!strcmp(a, "--verbose")) { verbose = 1; continue; }
        if (!strcmp(a, "--compat")) { compat = 1; continue; }
        if (!strcmp(a, "--long")) { long_reads = 1; continue; }
        if (!strcmp(a, "--device-parse")) { device_parse = 1; continue; }
        if (!strcmp(a, "--rccl")) { want_rccl = 1; continue; }       /* the RCCL exchange or nothing (also with one device: a self-send) */
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
    if (ndev == 0) { devs[0] = device; ndev = 1; }
    return mode == 1 ? do_compress(files[0], files[1], files[2], block_reads, device, var_length, threads, verbose, compat, devs, ndev, long_reads, device_parse, want_rccl)
                     : cbc_cli_decompress(files[0], files[1], files[2], devs, ndev);
}
