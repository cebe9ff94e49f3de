/*
 * cbc_gpu.h -- C ABI of the MI355X-native cbc hot path (libcbc_gpu.so, built from cbc_amd/csrc).
 *
 * The reference (1mishra/cbc) has no plugin/FFI seam: its hot path sits behind a CLI and a file
 * format.  The entry points below are what a maintainer would bind in
 * place of the per-file encode loop
 *      compress()              src/compression.c:112-170
 *        compress_line()       src/compression.c:42-69
 *          compress_rname()    src/id_compression.c:39-65
 *          compress_read()     src/read_compression.c:15-44   (pos/flag/match/edits/var/chars)
 *            send_value_to_as  src/stream_model.c:53-76  + update_model :31-51
 *            arithmetic_encoder_step  src/Arithmetic_stream.c:274-345
 *        encoder_last_step()   src/Arithmetic_stream.c:348-371
 * operating on *packed* records (the output of the load_sam_line() tokeniser,
 * src/sam_file_allocation.c:437-529) instead of SAM text.
 *
 * Unit of work: a BLOCK = up to N consecutive records of one contig, with POS rebased so that
 * the block's first record has POS 1.  The payload produced for a block is byte-identical to the
 * file the reference encoder (built with -DDEBUG, i.e. constant WELL seed) writes when it is run
 * on that block alone: a SAM holding just those records with the rebased POS and a one-contig
 * FASTA holding the reference window that starts at the block's first base.  One block is one
 * independent arithmetic stream, coded by one workgroup (encode: a model wavefront feeding a coder
 * wavefront through LDS; decode: one wavefront).
 *
 * Plain C: pointers and sizes only, caller owns every buffer, no global state, every function
 * returns 0 on success or a negative CBC_E_* code.  One host thread per context.
 */
#ifndef CBC_GPU_H
#define CBC_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBC_ABI_VERSION 1

/* ---- error / status codes ---------------------------------------------------------------- */
#define CBC_OK              0
#define CBC_E_ARG          -1   /* bad argument                                                */
#define CBC_E_NODEV        -2   /* no HIP device / HIP runtime error (see cbc_gpu_last_error)   */
#define CBC_E_NOMEM        -3
#define CBC_E_BLOCK        -4   /* at least one block finished with status != CBC_ST_OK         */
#define CBC_E_INPUT        -5   /* input violates a reference limit (message in last_error)     */
#define CBC_E_IO           -6

/* per-block status written by the kernel (cbc_block_result.status) */
#define CBC_ST_OK           0
#define CBC_ST_OUT_FULL     1   /* out_cap too small                                            */
#define CBC_ST_ASSERT       2   /* the reference would abort here: zero-count symbol, symbol >= */
                                /* alphabet, unsorted POS, var context >= 65535 ...             */
#define CBC_ST_CAP_POS      3   /* more distinct POS deltas than lds.cap_pos                    */
#define CBC_ST_CAP_FLAG     4   /* more than CBC_CAP_FLAG distinct FLAG values                  */
#define CBC_ST_CAP_VAR      5   /* more var symbols than lds.cap_var                            */
#define CBC_ST_CAP_NAME     6   /* contig name longer than CBC_CAP_NAME-3                       */
#define CBC_ST_UNSUPPORTED  7   /* raw leading-S / '*' op in the tokens (the packer emits a leading  */
                                /* soft clip as an I op after rebuilding MD, quirk Q6); a block    */
                                /* of more than CBC_MAX_BLOCK_READS records; a LOSSY stream         */

/* ---- packed record layout (device and host share it) ---------------------------------------- */

/* One per mapped record, 16 bytes, array-of-structs so a wavefront loads 64 records with one
 * 16-byte-per-lane coalesced access.  Mirrors struct read_line_t (include/sam_block.h:177-185). */
typedef struct cbc_read_rec {
    uint32_t pos;       /* POS, 1-based, relative to the block's reference window               */
    uint16_t flag;      /* SAM FLAG (read_line_t.invFlag)                                       */
    uint16_t rlen;      /* strlen(SEQ), <= CBC_MAX_READ_LEN                                     */
    uint32_t seq_off;   /* byte offset of SEQ in seq[], relative to the block's seq_base        */
    uint32_t tok_off;   /* word offset of the CIGAR/MD tokens in tok[], relative to tok_base    */
} cbc_read_rec;

/* Token stream of one record at tok[tok_base + tok_off]:
 *   word 0            : n_cigar | (n_md << 16)
 *   word 1            : n_deleted_bases | (n_inserted_bases << 16)   (sum of D lengths; sum of I and
 *                       trailing-S lengths) -- the counts compress_edits() codes first (:557-565);
 *                       the number of SNPs is n_md (the packer rejects MD strings that are
 *                       inconsistent with the read)
 *   n_cigar words     : (len << 4) | op      op: CBC_OP_M/I/D/S/STAR; len = atoi() of the CIGAR
 *                       segment exactly as compress_edits() reads it (read_compression.c:308-352)
 *   n_md words        : (gap << 8) | letter  one per mismatch letter of MD:Z, gap = matched bases
 *                       since the previous mismatch (numbers on both sides of a '^' run add up,
 *                       add_snps_to_array read_compression.c:613-701); letter is the raw byte,
 *                       so the trailing '\n' of an MD that is the last column survives (quirk Q2)
 */
#define CBC_OP_M    0u
#define CBC_OP_I    1u
#define CBC_OP_D    2u
#define CBC_OP_S    3u
#define CBC_OP_STAR 4u

/* One per block, 64 bytes. */
typedef struct cbc_block_desc {
    uint64_t rec_base;     /* index of the block's first cbc_read_rec                           */
    uint64_t seq_base;     /* byte offset of the block's bases in seq[]                         */
    uint64_t tok_base;     /* word offset of the block's tokens in tok[]                        */
    uint64_t ref_off;      /* byte offset in the device reference of the base that is POS 1     */
    uint64_t out_off;      /* byte offset of this block's payload area in out[] (4-aligned)     */
    uint32_t out_cap;      /* bytes available at out_off (multiple of 256)                      */
    uint32_t n_reads;
    uint32_t name_off;     /* offset in names[] of the NUL-terminated contig name               */
    uint32_t read_length;  /* header read length L0 (get_read_length, sam_file_allocation.c:26) */
    uint32_t n_tok;        /* words of tok[] owned by this block (bounds the token prefetch)    */
    uint32_t reserved;     /* set by cbc_gpu_plan_output: bytes of the out area that are payload; */
                           /* the rest, up to out_cap, holds the block's var-event list          */
} cbc_block_desc;

typedef struct cbc_block_result {
    uint32_t nbytes;       /* payload bytes written at out_off                                  */
    uint32_t status;       /* CBC_ST_*                                                          */
    uint32_t n_symbols;    /* arithmetic-coder steps taken                                      */
    uint32_t fail_read;    /* block-local index of the record being coded when status was set   */
} cbc_block_result;

/* limits of the LDS-resident model tables; the packer cuts blocks so that they hold */
#define CBC_MAX_READ_LEN   252u    /* var context (((L+2)<<7)+L)*2+1 must stay < 65535 (sam_models.c:317) */
#define CBC_CAP_FLAG       64u     /* distinct FLAG values per block                              */
#define CBC_CAP_NAME       128u    /* (context,char) pairs of the contig-name model               */
#define CBC_MAX_BLOCK_READS 16384u /* keeps every adaptive total below the 2^20 rescale point     */
#define CBC_REF_PAD        512u    /* zero bytes the caller appends after every contig            */
#define CBC_WELL_SEED      0x55555555u  /* sam_file_allocation.c:399, the -DDEBUG constant         */

typedef struct cbc_lds_caps {
    uint32_t cap_pos;      /* entries of the POS-delta alphabet in LDS (>= max distinct deltas + 1) */
    uint32_t cap_var;      /* var symbols per block: sizes the event list, which lives in global    */
                           /* memory (behind the payload area on encode, in d_var_scratch on decode) */
} cbc_lds_caps;

/* ---- context ------------------------------------------------------------------------------------ */
typedef struct cbc_gpu_ctx cbc_gpu_ctx;

int  cbc_gpu_abi_version(void);
int  cbc_gpu_device_count(void);
int  cbc_gpu_init(int device_ordinal, cbc_gpu_ctx **ctx);
int  cbc_gpu_shutdown(cbc_gpu_ctx *ctx);
const char *cbc_gpu_last_error(cbc_gpu_ctx *ctx);

/* Upload the reference the blocks' ref_off point into: upper-cased contig bases, one byte per
 * base, each contig followed by CBC_REF_PAD zero bytes (what store_reference_in_memory,
 * src/read_decompression.c:17-53, keeps in `reference[]`, for all contigs at once). */
int  cbc_gpu_upload_reference(cbc_gpu_ctx *ctx, const uint8_t *bases, uint64_t nbytes);
/* The same from several host pieces, laid end to end on the device in the order given (piece k starts at the sum of the
 * sizes before it): a device that codes some contigs of a genome uploads those contigs, not all of it. */
int  cbc_gpu_upload_reference_parts(cbc_gpu_ctx *ctx, const uint8_t *const *parts, const uint64_t *bytes, uint32_t n_parts);

/* Host-buffer entry point: copies the batch to the device, codes every block, copies the payloads
 * back *compacted*: block b's payload is out[out_offsets[b] .. out_offsets[b+1]).  `blocks[].out_off`
 * and `.out_cap` are filled in by the library.  results may be NULL. */
typedef struct cbc_host_batch {
    const cbc_read_rec   *recs;   uint64_t n_recs;
    const uint8_t        *seq;    uint64_t seq_bytes;
    const uint32_t       *tok;    uint64_t n_tok;
    const uint8_t        *names;  uint32_t names_bytes;
    cbc_block_desc       *blocks; uint32_t n_blocks;
    cbc_lds_caps          caps;
} cbc_host_batch;

int  cbc_gpu_encode_blocks(cbc_gpu_ctx *ctx, const cbc_host_batch *batch,
                           uint8_t *out, uint64_t out_cap, uint64_t *out_offsets /* n_blocks+1 */,
                           cbc_block_result *results /* n_blocks or NULL */);

/* The host-buffer entry points run as a pipeline (chunked H2D on a copy stream overlapped with the launches of the chunks
 * already on the device; device arrays kept in the context between calls).  What the most recent one did: */
typedef struct cbc_e2e_times {
    double   total_s;        /* whole call                                                                  */
    double   alloc_s;        /* growing the context's device arenas (0 once they have their size)           */
    double   issue_s;        /* until the last chunk's copies and launch had been handed to the runtime     */
    double   kernels_done_s; /* until every block had been coded and the sizes were on the host            */
    uint64_t h2d_bytes, d2h_bytes;
    uint32_t n_chunks, reserved;
} cbc_e2e_times;
int  cbc_gpu_last_e2e(cbc_gpu_ctx *ctx, cbc_e2e_times *out);
/* Page-lock caller-owned host memory the entry points read from or write to (hipHostRegister): DMA then runs at the link
 * rate and asynchronously.  Worth it for buffers that are reused; the entry points accept pageable memory as well. */
int  cbc_gpu_host_register(cbc_gpu_ctx *ctx, const void *p, uint64_t bytes);
int  cbc_gpu_host_unregister(cbc_gpu_ctx *ctx, const void *p);

/* ---- several devices in one process: the exchange step -------------------------------------------
 * cbc_gpu_encode_blocks (and _2bit / _tokenised) with out == NULL keep the compacted bitstreams ON THE DEVICE, appended to the
 * context's stash (out_offsets are relative to the call, as always).  A group over one context per device then moves every
 * member's stash to member 0's device with grouped ncclSend / ncclRecv (RCCL over xGMI), checks a checksum taken on the
 * sending device against one taken on the receiving device, and hands the bytes to the host in member order.  librccl.so is
 * loaded when the first group is created.  A one-member group sends to itself (self-test of the call sites).
 * Without RCCL (or with two contexts on one device) cbc_gpu_stash_fetch() brings a member's stash back over PCIe instead. */
typedef struct cbc_gpu_group cbc_gpu_group;
int  cbc_gpu_group_create(cbc_gpu_ctx *const *ctxs, int n, cbc_gpu_group **out);       /* CBC_E_ARG when two contexts share a device */
int  cbc_gpu_group_gather(cbc_gpu_group *g, uint8_t *out, uint64_t out_cap, uint64_t *nbytes /* n */, uint64_t *sums /* n or NULL */);
void cbc_gpu_group_destroy(cbc_gpu_group *g);
const char *cbc_gpu_group_last_error(cbc_gpu_group *g);
int  cbc_gpu_stash_reset(cbc_gpu_ctx *ctx);
uint64_t cbc_gpu_stash_bytes(cbc_gpu_ctx *ctx);
int  cbc_gpu_stash_fetch(cbc_gpu_ctx *ctx, uint8_t *out, uint64_t out_cap);

/* Device-pointer entry point (what bench.py and the multi-GPU host use): every pointer is a
 * device address, the launch is asynchronous on `hip_stream`, a hipStream_t used exactly as given
 * (NULL is HIP's null stream, which is what torch's default stream is).  Everything the launch reads
 * or overwrites must be ordered before it ON THAT STREAM by the caller.  out_off/out_cap in d_blocks
 * must already be set (cbc_gpu_plan_output). */
typedef struct cbc_device_batch {
    const cbc_read_rec   *d_recs;
    const uint8_t        *d_seq;
    const uint32_t       *d_tok;
    const uint8_t        *d_names;
    const cbc_block_desc *d_blocks;  uint32_t n_blocks;
    const uint8_t        *d_ref;     uint64_t ref_bytes;
    uint8_t              *d_out;     uint64_t out_bytes;
    cbc_block_result     *d_results;
    uint64_t              seq_bytes;
    uint64_t              n_tok;
    uint64_t              n_recs;
    cbc_lds_caps          caps;
} cbc_device_batch;

int  cbc_gpu_encode_blocks_device(cbc_gpu_ctx *ctx, const cbc_device_batch *batch, void *hip_stream);

/* Device-side compaction of the per-block payload areas written by the encode launch: an exclusive
 * scan of cbc_block_result.nbytes into d_offsets[n_blocks+1], then a gather into d_packed
 * (block b -> d_packed[d_offsets[b] .. d_offsets[b+1])).  Asynchronous on `hip_stream`. */
int  cbc_gpu_compact_device(cbc_gpu_ctx *ctx, const uint8_t *d_scratch, const cbc_block_desc *d_blocks,
                            const cbc_block_result *d_results, uint32_t n_blocks, uint64_t *d_offsets,
                            uint8_t *d_packed, uint64_t packed_cap, void *hip_stream);

/* Checksum of payload bytes, for the exchange step of the multi-GPU path: every rank sums its
 * compacted payloads on the device before the gather, the receiver sums what arrived (on its device, or on the host with
 * libcbc_host's cbc_checksum64 -- same value) and compares.  sum over i of (byte[i] + 1) * (((i + 1) * 0x9E3779B97F4A7C15) | 1)
 * mod 2^64: order-free, so it is exact under any reduction order; position-weighted, so a shifted, truncated, padded or
 * permuted buffer does not pass.  *d_sum (device, 8 bytes) is valid once `hip_stream` has reached this point. */
#define CBC_CHECKSUM_TERM(i, byte) ((uint64_t)((uint32_t)(byte) + 1u) * ((((uint64_t)(i) + 1u) * 0x9E3779B97F4A7C15ull) | 1ull))
int  cbc_gpu_checksum_device(cbc_gpu_ctx *ctx, const uint8_t *d_bytes, uint64_t n, uint64_t *d_sum, void *hip_stream);

/* Fills blocks[b].out_off / out_cap with a worst-case bound (every coded symbol costs at most 20
 * bits because every model total stays below 2^20) and returns the scratch bytes needed. */
uint64_t cbc_gpu_plan_output(cbc_block_desc *blocks, uint32_t n_blocks,
                             const cbc_read_rec *recs, const uint32_t *tok);

/* The same bound from the caps alone, without reading the records: no block holds more than caps->cap_var - 1 edit
 * events (the packers cut blocks that way).  O(blocks); what the host-buffer entry points use, and what sizes their `out`. */
uint64_t cbc_gpu_plan_output_caps(cbc_block_desc *blocks, uint32_t n_blocks, const cbc_lds_caps *caps);
/* Grow the context's device buffers for a batch of this shape before the batch exists (allocation of gigabytes takes
 * tens of milliseconds the first time: a CLI does it on the device-init thread while the host still parses the text). */
int  cbc_gpu_reserve_encode(cbc_gpu_ctx *ctx, uint64_t n_recs, uint64_t seq_bytes, uint64_t n_tok, uint32_t n_blocks,
                            uint64_t scratch_bytes);

/* Dynamic LDS bytes one block's workgroup needs for `caps` (workgroups per CU = 160 KiB / this). */
uint32_t cbc_gpu_lds_bytes(const cbc_lds_caps *caps);

/* ---- decode direction (replaces decompress(), src/compression.c:173-216,
 *      decompress_read()/reconstruct_read(), src/read_decompression.c:59-529) ----------------------- */
typedef struct cbc_dec_block_desc {
    uint64_t in_off;       /* byte offset of the block's payload in in[]                          */
    uint64_t ref_off;      /* byte offset in the device reference of the base that is POS 1       */
    uint64_t rec_base;     /* index of the block's first output cbc_read_rec                      */
    uint64_t seq_base;     /* byte offset of the block's output bases (n_reads * seq_stride)      */
    uint32_t in_bytes;     /* payload bytes                                                       */
    uint32_t n_reads;      /* records the container index says the block holds                    */
    uint32_t read_length;  /* header read length L0 (checked against the stream)                  */
    uint32_t seq_stride;   /* bytes reserved per read in seq_out: multiple of 4, >= longest read  */
    uint32_t reserved[4];
} cbc_dec_block_desc;      /* 64 bytes */

typedef struct cbc_dec_device_batch {
    const uint8_t            *d_in;      uint64_t in_bytes;   /* payloads + >= 3 spare bytes        */
    const cbc_dec_block_desc *d_blocks;  uint32_t n_blocks;
    const uint8_t            *d_ref;     uint64_t ref_bytes;
    cbc_read_rec             *d_recs;    uint64_t n_recs;     /* out: pos (block-local), flag, rlen */
    uint8_t                  *d_seq;     uint64_t seq_bytes;  /* out: bases; >= sum + 8             */
    cbc_block_result         *d_results;                      /* nbytes = records decoded           */
    uint32_t                 *d_var_scratch; uint64_t var_scratch_words; /* n_blocks * caps.cap_var words */
    cbc_lds_caps              caps;                           /* from the container header          */
} cbc_dec_device_batch;

int  cbc_gpu_decode_blocks_device(cbc_gpu_ctx *ctx, const cbc_dec_device_batch *batch, void *hip_stream);

/* Host-buffer form: payloads in, records + bases out (recs[n_recs], seq[n_recs * seq_stride]). */
int  cbc_gpu_decode_blocks(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes,
                           cbc_dec_block_desc *blocks, uint32_t n_blocks, const cbc_lds_caps *caps,
                           cbc_read_rec *recs, uint64_t n_recs, uint8_t *seq, uint64_t seq_bytes,
                           cbc_block_result *results /* n_blocks or NULL */);
uint32_t cbc_gpu_decode_lds_bytes(const cbc_lds_caps *caps);

/* ---- whole-file stream ("compat" mode): the reference's own file format --------------------------------------
 * compress() / decompress(), src/compression.c:112-216: ONE arithmetic stream per file, models never reset.
 * `batch` is a cbc_host_batch packed with cbc_pack_opts.whole_file = 1: its `blocks` are SEGMENTS of the one
 * stream (POS not rebased).  The bytes written to `out` are the bytes `program -c 1 in.sam out ref.fa` (-DDEBUG
 * build) writes, decodable by the reference's `-x`.  One wavefront codes the stream with the general (rescaling)
 * form of every model (cbc_stream_body.h); throughput is that of one serial chain.
 * cbc_gpu_encode_stream_blocks() runs the same general-form coder over ordinary (rebased) blocks, one stream per
 * block, in cbc_gpu_encode_blocks' output format: the fallback for blocks of more than CBC_MAX_BLOCK_READS records. */
typedef struct cbc_stream_result {
    uint64_t nbytes;       /* stream bytes written / records decoded                               */
    uint32_t status;       /* CBC_ST_*                                                              */
    uint32_t fail_read;    /* index (in stream order, low 32 bits) of the record being coded        */
    uint64_t n_symbols;
} cbc_stream_result;

int  cbc_gpu_encode_stream(cbc_gpu_ctx *ctx, const cbc_host_batch *batch, uint8_t *out, uint64_t out_cap,
                           cbc_stream_result *result);
int  cbc_gpu_encode_stream_blocks(cbc_gpu_ctx *ctx, const cbc_host_batch *batch, uint8_t *out, uint64_t out_cap,
                                  uint64_t *out_offsets /* n_blocks+1 */, cbc_block_result *results /* or NULL */);

/* The decode twin of cbc_gpu_encode_stream_blocks: block b's payload (in[in_off .. + in_bytes)) is decoded as a one-contig
 * stream whose contig is the block's reference window; records and bases go where cbc_gpu_decode_blocks would put them
 * (recs[rec_base ..], seq[seq_base + r * seq_stride]; cbc_read_rec.seq_off = r * seq_stride, .tok_off = 0). */
int  cbc_gpu_decode_stream_blocks(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes, const cbc_dec_block_desc *blocks,
                                  uint32_t n_blocks, cbc_read_rec *recs, uint64_t n_recs, uint8_t *seq, uint64_t seq_bytes,
                                  cbc_block_result *results /* n_blocks or NULL */);

/* Decode a whole-file stream.  contig_off[c] / contig_len[c]: where contig c (FASTA order) starts in the uploaded
 * reference and its length; the stream names contigs only by "next one" (decompress_line, compression.c:71-108).
 * recs[r] = { POS (1-based in its contig), FLAG, length, r * seq_stride (mod 2^32), contig index }, bases of record
 * r at seq[r * seq_stride].  result->nbytes = records decoded; CBC_ST_OUT_FULL with rec_cap too small (the caller
 * retries with larger buffers). */
int  cbc_gpu_decode_stream(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes,
                           const uint64_t *contig_off, const uint64_t *contig_len, uint32_t n_contigs,
                           cbc_read_rec *recs, uint64_t rec_cap, uint8_t *seq, uint64_t seq_bytes, uint32_t seq_stride,
                           cbc_stream_result *result);
/* header read length of a whole-file stream (its first four bytes come out verbatim), 0 if too short */
uint32_t cbc_stream_read_length(const uint8_t *in, uint64_t in_bytes);

/* ---- long-read format extension (stream version 3) ------------
 * Reads up to 65535 bases, POS steps up to 2^31, edits derived from read vs reference along the CIGAR -- everything
 * the reference's limits (include/sam_block.h:38,54, src/sam_models.c:317) exclude, so this is a format of its own:
 * no reference parity exists for it (oracle/cbc_long.c is its CPU statement).  Batches come from cbc_pack_sam /
 * cbc_synth_long with cbc_pack_opts.long_reads = 1; cbc_device_batch / cbc_dec_device_batch are used as in block
 * mode (caps.cap_var is ignored).  cbc_dec_block_desc.reserved[0] = bases of the block, seq_base = where they go
 * (the decoder writes the bases compactly; cbc_read_rec.seq_off = offset inside the block). */
uint64_t cbc_gpu_long_plan_output(cbc_block_desc *blocks, uint32_t n_blocks, const cbc_read_rec *recs, uint32_t bytes_per_16_bases);
uint32_t cbc_gpu_long_lds_bytes(const cbc_lds_caps *caps);
int  cbc_gpu_long_encode_blocks_device(cbc_gpu_ctx *ctx, const cbc_device_batch *batch, void *hip_stream);
int  cbc_gpu_long_decode_blocks_device(cbc_gpu_ctx *ctx, const cbc_dec_device_batch *batch, void *hip_stream);
int  cbc_gpu_long_encode_blocks(cbc_gpu_ctx *ctx, const cbc_host_batch *batch, uint8_t *out, uint64_t out_cap,
                                uint64_t *out_offsets /* n_blocks+1 */, cbc_block_result *results /* or NULL */);
int  cbc_gpu_long_decode_blocks(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes, cbc_dec_block_desc *blocks,
                                uint32_t n_blocks, const cbc_lds_caps *caps, cbc_read_rec *recs, uint64_t n_recs,
                                uint8_t *seq, uint64_t seq_bytes, cbc_block_result *results /* or NULL */);

/* ---- 2-bit transport of bases over PCIe (layout: include/cbc_host.h cbc_2bit) ------------
 * The codec kernels read and write one byte per base in HBM; what crosses PCIe can be a quarter of that.  Host side:
 * cbc_2bit_pack() (libcbc_host).  Device side: an expand kernel (16 bases per lane: one code word in, one 16-byte
 * store out) followed by the exception runs, and the inverse pack kernel for decoded reads.
 *   cbc_gpu_upload_reference_2bit   reference as 2-bit codes + exception runs (N-runs, the pads) -> the same device
 *                                   reference cbc_gpu_upload_reference() makes (bit-identical bytes)
 *   cbc_gpu_encode_blocks_2bit      as cbc_gpu_encode_blocks, the batch's bases given as codes + runs (batch->seq may be
 *                                   NULL; batch->seq_bytes = number of bases incl. the 8 pad bytes)
 *   cbc_gpu_decode_blocks_2bit      as cbc_gpu_decode_blocks, but the bases come back as 2-bit rows: read r occupies
 *                                   words [r * row_words, (r + 1) * row_words) of codes_out (row_words = seq_stride / 16,
 *                                   seq_stride a multiple of 16), and every base of a read that is not A/C/G/T comes back
 *                                   in exc_idx[] (= r * seq_stride + position) / exc_val[]; *n_exc = how many (CBC_E_ARG if
 *                                   more than exc_cap) */
typedef struct cbc_2bit_run_dev { uint64_t start; uint32_t length; uint32_t byte; } cbc_2bit_run_dev;   /* = cbc_2bit_run */
int  cbc_gpu_upload_reference_2bit(cbc_gpu_ctx *ctx, const uint32_t *codes, uint64_t n_bases,
                                   const cbc_2bit_run_dev *runs, uint64_t n_runs);
int  cbc_gpu_encode_blocks_2bit(cbc_gpu_ctx *ctx, const cbc_host_batch *batch, const uint32_t *seq_codes,
                                const cbc_2bit_run_dev *seq_runs, uint64_t n_seq_runs,
                                uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, cbc_block_result *results);
int  cbc_gpu_decode_blocks_2bit(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes, cbc_dec_block_desc *blocks,
                                uint32_t n_blocks, const cbc_lds_caps *caps, cbc_read_rec *recs, uint64_t n_recs,
                                uint32_t *codes_out, uint64_t *exc_idx, uint8_t *exc_val, uint64_t exc_cap, uint64_t *n_exc,
                                cbc_block_result *results);

/* ---- SAM text -> packed records on the device (rules: cbc_amd/csrc/cbc_tok_core.h) ----------
 * load_sam_line(), src/sam_file_allocation.c:437-529, one GPU thread per line.  The text is copied to the device once;
 * bases and token words are produced in device memory (where cbc_gpu_encode_blocks_tokenised reads them) and the host
 * gets 16 bytes per mapped record + one change flag, from which libcbc_host's cbc_pack_from_device_tokens() cuts the
 * blocks exactly as cbc_pack_sam() does (the arrays are identical; tests).  status != 0: the first offending line
 * (bad_line, 0-based, '@' lines included) and why (CBC_TOK_* of cbc_tok_core.h: 3 = the file needs the host packer --
 * a leading soft clip or a record without MD -- 4.. = malformed input). */
typedef struct cbc_tok_record_summary { uint32_t pos; uint16_t flag, rl; uint32_t nt_ev /* words | var bound << 16 */; uint32_t line; } cbc_tok_record_summary;
typedef struct cbc_tok_result {
    uint64_t n_lines, n_recs, n_unmapped, seq_bytes /* without the 8 pad bytes */, n_tok;
    cbc_tok_record_summary *summaries;      /* host, n_recs                                                     */
    uint8_t  *rname_change;                 /* host, n_recs: RNAME differs from the previous mapped record's    */
    uint64_t *change_name_off; uint32_t *change_name_len; uint64_t n_changes;   /* host: where each new RNAME sits in the text */
    uint8_t  *d_seq; uint32_t *d_tok;       /* device: seq_bytes + 8 zero bytes; n_tok words                    */
    uint32_t status; uint64_t bad_line;
} cbc_tok_result;
int  cbc_gpu_tokenise_sam(cbc_gpu_ctx *ctx, const char *sam, uint64_t sam_len, uint64_t body_off, cbc_tok_result *out);
int  cbc_gpu_tokenise_fetch(cbc_gpu_ctx *ctx, const cbc_tok_result *t, uint8_t *seq /* seq_bytes + 8 */, uint32_t *tok /* n_tok */);
void cbc_gpu_tokenise_free(cbc_gpu_ctx *ctx, cbc_tok_result *t);
/* cbc_gpu_encode_blocks over a batch whose bases and tokens are the tokeniser's device arrays (batch->seq / tok unused) */
int  cbc_gpu_encode_blocks_tokenised(cbc_gpu_ctx *ctx, const cbc_tok_result *t, const cbc_host_batch *batch,
                                     uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, cbc_block_result *results);

/* Timing of the most recent encode launch made through this context, measured with HIP events
 * recorded on the launch stream around the kernel (valid after the stream has been synchronised). */
int  cbc_gpu_last_kernel_ms(cbc_gpu_ctx *ctx, float *ms);
/* Which register budget of the encode kernel the most recent launch used: 5 or 6 wavefronts per SIMD
 * (cbc_encode_blocks_kernel / cbc_encode_blocks_kernel_w6; chosen from blocks per CU), 0 before any launch. */
int  cbc_gpu_last_kernel_variant(cbc_gpu_ctx *ctx);
int  cbc_gpu_synchronize(cbc_gpu_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* CBC_GPU_H */
