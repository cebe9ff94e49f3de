/*
 * cbc_host.h -- host side of the cbc hot path (libcbc_host.so, plain C): the record packer that
 * stands where the reference's tokeniser and FASTA loader stand, the block container, and the
 * seeded synthetic workload generator used by bench.py and the parity tests.
 *
 *   load_sam_line()              src/sam_file_allocation.c:437-529  -> cbc_pack_sam()
 *   get_read_length()            src/sam_file_allocation.c:26-79    -> cbc_packed.read_length
 *   store_reference_in_memory()  src/read_decompression.c:17-53     -> cbc_packed.ref
 *
 * No arithmetic coding happens on the host: payload bytes only ever come from the HIP kernels.
 */
#ifndef CBC_HOST_H
#define CBC_HOST_H

#include "cbc_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cbc_block_info {
    uint32_t contig;        /* index of the contig (order of appearance = FASTA record order)   */
    uint32_t n_reads;
    uint64_t window_start;  /* 0-based offset in the contig of the base that is local POS 1      */
    uint64_t n_bases;       /* sum of SEQ lengths                                                */
} cbc_block_info;

typedef struct cbc_contig_info {
    uint64_t ref_off;       /* offset of the contig's first base in cbc_packed.ref               */
    uint64_t length;
    uint32_t name_off;      /* offset of its NUL-terminated SAM RNAME in cbc_packed.names        */
    uint32_t reserved;
} cbc_contig_info;

typedef struct cbc_packed {
    cbc_read_rec    *recs;    uint64_t n_recs;
    uint8_t         *seq;     uint64_t seq_bytes;     /* includes 8 trailing pad bytes            */
    uint32_t        *tok;     uint64_t n_tok;
    uint8_t         *names;   uint32_t names_bytes;
    cbc_block_desc  *blocks;  uint32_t n_blocks;
    cbc_block_info  *info;
    cbc_contig_info *contigs; uint32_t n_contigs;
    uint8_t         *ref;     uint64_t ref_bytes;     /* contigs, each + CBC_REF_PAD zero bytes   */
    cbc_lds_caps     caps;
    uint32_t         read_length;                     /* header read length L0                    */
    uint64_t         n_bases;
    uint64_t         n_skipped_unmapped;
    uint32_t         max_read_len;                    /* longest SEQ packed                        */
    uint32_t         whole_file;                      /* 1: packed for the whole-file stream (cbc_pack_opts.whole_file); 2: long-read format */
    /* allocation bookkeeping (private) */
    uint64_t cap_recs, cap_seq, cap_tok, cap_ref; uint32_t cap_names, cap_blocks, cap_contigs;
} cbc_packed;

typedef struct cbc_pack_opts {
    uint32_t block_reads;   /* records per block (default 4096, max CBC_MAX_BLOCK_READS)          */
    uint32_t max_cap_pos;   /* cut a block before it needs more POS-delta entries (default 2048)  */
    uint32_t max_cap_var;   /* cut a block before it can hold more var symbols (default 8192)     */
    uint32_t var_length;    /* reference's -l: header read length = max over the file             */
    uint32_t n_threads;     /* text-path worker threads: 0 = one per online CPU, 1 = serial       */
    uint32_t whole_file;    /* 1 = "compat" mode: pack for ONE stream per file, the reference's own output format
                             * (compress(), src/compression.c:112-170): POS is NOT rebased, `blocks` become SEGMENTS
                             * (consecutive records of one contig; a new segment only at a contig change or when a
                             * 32-bit offset would overflow), caps.cap_pos counts the distinct POS steps of the whole
                             * file.  Inputs the reference cannot represent are refused: a POS step of 5 000 000 or
                             * more (MAX_ALPHA, sam_block.h:54), more than CBC_CAP_FLAG distinct FLAG values. */
    uint32_t long_reads;    /* 1 = the long-read format extension (stream version 3):
                             * SEQ up to CBC_LONG_MAX_READ_LEN bases and SAM lines of any length (the reference's limits
                             * are 252 bases / 1023 bytes), tokens = the CIGAR only (MD is ignored: edits are derived from
                             * read vs reference), blocks cut at block_reads (default 64) or CBC_LONG_BLOCK_BASES bases. */
} cbc_pack_opts;

#define CBC_LONG_MAX_READ_LEN 65535u
#define CBC_LONG_BLOCK_BASES  (1u << 20)

void cbc_pack_default_opts(cbc_pack_opts *o);

/* Tokenise SAM text + FASTA text into packed blocks.  errbuf receives a message on failure. */
int  cbc_pack_sam(const char *sam, size_t sam_len, const char *fasta, size_t fasta_len,
                  const cbc_pack_opts *opts, cbc_packed **out, char *errbuf, size_t errlen);
void cbc_packed_free(cbc_packed *p);

/* Seeded synthetic workload: one uniform-ACGT contig of `contig_len`
 * bases named `name`, `n_reads` reads of `read_len` bases at sorted uniform positions, FLAG in
 * {0,16}, per-base substitution rate `sub_rate`, `indel_frac` of reads with one 1..3-base
 * insertion or deletion >= 10 bases from either end, CIGAR M/I/D, MD:Z then NM:i.
 * Records go through the same path as cbc_pack_sam().  If sam_out/fasta_out are non-NULL the
 * equivalent SAM and FASTA text is returned too (malloc'ed, caller frees with cbc_free). */
typedef struct cbc_synth_opts {
    uint64_t seed;
    uint64_t contig_len;
    uint64_t n_reads;
    uint32_t read_len;
    double   sub_rate;
    double   indel_frac;
    const char *name;
} cbc_synth_opts;

int  cbc_synth_packed(const cbc_synth_opts *so, const cbc_pack_opts *po, cbc_packed **out,
                      char **sam_out, size_t *sam_len, char **fasta_out, size_t *fasta_len,
                      char *errbuf, size_t errlen);

/* cfg5 workload: long reads of `read_len` bases at sorted uniform positions on one uniform-ACGT
 * contig, FLAG in {0,16}; every aligned base is, with probability `edit_rate`, the site of one edit: substitution,
 * 1-base insertion or 1-base deletion (one third each).  Packed for the long-read format (long_reads = 1); the
 * generator runs on `po->n_threads` threads.  sam_out / fasta_out as in cbc_synth_packed (small cases only). */
int  cbc_synth_long(const cbc_synth_opts *so, const cbc_pack_opts *po, cbc_packed **out,
                    char **sam_out, size_t *sam_len, char **fasta_out, size_t *fasta_len,
                    char *errbuf, size_t errlen);
void cbc_free(void *p);

/* ---- block container (block mode of the CLI) ----------------------------------------------
 * magic "CBCB", version, header read length, contig table, block index, then the payloads.
 * Every payload follows the reference's stream grammar byte for byte. */
#define CBC_CONTAINER_MAGIC 0x42434243u   /* "CBCB" little-endian */
#define CBC_CONTAINER_VERSION 2u
#define CBC_CONTAINER_VERSION_LONG 3u     /* long-read format: same layout, per-block base counts in the index */

int64_t cbc_container_size(const cbc_packed *p, const uint64_t *out_offsets);
int64_t cbc_container_write(const cbc_packed *p, const uint8_t *payloads, const uint64_t *out_offsets,
                            uint8_t *dst, uint64_t dst_cap);

/* ---- the serial half of packing after the DEVICE tokeniser ------------------------------
 * cbc_gpu_tokenise_sam() (libcbc_gpu) turns the SAM text into bases, token words and one 16-byte summary per mapped
 * record; what is inherently serial -- contig numbering in file order, block cutting -- happens here, with the very
 * code path cbc_pack_sam() uses, so the resulting cbc_packed is identical to cbc_pack_sam()'s (tests).  seq / tok:
 * host copies of the tokeniser's arrays (then owned by the result) or NULL when they stay on the device (the result
 * then carries the sizes only).  summaries / rname_change / change_name_*: cbc_tok_result's arrays. */
int  cbc_pack_from_device_tokens(const char *sam, size_t sam_len, const char *fasta, size_t fasta_len, const cbc_pack_opts *opts,
                                 const void *summaries /* cbc_tok_record_summary[n_recs] */, const uint8_t *rname_change,
                                 const uint64_t *change_name_off, const uint32_t *change_name_len, uint64_t n_recs, uint64_t n_unmapped,
                                 uint8_t *seq, uint64_t seq_bytes, uint32_t *tok, uint64_t n_tok,
                                 cbc_packed **out, char *errbuf, size_t errlen);
/* offset of the first record line (behind the '@' header lines): the tokeniser's body_off */
uint64_t cbc_sam_body_offset(const char *sam, size_t sam_len);

/* ---- 2-bit transport of bases ------------------------------------------------
 * Bases travel to the device (reference, reads) and back (decoded reads) at 2 bits each: A C G T = 0 1 2 3, sixteen
 * bases per 32-bit word, base i in bits 2 (i & 15) of word i >> 4.  Every byte that is not one of 'A' 'C' 'G' 'T'
 * ('N', other IUPAC letters, the zero pad behind a contig) is an EXCEPTION, kept exactly: runs of one repeated byte
 * as (start, length, byte) -- an N-run of a chromosome is one entry -- so that unpacking restores the byte array
 * bit for bit (the match test compares bytes: src/read_compression.c:291-296).  cbc_2bit_pack runs on n_threads
 * threads (0 = one per CPU); cbc_2bit_unpack is the host inverse (the device inverse is cbc_gpu_expand_2bit). */
typedef struct cbc_2bit_run { uint64_t start; uint32_t length; uint32_t byte; } cbc_2bit_run;
typedef struct cbc_2bit {
    uint32_t     *codes;   uint64_t n_bases;     /* (n_bases + 15) / 16 words */
    cbc_2bit_run *runs;    uint64_t n_runs;      /* sorted by start, disjoint  */
} cbc_2bit;
int  cbc_2bit_pack(const uint8_t *bases, uint64_t n_bases, uint32_t n_threads, cbc_2bit **out);
int  cbc_2bit_unpack(const cbc_2bit *p, uint8_t *bases /* n_bases */);
void cbc_2bit_free(cbc_2bit *p);

/* ---- sharding over devices ---------------------------------------------------------
 * Blocks are independent streams; a contig's blocks share its reference, so whole contigs are dealt to the parts,
 * largest first, each to the part with the least records so far (cfg4: chromosome-sharded).  part_of_contig[c]
 * receives the part of contig c.  Deterministic: ties go to the lower part index. */
int  cbc_assign_contigs(const cbc_packed *p, uint32_t n_parts, uint32_t *part_of_contig /* n_contigs */);
/* host twin of cbc_gpu_checksum_device (include/cbc_gpu.h, CBC_CHECKSUM_TERM): what the receiving side of the bitstream
 * gather recomputes when the bytes arrive in host memory */
uint64_t cbc_checksum64(const uint8_t *bytes, uint64_t n);

/* ---- the reference alone (whole-file stream decode: the stream names contigs only by "next one") ----
 * FASTA text -> upper-cased contig bases, each + CBC_REF_PAD zero bytes, and the contig table in file order
 * (store_reference_in_memory, src/read_decompression.c:17-53).  Free with cbc_reference_free. */
typedef struct cbc_reference {
    uint8_t *bases; uint64_t n_bytes;
    uint64_t *contig_off, *contig_len; uint32_t n_contigs;
} cbc_reference;
int  cbc_reference_load(const char *fasta, size_t fasta_len, uint32_t n_threads, cbc_reference **out, char *errbuf, size_t errlen);
void cbc_reference_free(cbc_reference *r);

/* ---- unpack side: container + FASTA -> decode launch plan -> text ------------------------- */
typedef struct cbc_unpack_plan {
    cbc_dec_block_desc *blocks;   uint32_t n_blocks;
    const uint8_t      *payloads; uint64_t payload_bytes;   /* points into the caller's container blob */
    uint8_t            *ref;      uint64_t ref_bytes;       /* owned: contigs + pads, as the packer lays them out */
    uint64_t           *window_start;                        /* per block: add to a decoded POS for the contig POS */
    cbc_lds_caps        caps;
    uint32_t            read_length, seq_stride;
    uint64_t            n_recs;
    uint32_t            long_reads;                          /* container version 3: blocks[b].reserved[0] = bases of block b, */
    uint32_t            max_read_len;                        /* blocks[b].seq_base = where they start in the output (compact)  */
    uint64_t            seq_total;                           /* bytes of bases the decode writes (+ 8 spare)                   */
} cbc_unpack_plan;

int     cbc_unpack_plan_create(const uint8_t *blob, uint64_t len, const char *fasta, size_t fasta_len,
                               cbc_unpack_plan **out, char *errbuf, size_t errlen);
void    cbc_unpack_plan_free(cbc_unpack_plan *u);
int64_t cbc_unpack_write_text(const cbc_unpack_plan *u, const cbc_read_rec *recs, const uint8_t *seq,
                              char *dst, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
