/*
 * cbc_plan.h -- host-side sizing helpers shared by the C ABI implementation (cbc_gpu.hip) and the
 * test emulation driver: LDS bytes per wavefront and the worst-case payload area per block.
 */
#ifndef CBC_PLAN_H
#define CBC_PLAN_H

#include <stdint.h>
#include <string.h>
#include "../../include/cbc_gpu.h"

/* LDS layout constants, in 32-bit words (the kernel bodies take them from here).
 * [0, CBC_PLAN_TABLE_WORDS): model tables shared by encoder and decoder bodies:
 *   256 rlength, 256 snps, 256 indels, 2 x CBC_CAP_NAME contig-name pairs, CBC_BLOOM_WORDS Bloom filter,
 *   2 x CBC_P0_WORDS var events of the "p = 0" contexts (one array per strand, two 16-bit events per word) */
#define CBC_BLOOM_WORDS 256u                        /* 8192 bits, two hash functions (power of two) */
#define CBC_P0_WORDS    512u                        /* per strand: 8 buckets (d & 7) of 64 words = 128 events of 16 bits each */
#define CBC_P0_BUCKET_WORDS 64u                     /* one LDS load per lane scans a whole bucket */
#define CBC_P0_CAP      128u                        /* events per bucket */
#define CBC_PLAN_TABLE_WORDS (768u + 2u * CBC_CAP_NAME + CBC_BLOOM_WORDS + 2u * CBC_P0_WORDS)
#ifndef CBC_BATCH_SLOTS
#define CBC_BATCH_SLOTS 2u                         /* encoder hand-off ring depth (power of two; 2 measured as good as 4) */
#endif
#define CBC_BATCH_WORDS 200u                       /* 64 lo + 64 cnt + 64 n + {len, flags, status, record, match mask x2} */
#ifndef CBC_POS_IDX_WORDS
#define CBC_POS_IDX_WORDS 256u                      /* 0: no index table (A/B) */
#endif
#define CBC_RING_WORDS  256u                       /* output bit ring of the coder wave (power of two) */
/* encoder: tables, hand-off ring, its two counters (8 words), output ring; then 3 x cap_pos */
#define CBC_PLAN_LDS_FIXED_WORDS (CBC_PLAN_TABLE_WORDS + CBC_BATCH_SLOTS * CBC_BATCH_WORDS + 8u + CBC_RING_WORDS)

/* LDS per wavefront: fixed tables + the POS alphabet.  The var-event list is NOT in LDS: it lives in
 * global memory behind the block's payload area (encode) / in the decode scratch, so caps->cap_var
 * only sizes those areas. */
static inline uint32_t cbc_plan_lds_bytes(const cbc_lds_caps *caps)
{
    /* pos_val, pos_occ, pos_pre; + the alphabet index of every POS delta below CBC_POS_IDX_WORDS (one LDS load per lane
     * instead of a walk over the alphabet in fixed_group()) */
    return 4u * (CBC_PLAN_LDS_FIXED_WORDS + 3u * caps->cap_pos + CBC_POS_IDX_WORDS);
}

/* decoder: tables, 80 words scratch read + 256 deletion positions, 512 pos_alpha histograms, 256 insertions;
 * then 2 x cap_pos.  It does not carry the encoder's rings: its LDS footprint decides how many blocks
 * a CU holds, and with one wavefront per block that is the decoder's only latency hiding. */
#define CBC_PLAN_DLDS_SCRATCH_WORDS 336u
#define CBC_PLAN_DLDS_FIXED_WORDS (CBC_PLAN_TABLE_WORDS + CBC_PLAN_DLDS_SCRATCH_WORDS + 768u)
static inline uint32_t cbc_plan_dec_lds_bytes(const cbc_lds_caps *caps)
{
    return 4u * (CBC_PLAN_DLDS_FIXED_WORDS + 2u * caps->cap_pos);
}

/* Upper bound on the payload of a block.  Every model total stays below 2^20, so one coded symbol
 * costs < 20 bits; 3 bytes per symbol leaves slack for the 26-bit flush.  Symbols per record:
 * same_ref 1 + rlength 4 + pos <=5 + flag 1 + match 1 = 12, plus for an imperfect read <= 4 count
 * symbols and 2 per edit (var + chars); stream header 136, contig name + sentinel <= 2*CAP_NAME. */
static inline uint64_t cbc_plan_output(cbc_block_desc *blocks, uint32_t n_blocks,
                                       const cbc_read_rec *recs, const uint32_t *tok)
{
    uint64_t off = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        cbc_block_desc *bd = &blocks[b];
        uint64_t nsym = 136u + 2u * CBC_CAP_NAME + 16ull * bd->n_reads, nev = 0;
        for (uint32_t r = 0; r < bd->n_reads; r++) {
            const cbc_read_rec *rr = &recs[bd->rec_base + r];
            const uint32_t *t = tok + bd->tok_base + rr->tok_off;
            uint32_t n_cig = t[0] & 0xffffu, n_md = t[0] >> 16;
            uint64_t ev = n_md;
            for (uint32_t k = 0; k < n_cig; k++) if ((t[2 + k] & 15u) != CBC_OP_M) ev += t[2 + k] >> 4;
            nsym += 2 * ev; nev += ev;
        }
        /* [0, payload_cap): payload; [payload_cap, out_cap): the block's var-event list (one word per
         * var symbol), kept in HBM/L2 instead of LDS */
        uint64_t payload_cap = (3 * nsym + 256 + 255) & ~255ull;
        uint64_t cap = payload_cap + ((4 * (nev + 64) + 255) & ~255ull);
        if (cap > 0xffffff00ull) { cap = 0xffffff00ull; payload_cap = cap / 2; payload_cap &= ~255ull; }
        bd->out_off = off; bd->out_cap = (uint32_t)cap; bd->reserved = (uint32_t)payload_cap;
        off += cap;
    }
    return off;
}

/* The same bound without reading a single record: the packer cuts blocks so that none holds more than caps->cap_var - 1
 * edit events, so cap_var bounds every block's event count.  O(blocks) instead of O(records + tokens) on the host -- the
 * per-record form cost more than the whole device pipeline of a cfg2-sized call (2 x 15 ms of 48, round 3). */
static inline uint64_t cbc_plan_output_caps(cbc_block_desc *blocks, uint32_t n_blocks, const cbc_lds_caps *caps)
{
    uint64_t off = 0;
    const uint64_t nev = caps->cap_var;
    for (uint32_t b = 0; b < n_blocks; b++) {
        cbc_block_desc *bd = &blocks[b];
        const uint64_t nsym = 136u + 2u * CBC_CAP_NAME + 16ull * bd->n_reads + 2 * nev;
        uint64_t payload_cap = (3 * nsym + 256 + 255) & ~255ull;
        uint64_t cap = payload_cap + ((4 * (nev + 64) + 255) & ~255ull);
        if (cap > 0xffffff00ull) { cap = 0xffffff00ull; payload_cap = cap / 2; payload_cap &= ~255ull; }
        bd->out_off = off; bd->out_cap = (uint32_t)cap; bd->reserved = (uint32_t)payload_cap;
        off += cap;
    }
    return off;
}

#endif
