/*
 * emu_encode.cpp -- runs the HIP kernel BODY (cbc_amd/csrc/cbc_encode_body.h) on the CPU through
 * the lock-step wave emulation.  DEBUGGING / TEST AID ONLY (see wave_emu.h): it exists so that the
 * kernel's indexing and control flow can be checked under ASan/UBSan and against the oracle before
 * anything is launched on a real GPU.  The product never links or loads this.
 */
#include <vector>
#include "wave_emu.h"
#include "../../cbc_amd/csrc/cbc_encode_body.h"
#include "../../cbc_amd/csrc/cbc_decode_body.h"
#include "../../cbc_amd/csrc/cbc_plan.h"
#include "../../cbc_amd/csrc/cbc_stream_body.h"
#include "../../cbc_amd/csrc/cbc_long_body.h"

static int g_emu_errors = 0;
#ifdef CBC_EMU_TRACE
extern "C" void cbc_emu_trace(uint32_t read, uint32_t lo, uint32_t cnt, uint32_t n)
{
    static FILE *f = NULL;
    if (!f) { const char *p = getenv("CBC_EMU_TRACE_FILE"); f = fopen(p ? p : "/tmp/cbc_emu_trace.txt", "w"); }
    if (f) { fprintf(f, "%u %u %u %u\n", read, lo, cnt, n); fflush(f); }
}
#endif
extern "C" void emu_oob(const char *what) { fprintf(stderr, "[emu] invariant violated: %s\n", what); g_emu_errors++; }

extern "C" __attribute__((visibility("default")))
int emu_encode_blocks(const cbc_device_batch *b)
{
    cbc_enc_args A;
    A.recs = b->d_recs; A.seq = b->d_seq; A.tok = b->d_tok; A.names = b->d_names; A.blocks = b->d_blocks;
    A.ref = b->d_ref; A.out = b->d_out; A.results = b->d_results;
    A.ref_bytes = b->ref_bytes; A.out_bytes = b->out_bytes; A.seq_bytes = b->seq_bytes; A.n_tok = b->n_tok;
    A.n_recs = b->n_recs; A.n_blocks = b->n_blocks; A.cap_pos = b->caps.cap_pos; A.cap_var = b->caps.cap_var;
    A.names_bytes = 0x7fffffffu;
    g_emu_errors = 0;
    uint32_t words = cbc_plan_lds_bytes(&b->caps) / 4;
    for (uint32_t blk = 0; blk < b->n_blocks; blk++) {
        std::vector<uint32_t> lds(words, 0xdeadbeefu);      /* LDS is not zero-initialised on the GPU either */
        cbc_encode_stream<WaveEmu, CBC_ROLE_FUSED>(A, blk, lds.data());
    }
    return g_emu_errors ? -100 : 0;
}
extern "C" __attribute__((visibility("default")))
uint64_t emu_plan_output(cbc_block_desc *blocks, uint32_t n_blocks, const cbc_read_rec *recs, const uint32_t *tok)
{ return cbc_plan_output(blocks, n_blocks, recs, tok); }

extern "C" __attribute__((visibility("default")))
int emu_decode_blocks(const cbc_dec_device_batch *b)
{
    cbc_dec_args A;
    A.in = b->d_in; A.blocks = b->d_blocks; A.ref = b->d_ref; A.recs = b->d_recs; A.seq = b->d_seq; A.results = b->d_results;
    A.in_bytes = b->in_bytes; A.ref_bytes = b->ref_bytes; A.n_recs = b->n_recs; A.seq_bytes = b->seq_bytes;
    A.n_blocks = b->n_blocks; A.cap_pos = b->caps.cap_pos; A.cap_var = b->caps.cap_var;
    A.var_scratch = b->d_var_scratch; A.var_scratch_words = b->var_scratch_words;
    g_emu_errors = 0;
    uint32_t words = cbc_plan_dec_lds_bytes(&b->caps) / 4;
    for (uint32_t blk = 0; blk < b->n_blocks; blk++) {
        std::vector<uint32_t> lds(words, 0xdeadbeefu);
        cbc_decode_stream<WaveEmu>(A, blk, lds.data());
    }
    return g_emu_errors ? -100 : 0;
}

/* whole-file stream (cbc_stream_body.h): args as the HIP library builds them; vtab zero-filled by the caller */
extern "C" __attribute__((visibility("default")))
int emu_encode_stream(const cbc_stream_args *A)
{
    g_emu_errors = 0;
    cbc_stream_caps caps = { A->cap_pos, A->cap_name };
    uint32_t words = cbc_stream_lds_bytes(&caps) / 4;
    uint32_t n_streams = A->per_segment ? A->n_segs : 1u;
    for (uint32_t s = 0; s < n_streams; s++) {
        std::vector<uint32_t> lds(words, 0xdeadbeefu);
        if (A->per_segment) memset(A->vtab, 0, (size_t)CBC_VTAB_WORDS * 4);
        cbc_encode_whole<WaveEmu>(*A, s, 0u, lds.data());
    }
    return g_emu_errors ? -100 : 0;
}

extern "C" __attribute__((visibility("default")))
int emu_decode_stream(const cbc_dstream_args *A)
{
    g_emu_errors = 0;
    cbc_stream_caps caps = { A->cap_pos, A->cap_name };
    std::vector<uint32_t> lds(cbc_stream_lds_bytes(&caps) / 4, 0xdeadbeefu);
    cbc_decode_whole<WaveEmu>(*A, lds.data());
    return g_emu_errors ? -100 : 0;
}

/* long-read format (cbc_long_body.h) */
extern "C" __attribute__((visibility("default")))
int emu_long_encode_blocks(const cbc_long_args *A)
{
    g_emu_errors = 0;
    for (uint32_t blk = 0; blk < A->n_blocks; blk++) {
        std::vector<uint32_t> lds(cbc_long_lds_bytes(A->cap_pos) / 4, 0xdeadbeefu);
        cbc_long_encode<WaveEmu>(*A, blk, lds.data());
    }
    return g_emu_errors ? -100 : 0;
}
extern "C" __attribute__((visibility("default")))
int emu_long_decode_blocks(const cbc_dec_device_batch *b)
{
    cbc_dec_args A;
    memset(&A, 0, sizeof A);
    A.in = b->d_in; A.blocks = b->d_blocks; A.ref = b->d_ref; A.recs = b->d_recs; A.seq = b->d_seq; A.results = b->d_results;
    A.in_bytes = b->in_bytes; A.ref_bytes = b->ref_bytes; A.n_recs = b->n_recs; A.seq_bytes = b->seq_bytes;
    A.n_blocks = b->n_blocks; A.cap_pos = b->caps.cap_pos; A.cap_var = b->caps.cap_var;
    g_emu_errors = 0;
    for (uint32_t blk = 0; blk < b->n_blocks; blk++) {
        std::vector<uint32_t> lds(cbc_long_lds_bytes(A.cap_pos) / 4, 0xdeadbeefu);
        cbc_long_decode<WaveEmu>(A, blk, lds.data());
    }
    return g_emu_errors ? -100 : 0;
}
