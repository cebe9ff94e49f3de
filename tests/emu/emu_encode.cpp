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
#include "../../cbc_amd/csrc/cbc_tok_core.h"

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
/* The two wavefronts of a block as two host threads: the model / coder roles with their LDS hand-off ring (publish, pull,
 * seg_consume, the group batches), which the fused emulation above never enters.  The counters are acquire / release
 * atomics on the shared table memory, exactly the ordering the kernel asks of LDS. */
#include <thread>
#include <atomic>
#include <cstdlib>
#include <chrono>
/* CBC_EMU_JITTER=<seed>: random pauses around every hand-off operation, a different sequence per thread -- explores
 * interleavings in which one wavefront runs far ahead of the other (tests/test_emu_parity.py) */
static void emu_jitter()
{
    static const char *env = getenv("CBC_EMU_JITTER");
    if (!env) return;
    static std::atomic<unsigned> thread_ids{0};
    static thread_local uint64_t st = 0;
    if (!st) st = 0x9E3779B97F4A7C15ull * (strtoull(env, NULL, 10) + 1u) + 0xD1B54A32D192ED03ull * (thread_ids.fetch_add(1) + 1u);
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    const unsigned r = (unsigned)(st >> 40) & 1023u;
    if (r < 32u) std::this_thread::sleep_for(std::chrono::microseconds(20u * r));      /* now and then a long pause */
    else if (r < 256u) std::this_thread::yield();
}
struct WaveEmu2 : WaveEmu {
    static uint32_t ctl_load(const uint32_t *p) { emu_jitter(); return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
    static void ctl_store(uint32_t *p, uint32_t v) { emu_jitter(); __atomic_store_n(p, v, __ATOMIC_RELEASE); emu_jitter(); }
    static void nap() { emu_jitter(); std::this_thread::yield(); }
    static void barrier()
    {
        static std::atomic<unsigned> arrived{0};
        const unsigned ticket = arrived.fetch_add(1, std::memory_order_acq_rel);
        const unsigned target = (ticket / 2u + 1u) * 2u;         /* two threads per block, blocks run one after the other */
        while (arrived.load(std::memory_order_acquire) < target) std::this_thread::yield();
    }
};
extern "C" __attribute__((visibility("default")))
int emu_encode_blocks_two_wave(const cbc_device_batch *b)
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
        std::vector<uint32_t> lds(words, 0xdeadbeefu);
        std::thread model([&]() { cbc_encode_stream<WaveEmu2, CBC_ROLE_MODEL>(A, blk, lds.data()); });
        cbc_encode_stream<WaveEmu2, CBC_ROLE_CODER>(A, blk, lds.data());
        model.join();
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
    std::vector<uint32_t> scratch((size_t)A->n_blocks * CBC_LONG_SCRATCH_WORDS + 64, 0xdeadbeefu);   /* the kernel zeroes what it uses */
    cbc_long_args B = *A;
    B.scratch = scratch.data();
    for (uint32_t blk = 0; blk < B.n_blocks; blk++) {
        std::vector<uint32_t> lds(cbc_long_lds_bytes(B.cap_pos) / 4, 0xdeadbeefu);
        cbc_long_encode<WaveEmu, CBC_ROLE_FUSED>(B, blk, lds.data());
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
    std::vector<uint32_t> scratch((size_t)b->n_blocks * CBC_LONG_TABLE_WORDS + 64, 0xdeadbeefu);
    A.var_scratch = scratch.data(); A.var_scratch_words = (uint64_t)b->n_blocks * CBC_LONG_TABLE_WORDS;
    for (uint32_t blk = 0; blk < b->n_blocks; blk++) {
        std::vector<uint32_t> lds(cbc_long_dec_lds_bytes(A.cap_pos) / 4, 0xdeadbeefu);
        cbc_long_decode<WaveEmu>(A, blk, lds.data());
    }
    return g_emu_errors ? -100 : 0;
}

/* The device tokeniser's per-line / per-record functions (cbc_tok_core.h) run line by line on the CPU, producing
 * what cbc_gpu_tokenise_sam() produces: summaries, change flags + names, bases, token words.  Returns the first
 * non-zero status (and its line) or 0. */
extern "C" __attribute__((visibility("default")))
int emu_tokenise(const uint8_t *sam, uint64_t len, uint64_t body_off, cbc_tok_summary *sum, uint8_t *chg, uint64_t *chg_off, uint32_t *chg_len,
                 uint8_t *seq, uint32_t *tok, uint64_t *counts /* n_lines, n_recs, n_unmapped, seq_bytes, n_tok, n_changes, bad_line */)
{
    uint64_t n_lines = 0, n_recs = 0, n_unm = 0, sb = 0, ntok = 0, nchg = 0;
    uint64_t prev_off = 0; uint32_t prev_len = 0; int have_prev = 0;
    std::vector<cbc_tok_line> split;                       /* every line's column split, as the device keeps it (NULL source: header lines) */
    std::vector<uint8_t> in_body;
    for (uint64_t b = 0; b < len; n_lines++) {
        uint64_t e = b;
        while (e < len && sam[e] != '\n') e++;
        if (e < len) e++;
        cbc_tok_line L;
        memset(&L, 0, sizeof L);
        uint32_t st;
        if (b < body_off) st = CBC_TOK_SKIP; else { cbc_tok_split(sam, b, e, &L); st = L.status; }
        split.push_back(L); in_body.push_back(b >= body_off);
        uint32_t nt = 0, ev = 0;
        if (st == CBC_TOK_OK && !L.has_md) {
            uint64_t md = 0; uint32_t md_len = 0;
            st = cbc_tok_md_source(n_lines, [&](uint64_t j) -> const cbc_tok_line * { return in_body[j] ? &split[j] : (const cbc_tok_line *)0; }, &md, &md_len);
            L.md = md; L.md_len = md_len;
        }
        if (st == CBC_TOK_OK) st = cbc_tok_record(sam, &L, tok + ntok, &nt, &ev);
        if (st >= CBC_TOK_NEEDS_HOST) { counts[6] = n_lines; return (int)st; }
        if (st == CBC_TOK_UNMAPPED) n_unm++;
        if (st == CBC_TOK_OK) {
            cbc_tok_summary s; s.pos = (uint32_t)L.pos; s.flag = (uint16_t)L.flag; s.rl = (uint16_t)L.seq_len; s.nt_ev = nt | (ev << 16); s.line = (uint32_t)n_lines;
            sum[n_recs] = s;
            memcpy(seq + sb, sam + L.seq, L.seq_len);
            int change = !have_prev || prev_len != L.rname_len || memcmp(sam + prev_off, sam + L.rname, prev_len) != 0;
            chg[n_recs] = (uint8_t)change;
            if (change) { chg_off[nchg] = L.rname; chg_len[nchg] = L.rname_len; nchg++; }
            prev_off = L.rname; prev_len = L.rname_len; have_prev = 1;
            n_recs++; sb += L.seq_len; ntok += nt;
        }
        b = e;
    }
    counts[0] = n_lines; counts[1] = n_recs; counts[2] = n_unm; counts[3] = sb; counts[4] = ntok; counts[5] = nchg; counts[6] = 0;
    return 0;
}
