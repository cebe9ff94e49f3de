/*
 * cbc_gpu.hip -- libcbc_gpu.so: the HIP kernels and the C ABI of include/cbc_gpu.h (gfx950 only).
 *
 * Launch shape: one workgroup per block = per arithmetic stream (encode: 128 threads = model wavefront +
 * coder wavefront; decode: 64 threads); the model tables live in dynamic LDS (cbc_gpu_lds_bytes()), so
 * workgroups per CU = 160 KiB / that.  The grid is the number of blocks (thousands) >> 256 CUs.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <new>
#include <time.h>
#include <dlfcn.h>
#include <rccl/rccl.h>              /* types and prototypes only: librccl.so (573 MB) is loaded when a group is created */

#include "../../include/cbc_gpu.h"
#include "cbc_wave_gpu.h"
#include "cbc_encode_body.h"
#include "cbc_decode_body.h"
#include "cbc_plan.h"
#include "cbc_stream_body.h"
#include "cbc_long_body.h"
#include "cbc_tokenise.h"

#define API extern "C" __attribute__((visibility("default")))
/* internal marker: "use the context's own stream" (host-buffer entry points only) */
#define CBC_CTX_STREAM ((void *)(uintptr_t)1)

/* ------------------------------------------------------------------------------------------------
 * kernels
 * ---------------------------------------------------------------------------------------------- */
extern __shared__ uint32_t cbc_lds[];

/* One workgroup = one block = one arithmetic stream, coded by TWO wavefronts: wavefront 0 (model)
 * runs the match test and the edit models and sends END-terminated segments of the symbol stream
 * through an LDS ring; wavefront 1 (coder) owns the per-record models, computes their symbols one
 * lane per record and runs the range coder (cbc_encode_body.h, CbcEnc::publish / pull).
 * Workgroups are dealt round-robin to the 8 XCDs; blocks of one contig are neighbours in the
 * batch and share nothing but read-only reference lines, so the identity map is kept and the
 * per-XCD L2s each see a strided slice of the record stream. */
static __device__ __forceinline__ void cbc_encode_block(const cbc_enc_args &A)
{
    uint32_t blk = blockIdx.x;
    if (blk >= A.n_blocks) return;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wid == 0u) cbc_encode_stream<WaveGPU, CBC_ROLE_MODEL>(A, blk, cbc_lds);
    else cbc_encode_stream<WaveGPU, CBC_ROLE_CODER>(A, blk, cbc_lds);
}

/* Two register budgets of the same code.  Up to ten blocks per CU (cfg2: 9.5) everything is resident at
 * 5 wavefronts per SIMD and the kernel is latency-bound: the 84-register build is the faster one.  With
 * more blocks than that (cfg3-sized input) the CU is throughput-bound and a sixth wavefront per SIMD (80
 * registers, one spilled) gains 6 %; it costs 5 % at cfg2.  One wave fewer than 5 costs 25-30 %. */
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(5)))
cbc_encode_blocks_kernel(cbc_enc_args A) { cbc_encode_block(A); }

__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(6)))
cbc_encode_blocks_kernel_w6(cbc_enc_args A) { cbc_encode_block(A); }

__global__ void __launch_bounds__(64)
cbc_decode_blocks_kernel(cbc_dec_args A)
{
    uint32_t blk = blockIdx.x;
    if (blk >= A.n_blocks) return;
    cbc_decode_stream<WaveGPU>(A, blk, cbc_lds);
}

/* Whole-file stream / general-form fallback (cbc_stream_body.h): one wavefront per stream.  Workgroup w codes streams
 * w, w + gridDim, ... with var table w of the pool, which it re-zeroes between streams. */
__global__ void __launch_bounds__(64)
cbc_encode_whole_kernel(cbc_stream_args A)
{
    const uint32_t n_streams = A.per_segment ? A.n_segs : 1u;
    for (uint32_t s = blockIdx.x; s < n_streams; s += gridDim.x) {
        if (s != blockIdx.x) {
            uint4 *t = (uint4 *)(A.vtab + (uint64_t)blockIdx.x * CBC_VTAB_WORDS);
            for (uint64_t i = threadIdx.x; i < CBC_VTAB_WORDS / 4; i += 64) t[i] = make_uint4(0, 0, 0, 0);
            __threadfence();
        }
        cbc_encode_whole<WaveGPU>(A, s, blockIdx.x, cbc_lds);
    }
}

__global__ void __launch_bounds__(64)
cbc_decode_whole_kernel(cbc_dstream_args A) { cbc_decode_whole<WaveGPU>(A, cbc_lds); }

/* long-read format (cbc_long_body.h): encode = three wavefronts per block (model, coder, walker), decode = one */
#ifndef CBC_LONG_ENC_WAVES
#define CBC_LONG_ENC_WAVES 8           /* wavefronts per SIMD the register budget is cut for (A/B: profiles/r03_ab_kernels.log) */
#endif
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(CBC_LONG_ENC_WAVES)))
cbc_long_encode_kernel(cbc_long_args A)
{
    if (blockIdx.x >= A.n_blocks) return;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wid == 0u) cbc_long_encode<WaveGPU, CBC_ROLE_MODEL>(A, blockIdx.x, cbc_lds);
    else if (wid == 1u) cbc_long_encode<WaveGPU, CBC_ROLE_CODER>(A, blockIdx.x, cbc_lds);
    else cbc_long_encode<WaveGPU, CBC_ROLE_WALKER>(A, blockIdx.x, cbc_lds);
}
__global__ void __launch_bounds__(64)
cbc_long_decode_kernel(cbc_dec_args A) { if (blockIdx.x < A.n_blocks) cbc_long_decode<WaveGPU>(A, blockIdx.x, cbc_lds); }

/* ---- 2-bit transport (include/cbc_gpu.h): expand = one code word per lane -> 16 bases, one 16-byte store per lane ---- */
__global__ void __launch_bounds__(256)
cbc_expand_2bit_kernel(const uint32_t *__restrict__ codes, uint64_t n_words, uint8_t *__restrict__ out, uint64_t n_bases)
{
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t c = codes[w];
    uint32_t v[4];
    for (int q = 0; q < 4; q++) {
        uint32_t x = 0;
        for (int k = 0; k < 4; k++) {
            const uint32_t code = (c >> (2 * (4 * q + k))) & 3u;
            x |= (uint32_t)("ACGT"[code]) << (8 * k);
        }
        v[q] = x;
    }
    const uint64_t b0 = w * 16;
    if (b0 + 16 <= n_bases && ((uintptr_t)(out + b0) & 15) == 0) *(uint4 *)(out + b0) = make_uint4(v[0], v[1], v[2], v[3]);
    else for (int k = 0; k < 16 && b0 + k < n_bases; k++) out[b0 + k] = (uint8_t)(v[k >> 2] >> (8 * (k & 3)));
}
/* exception runs: workgroup = run, threads stride over its bytes */
__global__ void __launch_bounds__(256)
cbc_apply_runs_kernel(const cbc_2bit_run_dev *__restrict__ runs, uint64_t n_runs, uint8_t *__restrict__ out, uint64_t n_bases,
                      uint64_t lo, uint64_t hi /* only the bytes of [lo, hi): the chunk that has just been expanded */)
{
    const uint64_t r = blockIdx.x;
    if (r >= n_runs) return;
    const uint64_t s0 = runs[r].start; const uint32_t len = runs[r].length; const uint8_t b = (uint8_t)runs[r].byte;
    if (s0 > n_bases || len > n_bases - s0 || s0 >= hi || s0 + len <= lo) return;
    for (uint32_t i = threadIdx.x; i < len; i += 256) if (s0 + i >= lo && s0 + i < hi) out[s0 + i] = b;
}
/* pack decoded reads: thread = one 16-base word of one read row; bases past the read's length are not looked at */
__global__ void __launch_bounds__(256)
cbc_pack_2bit_kernel(const uint8_t *__restrict__ seq, const cbc_read_rec *__restrict__ recs, uint64_t rec0, uint64_t rec1, uint32_t stride,
                     uint32_t *__restrict__ codes, uint64_t *__restrict__ exc_idx, uint8_t *__restrict__ exc_val,
                     uint64_t exc_cap, unsigned long long *__restrict__ n_exc)
{
    const uint32_t row_words = stride >> 4;
    const uint64_t w = rec0 * row_words + (uint64_t)blockIdx.x * 256 + threadIdx.x;      /* records [rec0, rec1): one chunk of the decode */
    if (w >= rec1 * row_words) return;
    const uint64_t r = w / row_words; const uint32_t k0 = (uint32_t)(w % row_words) * 16u;
    const uint32_t rl = recs[r].rlen;
    const uint4 raw = *(const uint4 *)(seq + r * stride + k0);             /* rows are 16-byte aligned (stride % 16 == 0) */
    const uint32_t v[4] = { raw.x, raw.y, raw.z, raw.w };
    uint32_t c = 0;
    for (uint32_t k = 0; k < 16u; k++) {
        if (k0 + k >= rl) break;
        const uint8_t b = (uint8_t)(v[k >> 2] >> (8 * (k & 3)));
        const uint32_t code = b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u;
        if (code < 4u) c |= code << (2 * k);
        else {
            const unsigned long long at = atomicAdd(n_exc, 1ull);
            if (at < exc_cap) { exc_idx[at] = r * stride + k0 + k; exc_val[at] = b; }
        }
    }
    codes[w] = c;
}

/* exclusive scan of the per-block payload sizes -> offsets[n_blocks+1]; one workgroup */
__global__ void __launch_bounds__(1024)
cbc_scan_sizes_kernel(const cbc_block_result *__restrict__ results, uint64_t *__restrict__ offsets, uint32_t n_blocks)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n_blocks + 1023u) / 1024u;
    const uint32_t b0 = t * per, b1 = min(n_blocks, b0 + per);
    uint64_t s = 0;
    for (uint32_t b = b0; b < b1; b++) s += results[b].status == CBC_ST_OK ? results[b].nbytes : 0u;
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = part[t] - s;
    for (uint32_t b = b0; b < b1; b++) { offsets[b] = run; run += results[b].status == CBC_ST_OK ? results[b].nbytes : 0u; }
    if (t == 1023u) offsets[n_blocks] = part[1023];
}

/* gather the per-block payload areas into one compacted buffer, 16 bytes per lane where aligned */
__global__ void __launch_bounds__(256)
cbc_compact_kernel(const uint8_t *__restrict__ scratch, const cbc_block_desc *__restrict__ blocks,
                   const uint64_t *__restrict__ dst_off, uint8_t *__restrict__ dst, uint64_t dst_cap, uint32_t n_blocks)
{
    uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint64_t so = blocks[b].out_off, d0 = dst_off[b];
    uint64_t n = dst_off[b + 1] - d0;
    if (d0 + n > dst_cap) return;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) dst[d0 + i] = scratch[so + i];
}

/* 64-bit checksum of a byte range (cbc_gpu_checksum_device): sum of CBC_CHECKSUM_TERM(i, byte) mod 2^64 -- the sum is
 * order-free, so lanes, wavefronts and workgroups add their parts in any order and the value is exact */
__global__ void __launch_bounds__(256)
cbc_checksum_kernel(const uint8_t *__restrict__ p, uint64_t n, unsigned long long *__restrict__ sum)
{
    uint64_t acc = 0;
    const uint64_t n16 = n / 16;                              /* 16 bytes per lane where the base is aligned */
    if (((uintptr_t)p & 15) == 0) {
        for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < n16; w += (uint64_t)gridDim.x * 256) {
            const uint4 v = ((const uint4 *)p)[w];
            const uint32_t q[4] = { v.x, v.y, v.z, v.w };
            for (uint32_t k = 0; k < 16u; k++) acc += CBC_CHECKSUM_TERM(w * 16 + k, (uint8_t)(q[k >> 2] >> (8 * (k & 3))));
        }
        for (uint64_t i = n16 * 16 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc += CBC_CHECKSUM_TERM(i, p[i]);
    } else
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc += CBC_CHECKSUM_TERM(i, p[i]);
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down((unsigned long long)acc, d, 64);
    if ((threadIdx.x & 63u) == 0u && acc) atomicAdd(sum, (unsigned long long)acc);
}

/* ------------------------------------------------------------------------------------------------
 * context
 * ---------------------------------------------------------------------------------------------- */
/* grow-only device buffer owned by the context: the host-buffer entry points keep their device arrays between calls
 * (a hipMalloc / hipFree pair per array and call cost more than the copies they framed: profiles/r02_final_pcie.log) */
struct cbc_arena { void *p; uint64_t cap; };
enum { A_RECS, A_SEQ, A_TOK, A_NAMES, A_BLOCKS, A_OUT, A_RES, A_OFF, A_PACKED, A_CODES, A_RUNS, A_VS, A_IN, A_EXC_I, A_EXC_V, A_CNT, A_LSCR, A_STASH, A_GATHER, A_COUNT };
#define CBC_MAX_CHUNKS 8
#define CBC_N_KSTREAMS 8           /* every chunk's launch on a stream of its own: launches of different chunks share the chip */

struct cbc_gpu_ctx {
    int device;
    hipStream_t stream;
    hipStream_t s_copy;            /* H2D / D2H of the chunked host-buffer paths */
    hipStream_t s_k[CBC_N_KSTREAMS];   /* their kernel launches, chunk c on stream c % CBC_N_KSTREAMS */
    hipEvent_t ev0, ev1;
    hipEvent_t ev_chunk[CBC_MAX_CHUNKS], ev_done[CBC_N_KSTREAMS];
    int have_timing;
    int last_variant;              /* waves per SIMD of the encode build launched last */
    int n_cus;                     /* compute units of the device (block residency decides the kernel build) */
    uint8_t *d_ref; uint64_t ref_bytes;
    cbc_arena arena[A_COUNT];
    uint64_t stash_len;            /* bitstreams kept on the device by encode calls with out == NULL (cbc_gpu_group_gather moves them) */
    cbc_e2e_times last_e2e;
    char err[512];
};

static int set_err(cbc_gpu_ctx *c, int code, const char *what, hipError_t e)
{
    if (c) snprintf(c->err, sizeof c->err, "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
    return code;
}
#define HIPCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err(ctx, CBC_E_NODEV, what, e_); } while (0)

/* at least `bytes` in arena k; growing frees the old buffer (hipFree waits for the device) */
static int arena_need(cbc_gpu_ctx *ctx, int k, uint64_t bytes, const char *what)
{
    cbc_arena *a = &ctx->arena[k];
    if (a->p && a->cap >= bytes) return CBC_OK;
    if (a->p) { (void)hipFree(a->p); a->p = NULL; a->cap = 0; }
    const uint64_t want = (bytes + (bytes >> 3) + (2ull << 20)) & ~((2ull << 20) - 1);      /* 1/8 headroom, 2 MiB granules */
    HIPCHK(hipMalloc(&a->p, want), what);
    a->cap = want;
    return CBC_OK;
}
static double wall_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

API int cbc_gpu_abi_version(void) { return CBC_ABI_VERSION; }

API int cbc_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

API int cbc_gpu_init(int device_ordinal, cbc_gpu_ctx **out)
{
    if (!out) return CBC_E_ARG;
    *out = NULL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return CBC_E_NODEV;
    cbc_gpu_ctx *ctx = new (std::nothrow) cbc_gpu_ctx();
    if (!ctx) return CBC_E_NOMEM;
    memset(ctx, 0, sizeof *ctx);
    ctx->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->s_copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx; return CBC_E_NODEV;
    }
    for (int k = 0; k < CBC_N_KSTREAMS; k++)
        if (hipStreamCreateWithFlags(&ctx->s_k[k], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_done[k], hipEventDisableTiming) != hipSuccess) { delete ctx; return CBC_E_NODEV; }
    for (int k = 0; k < CBC_MAX_CHUNKS; k++)
        if (hipEventCreateWithFlags(&ctx->ev_chunk[k], hipEventDisableTiming) != hipSuccess) { delete ctx; return CBC_E_NODEV; }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) != hipSuccess || cus <= 0) cus = 256;
        ctx->n_cus = cus;
    }
    /* the kernel's dynamic LDS can exceed the 64 KiB default */
    (void)hipFuncSetAttribute((const void *)cbc_encode_blocks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)cbc_encode_blocks_kernel_w6, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)cbc_decode_blocks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)cbc_encode_whole_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)cbc_long_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)cbc_long_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)cbc_decode_whole_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *out = ctx;
    return CBC_OK;
}

API int cbc_gpu_shutdown(cbc_gpu_ctx *ctx)
{
    if (!ctx) return CBC_E_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->d_ref) (void)hipFree(ctx->d_ref);
    for (int k = 0; k < A_COUNT; k++) if (ctx->arena[k].p) (void)hipFree(ctx->arena[k].p);
    (void)hipEventDestroy(ctx->ev0); (void)hipEventDestroy(ctx->ev1);
    for (int k = 0; k < CBC_MAX_CHUNKS; k++) (void)hipEventDestroy(ctx->ev_chunk[k]);
    for (int k = 0; k < CBC_N_KSTREAMS; k++) { (void)hipEventDestroy(ctx->ev_done[k]); (void)hipStreamDestroy(ctx->s_k[k]); }
    (void)hipStreamDestroy(ctx->s_copy);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CBC_OK;
}

API const char *cbc_gpu_last_error(cbc_gpu_ctx *ctx) { return ctx ? ctx->err : "no context"; }

API int cbc_gpu_upload_reference(cbc_gpu_ctx *ctx, const uint8_t *bases, uint64_t nbytes)
{
    if (!ctx || !bases || nbytes == 0) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->d_ref) { (void)hipFree(ctx->d_ref); ctx->d_ref = NULL; ctx->ref_bytes = 0; }
    HIPCHK(hipMalloc((void **)&ctx->d_ref, nbytes), "hipMalloc(reference)");
    HIPCHK(hipMemcpy(ctx->d_ref, bases, nbytes, hipMemcpyHostToDevice), "hipMemcpy(reference)");
    ctx->ref_bytes = nbytes;
    return CBC_OK;
}

/* the reference as several host pieces laid end to end on the device (a device that owns some contigs uploads just those) */
API int cbc_gpu_upload_reference_parts(cbc_gpu_ctx *ctx, const uint8_t *const *parts, const uint64_t *bytes, uint32_t n_parts)
{
    if (!ctx || !parts || !bytes || n_parts == 0) return CBC_E_ARG;
    uint64_t total = 0;
    for (uint32_t k = 0; k < n_parts; k++) { if (!parts[k] && bytes[k]) return CBC_E_ARG; total += bytes[k]; }
    if (total == 0) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->d_ref) { (void)hipFree(ctx->d_ref); ctx->d_ref = NULL; ctx->ref_bytes = 0; }
    HIPCHK(hipMalloc((void **)&ctx->d_ref, total + 16), "hipMalloc(reference)");
    uint64_t at = 0;
    for (uint32_t k = 0; k < n_parts; k++) {
        if (bytes[k]) HIPCHK(hipMemcpyAsync(ctx->d_ref + at, parts[k], bytes[k], hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(reference part)");
        at += bytes[k];
    }
    HIPCHK(hipMemsetAsync(ctx->d_ref + total, 0, 16, ctx->stream), "memset reference pad");
    HIPCHK(hipStreamSynchronize(ctx->stream), "reference upload");
    ctx->ref_bytes = total;
    return CBC_OK;
}

API uint64_t cbc_gpu_plan_output(cbc_block_desc *blocks, uint32_t n_blocks, const cbc_read_rec *recs, const uint32_t *tok)
{
    return cbc_plan_output(blocks, n_blocks, recs, tok);
}
API uint64_t cbc_gpu_plan_output_caps(cbc_block_desc *blocks, uint32_t n_blocks, const cbc_lds_caps *caps)
{
    return (blocks && caps) ? cbc_plan_output_caps(blocks, n_blocks, caps) : 0;
}
API uint32_t cbc_gpu_lds_bytes(const cbc_lds_caps *caps) { return caps ? cbc_plan_lds_bytes(caps) : 0; }

/* grow the context's arenas for a batch of this shape before the batch exists (a CLI does it while the host still parses) */
API int cbc_gpu_reserve_encode(cbc_gpu_ctx *ctx, uint64_t n_recs, uint64_t seq_bytes, uint64_t n_tok, uint32_t n_blocks, uint64_t scratch_bytes)
{
    if (!ctx) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    int rc;
    if ((rc = arena_need(ctx, A_RECS, n_recs * sizeof(cbc_read_rec) + 16, "hipMalloc recs"))) return rc;
    if ((rc = arena_need(ctx, A_SEQ, seq_bytes + 32, "hipMalloc seq"))) return rc;
    if ((rc = arena_need(ctx, A_TOK, (n_tok ? n_tok : 1) * 4 + 16, "hipMalloc tok"))) return rc;
    if ((rc = arena_need(ctx, A_BLOCKS, (uint64_t)(n_blocks ? n_blocks : 1) * sizeof(cbc_block_desc), "hipMalloc blocks"))) return rc;
    if ((rc = arena_need(ctx, A_RES, (uint64_t)(n_blocks ? n_blocks : 1) * sizeof(cbc_block_result), "hipMalloc results"))) return rc;
    if ((rc = arena_need(ctx, A_OFF, ((uint64_t)n_blocks + 1) * 8, "hipMalloc offsets"))) return rc;
    if ((rc = arena_need(ctx, A_OUT, scratch_bytes ? scratch_bytes : 1, "hipMalloc out scratch"))) return rc;
    return CBC_OK;
}

static int encode_blocks_launch(cbc_gpu_ctx *ctx, const cbc_device_batch *b, void *hip_stream, uint64_t resident_blocks);
API int cbc_gpu_encode_blocks_device(cbc_gpu_ctx *ctx, const cbc_device_batch *b, void *hip_stream)
{
    return encode_blocks_launch(ctx, b, hip_stream, b ? b->n_blocks : 0);
}
/* resident_blocks: how many blocks compete for the chip while this launch runs (the chunked host-buffer path makes several
 * launches that run side by side): it, not the launch's own grid, decides the register budget of the build */
static int encode_blocks_launch(cbc_gpu_ctx *ctx, const cbc_device_batch *b, void *hip_stream, uint64_t resident_blocks)
{
    if (!ctx || !b) return CBC_E_ARG;
    if (b->n_blocks == 0) return CBC_OK;
    if (!b->d_recs || !b->d_seq || !b->d_tok || !b->d_names || !b->d_blocks || !b->d_ref || !b->d_out || !b->d_results)
        return set_err(ctx, CBC_E_ARG, "null device pointer in cbc_device_batch", hipSuccess);
    if (b->caps.cap_pos < 2 || b->caps.cap_pos > 8192 || b->caps.cap_var < 1 || b->caps.cap_var > 32768)
        return set_err(ctx, CBC_E_ARG, "lds caps out of range", hipSuccess);
    const uint32_t lds = cbc_plan_lds_bytes(&b->caps);
    if (lds > 160u * 1024u) return set_err(ctx, CBC_E_ARG, "lds caps need more than 160 KiB", hipSuccess);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t s = hip_stream == CBC_CTX_STREAM ? ctx->stream : (hipStream_t)hip_stream;
    cbc_enc_args A;
    A.recs = b->d_recs; A.seq = b->d_seq; A.tok = b->d_tok; A.names = b->d_names; A.blocks = b->d_blocks;
    A.ref = b->d_ref; A.out = b->d_out; A.results = b->d_results;
    A.ref_bytes = b->ref_bytes; A.out_bytes = b->out_bytes; A.seq_bytes = b->seq_bytes; A.n_tok = b->n_tok;
    A.n_recs = b->n_recs; A.n_blocks = b->n_blocks; A.cap_pos = b->caps.cap_pos; A.cap_var = b->caps.cap_var;
    A.names_bytes = 0x7fffffffu;   /* names are NUL-terminated; bounded by CBC_CAP_NAME in the kernel */
    HIPCHK(hipEventRecord(ctx->ev0, s), "hipEventRecord");
    if (resident_blocks > 10ull * (uint64_t)ctx->n_cus)           /* more blocks than are resident at 5 waves per SIMD */
        { hipLaunchKernelGGL(cbc_encode_blocks_kernel_w6, dim3(b->n_blocks), dim3(128), lds, s, A); ctx->last_variant = 6; }
    else
        { hipLaunchKernelGGL(cbc_encode_blocks_kernel, dim3(b->n_blocks), dim3(128), lds, s, A); ctx->last_variant = 5; }
    HIPCHK(hipGetLastError(), "launch cbc_encode_blocks_kernel");
    HIPCHK(hipEventRecord(ctx->ev1, s), "hipEventRecord");
    ctx->have_timing = 1;
    return CBC_OK;
}

API int cbc_gpu_last_kernel_ms(cbc_gpu_ctx *ctx, float *ms)
{
    if (!ctx || !ms || !ctx->have_timing) return CBC_E_ARG;
    HIPCHK(hipEventSynchronize(ctx->ev1), "hipEventSynchronize");
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1), "hipEventElapsedTime");
    return CBC_OK;
}

API int cbc_gpu_last_kernel_variant(cbc_gpu_ctx *ctx) { return ctx ? ctx->last_variant : 0; }

API int cbc_gpu_synchronize(cbc_gpu_ctx *ctx)
{
    if (!ctx) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return CBC_OK;
}

API int cbc_gpu_compact_device(cbc_gpu_ctx *ctx, const uint8_t *d_scratch, const cbc_block_desc *d_blocks,
                               const cbc_block_result *d_results, uint32_t n_blocks, uint64_t *d_offsets,
                               uint8_t *d_packed, uint64_t packed_cap, void *hip_stream)
{
    if (!ctx || !d_scratch || !d_blocks || !d_results || !d_offsets || !d_packed) return CBC_E_ARG;
    if (n_blocks == 0) return CBC_OK;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t s = hip_stream == CBC_CTX_STREAM ? ctx->stream : (hipStream_t)hip_stream;
    hipLaunchKernelGGL(cbc_scan_sizes_kernel, dim3(1), dim3(1024), 0, s, d_results, d_offsets, n_blocks);
    HIPCHK(hipGetLastError(), "launch cbc_scan_sizes_kernel");
    hipLaunchKernelGGL(cbc_compact_kernel, dim3(n_blocks), dim3(256), 0, s, d_scratch, d_blocks,
                       (const uint64_t *)d_offsets, d_packed, packed_cap, n_blocks);
    HIPCHK(hipGetLastError(), "launch cbc_compact_kernel");
    return CBC_OK;
}

API int cbc_gpu_checksum_device(cbc_gpu_ctx *ctx, const uint8_t *d_bytes, uint64_t n, uint64_t *d_sum, void *hip_stream)
{
    if (!ctx || !d_sum || (n && !d_bytes)) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t s = hip_stream == CBC_CTX_STREAM ? ctx->stream : (hipStream_t)hip_stream;
    HIPCHK(hipMemsetAsync(d_sum, 0, 8, s), "memset checksum");
    if (n == 0) return CBC_OK;
    const uint64_t want = (n / 16 + 255) / 256 + 1;
    const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(cbc_checksum_kernel, dim3(grid), dim3(256), 0, s, d_bytes, n, (unsigned long long *)d_sum);
    HIPCHK(hipGetLastError(), "launch cbc_checksum_kernel");
    return CBC_OK;
}

static int expand_2bit(cbc_gpu_ctx *ctx, const uint32_t *codes, uint64_t n_bases, const cbc_2bit_run_dev *runs, uint64_t n_runs,
                       uint8_t *d_out, void **d_tmp_codes, void **d_tmp_runs)
{
    const uint64_t n_words = (n_bases + 15) / 16;
    if (n_words == 0) return CBC_OK;
    if (n_runs > 0x7fffffffull || n_words > 0x7fffffffull * 256ull) return set_err(ctx, CBC_E_ARG, "2-bit transport: too many words or runs", hipSuccess);
    HIPCHK(hipMalloc(d_tmp_codes, n_words * 4), "hipMalloc 2-bit codes");
    HIPCHK(hipMemcpyAsync(*d_tmp_codes, codes, n_words * 4, hipMemcpyHostToDevice, ctx->stream), "H2D 2-bit codes");
    hipLaunchKernelGGL(cbc_expand_2bit_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t *)*d_tmp_codes, n_words, d_out, n_bases);
    HIPCHK(hipGetLastError(), "launch cbc_expand_2bit_kernel");
    if (n_runs) {
        HIPCHK(hipMalloc(d_tmp_runs, n_runs * sizeof(cbc_2bit_run_dev)), "hipMalloc 2-bit runs");
        HIPCHK(hipMemcpyAsync(*d_tmp_runs, runs, n_runs * sizeof(cbc_2bit_run_dev), hipMemcpyHostToDevice, ctx->stream), "H2D 2-bit runs");
        hipLaunchKernelGGL(cbc_apply_runs_kernel, dim3((unsigned)n_runs), dim3(256), 0, ctx->stream,
                           (const cbc_2bit_run_dev *)*d_tmp_runs, n_runs, d_out, n_bases, (uint64_t)0, n_bases);
        HIPCHK(hipGetLastError(), "launch cbc_apply_runs_kernel");
    }
    return CBC_OK;
}

API int cbc_gpu_upload_reference_2bit(cbc_gpu_ctx *ctx, const uint32_t *codes, uint64_t n_bases, const cbc_2bit_run_dev *runs, uint64_t n_runs)
{
    if (!ctx || !codes || n_bases == 0 || (n_runs && !runs)) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->d_ref) { (void)hipFree(ctx->d_ref); ctx->d_ref = NULL; ctx->ref_bytes = 0; }
    HIPCHK(hipMalloc((void **)&ctx->d_ref, n_bases + 16), "hipMalloc(reference)");
    void *d_codes = NULL, *d_runs = NULL;
    int rc = expand_2bit(ctx, codes, n_bases, runs, n_runs, ctx->d_ref, &d_codes, &d_runs);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_err(ctx, CBC_E_NODEV, "2-bit reference expansion", hipGetLastError());
    if (d_codes) (void)hipFree(d_codes); if (d_runs) (void)hipFree(d_runs);
    if (rc) { (void)hipFree(ctx->d_ref); ctx->d_ref = NULL; return rc; }
    ctx->ref_bytes = n_bases;
    return CBC_OK;
}

static int encode_blocks_impl(cbc_gpu_ctx *ctx, const cbc_host_batch *hb, const uint32_t *seq_codes, const cbc_2bit_run_dev *seq_runs,
                              uint64_t n_seq_runs, uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, cbc_block_result *results,
                              const uint8_t *d_seq_ext = NULL, const uint32_t *d_tok_ext = NULL, const cbc_tok_record_summary *sums = NULL);

/* cbc_plan_output() when the tokens are on the device: the per-record var-symbol bound travels in the summaries */
static uint64_t plan_output_from_summaries(cbc_block_desc *blocks, uint32_t n_blocks, const cbc_tok_record_summary *sums)
{
    uint64_t off = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        cbc_block_desc *bd = &blocks[b];
        uint64_t nev = 0;
        for (uint32_t r = 0; r < bd->n_reads; r++) nev += sums[bd->rec_base + r].nt_ev >> 16;
        const uint64_t nsym = 136u + 2u * CBC_CAP_NAME + 16ull * bd->n_reads + 2 * nev;
        uint64_t payload_cap = (3 * nsym + 256 + 255) & ~255ull;
        uint64_t cap = payload_cap + ((4 * (nev + 64) + 255) & ~255ull);
        if (cap > 0xffffff00ull) { cap = 0xffffff00ull; payload_cap = cap / 2; payload_cap &= ~255ull; }
        bd->out_off = off; bd->out_cap = (uint32_t)cap; bd->reserved = (uint32_t)payload_cap;
        off += cap;
    }
    return off;
}

/* host-buffer entry point: H2D, encode, size scan, device-side compaction, D2H */
API int cbc_gpu_encode_blocks(cbc_gpu_ctx *ctx, const cbc_host_batch *hb, uint8_t *out, uint64_t out_cap,
                              uint64_t *out_offsets, cbc_block_result *results)
{
    if (!hb || !hb->seq) return CBC_E_ARG;
    return encode_blocks_impl(ctx, hb, NULL, NULL, 0, out, out_cap, out_offsets, results);
}
API int cbc_gpu_encode_blocks_2bit(cbc_gpu_ctx *ctx, const cbc_host_batch *hb, const uint32_t *seq_codes, const cbc_2bit_run_dev *seq_runs,
                                   uint64_t n_seq_runs, uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, cbc_block_result *results)
{
    if (!seq_codes || (n_seq_runs && !seq_runs)) return CBC_E_ARG;
    return encode_blocks_impl(ctx, hb, seq_codes, seq_runs, n_seq_runs, out, out_cap, out_offsets, results);
}

/* The host-buffer encode path (timed region ii), as a pipeline:
 *   - device arrays are the context's grow-only arenas (no hipMalloc / hipFree per call once they have their size);
 *   - the batch is cut into up to CBC_MAX_CHUNKS runs of consecutive blocks; chunk c's records, bases (bytes or 2-bit
 *     codes) and tokens go H2D on the copy stream, an event later its expansion (2-bit) and its encode launch run on
 *     kernel stream c % 2 -- so chunk c + 1 crosses PCIe while chunk c is being coded, and launches of neighbouring
 *     chunks share the chip (a block is one serial chain: a launch of few blocks cannot fill it alone);
 *   - one size scan + compaction over all blocks and one D2H of the compacted bitstreams (2 bytes per read) end it.
 * Source buffers that are page-locked (cbc_gpu_host_register, or the caller's own hipHostMalloc) are read by DMA at the
 * link rate; pageable ones go through the runtime's staging, which blocks this thread but not the launches already made.
 * Needs the blocks' bases in ascending order with rec / seq / tok ranges contiguous from block to block (what the packers
 * produce); any other descriptor list is sent as one chunk covering the arrays whole. */
static int encode_blocks_impl(cbc_gpu_ctx *ctx, const cbc_host_batch *hb, const uint32_t *seq_codes, const cbc_2bit_run_dev *seq_runs,
                              uint64_t n_seq_runs, uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, cbc_block_result *results,
                              const uint8_t *d_seq_ext, const uint32_t *d_tok_ext, const cbc_tok_record_summary *sums)
{
    if (!ctx || !hb || !out_offsets) return CBC_E_ARG;     /* out == NULL: the bitstreams stay on the device (the context's stash) */
    if (!ctx->d_ref) return set_err(ctx, CBC_E_ARG, "cbc_gpu_upload_reference has not been called", hipSuccess);
    const uint32_t nb = hb->n_blocks;
    out_offsets[0] = 0;
    if (nb == 0) return CBC_OK;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const double T0 = wall_now();
    cbc_e2e_times tm; memset(&tm, 0, sizeof tm);
    /* payload areas from the caps alone (O(blocks)); a batch whose cap_var is not a bound -- not from the packers -- gets
     * OUT_FULL / CAP_VAR statuses from the kernel, never a wrong byte */
    const uint64_t scratch = sums ? plan_output_from_summaries(hb->blocks, nb, sums) : cbc_plan_output_caps(hb->blocks, nb, &hb->caps);
    cbc_block_result *res = NULL;
    uint8_t *d_compact = NULL;
    int rc = CBC_OK;
    uint64_t total = 0;
    const uint64_t ntok = hb->n_tok ? hb->n_tok : 1;
    const cbc_block_desc *B = hb->blocks;
#define GO(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = set_err(ctx, CBC_E_NODEV, what, e_); goto done; } } while (0)
#define NEED(k, bytes, what) do { rc = arena_need(ctx, k, bytes, what); if (rc) goto done; } while (0)
    NEED(A_RECS, hb->n_recs * sizeof(cbc_read_rec) + 16, "hipMalloc recs");
    if (!d_seq_ext) NEED(A_SEQ, hb->seq_bytes + 32, "hipMalloc seq");
    if (!d_tok_ext) NEED(A_TOK, ntok * 4 + 16, "hipMalloc tok");
    NEED(A_NAMES, hb->names_bytes + 16, "hipMalloc names");
    NEED(A_BLOCKS, (uint64_t)nb * sizeof(cbc_block_desc), "hipMalloc blocks");
    NEED(A_OUT, scratch, "hipMalloc out scratch");
    NEED(A_RES, (uint64_t)nb * sizeof(cbc_block_result), "hipMalloc results");
    NEED(A_OFF, ((uint64_t)nb + 1) * 8, "hipMalloc offsets");
    if (seq_codes && !d_seq_ext) {
        NEED(A_CODES, ((hb->seq_bytes + 15) / 16) * 4 + 16, "hipMalloc 2-bit codes");
        if (n_seq_runs) NEED(A_RUNS, n_seq_runs * sizeof(cbc_2bit_run_dev), "hipMalloc 2-bit runs");
        if (n_seq_runs > 0x7fffffffull) { rc = set_err(ctx, CBC_E_ARG, "2-bit transport: too many runs", hipSuccess); goto done; }
    }
    tm.alloc_s = wall_now() - T0;
    {
        uint8_t *d_recs = (uint8_t *)ctx->arena[A_RECS].p, *d_seq = d_seq_ext ? NULL : (uint8_t *)ctx->arena[A_SEQ].p;
        uint32_t *d_tok = d_tok_ext ? NULL : (uint32_t *)ctx->arena[A_TOK].p, *d_codes = (uint32_t *)ctx->arena[A_CODES].p;
        cbc_block_desc *d_blocks = (cbc_block_desc *)ctx->arena[A_BLOCKS].p;
        cbc_block_result *d_res = (cbc_block_result *)ctx->arena[A_RES].p;
        hipStream_t sc = ctx->s_copy;
        /* chunks: runs of consecutive blocks of about equal H2D volume, when the descriptor list is contiguous */
        uint32_t n_chunks = 1, cut[CBC_MAX_CHUNKS + 1];
        bool contiguous = true;
        for (uint32_t b = 0; b + 1 < nb && contiguous; b++)
            contiguous = B[b + 1].rec_base == B[b].rec_base + B[b].n_reads && B[b + 1].seq_base >= B[b].seq_base && B[b + 1].tok_base >= B[b].tok_base;
        contiguous = contiguous && B[0].rec_base + 0 <= hb->n_recs && B[nb - 1].rec_base + B[nb - 1].n_reads <= hb->n_recs
                     && B[nb - 1].seq_base <= hb->seq_bytes && B[nb - 1].tok_base <= ntok;
        const uint64_t vol = hb->n_recs * 16 + (d_seq_ext ? 0 : seq_codes ? hb->seq_bytes / 4 : hb->seq_bytes) + (d_tok_ext ? 0 : ntok * 4);
        if (contiguous && nb >= 512) {
            /* A block is one serial chain (~5.6 ms for 4096 reads) however few blocks a launch holds, and the link moves
             * ~57 GB/s: chunks of >= 64 MB and >= 256 blocks, each launched on its own stream the moment it has arrived */
            uint64_t want = vol / (64ull << 20);
            if (want > CBC_MAX_CHUNKS) want = CBC_MAX_CHUNKS;
            if (want > nb / 256) want = nb / 256;
            if (want >= 2) n_chunks = (uint32_t)want;
        }
        cut[0] = 0; cut[n_chunks] = nb;
        for (uint32_t c = 1; c < n_chunks; c++) {                 /* equal record counts ~ equal bytes */
            const uint64_t target = B[0].rec_base + (B[nb - 1].rec_base + B[nb - 1].n_reads - B[0].rec_base) * c / n_chunks;
            uint32_t lo = cut[c - 1] + 1, hi = nb - (n_chunks - c);
            while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (B[mid].rec_base < target) lo = mid + 1; else hi = mid; }
            cut[c] = lo;
        }
        tm.n_chunks = n_chunks;
        /* small things first: descriptors, names, result slots, exception runs */
        GO(hipMemcpyAsync(d_blocks, B, (uint64_t)nb * sizeof(cbc_block_desc), hipMemcpyHostToDevice, sc), "H2D blocks");
        GO(hipMemcpyAsync(ctx->arena[A_NAMES].p, hb->names, hb->names_bytes, hipMemcpyHostToDevice, sc), "H2D names");
        GO(hipMemsetAsync(d_res, 0xff, (uint64_t)nb * sizeof(cbc_block_result), sc), "memset results");
        if (seq_codes && !d_seq_ext && n_seq_runs)
            GO(hipMemcpyAsync(ctx->arena[A_RUNS].p, seq_runs, n_seq_runs * sizeof(cbc_2bit_run_dev), hipMemcpyHostToDevice, sc), "H2D 2-bit runs");
        tm.h2d_bytes = (uint64_t)nb * sizeof(cbc_block_desc) + hb->names_bytes + (seq_codes && !d_seq_ext ? n_seq_runs * sizeof(cbc_2bit_run_dev) : 0);
        uint64_t words_done = 0;                                  /* 2-bit words expanded so far (chunks meet inside a word) */
        for (uint32_t c = 0; c < n_chunks; c++) {
            const uint32_t c0 = cut[c], c1 = cut[c + 1];
            const bool whole = !contiguous || n_chunks == 1;
            const uint64_t r0 = whole ? 0 : B[c0].rec_base, r1 = whole ? hb->n_recs : B[c1 - 1].rec_base + B[c1 - 1].n_reads;
            const uint64_t s0 = whole ? 0 : B[c0].seq_base, s1 = whole || c1 == nb ? hb->seq_bytes : B[c1].seq_base;
            const uint64_t t0 = whole ? 0 : B[c0].tok_base, t1 = whole || c1 == nb ? hb->n_tok : B[c1].tok_base;
            GO(hipMemcpyAsync(d_recs + r0 * 16, (const uint8_t *)hb->recs + r0 * 16, (r1 - r0) * 16, hipMemcpyHostToDevice, sc), "H2D recs");
            tm.h2d_bytes += (r1 - r0) * 16;
            uint64_t w0 = 0, w1 = 0;
            if (d_seq_ext) { /* already resident */ }
            else if (seq_codes) {                                  /* 2-bit transport: a quarter of the bytes cross PCIe, expanded on the device */
                w0 = words_done; w1 = (s1 + 15) / 16; if (w1 < w0) w1 = w0;
                if (w1 > w0) GO(hipMemcpyAsync(d_codes + w0, seq_codes + w0, (w1 - w0) * 4, hipMemcpyHostToDevice, sc), "H2D 2-bit codes");
                tm.h2d_bytes += (w1 - w0) * 4; words_done = w1;
            } else if (s1 > s0) {
                GO(hipMemcpyAsync(d_seq + s0, hb->seq + s0, s1 - s0, hipMemcpyHostToDevice, sc), "H2D seq");
                tm.h2d_bytes += s1 - s0;
            }
            if (!d_tok_ext && t1 > t0) {
                GO(hipMemcpyAsync(d_tok + t0, hb->tok + t0, (t1 - t0) * 4, hipMemcpyHostToDevice, sc), "H2D tok");
                tm.h2d_bytes += (t1 - t0) * 4;
            }
            GO(hipEventRecord(ctx->ev_chunk[c], sc), "hipEventRecord");
            hipStream_t ks = ctx->s_k[c % CBC_N_KSTREAMS];
            GO(hipStreamWaitEvent(ks, ctx->ev_chunk[c], 0), "hipStreamWaitEvent");
            if (seq_codes && !d_seq_ext && w1 > w0) {
                if (w1 - w0 > 0x7fffffffull * 256ull) { rc = set_err(ctx, CBC_E_ARG, "2-bit transport: too many words", hipSuccess); goto done; }
                hipLaunchKernelGGL(cbc_expand_2bit_kernel, dim3((unsigned)((w1 - w0 + 255) / 256)), dim3(256), 0, ks,
                                   (const uint32_t *)(d_codes + w0), w1 - w0, d_seq + w0 * 16, hb->seq_bytes - w0 * 16);
                GO(hipGetLastError(), "launch cbc_expand_2bit_kernel");
                if (n_seq_runs) {
                    hipLaunchKernelGGL(cbc_apply_runs_kernel, dim3((unsigned)n_seq_runs), dim3(256), 0, ks,
                                       (const cbc_2bit_run_dev *)ctx->arena[A_RUNS].p, n_seq_runs, d_seq, hb->seq_bytes, w0 * 16, w1 * 16);
                    GO(hipGetLastError(), "launch cbc_apply_runs_kernel");
                }
            }
            cbc_device_batch db;
            memset(&db, 0, sizeof db);
            db.d_recs = (const cbc_read_rec *)d_recs; db.d_seq = d_seq_ext ? d_seq_ext : d_seq; db.d_tok = d_tok_ext ? d_tok_ext : d_tok;
            db.d_names = (const uint8_t *)ctx->arena[A_NAMES].p; db.d_blocks = d_blocks + c0; db.n_blocks = c1 - c0;
            db.d_ref = ctx->d_ref; db.ref_bytes = ctx->ref_bytes; db.d_out = (uint8_t *)ctx->arena[A_OUT].p; db.out_bytes = scratch;
            db.d_results = d_res + c0; db.seq_bytes = hb->seq_bytes; db.n_tok = ntok; db.n_recs = hb->n_recs;
            db.caps = hb->caps;
            rc = encode_blocks_launch(ctx, &db, ks, nb);
            if (rc) goto done;
        }
        /* the context's own stream joins the kernel streams, then: sizes -> offsets -> compaction -> D2H */
        for (int k = 0; k < CBC_N_KSTREAMS; k++) {
            GO(hipEventRecord(ctx->ev_done[k], ctx->s_k[k]), "hipEventRecord");
            GO(hipStreamWaitEvent(ctx->stream, ctx->ev_done[k], 0), "hipStreamWaitEvent");
        }
        tm.issue_s = wall_now() - T0;
        hipLaunchKernelGGL(cbc_scan_sizes_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const cbc_block_result *)d_res,
                           (uint64_t *)ctx->arena[A_OFF].p, nb);
        GO(hipGetLastError(), "launch cbc_scan_sizes_kernel");
        res = results ? results : (cbc_block_result *)malloc((size_t)nb * sizeof(cbc_block_result));
        if (!res) { rc = CBC_E_NOMEM; goto done; }
        GO(hipMemcpyAsync(res, d_res, (uint64_t)nb * sizeof(cbc_block_result), hipMemcpyDeviceToHost, ctx->stream), "D2H results");
        GO(hipMemcpyAsync(out_offsets, ctx->arena[A_OFF].p, ((uint64_t)nb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream), "D2H offsets");
        GO(hipStreamSynchronize(ctx->stream), "encode kernel");
        tm.kernels_done_s = wall_now() - T0;
        /* the compacted bitstreams (2 bytes per read) go where the batch's bases were: they are dead now, and a second
         * worst-case-sized buffer would cost more to allocate than the whole call takes */
        total = out_offsets[nb];
        if (!d_seq_ext && total <= ctx->arena[A_SEQ].cap) d_compact = (uint8_t *)ctx->arena[A_SEQ].p;
        else { NEED(A_PACKED, total + 16, "hipMalloc packed"); d_compact = (uint8_t *)ctx->arena[A_PACKED].p; }
        if (total) {
            hipLaunchKernelGGL(cbc_compact_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const uint8_t *)ctx->arena[A_OUT].p,
                               (const cbc_block_desc *)d_blocks, (const uint64_t *)ctx->arena[A_OFF].p, d_compact, total, nb);
            GO(hipGetLastError(), "launch cbc_compact_kernel");
        }
    }
    for (uint32_t b = 0; b < nb; b++) {
        if (res[b].status != CBC_ST_OK && rc == CBC_OK) {
            snprintf(ctx->err, sizeof ctx->err, "block %u failed with status %u at record %u", b, res[b].status, res[b].fail_read);
            rc = CBC_E_BLOCK;
        }
    }
    if (!out) {
        /* keep them: appended to the stash (a grow-with-copy buffer), for cbc_gpu_group_gather / cbc_gpu_stash_fetch */
        if (total) {
            const uint64_t need = ctx->stash_len + total + 16;
            if (ctx->arena[A_STASH].cap < need) {
                void *np = NULL; const uint64_t want = (need + (need >> 1) + (4ull << 20)) & ~((2ull << 20) - 1);
                GO(hipMalloc(&np, want), "hipMalloc stash");
                if (ctx->stash_len) GO(hipMemcpyAsync(np, ctx->arena[A_STASH].p, ctx->stash_len, hipMemcpyDeviceToDevice, ctx->stream), "stash copy");
                GO(hipStreamSynchronize(ctx->stream), "stash copy");
                if (ctx->arena[A_STASH].p) (void)hipFree(ctx->arena[A_STASH].p);
                ctx->arena[A_STASH].p = np; ctx->arena[A_STASH].cap = want;
            }
            GO(hipMemcpyAsync((uint8_t *)ctx->arena[A_STASH].p + ctx->stash_len, d_compact, total, hipMemcpyDeviceToDevice, ctx->stream), "D2D stash");
            GO(hipStreamSynchronize(ctx->stream), "D2D stash");
            ctx->stash_len += total;
        }
    } else {
    if (total > out_cap) { rc = set_err(ctx, CBC_E_ARG, "out_cap too small for the compacted payloads", hipSuccess); goto done; }
    if (total) {
        GO(hipMemcpyAsync(out, d_compact, total, hipMemcpyDeviceToHost, ctx->stream), "D2H payloads");
        GO(hipStreamSynchronize(ctx->stream), "D2H payloads");
    }
    }
    tm.d2h_bytes = total + (uint64_t)nb * (sizeof(cbc_block_result) + 8) + 8;
done:
#undef GO
#undef NEED
    if (rc && rc != CBC_E_BLOCK) (void)hipDeviceSynchronize();   /* nothing of a failed call may still be running over the arenas */
    if (res && res != results) free(res);
    tm.total_s = wall_now() - T0;
    ctx->last_e2e = tm;
    return rc;
}

API int cbc_gpu_last_e2e(cbc_gpu_ctx *ctx, cbc_e2e_times *out)
{
    if (!ctx || !out) return CBC_E_ARG;
    *out = ctx->last_e2e;
    return CBC_OK;
}

API int cbc_gpu_host_register(cbc_gpu_ctx *ctx, const void *p, uint64_t bytes)
{
    if (!ctx || !p || !bytes) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipHostRegister((void *)p, bytes, hipHostRegisterDefault), "hipHostRegister");
    return CBC_OK;
}
API int cbc_gpu_host_unregister(cbc_gpu_ctx *ctx, const void *p)
{
    if (!ctx || !p) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipHostUnregister((void *)p), "hipHostUnregister");
    return CBC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * the exchange step of the multi-device path: every device's bitstreams to device 0 over RCCL
 * (grouped ncclSend / ncclRecv over xGMI), checksummed on both sides, then one D2H.  One process, one context per device --
 * the shape of `cbc --devices a,b,...`; bench.py's one-process-per-GPU form makes the same exchange through
 * torch.distributed's "nccl" backend, which is this library too.  librccl.so is loaded here, not at program start.
 * ---------------------------------------------------------------------------------------------- */
struct cbc_rccl_api {
    void *so;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
};
struct cbc_gpu_group { int n; cbc_gpu_ctx **ctx; ncclComm_t *comm; cbc_rccl_api api; char err[256]; };

static int rccl_load(cbc_rccl_api *a, char *err, size_t errlen)
{
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    a->so = NULL;
    for (unsigned k = 0; k < 3 && !a->so; k++) a->so = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!a->so) { snprintf(err, errlen, "cannot load librccl: %s", dlerror()); return CBC_E_NODEV; }
#define SYM(field, name) do { *(void **)&a->field = dlsym(a->so, name); if (!a->field) { snprintf(err, errlen, "librccl has no %s", name); return CBC_E_NODEV; } } while (0)
    SYM(CommInitAll, "ncclCommInitAll"); SYM(CommDestroy, "ncclCommDestroy"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return CBC_OK;
}

API const char *cbc_gpu_group_last_error(cbc_gpu_group *g) { return g ? g->err : "no group"; }

API void cbc_gpu_group_destroy(cbc_gpu_group *g)
{
    if (!g) return;
    if (g->comm) { for (int k = 0; k < g->n; k++) if (g->comm[k]) { (void)hipSetDevice(g->ctx[k]->device); (void)g->api.CommDestroy(g->comm[k]); } }
    free(g->comm); free(g->ctx);
    /* the library stays loaded: unloading RCCL while HIP is alive is not worth the risk */
    delete g;
}

API int cbc_gpu_group_create(cbc_gpu_ctx *const *ctxs, int n, cbc_gpu_group **out)
{
    if (!ctxs || !out || n < 1 || n > 64) return CBC_E_ARG;
    *out = NULL;
    for (int k = 0; k < n; k++) { if (!ctxs[k]) return CBC_E_ARG; for (int j = 0; j < k; j++) if (ctxs[j]->device == ctxs[k]->device) return CBC_E_ARG; }   /* one rank per device */
    cbc_gpu_group *g = new (std::nothrow) cbc_gpu_group();
    if (!g) return CBC_E_NOMEM;
    memset(g, 0, sizeof *g);
    g->n = n;
    g->ctx = (cbc_gpu_ctx **)calloc((size_t)n, sizeof(cbc_gpu_ctx *)); g->comm = (ncclComm_t *)calloc((size_t)n, sizeof(ncclComm_t));
    int *devs = (int *)calloc((size_t)n, sizeof(int));
    int rc = (g->ctx && g->comm && devs) ? CBC_OK : CBC_E_NOMEM;
    if (!rc) rc = rccl_load(&g->api, g->err, sizeof g->err);
    if (!rc) {
        for (int k = 0; k < n; k++) { g->ctx[k] = ctxs[k]; devs[k] = ctxs[k]->device; }
        const ncclResult_t r = g->api.CommInitAll(g->comm, n, devs);
        if (r != ncclSuccess) { snprintf(g->err, sizeof g->err, "ncclCommInitAll: %s", g->api.GetErrorString(r)); rc = CBC_E_NODEV; memset(g->comm, 0, (size_t)n * sizeof(ncclComm_t)); }
    }
    free(devs);
    if (rc) { if (ctxs[0]) snprintf(ctxs[0]->err, sizeof ctxs[0]->err, "%s", g->err); cbc_gpu_group_destroy(g); return rc; }
    *out = g;
    return CBC_OK;
}

API int cbc_gpu_stash_reset(cbc_gpu_ctx *ctx) { if (!ctx) return CBC_E_ARG; ctx->stash_len = 0; return CBC_OK; }
API uint64_t cbc_gpu_stash_bytes(cbc_gpu_ctx *ctx) { return ctx ? ctx->stash_len : 0; }
API int cbc_gpu_stash_fetch(cbc_gpu_ctx *ctx, uint8_t *out, uint64_t out_cap)
{
    if (!ctx || (ctx->stash_len && !out) || out_cap < ctx->stash_len) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->stash_len) HIPCHK(hipMemcpy(out, ctx->arena[A_STASH].p, ctx->stash_len, hipMemcpyDeviceToHost), "D2H stash");
    return CBC_OK;
}

static int device_checksum_now(cbc_gpu_ctx *ctx, const uint8_t *d, uint64_t n, uint64_t *sum)
{
    int rc = arena_need(ctx, A_CNT, 8, "hipMalloc checksum"); if (rc) return rc;
    rc = cbc_gpu_checksum_device(ctx, d, n, (uint64_t *)ctx->arena[A_CNT].p, CBC_CTX_STREAM); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(sum, ctx->arena[A_CNT].p, 8, hipMemcpyDeviceToHost, ctx->stream), "D2H checksum");
    HIPCHK(hipStreamSynchronize(ctx->stream), "checksum");
    return CBC_OK;
}

/* Member k's stash -> out[sum of the earlier members' bytes ...], through member 0's device.  nbytes[k] / sums[k] (may be
 * NULL): what member k held and the checksum it took of it before the exchange; the call fails with CBC_E_IO when what
 * member 0 received sums to something else. */
API int cbc_gpu_group_gather(cbc_gpu_group *g, uint8_t *out, uint64_t out_cap, uint64_t *nbytes, uint64_t *sums)
{
    if (!g || !nbytes) return CBC_E_ARG;
    cbc_gpu_ctx *c0 = g->ctx[0];
    cbc_gpu_ctx *ctx = c0;                                      /* for HIPCHK / set_err */
    uint64_t total = 0, sent[64], got = 0;
    for (int k = 0; k < g->n; k++) { nbytes[k] = g->ctx[k]->stash_len; total += nbytes[k]; }
    if (total > out_cap || (total && !out)) return set_err(c0, CBC_E_ARG, "out_cap too small for the gathered bitstreams", hipSuccess);
    for (int k = 0; k < g->n; k++) {                          /* every member sums its own bytes on its own device */
        cbc_gpu_ctx *ck = g->ctx[k];
        HIPCHK(hipSetDevice(ck->device), "hipSetDevice");
        int rc = device_checksum_now(ck, (const uint8_t *)ck->arena[A_STASH].p, nbytes[k], &sent[k]); if (rc) return rc;
        if (sums) sums[k] = sent[k];
    }
    HIPCHK(hipSetDevice(c0->device), "hipSetDevice");
    { int rc = arena_need(c0, A_GATHER, total + 16, "hipMalloc gather"); if (rc) return rc; }
    uint8_t *dst = (uint8_t *)c0->arena[A_GATHER].p;
    /* one group: every send and its receive (a one-member group sends to itself: the self-test of the call sites) */
    ncclResult_t r = g->api.GroupStart();
    uint64_t at = 0;
    for (int k = 0; k < g->n && r == ncclSuccess; k++) {
        if (nbytes[k]) {
            (void)hipSetDevice(g->ctx[k]->device);
            r = g->api.Send(g->ctx[k]->arena[A_STASH].p, (size_t)nbytes[k], ncclUint8, 0, g->comm[k], g->ctx[k]->stream);
            if (r == ncclSuccess) { (void)hipSetDevice(c0->device); r = g->api.Recv(dst + at, (size_t)nbytes[k], ncclUint8, k, g->comm[0], c0->stream); }
        }
        at += nbytes[k];
    }
    { const ncclResult_t e = g->api.GroupEnd(); if (r == ncclSuccess) r = e; }
    if (r != ncclSuccess) { snprintf(g->err, sizeof g->err, "RCCL send/recv: %s", g->api.GetErrorString(r)); return set_err(c0, CBC_E_NODEV, g->err, hipSuccess); }
    for (int k = 0; k < g->n; k++) { HIPCHK(hipSetDevice(g->ctx[k]->device), "hipSetDevice"); HIPCHK(hipStreamSynchronize(g->ctx[k]->stream), "RCCL exchange"); }
    HIPCHK(hipSetDevice(c0->device), "hipSetDevice");
    at = 0;
    for (int k = 0; k < g->n; k++) {                          /* ... and member 0 sums what arrived */
        int rc = device_checksum_now(c0, dst + at, nbytes[k], &got); if (rc) return rc;
        if (got != sent[k]) { snprintf(g->err, sizeof g->err, "member %d: checksum %016llx sent, %016llx received", k, (unsigned long long)sent[k], (unsigned long long)got); return set_err(c0, CBC_E_IO, g->err, hipSuccess); }
        at += nbytes[k];
    }
    if (total) HIPCHK(hipMemcpy(out, dst, total, hipMemcpyDeviceToHost), "D2H gathered bitstreams");
    return CBC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * decode direction
 * ---------------------------------------------------------------------------------------------- */
API uint32_t cbc_gpu_decode_lds_bytes(const cbc_lds_caps *caps) { return caps ? cbc_plan_dec_lds_bytes(caps) : 0; }

API int cbc_gpu_decode_blocks_device(cbc_gpu_ctx *ctx, const cbc_dec_device_batch *b, void *hip_stream)
{
    if (!ctx || !b) return CBC_E_ARG;
    if (b->n_blocks == 0) return CBC_OK;
    if (!b->d_in || !b->d_blocks || !b->d_ref || !b->d_recs || !b->d_seq || !b->d_results || !b->d_var_scratch)
        return set_err(ctx, CBC_E_ARG, "null device pointer in cbc_dec_device_batch", hipSuccess);
    if (b->caps.cap_pos < 2 || b->caps.cap_pos > 8192 || b->caps.cap_var < 1 || b->caps.cap_var > 32768)
        return set_err(ctx, CBC_E_ARG, "lds caps out of range", hipSuccess);
    const uint32_t lds = cbc_plan_dec_lds_bytes(&b->caps);
    if (lds > 160u * 1024u) return set_err(ctx, CBC_E_ARG, "lds caps need more than 160 KiB", hipSuccess);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t s = hip_stream == CBC_CTX_STREAM ? ctx->stream : (hipStream_t)hip_stream;
    cbc_dec_args A;
    A.in = b->d_in; A.blocks = b->d_blocks; A.ref = b->d_ref; A.recs = b->d_recs; A.seq = b->d_seq; A.results = b->d_results;
    A.in_bytes = b->in_bytes; A.ref_bytes = b->ref_bytes; A.n_recs = b->n_recs; A.seq_bytes = b->seq_bytes;
    A.n_blocks = b->n_blocks; A.cap_pos = b->caps.cap_pos; A.cap_var = b->caps.cap_var;
    A.var_scratch = b->d_var_scratch; A.var_scratch_words = b->var_scratch_words;
    HIPCHK(hipEventRecord(ctx->ev0, s), "hipEventRecord");
    hipLaunchKernelGGL(cbc_decode_blocks_kernel, dim3(b->n_blocks), dim3(64), lds, s, A);
    HIPCHK(hipGetLastError(), "launch cbc_decode_blocks_kernel");
    HIPCHK(hipEventRecord(ctx->ev1, s), "hipEventRecord");
    ctx->have_timing = 1;
    return CBC_OK;
}

/* The host-buffer decode path as a pipeline, mirror of encode_blocks_impl: the payloads (2 bytes per read) go H2D at once;
 * the blocks are decoded in chunks on the two kernel streams, and chunk c's records and bases (bytes, or 2-bit rows packed by
 * cbc_pack_2bit_kernel) come back on the copy stream while the later chunks are still being decoded.  Device arrays are the
 * context's arenas. */
static int decode_blocks_impl(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes, cbc_dec_block_desc *blocks,
                              uint32_t n_blocks, const cbc_lds_caps *caps, cbc_read_rec *recs, uint64_t n_recs,
                              uint8_t *seq, uint64_t seq_bytes, uint32_t *codes_out, uint64_t *exc_idx, uint8_t *exc_val,
                              uint64_t exc_cap, uint64_t *n_exc, cbc_block_result *results)
{
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const double T0 = wall_now();
    cbc_e2e_times tm; memset(&tm, 0, sizeof tm);
    const bool two_bit = codes_out != NULL;
    const uint32_t stride = blocks[0].seq_stride;
    cbc_block_result *res = NULL;
    int rc = CBC_OK;
    const uint64_t vs_words = (uint64_t)n_blocks * caps->cap_var;
    const uint64_t n_words = two_bit ? n_recs * (stride >> 4) : 0;
    unsigned long long got = 0;
#define GO(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = set_err(ctx, CBC_E_NODEV, what, e_); goto done; } } while (0)
#define NEED(k, bytes, what) do { rc = arena_need(ctx, k, bytes, what); if (rc) goto done; } while (0)
    NEED(A_VS, vs_words * 4 + 16, "hipMalloc var scratch");
    NEED(A_IN, in_bytes + 16, "hipMalloc in");
    NEED(A_BLOCKS, (uint64_t)n_blocks * sizeof(cbc_dec_block_desc), "hipMalloc blocks");
    NEED(A_RECS, n_recs * sizeof(cbc_read_rec) + 16, "hipMalloc recs");
    NEED(A_SEQ, seq_bytes + 32, "hipMalloc seq");
    NEED(A_RES, (uint64_t)n_blocks * sizeof(cbc_block_result), "hipMalloc results");
    if (two_bit) {
        NEED(A_CODES, n_words * 4 + 16, "hipMalloc codes");
        NEED(A_EXC_I, (exc_cap ? exc_cap : 1) * 8, "hipMalloc exceptions");
        NEED(A_EXC_V, (exc_cap ? exc_cap : 1), "hipMalloc exceptions");
        NEED(A_CNT, 8, "hipMalloc counter");
    }
    tm.alloc_s = wall_now() - T0;
    {
        uint8_t *d_in = (uint8_t *)ctx->arena[A_IN].p, *d_seq = (uint8_t *)ctx->arena[A_SEQ].p;
        cbc_dec_block_desc *d_blocks = (cbc_dec_block_desc *)ctx->arena[A_BLOCKS].p;
        cbc_read_rec *d_recs = (cbc_read_rec *)ctx->arena[A_RECS].p;
        cbc_block_result *d_res = (cbc_block_result *)ctx->arena[A_RES].p;
        hipStream_t sc = ctx->s_copy;
        GO(hipMemsetAsync(d_in + in_bytes, 0, 16, sc), "memset pad");
        GO(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, sc), "H2D payloads");
        GO(hipMemcpyAsync(d_blocks, blocks, (uint64_t)n_blocks * sizeof(cbc_dec_block_desc), hipMemcpyHostToDevice, sc), "H2D blocks");
        GO(hipMemsetAsync(d_res, 0xff, (uint64_t)n_blocks * sizeof(cbc_block_result), sc), "memset results");
        if (two_bit) GO(hipMemsetAsync(ctx->arena[A_CNT].p, 0, 8, sc), "memset counter");
        GO(hipEventRecord(ctx->ev_done[0], sc), "hipEventRecord");       /* inputs are on the device */
        tm.h2d_bytes = in_bytes + (uint64_t)n_blocks * sizeof(cbc_dec_block_desc);
        /* chunks of consecutive blocks whose outputs are consecutive too */
        uint32_t n_chunks = 1, cut[CBC_MAX_CHUNKS + 1];
        bool contiguous = true;
        for (uint32_t b = 0; b + 1 < n_blocks && contiguous; b++)
            contiguous = blocks[b + 1].rec_base == blocks[b].rec_base + blocks[b].n_reads && blocks[b + 1].seq_base >= blocks[b].seq_base;
        contiguous = contiguous && blocks[n_blocks - 1].rec_base + blocks[n_blocks - 1].n_reads <= n_recs && blocks[n_blocks - 1].seq_base <= seq_bytes;
        if (contiguous && n_blocks >= 512) {
            uint64_t want = (n_recs * 16 + (two_bit ? n_words * 4 : seq_bytes)) / (64ull << 20);
            if (want > CBC_MAX_CHUNKS) want = CBC_MAX_CHUNKS;
            if (want > n_blocks / 256) want = n_blocks / 256;
            if (want >= 2) n_chunks = (uint32_t)want;
        }
        cut[0] = 0; cut[n_chunks] = n_blocks;
        for (uint32_t c = 1; c < n_chunks; c++) {
            const uint64_t target = blocks[0].rec_base + (blocks[n_blocks - 1].rec_base + blocks[n_blocks - 1].n_reads - blocks[0].rec_base) * c / n_chunks;
            uint32_t lo = cut[c - 1] + 1, hi = n_blocks - (n_chunks - c);
            while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (blocks[mid].rec_base < target) lo = mid + 1; else hi = mid; }
            cut[c] = lo;
        }
        tm.n_chunks = n_chunks;
        for (uint32_t c = 0; c < n_chunks; c++) {
            const uint32_t c0 = cut[c], c1 = cut[c + 1];
            const bool whole = !contiguous || n_chunks == 1;
            const uint64_t s0 = whole ? 0 : blocks[c0].seq_base, s1 = whole || c1 == n_blocks ? seq_bytes : blocks[c1].seq_base;
            hipStream_t ks = ctx->s_k[c % CBC_N_KSTREAMS];
            GO(hipStreamWaitEvent(ks, ctx->ev_done[0], 0), "hipStreamWaitEvent");
            if (s1 > s0) GO(hipMemsetAsync(d_seq + s0, 0, s1 - s0 + (c1 == n_blocks || whole ? 32 : 0), ks), "memset seq");
            cbc_dec_device_batch db;
            memset(&db, 0, sizeof db);
            db.d_in = d_in; db.in_bytes = in_bytes + 16; db.d_blocks = d_blocks + c0; db.n_blocks = c1 - c0;
            db.d_ref = ctx->d_ref; db.ref_bytes = ctx->ref_bytes; db.d_recs = d_recs; db.n_recs = n_recs;
            db.d_seq = d_seq; db.seq_bytes = seq_bytes + 32; db.d_results = d_res + c0;
            db.caps = *caps; db.d_var_scratch = (uint32_t *)ctx->arena[A_VS].p + (uint64_t)c0 * caps->cap_var;
            db.var_scratch_words = (uint64_t)(c1 - c0) * caps->cap_var;
            rc = cbc_gpu_decode_blocks_device(ctx, &db, ks);
            if (rc) goto done;
            const uint64_t r0 = whole ? 0 : blocks[c0].rec_base, r1 = whole ? n_recs : blocks[c1 - 1].rec_base + blocks[c1 - 1].n_reads;
            if (two_bit && r1 > r0) {
                const uint64_t w0 = r0 * (stride >> 4), w1 = r1 * (stride >> 4);
                hipLaunchKernelGGL(cbc_pack_2bit_kernel, dim3((unsigned)((w1 - w0 + 255) / 256)), dim3(256), 0, ks,
                                   (const uint8_t *)d_seq, (const cbc_read_rec *)d_recs, r0, r1, stride, (uint32_t *)ctx->arena[A_CODES].p,
                                   (uint64_t *)ctx->arena[A_EXC_I].p, (uint8_t *)ctx->arena[A_EXC_V].p, exc_cap, (unsigned long long *)ctx->arena[A_CNT].p);
                GO(hipGetLastError(), "launch cbc_pack_2bit_kernel");
            }
            GO(hipEventRecord(ctx->ev_chunk[c], ks), "hipEventRecord");
        }
        tm.issue_s = wall_now() - T0;
        for (uint32_t c = 0; c < n_chunks; c++) {                 /* the chunks come back in order while later ones are being decoded */
            const uint32_t c0 = cut[c], c1 = cut[c + 1];
            const bool whole = !contiguous || n_chunks == 1;
            const uint64_t r0 = whole ? 0 : blocks[c0].rec_base, r1 = whole ? n_recs : blocks[c1 - 1].rec_base + blocks[c1 - 1].n_reads;
            const uint64_t s0 = whole ? 0 : blocks[c0].seq_base, s1 = whole || c1 == n_blocks ? seq_bytes : blocks[c1].seq_base;
            GO(hipStreamWaitEvent(sc, ctx->ev_chunk[c], 0), "hipStreamWaitEvent");
            if (r1 > r0) GO(hipMemcpyAsync(recs + r0, d_recs + r0, (r1 - r0) * sizeof(cbc_read_rec), hipMemcpyDeviceToHost, sc), "D2H recs");
            tm.d2h_bytes += (r1 - r0) * sizeof(cbc_read_rec);
            if (two_bit) {
                const uint64_t w0 = r0 * (stride >> 4), w1 = r1 * (stride >> 4);
                if (w1 > w0) GO(hipMemcpyAsync(codes_out + w0, (uint32_t *)ctx->arena[A_CODES].p + w0, (w1 - w0) * 4, hipMemcpyDeviceToHost, sc), "D2H codes");
                tm.d2h_bytes += (w1 - w0) * 4;
            } else if (s1 > s0) {
                GO(hipMemcpyAsync(seq + s0, d_seq + s0, s1 - s0, hipMemcpyDeviceToHost, sc), "D2H seq");
                tm.d2h_bytes += s1 - s0;
            }
        }
        res = results ? results : (cbc_block_result *)malloc((size_t)n_blocks * sizeof(cbc_block_result));
        if (!res) { rc = CBC_E_NOMEM; goto done; }
        GO(hipMemcpyAsync(res, d_res, (uint64_t)n_blocks * sizeof(cbc_block_result), hipMemcpyDeviceToHost, sc), "D2H results");
        if (two_bit) GO(hipMemcpyAsync(&got, ctx->arena[A_CNT].p, 8, hipMemcpyDeviceToHost, sc), "D2H counter");
        GO(hipStreamSynchronize(sc), "decode kernel");
        tm.kernels_done_s = wall_now() - T0;
        if (two_bit) {
            *n_exc = got;
            if (got > exc_cap) { rc = set_err(ctx, CBC_E_ARG, "more non-ACGT bases than exc_cap", hipSuccess); goto done; }
            if (got) {
                GO(hipMemcpyAsync(exc_idx, ctx->arena[A_EXC_I].p, got * 8, hipMemcpyDeviceToHost, sc), "D2H exceptions");
                GO(hipMemcpyAsync(exc_val, ctx->arena[A_EXC_V].p, got, hipMemcpyDeviceToHost, sc), "D2H exceptions");
                GO(hipStreamSynchronize(sc), "D2H exceptions");
                tm.d2h_bytes += got * 9;
            }
        }
    }
    for (uint32_t b = 0; b < n_blocks; b++)
        if (res[b].status != CBC_ST_OK && rc == CBC_OK) {
            snprintf(ctx->err, sizeof ctx->err, "block %u failed to decode with status %u at record %u", b, res[b].status, res[b].fail_read);
            rc = CBC_E_BLOCK;
        }
done:
#undef GO
#undef NEED
    if (rc && rc != CBC_E_BLOCK) (void)hipDeviceSynchronize();
    if (res && res != results) free(res);
    tm.total_s = wall_now() - T0;
    ctx->last_e2e = tm;
    return rc;
}

API int cbc_gpu_decode_blocks(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes, cbc_dec_block_desc *blocks,
                              uint32_t n_blocks, const cbc_lds_caps *caps, cbc_read_rec *recs, uint64_t n_recs,
                              uint8_t *seq, uint64_t seq_bytes, cbc_block_result *results)
{
    if (!ctx || !in || !blocks || !caps || !recs || !seq) return CBC_E_ARG;
    if (!ctx->d_ref) return set_err(ctx, CBC_E_ARG, "cbc_gpu_upload_reference has not been called", hipSuccess);
    if (n_blocks == 0) return CBC_OK;
    return decode_blocks_impl(ctx, in, in_bytes, blocks, n_blocks, caps, recs, n_recs, seq, seq_bytes, NULL, NULL, NULL, 0, NULL, results);
}

/* ------------------------------------------------------------------------------------------------
 * whole-file stream ("compat" mode) and the general-form fallback over blocks
 * ---------------------------------------------------------------------------------------------- */
API uint32_t cbc_stream_read_length(const uint8_t *in, uint64_t in_bytes)
{
    /* the header's first int goes through four untouched 256-symbol models: its bytes come out verbatim */
    if (!in || in_bytes < 4) return 0;
    return ((uint32_t)in[0] << 24) | ((uint32_t)in[1] << 16) | ((uint32_t)in[2] << 8) | in[3];
}

static int stream_encode(cbc_gpu_ctx *ctx, const cbc_host_batch *hb, int per_segment, uint8_t *out, uint64_t out_cap,
                         uint64_t *out_offsets, cbc_block_result *results, cbc_stream_result *sres)
{
    if (!ctx || !hb || !out) return CBC_E_ARG;
    if (!ctx->d_ref) return set_err(ctx, CBC_E_ARG, "cbc_gpu_upload_reference has not been called", hipSuccess);
    const uint32_t nb = hb->n_blocks;
    if (nb == 0) return set_err(ctx, CBC_E_ARG, "no records", hipSuccess);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const uint32_t n_streams = per_segment ? nb : 1u;
    cbc_block_desc *segs = (cbc_block_desc *)malloc((size_t)nb * sizeof(cbc_block_desc));
    cbc_block_result *res = (cbc_block_result *)malloc((size_t)n_streams * sizeof(cbc_block_result));
    if (!segs || !res) { free(segs); free(res); return CBC_E_NOMEM; }
    memcpy(segs, hb->blocks, (size_t)nb * sizeof(cbc_block_desc));
    /* output areas: < 20 bits per coded symbol; symbols per record <= 17 + 2 per edit (cf. cbc_plan_output) */
    uint64_t scratch = 0;
    {
        uint64_t whole = 4096;
        for (uint32_t b = 0; b < nb; b++) {
            uint64_t cap = (4096 + 48ull * segs[b].n_reads + 8ull * segs[b].n_tok + 255) & ~255ull;
            if (per_segment) {
                if (cap > 0xffffff00ull) cap = 0xffffff00ull;
                segs[b].out_off = scratch; segs[b].out_cap = (uint32_t)cap; scratch += cap;
            } else whole += cap;
        }
        if (!per_segment) {
            if (whole > 0xffffff00ull) whole = 0xffffff00ull;
            whole &= ~255ull;
            for (uint32_t b = 0; b < nb; b++) { segs[b].out_off = 0; segs[b].out_cap = (uint32_t)whole; }
            scratch = whole;
        }
    }
    cbc_stream_caps caps; caps.cap_pos = hb->caps.cap_pos < 64 ? 64 : hb->caps.cap_pos; caps.cap_name = hb->names_bytes + 2u * (per_segment ? 1u : nb) + 16u;
    const uint32_t lds = cbc_stream_lds_bytes(&caps);
    int rc = CBC_OK;
    void *d_recs = NULL, *d_seq = NULL, *d_tok = NULL, *d_names = NULL, *d_segs = NULL, *d_out = NULL, *d_res = NULL, *d_vtab = NULL, *d_aux = NULL;
    const uint64_t ntok = hb->n_tok ? hb->n_tok : 1;
    uint32_t grid = n_streams < 32u ? n_streams : 32u;            /* pool of var tables: 67 MB each */
    if (caps.cap_pos > CBC_STREAM_POS_MAX) { free(segs); free(res); return set_err(ctx, CBC_E_ARG, "cap_pos beyond MAX_ALPHA", hipSuccess); }
    if (lds > 160u * 1024u) { free(segs); free(res); return set_err(ctx, CBC_E_ARG, "stream tables need more than 160 KiB of LDS", hipSuccess); }
#define GO(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = set_err(ctx, CBC_E_NODEV, what, e_); goto done; } } while (0)
    GO(hipMalloc(&d_recs, hb->n_recs * sizeof(cbc_read_rec) + 16), "hipMalloc recs");
    GO(hipMalloc(&d_seq, hb->seq_bytes + 16), "hipMalloc seq");
    GO(hipMalloc(&d_tok, ntok * 4 + 16), "hipMalloc tok");
    GO(hipMalloc(&d_names, hb->names_bytes + 16), "hipMalloc names");
    GO(hipMalloc(&d_segs, (uint64_t)nb * sizeof(cbc_block_desc)), "hipMalloc segments");
    GO(hipMalloc(&d_out, scratch), "hipMalloc out");
    GO(hipMalloc(&d_res, (uint64_t)n_streams * sizeof(cbc_block_result)), "hipMalloc results");
    GO(hipMalloc(&d_vtab, (uint64_t)grid * CBC_VTAB_WORDS * 4), "hipMalloc var tables");
    GO(hipMemsetAsync(d_vtab, 0, (uint64_t)grid * CBC_VTAB_WORDS * 4, ctx->stream), "memset var tables");
    GO(hipMalloc(&d_aux, (uint64_t)grid * cbc_stream_aux_words(caps.cap_pos) * 4), "hipMalloc flag / pos overflow tables");
    GO(hipMemcpyAsync(d_recs, hb->recs, hb->n_recs * sizeof(cbc_read_rec), hipMemcpyHostToDevice, ctx->stream), "H2D recs");
    GO(hipMemcpyAsync(d_seq, hb->seq, hb->seq_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D seq");
    GO(hipMemcpyAsync(d_tok, hb->tok, hb->n_tok * 4, hipMemcpyHostToDevice, ctx->stream), "H2D tok");
    GO(hipMemcpyAsync(d_names, hb->names, hb->names_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D names");
    GO(hipMemcpyAsync(d_segs, segs, (uint64_t)nb * sizeof(cbc_block_desc), hipMemcpyHostToDevice, ctx->stream), "H2D segments");
    GO(hipMemsetAsync(d_res, 0xff, (uint64_t)n_streams * sizeof(cbc_block_result), ctx->stream), "memset results");
    {
        cbc_stream_args A;
        memset(&A, 0, sizeof A);
        A.recs = (const cbc_read_rec *)d_recs; A.seq = (const uint8_t *)d_seq; A.tok = (const uint32_t *)d_tok;
        A.names = (const uint8_t *)d_names; A.segs = (const cbc_block_desc *)d_segs; A.ref = ctx->d_ref;
        A.out = (uint8_t *)d_out; A.results = (cbc_block_result *)d_res; A.vtab = (uint32_t *)d_vtab; A.aux = (uint32_t *)d_aux;
        A.ref_bytes = ctx->ref_bytes; A.out_bytes = scratch; A.seq_bytes = hb->seq_bytes; A.n_tok = ntok; A.n_recs = hb->n_recs;
        A.n_segs = nb; A.cap_pos = caps.cap_pos; A.cap_name = caps.cap_name; A.names_bytes = hb->names_bytes;
        A.per_segment = per_segment ? 1u : 0u; A.n_vtab = grid;
        GO(hipEventRecord(ctx->ev0, ctx->stream), "hipEventRecord");
        hipLaunchKernelGGL(cbc_encode_whole_kernel, dim3(grid), dim3(64), lds, ctx->stream, A);
        GO(hipGetLastError(), "launch cbc_encode_whole_kernel");
        GO(hipEventRecord(ctx->ev1, ctx->stream), "hipEventRecord");
        ctx->have_timing = 1;
    }
    GO(hipMemcpyAsync(res, d_res, (uint64_t)n_streams * sizeof(cbc_block_result), hipMemcpyDeviceToHost, ctx->stream), "D2H results");
    GO(hipStreamSynchronize(ctx->stream), "stream encode kernel");
    {
        uint64_t off = 0;
        if (out_offsets) out_offsets[0] = 0;
        for (uint32_t s = 0; s < n_streams; s++) {
            if (res[s].status != CBC_ST_OK) {
                if (rc == CBC_OK) { snprintf(ctx->err, sizeof ctx->err, "stream %u failed with status %u at record %u", s, res[s].status, res[s].fail_read); rc = CBC_E_BLOCK; }
                res[s].nbytes = 0;
            }
            if (off + res[s].nbytes > out_cap) { rc = set_err(ctx, CBC_E_ARG, "out_cap too small for the stream", hipSuccess); goto done; }
            if (res[s].nbytes) GO(hipMemcpyAsync(out + off, (uint8_t *)d_out + segs[per_segment ? s : 0].out_off, res[s].nbytes, hipMemcpyDeviceToHost, ctx->stream), "D2H stream");
            off += res[s].nbytes;
            if (out_offsets) out_offsets[s + 1] = off;
            if (results) results[s] = res[s];
        }
        GO(hipStreamSynchronize(ctx->stream), "D2H stream");
        if (sres) { sres->nbytes = off; sres->status = res[0].status; sres->fail_read = res[0].fail_read; sres->n_symbols = res[0].n_symbols; }
    }
done:
#undef GO
    free(segs); free(res);
    if (d_recs) (void)hipFree(d_recs); if (d_seq) (void)hipFree(d_seq); if (d_tok) (void)hipFree(d_tok);
    if (d_names) (void)hipFree(d_names); if (d_segs) (void)hipFree(d_segs); if (d_out) (void)hipFree(d_out);
    if (d_res) (void)hipFree(d_res); if (d_vtab) (void)hipFree(d_vtab); if (d_aux) (void)hipFree(d_aux);
    return rc;
}

API int cbc_gpu_encode_stream(cbc_gpu_ctx *ctx, const cbc_host_batch *hb, uint8_t *out, uint64_t out_cap, cbc_stream_result *result)
{
    return stream_encode(ctx, hb, 0, out, out_cap, NULL, NULL, result);
}
API int cbc_gpu_encode_stream_blocks(cbc_gpu_ctx *ctx, const cbc_host_batch *hb, uint8_t *out, uint64_t out_cap,
                                     uint64_t *out_offsets, cbc_block_result *results)
{
    if (!out_offsets) return CBC_E_ARG;
    return stream_encode(ctx, hb, 1, out, out_cap, out_offsets, results, NULL);
}

API int cbc_gpu_decode_stream(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes,
                              const uint64_t *contig_off, const uint64_t *contig_len, uint32_t n_contigs,
                              cbc_read_rec *recs, uint64_t rec_cap, uint8_t *seq, uint64_t seq_bytes, uint32_t seq_stride,
                              cbc_stream_result *result)
{
    if (!ctx || !in || !contig_off || !contig_len || !recs || !seq || !result || n_contigs == 0) return CBC_E_ARG;
    if (!ctx->d_ref) return set_err(ctx, CBC_E_ARG, "cbc_gpu_upload_reference has not been called", hipSuccess);
    const uint32_t L0 = cbc_stream_read_length(in, in_bytes);
    if (L0 < 1 || L0 > 256 || seq_stride < 4 || seq_stride > 256 || (seq_stride & 3u) || rec_cap == 0 || rec_cap > 0xffffffffull ||
        seq_bytes < rec_cap * seq_stride + 8) return set_err(ctx, CBC_E_ARG, "bad stream header or buffer sizes", hipSuccess);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    /* the stream does not say how many distinct POS steps it holds: room for the reference's whole alphabet (MAX_ALPHA),
     * CBC_STREAM_POS_LDS entries of it in LDS, the rest in the aux area (40 MB) */
    cbc_stream_caps caps; caps.cap_pos = CBC_STREAM_POS_MAX; caps.cap_name = 2048;
    const uint32_t lds = cbc_stream_lds_bytes(&caps);
    void *d_in = NULL, *d_co = NULL, *d_cl = NULL, *d_recs = NULL, *d_seq = NULL, *d_res = NULL, *d_vtab = NULL, *d_aux = NULL;
    cbc_block_result res; memset(&res, 0xff, sizeof res);
    int rc = CBC_OK;
#define GO(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = set_err(ctx, CBC_E_NODEV, what, e_); goto done; } } while (0)
    GO(hipMalloc(&d_in, in_bytes + 16), "hipMalloc in");
    GO(hipMalloc(&d_co, (uint64_t)n_contigs * 8), "hipMalloc contigs");
    GO(hipMalloc(&d_cl, (uint64_t)n_contigs * 8), "hipMalloc contigs");
    GO(hipMalloc(&d_recs, rec_cap * sizeof(cbc_read_rec) + 16), "hipMalloc recs");
    GO(hipMalloc(&d_seq, seq_bytes + 16), "hipMalloc seq");
    GO(hipMalloc(&d_res, sizeof(cbc_block_result)), "hipMalloc result");
    GO(hipMalloc(&d_vtab, CBC_VTAB_WORDS * 4), "hipMalloc var table");
    GO(hipMemsetAsync(d_vtab, 0, CBC_VTAB_WORDS * 4, ctx->stream), "memset var table");
    GO(hipMalloc(&d_aux, cbc_stream_aux_words(caps.cap_pos) * 4), "hipMalloc flag / pos overflow tables");
    GO(hipMemsetAsync((uint8_t *)d_in + in_bytes, 0, 16, ctx->stream), "memset pad");
    GO(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D stream");
    GO(hipMemcpyAsync(d_co, contig_off, (uint64_t)n_contigs * 8, hipMemcpyHostToDevice, ctx->stream), "H2D contigs");
    GO(hipMemcpyAsync(d_cl, contig_len, (uint64_t)n_contigs * 8, hipMemcpyHostToDevice, ctx->stream), "H2D contigs");
    GO(hipMemsetAsync(d_res, 0xff, sizeof(cbc_block_result), ctx->stream), "memset result");
    {
        cbc_dstream_args A;
        memset(&A, 0, sizeof A);
        A.in = (const uint8_t *)d_in; A.ref = ctx->d_ref; A.contig_off = (const uint64_t *)d_co; A.contig_len = (const uint64_t *)d_cl;
        A.recs = (cbc_read_rec *)d_recs; A.seq = (uint8_t *)d_seq; A.results = (cbc_block_result *)d_res; A.vtab = (uint32_t *)d_vtab;
        A.aux = (uint32_t *)d_aux;
        A.in_bytes = in_bytes; A.ref_bytes = ctx->ref_bytes; A.rec_cap = rec_cap; A.seq_bytes = seq_bytes + 16;
        A.n_contigs = n_contigs; A.cap_pos = caps.cap_pos; A.cap_name = caps.cap_name; A.seq_stride = seq_stride; A.read_length = L0;
        GO(hipEventRecord(ctx->ev0, ctx->stream), "hipEventRecord");
        hipLaunchKernelGGL(cbc_decode_whole_kernel, dim3(1), dim3(64), lds, ctx->stream, A);
        GO(hipGetLastError(), "launch cbc_decode_whole_kernel");
        GO(hipEventRecord(ctx->ev1, ctx->stream), "hipEventRecord");
        ctx->have_timing = 1;
    }
    GO(hipMemcpyAsync(&res, d_res, sizeof res, hipMemcpyDeviceToHost, ctx->stream), "D2H result");
    GO(hipStreamSynchronize(ctx->stream), "stream decode kernel");
    result->nbytes = res.nbytes; result->status = res.status; result->fail_read = res.fail_read; result->n_symbols = res.n_symbols;
    if (res.nbytes <= rec_cap && res.nbytes) {
        GO(hipMemcpyAsync(recs, d_recs, (uint64_t)res.nbytes * sizeof(cbc_read_rec), hipMemcpyDeviceToHost, ctx->stream), "D2H recs");
        GO(hipMemcpyAsync(seq, d_seq, (uint64_t)res.nbytes * seq_stride, hipMemcpyDeviceToHost, ctx->stream), "D2H seq");
        GO(hipStreamSynchronize(ctx->stream), "D2H");
    }
    if (res.status != CBC_ST_OK) {
        snprintf(ctx->err, sizeof ctx->err, "stream decode stopped with status %u at record %u", res.status, res.fail_read);
        rc = CBC_E_BLOCK;
    }
done:
#undef GO
    if (d_in) (void)hipFree(d_in); if (d_co) (void)hipFree(d_co); if (d_cl) (void)hipFree(d_cl); if (d_recs) (void)hipFree(d_recs);
    if (d_seq) (void)hipFree(d_seq); if (d_res) (void)hipFree(d_res); if (d_vtab) (void)hipFree(d_vtab); if (d_aux) (void)hipFree(d_aux);
    return rc;
}

/* The decode twin of cbc_gpu_encode_stream_blocks: every block's payload is a stream of its own in the general (rescaling)
 * form of the models -- what a block of more than CBC_MAX_BLOCK_READS records needs, and what the block decoder refuses.
 * A block is decoded as a one-contig file whose contig is the block's reference window (decompress(), src/compression.c:
 * 173-216; the models never rescale-free here: src/stream_model.c:78-117).  One launch per block: a rare path. */
API int cbc_gpu_decode_stream_blocks(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes, const cbc_dec_block_desc *blocks,
                                     uint32_t n_blocks, cbc_read_rec *recs, uint64_t n_recs, uint8_t *seq, uint64_t seq_bytes,
                                     cbc_block_result *results)
{
    if (!ctx || !in || !blocks || !recs || !seq) return CBC_E_ARG;
    if (!ctx->d_ref) return set_err(ctx, CBC_E_ARG, "cbc_gpu_upload_reference has not been called", hipSuccess);
    int rc = CBC_OK;
    for (uint32_t b = 0; b < n_blocks; b++) {
        const cbc_dec_block_desc *bd = &blocks[b];
        cbc_stream_result sr; memset(&sr, 0, sizeof sr);
        int one = CBC_E_ARG;
        if (bd->in_off <= in_bytes && bd->in_bytes <= in_bytes - bd->in_off && bd->rec_base <= n_recs && bd->n_reads <= n_recs - bd->rec_base &&
            bd->seq_stride >= 4 && bd->seq_stride <= 256 && (bd->seq_stride & 3u) == 0 && bd->seq_base <= seq_bytes &&
            (uint64_t)bd->n_reads * bd->seq_stride + 8 <= seq_bytes - bd->seq_base && bd->ref_off + CBC_REF_PAD < ctx->ref_bytes) {
            const uint64_t co = bd->ref_off, cl = ctx->ref_bytes - bd->ref_off - CBC_REF_PAD;      /* the window: from POS 1 of the block to the end */
            one = cbc_gpu_decode_stream(ctx, in + bd->in_off, bd->in_bytes, &co, &cl, 1, recs + bd->rec_base, bd->n_reads,
                                        seq + bd->seq_base, (uint64_t)bd->n_reads * bd->seq_stride + 8, bd->seq_stride, &sr);
            if (one == CBC_OK && sr.nbytes != bd->n_reads) { sr.status = CBC_ST_ASSERT; one = CBC_E_BLOCK; }     /* the index and the stream disagree */
        }
        if (results) { results[b].nbytes = (uint32_t)sr.nbytes; results[b].status = one == CBC_E_ARG ? CBC_ST_ASSERT : sr.status;
                       results[b].n_symbols = (uint32_t)sr.n_symbols; results[b].fail_read = sr.fail_read; }
        if (one != CBC_OK && rc == CBC_OK) {
            rc = one == CBC_E_ARG ? CBC_E_ARG : CBC_E_BLOCK;
            if (one == CBC_E_ARG) (void)set_err(ctx, CBC_E_ARG, "block descriptor outside the buffers", hipSuccess);
        }
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * long-read format extension
 * ---------------------------------------------------------------------------------------------- */
API uint32_t cbc_gpu_long_lds_bytes(const cbc_lds_caps *caps) { return caps ? cbc_long_lds_bytes(caps->cap_pos) : 0; }

/* payload areas: 4096 + 64 per read + bytes_per_16_bases / 16 per base (typical streams need < 1 bit per base; the
 * worst case, every base an edit with an escaped gap, is 5 symbols of < 20 bits = 12.5 bytes per base = 200) */
API uint64_t cbc_gpu_long_plan_output(cbc_block_desc *blocks, uint32_t n_blocks, const cbc_read_rec *recs, uint32_t bytes_per_16_bases)
{
    uint64_t off = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint64_t bases = 0;
        for (uint32_t r = 0; r < blocks[b].n_reads; r++) bases += recs[blocks[b].rec_base + r].rlen;
        uint64_t cap = (4096 + 64ull * blocks[b].n_reads + bases * bytes_per_16_bases / 16 + 255) & ~255ull;
        if (cap > 0xffffff00ull) cap = 0xffffff00ull;
        blocks[b].out_off = off; blocks[b].out_cap = (uint32_t)cap; blocks[b].reserved = (uint32_t)cap;
        off += cap;
    }
    return off;
}

API int cbc_gpu_long_encode_blocks_device(cbc_gpu_ctx *ctx, const cbc_device_batch *b, void *hip_stream)
{
    if (!ctx || !b) return CBC_E_ARG;
    if (b->n_blocks == 0) return CBC_OK;
    if (!b->d_recs || !b->d_seq || !b->d_tok || !b->d_names || !b->d_blocks || !b->d_ref || !b->d_out || !b->d_results)
        return set_err(ctx, CBC_E_ARG, "null device pointer in cbc_device_batch", hipSuccess);
    if (b->caps.cap_pos < 2 || b->caps.cap_pos > 8192) return set_err(ctx, CBC_E_ARG, "lds caps out of range", hipSuccess);
    const uint32_t lds = cbc_long_lds_bytes(b->caps.cap_pos);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t s = hip_stream == CBC_CTX_STREAM ? ctx->stream : (hipStream_t)hip_stream;
    cbc_long_args A;
    A.recs = b->d_recs; A.seq = b->d_seq; A.tok = b->d_tok; A.names = b->d_names; A.blocks = b->d_blocks;
    A.ref = b->d_ref; A.out = b->d_out; A.results = b->d_results;
    A.ref_bytes = b->ref_bytes; A.out_bytes = b->out_bytes; A.seq_bytes = b->seq_bytes; A.n_tok = b->n_tok;
    A.n_recs = b->n_recs; A.n_blocks = b->n_blocks; A.cap_pos = b->caps.cap_pos; A.names_bytes = 0x7fffffffu;
    /* the blocks' global-memory table parts (gap symbols 64.., gx): a buffer of the context, grown when a larger batch
     * comes (which waits for the device once); the kernel zeroes what it uses */
    { int rc_ = arena_need(ctx, A_LSCR, (uint64_t)b->n_blocks * CBC_LONG_SCRATCH_WORDS * 4 + 256, "hipMalloc long-read table scratch"); if (rc_) return rc_; }
    A.scratch = (uint32_t *)ctx->arena[A_LSCR].p;
    HIPCHK(hipEventRecord(ctx->ev0, s), "hipEventRecord");
    hipLaunchKernelGGL(cbc_long_encode_kernel, dim3(b->n_blocks), dim3(192), lds, s, A);
    HIPCHK(hipGetLastError(), "launch cbc_long_encode_kernel");
    HIPCHK(hipEventRecord(ctx->ev1, s), "hipEventRecord");
    ctx->have_timing = 1; ctx->last_variant = 0;
    return CBC_OK;
}

API int cbc_gpu_long_decode_blocks_device(cbc_gpu_ctx *ctx, const cbc_dec_device_batch *b, void *hip_stream)
{
    if (!ctx || !b) return CBC_E_ARG;
    if (b->n_blocks == 0) return CBC_OK;
    if (!b->d_in || !b->d_blocks || !b->d_ref || !b->d_recs || !b->d_seq || !b->d_results)
        return set_err(ctx, CBC_E_ARG, "null device pointer in cbc_dec_device_batch", hipSuccess);
    if (b->caps.cap_pos < 2 || b->caps.cap_pos > 8192) return set_err(ctx, CBC_E_ARG, "lds caps out of range", hipSuccess);
    const uint32_t lds = cbc_long_dec_lds_bytes(b->caps.cap_pos);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t s = hip_stream == CBC_CTX_STREAM ? ctx->stream : (hipStream_t)hip_stream;
    cbc_dec_args A;
    memset(&A, 0, sizeof A);
    A.in = b->d_in; A.blocks = b->d_blocks; A.ref = b->d_ref; A.recs = b->d_recs; A.seq = b->d_seq; A.results = b->d_results;
    A.in_bytes = b->in_bytes; A.ref_bytes = b->ref_bytes; A.n_recs = b->n_recs; A.seq_bytes = b->seq_bytes;
    A.n_blocks = b->n_blocks; A.cap_pos = b->caps.cap_pos; A.cap_var = 0;
    /* the blocks' global-memory table parts (cbc_long_body.h): the context's own buffer -- batch->d_var_scratch is not used */
    { int rc_ = arena_need(ctx, A_LSCR, (uint64_t)b->n_blocks * CBC_LONG_TABLE_WORDS * 4 + 256, "hipMalloc long-read table scratch"); if (rc_) return rc_; }
    A.var_scratch = (uint32_t *)ctx->arena[A_LSCR].p; A.var_scratch_words = (uint64_t)b->n_blocks * CBC_LONG_TABLE_WORDS;
    HIPCHK(hipEventRecord(ctx->ev0, s), "hipEventRecord");
    hipLaunchKernelGGL(cbc_long_decode_kernel, dim3(b->n_blocks), dim3(64), lds, s, A);
    HIPCHK(hipGetLastError(), "launch cbc_long_decode_kernel");
    HIPCHK(hipEventRecord(ctx->ev1, s), "hipEventRecord");
    ctx->have_timing = 1;
    return CBC_OK;
}

/* host buffers in, compacted payloads out; a block whose area was too small (CBC_ST_OUT_FULL) makes the whole batch
 * run once more with the worst-case areas */
API int cbc_gpu_long_encode_blocks(cbc_gpu_ctx *ctx, const cbc_host_batch *hb, uint8_t *out, uint64_t out_cap,
                                   uint64_t *out_offsets, cbc_block_result *results)
{
    if (!ctx || !hb || !out || !out_offsets) return CBC_E_ARG;
    if (!ctx->d_ref) return set_err(ctx, CBC_E_ARG, "cbc_gpu_upload_reference has not been called", hipSuccess);
    const uint32_t nb = hb->n_blocks;
    out_offsets[0] = 0;
    if (nb == 0) return CBC_OK;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    void *d_recs = NULL, *d_seq = NULL, *d_tok = NULL, *d_names = NULL, *d_blocks = NULL, *d_out = NULL, *d_res = NULL, *d_off = NULL, *d_packed = NULL;
    cbc_block_result *res = NULL;
    int rc = CBC_OK;
    uint64_t total = 0, scratch = 0;
    const uint64_t ntok = hb->n_tok ? hb->n_tok : 1;
#define GO(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = set_err(ctx, CBC_E_NODEV, what, e_); goto done; } } while (0)
    GO(hipMalloc(&d_recs, hb->n_recs * sizeof(cbc_read_rec) + 16), "hipMalloc recs");
    GO(hipMalloc(&d_seq, hb->seq_bytes + 16), "hipMalloc seq");
    GO(hipMalloc(&d_tok, ntok * 4 + 16), "hipMalloc tok");
    GO(hipMalloc(&d_names, hb->names_bytes + 16), "hipMalloc names");
    GO(hipMalloc(&d_blocks, (uint64_t)nb * sizeof(cbc_block_desc)), "hipMalloc blocks");
    GO(hipMalloc(&d_res, (uint64_t)nb * sizeof(cbc_block_result)), "hipMalloc results");
    GO(hipMalloc(&d_off, ((uint64_t)nb + 1) * 8), "hipMalloc offsets");
    GO(hipMemcpyAsync(d_recs, hb->recs, hb->n_recs * sizeof(cbc_read_rec), hipMemcpyHostToDevice, ctx->stream), "H2D recs");
    GO(hipMemcpyAsync(d_seq, hb->seq, hb->seq_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D seq");
    GO(hipMemcpyAsync(d_tok, hb->tok, hb->n_tok * 4, hipMemcpyHostToDevice, ctx->stream), "H2D tok");
    GO(hipMemcpyAsync(d_names, hb->names, hb->names_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D names");
    res = results ? results : (cbc_block_result *)malloc((size_t)nb * sizeof(cbc_block_result));
    if (!res) { rc = CBC_E_NOMEM; goto done; }
    for (uint32_t attempt = 0; attempt < 2; attempt++) {
        scratch = cbc_gpu_long_plan_output(hb->blocks, nb, hb->recs, attempt == 0 ? 8u : 200u);     /* 0.5, then 12.5 bytes per base */
        if (d_out) { (void)hipFree(d_out); d_out = NULL; }
        GO(hipMalloc(&d_out, scratch), "hipMalloc out scratch");
        GO(hipMemcpyAsync(d_blocks, hb->blocks, (uint64_t)nb * sizeof(cbc_block_desc), hipMemcpyHostToDevice, ctx->stream), "H2D blocks");
        GO(hipMemsetAsync(d_res, 0xff, (uint64_t)nb * sizeof(cbc_block_result), ctx->stream), "memset results");
        cbc_device_batch db;
        memset(&db, 0, sizeof db);
        db.d_recs = (const cbc_read_rec *)d_recs; db.d_seq = (const uint8_t *)d_seq; db.d_tok = (const uint32_t *)d_tok;
        db.d_names = (const uint8_t *)d_names; db.d_blocks = (const cbc_block_desc *)d_blocks; db.n_blocks = nb;
        db.d_ref = ctx->d_ref; db.ref_bytes = ctx->ref_bytes; db.d_out = (uint8_t *)d_out; db.out_bytes = scratch;
        db.d_results = (cbc_block_result *)d_res; db.seq_bytes = hb->seq_bytes; db.n_tok = ntok; db.n_recs = hb->n_recs; db.caps = hb->caps;
        rc = cbc_gpu_long_encode_blocks_device(ctx, &db, CBC_CTX_STREAM);
        if (rc) goto done;
        GO(hipMemcpyAsync(res, d_res, (uint64_t)nb * sizeof(cbc_block_result), hipMemcpyDeviceToHost, ctx->stream), "D2H results");
        GO(hipStreamSynchronize(ctx->stream), "long encode kernel");
        int full = 0;
        for (uint32_t b = 0; b < nb; b++) if (res[b].status == CBC_ST_OUT_FULL) full = 1;
        if (!full) break;
    }
    GO(hipMalloc(&d_packed, scratch), "hipMalloc packed");
    rc = cbc_gpu_compact_device(ctx, (const uint8_t *)d_out, (const cbc_block_desc *)d_blocks, (const cbc_block_result *)d_res,
                                nb, (uint64_t *)d_off, (uint8_t *)d_packed, scratch, CBC_CTX_STREAM);
    if (rc) goto done;
    GO(hipMemcpyAsync(out_offsets, d_off, ((uint64_t)nb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream), "D2H offsets");
    GO(hipStreamSynchronize(ctx->stream), "compaction");
    for (uint32_t b = 0; b < nb; b++)
        if (res[b].status != CBC_ST_OK && rc == CBC_OK) {
            snprintf(ctx->err, sizeof ctx->err, "block %u failed with status %u at record %u", b, res[b].status, res[b].fail_read);
            rc = CBC_E_BLOCK;
        }
    total = out_offsets[nb];
    if (total > out_cap) { rc = set_err(ctx, CBC_E_ARG, "out_cap too small for the compacted payloads", hipSuccess); goto done; }
    if (total) {
        GO(hipMemcpyAsync(out, d_packed, total, hipMemcpyDeviceToHost, ctx->stream), "D2H payloads");
        GO(hipStreamSynchronize(ctx->stream), "D2H payloads");
    }
done:
#undef GO
    if (res && res != results) free(res);
    if (d_recs) (void)hipFree(d_recs); if (d_seq) (void)hipFree(d_seq); if (d_tok) (void)hipFree(d_tok);
    if (d_names) (void)hipFree(d_names); if (d_blocks) (void)hipFree(d_blocks); if (d_out) (void)hipFree(d_out);
    if (d_res) (void)hipFree(d_res); if (d_off) (void)hipFree(d_off); if (d_packed) (void)hipFree(d_packed);
    return rc;
}

API int cbc_gpu_long_decode_blocks(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes, cbc_dec_block_desc *blocks,
                                   uint32_t n_blocks, const cbc_lds_caps *caps, cbc_read_rec *recs, uint64_t n_recs,
                                   uint8_t *seq, uint64_t seq_bytes, cbc_block_result *results)
{
    if (!ctx || !in || !blocks || !caps || !recs || !seq) return CBC_E_ARG;
    if (!ctx->d_ref) return set_err(ctx, CBC_E_ARG, "cbc_gpu_upload_reference has not been called", hipSuccess);
    if (n_blocks == 0) return CBC_OK;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    void *d_in = NULL, *d_blocks = NULL, *d_recs = NULL, *d_seq = NULL, *d_res = NULL;
    cbc_block_result *res = NULL;
    int rc = CBC_OK;
#define GO(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = set_err(ctx, CBC_E_NODEV, what, e_); goto done; } } while (0)
    GO(hipMalloc(&d_in, in_bytes + 16), "hipMalloc in");
    GO(hipMalloc(&d_blocks, (uint64_t)n_blocks * sizeof(cbc_dec_block_desc)), "hipMalloc blocks");
    GO(hipMalloc(&d_recs, n_recs * sizeof(cbc_read_rec) + 16), "hipMalloc recs");
    GO(hipMalloc(&d_seq, seq_bytes + 16), "hipMalloc seq");
    GO(hipMalloc(&d_res, (uint64_t)n_blocks * sizeof(cbc_block_result)), "hipMalloc results");
    GO(hipMemsetAsync((uint8_t *)d_in + in_bytes, 0, 16, ctx->stream), "memset pad");
    GO(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D payloads");
    GO(hipMemcpyAsync(d_blocks, blocks, (uint64_t)n_blocks * sizeof(cbc_dec_block_desc), hipMemcpyHostToDevice, ctx->stream), "H2D blocks");
    GO(hipMemsetAsync(d_res, 0xff, (uint64_t)n_blocks * sizeof(cbc_block_result), ctx->stream), "memset results");
    {
        cbc_dec_device_batch db;
        memset(&db, 0, sizeof db);
        db.d_in = (const uint8_t *)d_in; db.in_bytes = in_bytes + 16; db.d_blocks = (const cbc_dec_block_desc *)d_blocks;
        db.n_blocks = n_blocks; db.d_ref = ctx->d_ref; db.ref_bytes = ctx->ref_bytes; db.d_recs = (cbc_read_rec *)d_recs;
        db.n_recs = n_recs; db.d_seq = (uint8_t *)d_seq; db.seq_bytes = seq_bytes + 16; db.d_results = (cbc_block_result *)d_res;
        db.caps = *caps;
        rc = cbc_gpu_long_decode_blocks_device(ctx, &db, CBC_CTX_STREAM);
        if (rc) goto done;
    }
    res = results ? results : (cbc_block_result *)malloc((size_t)n_blocks * sizeof(cbc_block_result));
    if (!res) { rc = CBC_E_NOMEM; goto done; }
    GO(hipMemcpyAsync(res, d_res, (uint64_t)n_blocks * sizeof(cbc_block_result), hipMemcpyDeviceToHost, ctx->stream), "D2H results");
    GO(hipMemcpyAsync(recs, d_recs, n_recs * sizeof(cbc_read_rec), hipMemcpyDeviceToHost, ctx->stream), "D2H recs");
    GO(hipMemcpyAsync(seq, d_seq, seq_bytes, hipMemcpyDeviceToHost, ctx->stream), "D2H seq");
    GO(hipStreamSynchronize(ctx->stream), "long decode kernel");
    for (uint32_t b = 0; b < n_blocks; b++)
        if (res[b].status != CBC_ST_OK && rc == CBC_OK) {
            snprintf(ctx->err, sizeof ctx->err, "block %u failed to decode with status %u at record %u", b, res[b].status, res[b].fail_read);
            rc = CBC_E_BLOCK;
        }
done:
#undef GO
    if (res && res != results) free(res);
    if (d_in) (void)hipFree(d_in); if (d_blocks) (void)hipFree(d_blocks); if (d_recs) (void)hipFree(d_recs);
    if (d_seq) (void)hipFree(d_seq); if (d_res) (void)hipFree(d_res);
    return rc;
}

/* decode with the bases returned as 2-bit rows (include/cbc_gpu.h) */
API int cbc_gpu_decode_blocks_2bit(cbc_gpu_ctx *ctx, const uint8_t *in, uint64_t in_bytes, cbc_dec_block_desc *blocks,
                                   uint32_t n_blocks, const cbc_lds_caps *caps, cbc_read_rec *recs, uint64_t n_recs,
                                   uint32_t *codes_out, uint64_t *exc_idx, uint8_t *exc_val, uint64_t exc_cap, uint64_t *n_exc,
                                   cbc_block_result *results)
{
    if (!ctx || !in || !blocks || !caps || !recs || !codes_out || !n_exc || (exc_cap && (!exc_idx || !exc_val))) return CBC_E_ARG;
    if (!ctx->d_ref) return set_err(ctx, CBC_E_ARG, "cbc_gpu_upload_reference has not been called", hipSuccess);
    *n_exc = 0;
    if (n_blocks == 0) return CBC_OK;
    const uint32_t stride = blocks[0].seq_stride;
    if (stride < 16 || stride > 256 || (stride & 15u)) return set_err(ctx, CBC_E_ARG, "2-bit decode wants seq_stride to be a multiple of 16", hipSuccess);
    for (uint32_t b = 0; b < n_blocks; b++) if (blocks[b].seq_stride != stride || blocks[b].seq_base != blocks[b].rec_base * stride)
        return set_err(ctx, CBC_E_ARG, "2-bit decode wants one stride and seq_base = rec_base * stride", hipSuccess);
    return decode_blocks_impl(ctx, in, in_bytes, blocks, n_blocks, caps, recs, n_recs, NULL, n_recs * stride, codes_out, exc_idx, exc_val,
                              exc_cap, n_exc, results);
}

/* ------------------------------------------------------------------------------------------------
 * SAM text -> packed records on the device (cbc_tokenise.h)
 * ---------------------------------------------------------------------------------------------- */
static int scan_u32(cbc_gpu_ctx *ctx, const uint32_t *d_v, uint64_t n, uint64_t *d_out, uint64_t *d_tmp /* >= n/1024 + 2 */, uint64_t *grand)
{
    if (n == 0) { *grand = 0; return CBC_OK; }
    const uint64_t nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(cbc_scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_v, n, d_out, d_tmp);
    hipLaunchKernelGGL(cbc_scan_totals_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_tmp, nb, d_tmp + nb);
    hipLaunchKernelGGL(cbc_scan_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_out, n, (const uint64_t *)d_tmp);
    HIPCHK(hipGetLastError(), "launch scan kernels");
    HIPCHK(hipMemcpyAsync(grand, d_tmp + nb, 8, hipMemcpyDeviceToHost, ctx->stream), "D2H scan total");
    HIPCHK(hipStreamSynchronize(ctx->stream), "scan");
    return CBC_OK;
}

__global__ void __launch_bounds__(256)
cbc_tok_status_kernel(const cbc_tok_perline *__restrict__ pl, uint64_t n_lines, unsigned long long *__restrict__ first_bad, unsigned long long *__restrict__ n_unmapped)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_lines) return;
    const uint32_t st = pl[k].status;
    if (st >= CBC_TOK_NEEDS_HOST) atomicMin(first_bad, (unsigned long long)k);
    if (st == CBC_TOK_UNMAPPED) atomicAdd(n_unmapped, 1ull);
}

API void cbc_gpu_tokenise_free(cbc_gpu_ctx *ctx, cbc_tok_result *t)
{
    if (!t) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    free(t->summaries); free(t->rname_change); free(t->change_name_off); free(t->change_name_len);
    if (t->d_seq) (void)hipFree(t->d_seq); if (t->d_tok) (void)hipFree(t->d_tok);
    memset(t, 0, sizeof *t);
}

API int cbc_gpu_tokenise_sam(cbc_gpu_ctx *ctx, const char *sam, uint64_t len, uint64_t body_off, cbc_tok_result *out)
{
    if (!ctx || !sam || !out || body_off > len) return CBC_E_ARG;
    memset(out, 0, sizeof *out);
    if (len == 0) return CBC_OK;
    if (len > 0xffffffffull * 64) return set_err(ctx, CBC_E_ARG, "SAM text too large for one launch", hipSuccess);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const uint64_t n_tiles = (len + CBC_TOK_TILE - 1) / CBC_TOK_TILE;
    void *d_sam = NULL, *d_tc = NULL, *d_tb = NULL, *d_tmp = NULL, *d_ls = NULL, *d_pl = NULL, *d_isrec = NULL, *d_vrl = NULL, *d_vnt = NULL,
         *d_recof = NULL, *d_seqof = NULL, *d_tokof = NULL, *d_sum = NULL, *d_chg = NULL, *d_cnt = NULL;
    int rc = CBC_OK;
    uint64_t n_nl = 0, n_lines = 0, n_recs = 0, seq_bytes = 0, n_tok = 0;
    unsigned long long cnt[2] = { ~0ull, 0ull };
    cbc_tok_perline bad_pl;
#define GO(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = set_err(ctx, CBC_E_NODEV, what, e_); goto done; } } while (0)
#define GOR(call) do { rc = (call); if (rc) goto done; } while (0)
    const bool times = getenv("CBC_TOK_TIMES") != NULL;            /* diagnostic: stage wall times on stderr */
    auto now = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    double T0 = now(), T1 = T0, T2 = T0, T3 = T0;
    GO(hipMalloc(&d_sam, len + 64), "hipMalloc SAM text");
    GO(hipMemcpyAsync(d_sam, sam, len, hipMemcpyHostToDevice, ctx->stream), "H2D SAM text");
    if (times) { GO(hipStreamSynchronize(ctx->stream), "H2D SAM text"); T1 = now(); }
    GO(hipMalloc(&d_tc, n_tiles * 4 + 16), "hipMalloc tiles");
    GO(hipMalloc(&d_tb, n_tiles * 8 + 16), "hipMalloc tiles");
    GO(hipMalloc(&d_tmp, (len / 1024 + n_tiles / 1024 + 16) * 8), "hipMalloc scan scratch");   /* block totals of the largest scan: n_lines <= len */
    hipLaunchKernelGGL(cbc_tok_count_kernel, dim3((unsigned)n_tiles), dim3(64), 0, ctx->stream, (const uint8_t *)d_sam, len, (uint32_t *)d_tc);
    GO(hipGetLastError(), "launch cbc_tok_count_kernel");
    GOR(scan_u32(ctx, (const uint32_t *)d_tc, n_tiles, (uint64_t *)d_tb, (uint64_t *)d_tmp, &n_nl));
    n_lines = n_nl + ((uint8_t)sam[len - 1] != '\n' ? 1 : 0);
    GO(hipMalloc(&d_ls, (n_lines + 2) * 8), "hipMalloc line starts");
    {
        const uint64_t zero = 0;
        GO(hipMemcpyAsync(d_ls, &zero, 8, hipMemcpyHostToDevice, ctx->stream), "line start 0");
        GO(hipMemcpyAsync((uint64_t *)d_ls + n_lines, &len, 8, hipMemcpyHostToDevice, ctx->stream), "line start n");
    }
    hipLaunchKernelGGL(cbc_tok_lines_kernel, dim3((unsigned)n_tiles), dim3(64), 0, ctx->stream, (const uint8_t *)d_sam, len, (const uint64_t *)d_tb, (uint64_t *)d_ls);
    GO(hipGetLastError(), "launch cbc_tok_lines_kernel");
    GO(hipMalloc(&d_pl, n_lines * sizeof(cbc_tok_perline) + 16), "hipMalloc per-line");
    GO(hipMalloc(&d_isrec, n_lines * 4 + 16), "hipMalloc"); GO(hipMalloc(&d_vrl, n_lines * 4 + 16), "hipMalloc"); GO(hipMalloc(&d_vnt, n_lines * 4 + 16), "hipMalloc");
    GO(hipMalloc(&d_recof, n_lines * 8 + 16), "hipMalloc"); GO(hipMalloc(&d_seqof, n_lines * 8 + 16), "hipMalloc"); GO(hipMalloc(&d_tokof, n_lines * 8 + 16), "hipMalloc");
    GO(hipMalloc(&d_cnt, 16), "hipMalloc counters");
    GO(hipMemcpyAsync(d_cnt, cnt, 16, hipMemcpyHostToDevice, ctx->stream), "H2D counters");
    hipLaunchKernelGGL(cbc_tok_split_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t *)d_sam,
                       (const uint64_t *)d_ls, n_lines, body_off, (cbc_tok_perline *)d_pl);
    GO(hipGetLastError(), "launch cbc_tok_split_kernel");
    hipLaunchKernelGGL(cbc_tok_parse_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t *)d_sam,
                       (const uint64_t *)d_ls, n_lines, body_off, (cbc_tok_perline *)d_pl, (uint32_t *)d_isrec, (uint32_t *)d_vrl, (uint32_t *)d_vnt);
    GO(hipGetLastError(), "launch cbc_tok_parse_kernel");
    hipLaunchKernelGGL(cbc_tok_status_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, ctx->stream, (const cbc_tok_perline *)d_pl,
                       n_lines, (unsigned long long *)d_cnt, (unsigned long long *)d_cnt + 1);
    GO(hipGetLastError(), "launch cbc_tok_status_kernel");
    GOR(scan_u32(ctx, (const uint32_t *)d_isrec, n_lines, (uint64_t *)d_recof, (uint64_t *)d_tmp, &n_recs));
    GOR(scan_u32(ctx, (const uint32_t *)d_vrl, n_lines, (uint64_t *)d_seqof, (uint64_t *)d_tmp, &seq_bytes));
    GOR(scan_u32(ctx, (const uint32_t *)d_vnt, n_lines, (uint64_t *)d_tokof, (uint64_t *)d_tmp, &n_tok));
    GO(hipMemcpyAsync(cnt, d_cnt, 16, hipMemcpyDeviceToHost, ctx->stream), "D2H counters");
    GO(hipStreamSynchronize(ctx->stream), "tokenise pass 1");
    T2 = now();
    out->n_lines = n_lines; out->n_recs = n_recs; out->n_unmapped = cnt[1]; out->seq_bytes = seq_bytes; out->n_tok = n_tok;
    if (cnt[0] != ~0ull) {                                    /* a line the device path does not take: say which and why */
        GO(hipMemcpy(&bad_pl, (cbc_tok_perline *)d_pl + cnt[0], sizeof bad_pl, hipMemcpyDeviceToHost), "D2H status");
        out->status = bad_pl.status; out->bad_line = cnt[0];
        goto done;
    }
    GO(hipMalloc((void **)&out->d_seq, seq_bytes + 16), "hipMalloc seq");
    GO(hipMalloc((void **)&out->d_tok, (n_tok + 4) * 4), "hipMalloc tok");
    GO(hipMemsetAsync(out->d_seq + seq_bytes, 0, 16, ctx->stream), "memset seq pad");
    GO(hipMalloc(&d_sum, (n_recs + 1) * sizeof(cbc_tok_summary)), "hipMalloc summaries");
    GO(hipMalloc(&d_chg, n_recs + 16), "hipMalloc change flags");
    if (n_recs) {
        hipLaunchKernelGGL(cbc_tok_emit_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t *)d_sam,
                           (const uint64_t *)d_ls, n_lines, (const cbc_tok_perline *)d_pl, (const uint64_t *)d_recof, (const uint64_t *)d_tokof,
                           out->d_tok, (cbc_tok_summary *)d_sum);
        GO(hipGetLastError(), "launch cbc_tok_emit_kernel");
        hipLaunchKernelGGL(cbc_tok_seq_kernel, dim3((unsigned)((n_recs + 15) / 16)), dim3(64), 0, ctx->stream, (const uint8_t *)d_sam,
                           (const cbc_tok_perline *)d_pl, (const cbc_tok_summary *)d_sum, n_recs, (const uint64_t *)d_seqof, out->d_seq);
        GO(hipGetLastError(), "launch cbc_tok_seq_kernel");
        hipLaunchKernelGGL(cbc_tok_names_kernel, dim3((unsigned)((n_recs + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t *)d_sam,
                           (const cbc_tok_perline *)d_pl, (const cbc_tok_summary *)d_sum, n_recs, (uint8_t *)d_chg);
        GO(hipGetLastError(), "launch cbc_tok_names_kernel");
    }
    out->summaries = (cbc_tok_record_summary *)malloc((size_t)(n_recs + 1) * sizeof(cbc_tok_record_summary));
    out->rname_change = (uint8_t *)malloc((size_t)n_recs + 1);
    if (!out->summaries || !out->rname_change) { rc = CBC_E_NOMEM; goto done; }
    GO(hipMemcpyAsync(out->summaries, d_sum, n_recs * sizeof(cbc_tok_summary), hipMemcpyDeviceToHost, ctx->stream), "D2H summaries");
    GO(hipMemcpyAsync(out->rname_change, d_chg, n_recs, hipMemcpyDeviceToHost, ctx->stream), "D2H change flags");
    GO(hipStreamSynchronize(ctx->stream), "tokenise pass 2");
    T3 = now();
    if (times) fprintf(stderr, "tokenise: H2D of %.0f MB %.3f s, line index + parse + scans %.3f s, emit + bases + D2H of the summaries %.3f s\n",
                       (double)len / 1e6, T1 - T0, T2 - T1, T3 - T2);
    {   /* where each new RNAME sits in the text: one small copy per contig change */
        uint64_t nc = 0;
        for (uint64_t r = 0; r < n_recs; r++) nc += out->rname_change[r] != 0;
        out->change_name_off = (uint64_t *)malloc((size_t)(nc + 1) * 8); out->change_name_len = (uint32_t *)malloc((size_t)(nc + 1) * 4);
        if (!out->change_name_off || !out->change_name_len) { rc = CBC_E_NOMEM; goto done; }
        uint64_t k = 0;
        for (uint64_t r = 0; r < n_recs; r++) if (out->rname_change[r]) {
            cbc_tok_perline one;
            GO(hipMemcpy(&one, (cbc_tok_perline *)d_pl + out->summaries[r].line, sizeof one, hipMemcpyDeviceToHost), "D2H contig name");
            out->change_name_off[k] = one.rname; out->change_name_len[k] = one.rname_len; k++;
        }
        out->n_changes = nc;
    }
done:
#undef GO
#undef GOR
    if (d_sam) (void)hipFree(d_sam); if (d_tc) (void)hipFree(d_tc); if (d_tb) (void)hipFree(d_tb); if (d_tmp) (void)hipFree(d_tmp);
    if (d_ls) (void)hipFree(d_ls); if (d_pl) (void)hipFree(d_pl); if (d_isrec) (void)hipFree(d_isrec); if (d_vrl) (void)hipFree(d_vrl);
    if (d_vnt) (void)hipFree(d_vnt); if (d_recof) (void)hipFree(d_recof); if (d_seqof) (void)hipFree(d_seqof); if (d_tokof) (void)hipFree(d_tokof);
    if (d_sum) (void)hipFree(d_sum); if (d_chg) (void)hipFree(d_chg); if (d_cnt) (void)hipFree(d_cnt);
    if (rc) cbc_gpu_tokenise_free(ctx, out);
    return rc;
}

API int cbc_gpu_tokenise_fetch(cbc_gpu_ctx *ctx, const cbc_tok_result *t, uint8_t *seq, uint32_t *tok)
{
    if (!ctx || !t || !seq || !tok) return CBC_E_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (t->d_seq) HIPCHK(hipMemcpy(seq, t->d_seq, t->seq_bytes + 8, hipMemcpyDeviceToHost), "D2H seq");
    if (t->d_tok && t->n_tok) HIPCHK(hipMemcpy(tok, t->d_tok, t->n_tok * 4, hipMemcpyDeviceToHost), "D2H tok");
    return CBC_OK;
}

API int cbc_gpu_encode_blocks_tokenised(cbc_gpu_ctx *ctx, const cbc_tok_result *t, const cbc_host_batch *hb,
                                        uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, cbc_block_result *results)
{
    if (!t || !t->d_seq || !t->d_tok || !hb) return CBC_E_ARG;
    if (hb->seq_bytes != t->seq_bytes + 8 || hb->n_tok != t->n_tok) return set_err(ctx, CBC_E_ARG, "batch and tokeniser result disagree", hipSuccess);
    if (!t->summaries || hb->n_recs != t->n_recs) return set_err(ctx, CBC_E_ARG, "batch and tokeniser result disagree", hipSuccess);
    return encode_blocks_impl(ctx, hb, NULL, NULL, 0, out, out_cap, out_offsets, results, t->d_seq, t->d_tok, t->summaries);
}
