/* vidmem.h - C ABI of libvidmem.so: the MI355X (gfx950) replacement for the frame-embedding +
 * cosine/top-k hot path of VidGraph (RaphaelHaddad/Real-Time-Brain-Inspired-Video-Memory).
 *
 * The reference is pure Python and has no FFI; each entry point below names the reference call site whose
 * arithmetic it replaces (paths relative to the reference repository root).  INTEGRATION.md shows the ctypes
 * stubs a maintainer adds at those call sites.
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, sizes.  No torch / HIP types (streams travel as void*).
 *   - every data pointer is a DEVICE pointer unless the parameter name ends in _host.
 *   - return 0 = VM_OK, negative = vm_status; text via vm_last_error(ctx).
 *   - the caller owns every buffer; the library allocates only inside vm_*_create (freed by *_destroy).
 *   - vm_preprocess / vm_encode / vm_memory_append / vm_topk_* enqueue work on the given stream and return:
 *     no hidden synchronisation, no allocation, no host threads -> capturable into a hipGraph.
 *   - calls on one handle must be stream-ordered by the caller.
 */
#ifndef VIDMEM_H
#define VIDMEM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vm_ctx vm_ctx;
typedef struct vm_encoder vm_encoder;
typedef struct vm_memory vm_memory;

enum vm_status {
    VM_OK = 0,
    VM_ERR_INVALID = -1,     /* bad argument (shape / dtype / null) - the Python adapters raise on this   */
    VM_ERR_HIP = -2,         /* a HIP runtime call failed                                                */
    VM_ERR_NOMEM = -3,       /* device allocation failed or caller workspace too small                   */
    VM_ERR_UNSUPPORTED = -4, /* shape outside what the kernels are built for                             */
    VM_ERR_NO_DEVICE = -5    /* no gfx950 device visible                                                 */
};
enum vm_dtype { VM_F16 = 0, VM_BF16 = 1, VM_F32 = 2 /* vm_cosine_exact operands only */ };
enum vm_act { VM_ACT_GELU = 0, VM_ACT_QUICK_GELU = 1 };
enum vm_layout { VM_LAYOUT_CHW = 0, VM_LAYOUT_PATCHES = 1 };
/* Neo4j's vector.similarity.cosine (retriever_hybrid.py:296) is third-party and unpinned: the score mapping is
 * an explicit parameter instead of a guess.  RAW = cosine, UNIT_INTERVAL = (1 + cosine) / 2. */
enum vm_score_mode { VM_SCORE_RAW = 0, VM_SCORE_UNIT_INTERVAL = 1 };
enum vm_topk_flag { VM_FLAG_CERTIFIED = 0, VM_FLAG_GAP = 1, VM_FLAG_OVERFLOW = 3 };   /* out_query_flags values */

/* ---- context ------------------------------------------------------------------------------------------ */
int vm_init(int device, vm_ctx **out);
void vm_destroy(vm_ctx *ctx);
const char *vm_last_error(vm_ctx *ctx);
/* ABI version of this header (bumped on any signature change; 3: vm_encode_micro_batch, VM_PROF_GEMM_CLS - the arrays
 * vm_profile_read fills grew to 14 entries; 4: vm_encoder_set_option / vm_encoder_get_option replace the environment
 * variables an encoder used to read when it was created, vm_probe_mfma, vm_topk_select). */
int vm_abi_version(void);

/* ---- frame preprocessing ------------------------------------------------------------------------------
 * Replaces the CPU frame handling between cv2.VideoCapture.read and the VLM request:
 * src/pipeline/vlm_extractor.py:110-128 (BGR uint8 HWC frames; JPEG/base64 is NOT reproduced).
 * frames: uint8 [B,H,W,3] BGR.  Bilinear resize (half-pixel centres, edge clamp, no antialias) to
 * out_S x out_S, BGR->RGB, x*(1/(255*std)) - mean/std, cast to `dtype`.
 * layout CHW    : out [B,3,S,S]
 * layout PATCHES: out [B,(S/patch)^2,k_pad], k index = c*patch*patch + py*patch + px, zero-padded to k_pad
 *                 (the patch-embed GEMM's A operand; k_pad = vm_encoder_patch_k(enc)). */
int vm_preprocess(vm_ctx *ctx, const uint8_t *frames_hwc_bgr, int B, int H, int W, const float mean_host[3],
                  const float std_host[3], int out_S, int dtype, int layout, int patch, int k_pad, void *out,
                  void *stream);

/* ---- vision encoder -----------------------------------------------------------------------------------
 * Replaces the remote model behind VLMExtractor._call_vlm_api (src/pipeline/vlm_extractor.py:130-185) and
 * behind OpenAIEmbeddings.aembed_query (src/components/neo4j_handler.py:27-31,333;
 * src/components/pre_llm_injector.py:207-221): frames -> one embedding vector each. */
typedef struct vm_encoder_desc {
    int image;      /* input side, 224 or 336                          */
    int patch;      /* 16 or 14                                        */
    int hidden;     /* 768 / 1024                                      */
    int layers;     /* 12 / 24                                         */
    int heads;      /* 12 / 16 (head dim must be 64)                   */
    int mlp;        /* 3072 / 4096                                     */
    int act;        /* vm_act                                          */
    int pre_ln;     /* 1: LayerNorm after the embeddings (CLIP)        */
    int patch_bias; /* 1: patch-embed conv has a bias (ViT)            */
    int proj_dim;   /* 0: none, else output projection rows            */
    int dtype;      /* vm_dtype of GEMM operands and of the output     */
    float ln_eps;
} vm_encoder_desc;

/* weights_host: array of DEVICE pointers, in this order (n = 9 + 12*layers):
 *   0 patch_w [hidden, k_pad] dtype (k = c*p*p+py*p+px, zero-padded)   1 patch_b [hidden] f32
 *   2 cls [hidden] f32        3 pos [tokens, hidden] f32
 *   4 pre_ln_g  5 pre_ln_b  (f32 [hidden]; ignored unless pre_ln)
 *   6 ln_g  7 ln_b (final LayerNorm, f32)      8 proj_w [proj_dim, hidden] dtype (ignored if proj_dim == 0)
 *   then per layer l, base = 9 + 12*l:
 *   +0 ln1_g +1 ln1_b (f32)  +2 qkv_w [3*hidden, hidden] dtype  +3 qkv_b [3*hidden] f32
 *   +4 proj_w [hidden, hidden] dtype  +5 proj_b f32  +6 ln2_g +7 ln2_b (f32)
 *   +8 fc1_w [mlp, hidden] dtype  +9 fc1_b f32  +10 fc2_w [hidden, mlp] dtype  +11 fc2_b f32
 * All matrices are row-major [out_features][in_features] (torch.nn.Linear layout).  The library copies them
 * (device-to-device) during create; the caller may free its copies afterwards. */
int vm_encoder_create(vm_ctx *ctx, const vm_encoder_desc *desc, const void *const *weights_host, int n_weights,
                      vm_encoder **out);
void vm_encoder_destroy(vm_encoder *enc);
/* Per-encoder options (ABI 4).  The release library reads NO environment variable: what a deployment may choose is
 * set here, on the handle, between calls (not while a vm_encode of this encoder is being captured or is in flight).
 * Every value of every option produces the same embeddings bit for bit (tests/test_encoder_gpu.py).
 *   VM_ENC_OPT_SCHEDULE     vm_encoder_schedule.  TWO_STREAMS: consecutive micro-batch passes of one vm_encode call
 *                           alternate between two internal streams (forked from / joined to the caller's stream inside
 *                           the call, still capturable; one workspace per stream), so that one pass's bandwidth-bound
 *                           LayerNorms run beside the other pass's matrix-bound GEMMs.  ONE_STREAM: every kernel on the
 *                           caller's stream.  AUTO (default): TWO_STREAMS for calls of two or more passes, except
 *                           while vm_profile_enable(ctx, > 0) is in force - with two streams a kernel's event
 *                           duration includes its wait for the other stream's kernels, so timing runs take one stream.
 *   VM_ENC_OPT_MICRO_BATCH  frames per pass; 0 (default) = chosen by the library (see vm_encode_micro_batch).
 *   VM_ENC_OPT_LAST_LAYER   which rows the LAST layer computes behind its keys and values: 3 (default) = the CLS rows
 *                           only (the embedding is pooled from them), 1 = CLS rows behind the attention, 0 = every row. */
enum vm_encoder_option { VM_ENC_OPT_SCHEDULE = 0, VM_ENC_OPT_MICRO_BATCH = 1, VM_ENC_OPT_LAST_LAYER = 2 };
enum vm_encoder_schedule { VM_SCHED_AUTO = 0, VM_SCHED_ONE_STREAM = 1, VM_SCHED_TWO_STREAMS = 2 };
int vm_encoder_set_option(vm_encoder *enc, int option, int value);
int vm_encoder_get_option(const vm_encoder *enc, int option);   /* negative = vm_status */
int vm_encoder_tokens(const vm_encoder *enc);    /* patches + 1                        */
int vm_encoder_patch_k(const vm_encoder *enc);   /* 3*patch*patch rounded up to 64     */
int vm_encoder_out_dim(const vm_encoder *enc);   /* proj_dim ? proj_dim : hidden       */
size_t vm_encode_workspace_bytes(const vm_encoder *enc, int B);
/* Frames vm_encode runs per pass for a call with B frames (it walks B in micro-batches: a whole number of GEMM tile
 * rounds, and - for sequences that take one attention workgroup per (frame, head) - of attention rounds).
 * vm_encode_workspace_bytes returns twice the single-pass size for calls of more than one pass unless the schedule is
 * VM_SCHED_ONE_STREAM (one workspace per internal stream); a call that is handed less runs on one stream. */
int vm_encode_micro_batch(const vm_encoder *enc, int B);
/* patches: [B, tokens-1, patch_k] dtype (vm_preprocess, PATCHES layout).  out_emb: [B, out_dim] dtype.
 * l2_normalise: divide each embedding by its L2 norm (fp32) before the cast. */
int vm_encode(vm_encoder *enc, const void *patches, int B, void *out_emb, int l2_normalise, void *workspace,
              size_t workspace_bytes, void *stream);

/* ---- embedding memory ---------------------------------------------------------------------------------
 * Replaces the Chunk.embedding store: append = src/components/neo4j_handler.py:229-242
 * (MERGE ... SET c.embedding), bulk read-back = src/components/pre_llm_injector.py:390-412.
 * Rows are kept resident in HBM as [capacity, D] dtype, plus per row an exact fp64 norm and an fp32 reciprocal
 * norm.  Row id = append order (0,1,2,...).  ring=1: after `capacity` rows the oldest are overwritten; ids keep
 * counting, only the newest `capacity` ids are searchable. */
int vm_memory_create(vm_ctx *ctx, int64_t capacity_rows, int D, int dtype, int ring, vm_memory **out);
void vm_memory_destroy(vm_memory *mem);
/* rows: [B, D] dtype.  *out_first_row_host (optional) receives the id of the first appended row. */
int vm_memory_append(vm_memory *mem, const void *rows, int B, int64_t *out_first_row_host, void *stream);
int64_t vm_memory_size(const vm_memory *mem);  /* rows appended so far (host mirror)                      */
int64_t vm_memory_capacity(const vm_memory *mem);
int vm_memory_dim(const vm_memory *mem);
int vm_memory_reset(vm_memory *mem, void *stream);
/* Re-read the DEVICE row counter into the host mirror and return it (negative = vm_status).  Needed after hipGraph
 * replays of vm_memory_append (they advance only the device counter) and after a graph CAPTURE of it (which advanced
 * only the host mirror).  Synchronises `stream`; not capturable.  The reference has no counterpart (its store is a
 * database, src/components/neo4j_handler.py:229-242); this is the price of the capturable append. */
int64_t vm_memory_sync(vm_memory *mem, void *stream);
const void *vm_memory_rows(const vm_memory *mem); /* device pointer to the [capacity, D] row store          */

/* ---- cosine top-k over the memory ---------------------------------------------------------------------
 * Replaces PreLLMInjector._calculate_batch_similarities + _cosine_similarity
 * (src/components/pre_llm_injector.py:346-388) and the Cypher scan of HybridRetriever._vector_search_chunks
 * (src/pipeline/retriever_hybrid.py:293-306).
 *
 * queries [Q, D] dtype.  For every query: score each stored row with the reference cosine, order by
 * (score descending, row id ascending) - Python's stable sort over memory order - and keep the first k.
 * out_scores [Q,k] are the reference's fp64 values BIT FOR BIT (sequential fp64 sums over the stored 16-bit
 * values); out_rows [Q,k] int64 row ids, -1 padded (scores 0.0 padded).
 *   global row id written = row_id * row_stride + row_offset   (row-sharded memory; use 1, 0 on one GPU)
 *   use_min_score: keep only score > min_score (after the score_mode mapping).
 * Two-stage: an fp32 MFMA scan keeps k+slack candidates per query, which are re-scored exactly.  A query whose
 * result cannot be PROVEN equal to the exhaustive answer (fp32 error bound vs the gap to the best rejected
 * row) is counted in *out_uncertified (device int32, accumulates, may be NULL) and marked in
 * out_query_flags[q] != 0 (device int32 [Q], rewritten for every query on every call, may be NULL; the value says
 * why, vm_topk_flag: VM_FLAG_GAP = the exact k-th score does not clear the best rejected fp32 score by the error
 * bound 2 (D + 8) 2^-24 - near-ties between rank k and rank k + slack, e.g. more than `slack` exact duplicates;
 * VM_FLAG_OVERFLOW = more candidates at or above a query's cut than its buffer holds).  Pass the flags
 * to vm_topk_redo_flagged on the same stream: the reference always returns the exhaustive answer
 * (src/components/pre_llm_injector.py:356-370), so the pair {vm_topk_cosine, vm_topk_redo_flagged} is the drop-in. */
size_t vm_topk_workspace_bytes(const vm_memory *mem, int Q, int k);
int vm_topk_cosine(vm_memory *mem, const void *queries, int Q, int k, int use_min_score, double min_score,
                   int score_mode, int64_t row_stride, int64_t row_offset, double *out_scores,
                   int64_t *out_rows, int32_t *out_uncertified, int32_t *out_query_flags, void *workspace,
                   size_t workspace_bytes, void *stream);
/* Exhaustive fp64 redo of the flagged queries only (query_flags: device int32 [Q], as written by vm_topk_cosine):
 * their rows of out_scores / out_rows are overwritten with the exhaustive answer under the same contract; the
 * other queries are left untouched.  Row count and flags are read on the device: no host read-back, no
 * allocation, capturable into a hipGraph, and two near-empty launches when nothing is flagged.  k <= 64.
 * Workspace: vm_topk_redo_workspace_bytes (slice winners, 16 bytes x blocks x Q x k; a few MB). */
size_t vm_topk_redo_workspace_bytes(const vm_memory *mem, int Q, int k);
int vm_topk_redo_flagged(vm_memory *mem, const void *queries, int Q, int k, int use_min_score, double min_score,
                         int score_mode, int64_t row_stride, int64_t row_offset, const int32_t *query_flags,
                         double *out_scores, int64_t *out_rows, void *workspace, size_t workspace_bytes,
                         void *stream);
/* Exhaustive fp64 version of the same contract for ALL queries (every pair scored exactly; slow, always exact;
 * workspace Q x rows x 8 bytes; sizes its grid from the host row count, so not for graph replay). */
size_t vm_topk_exact_workspace_bytes(const vm_memory *mem, int Q, int k);
int vm_topk_cosine_exact(vm_memory *mem, const void *queries, int Q, int k, int use_min_score,
                         double min_score, int score_mode, int64_t row_stride, int64_t row_offset,
                         double *out_scores, int64_t *out_rows, void *workspace, size_t workspace_bytes,
                         void *stream);
/* All-pairs exact cosine, out [Q, S] fp64: the post-compression filter of
 * src/pipeline/retriever_hybrid.py:494-504 (query vs segment embeddings) and a checker for the scan.
 * rows [S, D] dtype need not live in a vm_memory.  dtype VM_F32 takes fp32 operands: an embedder that returns fp32
 * values (every OpenAI-compatible server does) is then scored on its UN-rounded vectors, as the reference scores them
 * (:497), so `>= compression_threshold` decisions cannot flip on a 16-bit rounding. */
int vm_cosine_exact(vm_ctx *ctx, const void *queries, int Q, const void *rows, int64_t S, int D, int dtype,
                    double *out, void *stream);
/* The k best columns of every row of an all-pairs score matrix (vm_cosine_exact's output), by the order relation of
 * vm_topk_cosine: (score descending, column ascending).  scores [Q,S] fp64; col_limit (device int64 [Q], may be NULL):
 * query q ranks only columns < col_limit[q].  out_rows = row_base + column, -1 / 0.0 padded.  With vm_cosine_exact
 * and vm_topk_merge this ranks the frames of a look-ahead group against the group's own EARLIER chunks, rows that are
 * about to be appended: what the reference's chunk-by-chunk loop would have stored by the time each chunk is searched
 * (src/pipeline/vlm_extractor.py:44-74; the same stable sort as src/components/pre_llm_injector.py:369). */
int vm_topk_select(vm_ctx *ctx, const double *scores, int Q, int64_t S, const int64_t *col_limit, int k,
                   int64_t row_base, double *out_scores, int64_t *out_rows, void *stream);
/* Merge `parts` per-shard results (each [Q,k], sorted as above, -1 padded) into the global top-k:
 * the step after the RCCL all-gather (and the cross-query max-merge input of pre_llm_injector.py:238-249).
 * scores [parts,Q,k] fp64, rows [parts,Q,k] int64. */
int vm_topk_merge(vm_ctx *ctx, const double *scores, const int64_t *rows, int parts, int Q, int k,
                  double *out_scores, int64_t *out_rows, void *stream);

/* ---- measurement ---------------------------------------------------------------------------------------
 * Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg; the
 * reference's counterpart is MetricsTracker.record_timing, src/core/metrics.py:18-24).  While enabled, every
 * kernel launch of this context is bracketed by two events from a preallocated pool (no sync, no allocation);
 * vm_profile_read synchronises the device, folds the pool into per-category totals and clears it.
 * Not capturable into a hipGraph while enabled. */
enum vm_prof_cat {
    VM_PROF_PREPROCESS = 0, VM_PROF_GEMM_PATCH, VM_PROF_GEMM_QKV, VM_PROF_GEMM_ACT, VM_PROF_GEMM_RESID,
    VM_PROF_ATTENTION, VM_PROF_LAYERNORM, VM_PROF_POOL, VM_PROF_APPEND, VM_PROF_TOPK_SCAN, VM_PROF_TOPK_FINALIZE,
    VM_PROF_TOPK_EXACT, VM_PROF_TOPK_MERGE,
    VM_PROF_GEMM_CLS,   /* the last encoder layer's GEMMs over the CLS rows only (one row per frame) */
    VM_PROF_NCAT
};
int vm_profile_enable(vm_ctx *ctx, int max_events);   /* 0 disables and frees the pool */
int vm_profile_read(vm_ctx *ctx, double *total_ms_host /*[VM_PROF_NCAT]*/, int64_t *launches_host /*[VM_PROF_NCAT]*/);
/* Restrict event recording to the categories whose bit (1u << vm_prof_cat) is set (default: all).  Two event
 * records cost several microseconds per launch on this stack, so a timed run enables only the kernel it reports. */
int vm_profile_mask(vm_ctx *ctx, uint32_t category_mask);

/* What this GPU sustains on 16-bit MFMA work, measured here and now (bench.py's `mfma_ceiling`).  The
 * chip lowers its clock under matrix load, so the 2.5 PFLOP/s of the data sheet is not what any kernel can reach on
 * random operands; these loops put a number on what can.  Runs back-to-back launches of a synthetic loop for about
 * `seconds` on `stream`, synchronises, and writes the mean rate in TFLOP/s to *tflops_host.
 *   variant 0: register-only v_mfma_f32_16x16x32_f16 loop with the operand pattern of a 128 x 64 wave tile, two waves
 *              per SIMD on every CU - no LDS, no memory;
 *   variant 1: + the fragment reads of the 256 x 256 GEMM tile (12 conflict-free ds_read_b128 per 32 MFMAs);
 *   variant 2: + its staging (4 KiB of LDS-DMA per wave and 32 MFMAs out of an L2-resident buffer, counted wait):
 *              the GEMM's K loop with no epilogue, barrier, tile boundary, miss or store.
 * zero_operands = 1 fills the operands with zeros (the same cycles at the clock an idle data path allows).
 * No reference counterpart (the reference has no kernels). */
int vm_probe_mfma(vm_ctx *ctx, int variant, int zero_operands, double seconds, double *tflops_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDMEM_H */
