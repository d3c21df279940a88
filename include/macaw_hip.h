/*
 * macaw_hip.h — C ABI of the MI355X-native (gfx950) kernels behind Macaw-LLM's
 * multimodal forward/backward hot path.
 *
 * The reference (lyuchenyang/Macaw-LLM) has no FFI of its own: the hot path is
 * eager PyTorch inside modeling.py.  Each entry point below therefore cites the
 * reference expression (modeling.py:line, or the un-vendored torch/transformers
 * module it calls) whose arithmetic it replaces.  The Python mirror of the
 * reference surface (macaw_llm_amd/modeling.py) calls these through ctypes.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     unless noted; nothing is allocated or retained; asynchronous on `stream`
 *     (a hipStream_t passed as void*).
 *   - return value: 0 on success, <0 on error (MK_ERR_*). Never aborts.
 *   - dtype codes: 0 = f32, 1 = bf16, 2 = f16.  Every kernel exists for all three (the reference's
 *     scripts run fp16: train.sh:36, llm_trainer.py:411-412) except where a comment says otherwise
 *     (fp8 quantisation and the host-input pipeline take bf16 / f32); accumulation is always f32.
 *   - matrices are row-major with an explicit leading dimension (elements).
 */
#ifndef MACAW_HIP_H
#define MACAW_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MK_OK 0
#define MK_ERR_BAD_ARG (-1)
#define MK_ERR_UNSUPPORTED (-2)
#define MK_ERR_LAUNCH (-3)

#define MK_F32 0
#define MK_BF16 1
#define MK_F16 2
#define MK_FP8 3 /* OCP e4m3fn bytes: mk_gemm operands / mk_fp8_quantize output only */

/* library identification: returns MK_ABI_VERSION */
#define MK_ABI_VERSION 6
int mk_abi_version(void);

/* ------------------------------------------------------------------ GEMM --
 * C[z][M,N] = act(alpha * opA(A[z]) * opB(B[z])^T + bias) + R[z] (+ C[z] if accumulate)
 *
 * Logical operands: A is M x K, B is N x K ("NT": both reduce over their 2nd
 * index).  a_red_major / b_red_major = 1 means the operand is STORED with the
 * reduction index as the row index (i.e. A stored as [K][M], B as [K][N]), so
 * the four BLAS layouts are covered without separate transposes:
 *   nn.Linear forward  y = x W^T        : A=x[M,K]        B=W[N,K]       (0,0)
 *   grad input         dx = dy W        : A=dy[M,N]       B=W[N,K] as [red=N][out=K] (0,1)
 *   grad weight        dW = dy^T x      : A=dy[M,N] as [red=M][out=N], B=x (1,1)
 * Replaces nn.Linear (modeling.py:134-140,159-162,530,912-917), torch.matmul
 * in LlamaAttention (modeling.py:197,215), the packed in-proj / out-proj of
 * nn.MultiheadAttention (modeling.py:882-910) and the HF CLIP / Whisper
 * projections (modeling.py:1073,1082,1092).
 * Batch: grid z = z1*nb2 + z2; pointer offset = z1*s?1 + z2*s?2 (elements).
 */
typedef struct mk_gemm_desc {
  const void* A;
  const void* B;
  void* C;
  const void* R;    /* optional residual added after activation, same dtype as C */
  const void* bias; /* optional, f32 or same dtype as C (see bias_dtype) */
  int32_t M, N, K;
  int64_t lda, ldb, ldc, ldr;
  int32_t a_red_major, b_red_major;
  int32_t nb1, nb2; /* batch = nb1*nb2 (>=1 each) */
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
  float alpha;
  int32_t bias_mode;  /* 0 none, 1 per output column (n), 2 per output row (m) */
  int32_t act;        /* 0 none, 1 gelu(erf), 2 quick_gelu (x*sigmoid(1.702x)) */
  int32_t accumulate; /* 1: C += result */
  int32_t dtype;      /* MK_F32 or MK_BF16: type of A,B,C,R,bias.  MK_FP8: A and B are OCP e4m3 bytes,
                         both K-major (a_red_major = b_red_major = 0), K % 128 == 0, lda / ldb in
                         elements (= bytes) and multiples of 16; C, R, bias are bf16 */
  void* ws;           /* optional device scratch for the stream-K tail (fp32 partial tiles +
                         arrival counters); NULL disables it. Must not be shared by GEMMs that
                         run concurrently on different streams.  Its first 4096 bytes (the
                         counters) must be ZERO before the first use; every launch leaves
                         them zero again (the last arriver of a tile resets its counter). */
  int64_t ws_bytes;
  const float* scale_a; /* optional DEVICE scalars multiplied into alpha in the epilogue (the     */
  const float* scale_b; /* per-tensor de-quantisation scales of fp8 operands; no host sync)       */
  int32_t flags;        /* MK_GEMM_*_KPAD_ZERO: that K-major operand's rows are readable and ZERO from
                           column K up to the next multiple of 64 (a pitched buffer whose pad columns
                           are kept zero, e.g. d(logits) [tokens, 32064] for V = 32007), so a K that
                           is not a multiple of 64 can still take the MFMA tile kernels (the other
                           operand must be reduction-major or padded the same way) */
} mk_gemm_desc;
#define MK_GEMM_SCALE_VEC 4   /* scale_a / scale_b are VECTORS: scale_a[M] per row of A (= output row),
                                 scale_b[N] per row of B (= output column): C = alpha-free
                                 act(acc * scale_a[m] * scale_b[n] + bias) ... (per-row / per-channel
                                 fp8 de-quantisation, mk_fp8_quantize_rows / _cols_t); MK_FP8_E4M3
                                 operands only, MK_ERR_UNSUPPORTED otherwise */
#define MK_GEMM_A_KPAD_ZERO 1
#define MK_GEMM_B_KPAD_ZERO 2
int mk_gemm(const mk_gemm_desc* d, void* stream);
/* Tuning / A-B hook: force the bf16 kernel configuration of the following mk_gemm calls
 * (11 = 256x256 v7 with eight waves, 15 = 256x256 v9: four waves, one per SIMD, hand-placed inline-asm K loop (whole tiles
 * only), 14 = its hipcc-scheduled predecessor v8 (experiment builds only), 5 = 128x128 v2, 7 = v2 BK32, 0 = generic;
 * -1 = automatic).  A forced configuration is still replaced where it is not legal for the problem.
 * Replaces cuBLAS behind nn.Linear of /root/reference/modeling.py:134-140,159-162,597. */
int mk_gemm_set_cfg(int cfg);
/* 1 if kernel configuration `cfg` is compiled into this library (14 only with MK_EXPERIMENTS=1 at build time). */
int mk_gemm_has_cfg(int cfg);
/* How many CUs the tile kernels may plan for (0 = all of the device, the default).  The 256x256 kernels hold one
 * workgroup per CU with all of its LDS, so a CU occupied by another resident kernel -- an RCCL channel of the
 * gradient reduce-scatter running beside the backward (train.sh:14, configs/deepspeed_config.json:22-41) -- is
 * lost to them: rounds of exactly 256 tiles then take two passes.  The step runtime sets this to
 * (CUs - collective channels) while collectives overlap the backward; round sizes, the spatial tail and the
 * kernel choice follow.  Returns the previous value. */
int mk_gemm_set_cus(int n_cus);
/* Optional live timing of every mk_gemm launch with HIP events on the launch stream
 * (bench.py roofline): begin, run, then end() synchronises and returns the sums. */
int mk_prof_begin(void);
int mk_prof_end(double* total_ms, double* total_flops, int64_t* launches);
/* the same sums for one launch kind since mk_prof_begin (0 = mk_gemm, 1 = mk_flash_attn_fwd,
 * 2 = mk_flash_attn_bwd; attention FLOPs are algorithmic: causal = lower triangle); call before
 * mk_prof_end */
int mk_prof_sum(int kind, double* total_ms, double* total_flops, int64_t* launches);
/* per-shape CSV breakdown (host path) of the launches since mk_prof_begin; call before end */
int mk_prof_report(const char* path);

/* 2-D (batched) transpose out[z][c][r] = in[z][r][c]; elem_size 2 or 4. */
int mk_transpose(const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in,
                 int64_t ld_out, int32_t batch, int64_t s_in, int64_t s_out, int32_t elem_size,
                 void* stream);

/* ------------------------------------------------------------ norms ------
 * RMSNorm (modeling.py:311-319): y = w * cast(x * rsqrt(mean(x^2, fp32) + eps)).
 * Optional fused residual: h = x + res is written to h_out and normalised
 * (LlamaDecoderLayer residual adds, modeling.py:283,289). rstd[rows] is f32.
 */
int mk_rmsnorm_fwd(const void* x, const void* res, const void* w, void* h_out, void* y,
                   float* rstd, int32_t rows, int32_t cols, float eps, int32_t dtype, void* stream);
/* dx = dres_in + rmsnorm_bwd(dy; h, rstd, w); dw_partial[nblk][cols] f32 partial
 * column sums (reduced by mk_colsum_partials). dres_in may be NULL. */
int mk_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd,
                   const void* dres_in, void* dx, float* dw_partial, int32_t nblk, int32_t rows,
                   int32_t cols, int32_t dtype, void* stream);
/* LayerNorm with bias (torch nn.LayerNorm inside HF CLIP / Whisper encoder layers). */
int mk_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean,
                     float* rstd, int32_t rows, int32_t cols, float eps, int32_t dtype,
                     void* stream);
int mk_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean,
                     const float* rstd, const void* dres_in, void* dx, float* dw_partial,
                     float* db_partial, int32_t nblk, int32_t rows, int32_t cols, int32_t dtype,
                     void* stream);
/* out[c] (+)= sum_b partial[b][c]; out dtype = `dtype`. */
int mk_colsum_partials(const float* partial, void* out, int32_t nblk, int32_t cols,
                       int32_t accumulate, int32_t dtype, void* stream);
/* out[c] (+)= sum_r x[r][c] (bias gradients). ws: f32 [nblk*cols] scratch. */
int mk_colsum(const void* x, int64_t ld, void* out, float* ws, int32_t nblk, int32_t rows,
              int32_t cols, int32_t accumulate, int32_t dtype, void* stream);

/* ----------------------------------------------------------- pointwise ---
 * RoPE (modeling.py:76-123): x layout [B,S,H,hd] (row pitch ld between
 * tokens), half-split rotation, cos/sin tables [max_pos, hd] in `dtype`
 * (already cast as modeling.py:121-122 does); pos[B*S] int32 position ids;
 * inverse=1 applies the transpose rotation (backward).
 */
int mk_rope(void* x, const void* cos_t, const void* sin_t, const int32_t* pos, int32_t tokens,
            int32_t heads, int32_t hd, int64_t ld, int32_t inverse, int32_t dtype, void* stream);
/* SwiGLU (modeling.py:140): a = silu(g) * u ; backward gives dg, du. */
int mk_swiglu_fwd(const void* g, const void* u, void* a, int64_t n, int32_t dtype, void* stream);
int mk_swiglu_bwd(const void* g, const void* u, const void* da, void* dg, void* du, int64_t n,
                  int32_t dtype, void* stream);
/* pitched 2-D forms for the fused gate|up projection buffer [rows, 2*cols] */
int mk_swiglu2d_fwd(const void* g, const void* u, void* a, int64_t rows, int32_t cols,
                    int64_t ld_in, int64_t ld_out, int32_t dtype, void* stream);
int mk_swiglu2d_bwd(const void* g, const void* u, const void* da, void* dg, void* du, int64_t rows,
                    int32_t cols, int64_t ld_gu, int64_t ld_a, int32_t dtype, void* stream);
/* y = act(x): act 1 gelu(erf) (Whisper), 2 quick_gelu (CLIP). */
int mk_act_fwd(const void* x, void* y, int64_t n, int32_t act, int32_t dtype, void* stream);
/* activation backward: dx = dy * act'(x_pre); act 1 gelu(erf), 2 quick_gelu. x_pre is the
 * pre-activation INCLUDING bias. */
int mk_act_bwd(const void* x_pre, const void* dy, void* dx, int64_t n, int32_t act, int32_t dtype,
               void* stream);
/* y = a + b (b may be broadcast over rows with period `period` elements; 0 = none) */
int mk_add(const void* a, const void* b, void* y, int64_t n, int64_t period, int32_t dtype,
           void* stream);
/* dtype conversion (inputs arrive as fp16 from llm_trainer.py:366-368). */
int mk_cast(const void* in, int32_t in_dtype, void* out, int32_t out_dtype, int64_t n,
            void* stream);
int mk_fill(void* p, float v, int64_t n, int32_t dtype, void* stream);
/* out[0] (+)= sum_i x[i]^2 in fp32, deterministic (global gradient norm for clipping: the
 * reference trains with HF max_grad_norm / DeepSpeed gradient_clipping, train.sh + configs/).
 * ws: f32 [1024] scratch; x 16-byte aligned. */
int mk_sumsq(const void* x, int64_t n, float* ws, float* out, int32_t accumulate, int32_t dtype,
             void* stream);
/* dst[z][r][0:cols] = src[z][r][0:cols] (pitched rows, batch strides; s_src = 0 broadcasts):
 * the torch.cat / slice / repeat plumbing of modeling.py:974-1046 without eager kernels. */
int mk_copy2d(const void* src, void* dst, int32_t rows, int32_t cols, int64_t ld_src,
              int64_t ld_dst, int32_t batch, int64_t s_src, int64_t s_dst, int32_t elem_size,
              void* stream);

/* Embedding gather (modeling.py:972,979-980): out[t] = table[ids[t]]; ids int64. */
int mk_embedding_fwd(const void* table, const int64_t* ids, void* out, int32_t tokens,
                     int32_t dim, int64_t ld_out, int32_t vocab, int32_t dtype, void* stream);
/* dtable[ids[t]] += dout[t], deterministic (no atomics): the first occurrence of each id
 * sums all of its occurrences in fp32 and updates the row once. Rows equal to padding_idx
 * (nn.Embedding padding_idx, modeling.py:358) are skipped; pass -1 for none. */
int mk_embedding_bwd(const void* dout, int64_t ld, const int64_t* ids, void* dtable,
                     int32_t tokens, int32_t dim, int32_t vocab, int64_t padding_idx,
                     int32_t dtype, void* stream);

/* Generic strided-window gather ("im2col") used for Conv2d patch embedding
 * (HF CLIPVisionEmbeddings), Whisper conv1/conv2 and the project_* Conv1d
 * (modeling.py:919-924): out[(b,j)][c*kw + t] = x[b*sb + c*sc + (j*stride + t - pad)*st]
 * (zero outside [0,T)), rows padded with zeros up to ld_out. 2-D patches are
 * expressed by the caller as two nested calls or by mk_patchify. */
int mk_im2col1d(const void* x, void* out, int32_t B, int32_t C, int32_t T, int32_t kw,
                int32_t stride, int32_t pad, int32_t Lout, int64_t sb, int64_t sc, int64_t st,
                int64_t ld_out, int32_t dtype, void* stream);
/* adjoint of mk_im2col1d (gather form, no atomics): dx = col2im(dcols). */
int mk_col2im1d(const void* dcols, void* dx, int32_t B, int32_t C, int32_t T, int32_t kw,
                int32_t stride, int32_t pad, int32_t Lout, int64_t sb, int64_t sc, int64_t st,
                int64_t ld_cols, int32_t dtype, void* stream);
/* Non-overlapping 2-D patches: out[(b,py,px)][c*P*P + dy*P + dx] = img[b][c][py*P+dy][px*P+dx] */
int mk_patchify(const void* img, void* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t P,
                int64_t ld_out, int32_t dtype, void* stream);
int mk_unpatchify(const void* dcols, void* dimg, int32_t B, int32_t C, int32_t H, int32_t W,
                  int32_t P, int64_t ld_cols, int32_t dtype, void* stream);

/* ------------------------------------------------------------- softmax ---
 * Row softmax over scores[z][q][k] with the reference's masking semantics
 * (modeling.py:205-214): causal (k > q + (Lk-Lq) masked) and/or key padding
 * mask kmask[b][Lk] (int32, 0 = masked); masked logits become finfo.min and
 * softmax runs in fp32 then casts.  Optional dropout on the probabilities
 * (nn.MultiheadAttention dropout=0.1, modeling.py:879): keep iff
 * hash(seed, element index) < keep_threshold, kept values scaled by
 * 1/(1-p).  In-place (probs may alias scores).  heads = rows of z per batch
 * element (kmask index = z / heads).
 */
int mk_softmax_fwd(const void* scores, void* probs, void* probs_dropped /* NULL if p == 0 */,
                   const int32_t* kmask, int32_t nz, int32_t heads, int32_t Lq, int32_t Lk,
                   int64_t ld, int32_t causal, float dropout_p, uint64_t seed, int32_t dtype,
                   void* stream);
/* dscores = (P .* (g - rowsum(g .* P))) * scale, g = dP_dropped .* keep/(1-p); in place on
 * dprobs.  probs are the PRE-dropout probabilities; the dropout mask is regenerated from
 * (seed, element index). */
int mk_softmax_bwd(const void* probs, void* dprobs, int32_t nz, int32_t Lq, int32_t Lk,
                   int64_t ld, float scale, float dropout_p, uint64_t seed, int32_t dtype,
                   void* stream);

/* Fused (flash) attention forward, bf16, head_dim 64 or 128: o = softmax(scale q k^T + mask) v
 * per (batch, head) without materialising the scores (modeling.py:197-215 / HF encoder
 * attention).  q/k/v/o are addressed as base + b*bs + token*ld + h*hd (elements); kmask
 * [B, Lk] int32 (0 = masked) optional; causal masks key > query + (Lk - Lq); lse [B, H, Lq]
 * f32 optional (log-sum-exp of the scaled, masked scores). */
int mk_flash_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                      const int32_t* kmask, int32_t B, int32_t H, int32_t Lq, int32_t Lk,
                      int32_t hd, int64_t q_ld, int64_t q_bs, int64_t k_ld, int64_t k_bs,
                      int64_t v_ld, int64_t v_bs, int64_t o_ld, int64_t o_bs, float scale,
                      int32_t causal, int32_t dtype, void* stream);

/* Fused attention backward (recompute form): dq, dk, dv from q, k, v, o, do and the forward's
 * lse; never stores P.  dq/dk/dv/do use the geometry of q/k/v/o.  dvec: f32 [B*H*Lq] scratch. */
int mk_flash_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                      const float* lse, float* dvec, void* dq, void* dk, void* dv,
                      const int32_t* kmask, int32_t B, int32_t H, int32_t Lq, int32_t Lk,
                      int32_t hd, int64_t q_ld, int64_t q_bs, int64_t k_ld, int64_t k_bs,
                      int64_t v_ld, int64_t v_bs, int64_t o_ld, int64_t o_bs, float scale,
                      int32_t causal, int32_t dtype, void* stream);

/* mk_flash_attn_fwd / _bwd with the rotary embedding of q and k (modeling.py:76-91, 167-170) folded in: q and k are
 * the UNROTATED projections, rotated on their way into the kernel with mk_rope's arithmetic (each product and the
 * sum rounded to the element type: q, k as the products see them are bit-identical to mk_rope followed by
 * mk_flash_attn_*); dq and dk come back as gradients of the unrotated tensors (rounded to the element type, then
 * rotated back -- what mk_rope(inverse) after mk_flash_attn_bwd yields).  cos_t / sin_t [positions][hd] of the
 * element type, 16-byte aligned; pos[b * Lq + token] int32.  Only where every q / k element enters the kernel once:
 * hd == 128, Lq == Lk <= 160 (the one-workgroup-per-(b, h) kernels); otherwise MK_ERR_UNSUPPORTED and the caller
 * runs mk_rope itself.  mk_flash_attn_rope_bwd with qk_rotated = 1: q and k ARE the rotated tensors (mk_rope ran
 * before mk_flash_attn_fwd) and only the rotation of dq / dk back is folded into the kernel's stores -- the form the
 * training step uses (one rotation per element instead of three: profiles/r06_rope_fuse.txt). */
int mk_flash_attn_rope_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                           const int32_t* kmask, const void* cos_t, const void* sin_t,
                           const int32_t* pos, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                           int64_t q_ld, int64_t q_bs, int64_t k_ld, int64_t k_bs, int64_t v_ld,
                           int64_t v_bs, int64_t o_ld, int64_t o_bs, float scale, int32_t causal,
                           int32_t dtype, void* stream);
int mk_flash_attn_rope_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                           const float* lse, float* dvec, void* dq, void* dk, void* dv,
                           const int32_t* kmask, const void* cos_t, const void* sin_t,
                           const int32_t* pos, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                           int64_t q_ld, int64_t q_bs, int64_t k_ld, int64_t k_bs, int64_t v_ld,
                           int64_t v_bs, int64_t o_ld, int64_t o_bs, float scale, int32_t causal,
                           int32_t qk_rotated, int32_t dtype, void* stream);

/* Shifted cross-entropy (modeling.py:600-610). The caller passes labels already shifted
 * (row r predicts labels[r]; -100 = ignore).  row_loss[r] = lse_r - logit[r][label] (0 when
 * ignored), row_lse[r] kept for backward, loss_sum_cnt = {sum of row losses, number of valid
 * rows, mean loss, 0} (f32[4], reduced deterministically on device). */
int mk_cross_entropy(const void* logits, const int64_t* labels, float* row_loss, float* row_lse,
                     float* loss_sum_cnt, int32_t rows, int32_t V, int64_t ld, int32_t dtype,
                     void* stream);
/* dlogits[r][c] = (softmax(logits[r])[c] - [c == label]) * grad_scale / n_valid, zero for
 * ignored rows and for pad columns c in [V, ld).  dlogits may alias logits. */
int mk_cross_entropy_bwd(const void* logits, void* dlogits, const int64_t* labels,
                         const float* row_lse, const float* loss_sum_cnt, float grad_scale,
                         const float* grad_scale_dev /* optional device scalar multiplied in */,
                         int32_t rows, int32_t V, int64_t ld, int32_t dtype, void* stream);

/* Greedy decoding (modeling.py:959, HF greedy_search): out[r] = index of the first maximum of
 * row r of x[rows][cols] (pitch ld). */
int mk_argmax_rows(const void* x, int64_t ld, int32_t rows, int32_t cols, int64_t* out,
                   int32_t dtype, void* stream);

/* Decode step with the position in DEVICE memory (a fixed launch sequence per token: capturable in
 * a hipGraph; modeling.py:190-195 appends with torch.cat and modeling.py:954-960 re-launches the
 * eager step per token).
 * mk_kv_append: cache[b][t][0:cols] = src[b][0:cols] for b < batch, t = clamp(*t_dev, 0, t_max - 1);
 *   strides in elements: s_src / s_cache between samples, ld_cache between cache rows; 16-byte
 *   aligned rows.
 * mk_decode_attn: o[b][h][:] = softmax(scale * q[b][h] . k[b][0:T][h]) v[b][0:T][h] with
 *   T = clamp(*t_dev + t_add, 1, t_max); one query row per (sample, head); q / o rows are
 *   [H * hd] at batch strides q_bs / o_bs, keys and values [t_max][H * hd] at pitches k_ld / v_ld and
 *   batch strides k_bs / v_bs.  bf16, hd in {16, 32, 64, 128}, t_max <= 15360 (scores in LDS). */
/* mk_decode_linear: y[M][N] = prologue(x) W^T (+ residual) for M <= 16 token rows (M <= 32 without a
 *   prologue; bf16, K % 64 == 0,
 *   16-byte aligned rows), the weight-streaming kernel of mk_gemm's M <= 16 path with the operation
 *   that precedes the linear folded in: prologue 0 = none (x [M][K]); 1 = RMSNorm (x [M][K], norm_w
 *   [K], eps: y = (norm_w * rnd(x * rstd)) W^T, modeling.py:100-105); 2 = SwiGLU (x [M][2K] =
 *   [gate | up]: (rnd(silu(gate)) * up) W^T, modeling.py:140).  With a prologue the prepared token rows
 *   are staged in LDS: M * (K + 8) * 2 bytes <= 40 KiB, else MK_ERR_UNSUPPORTED (use the separate
 *   kernels).  17 <= M <= 32: 32 weight rows x 32 token rows per workgroup where N >= 8192, two
 *   16-token tiles per 16 weight rows below (the same choice mk_gemm makes for M <= 32). */
int mk_decode_linear(const void* x, int64_t ldx, const void* W, int64_t ldw, void* y, int64_t ldy,
                     const void* residual, int64_t ldr, int32_t M, int32_t N, int32_t K,
                     int32_t prologue, const void* norm_w, float eps, int32_t dtype, void* stream);
/* mk_decode_emit: greedy selection + bookkeeping of one decode step (modeling.py:959, HF greedy_search):
 *   for every sample b: nxt = done[b] ? pad : first argmax of logits[b][0:V]; out[b][state[1]] = nxt;
 *   done[b] |= (nxt == eos); tok[b] = nxt.  Then, once: state[0] += 1 (the position the other decode
 *   kernels read), state[1] += 1 (output column); state[2] is an arrival counter that must be zero
 *   before the first launch and is left zero.  done: one byte per sample. */
int mk_decode_emit(const void* logits, int64_t ld, int32_t V, int32_t B, int64_t pad, int64_t eos,
                   int64_t* tok, void* done, int64_t* out, int64_t out_ld, int32_t* state, int32_t dtype,
                   void* stream);
/* mk_decode_step_attn: the attention block of one decode step in one launch: p = clamp(*t_dev, 0,
 *   t_max - 1); q and k_new (rows [H * hd] at batch stride in_bs) are rotated with rows p of the
 *   [positions][hd] cos / sin tables (modeling.py:76-91, rope rounding points of mk_rope); the rotated
 *   key and v_new go to row p of the caches; o = attention of the rotated query over keys 0 ... p. */
int mk_decode_step_attn(const void* q, const void* k_new, const void* v_new, int64_t in_bs,
                        const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                        int64_t kv_ld, int64_t kv_bs, void* o, int64_t o_bs, const int32_t* t_dev,
                        int32_t t_max, int32_t B, int32_t H, int32_t hd, float scale, int32_t dtype,
                        void* stream);
int mk_kv_append(const void* src, void* cache, int32_t cols, int32_t batch, int64_t s_src,
                 int64_t s_cache, int64_t ld_cache, const int32_t* t_dev, int32_t t_max,
                 int32_t elem_size, void* stream);
int mk_decode_attn(const void* q, const void* k, const void* v, void* o, const int32_t* t_dev,
                   int32_t t_add, int32_t t_max, int32_t B, int32_t H, int32_t hd, int64_t q_bs,
                   int64_t k_ld, int64_t k_bs, int64_t v_ld, int64_t v_bs, int64_t o_bs, float scale,
                   int32_t dtype, void* stream);

/* ------------------------------------------------------------ optimizer --
 * Fused AdamW over a flat shard (replaces DeepSpeed CPU-offloaded Adam,
 * configs/deepspeed_config.json:2-13): fp32 master/m/v, `dtype` grads and
 * model weights.  grad_scale multiplies grads first (loss scaling / 1/world). */
int mk_adamw(void* param, float* master, float* m, float* v, const void* grad, int64_t n,
             float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
             float grad_scale, int32_t dtype, void* stream);

/* Caps mk_adamw's grid (blocks of 256 threads; 0 = default).  A small grid turns the grid-stride update into a
 * persistent kernel that holds only a few CUs: the per-bucket updates of one rank then run BESIDE the backward's
 * GEMMs (planned for fewer CUs through mk_gemm_set_cus) instead of as a serial tail -- the single-GPU counterpart of
 * overlap_comm (configs/deepspeed_config.json:32).  Process-global like mk_gemm_set_cus; returns the previous cap. */
int mk_adamw_set_max_blocks(int32_t n_blocks);

/* A whole training step replayed from a hipGraph (macaw_llm_amd.train.GraphedStep) cannot change
 * kernel ARGUMENTS between steps; the two per-step scalars of the path live in device memory:
 *  - mk_adamw_multi_dev: mk_adamw_multi with hyper_dev = {lr, 1 - beta1^step, 1 - beta2^step,
 *    grad_scale} (device, f32[4]); mk_adamw_bias_correction fills the two corrections on the HOST
 *    with mk_adamw_multi's own arithmetic (bit-identical updates);
 *  - mk_set_dropout_seed_offset: with a device pointer registered every mk_softmax_fwd / _bwd launch
 *    adds *dev_ptr to its seed when it executes; NULL (default) switches it off. */
int mk_adamw_bias_correction(float beta1, float beta2, int32_t step, float* out2);
int mk_adamw_multi_dev(const void* items, const int64_t* chunk_start, int32_t n_items, int64_t n_chunks,
                       float beta1, float beta2, float eps, float weight_decay, const float* hyper_dev,
                       int32_t dtype, void* stream);
int mk_set_dropout_seed_offset(const uint64_t* dev_ptr);

/* ------------------------------------------------------------------ fp8 --
 * BASELINE cfg 5 ("fp8 MFMA for alignment-attn and QKV GEMMs"): per-tensor scaled OCP e4m3.
 * q[i] = e4m3(clamp(x[i] * 448 / amax(|x|), +-448)); *dequant_scale = amax / 448 (1 if amax == 0),
 * written on the DEVICE and handed to mk_gemm as scale_a / scale_b: no host round trip.
 * amax_ws: device float[1] scratch.  n % 8 == 0, 16-byte aligned x, 8-byte aligned q. */
int mk_fp8_quantize(const void* x, int64_t n, int32_t dtype, uint8_t* q, float* amax_ws,
                    float* dequant_scale, void* stream);
/* Per-ROW scaled e4m3 of a row-major (pitched) [rows, cols] bf16 matrix (activations, gradients,
 * K-major weights = one scale per output channel): q[r, c] = e4m3(x[r, c] * 448 / amax_r),
 * scales[r] = amax_r / 448 (1 for a zero row).  cols % 8 == 0; GEMM operands need cols % 128 == 0. */
int mk_fp8_quantize_rows(const void* x, int32_t rows, int32_t cols, int64_t ld, int32_t dtype,
                         uint8_t* q, int64_t ldq, float* scales, void* stream);
/* mk_rmsnorm_fwd (LlamaRMSNorm, modeling.py:311-319) with mk_fp8_quantize_rows of its output y folded in: h_out (with
 * res), y, rstd as mk_rmsnorm_fwd writes them, plus q [rows, cols] e4m3 (pitch ldq) and scales[rows] of y -- the
 * activation operand of the fp8 q|k|v GEMM (BASELINE cfg 5) without a second pass over y.  Bit-identical to the two
 * calls.  bf16, cols % 8 == 0, cols <= 16384, contiguous rows, 16-byte aligned. */
int mk_rmsnorm_fwd_fp8(const void* x, const void* res, const void* w, void* h_out, void* y, float* rstd,
                       uint8_t* q, int64_t ldq, float* scales, int32_t rows, int32_t cols, float eps,
                       int32_t dtype, void* stream);
/* Per-COLUMN scaled e4m3 with TRANSPOSED output: qt[c, r] = e4m3(x[r, c] * 448 / amax_c),
 * scales[c] = amax_c / 448 -- the operand of the fp8 grad-input GEMM dx = dy W (the reduction runs
 * over the rows of W [out, in], modeling.py:159-162: nn.Linear stores [out, in]), i.e. W^T K-major
 * with one scale per input channel.  amax_ws: device float[cols] scratch.  rows, cols % 8 == 0. */
int mk_fp8_quantize_cols_t(const void* x, int32_t rows, int32_t cols, int64_t ld, int32_t dtype,
                           uint8_t* qt, int64_t ldqt, float* scales, float* amax_ws, void* stream);

/* ------------------------------------------------------ host-input pipeline --
 * The per-step CPU work of llm_trainer.py:306-381 (get_self_inputs) moved to the GPU.
 *
 * mk_image_transform = CLIP `_transform(n_px)` (llm_trainer.py:150-157): torchvision
 * Resize(n_px, BICUBIC) on a PIL image -> CenterCrop(n_px) -> ToTensor -> Normalize, for a batch
 * of RGB uint8 HWC images of different sizes packed back to back in `src`.  Bit-exact with
 * Pillow 12.2 ImagingResample (libImaging/Resample.c): the host computes Pillow's 22-bit
 * fixed-point bicubic coefficients (macaw_llm_amd/preprocess.py) for the cropped window only.
 *   descs  : n_images x 12 int64 {src_off, H, W, tmp_off, row0, nrows, hk_off, hb_off, hks,
 *            vk_off, vb_off, vks}  (offsets into src/tmp in bytes, into coef in int32 elements)
 *   coef   : int32 coefficient rows [out_px][ks] and bounds [out_px][2] = {first tap, n taps}
 *   lut    : float[3][256] = ((v / 255) - mean[c]) / std[c] as the reference evaluates it
 *   tmp    : scratch for the horizontally resampled rows (sum of nrows * out_px * 3 bytes)
 *   out    : [n_images][3][out_px][out_px] in `dtype` (MK_F32 / MK_BF16 / MK_F16) */
int mk_image_transform(const uint8_t* src, uint8_t* tmp, const int64_t* descs, const int32_t* coef,
                       const float* lut, void* out, int32_t n_images, int32_t out_px,
                       int64_t max_tmp_rows, int32_t dtype, void* stream);

/* mk_log_mel = whisper.log_mel_spectrogram (llm_trainer.py:343, openai-whisper audio.py):
 * torch.stft(audio, 400, 160, hann, center/reflect) -> |.|^2 without the last frame ->
 * mel_filters[n_mels][201] @ . -> log10(clamp 1e-10) -> max(., per-clip max - 8) -> (. + 4) / 4.
 * audio [n_clips][ld] f32 with n_samples (multiple of 160) valid samples per clip;
 * window f32[400]; twiddle f64[400][2] = (cos, sin)(2 pi j / 400); mel_lo/mel_hi = first / one
 * past last non-zero bin of each filter; ws_logspec f32[n_clips * n_mels * n_samples/160],
 * ws_max int32[n_clips]; out [n_clips][n_mels][n_samples/160] in `dtype`. */
int mk_log_mel(const float* audio, int64_t ld, int32_t n_clips, int32_t n_samples,
               const float* window, const double* twiddle, const float* mel_filters,
               const int32_t* mel_lo, const int32_t* mel_hi, int32_t n_mels, float* ws_logspec,
               int32_t* ws_max, void* out, int32_t dtype, void* stream);

/* Multi-tensor AdamW: one launch for a whole list of tensors (same arithmetic per element as
 * mk_adamw).  items: DEVICE array of n_items records {param, master, m, v, grad, n} (6 x 8 bytes,
 * every pointer 16-byte aligned); chunk_start: DEVICE int64[n_items + 1], prefix sum of
 * ceil(n / mk_adamw_chunk()) per item; n_chunks = chunk_start[n_items] (= grid size).  The gradient is read with the
 * non-temporal hint (touched once per step); every store is a plain one. */
/* elements per workgroup slice of mk_adamw_multi / _dev: chunk_start[i] = sum over items j < i of ceil(n_j / mk_adamw_chunk()) */
int mk_adamw_chunk(void);
int mk_adamw_multi(const void* items, const int64_t* chunk_start, int32_t n_items, int64_t n_chunks,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                   float grad_scale, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MACAW_HIP_H */
