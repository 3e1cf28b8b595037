/*
 * a3vlm_hip.h -- C ABI of liba3vlm_hip.so: the MI355X (gfx950 / CDNA4) kernels of the
 * A3VLM multimodal hot path (ViT patch-embed + encoder -> projector -> Llama decoder
 * -> LM head / CE / greedy decode).
 *
 * The reference (changhaonan/A3VLM) is pure Python on PyTorch and has no FFI of its
 * own: its "native" layer is whatever ATen/cuBLAS/flash-attn kernel each torch call
 * reaches.  Each entry point below therefore cites the
 * reference call site (path:line under model/accessory/) whose arithmetic it
 * replaces.  The reference-side binding is a ctypes stub.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are
 *    enqueued asynchronously on it and never synchronise.
 *  - never allocates; scratch is passed in by the caller.
 *  - re-entrant from any host thread.  The only process-wide state is (i) the table of scratch registrations of
 *    a3v_gemm_set_workspace_for (mutex-protected, keyed by device and stream: two streams never share split-K planes) and (ii) the
 *    cached values of the A3V_* environment switches (A/B runs only; a3v_reload_env re-reads them).  No kernel result depends on
 *    state left by an earlier call on another stream.
 *  - return value: 0 on success, a hipError_t (>0) from the launch, or a negative
 *    A3V_ERR_* for argument errors.  The Python host raises RuntimeError on != 0.
 *  - dtype codes: A3V_BF16 activations/weights are bfloat16 with fp32 accumulation
 *    (the throughput path); A3V_F32 is the fp32 parity path (same op order).
 *  - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef A3VLM_HIP_H
#define A3VLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { A3V_BF16 = 0, A3V_F32 = 1 };

enum {
  A3V_OK = 0,
  A3V_ERR_SHAPE = -1,   /* unsupported size / alignment */
  A3V_ERR_DTYPE = -2,
  A3V_ERR_ARG = -3
};

/* GEMM epilogue flags (bit-or) */
enum {
  A3V_EPI_NONE = 0,
  A3V_EPI_BIAS = 1,        /* + bias[n]                                             */
  A3V_EPI_GELU = 2,        /* erf GELU   (open_clip mlp)          */
  A3V_EPI_QUICKGELU = 4,   /* x*sigmoid(1.702x) (open_clip 'openai' pretrained cfg)  */
  A3V_EPI_RESIDUAL = 8,    /* out = residual + y  (LLM/llama_ens5.py:238,241)        */
  A3V_EPI_SWIGLU = 16,     /* W rows interleaved [w1 blk16 | w3 blk16]: out[:, N/2] =
                              silu(x w1^T) * (x w3^T)   (LLM/llama_ens5.py:213-217)   */
  A3V_EPI_OUT_F32 = 32,    /* store fp32 (logits .float(), LLM/llama_ens5.py:531)     */
  A3V_EPI_RES_F32 = 64,    /* residual stream (and output) kept in fp32 (training
                              under autocast: engine_finetune.py:44-50)               */
  A3V_EPI_SWIGLU_BWD = 128, /* the GEMM's [M, N] product is d(act) of the SwiGLU (the input gradient of w2, llama_ens5.py:213-217
                               backward): `residual` = gu [M, 2N] bf16 (gate columns 0..N-1, up columns N..2N-1, as the forward's
                               un-fused w1|w3 output keeps them), C [M, >= 2N] bf16 receives d(gate) in columns 0..N-1 and d(up)
                               in N..2N-1 -- a3v_swiglu_bwd applied to the bf16-rounded product, bit for bit, without the
                               d(act) round trip through HBM.  bf16 only; N % 8 == 0.                                   */
  A3V_EPI_TILE_128 = 1 << 16,  /* force the 128x128 tile kernel (tuning / tests)       */
  A3V_EPI_TILE_256 = 1 << 17,  /* force the 256x256 tile kernel (tuning / tests)       */
  A3V_EPI_TILE_256PP = 1 << 18,/* force the 256x256 ping-pong kernel (tuning / tests)  */
  A3V_EPI_TILE_256PP32 = 1 << 19, /* ... its 32x32x16-MFMA form                        */
  A3V_EPI_TILE_192PP = 1 << 23  /* force the ring kernel's 192 x 256 tile form (round 5; tuning / tests) */
};

int a3v_version(void);

/* A3V_* environment switches (A/B runs, tuning scripts, equality tests) are read once per call site and
 * cached: a process that changes one after its first launch calls this to have them re-read.  Returns the new generation.
 * (No reference counterpart: the reference has no native code.) */
int a3v_reload_env(void);
/* bit 0: the library was built with -DA3V_EXPERIMENTS (`make EXPERIMENTS=1`): the measured-and-not-dispatched GEMM kernels
 * (two-stage ping-pong, one wave per SIMD, overlapped, 32x32x16 forms, stamped builds) and their switches exist. */
int a3v_build_flags(void);
/* C[M,N] = epilogue(A[M,K] @ W[N,K]^T).  Replaces every F.linear on the path:
 * wq/wk/wv/wo (LLM/llama_ens5.py:63-90,112,169), w1/w2/w3 (:202-217), output (:267-269,
 * 486,530), visual_proj[0] (:330-333), the open_clip in_proj/out_proj/c_fc/c_proj, and
 * conv1 as an im2col GEMM (:354).  bf16: K % 64 == 0, lda/ldw % 8 == 0, N % 4 == 0.
 * With A3V_EPI_SWIGLU, N counts the interleaved rows (2*ffn) and C has N/2 columns. */
int a3v_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                int epilogue, int dtype, void* stream);

/* Prefill qkv projection with the rotary embedding and the KV-cache write in the GEMM epilogue (LLM/llama_ens5.py:155-176:
 * xq, xk, xv = wq(x), wk(x), wv(x); apply_rotary_emb; cache_k/cache_v[:bsz, start_pos:start_pos+seqlen] = xk/xv) --
 * a3v_gemm_nt on the fused [(H+2Hkv)*hd, K] weight followed by a3v_rope_kvcache, value for value (the accumulator is rounded
 * to the bf16 activation before the rotation), minus one HBM round trip of the [B*S, (H+2Hkv)*hd] activation.
 * A [B*S, K] bf16; q_out [B*S, ldq] receives the rotated q in columns 0..H*hd; caches as a3v_rope_kvcache; hd 64 or 128.
 * v_rows (optional, [B*S, ldv]): v also stored token-major, the layout the attention backward reads (training forward).
 * delta (optional, bf16 [B*S, ldd]): an additive term of the projection -- the LoRA branch lora_b(lora_a(x)) of model/peft.py:89-95
 * -- added to the bf16 linear output (and rounded, as the reference does) before the rotation; may alias v_rows' buffer. */
int a3v_gemm_qkv_rope(const void* A, int64_t lda, const void* W, int64_t ldw, int K, void* q_out, int64_t ldq,
                      void* k_cache, void* vt_cache, void* v_rows, int64_t ldv, const void* delta, int64_t ldd,
                      const float* cos_sin, int B, int S, int H, int Hkv, int hd, int Smax, int start_pos, int rope_pos0,
                      void* stream);

/* fp8 (OCP e4m3fn) W8A8 prefill GEMM (BASELINE config 5 "quantised inference"; the reference's quantised path is bitsandbytes
 * NF4/INT8, util/quant.py:95-163, so there is no reference oracle: parity is stated against the exact product of the
 * dequantised operands): C = epilogue((Aq . Wq^T) * sa[m] * sw[n]) on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales.
 * Aq [M, K] / Wq [N, K] fp8 bytes (lda, ldw % 16 == 0, K % 128 == 0), sa [M] / sw [N] fp32 per-row scales; C bf16 (or fp32 with
 * OUT_F32 / RES_F32); epilogues BIAS / GELU / QUICKGELU / RESIDUAL / SWIGLU / OUT_F32 / RES_F32 as a3v_gemm_nt. */
int a3v_gemm_nt_fp8(const void* Aq, int64_t lda, const float* sa, const void* Wq, int64_t ldw, const float* sw, void* C,
                    int64_t ldc, int M, int N, int K, const void* bias, const void* residual, int64_t ldr, int epilogue,
                    void* stream);
/* a3v_gemm_qkv_rope with fp8 operands. */
int a3v_gemm_qkv_rope_fp8(const void* Aq, int64_t lda, const float* sa, const void* Wq, int64_t ldw, const float* sw, int K,
                          void* q_out, int64_t ldq, void* k_cache, void* vt_cache, const float* cos_sin, int B, int S,
                          int H, int Hkv, int hd, int Smax, int start_pos, int rope_pos0, void* stream);
/* Dynamic per-row activation quantisation for a3v_gemm_nt_fp8: scales[r] = max|y[r,:]| / 448, q[r,k] = fp8(y[r,k] / scales[r]),
 * y = x (bf16 rows) or, with norm_w != NULL, the bf16 RMSNorm of x (model/components.py:39,52-53) without writing it out.
 * dim % 8 == 0, dim <= 16384. */
int a3v_quantize_rows_fp8(const void* x, int64_t ldx, const void* norm_w, float eps, void* q, int64_t ldq, float* scales,
                          int rows, int dim, int x_dtype, void* stream);

/* "TN" GEMM: C[M,N] = epilogue(At^T . Wt) with At [K, M] and Wt [K, N] (the contracted index is the ROW index of both
 * operands): the weight gradient dW = dY^T . X on the token-major activations autograd holds (engine_finetune.py:55-57
 * loss.backward()), without transposing either.  lda, ldw % 8 == 0, M % 8 == 0; epilogues NONE / RESIDUAL / RES_F32 / OUT_F32. */
int a3v_gemm_tn(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int M, int N,
                int K, const void* residual, int64_t ldr, int epilogue, void* stream);

/* a3v_gemm_tn with an fp32 output kind (A3V_EPI_OUT_F32 / A3V_EPI_RES_F32: the weight gradients) that also leaves the sum of
 * squares of the values it stored, as a3v_gemm_tn_sumsq_slots(M, N) partial sums in `sumsq` (the caller zeroes the buffer once per
 * step; slots of tiles outside C are not written).  The global-norm clip (reference util/clip_grad.py:59-210) then needs no pass
 * over these gradients at all. */
int64_t a3v_gemm_tn_sumsq_slots(int M, int N);
int a3v_gemm_tn_sumsq(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                      const void* residual, int64_t ldr, int epilogue, float* sumsq, int64_t sumsq_cap, void* stream);

/* "NN" GEMM: C[M,N] = epilogue(A . Wt), A [M, K] and Wt [K, N] both row-major (the contracted index is Wt's ROW index): the
 * input gradient dX = dY . W of F.linear (autograd of LLM/llama_ens5.py:112,169,214-217) on the weight image the forward pass
 * uses -- no transposed copy of W.  K % 64 == 0, N % 8 == 0; epilogues NONE / RESIDUAL / RES_F32 / OUT_F32. */
int a3v_gemm_nn(const void* A, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                const void* residual, int64_t ldr, int epilogue, void* stream);

/* Split-K form of a3v_gemm_tn for adapter-sized outputs (the LoRA weight gradients, model/peft.py:40-64 under autograd):
 * partial [S][M][N] raw fp32 planes, to be summed by a3v_splitk_reduce. */
int a3v_gemm_tn_splitk(const void* At, int64_t lda, const void* Wt, int64_t ldw, float* partial, int M, int N, int K,
                       int S, void* stream);

/* Optional: register a scratch buffer (device memory, >= 32 MiB recommended; NULL unregisters) that a3v_gemm_nt / _nn / _tn /
 * _nt_fp8 may use for the split-K planes of their hybrid tile dispatch (the few-hundred-row tail, few-tile problems).  The library
 * itself never allocates.  Registrations are keyed by (current device, stream): GEMM calls issued on `stream` use that buffer and no
 * other, so two streams (a second model, an optimizer stream, a second device of one process) never share planes.  The buffer must
 * outlive every later GEMM call on that stream. */
int a3v_gemm_set_workspace_for(void* stream, void* ptr, int64_t bytes);
/* Legacy form for callers that do not name a stream: the buffer is bound to the FIRST (device, stream) whose GEMM call uses it; calls
 * on any other stream without a registration of their own get no scratch (plain launches), never this buffer. */
int a3v_gemm_set_workspace(void* ptr, int64_t bytes);

/* Split-K form for skinny products with a long K (the LoRA adapter GEMMs of model/peft.py:84-99 and their gradients:
 * N or M = 64, K = 4096 ... 22016): slice s of S writes the fp32 plane partial[s][M][N]; a3v_splitk_reduce sums the
 * planes in order, optionally accumulates into `out` (fp32 gradients) and rounds once to out_dtype.
 * N <= 64, M >= 512: the streamed operand A goes through a multi-stage LDS ring (k-tiles stay in flight across the block barrier) as
 * 256-row blocks when S * ceil(M / 256) fills between half and all of the CUs -- one resident block per CU, the caller picks S for that
 * (S = CUs / ceil(M / 256)) -- else as 64-row blocks; every form writes the same planes bit for bit. */
int a3v_gemm_nt_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, float* partial, int M, int N,
                       int K, int S, void* stream);
int a3v_splitk_reduce(const float* partial, int S, int M, int N, void* out, int64_t ldo, int out_dtype,
                      int accumulate, void* stream);

/* Skinny-M (decode) form of the same contract, M <= 16: streams W once from HBM (one launch).
 * `partial` is a workspace of a3v_gemm_skinny_ws_bytes(M, N, K) bytes: the first 16384 bytes are
 * split-K arrival counters that the caller zero-fills ONCE; every call leaves them zero.  The
 * workspace must not be shared by calls that can run concurrently (different streams).
 * a3v_gemm_skinny_split reports the split-K factor used.  Epilogues: NONE, RESIDUAL, SWIGLU, OUT_F32. */
int a3v_gemm_skinny_split(int M, int N, int K);
int64_t a3v_gemm_skinny_ws_bytes(int M, int N, int K);
/* Weight-only fp8 form (BASELINE config 5): Wq [N,K] OCP e4m3fn bytes (row stride ldw in BYTES,
 * % 16 == 0), wscale [N] fp32 per-row dequantisation scales; C = epilogue((A . float(Wq)^T) * wscale), K % 256 == 0.
 * The reference's quantised path is CUDA-only bitsandbytes NF4 (util/quant.py:95-163): there is NO reference oracle for
 * fp8 -- parity is stated against the bf16 kernel on the dequantised weights. */
int a3v_gemm_skinny_fp8(const void* A, int64_t lda, const void* Wq, int64_t ldw, const float* wscale, void* C,
                        int64_t ldc, int M, int N, int K, const void* residual, int64_t ldr, int epilogue,
                        void* workspace, void* stream);
int a3v_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                    int M, int N, int K, const void* residual, int64_t ldr, int epilogue,
                    void* partial, void* stream);

/* y = rmsnorm(x) * w  with the reference's rounding order (model/components.py:39,52-53):
 * fp32 normalise -> cast to x's dtype -> multiply by weight.  x_dtype may be A3V_F32 with
 * y bf16 (training under autocast: fp32 residual stream feeding bf16 GEMMs). */
int a3v_rmsnorm(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int rows, int dim,
                float eps, int x_dtype, int w_dtype, int y_dtype, void* stream);

/* y[row_map ? row_map[r] : r] = layernorm(x[r]) * w + b   (torch.nn.LayerNorm, eps 1e-5):
 * open_clip ln_pre/ln_1/ln_2/ln_post and the projector LayerNorm (LLM/llama_ens5.py:325-333,
 * 363,370).  row_map (int32, device) lets the projector write straight into the
 * [BOS | image tokens | text] sequence buffer (:471-479). */
int a3v_layernorm(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy,
                  const int32_t* row_map, int rows, int dim, float eps, int x_dtype, int p_dtype,
                  int y_dtype, void* stream);

/* RoPE on q,k (interleaved pairs, fp32 math, LLM/llama_ens5.py:118 + restated llama.py) and
 * KV-cache write (:124-129) from a fused qkv activation [B*S, (H+2*Hkv)*hd]:
 *   q_out [B*S, ldq]            rotated q (may alias qkv: in-place, ldq = ldqkv)
 *   k_cache[B,Hkv,Smax,hd]      rotated k at positions start_pos..start_pos+S-1
 *   vt_cache[B,Hkv,hd,Smax]     v TRANSPOSED (this library's cache layout: the attention
 *                               kernels consume V^T tiles with no in-loop transpose)
 * cos_sin: fp32 table [n_pos, hd/2, 2]; row = rope_pos0 + s. */
int a3v_rope_kvcache(const void* qkv, int64_t ldqkv, void* q_out, int64_t ldq, void* k_cache,
                     void* vt_cache, const float* cos_sin, int B, int S, int H, int Hkv, int hd,
                     int Smax, int start_pos, int rope_pos0, int dtype, void* stream);

/* v [N, L, H*hd] (row stride ldv, e.g. the v third of nn.MultiheadAttention's packed
 * in_proj output) -> vt [N, H, hd, Lpad] zero padded: the V^T operand of a3v_attention. */
int a3v_vt_pack(const void* v, int64_t ldv, void* vt, int N, int L, int H, int hd, int Lpad,
                int dtype, void* stream);

/* softmax(q k^T / sqrt(hd) [+ right-aligned causal mask]) v, flash style, fp32 softmax.
 * Replaces F.scaled_dot_product_attention / flash_attn_func (LLM/llama_ens5.py:142-167,
 * mask :181-185) and nn.MultiheadAttention's core (open_clip resblocks).
 * strides (ELEMENTS, host array of 12): q {batch, seq, head}, k {batch, kv-head, seq},
 * vt {batch, kv-head, d}, out {batch, seq, head}; hd is contiguous in q/k/out and seq is
 * contiguous in vt.  bf16: hd in {64,128}, strides multiples of 8.  Sq == 1 selects the
 * split-KV decode kernel (`scratch`: fp32, a3v_attention_scratch_floats() entries; may be
 * NULL when Sq > 1).  causal requires Sk >= Sq (queries are the LAST Sq positions). */
int64_t a3v_attention_scratch_floats(int B, int H, int hd, int Sk);
int a3v_attention(const void* q, const void* k, const void* vt, void* out, int B, int Sq, int Sk,
                  int H, int Hkv, int hd, const int64_t* strides, int causal, float* scratch,
                  int dtype, void* stream);

/* h[b, s, :] for the sequence [BOS | image words | text]: rows s==0 and s>W come from the
 * embedding table (tok_embeddings, LLM/llama_ens5.py:464,495), rows 1..W are left untouched
 * (written by a3v_layernorm(row_map) / a3v_fill_rows).  tokens int64 [B,T]. */
int a3v_embed_assemble(const int64_t* tokens, int64_t ld_tok, const void* table, void* h, int B,
                       int T, int W, int dim, int vocab, int table_dtype, int h_dtype, void* stream);

/* dst[row_idx[i], :] = src[0, :]  (start_img / end_img tags, LLM/llama_ens5.py:472-474) */
int a3v_fill_rows(const void* src, void* dst, int64_t ldd, const int32_t* row_idx, int n_rows,
                  int dim, int src_dtype, int dst_dtype, void* stream);

/* im2col for the k=s=P, bias-free patch-embed conv (LLM/llama_ens5.py:354): img [N,3,Hi,Wi]
 * -> cols [N*g*g, Kpad] with k = c*P*P + py*P + px (conv1.weight.view(width,-1) order),
 * zero padded to Kpad. */
int a3v_patch_im2col(const void* img, void* cols, int N, int Hi, int Wi, int P, int Kpad,
                     int in_dtype, int out_dtype, void* stream);

/* LLM/llama_ens5.py:383-385: img [B,3,2c,2c] -> out [5B,3,c,c] = [fp16 bicubic 2x downsample
 * of the whole image ; top-left ; top-right ; bottom-left ; bottom-right]. */
int a3v_split_views(const void* img, void* out, int B, int crop, int in_dtype, int out_dtype,
                    void* stream);

/* x[n, 0, :] = cls + pos[0]; x[n, 1+t, :] = patch[n*T+t, :] + pos[1+t]  (:358-362) */
int a3v_vit_embed(const void* patch, const void* cls, const void* pos, void* x, int N, int T,
                  int width, int dtype, void* stream);

/* Image preprocessing of the input contract on the device (data/transform.py:13-68: PadToSquare with the CLIP-mean fill ->
 * Resize(bicubic) -> ToTensor -> Normalize): src = one decoded RGB image, uint8 [H][W][3]; it sits at (pad_x, pad_y) inside a
 * side x side square filled with fill_rgb; the square is resized to out_size x out_size with Pillow's 8-bit two-pass resampling
 * (the reference transform runs torchvision on PIL images, i.e. Pillow's ImagingResample): kx / ky = the fixed-point (2^22)
 * coefficient rows of the horizontal / vertical pass, ksize_* ints per output sample, bx / by = (first input index, tap count)
 * per output sample -- tables computed on the host from (side, out_size) exactly as Resample.c does (a3vlm_amd/data/transform.py).
 * tmp: uint8 scratch [side][out_size][3].  dst: [3][out_size][out_size] in dst_dtype, ((v / 255) - mean[c]) / std[c] with fp32
 * arithmetic.  Host-side scalars (fill_rgb, mean, std) are HOST pointers; everything else is device memory.  Bit-identical to
 * the PIL path (tests/test_gpu_preprocess.py). */
int a3v_preprocess_image(const uint8_t* src, int H, int W, int side, int pad_x, int pad_y, const int* fill_rgb,
                         const int32_t* kx, const int32_t* bx, int ksize_x, const int32_t* ky, const int32_t* by, int ksize_y,
                         int out_size, uint8_t* tmp, void* dst, int dst_dtype, const float* mean, const float* std, void* stream);

/* The same transform for a BATCH of decoded images (the loaders of eval_affordance_v2.py:109-180 and main_finetune.py's dataset
 * build: workers decode to uint8 HWC, the device does the rest), two launches per 16 images.  Each image carries the tables of ITS
 * padded side (coeffs / bounds of a3vlm_amd/data/transform.py:pillow_bicubic_coeffs, used for both passes: the square is resized
 * to a square).  `images` is a HOST array of n descriptors whose pointers are DEVICE pointers.  tmp: uint8 scratch, tmp_stride bytes
 * per image (>= side x out_size x 3); dst: [n][3][out_size][out_size] in dst_dtype, dst_stride ELEMENTS per image.  Same arithmetic,
 * bit for bit, as a3v_preprocess_image. */
typedef struct {
  const uint8_t* src;        /* [H][W][3] */
  const int32_t* coeffs;     /* [out_size][ksize] */
  const int32_t* bounds;     /* [out_size][2] */
  int H, W, side, pad_x, pad_y, ksize;
} a3v_image_desc;
int a3v_preprocess_batch(const a3v_image_desc* images, int n, const int* fill_rgb, int out_size, uint8_t* tmp, int64_t tmp_stride,
                         void* dst, int64_t dst_stride, int dst_dtype, const float* mean, const float* std, void* stream);

/* One step of the token bookkeeping of MetaModel.generate (model/meta.py:456-477), one launch for the whole batch: greedy argmax of
 * logits [B, V] (fp32, row stride ld) -- or, when `sampled` is non-NULL, the externally sampled ids of the top-p branch (:457-459) --
 * then, per row: teacher forcing of prompt positions (text_mask[row, cur_pos], :463-465), tokens[row, cur_pos] = next,
 * stop_pos / stopped update and the multi-token stop-sequence match (:468-475; stop_seq = all sequences concatenated, stop_off =
 * n_stop + 1 offsets, tried in order).  text_mask / stopped are one byte per element (torch.bool).  `live` (optional): device
 * counter of rows still running, decremented when a row stops -- the host reads that one word every k steps instead of
 * `stopped.all()` every step (:476).  Values are bit-identical to the reference's torch ops (integer work). */
int a3v_generate_step(const float* logits, int64_t ld, const int64_t* sampled, int B, int V, int64_t* tokens, int64_t ld_tok,
                      const uint8_t* text_mask, int64_t ld_mask, int cur_pos, const int64_t* stop_seq, const int32_t* stop_off,
                      int n_stop, uint8_t* stopped, int64_t* stop_pos, int32_t* live, void* stream);

/* The sampled branch of MetaModel.generate (model/meta.py:456-459) with sample_top_p (model/meta.py:568-583) on the device, one
 * launch for the batch and no full-vocabulary sort: probs = softmax(logits / temperature); a token is kept iff the probability
 * mass ranked before it (value descending, index ascending on ties) is <= top_p; out[b] = the kept token the inverse CDF of the
 * renormalised nucleus reaches at u[b] (u[b] uniform in [0, 1), supplied by the caller's generator: torch.multinomial's own
 * random stream is not part of the contract).  logits fp32 [B, V] (V <= 65536), temperature > 0, top_p > 0 (>= 1 keeps all). */
int a3v_sample_top_p(const float* logits, int64_t ld, int B, int V, float temperature, float top_p, const float* u,
                     int64_t* out, void* stream);

/* torch.argmax(logits, -1) with first-index tie break (model/meta.py:460).  fp32 [B,V]. */
int a3v_argmax(const float* logits, int64_t ld, int64_t* out, int B, int V, void* stream);

/* CrossEntropyLoss(ignore_index=0) pieces (model/meta.py:67,256-262): per-row loss (fp32)
 * and, if dlogits != NULL, d(mean loss)/dlogits scaled by grad_scale / n_valid where
 * n_valid is read from n_valid_dev (device int32, produced by a3v_count_valid).
 * logits bf16 or f32 [rows, V]; labels int64 (already shifted). */
int a3v_count_valid(const int64_t* labels, int rows, int32_t* n_valid_dev, void* stream);
int a3v_cross_entropy(const void* logits, int64_t ld, const int64_t* labels, float* row_loss,
                      void* dlogits, int64_t ldd, const int32_t* n_valid_dev, float grad_scale,
                      int rows, int V, int dtype, void* stream);

/* One decode step (seqlen == 1) of the whole decoder stack from a single host call: per layer
 * RMSNorm -> fused QKV (skinny GEMM) -> RoPE + KV-cache write at `pos` -> split-KV attention over
 * pos+1 keys -> WO (+residual) -> RMSNorm -> W1|W3 SwiGLU -> W2 (+residual)  (LLM/llama_ens5.py:
 * 220-249, 490-531).  h [B,dim] bf16 is updated in place; xn/qkv/att/act are caller workspaces of
 * [B,dim], [B,(H+2Hkv)hd], [B,H*hd], [B,ffn]; attn_scratch as for a3v_attention at Sk = Smax;
 * skinny_ws as for a3v_gemm_skinny, sized for the largest of the four linears. */
typedef struct a3v_llama_layer {
  const void* attn_norm_w;
  const void* wqkv;     /* [wq;wk;wv] rows */
  const void* wo;
  const void* ffn_norm_w;
  const void* w13;      /* w1/w3 interleaved in 16-row blocks */
  const void* w2;
  void* k_cache;        /* [B,Hkv,Smax,hd] */
  void* vt_cache;       /* [B,Hkv,hd,Smax] */
  /* optional weight-only fp8 (OCP e4m3fn) images of the four matrices, same row order as the bf16 images, with one fp32
   * scale per row (a3v_gemm_skinny_fp8); all NULL = bf16.  Used only by the fused step form (dims % 256 == 0). */
  const void* wqkv_q; const float* wqkv_s;
  const void* wo_q;   const float* wo_s;
  const void* w13_q;  const float* w13_s;
  const void* w2_q;   const float* w2_s;
} a3v_llama_layer;
int a3v_llama_decode_step(const a3v_llama_layer* layers, int n_layers, void* h, void* xn, void* qkv,
                          void* att, void* act, float* attn_scratch, void* skinny_ws, const float* cos_sin, int B,
                          int dim, int H, int Hkv, int hd, int ffn, int Smax, int pos, float eps,
                          void* stream);
/* Which form a3v_llama_decode_step takes for a geometry, decided before anything is launched: 2 = the fused five-launch form,
 * 1 = the per-kernel form (<= 16 rows, bf16 weights), 0 = not taken (the call returns A3V_ERR_SHAPE with every buffer untouched and
 * the host runs the general kernels: llama_ens5.py:490-531 has no batch limit below max_batch_size). */
int a3v_llama_decode_step_form(int B, int dim, int H, int Hkv, int hd, int ffn, int w8);

/* ---------------------------------------------------------------------------------------
 * Training (backward) entry points.  Reference: autograd through the same modules under
 * autocast(bf16) with fp32 master weights (engine_finetune.py:44-68, main_finetune.py:212-217).
 * Convention: activations and their grads use `act_dtype` (bf16 / f32 parity); the residual
 * stream h, master weights and weight grads are fp32.  GEMM-shaped backward work goes through
 * a3v_gemm_nt on transposed images built with a3v_transpose.
 * ------------------------------------------------------------------------------------- */

/* a3v_attention + per-row log-sum-exp lse [B,H,Sq] (fp32) kept for the backward pass. */
int a3v_attention_lse(const void* q, const void* k, const void* vt, void* out, float* lse, int B,
                      int Sq, int Sk, int H, int Hkv, int hd, const int64_t* strides, int causal,
                      int dtype, void* stream);

/* dst[b][c][r] = src[b][r][c] for r < R (zero for R <= r < Rpad); leading dims / batch strides
 * in elements.  Builds W^T for dgrad and X^T / dY^T for wgrad. */
int a3v_transpose(const void* src, int64_t ld_src, int64_t bs_src, void* dst, int64_t ld_dst,
                  int64_t bs_dst, int R, int C, int Rpad, int batch, int dtype, void* stream);

/* backward of a3v_rmsnorm with fp32 x, w: dh += d/dx ; dw += d/dw (may be NULL). dy in act_dtype.
 * dw_scratch (may be NULL): a3v_rmsnorm_bwd_scratch_floats(rows, dim) floats; with it the per-block weight-gradient
 * partials are written as rows and column-summed by a second small kernel instead of ~rows/8 x dim atomics on dw. */
int64_t a3v_rmsnorm_bwd_scratch_floats(int rows, int dim);
int a3v_rmsnorm_bwd(const float* x, int64_t ldx, const float* w, const void* dy, int64_t lddy,
                    float* dh, int64_t lddh, float* dw, float* dw_scratch, int rows, int dim, float eps,
                    int act_dtype, void* stream);
/* The same, and the updated dh also as bf16 (dh_bf16 [rows][ld_bf16]): the operand of the weight / input gradient GEMMs that
 * follow (autograd casts the fp32 stream gradient for the autocast linears, engine_finetune.py:49) without a pass of its own. */
int a3v_rmsnorm_bwd_cast(const float* x, int64_t ldx, const float* w, const void* dy, int64_t lddy,
                         float* dh, int64_t lddh, float* dw, float* dw_scratch, int rows, int dim, float eps,
                         int act_dtype, void* dh_bf16, int64_t ld_bf16, void* stream);

/* bf16 residual stream: under FSDP MixedPrecision(param_dtype = bf16) + autocast (main_finetune.py:241-263, engine_finetune.py:44-50)
 * the reference's embeddings, block outputs and hence the residual stream AND ITS GRADIENT are bf16 tensors; these forms take the
 * stream (x) and the accumulated stream gradient (dh, read-modify-write, rounded to bf16 as autograd rounds it at the residual add) in
 * bf16, with fp32 arithmetic inside.  dy bf16.  Same formulas as a3v_rmsnorm_bwd / a3v_layernorm_bwd / a3v_embed_bwd. */
int a3v_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const float* w, const void* dy, int64_t lddy, void* dh, int64_t lddh,
                         float* dw, float* dw_scratch, int rows, int dim, float eps, void* stream);
int a3v_layernorm_bwd_bf16(const void* x, int64_t ldx, const float* w, const void* dy, int64_t lddy, const int32_t* row_map,
                           void* dx, int64_t lddx, float* dw, float* db, int rows, int dim, float eps, void* stream);
int a3v_embed_bwd_bf16(const int64_t* tokens, int64_t ld_tok, const void* dh, float* dtable, int B, int T, int W, int dim,
                       int vocab, void* stream);

/* backward of a3v_layernorm (x in act_dtype, fp32 w; dy fp32 rows gathered through row_map):
 * dx (act_dtype), dw += , db += . */
int a3v_layernorm_bwd(const void* x, int64_t ldx, const float* w, const float* dy, int64_t lddy,
                      const int32_t* row_map, void* dx, int64_t lddx, float* dw, float* db, int rows,
                      int dim, float eps, int act_dtype, void* stream);

/* SwiGLU on gu [rows, 2F] (the un-fused form of A3V_EPI_SWIGLU, kept for the backward pass):
 * interleaved = 1: 16-column blocks [gate|up] (inference weight image); 0: [all gates | all ups]
 * (training image = cat(w1, w3), whose fp32 grad buffer splits into w1.grad / w3.grad views).
 * act = silu(g)*u ; dgu from dact. */
int a3v_swiglu_fwd(const void* gu, int64_t ldg, void* act, int64_t lda, int rows, int F,
                   int interleaved, int dtype, void* stream);
int a3v_swiglu_bwd(const void* gu, int64_t ldg, const void* dact, int64_t lda, void* dgu,
                   int64_t lddg, int rows, int F, int interleaved, int dtype, void* stream);

/* inverse RoPE on dq [B,S,H,hd], dk [B,Hkv,S,hd]; dv [B,Hkv,S,hd] copied; packed into
 * dqkv [B*S, (H+2Hkv)*hd] (the gradient of the fused QKV activation). */
int a3v_rope_bwd_pack(const void* dq, const void* dk, const void* dv, void* dqkv, int64_t ld,
                      const float* cos_sin, int B, int S, int H, int Hkv, int hd, int rope_pos0,
                      int dtype, void* stream);

/* backward of causal / full self-attention (Sq == Sk == S).  q,out,dout,dq [B,S,H,hd]; k,dk,dv
 * [B,Hkv,S,hd]; k rows [b][hk][s] at k + b*k_sb + hk*k_sh + s*hd; v rows addressed by element
 * strides (batch, seq, kv-head); lse [B,H,S]; D: fp32 scratch [B,S,H]. */
int a3v_attention_bwd(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v,
                      int64_t v_sb, int64_t v_ss, int64_t v_sh, const void* out, const void* dout,
                      const float* lse, float* D, void* dq, void* dk, void* dv, void* workspace,
                      int B, int S, int H, int Hkv, int hd, int causal, int dtype, void* stream);
/* a3v_attention_bwd + a3v_rope_bwd_pack in one pass (bf16, hd 64 / 128): the gradient of the fused qkv activation
 * dqkv [B*S, ld_qkv] = [dq | dk | dv] with the q / k parts rotated back by -theta (autograd of apply_rotary_emb and of the
 * xq / xk / xv views, LLM/llama_ens5.py:112-118); no dq / dk / dv buffers.  out [B*S, ld_out] (token stride ld_out >= H*hd, 16-B
 * aligned), dout contiguous.  D: scratch [B, S, H] floats. */
int a3v_attention_bwd_packed(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb,
                             int64_t v_ss, int64_t v_sh, const void* out, int64_t ld_out, const void* dout, const float* lse, float* D,
                             void* dqkv, int64_t ld_qkv, const float* cos_sin, int rope_pos0, int B, int S, int H, int Hkv,
                             int hd, int causal, int dtype, void* stream);
/* bytes of `workspace` for the bf16 MFMA path: a non-NULL workspace selects it (NULL = the generic, slow kernels).  The MFMA
 * kernels no longer keep transposed K / Q / dO images there (transpose reads on the row tiles), so this is a token 256 bytes. */
int64_t a3v_attention_bwd_workspace_bytes(int B, int S, int H, int Hkv, int hd);
int a3v_attention_bwd_mfma(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v,
                           int64_t v_sb, int64_t v_ss, int64_t v_sh, const void* dout, const float* lse,
                           const float* D, void* dq, void* dk, void* dv, void* workspace, int B, int S,
                           int H, int Hkv, int hd, int causal, void* stream);

/* dst = (dst_dtype) src, 2-D with leading dimensions (fp32 grad stream -> bf16 GEMM operand). */
/* dst[r, c] += src[r, c] (fp32 or bf16, cols % 4 == 0): accumulates the diagonal blocks of the fused LoRA gradient GEMMs
 * into lora_a.weight.grad / lora_b.weight.grad (model/peft.py adapters) and performs the block's residual add
 * `h + out` (LLM/llama_ens5.py:238,241) when the adapter sum must be rounded before it (peft.py:95). */
int a3v_add2d(void* dst, int64_t ld_dst, const void* src, int64_t ld_src, int rows, int cols, int dtype, void* stream);
int a3v_cast(const void* src, int64_t ld_src, int src_dtype, void* dst, int64_t ld_dst, int dst_dtype,
             int rows, int cols, void* stream);

/* d_table[token] += dh[row] over the text rows of [BOS | W image words | text] (fp32). */
int a3v_embed_bwd(const int64_t* tokens, int64_t ld_tok, const float* dh, float* dtable, int B, int T,
                  int W, int dim, int vocab, void* stream);

/* out[c] += sum_i src[row_idx ? row_idx[i] : i][c]  (bias grads, start_img / end_img grads). */
int a3v_rows_sum(const void* src, int64_t ld, const int32_t* row_idx, int n_rows, int dim, float* out,
                 int dtype, void* stream);

/* One AdamW step on one fp32 parameter tensor (torch.optim.AdamW semantics, decoupled weight decay; the reference optimizer of
 * main_finetune.py:138 with engine_finetune.py:63 optimizer.step()): step = 1-based update count of this parameter.  All four
 * arrays 16-B aligned.  bf16_image (optional): also store the updated parameter rounded to bf16 (the next step's GEMM operand). */
int a3v_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
              float eps, float weight_decay, int64_t step, void* bf16_image, void* stream);

/* a3v_adamw with the gradient multiplied by *grad_scale (a DEVICE fp32 scalar, NULL = 1) as it is read: the clip of
 * NativeScalerWithGradNormCount.__call__ (util/misc.py:302-315) / clip_grad_norm (util/clip_grad.py:187-193: coef =
 * max_norm / (norm + 1e-6) clamped to 1, every gradient multiplied by it) folded into the optimizer pass -- same values as
 * grad.mul_(coef) followed by a3v_adamw, without the extra read + write of every gradient and without a host read of the norm. */
/* A NEGATIVE or NaN *grad_scale makes the call a no-op: the trainer folds "loss and gradient norm of this step are finite" into
 * the coefficient on the device, so a bad step never reaches the fp32 masters, the moments or the bf16 weight images (the
 * reference stops before backward on a non-finite loss, engine_finetune.py:56-58; here the host reads the flag later). */
int a3v_adamw_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int64_t step, void* bf16_image, const float* grad_scale, void* stream);

/* Adapter weight gradients of a fused LoRA group (model/peft.py:58-159: lora_b of the n_mods <= 4 linears that share an input):
 * gbt = t^T . dy, fp32 [>= n_mods*r, N] with row stride ld, holds the group's dB^T; module j owns rows j*r..(j+1)*r of columns
 * row0[j]..row0[j]+nj[j]; dst[j] (fp32 [nj[j], r], contiguous: the parameter's gradient) += that block transposed.  dst / row0 / nj
 * are HOST arrays (dst[j] device pointers). */
int a3v_lora_gb_scatter(const float* gbt, int64_t ld, int r, int n_mods, float* const* dst, const int* row0, const int* nj, void* stream);

/* The adapter weight gradients as a streaming kernel: partial[s][R][N] = T[k in slice s][0..R)^T . X[k in slice s][0..N), S token
 * slices (raw fp32 planes, summed / rounded / accumulated by a3v_splitk_reduce exactly like those of a3v_gemm_tn_splitk).  T: bf16
 * [Kt][ldt] with R <= 64 valid columns (the padded rank of a fused adapter group; the kernel always loads 64 columns per row, so
 * ldt >= 64 and the row's first 64 elements must be readable), X: bf16 [Kt][ldx].  dB^T = t^T . dy and dA = dt^T . x of
 * model/peft.py:58-159 (autograd's gradients of lora_b / lora_a, engine_finetune.py:55-57).  Small blocks (64 x 128 tile, 48 KiB of
 * LDS: three per CU) so that the stream of X has several stages in flight per CU. */
int a3v_gemm_tn_strip(const void* T, int64_t ldt, const void* X, int64_t ldx, float* partial, int R, int N, int Kt, int S, void* stream);

/* a3v_adamw_scaled for a [rows, cols] matrix (rows, cols multiples of 64) that ALSO keeps the transposed bf16 image current:
 * bf16_image_t points at element [0][first row of this parameter] of W^T [cols][ldt] (ldt >= rows, multiple of 4; 8-B aligned), the
 * operand of the full fine-tune's input-gradient GEMMs on the NT kernel (loss.backward() through F.linear, engine_finetune.py:55-57;
 * optimizer.step(), :63).  bf16_image (optional) as in a3v_adamw.  Arithmetic identical to a3v_adamw_scaled. */
int a3v_adamw_scaled_t(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int rows, int cols, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int64_t step, void* bf16_image, void* bf16_image_t, int64_t ldt,
                       const float* grad_scale, void* stream);

/* The same update for MANY small tensors of one torch param group in ONE launch (a LoRA step: ~520 adapter / norm tensors; was one
 * launch per tensor plus one a3v_lora_refresh per adapter).  `table` is a DEVICE array of n_tensors descriptors (built once by the
 * host, a3vlm_amd/optim.py); every pointer 16-byte aligned, fp32 contiguous p / g / m / v of n elements viewed as [n / cols, cols].
 * The bf16 value of updated element (i, j) is also stored to d1[i*s1r + j*s1c] and d2[i*s2r + j*s2c] when those are non-NULL
 * (element strides): a same-shape image (s1r = cols, s1c = 1), or an adapter's rows / columns inside the fused group images and
 * their transposes (model/peft.py:40-64 parameters; reference optimizer: main_finetune.py:138).  max_n = the largest n.
 * Arithmetic identical to a3v_adamw_scaled. */
typedef struct {
  float* p; const float* g; float* m; float* v;
  int64_t n, cols;
  void* d1; int64_t s1r, s1c;
  void* d2; int64_t s2r, s2c;
} a3v_adamw_tensor;
int a3v_adamw_multi(const a3v_adamw_tensor* table, int n_tensors, int64_t max_n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, const float* grad_scale, void* stream);

/* dst[i] = (dst_dtype)(src[i] * scale), n contiguous elements, both 16-B aligned: the wire-dtype conversions of the DP gradient
 * reducer (fp32 bucket -> bf16 wire bucket pre-scaled by 1/world, and back) -- FSDP's reduce_dtype = bf16 averaging
 * (main_finetune.py:251-255) without separate scale / cast / copy passes over the gradient buffer. */
int a3v_scale_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, float scale, void* stream);

/* The exchange step of data-parallel fine-tuning for a host that owns an RCCL communicator itself (the
 * reference reaches NCCL through FSDP's gradient reduction, main_finetune.py:241-263, reduce_dtype :251-255; the Python host of this
 * repository goes through torch.distributed, a3vlm_amd/dp.py).  `grad`: one bucket (n fp32 elements, 16-byte aligned) of the flat
 * gradient buffer, reduced IN PLACE over the ranks of `comm` (an ncclComm_t) on `stream`: average != 0 -> ncclAvg (FSDP's
 * gradient average), else ncclSum.  `wire_bf16` (optional, n bf16 elements): the bucket crosses the wire in bf16 -- one fused cast
 * in, all-reduce of the bf16 buffer, one widening cast back.  Skipping the call on accumulation micro-steps (util/misc.py:311-313)
 * is the caller's decision.  RCCL is not linked into the library: ncclAllReduce is resolved at run time from the librccl already
 * in the process (PyTorch-ROCm's), A3V_RCCL_LIB, or the loader path; A3V_ERR_ARG when none is found. */
int a3v_grad_bucket_allreduce(void* comm, float* grad, int64_t n, void* wire_bf16, int average, void* stream);

/* ZeRO-1 through the same boundary (round 4; the reference trains configs[3] under FSDP SHARD_GRAD_OP, main_finetune.py:241-263:
 * gradients and AdamW state sharded over DP; the Python host drives these two collectives through torch.distributed,
 * a3vlm_amd/zero1.py).  a3v_grad_bucket_reduce_scatter: `grad` = a bucket's sharded span, world * n_per_rank fp32 elements (world =
 * ncclCommCount(comm)); rank r receives the average (or sum) of elements [r n_per_rank, (r + 1) n_per_rank) in `shard` (fp32).
 * wire_bf16 (world * n_per_rank bf16) + wire_shard_bf16 (n_per_rank bf16), both or neither: the span crosses the wire in bf16.
 * a3v_param_shard_all_gather: every rank's updated bf16 parameter slice (n_per_rank elements) gathered into the flat bf16 parameter
 * buffer (world * n_per_rank elements; in place when shard_bf16 = flat_bf16 + rank * n_per_rank).  a3v_rccl_comm_count: the
 * communicator's size (-1 without RCCL).  Same run-time RCCL resolution as a3v_grad_bucket_allreduce. */
int a3v_grad_bucket_reduce_scatter(void* comm, const float* grad, int64_t n_per_rank, float* shard, void* wire_bf16, void* wire_shard_bf16,
                                   int average, void* stream);
int a3v_param_shard_all_gather(void* comm, const void* shard_bf16, void* flat_bf16, int64_t n_per_rank, void* stream);
int a3v_rccl_comm_count(void* comm);
int a3v_rccl_available(void);

/* Partial sums of squares of an fp32 range: out[0 .. A3V_SUMSQ_SLOTS) (every slot written).  The global-norm gradient clip
 * (reference util/clip_grad.py:59-210) = sqrt(sum of the partials of all gradient buckets); called per bucket on a side stream
 * while the backward still runs.  x 16-byte aligned. */
#define A3V_SUMSQ_SLOTS 1024
int a3v_sumsq_partials(const float* x, int64_t n, float* out, void* stream);

/* Re-write one adapter's rows / columns of a fused LoRA group's bf16 images from its fp32 parameters (model/peft.py:40-64
 * lora_a [r, in], lora_b [nj, r]): A[col0+i, :] and At[:, col0+i] from lora_a, B[row0+n, col0+i] and Bt[col0+i, row0+n] from lora_b. */
int a3v_lora_refresh(const float* lora_a, const float* lora_b, int r, int in_f, int nj, void* A, int64_t lda, void* At, int64_t ldat,
                     void* B, int64_t ldb, void* Bt, int64_t ldbt, int col0, int row0, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* A3VLM_HIP_H */
