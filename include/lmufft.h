/*
 * lmufft.h -- C ABI of the MI355X (gfx950) implementation of the parallel
 * Legendre Memory Unit layer ("LMUFFT").
 *
 * The library (asr_amd/csrc/liblmufft_hip.so) replaces the arithmetic of the
 * reference module `LMUFFT` (reference: src/lmu.py:208-320).  The reference is
 * pure Python and has no FFI of its own for this path; the Python side of this
 * repository (asr_amd/lmu.py) binds these symbols with ctypes and keeps the
 * reference module's constructor / forward / state_dict surface.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) owned by the caller; the library
 *     neither frees nor retains it beyond the call;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     all work is enqueued on it, nothing synchronises the device;
 *   - tensors are dense row-major: x[B][T][D], h[B][T][Hd], m[B][T][M],
 *     u[B][T], W_h[Hd][M+D] (memory columns first, then input columns),
 *     w_u[D], H[M][ldH] with ldH >= T (the module's seq_len);
 *   - return value 0 = success; non-zero = error, text via lmufft_last_error();
 *   - no function allocates device memory: scratch comes in through
 *     `workspace` (size from lmufft_*_workspace_bytes).
 */
#ifndef LMUFFT_H_
#define LMUFFT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMUFFT_ABI_VERSION 31

/* element types of the activations (x, h, gh, gx, m) */
#define LMUFFT_F32 0
#define LMUFFT_BF16 1

/* causal-convolution algorithms */
#define LMUFFT_CONV_AUTO 0
#define LMUFFT_CONV_TOEPLITZ 1 /* blocked Toeplitz contraction on MFMA */
#define LMUFFT_CONV_FFT 2      /* LDS-resident radix FFT */
#define LMUFFT_CONV_SCAN 3     /* bf16 only: state-space chunks, two dense GEMMs (long sequences) */

int lmufft_abi_version(void);

/* Text of the last error raised on the calling thread ("" if none). */
const char* lmufft_last_error(void);

/* Shape of one layer call. */
typedef struct lmufft_shape {
  int32_t B;   /* batch                                   */
  int32_t T;   /* frames in this call (<= ldH)            */
  int32_t D;   /* input_size                              */
  int32_t M;   /* memory_size                             */
  int32_t Hd;  /* hidden_size                             */
  int32_t ldH; /* row stride of H = the module's seq_len  */
} lmufft_shape;

/* ------------------------------------------------------------------------
 * Forward of the whole layer.  Replaces LMUFFT.forward, src/lmu.py:290-320:
 *   u = relu(x . w_u + b_u)                       (src/lmu.py:300)
 *   m[b,t,k] = sum_{s<=t} H[k,t-s] u[b,s]         (src/lmu.py:303-312)
 *   h = relu([m;x] W_h^T + b_h), h_n = h[:,T-1,:] (src/lmu.py:315-318)
 * Outputs u [B][T] and m [B][T][M] are what the backward needs; h [B][T][Hd],
 * h_n [B][Hd].  All f32.
 * ---------------------------------------------------------------------- */
size_t lmufft_forward_workspace_bytes(const lmufft_shape* s, int dtype);

int lmufft_forward_f32(const lmufft_shape* s, const float* x, const float* w_u,
                       const float* b_u, const float* W_h, const float* b_h,
                       const float* H, const void* fftprep, float* u, float* m,
                       float* h, float* h_n, int conv_algo, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Backward of the whole layer (what autograd derives from src/lmu.py:300-318):
 *   g = gh (+ gh_n on the last frame); gz = g*(h>0);
 *   gW_h = gz^T [m;x]; gb_h = sum gz; [gm;gx1] = gz W_h;
 *   gu[b,s] = sum_{t>=s} sum_k H[k,t-s] gm[b,t,k]; ga = gu*(u>0);
 *   gw_u = sum ga x; gb_u = sum ga; gx = gx1 + ga w_u.
 * gh and gh_n may each be NULL (treated as zero).  gx may be NULL when the
 * input needs no gradient.  gm [B][T][M] is caller-provided scratch/output.
 * ---------------------------------------------------------------------- */
size_t lmufft_backward_workspace_bytes(const lmufft_shape* s, int dtype);

int lmufft_backward_f32(const lmufft_shape* s, const float* x, const float* w_u,
                        const float* W_h, const float* H, const void* fftprep,
                        const float* u, const float* m, const float* h, const float* gh,
                        const float* gh_n, float* gx, float* gw_u, float* gb_u,
                        float* gW_h, float* gb_h, float* gm, int conv_algo,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * The causal convolution on its own (eq. 26, src/lmu.py:303-312) and its
 * adjoint, exposed for kernel-level parity and roofline measurement.
 *   forward : u [B][T], H [M][ldH]  ->  m [B][T][M]
 *   backward: gm [B][T][M]          ->  gu [B][T]
 * Only B, T, M, ldH of the shape are read.  Both directions take a workspace
 * of lmufft_conv_workspace_bytes (the FFT route keeps spectra there; the
 * Toeplitz forward uses none).
 * ---------------------------------------------------------------------- */
size_t lmufft_conv_workspace_bytes(const lmufft_shape* s, int dtype, int conv_algo);

int lmufft_conv_forward_f32(const lmufft_shape* s, const float* u, const float* H,
                            const void* fftprep, float* m, int conv_algo,
                            void* workspace, size_t workspace_bytes, void* stream);

int lmufft_conv_backward_f32(const lmufft_shape* s, const float* gm, const float* H,
                             const void* fftprep, float* gu, int conv_algo,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Allocation-free form of lmufft_conv_backward_f32.  It has no counterpart in
 * the reference; it exists for callers that cannot allocate device memory or
 * carry a workspace: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Same arguments and result as
 * lmufft_conv_backward_f32 minus the workspace: the partial slabs / partial
 * spectra live in a device array of LMUFFT_NOALLOC_SCRATCH_BYTES that is part
 * of the library's code object (every GPU the library runs on holds its own
 * copy until the process ends).  The call allocates and frees nothing.
 *   - shapes whose lmufft_conv_workspace_bytes exceeds
 *     LMUFFT_NOALLOC_SCRATCH_BYTES are rejected with an error;
 *   - all calls on one GPU share that GPU's array: calls that overlap in time
 *     on one GPU (different streams, or threads) race -- serialise them;
 *   - the device that is current at the call must be the stream's device; the
 *     first call on a device looks the array's address up (make that call
 *     outside graph capture).
 * ---------------------------------------------------------------------- */
#define LMUFFT_NOALLOC_SCRATCH_BYTES (8u << 20)
int lmufft_conv_backward_noalloc_f32(const lmufft_shape* s, const float* gm, const float* H,
                                     const void* fftprep, float* gu, int conv_algo, void* stream);

/* ------------------------------------------------------------------------
 * Spectra of the impulse response for the FFT route (the counterpart of the
 * reference's `fft_H` buffer, src/lmu.py:285, in the layout and transform
 * length the kernels use).  They depend on H and on the call's T (the filter
 * is truncated to the first T taps), so a caller prepares them once per
 * (H, T): allocate lmufft_fftprep_bytes(s) (0 = this T is not supported by the
 * FFT kernels), call lmufft_fftprep_f32, pass the buffer as `fftprep` above.
 * fftprep == NULL selects the Toeplitz kernels (conv_algo AUTO) or is an error
 * (conv_algo FFT).
 * ---------------------------------------------------------------------- */
size_t lmufft_fftprep_bytes(const lmufft_shape* s);
int lmufft_fftprep_f32(const lmufft_shape* s, const float* H, void* fftprep, void* stream);

/* ------------------------------------------------------------------------
 * bf16 variant: activations x, m, h, h_n, gh, gh_n, gx, gm are bf16 (passed as
 * void*), every sum is accumulated in fp32; parameters, their gradients, u and
 * H stay fp32.  The reference has no bf16 GPU path for these lines (torch.fft
 * rejects bf16); semantics = src/lmu.py:297-320 on bf16-rounded operands.
 *
 * `hprep` holds bf16 copies of H in both orientations.  H is a constant
 * buffer of the module (src/lmu.py:243-245), so it is prepared once:
 * allocate lmufft_hprep_bytes(s) (M and ldH are read), call lmufft_hprep_bf16,
 * keep the buffer for every later forward/backward with that H.
 *
 * `wprep` holds bf16 copies of W_h in both orientations (the forward reads
 * W_h[n][k], the data gradient its transpose).  W_h changes every optimiser
 * step, so it is prepared once per step: lmufft_wprep_bf16 before the
 * forward, the same buffer handed to the backward.
 * ---------------------------------------------------------------------- */
size_t lmufft_hprep_bytes(const lmufft_shape* s);
int lmufft_hprep_bf16(const lmufft_shape* s, const float* H, void* hprep, void* stream);
size_t lmufft_wprep_bytes(const lmufft_shape* s);
int lmufft_wprep_bf16(const lmufft_shape* s, const float* W_h, void* wprep, void* stream);

/* W_h: NULL, or the fp32 weights [Hd][M+D] -- the call then rebuilds `wprep` from them itself,
 * in the same launch as the encoder kernel (one launch fewer per step than a separate
 * lmufft_wprep_bf16 call; `wprep` is written in that case). */
/* scanprep: NULL, or the operands of the state-space convolution route (below).  conv_algo AUTO takes
 * that route whenever scanprep is given and the Toeplitz kernels otherwise; SCAN requires it.  The
 * workspace (lmufft_forward_workspace_bytes / lmufft_backward_workspace_bytes) is only read on that route. */
/* hmask: NULL, or [B*T][Hd/8] bytes (Hd a multiple of 8) that receive the relu pattern of h, bit j of byte
 * (row, c) = h[row][8 c + j] > 0.  Handed to the backward it replaces the read of h there (1/16 of the bytes). */
int lmufft_forward_bf16(const lmufft_shape* s, const void* x, const float* w_u,
                        const float* b_u, void* wprep, const float* W_h, const float* b_h,
                        const void* hprep, const void* scanprep, float* u, void* m, void* h,
                        void* h_n, void* hmask, int conv_algo, void* workspace, size_t workspace_bytes,
                        void* stream);

/* The bf16 forward followed by the layer norm the encoder applies to the layer's output
 * (`x = layer_norm(x)`, src/models/lmu_encoder.py:274; dropout is the identity in eval mode), with
 * the normalisation done in the projection kernel's epilogue: y = LN(h) * gamma + beta (fp32
 * statistics over the bf16 h, biased variance), y [B][T][Hd] f32 or bf16 (y_dtype).
 * `h` may be NULL when only y and h_n are wanted (inference).  Needs Hd <= 384 and M, D, Hd
 * multiples of 8; otherwise it fails and the caller runs lmufft_add_layernorm after the layer. */
int lmufft_forward_ln_bf16(const lmufft_shape* s, const void* x, const float* w_u,
                           const float* b_u, void* wprep, const float* W_h, const float* b_h,
                           const void* hprep, const void* scanprep, float* u, void* m, void* h,
                           void* h_n, const float* gamma, const float* beta, float eps, void* y,
                           int y_dtype, int conv_algo, void* workspace,
                           size_t workspace_bytes, void* stream);

/* The bf16 causal convolution (eq. 26, src/lmu.py:303-312) and its adjoint on their own, for
 * kernel-level parity and roofline measurement: u [B][T] f32 -> m [B][T][M] bf16; gm [B][T][M]
 * bf16 -> gu [B][T] f32 (unmasked).  Only B, T, M, ldH of the shape are read.  The backward takes
 * a workspace of lmufft_conv_workspace_bytes(s, LMUFFT_BF16, LMUFFT_CONV_AUTO). */
int lmufft_conv_forward_bf16(const lmufft_shape* s, const float* u, const void* hprep, void* m,
                             void* stream);
int lmufft_conv_backward_bf16(const lmufft_shape* s, const void* gm, const void* hprep, float* gu,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * State-space ("scan") route of the bf16 causal convolution, for long sequences.  Same result as
 * src/lmu.py:303-312 -- the reference's H is the impulse response H[:,t] = A^t B of
 * x_t = A x_{t-1} + B u_t (src/lmu.py:275-283) -- evaluated in chunks of lmufft_scan_chunk() frames:
 * a chunk needs from all earlier frames only the state at its start, so the convolution becomes one
 * dense GEMM of [chunk state | chunk samples] rows against a constant matrix (powers of A, the
 * chunk's own Toeplitz block) and a short recurrence over chunks; the adjoint is the transposed
 * GEMM and the reverse recurrence.  Work per frame is M (M + chunk) multiply-adds instead of M T / 2.
 * `scanprep` holds those constants.  They depend on the module's buffers A [M][M] and B [M]
 * (src/lmu.py:240-241) only: allocate lmufft_scanprep_bytes(M) bytes of HOST memory (0 = this
 * memory size is not supported: multiples of 8 up to 256), fill them with
 * lmufft_scanprep_build_host (host pointers, double-precision powers; about a second for M = 192),
 * copy the block to 16-byte aligned device memory once and pass that pointer as `scanprep`.
 * The conv-only entry points mirror lmufft_conv_forward_bf16 / lmufft_conv_backward_bf16; both take
 * a workspace of lmufft_conv_scan_workspace_bytes(s) (B, T, M are read).
 * ---------------------------------------------------------------------- */
int32_t lmufft_scan_chunk(void);
size_t lmufft_scanprep_bytes(int32_t M);
int lmufft_scanprep_build_host(const float* A_host, const float* B_host, int32_t M, void* scanprep_host);
size_t lmufft_conv_scan_workspace_bytes(const lmufft_shape* s);
int lmufft_conv_forward_scan_bf16(const lmufft_shape* s, const float* u, const void* scanprep, void* m,
                                  void* workspace, size_t workspace_bytes, void* stream);
int lmufft_conv_backward_scan_bf16(const lmufft_shape* s, const void* gm, const void* scanprep, float* gu,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* gm: [B][T][M] bf16 scratch that receives the memory gradient gz W_h[:, :M], or NULL.  With gm == NULL the
 * encoder gradient is taken straight from the masked upstream gradient: the two linear steps of autograd of
 * src/lmu.py:303-316 (gm = gz W_h[:, :M], then the adjoint convolution with H) are composed into one GEMM against
 * Q^T, Q[n][tau] = sum_k W_h[n][k] H[k][tau], built once per call, and gm is never formed.  That route takes
 * sequences of up to 512 frames, a memory size that is a multiple of 8, a hidden size that is a multiple of 32 and
 * 16-byte aligned h / gh, and is
 * not used with the state-space convolution; lmufft_backward_needs_gm_bf16 tells (1: pass a gm buffer, 0: gm may
 * be NULL) for a shape, conv_algo and whether scanprep is given. */
int lmufft_backward_needs_gm_bf16(const lmufft_shape* s, int conv_algo, int has_scanprep);
/* hmask: NULL, or the relu pattern the forward wrote (see lmufft_forward_bf16); h must still be given. */
int lmufft_backward_bf16(const lmufft_shape* s, const void* x, const float* w_u,
                         const void* wprep, const void* hprep, const void* scanprep, const float* u,
                         const void* m, const void* h, const void* hmask, const void* gh,
                         const void* gh_n, void* gx, float* gw_u, float* gb_u,
                         float* gW_h, float* gb_h, void* gm, int conv_algo,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The bf16 backward in two calls, for callers that run the parameter gradients on another stream than the
 * rest (they depend on nothing after the input gradient, so they can run beside whatever consumes gx):
 *   lmufft_backward_data_bf16    everything of lmufft_backward_bf16 up to gx and gm; leaves the masked
 *                                upstream gradient and the encoder-activation gradient in `workspace`;
 *   lmufft_backward_weights_bf16 gw_u, gb_u, gW_h, gb_h from x, m and that SAME workspace, untouched in
 *                                between; the caller orders it after the data call (an event, if it
 *                                runs on another stream) and keeps x, m and the workspace alive until
 *                                it has finished.
 * lmufft_backward_bf16 is the two back to back on one stream. */
int lmufft_backward_data_bf16(const lmufft_shape* s, const void* x, const float* w_u,
                              const void* wprep, const void* hprep, const void* scanprep,
                              const float* u, const void* m, const void* h, const void* hmask,
                              const void* gh, const void* gh_n, void* gx, void* gm, int conv_algo, void* workspace,
                              size_t workspace_bytes, void* stream);
int lmufft_backward_weights_bf16(const lmufft_shape* s, const void* x, const void* m, float* gw_u,
                                 float* gb_u, float* gW_h, float* gb_h, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* lmufft_backward_bf16 with the last step left out: the weight-gradient kernel's partial slabs stay in the workspace
 * (gx is complete), and ONE later launch sums them for every layer of the step -- three launches less per step of the
 * reference's three-block encoder.  `items` is a HOST array; shape and workspace as given to the partial call. */
int lmufft_backward_partial_bf16(const lmufft_shape* s, const void* x, const float* w_u, const void* wprep,
                                 const void* hprep, const void* scanprep, const float* u, const void* m,
                                 const void* h, const void* hmask, const void* gh, const void* gh_n, void* gx,
                                 void* gm, int conv_algo, void* workspace, size_t workspace_bytes, void* stream);
/* lmufft_backward_bf16 with options (a bit set of LMUFFT_BWD_*; 0 = lmufft_backward_bf16).  The two masking options
 * fold the ReLU backward that sits between two directly stacked layers into the upper layer's input-gradient kernel
 * (no counterpart in the reference, where autograd runs that ReLU backward as a pass of its own, src/lmu.py:318):
 *   LMUFFT_BWD_MASK_GX     this layer's input x is the output of a ReLU whose backward is the next thing to happen to
 *                          gx: gx is returned already multiplied by (x > 0); `extras->x_mask`, if given, holds
 *                          that comparison as bits, [B*T][D/8] bytes in the layout of `hmask` (what the forward
 *                          of the layer that produced x wrote: 1/16 of the bytes of x), else x itself is read;
 *   LMUFFT_BWD_GH_MASKED   the caller vouches that gh is already zero wherever h <= 0 (the layer above was run with
 *                          LMUFFT_BWD_MASK_GX on this layer's h): gh is used as the masked gradient in place and the
 *                          masking pass is not run.  Needs gh_n == NULL.
 *   LMUFFT_BWD_PARTIAL_SUMS  as lmufft_backward_partial_bf16 (the four parameter-gradient pointers are ignored). */
/* `extras` (NULL = none of it).  The backward of sequences up to 512 frames multiplies the masked gradient by a
 * matrix Q^T that depends on the layer's weights only (lmufft_qt_bytes(shape) bytes; 0 = this shape does not take that
 * route).  It makes Q^T itself, on spare workgroups of its masking pass -- but a layer run with LMUFFT_BWD_GH_MASKED
 * has no such pass.  So Q^T can be made elsewhere: by lmufft_qprep_many_bf16 (one launch for several layers), or on
 * the masking pass of ANOTHER layer's backward (`q_items` of that call: up to three ride for free, more get a launch
 * of their own), any time after the forward that filled the item's `wprep`; the layer's own backward is then given
 * the finished matrix as `qt`. */
typedef struct lmufft_q_item {
  lmufft_shape shape;
  const void* hprep;
  const void* wprep;
  void* qt;
} lmufft_q_item;
typedef struct lmufft_bwd_extras {
  const void* x_mask;
  const void* qt;
  const lmufft_q_item* q_items; /* HOST array */
  int32_t q_count;
} lmufft_bwd_extras;
size_t lmufft_qt_bytes(const lmufft_shape* s);
int lmufft_qprep_many_bf16(const lmufft_q_item* items, int32_t count, void* stream);
#define LMUFFT_BWD_GH_MASKED 1u
#define LMUFFT_BWD_MASK_GX 2u
#define LMUFFT_BWD_PARTIAL_SUMS 4u
int lmufft_backward_flags_bf16(const lmufft_shape* s, const void* x, const float* w_u, const void* wprep,
                               const void* hprep, const void* scanprep, const float* u, const void* m, const void* h,
                               const void* hmask, const void* gh, const void* gh_n, void* gx, float* gw_u,
                               float* gb_u, float* gW_h, float* gb_h, void* gm, int conv_algo, void* workspace,
                               size_t workspace_bytes, void* stream, uint32_t flags,
                               const lmufft_bwd_extras* extras);
typedef struct lmufft_lmu_sum_item {
  lmufft_shape shape;
  float* gw_u;
  float* gb_u;
  float* gW_h;
  float* gb_h;
  const void* workspace;
  size_t workspace_bytes;
} lmufft_lmu_sum_item;
int lmufft_backward_weights_sum_many_bf16(const lmufft_lmu_sum_item* items, int32_t count, void* stream);

/* ------------------------------------------------------------------------
 * Log-mel frontend.  Replaces AudioPreprocessor.extract_mel_features and
 * _normalize_features, src/data/preprocessing.py:76-117, for the transform the
 * reference constructs at src/data/preprocessing.py:41-51 (16 kHz, n_fft =
 * win_length = 400, hop 160, 80 HTK mel bins, power 2, AmplitudeToDB):
 *   wave [B][ld_wave] f32 samples, utterance b holds lengths[b] <= L of them
 *   (lengths = NULL: all hold L; every utterance needs more than 200 samples,
 *   the reflect padding's limit);
 *   mel [B][Tmax][80] f32, Tmax >= lmufft_mel_frames(L); utterance b has
 *   lmufft_mel_frames(lengths[b]) = 1 + lengths[b]/160 frames, the rows after
 *   them are written as zero (the reference's collate_fn pads with zeros,
 *   src/data/preprocessing.py:351-353);
 *   normalize != 0: per utterance and mel bin (x - mean_t) / std_t, unbiased
 *   std, std == 0 -> 1 (src/data/preprocessing.py:111-117).
 * `melprep` = constants (twiddles, window, filterbank) in device memory:
 * allocate lmufft_melprep_bytes(), fill once with lmufft_melprep.
 * ---------------------------------------------------------------------- */
size_t lmufft_melprep_bytes(void);
int lmufft_melprep(void* melprep, void* stream);
int lmufft_mel_frames(int32_t samples);
int lmufft_mel_forward_f32(const float* wave, const int32_t* lengths, int32_t B,
                           int64_t ld_wave, int32_t L, float* mel, int32_t Tmax,
                           int normalize, const void* melprep, void* stream);

/* ------------------------------------------------------------------------
 * CTC head for inference.  Replaces CTCDecoder.forward (src/models/asr_model.py:
 * 34-50: log_probs = log_softmax(enc W^T + b)) followed by LMUASRModel.decode
 * with beam_width = 1 (src/models/asr_model.py:158-213: per utterance, arg-max
 * over the vocabulary for the first lengths[b] frames, then blanks and
 * consecutive repeats removed).
 *   enc [B][T][Hd] f32 or bf16 (dtype), W [V][Hd], bias [V] f32, V <= 64,
 *   lengths [B] (NULL = all T), blank = index of the blank (the reference uses
 *   vocab_size - 1);
 *   log_probs [B][T][V] f32 (may be NULL), tokens [B][T] (row b holds counts[b]
 *   tokens, then -1), counts [B].
 * ---------------------------------------------------------------------- */
size_t lmufft_ctc_workspace_bytes(int32_t B, int32_t T);
int lmufft_ctc_greedy(const void* enc, int dtype, const float* W, const float* bias,
                      const int32_t* lengths, int32_t B, int32_t T, int32_t Hd, int32_t V,
                      int32_t blank, float* log_probs, int32_t* tokens, int32_t* counts,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Attention core for training: forward with dropout on the attention weights and the backward.
 * Replaces MultiHeadAttention.forward's score / mask / softmax / dropout / context part,
 * src/models/lmu_encoder.py:72-90 (self.dropout at :45, :86), and what autograd derives from it.
 *   lmufft_attention_fwd_train: as lmufft_attention_fwd_ld; additionally element (b, h, query, key) of
 *   the weights is dropped with probability p_drop (kept weights scaled by 1/(1 - p_drop)), decided by a
 *   counter-based hash of the element's row (b, h, query), its key and (seed0, seed1): with
 *   g = row * ((T + 1) / 2) as a 64-bit number, x = fmix32((lo32(g) + key / 2) ^ seed0 ^ ((hi32(g) + seed1) * 0x9E3779B1))
 *   (fmix32: MurmurHash3's 32-bit finaliser) and the element is kept when the low (even key) or high (odd key)
 *   16 bits of x reach round(p_drop * 65536) -- no mask is stored -- and
 *   stats [B][H][T][2] f32 receives per query row the maximum score and 1 / sum exp(score - max).
 *   lmufft_attention_bwd: dq, dk, dv ([B][T][H*DK] with row stride ld_grad, dtype as the inputs) from
 *   q, k, v, the forward's ctx and stats, the upstream gradient dctx ([B][T][H*DK] dense) and the same
 *   (p_drop, seed0, seed1); dsum [B][H][T] f32 is scratch (receives dctx . ctx per row).
 *   Masked positions pass no gradient to q and k (masked_fill); rows past an utterance's end, which
 *   attend uniformly to all T positions, still pass gradient to v, as in the reference.
 * fp32 arithmetic for fp32 and bf16 tensors.  DK in {32, 64, 128}.
 * ---------------------------------------------------------------------- */
int lmufft_attention_fwd_train(const void* q, const void* k, const void* v, int dtype, int64_t ld,
                               const int32_t* lengths, int32_t B, int32_t T, int32_t H, int32_t DK,
                               float p_drop, uint32_t seed0, uint32_t seed1, void* ctx, float* stats,
                               void* stream);
int lmufft_attention_bwd(const void* q, const void* k, const void* v, const void* ctx, const void* dctx,
                         const float* stats, int dtype, int64_t ld, const int32_t* lengths, int32_t B,
                         int32_t T, int32_t H, int32_t DK, float p_drop, uint32_t seed0, uint32_t seed1,
                         void* dq, void* dk, void* dv, int64_t ld_grad, float* dsum, void* stream);

/* ------------------------------------------------------------------------
 * Backward of the residual add + LayerNorm (training).  Replaces what autograd derives for
 * layer_norm(output + residual), src/models/lmu_encoder.py:97, and layer_norm(x), :274:
 *   ds = rstd (gy - mean(gy) - xhat mean(gy xhat)), gy = dy gamma, xhat = (x + res - mean) rstd  -> dx (dtype of x)
 *   and, when dres != NULL, the same values in the dtype of res; dgamma = sum_rows dy xhat, dbeta = sum_rows dy (f32).
 * x, res (may be NULL), dy: [rows][C] f32 or bf16 each; mean and rstd are recomputed from x (+ res) with the
 * forward's arithmetic.  C a multiple of 4, at most 1024.  Sums in a fixed order.
 * ---------------------------------------------------------------------- */
size_t lmufft_layernorm_bwd_workspace_bytes(int32_t C);
int lmufft_layernorm_bwd(const void* x, int x_dtype, const void* res, int res_dtype, const void* dy, int dy_dtype,
                         const float* gamma, float eps, int64_t rows, int32_t C, void* dx, void* dres,
                         float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Layer norm followed by dropout (training), and its backward.  Replaces
 * `x = self.dropout_layer(layer_norm(x))`, src/models/lmu_encoder.py:274-277 (nn.Dropout, :232), as one pass
 * over HBM in each direction: the forward multiplies element (row, column) of the normalised output by
 * keep / (1 - p_drop), keep decided by the library's counter-based hash of the row, the column and the call's two
 * seeds (the function given at lmufft_attention_fwd_train with rows of C columns; no mask is stored); the backward
 * sends dy through the same decisions before the layer-norm backward above.  p_drop in [0, 1); 0 gives the plain
 * kernels.  With p_drop > 0 `out` may not alias x or res (the backward recomputes from them).
 * out_bf16 (may be NULL; needs a float32 `out`): receives the same output rounded to bf16 -- the copy the next
 * block's projections read under bf16 autocast while its residual connection keeps the float32 one (the reference
 * makes it with a cast kernel at every consumer).  dy2 (may be NULL; f32 or bf16): a second upstream gradient,
 * added to dy on load -- the gradient that arrived through that second copy.
 * ---------------------------------------------------------------------- */
int lmufft_add_layernorm_dropout(const void* x, int x_dtype, const void* res, int res_dtype, const float* gamma,
                                 const float* beta, float eps, int64_t rows, int32_t C, void* out, int out_dtype,
                                 float p_drop, uint32_t seed0, uint32_t seed1, void* out_bf16, void* stream);
int lmufft_layernorm_dropout_bwd(const void* x, int x_dtype, const void* res, int res_dtype, const void* dy,
                                 int dy_dtype, const void* dy2, int dy2_dtype, const float* gamma, float eps,
                                 int64_t rows, int32_t C, float p_drop, uint32_t seed0, uint32_t seed1, void* dx,
                                 void* dres, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* lmufft_layernorm_dropout_bwd with one more option: relu_x != 0 -- x is the output of a ReLU whose backward is the
 * next thing to happen to dx (in the encoder: the LMUFFT layer's h, src/models/lmu_encoder.py:266-277), and dx is
 * stored already multiplied by (x > 0), so that the layer below can be run with LMUFFT_BWD_GH_MASKED.  dres, if any,
 * is not masked (a residual input then needs its own dres buffer). */
int lmufft_layernorm_dropout_bwd_relu(const void* x, int x_dtype, const void* res, int res_dtype, const void* dy,
                                      int dy_dtype, const void* dy2, int dy2_dtype, const float* gamma, float eps,
                                      int64_t rows, int32_t C, float p_drop, uint32_t seed0, uint32_t seed1, void* dx,
                                      void* dres, float* dgamma, float* dbeta, void* workspace,
                                      size_t workspace_bytes, void* stream, int relu_x);
/* dgamma == dbeta == NULL in either backward: the workgroups' partial sums of the parameter gradients stay in the
 * workspace, and ONE later launch sums them for every layer norm of the step (six per step in the reference's
 * encoder).  `items` is a HOST array; rows and C as given to the backward call whose workspace the item names. */
typedef struct lmufft_ln_sum_item {
  int64_t rows;
  int32_t C;
  float* dgamma;
  float* dbeta;
  const void* workspace;
} lmufft_ln_sum_item;
int lmufft_layernorm_param_sum_many(const lmufft_ln_sum_item* items, int32_t count, void* stream);

/* ------------------------------------------------------------------------
 * Dropout masks under HIP graph replay.  The reference draws a fresh mask at every call of nn.Dropout
 * (src/models/lmu_encoder.py:45, 86, 232, 274-277); the seeds of lmufft_attention_fwd_train / lmufft_attention_bwd /
 * lmufft_add_layernorm_dropout / lmufft_layernorm_dropout_bwd are by-value arguments, which a captured graph
 * freezes.  lmufft_dropout_epoch registers, for the CURRENT device, one 32-bit counter in device memory (NULL
 * unregisters): every dropout kernel launched on that device afterwards adds its value, read when the kernel runs,
 * to seed1.  A step captured in a graph increments the counter inside the graph (any captured device-side add), so
 * every replay draws new masks; the forward and the backward of one step read the same value.  The counter must stay
 * allocated while such kernels can run.  No counterpart call in the reference.
 * ---------------------------------------------------------------------- */
int lmufft_dropout_epoch(const uint32_t* epoch_dev);

/* ------------------------------------------------------------------------
 * Weight and bias gradient of a Linear layer under bf16 autocast (training).  Replaces what autograd runs
 * for the reference's nn.Linear layers (src/models/lmu_encoder.py:40-43, 184, 229):
 *   dW[out][in] = sum_rows dy[row][out] x[row][in],  db[out] = sum_rows dy[row][out]   (db may be NULL)
 * with x [rows][in] and dy [rows][out] bf16 (16-byte aligned, in and out multiples of 8), fp32 results,
 * fp32 accumulation in a fixed order.  It is the layer's own weight-gradient kernel with an empty memory block.
 * ---------------------------------------------------------------------- */
size_t lmufft_linear_wgrad_workspace_bytes(int64_t rows, int32_t in, int32_t out);
int lmufft_linear_wgrad_bf16(const void* x, const void* dy, int64_t rows, int32_t in, int32_t out,
                             float* dW, float* db, void* workspace, size_t workspace_bytes, void* stream);
/* The same gradient in two calls: the partial sums of the row ranges into the workspace (the GEMM), then their sum
 * into dW / db (a short memory-bound kernel that only has to finish before the gradient is consumed, so a caller may
 * run it on another stream, beside the rest of the backward, once that stream waits for the first call).  Same
 * workspace for both; the results are the numbers lmufft_linear_wgrad_bf16 writes. */
int lmufft_linear_wgrad_slabs_bf16(const void* x, const void* dy, int64_t rows, int32_t in, int32_t out,
                                   void* workspace, size_t workspace_bytes, void* stream);
int lmufft_linear_wgrad_sum_bf16(int64_t rows, int32_t in, int32_t out, float* dW, float* db, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* The second call for several layers at once: one launch sums every item's partial results (the weight gradients of
 * a training step's Linear layers, at the end of the backward).  `items` is a HOST array; an item's fields mean what
 * the arguments of lmufft_linear_wgrad_sum_bf16 mean. */
typedef struct lmufft_wgrad_sum_item {
  int64_t rows;
  int32_t in, out;
  float* dW;
  float* db;
  void* workspace;
  size_t workspace_bytes;
} lmufft_wgrad_sum_item;
int lmufft_linear_wgrad_sum_many_bf16(const lmufft_wgrad_sum_item* items, int32_t count, void* stream);

/* ------------------------------------------------------------------------
 * CTC target packing.  Replaces the label concatenation at the head of
 * LMUASRModel.compute_loss, src/models/asr_model.py:145-151 (a Python loop with one
 * device-to-host .tolist() per utterance):
 *   targets [B][ld] int64 labels, utterance b holds target_lengths[b] <= Lmax of them
 *   (lengths are clamped to 0..Lmax); packed [sum of lengths] (B * Lmax always
 *   suffices); offsets [B + 1] int32: first packed position of every utterance,
 *   offsets[B] = the total.  Everything stays on the device.
 * ---------------------------------------------------------------------- */
int lmufft_ctc_pack_targets(const int64_t* targets, const int32_t* target_lengths, int32_t B,
                            int32_t Lmax, int64_t ld, int64_t* packed, int32_t* offsets,
                            void* stream);

/* ------------------------------------------------------------------------
 * CTC loss with its gradient (training).  Replaces nn.CTCLoss(blank = vocab_size - 1, reduction = 'mean',
 * zero_infinity = True) as LMUASRModel.compute_loss calls it, src/models/asr_model.py:105, 130-156, and its
 * backward:
 *   log_probs [B][T][V] f32 (V <= 64), targets [B][ld] int64 (utterance b holds target_lengths[b] <= Lmax labels,
 *   no packing needed), input_lengths [B];
 *   nll [B]: -log p(targets | log_probs) per utterance (+inf when nothing aligns);
 *   grad [B][T][V]: the gradient torch's ctc_loss returns for log_probs (the gradient through the log-softmax
 *   that produced them), times 1 / (B max(L_b, 1)) when reduction_mean != 0; rows t >= input_lengths[b] are zero;
 *   with zero_infinity an utterance whose nll is infinite gets a zero gradient.
 * The loss of the 'mean' reduction is mean_b(nll_b / max(L_b, 1)), infinite terms replaced by 0 under
 * zero_infinity; multiply grad by the upstream gradient of the loss.
 * The forward/backward recursions run on doubles in the linear domain, rescaled by an exact power of two at every
 * frame; the workspace (16-byte aligned, lmufft_ctc_loss_workspace_bytes) holds both sets of variables and their
 * scales for every frame.
 * ---------------------------------------------------------------------- */
size_t lmufft_ctc_loss_workspace_bytes(int32_t B, int32_t T, int32_t Lmax);
int lmufft_ctc_loss(const float* log_probs, const int64_t* targets, int64_t ld, const int32_t* input_lengths,
                    const int32_t* target_lengths, int32_t B, int32_t T, int32_t V, int32_t Lmax, int32_t blank,
                    int reduction_mean, int zero_infinity, float* nll, float* grad, void* workspace,
                    size_t workspace_bytes, void* stream);
/* loss[0] = mean_b(nll[b] / max(target_lengths[b], 1)), infinite terms replaced by 0 when zero_infinity != 0: the
 * 'mean' reduction of nn.CTCLoss (src/models/asr_model.py:105, 144-154) from lmufft_ctc_loss's nll, one launch, a
 * fixed summation order; nll, target_lengths and loss in device memory. */
int lmufft_ctc_mean_loss(const float* nll, const int32_t* target_lengths, int32_t B, int zero_infinity, float* loss,
                         void* stream);

/* Log-softmax over the vocabulary and its backward, around the CTC loss: `torch.log_softmax(logits, dim=-1)` of
 * LMUASRModel.forward (src/models/asr_model.py:126) on the padded bf16 (or f32) logits the projection kernel writes.
 *   logits / dlogits: [rows][ld], V <= ld <= 64 (columns V .. ld-1 are padding: ignored forward, zeroed backward);
 *   log_probs / dlog_probs: dense f32 [rows][V]. */
int lmufft_log_softmax_rows(const void* logits, int dtype, int64_t ld, int64_t rows, int32_t V, float* log_probs,
                            void* stream);
int lmufft_log_softmax_rows_bwd(const float* dlog_probs, const float* log_probs, int64_t rows, int32_t V, void* dlogits,
                                int dtype, int64_t ld, void* stream);

/* ------------------------------------------------------------------------
 * Optimizer step of the training loop on flat fp32 arenas.  Replaces the per-parameter
 * gradient-norm loop, the NaN check, clip_grad_norm_ and torch.optim.AdamW.step of
 * DistributedTrainer.train_epoch, src/training/distributed.py:207-231 (optimizer
 * built at :92-98; clipping in src/training/utils.py:653-665).
 *   lmufft_grad_norm2_f32: norm2[0] (device) = sum of squares of grads[0..n), summed in
 *   a fixed order; workspace of lmufft_grad_norm_workspace_bytes().
 *   lmufft_adamw_step_f32: decoupled weight decay, bias-corrected moments, eps 1e-8
 *   style denominator sqrt(v)/sqrt(1 - beta2^step) + eps, as torch.optim.AdamW;
 *   step counts from 1.  norm2 != NULL: the gradients are scaled by
 *   min(1, max_norm / (sqrt(norm2[0]) + 1e-6)) on the fly, and the whole update is
 *   skipped when the norm is NaN (the reference skips such a batch); norm2 == NULL:
 *   no clipping.  params, grads, exp_avg, exp_avg_sq: n floats each, 16-byte aligned,
 *   same element order.
 * ---------------------------------------------------------------------- */
size_t lmufft_grad_norm_workspace_bytes(void);
int lmufft_grad_norm2_f32(const float* grads, int64_t n, float* norm2, void* workspace,
                          size_t workspace_bytes, void* stream);
int lmufft_adamw_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                          int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int32_t step, const float* norm2, float max_norm,
                          void* stream);

/* The same update with its clock in device memory, for steps replayed from a HIP graph (launch arguments
 * are frozen at capture; the learning rate and the step count must not be).
 *   clock: 8 int32 on the device, zeroed once by the caller and then owned by this call:
 *     [0] scheduler steps taken, [1] optimizer steps taken (a step skipped for a NaN norm does not count, as
 *     the reference does not call optimizer.step() then, src/training/distributed.py:222-231), [2..5] the
 *     current step's learning rate, bias corrections and clip coefficient as floats (for monitoring).
 *   schedule (host memory, read during the call): the warm-up / staircase-decay rule of
 *     LearningRateScheduler.step, src/training/utils.py:246-263, evaluated on the device in double.  The rate
 *     of optimizer step k is the one the scheduler set after step k - 1 -- the loop calls optimizer.step(),
 *     then lr_scheduler.step() (src/training/distributed.py:229-234) -- i.e. lr0 for the first step.
 *     scheduled == 0: lr0 at every step. */
typedef struct lmufft_lr_schedule {
  double lr0;           /* the optimizer's constructor rate */
  double max_lr;
  double decay_rate;
  int32_t warmup_steps;
  int32_t decay_steps;
  int32_t scheduled;
} lmufft_lr_schedule;
int lmufft_adamw_step_clock_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                int64_t n, int32_t* clock, const lmufft_lr_schedule* schedule,
                                float beta1, float beta2, float eps, float weight_decay,
                                const float* norm2, float max_norm, void* stream);

/* ------------------------------------------------------------------------
 * Masked self-attention core for inference.  Replaces the score / softmax /
 * context part of MultiHeadAttention.forward, src/models/lmu_encoder.py:68-91
 * (scores = Q K^T / sqrt(d_k); masked_fill(mask == 0, -1e9) with mask[b,i,j] =
 * i < len_b and j < len_b, src/models/lmu_encoder.py:247-252; softmax;
 * weights V; heads concatenated), without the [B,H,T,T] tensors.
 *   q, k, v, ctx: [B][T][H*DK], f32 or bf16 (dtype), head h in columns
 *   h*DK .. (h+1)*DK-1 (the layout the reference's Linear layers produce and
 *   its output projection consumes); lengths [B] or NULL (no mask);
 *   DK in {32, 64, 128}.  fp32 arithmetic.  Dropout is not applied (eval).
 * ---------------------------------------------------------------------- */
int lmufft_attention_fwd(const void* q, const void* k, const void* v, int dtype,
                         const int32_t* lengths, int32_t B, int32_t T, int32_t H,
                         int32_t DK, void* ctx, void* stream);
/* The same with q, k, v rows `ld` elements apart (ld >= H*DK; ctx stays dense): the three tensors
 * may be column blocks of ONE projection output [B][T][3*H*DK] (q = base, k = base + H*DK,
 * v = base + 2*H*DK, ld = 3*H*DK), i.e. w_q, w_k, w_v of src/models/lmu_encoder.py:41-43,68-70
 * applied as a single GEMM. */
int lmufft_attention_fwd_ld(const void* q, const void* k, const void* v, int dtype, int64_t ld,
                            const int32_t* lengths, int32_t B, int32_t T, int32_t H,
                            int32_t DK, void* ctx, void* stream);

/* ------------------------------------------------------------------------
 * Residual add + layer normalisation for inference, one pass over HBM.
 * Replaces `self.layer_norm(output + residual)`, src/models/lmu_encoder.py:97
 * (attention block), and `x = layer_norm(x)` after every LMU layer,
 * src/models/lmu_encoder.py:274 (res = NULL); the dropout that follows each of
 * them is the identity in eval mode.
 *   out[r][:] = (v - mean(v)) / sqrt(var(v) + eps) * gamma + beta,  v = x[r][:] (+ res[r][:]),
 *   biased variance, statistics and arithmetic in fp32 as torch.nn.LayerNorm.
 *   x, res, out: [rows][C] contiguous, each f32 or bf16 (x_dtype, res_dtype,
 *   out_dtype; res may be NULL); gamma, beta: f32 [C]; C % 4 == 0, C <= 1024;
 *   out may alias x or res when their dtypes agree.
 * ---------------------------------------------------------------------- */
int lmufft_add_layernorm(const void* x, int x_dtype, const void* res, int res_dtype,
                         const float* gamma, const float* beta, float eps, int64_t rows,
                         int32_t C, void* out, int out_dtype, void* stream);

/* ------------------------------------------------------------------------
 * Linear layers of the encoder for bf16 inference: out = x W^T + bias with bf16
 * operands, fp32 accumulation, the bias added in fp32 and one rounding to bf16 --
 * the arithmetic of torch.nn.Linear under bf16 autocast.  Replaces
 * input_projection / output_projection (src/models/lmu_encoder.py:184, 229;
 * calls at :244, :278) and the attention block's w_q, w_k, w_v, w_o
 * (src/models/lmu_encoder.py:40-43; calls at :68-70, :94); the three input
 * projections run as one call over the row-concatenated weights.
 *   x: bf16 [rows][K]; W: bf16 [N][K] (nn.Linear.weight layout); bias: f32 [N]
 *   or NULL; out: bf16 [rows][N].  K % 8 == 0, N % 8 == 0, 16-byte aligned
 *   x, W and out.
 * ---------------------------------------------------------------------- */
int lmufft_linear_bf16(const void* x, const void* W, const float* bias, int64_t rows,
                       int32_t K, int32_t N, void* out, void* stream);

/* Operands of a Linear layer's training step under bf16 autocast from its float32 master weight, one launch
 * (autocast casts weight and bias at every call, and the input gradient needs W^T: four kernels per layer):
 *   W [N][K] f32, bias [N] f32 or NULL;  Wb [N_pad][K] bf16 (rows >= N zero: N_pad pads an output width that is not
 *   a multiple of 8);  WbT [K][ldT] bf16, columns 0 .. N_pad-1 written (ldT > N_pad: the layer is a column block of
 *   a wider concatenated weight);  bias_rounded [N_pad] f32 (may be NULL) = the bias rounded to bf16, widened. */
int lmufft_linear_wprep_bf16(const float* W, const float* bias, int32_t N, int32_t K, int32_t N_pad, void* Wb,
                             void* WbT, int64_t ldT, float* bias_rounded, void* stream);

/* The same conversion for several weights in ONE launch: a training step under autocast casts every Linear layer's
 * weight (src/models/lmu_encoder.py:40-43, 184, 229; src/models/asr_model.py's CTC projection) -- fifteen launches of
 * ~4 us at the reference's encoder sizes.  `items` is a HOST array (copied into the kernel's arguments); every item's
 * fields mean what the arguments of lmufft_linear_wprep_bf16 mean. */
typedef struct lmufft_wprep_item {
  const float* W;
  const float* bias;
  int32_t N, K, N_pad;
  void* Wb;
  void* WbT;
  int64_t ldT;
  float* bias_rounded;
} lmufft_wprep_item;
int lmufft_linear_wprep_many_bf16(const lmufft_wprep_item* items, int32_t count, void* stream);

/* The same GEMM followed, in its epilogue, by the residual add and layer norm of
 * the attention block: y = LN(bf16(x W^T + bias) + res) * gamma + beta
 * (`output = self.layer_norm(output + residual)`, src/models/lmu_encoder.py:94-97;
 * the dropout between them is the identity in eval mode).  The sum and the
 * statistics are fp32, the variance is biased.
 *   res: [rows][N] f32 or bf16 (res_dtype), or NULL; gamma, beta: f32 [N];
 *   y: [rows][N] f32 or bf16 (y_dtype); out: bf16 [rows][N], or NULL when the
 *   projection itself is not wanted.  Needs N <= 384; y may alias res when
 *   their dtypes agree. */
int lmufft_linear_ln_bf16(const void* x, const void* W, const float* bias, int64_t rows,
                          int32_t K, int32_t N, void* out, const void* res, int res_dtype,
                          const float* gamma, const float* beta, float eps, void* y,
                          int y_dtype, void* stream);

/* ------------------------------------------------------------------------
 * Per-kernel timing with HIP events on the caller's stream (no counterpart in
 * the reference, which has no profiling hooks).  While enabled, every kernel
 * launch made by this library is bracketed by two events; the totals are read
 * back per kernel name.  Off by default; enabling it makes calls allocate
 * events, so keep it off in production and under graph capture.
 * ---------------------------------------------------------------------- */
int lmufft_profile_enable(int on);
void lmufft_profile_reset(void);
int lmufft_profile_count(void);
int lmufft_profile_get(int index, char* name, size_t name_len, double* total_ms,
                       int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* LMUFFT_H_ */
