/*
 * tssep_hip.h -- C ABI of libtssep_hip.so: the MI355X (gfx950) kernels behind the
 * TS-VAD / TS-SEP forward/backward hot path.
 *
 * The reference (merlresearch/tssep) has no FFI: every operation below replaces a
 * PyTorch ATen call made from the reference's Python hot path.  Each entry point
 * cites the reference call site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - plain device pointers + explicit sizes/leading dimensions; no torch types
 *   - the CALLER owns every buffer (inputs, outputs, workspace); nothing is
 *     allocated, nothing global is mutated, no host synchronisation inside, and the
 *     library reads no environment variable: what runs is a function of the arguments
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*)
 *   - return value: 0 = ok, <0 = TSSEP_E_* (invalid shape / alignment / unsupported);
 *     never throws across the boundary
 *   - all floating point is IEEE fp32 (complex = interleaved re,im fp32)
 *   - re-entrant across streams/devices; no internal locking
 */
#ifndef TSSEP_HIP_H
#define TSSEP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSSEP_OK 0
#define TSSEP_E_SHAPE (-1)       /* a size is out of the supported range        */
#define TSSEP_E_ALIGN (-2)       /* a pointer / leading dimension is misaligned */
#define TSSEP_E_UNSUPPORTED (-3) /* valid request the kernels do not cover      */
#define TSSEP_E_LAUNCH (-4)      /* hipLaunch reported an error                 */
#define TSSEP_E_NULL (-5)        /* required pointer is NULL                    */

/* Library identification: returns ABI version (bumped on any signature change). */
int tssep_abi_version(void);
/* Name of the code object target the library was built for ("gfx950"). */
const char* tssep_arch(void);

/* ------------------------------------------------------------------ probes ---
 * Hardware self-checks used by tests: dump the lane<->element maps of the MFMA
 * instructions the kernels rely on.  out: device float[...] (see probe.hip). */
int tssep_probe_mfma(float* out_4x4, float* out_32x32, void* stream);
/* out[b] = XCD (HW_REG_XCC_ID) that ran workgroup b of an nblocks x 512-thread launch. */
int tssep_probe_xcc(int* out, int nblocks, void* stream);
/* Shader clock under load: out[2b] = s_memtime ticks, out[2b+1] = 100 MHz reference ticks spent by
 * block b in `iters` x 8 bf16 MFMAs per wave (heavy != 0) or as many s_sleep (heavy == 0). */
int tssep_probe_clock(int64_t* out, int nblocks, int iters, int heavy, void* stream);
/* Store-flavour probe: every one of `nblocks` workgroups rewrites its own bytes_per_wg (multiple of 4096) of buf
 * `reps` times with 16-byte stores of one flavour (0 plain, 1 sc0, 2 sc1, 3 sc0 sc1, 4 nt), streams `pressure`
 * bytes (0: none) of stream_src [2 x stream_bytes: the upper half receives 2/3 x pressure bytes of non-temporal
 * stores per rewrite] through the L2 between two rewrites, and -- read_flavour != 0 -- reads
 * the region of the workgroup 8 blocks on (same XCD) with 16-byte loads (1 sc1, 2 nt, 3 sc0 sc1, 4 sc0) after every
 * rewrite.  Under rocprofv3 --pmc WRITE_SIZE it tells whether rewritten lines reach the memory side once or every
 * time, and whether a peer's read forces them out (the exchange granules of the W-stationary recurrences: DESIGN 4.2). */
int tssep_probe_rewrite(float* buf, int nblocks, int bytes_per_wg, int reps, int flavour, int read_flavour,
                        const float* stream_src, int64_t stream_bytes, int pressure, float* sink, void* stream);

/* ------------------------------------------------------------------- STFT ----
 * paderbox-semantics STFT (fading + end padding + periodic window + rfft, no
 * scaling).  Replaces fe.stft at tssep/train/model.py:503-504.
 *   x      [rows, N]            real input (rows = B*C)
 *   window [size]               analysis window
 *   tw     [size/2 + size/4..]  twiddle table built by tssep_fft_twiddles
 *   X      [rows, T, size/2+1]  complex64 out, T = tssep_stft_frames(N,...)
 * FFT plans (tssep_stft_plan): 1 = size 1024 / shift 256, the plan of the reference's configs
 * (tssep/exp/init_cfg_common.yaml:33-43; specialised kernels, the only plan of the fused tssep_mask_istft_* entry points);
 * 2 = the general plan behind tssep_stft_fwd / tssep_istft_fwd / tssep_istft_bwd: size even, size / 2 = 2^a 3^b 5^c <= 2048,
 * 1 <= shift <= min(size, 512) -- 512 / 128, TorchMFCC's own default 400 / 200
 * (tssep/train/feature_extractor_torchaudio.py:24-25), ...; 0 = not built -> TSSEP_E_UNSUPPORTED. */
int tssep_stft_plan(int size, int shift);
int64_t tssep_stft_frames(int64_t N, int size, int shift, int window_length,
                          int pad, int fading);
int tssep_fft_twiddles(int size, float* host_out /* HOST buffer, 2*(size/2 + size/2+1) floats */);
int tssep_stft_fwd(const float* x, int64_t rows, int64_t N, int size, int shift,
                   int fading, const float* window, const float* tw,
                   float* X, int64_t T, void* stream);

/* Inverse STFT (irfft, biorthogonal synthesis window, overlap-add, un-fade,
 * truncate).  Replaces fe.istft at tssep/train/model.py:661-664.
 *   X [rows, T, F] complex64, wsyn [size], y [rows, N]
 * Optional fused LogMAE partial sums (tssep/train/loss.py:244-247): with tgt [rows,N] and
 * abs_partial [rows, tssep_istft_chunks(N)] non-NULL, abs_partial[row,c] = sum over the
 * chunk's samples of |y - tgt| (summed in a fixed order -> deterministic). */
int64_t tssep_istft_chunks(int64_t N);
int tssep_istft_fwd(const float* X, int64_t rows, int64_t T, int size, int shift,
                    int fading, const float* wsyn, const float* tw,
                    float* y, int64_t N, const float* tgt, float* abs_partial, void* stream);
/* Adjoint of tssep_istft_fwd: dy [rows,N] -> dX [rows,T,F] complex64
 * (torch convention: dRe + i dIm). */
int tssep_istft_bwd(const float* dy, int64_t rows, int64_t N, int size, int shift,
                    int fading, const float* wsyn, const float* tw,
                    float* dX, int64_t T, void* stream);

/* Mask head fused with the inverse STFT and with its adjoint: the chain
 *   mask = sigmoid(logit)                      tssep/train/net.py:983
 *   stft_estimate = Observation * mask         tssep/train/enhancer.py:98-100
 *   time_estimate = fe.istft(stft_estimate)    tssep/train/model.py:661-664
 * without writing mask or stft_estimate (results identical to tssep_maskhead_fwd followed by
 * tssep_istft_fwd; the unfused entry points remain for callers that want the intermediates).
 *   logit [B, K, T, F] fp32, obs [B, T, F] complex64 (reference channel), y [B*K, N]
 *   tgt / abs_partial: as tssep_istft_fwd (rows = B*K).
 * Backward: dy [B*K, N] -> dlogit [B, K, T, F] = Re(conj(obs) dEst) m (1 - m), dEst = adjoint(dy)
 * (identical to tssep_istft_bwd followed by tssep_maskhead_bwd with dmask = NULL). */
int tssep_mask_istft_fwd(const float* logit, const float* obs, int64_t B, int64_t K, int64_t T,
                         int size, int shift, int fading, const float* wsyn, const float* tw,
                         float* y, int64_t N, const float* tgt, float* abs_partial, void* stream);
int tssep_mask_istft_bwd(const float* dy, const float* logit, const float* obs, int64_t B,
                         int64_t K, int64_t N, int size, int shift, int fading,
                         const float* wsyn, const float* tw, float* dlogit, int64_t T, void* stream);
/* The same backward with the time-domain loss in front and the final Linear behind it folded in:
 *   input : est, tgt [B*K, N] and the loss's own backward arguments instead of dy -- the frame samples are
 *           d LogMAE / d est = gout[b] sign(est - tgt) / (N ln10 sums[b])  (tssep/train/loss.py:244-247;
 *           sums == NULL: MAE, gout[b] sign(est - tgt) / N, loss.py:214-216), i.e. what tssep_logmae_bwd
 *           would have written to a [B, K, N] buffer first (tgt == NULL: `est` IS dy, as above);
 *   output: bt_major = 0: dlogit [B, K, T, F] as above; bt_major = 1: rows (b, t) x (speaker position
 *           iperm[b*K + k] (NULL: k), f) -- the layout the final Linear's backward GEMMs read, i.e. what
 *           tssep_logit_map_bwd (trials = 1, 'tf' resolution) would have produced from dlogit
 *           (tssep/train/net.py:637-641, 957-967 backwards). */
int tssep_mask_istft_bwd_loss(const float* est, const float* tgt, const float* sums, const float* gout,
                              const float* logit, const float* obs, int64_t B, int64_t K, int64_t N,
                              int size, int shift, int fading, const float* wsyn, const float* tw,
                              const int32_t* iperm, int bt_major, float* dlogit, int64_t T, void* stream);

/* --------------------------------------------------------------- features ----
 * ConcaternatedSTFTFeatures(TorchMFCC, Log1pMaxNormAbsSTFT).stft_to_feature
 * (tssep/train/feature_extractor.py:352-360, feature_extractor_torchaudio.py:93-106,
 * feature_extractor.py:233-248).
 *   X     [B, T, F] complex64 (reference channel already selected)
 *   fb    [F, n_mels], dct [n_mels, n_mfcc]    (fp32, row major)
 *   out   [B, T, ld_out] : cols [0,n_mfcc) = MFCC, [n_mfcc, n_mfcc+F) = log1p feature
 *   ws    workspace, tssep_feat_workspace_bytes(B,T,n_mels,F,stat_axis) bytes
 * n_mfcc == 0 skips the MFCC block (plain Log1pMaxNormAbsSTFT).
 * stat_axis: the maximum the log1p feature is normalised by, Log1pMaxNormAbsSTFT's `statistics_axis`
 * (feature_extractor.py:239-242): 0 = 'tf' one per utterance (every shipped config), 1 = 't' one per
 * (utterance, frequency) over the frames, 2 = 'f' one per frame over the frequencies.
 * The dB floor (top_db) is taken over the WHOLE batch, as torchaudio's
 * AmplitudeToDB does for the 3-D input the reference passes.  top_db < 0 selects TorchMFCC's `log_mels`
 * (feature_extractor_torchaudio.py:98-100): log(mel + 1e-6) instead of dB, no floor.  The filterbank may be any
 * matrix whose columns are non-zero on ONE contiguous band (HTK or Slaney scale, with or without area normalisation). */
int64_t tssep_feat_workspace_bytes(int64_t B, int64_t T, int n_mels, int F, int stat_axis);
int tssep_feat_fwd(const float* X, int64_t B, int64_t T, int F,
                   const float* fb, const float* dct, int n_mels, int n_mfcc,
                   float top_db, int stat_axis, float* out, int64_t ld_out, void* ws, void* stream);

/* ------------------------------------------------------------------- GEMM ----
 * Exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32): C = epilogue(A x B).
 * Replaces nn.Linear / the LSTM input GEMM (tssep/train/rnnp.py:88-96,146-161,
 * tssep/train/net.py:663-666) and their autograd backward GEMMs.
 *   op A: a_kmajor=0 -> A(m,k)=A[m*lda+k]   ; a_kmajor=1 -> A(m,k)=A[k*lda+m]
 *   op B: b_kmajor=0 -> B(k,n)=B[n*ldb+k]   ; b_kmajor=1 -> B(k,n)=B[k*ldb+n]
 *   (kmajor=0 is the "row x row" form y = x W^T; kmajor=1 reads the transposed
 *    operand in place, used by dgrad / wgrad)
 * see struct for the fused prologue / epilogue options. */
typedef struct tssep_gemm_args {
  const float* A; const float* B; float* C;
  int64_t M, N, K;
  int64_t lda, ldb, ldc;          /* lda, ldb multiples of 4; A, B 16-byte aligned */
  int32_t a_kmajor, b_kmajor;     /* (1,0) is not built */
  /* time shift on the reduction index of a k-major B (wgrad of W_hh pairs dgates_t with
   * h_{t-1}): B row k is read at k+b_kshift and taken as 0 when (k % kperiod)+b_kshift
   * falls outside [0,kperiod).  kperiod = 0 disables. */
  int64_t b_kshift, kperiod;
  /* epilogue */
  const float* bias;              /* [N] added to every row; NULL = off */
  int32_t act;                    /* 0 none, 1 tanh (Tanh between post-net layers, net.py:623-625; computed as
                                   * 1 - 2 / (1 + e^2x) on the hardware exp / rcp: absolute error <= 3e-7),
                                   * 2 multiply by 1 - aux^2: the BACKWARD of that Tanh folded into the store of
                                   * the d(input) GEMM of the layer that consumed its output (aux = that input) */
  int32_t accumulate;             /* C += result */
  /* output remap (c_remap != 0): row m = (b*c_K + k)*c_T + t and column n = q*c_cm + r are
   * stored at C[b*c_sb + k*c_sk + t*c_st + q'*c_co + r], q' = c_perm ? c_perm[b*c_perm_ld+q] : q.
   * Covers 'spk time feature -> 1 time (spk feature)' (net.py:608-611) and
   * '1 time (spk mask freq) -> spk mask time freq' + the speaker un-permutation
   * (net.py:637-641, 957-967) inside the producing GEMM's store.
   * c_remap = 2: the same map through the plain 4-byte-per-lane store only (the reference the 16-byte variants
   * are tested against; same values). */
  int32_t c_remap;
  int64_t c_T, c_K, c_sb, c_sk, c_st, c_cm, c_co;
  const int32_t* c_perm; int64_t c_perm_ld;
  /* split-K: gridDim.z = splitk partial products are written to C + z*c_split_stride
   * (no bias/act/remap); the caller reduces them.  <=1 = single pass. */
  int32_t splitk; int64_t c_split_stride;
  /* arithmetic: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
   * 1 = split-bf16 "bf16x3": every fp32 operand is split on the fly into bf16 hi + bf16 lo and
   *     the product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
   *     (per-product relative error <= ~2^-16, i.e. fp32-class for the 1e-3 parity bar);
   * 2 = weight gradients only (a_kmajor = b_kmajor = 1): as 1 with the A_lo * B_hi product dropped -- dY enters as
   *     plain bf16, X keeps hi + lo (opt-in side line of bench.py, never the default);
   * 3 = plain bf16 (opt-in side line, never the default): both operands rounded to bf16, ONE product per k-step, fp32
   *     accumulation -- in the kernels that have the variant (the persistent row x row kernels TSSEP_GEMM_BIG_P /
   *     _BIG_P320); every other kernel computes a precision-3 request as precision 1 (at least as accurate). */
  int32_t precision;
  /* k-major B only, precision 1 only: column N-1 of B is VIRTUAL and reads as 1.0 for every valid
   * k, so column N-1 of C holds the column sums of A -- the bias gradient comes out of the
   * weight-gradient GEMM that streams d(gates) anyway (B then has N-1 real columns). */
  int32_t b_ones_col;
  /* act == 2 only: aux[m*ldaux + n], indexed like the UNREMAPPED C (row m, column n) */
  const float* aux; int64_t ldaux;
} tssep_gemm_args;
int tssep_gemm_f32(const tssep_gemm_args* args, void* stream);

/* The kernels behind tssep_gemm_f32.  The library picks one per request from the request alone (shape, layout,
 * epilogue): no environment variable, no global state.  A caller that wants to know, or to decide itself
 * (autotuning sweeps, A/B timing), uses the three calls below; every kernel of the split-bf16 family gives
 * bit-identical results for the same request (same k order per output element, same epilogue arithmetic). */
#define TSSEP_GEMM_AUTO 0        /* let the library choose                                                    */
#define TSSEP_GEMM_F32 1         /* precision 0: exact-fp32 MFMA, 128 x 128 tiles                              */
#define TSSEP_GEMM_PIPE 2        /* 128 x 128 x 32, every layout / epilogue                                    */
#define TSSEP_GEMM_TALL2 3       /* row x row, 256 x 128, four waves                                           */
#define TSSEP_GEMM_TALL4 4       /* row x row, 256 x 256, eight waves                                          */
#define TSSEP_GEMM_TALL4_XCOL 5  /* ... N = 256 q + 1: the last column on the VALU                             */
#define TSSEP_GEMM_BIG 6         /* row x row, 256 x 256, four waves of 128 x 128 (one per SIMD)               */
#define TSSEP_GEMM_STREAM 7      /* row x row, persistent, plain stores hidden behind the next tile            */
#define TSSEP_GEMM_NT_W160 8     /* row x row, 256 x 160                                                       */
#define TSSEP_GEMM_TN 9          /* weight gradient (both operands k-major), 128 x 128                         */
#define TSSEP_GEMM_TN_TALL 10    /* weight gradient, 256 x 128                                                 */
#define TSSEP_GEMM_TN_BIG 11     /* weight gradient, 512 x 128, four waves of 128 x 128                        */
#define TSSEP_GEMM_TN_W160 12    /* weight gradient, 256 x 160; eight waves on 256 x 320 / 256 x 256 (also swapped)  */
#define TSSEP_GEMM_TN_H160 13    /* weight gradient, 320 x 128                                                 */
#define TSSEP_GEMM_BIG_P 14      /* row x row, 256 x 256 persistent: plain / bias / Tanh store hidden behind tiles */
#define TSSEP_GEMM_BIG_P320 15   /* row x row, 192 x 320 persistent (N = 320 q): plain / bias / Tanh / its backward  */
#define TSSEP_GEMM_TN_P320 16    /* weight gradient, 192 x 320 (N = 320 q, + the ones column)                  */
#define TSSEP_GEMM_KERNEL_LAST 16
/* As tssep_gemm_f32, on the kernel named (TSSEP_GEMM_AUTO = tssep_gemm_f32); TSSEP_E_UNSUPPORTED when that
 * kernel does not cover the request. */
int tssep_gemm_f32_on(const tssep_gemm_args* args, int32_t kernel, void* stream);
/* *kernel = the kernel tssep_gemm_f32_on(args, force, .) would launch; nothing is launched, C may be NULL. */
int tssep_gemm_plan(const tssep_gemm_args* args, int32_t force, int32_t* kernel);
const char* tssep_gemm_kernel_name(int32_t kernel);
/* Recommended splitk (>= 1; < 0 = TSSEP_E_*) for the weight gradient dW[M,N] = dY[K,M]^T X[K,N] described by
 * `args` (a_kmajor = b_kmajor = 1; splitk, c_split_stride and C are ignored): follows the kernel the library
 * picks for it. */
int tssep_gemm_wgrad_splits(const tssep_gemm_args* args);
/* The split count the split rule OF kernel `kernel_id` (TSSEP_GEMM_*) gives this request: tssep_gemm_wgrad_splits returns
 * an S for which tssep_gemm_plan(.., splitk = S) names a kernel k with tssep_gemm_wgrad_split_rule(g, k) == S (a fixed
 * point), or 8 when there is none (ABI 4). */
int tssep_gemm_wgrad_split_rule(const tssep_gemm_args* g, int32_t kernel_id);

/* column sums: out[n] (+)= sum_m A[m*lda+n]  (bias gradients) */
int tssep_colsum_f32(const float* A, int64_t M, int64_t N, int64_t lda, float* out,
                     int accumulate, void* ws, void* stream);
int64_t tssep_colsum_workspace_bytes(int64_t M, int64_t N);

/* ------------------------------------------------------------------ BLSTM ----
 * One bidirectional LSTM layer, zero initial state (torch.nn.LSTM as used at
 * tssep/train/rnnp.py:88-95,146-153).  The input projection x W_ih^T + b is done by
 * tssep_gemm_f32 into `gates`; these kernels run the T-sequential recurrence.
 *
 * Packed layouts (built by tssep_lstm_pack from torch-layout parameters):
 *   gate columns are ordered [dir][unit][gate(i,f,g,o)]  (4H per direction)
 *   wih_p  [2*4H, ld_i]   rows permuted to that order, K zero-padded to ld_i
 *   bias_p [2*4H]         b_ih + b_hh, permuted
 *   whh_f / whh_b         streaming layouts for the forward / backward recurrence
 * sizes in floats: tssep_lstm_pack_sizes(). */
typedef struct tssep_lstm_sizes {
  int64_t wih_p, bias_p, whh_f, whh_b;   /* floats */
} tssep_lstm_sizes;
int tssep_lstm_pack_sizes(int H, int I, int64_t ld_i, tssep_lstm_sizes* out);
/* torch-layout parameters of both directions (suffix _f = forward, _r = "_reverse"):
 * w_ih [4H,I], w_hh [4H,H], b_ih [4H], b_hh [4H]  (gate-major rows i,f,g,o). */
int tssep_lstm_pack(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f,
                    const float* b_hh_f, const float* w_ih_r, const float* w_hh_r,
                    const float* b_ih_r, const float* b_hh_r, int H, int I, int64_t ld_i,
                    float* wih_p, float* bias_p, float* whh_f, float* whh_b, void* stream);
/* gates [N,T,2,H,4]: in = pre-activations from the input GEMM; out = activated gates
 * (kept for backward).  cell [N,T,2,H]; hout [N,T,ldo] (cols d*dstride+u).  H <= 512 (round 4; the
 * W-stationary families below cover H <= 320 / 304, which is where the reference's configs live). */
int tssep_blstm_fwd(float* gates, float* cell, float* hout, int64_t ldo, int64_t dstride,
                    const float* whh_f, int64_t N, int64_t T, int H, void* stream);
/* dhout [N,T,ldo] -> gates is overwritten with d(pre-activation) in the same layout. */
int tssep_blstm_bwd(float* gates, const float* cell, const float* dhout, int64_t ldo,
                    int64_t dstride, const float* whh_b, int64_t N, int64_t T, int H, void* stream);
/* gradient unpack + split-K reduction: a matrix with packed (dir,unit,gate) rows back to
 * torch's gate-major rows:
 *   dst_d[(g*H + u)*ncols + k] (+)= sum_{s<nsplit} src[s*split_stride + (d*4H + 4u + g)*ld + k] */
int tssep_lstm_unpack(const float* src, int64_t ld, int nsplit, int64_t split_stride,
                      int H, int ncols, float* dst_f, float* dst_r, int accumulate, void* stream);

/* Cluster (W-stationary) recurrence -- latency-optimised alternative to tssep_blstm_fwd/bwd for
 * batches that do not fill the chip (same math, same tensor layouts; see lstm_cluster.hip).
 * A cluster of ceil(H/32) co-resident workgroups keeps W_hh in registers and exchanges h / dh
 * through 8-byte {tag,value} granules in `xbuf` (tssep_lstm_cluster_xbuf_bytes; zeroed inside
 * the call by a memset node on `stream`).  `max_wgs` = number of CUs (sizes the grid so that
 * neighbouring clusters hide each other's exchange latency; cluster membership is taken by
 * arrival ticket, so correctness does not depend on residency or dispatch order).
 * `ms` = sequences per cluster / 4: 2, 4, or 0 = automatic.  H <= 304, T < 65535.
 *
 * `err` (both W-stationary families, cluster and on-chip): device int[4], zeroed ONCE by the caller
 * and then owned by the library: err[0] is set non-zero when a bounded spin timed out (the workgroup
 * gives up and the outputs of that launch are garbage: the caller MUST read err[0] before it trusts
 * or checkpoints anything computed from them); err[1] is the launch epoch the granule tags carry,
 * advanced on the device in front of every launch (so captured hipGraphs replay with fresh epochs).
 *
 * CONCURRENCY CONTRACT: at most ONE W-stationary launch (tssep_blstm_cluster_* / tssep_blstm_onchip_*)
 * may be in flight per device at any time, across all streams and processes that share `err`: the
 * clusters are formed from the workgroups of one launch by arrival ticket and spin on their peers,
 * and two such grids competing for CUs can starve each other into the timeout.  Launch them on one
 * stream (other kernels may run concurrently on other streams). */
int tssep_lstm_cluster_supported(int H);
int64_t tssep_lstm_cluster_pack_floats(int H, int which /* 0: whh_cf, 1: whh_cb */);
int tssep_lstm_pack_cluster(const float* w_hh_f, const float* w_hh_r, int H,
                            float* whh_cf, float* whh_cb, void* stream);
int64_t tssep_lstm_cluster_xbuf_bytes(int64_t N, int H, int backward, int max_wgs, int ms);
int tssep_blstm_cluster_fwd(float* gates, float* cell, float* hout, int64_t ldo, int64_t dstride,
                            const float* whh_cf, void* xbuf, int* err,
                            int64_t N, int64_t T, int H, int max_wgs, int ms, void* stream);
int tssep_blstm_cluster_bwd(float* gates, const float* cell, const float* dhout, int64_t ldo,
                            int64_t dstride, const float* whh_cb, void* xbuf, int* err,
                            int64_t N, int64_t T, int H, int max_wgs, int ms, void* stream);

/* On-chip-weights recurrence on the bf16 matrix cores (lstm_onchip.hip): a cluster of
 * ceil(H/64) workgroups keeps W_hh on chip as split bf16 (hi+lo) and evaluates the recurrent
 * product as hi*hi + hi*lo + lo*hi with fp32 accumulation (fp32-class accuracy), 32 sequences per
 * cluster and step; h / partial dh travel as fp32 in 8-byte {tag, value} granules (the tag carries
 * a per-launch epoch and the step: the data is the flag).  Clusters are formed from workgroups of
 * ONE XCD (HW_REG_XCC_ID), so the exchange stays in that XCD's L2.  Same tensor layouts as
 * tssep_blstm_fwd/bwd.  `layout` bits: 1 = experimental time-major-in-groups-of-32 row order,
 * 8 = force cross-XCD clusters (write-through exchange), 32 = forward only: keep the activation
 * stream temporal (default: non-temporal from 160 sequences up, so that it does not evict the
 * exchange granules from the L2).  xbuf: caller-owned 16-byte-aligned scratch of
 * tssep_lstm_onchip_xbuf_bytes() bytes (zeroed by the call); err: device int[4] and the concurrency
 * contract as stated for the cluster kernels above.  max_wgs: number of CUs the launch may occupy.
 * H <= 304. */
int tssep_lstm_onchip_supported(int H);
/* Longest sequence (frames) one launch takes for `seqs_per_item` sequences per work item (32: the kernels
 * declared here; 16: the interleaved tssep_blstm_onchip16_* below): 32-bit lane offsets into the gate tensor,
 * ((1 << 31) - 8192) / ((seqs - 1) 2H 16) -- 14 913 / 7 215 frames at H = 300.  Longer: TSSEP_E_SHAPE (the
 * streaming kernels tssep_blstm_fwd/bwd have no limit, like tssep/train/rnnp.py:111-173). */
int64_t tssep_lstm_onchip_max_steps(int H, int seqs_per_item);
int64_t tssep_lstm_onchip_pack_floats(int H, int which /* 0: forward, 1: backward */);
int tssep_lstm_pack_onchip(const float* w_hh_f, const float* w_hh_r, int H, float* wf, float* wb,
                           void* stream);
int64_t tssep_lstm_onchip_xbuf_bytes(int64_t N, int H, int backward);
int tssep_blstm_onchip_fwd(float* gates, float* cell, float* hout, int64_t ldo, int64_t dstride,
                           const float* wf, void* xbuf, int* err, int64_t N, int64_t T, int H,
                           int max_wgs, int layout, void* stream);
int tssep_blstm_onchip_bwd(float* gates, const float* cell, const float* dhout, int64_t ldo,
                           int64_t dstride, const float* wb, void* xbuf, int* err, int64_t N,
                           int64_t T, int H, int max_wgs, int layout, void* stream);

/* Interleaved forward of the same recurrence (round 3): `groups` (2 or 1; 4 on request) independent groups of 16 sequences per
 * cluster share the stationary W_hh in rotation, so that a group's exchange of h overlaps the other groups' MFMAs and
 * cell updates (the kernel above walks one group of 32 sequences through a serial chain).  Same tensors, same
 * semantics, same err / concurrency contract; own weight pack (16 x 16 x 32 MFMA fragments) and own exchange
 * buffer layout.  tssep_blstm_onchip16_groups(): the group count the launcher uses for N sequences on max_wgs CUs
 * (0: shape not supported -- H % 4, H > 320 -- call tssep_blstm_onchip_fwd instead); groups = 0 in the launch = that
 * choice.  ldo, dstride multiples of 4; gates / cell / hout 16-byte aligned; XCD-local clusters only. */
int64_t tssep_lstm_onchip16_pack_floats(int H);
int tssep_lstm_pack_onchip16(const float* w_hh_f, const float* w_hh_r, int H, float* wf, void* stream);
int64_t tssep_lstm_onchip16_xbuf_bytes(int64_t N, int H);
int tssep_blstm_onchip16_groups(int64_t N, int H, int max_wgs);
int tssep_blstm_onchip16_fwd(float* gates, float* cell, float* hout, int64_t ldo, int64_t dstride,
                             const float* wf, void* xbuf, int* err, int64_t N, int64_t T, int H,
                             int max_wgs, int layout, int groups, void* stream);
/* Interleaved backward: the same rotation for the reduce-scatter of dh (exchange waves / io waves as in the forward, a
 * ring of four LDS slots {gate activations, c_(t-1), dh} filled by asynchronous copies two phases ahead).  Own weight
 * pack (W_hh^T as 16 x 16 x 32 MFMA fragments) and exchange layout; H = 257 .. 320 (five workgroups per cluster),
 * H % 4 == 0; groups as for the forward (0 = the launcher's choice). */
int64_t tssep_lstm_onchip16_bwd_pack_floats(int H);
int tssep_lstm_pack_onchip16_bwd(const float* w_hh_f, const float* w_hh_r, int H, float* wb, void* stream);
int64_t tssep_lstm_onchip16_bwd_xbuf_bytes(int64_t N, int H);
int tssep_blstm_onchip16_bwd(float* gates, const float* cell, const float* dhout, int64_t ldo, int64_t dstride,
                             const float* wb, void* xbuf, int* err, int64_t N, int64_t T, int H, int max_wgs,
                             int layout, int groups, void* stream);

/* ---------------------------------------------------------- elementwise ------*/
/* d(pre-tanh) = dy * (1 - y^2) for the Tanh between post-net layers (tssep/train/net.py:623-625).
 * dz rows are (b,k,t) x P; combined_in != 0: dy and y live in the speaker-combined layout
 * [B,T,K*P] of 'spk time feature -> 1 time (spk feature)' (net.py:608-611). */
int tssep_tanh_bwd(const float* dy, const float* y, float* dz, int64_t rows, int64_t P,
                   int64_t K, int64_t T, int combined_in, void* stream);
/* Speaker conditioning (tssep/train/net.py:862-896) fused with the permutation-trial fold
 * (net.py:913-924): xs rows are (b, trial, k, t); trial tr holds speaker (k+tr)%K at position k.
 *   mul: xs[row, f] = pre[(b,t), f] * aux[(b,spk), f]
 *   cat: xs[row, :] = [pre[(b,t), :F] | aux[(b,spk), :E]]
 * bwd: dpre[(b,t), f] = sum over (trial,k) rows (aux has no gradient: aux_net is None). */
int tssep_cond_mul_fwd(const float* pre, int64_t ld_pre, const float* aux, int64_t ld_aux,
                       float* xs, int64_t ld_xs, int64_t B, int64_t K, int64_t T, int F,
                       int trials, void* stream);
int tssep_cond_mul_bwd(const float* dxs, int64_t ld_dxs, const float* aux, int64_t ld_aux,
                       float* dpre, int64_t ld_dpre, int64_t B, int64_t K, int64_t T, int F,
                       int trials, void* stream);
int tssep_cond_cat_fwd(const float* pre, int64_t ld_pre, const float* aux, int64_t ld_aux,
                       float* xs, int64_t ld_xs, int64_t B, int64_t K, int64_t T, int F, int E,
                       int trials, void* stream);
int tssep_cond_cat_bwd(const float* dxs, int64_t ld_dxs, float* dpre, int64_t ld_dpre,
                       int64_t B, int64_t K, int64_t T, int F, int trials, void* stream);

/* -------------------------------------------------------------- mask head ----
 * mask = sigmoid(logit); est = Obs * mask   (tssep/train/net.py:981-986 +
 * tssep/train/enhancer.py:98-100).   logit/mask [B,K,T,F] fp32, obs [B,T,F] c64,
 * est [B,K,T,F] c64.  Algorithmic HBM bytes per (b,t): 16*K*F + 8*F. */
int tssep_maskhead_fwd(const float* logit, const float* obs, float* mask, float* est,
                       int64_t B, int64_t K, int64_t T, int F, void* stream);
/* dlogit = (Re(conj(obs) * dest) + dmask) * m * (1 - m); dmask may be NULL. */
int tssep_maskhead_bwd(const float* dest, const float* dmask, const float* mask,
                       const float* obs, float* dlogit,
                       int64_t B, int64_t K, int64_t T, int F, void* stream);

/* Masking on a given mask (standalone enhancer call, tssep/train/enhancer.py:98-100):
 * est = Obs * mask, and its backward dmask = Re(conj(Obs) * dest). */
int tssep_mask_mul_fwd(const float* mask, const float* obs, float* est,
                       int64_t B, int64_t K, int64_t T, int F, void* stream);
int tssep_mask_mul_bwd(const float* dest, const float* obs, float* dmask,
                       int64_t B, int64_t K, int64_t T, int F, void* stream);

/* ------------------------------------------------------------------ losses ---
 * LogMAE (tssep/train/loss.py:244-247): loss[b] = log10(sum_k mean_n |est-tgt|).
 * Deterministic two-stage reduction; `sums[b]` (the argument of the log) is kept for bwd.
 * tssep_logmae_finalize consumes the partial sums tssep_istft_fwd can emit
 * (partial [B*K, nchunks]).  bwd: dest = gout[b] * sign(est-tgt) / (N ln10 sums[b]).
 * MAE (tssep/train/loss.py:194-216) is the same reduction without the logarithm: its value is
 * sums[b], and tssep_logmae_bwd with sums == NULL gives its gradient gout[b] * sign(est-tgt) / N. */
int64_t tssep_logmae_chunks(int64_t N);
int64_t tssep_logmae_workspace_bytes(int64_t B, int64_t K, int64_t N);
int tssep_logmae_fwd(const float* est, const float* tgt, int64_t B, int64_t K, int64_t N,
                     float* loss, float* sums, void* ws, void* stream);
int tssep_logmae_finalize(const float* partial, int64_t B, int64_t K, int64_t nchunks, int64_t N,
                          float* loss, float* sums, void* stream);
int tssep_logmae_bwd(const float* est, const float* tgt, const float* sums, const float* gout,
                     int64_t B, int64_t K, int64_t N, float* dest, void* stream);
/* VADSigmoidBCE with target 'Vad' (tssep/train/loss.py:329-345,302-310):
 * x = mean_f logit[b,k,t,:]; loss[b] = mean_{k,t} BCEWithLogits(x, vad).
 * xmean [B,K,T] is kept for bwd; ws: tssep_vadbce_workspace_bytes.
 * bwd writes dlogit[b,k,t,f] = gout[b]*(sigmoid(x)-y)/(K*T*F). */
int64_t tssep_vadbce_workspace_bytes(int64_t B, int64_t K, int64_t T);
int tssep_vadbce_fwd(const float* logit, const float* vad, int64_t B, int64_t K, int64_t T, int F,
                     float* loss, float* xmean, void* ws, void* stream);
int tssep_vadbce_bwd(const float* xmean, const float* vad, const float* gout,
                     int64_t B, int64_t K, int64_t T, int F, float* dlogit, void* stream);

/* -------------------------------------------------------- logit layout map ---
 * Tail of MaskEstimator_v2.forward: final einops rearrange / reduce-repeat
 * (tssep/train/net.py:631-659), mean over permutation trials (net.py:928-951) and the speaker
 * un-permutation (net.py:957-967):  raw GEMM output -> out [B,K,T,F].
 *   spk_rows = 0: raw[((b*trials+tr)*T + t) * (K*Fr) + k*Fr + fr]   (ts_vad combination layer)
 *   spk_rows = 1: raw[((b*K + k)*T + t) * Fr + fr]                   (ts_vad off, trials == 1)
 *   Fr = F ('tf') or 1 ('t': the value is repeated over frequency).
 * perm[b,s] = output index of speaker s, iperm its inverse (both NULL = identity). */
int tssep_logit_map_fwd(const float* raw, const int32_t* perm, const int32_t* iperm, int64_t B,
                        int trials, int64_t K, int64_t T, int F, int Fr, int spk_rows,
                        float* out, void* stream);
int tssep_logit_map_bwd(const float* dout, const int32_t* perm, const int32_t* iperm, int64_t B,
                        int trials, int64_t K, int64_t T, int F, int Fr, int spk_rows,
                        float* draw, void* stream);

/* ------------------------------------------------- mask-based MVDR beamformer ----
 * Eval-time enhancer TorchBF('mvdr_souden') of the reference, tssep/train/enhancer.py:140-265, in
 * complex128 as there (the reference asserts Observation.dtype == complex128, :224):
 *   psd_m[k,f]  = sum_t w_m[k,t,f] Y[:,t,f] Y[:,t,f]^H     w_0 = mask 0 (target),
 *                                                          w_1 = mask 1, or 1 - mask 0 when M == 1
 *   phi         = psd_1^-1 psd_0     (LU, partial pivoting = torch.linalg.solve / zgesv)
 *   bf[k,f,:]   = phi[:, reference_channel] / max(Re trace(phi), eps)
 *   enh[k,t,f]  = sum_d conj(bf[k,f,d]) Y[d,t,f]     (* max(mask 0, masking_eps) if masking)
 * obs  [B,D,T,F] complex128 (interleaved re,im doubles), D <= 8
 * masks [B,K,M,T,F] fp32 (mask_f64 = 0) or fp64 (mask_f64 = 1), M in {1,2}
 * enh  [B,K,T,F] complex128.      eps: pass DBL_MIN for the reference's eps=None.
 * info[0] (device int) receives the number of (b,k,f) systems with an exactly zero pivot
 * (torch.linalg.solve raises for those; the host mirror does the same).
 * Algorithmic HBM bytes: 2 * 16*D*T*F (Y for the statistics, Y again for the filtering)
 *   + K*M*T*F*sizeof(mask) + 16*K*T*F, per batch element.
 * The three stages are exported for tests and profiling; _souden_fwd runs them back to back on
 * `stream` out of one caller-owned workspace of tssep_mvdr_workspace_bytes().  No host sync. */
int64_t tssep_mvdr_partial_bytes(int64_t B, int K, int D, int64_t T, int F);
int64_t tssep_mvdr_workspace_bytes(int64_t B, int K, int D, int64_t T, int F);
int tssep_mvdr_psd(const double* obs, const void* masks, int mask_f64, double* partials,
                   int64_t B, int K, int M, int D, int64_t T, int F, void* stream);
/* wconj [B,K,D,F] complex128 = conj(bf), bin-contiguous.  Consumes `partials` (the chunk sums are
 * formed in place).  masking_eps is compared in double: a caller that holds fp32 masks passes the
 * fp32-rounded value to reproduce torch.clamp(mask_fp32, min=masking_eps) bit for bit. */
int tssep_mvdr_weights(double* partials, double* wconj, int* info, int64_t B, int K, int D,
                       int64_t T, int F, int reference_channel, double eps, void* stream);
int tssep_mvdr_apply(const double* obs, const double* wconj, const void* masks, int mask_f64,
                     double* enh, int64_t B, int K, int M, int D, int64_t T, int F, int masking,
                     double masking_eps, void* stream);
int tssep_mvdr_souden_fwd(const double* obs, const void* masks, int mask_f64, double* enh,
                          void* workspace, int* info, int64_t B, int K, int M, int D, int64_t T,
                          int F, int reference_channel, double eps, int masking,
                          double masking_eps, void* stream);

/* -------------------------------------------------------------- optimizer -----
 * One optimizer step on flat fp32 buffers: global-norm gradient clipping
 * (torch.nn.utils.clip_grad_norm_, max_norm <= 0 disables) + Adam (torch.optim.Adam update rule,
 * amsgrad off) -- the reference's trainer settings (tssep/train/experiment.py:147-150).  No host
 * synchronisation; norm_out[0] (optional) receives the pre-clip gradient norm.
 * step = 1-based update count; ws: tssep_adam_workspace_bytes(). */
int64_t tssep_adam_workspace_bytes(void);
int tssep_adam_step(float* param, float* exp_avg, float* exp_avg_sq, const float* grad, int64_t n,
                    int64_t step, float max_norm, float lr, float beta1, float beta2, float eps,
                    float weight_decay, float* norm_out, void* ws, void* stream);
/* The same step behind a device-side guard: err = the error flag of the W-stationary recurrences (err[0] != 0: a launch
 * of this step gave up on a peer, the gradient is garbage) -> the update is skipped on the device, parameters and
 * moments stay those of the last good step; the host raises at its next check of the flag.  err == NULL: no guard. */
int tssep_adam_step_guarded(float* param, float* exp_avg, float* exp_avg_sq, const float* grad, int64_t n,
                            int64_t step, float max_norm, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float* norm_out, void* ws, const int* err, void* stream);

/* split-K / slab reduction: dst[i] (+)= sum_{s<nsplit} src[s*stride + i] */
int tssep_reduce_splits(const float* src, int nsplit, int64_t stride, int64_t count, float* dst,
                        int accumulate, void* stream);
/* the same for the partials [nsplit][M][ldp] of a weight-gradient GEMM with b_ones_col (tssep_gemm_f32):
 * columns [0,N) -> dw [M, ld_w], column N (the column sums of dY = the bias gradient of an nn.Linear,
 * tssep/train/rnnp.py:96,161, net.py:663-666) -> db [M] */
int tssep_reduce_splits_bias(const float* src, int nsplit, int64_t stride, int64_t M, int64_t N,
                             int64_t ldp, float* dw, int64_t ld_w, float* db, int accumulate,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSSEP_HIP_H */
