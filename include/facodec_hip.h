/*
 * facodec_hip.h -- C ABI of libfacodec_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the FAcodec encode -> factorized-VQ -> decode (+ multi-scale
 * spectral loss) hot path.  The reference (Plachtaa/FAcodec) is pure Python/PyTorch and has
 * no FFI; what it binds to underneath its nn.Modules are ATen ops.  Each entry point below
 * names the reference op (file:line under /root/reference) it replaces.  The Python host
 * (facodec_amd/) keeps the reference's module surface (build_model / encoder / quantizer /
 * decoder, modules/commons.py:283-348) and calls these through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (torch tensor storage);
 *     dense row-major fp32 unless noted; codes are int64.
 *   - every call is asynchronous on the hipStream_t passed as `stream` (void*); the library
 *     never allocates, frees or synchronises device memory.
 *   - return 0 on success, negative on error; fac_last_error() gives a thread-local message.
 *   - activations use the reference layout (B, C, T), T contiguous.
 */
#ifndef FACODEC_HIP_H
#define FACODEC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fac_stream_t; /* hipStream_t */

#define FAC_OK 0
#define FAC_ERR_ARG (-1)
#define FAC_ERR_LAUNCH (-2)

#define FAC_PAD_ZERO 0
#define FAC_PAD_REFLECT 1

#define FAC_ACT_NONE 0
#define FAC_ACT_TANH 1  /* nn.Tanh, dac/model/dac.py:159 */
#define FAC_ACT_MISH 2  /* x*tanh(softplus(x)), modules/style_encoder.py:6-10 */
#define FAC_ACT_LOG_MEL 3 /* (log(1e-5+x)+4)/4, modules/quantize.py:241 */
/* The two epilogues of a WaveNet layer (modules/wavenet.py:138-166), taken only by the few-column split-reduction launches of
 * the streaming hop (B * T_out <= 640; fac_conv1d_fwd fails for any other shape -- there the elementwise kernels
 * fac_gate_tanh_sigmoid / fac_wn_res_skip do the same arithmetic):
 *   FAC_ACT_GATE        : y (B, C_out / 2, T) = tanh(c[:, :C_out/2]) * sigmoid(c[:, C_out/2:]) of c = conv + bias
 *                         (fused_add_tanh_sigmoid_multiply without conditioning, modules/commons.py:113-120); C_out % 256 == 0.
 *   FAC_ACT_WN_RES_SKIP : c = conv + bias; y (B, C_out / 2, T) = res + c[:, :C_out/2] (res may be y itself) and
 *                         y2 (B, C_out / 2, T) += c[:, C_out/2:] (no alpha_y2; y2 is the running skip sum). */
#define FAC_ACT_GATE 4
#define FAC_ACT_WN_RES_SKIP 5

/* ABI version; bumped whenever a signature changes. */
int fac_version(void);
/* Thread-local description of the last failure (never NULL). */
const char* fac_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K6  weight-norm materialisation + packing to the GEMM-friendly layout the conv kernels read.
 * Replaces torch.nn.utils.weight_norm's  w = v * (g / ||v||)  recomputed every forward
 * (dac/model/encodec.py:42-51, dac/nn/layers.py:9-14).
 * ---------------------------------------------------------------------------------------- */

/* scale[i] = g[i] / ||v[i,:]||_2  for i < n_slices (slice = all dims but 0, contiguous,
 * slice_len floats).  g == NULL -> scale[i] = 1 (plain, un-normed weight). */
int fac_wn_scale(const float* v, const float* g, float* scale, int n_slices, int slice_len,
                 fac_stream_t stream);

/* Conv1d / Linear weight v (C_out, C_in, K) -> packed (C_in_pad, K, C_out_pad), co fastest,
 * packed[ci][k][co] = v[co][ci][k] * scale[co]; columns C_out..C_out_pad-1 and rows
 * C_in..C_in_pad-1 are zero: this call writes the WHOLE (C_in_pad, K, C_out_pad) buffer, padding included (an
 * uninitialised allocation is fine).  C_out_pad = fac_pad32(C_out), C_in_pad = fac_cin_pad(C_in). scale may be NULL (== 1). */
int fac_pack_conv_w(const float* v, const float* scale, float* packed, int C_out, int C_in,
                    int K, int C_out_pad, fac_stream_t stream);

/* ConvTranspose1d weight v (C_in, C_out, K = 2*stride) with weight-norm over dim 0 (= C_in,
 * dac/model/encodec.py:163) -> `stride` polyphase 2-tap sub-filters:
 * packed[p][ci][j][co] = v[ci][co][p + stride*(1-j)] * scale[ci],  p < stride, j in {0,1},
 * ci < C_in_pad (rows >= C_in written as zeros, as above).
 * Output phase p of the transposed conv is then a causal 2-tap conv (see fac_conv1d_fwd). */
int fac_pack_convtr_w(const float* v, const float* scale, float* packed, int C_in, int C_out,
                      int stride, int C_out_pad, fac_stream_t stream);

/* The same sub-filters for the all-phases-per-workgroup launch (fac_conv_desc.row_phases = stride): rows r = (co, p), phase
 * fastest, in tiles of 128 rows holding cpt = 128 / stride channels each (rows cpt * stride .. 127 of a tile are zero):
 * packed[ci][j][tile * 128 + (co % cpt) * stride + p] = v[ci][co][p + stride * (1 - j)] * scale[ci], tile = co / cpt;
 * the buffer is (fac_cin_pad(C_in), 2, fac_convtr_rows(C_out, stride)) floats, written completely. */
static inline int fac_convtr_rows(int C_out, int stride) {
  const int cpt = 128 / stride;
  return ((C_out + cpt - 1) / cpt) * 128;
}
int fac_pack_convtr_w_rows(const float* v, const float* scale, float* packed, int C_in, int C_out, int stride,
                           fac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K1-K4  Conv1d / polyphase ConvTranspose1d as implicit GEMM on fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), with fused Snake prologue, bias, Snake/tanh epilogue and
 * residual add.  Replaces SConv1d.forward (dac/model/encodec.py:212-228: pad1d + F.conv1d),
 * SConvTranspose1d.forward (:248-270), snake (dac/nn/layers.py:18-24), ResidualUnit's skip
 * add (dac/model/dac.py:38-42) and nn.Conv1d / nn.Linear calls of the quantizer.
 *
 *   y[b, co, (t*y_tstride + y_toff)] = act( snake_out( bias[co] +
 *        sum_{ci,k} w[phase][ci][k][co] * snake_in( xpad[b, ci, t*stride + k*dilation - pad_left] ) ) )
 *        + res[b, co, ...]
 *
 * xpad is x extended by reflection (FAC_PAD_REFLECT: index -j -> j on the left, T-1+j -> T-1-j
 * on the right, zero where the mirror falls outside x as pad1d :104-111 does) or zeros.
 * ---------------------------------------------------------------------------------------- */
#define FAC_CONV_WS_BYTES (32ll << 20)

typedef struct fac_conv_desc {
  const float* x;         /* (B, C_in, T_in); element (b,c,t) at x[b*x_bs + c*x_cs + t] */
  const float* w;         /* packed weights, see fac_pack_conv_w / fac_pack_convtr_w */
  const float* bias;      /* (C_out) or NULL */
  const float* alpha_in;  /* (C_in) Snake alpha applied to x on load, or NULL */
  const float* alpha_out; /* (C_out) Snake alpha applied after bias, or NULL */
  const float* res;       /* residual, same indexing as y, or NULL */
  float* y;               /* element (b,c,u) at y[b*y_bs + c*y_cs + u]; may be NULL when y2 is given */
  float* y2;              /* optional second output, same indexing: y2 = snake(y, alpha_y2) -- the
                             pre-activated copy the next Snake->conv consumes, so that consumer can
                             stage its input by pure LDS-DMA (no VALU work beside the MFMAs) */
  const float* alpha_y2;  /* (C_out) Snake alpha of y2; required iff y2 != NULL */
  const float* w_k1;      /* optional fused ResidualUnit tail (dac/model/dac.py:33-34): packed 1x1 weights
                             (fac_pack_conv_w of (C, C, 1)); then y = w_k1 * snake(conv + bias, alpha_out)
                             + bias_k1 + res.  Needs C_in == C_out in {64, 96, 128}, K = 7, stride 1. */
  const float* bias_k1;   /* (C_out) bias of the 1x1 conv, or NULL */
  int64_t x_bs, x_cs;
  int64_t y_bs, y_cs;
  int32_t B, C_in, T_in, C_out, C_out_pad;
  int32_t T_out;          /* number of t positions computed per phase */
  int32_t K, stride, dilation, pad_left, pad_mode;
  int32_t n_phase;        /* 1 for Conv1d; = upsampling stride for polyphase ConvTranspose1d */
  int32_t y_tstride;      /* 1 for Conv1d; = n_phase for ConvTranspose1d (y_toff = phase) */
  int32_t phase_shift;    /* non-causal ConvTranspose1d trims `phase_shift` samples on the LEFT
                             (dac/model/encodec.py:265-269): phase p then lands at offset p - shift, and
                             phases with p < shift read x[t], x[t+1] instead of x[t-1], x[t]; 0 = causal */
  int32_t act;            /* FAC_ACT_* applied last (before residual) */
  int32_t w_batched;      /* 0: shared weights; 1: per-b weights at w + b*w_bs (attention-style) */
  int64_t w_bs;
  /* Optional scratch for launches with few output columns (B*T_out <= 128: streaming hops, per-clip
   * Linears): lets the library split the C_in*K reduction across workgroups (conv1d_skinny.hip).  At least
   * FAC_CONV_WS_BYTES, owned by ONE stream at a time (partial sums live there between the two kernels of a
   * launch).  NULL: the tiled kernel is used for every shape. */
  void* ws;
  int64_t ws_bytes;
  /* Optional: the same weights in the split-bf16 layout of fac_pack_conv_w_split (K = 5 / 7) or fac_pack_gemm_w_split
   * (K = 1 / 2).  When given and the shape qualifies (stride 1, no Snake prologue, C_in % 16 == 0 resp. % 32 == 0, enough
   * columns), the conv runs on the bf16 matrix pipe with fp32-grade operand splitting (conv1d_bsplit.hip /
   * conv1d_gemm_split.hip); `w` is still required for every other shape. */
  const void* w_split;
  /* Optional two-level taps (0 = plain): tap k = k2 * K1 + k1 reads the input at offset k2 * dilation2 + k1 * dilation
   * (K % K1 == 0).  A (3, k) Conv2d over a row-concatenated (time, frequency) signal -- the multi-resolution
   * discriminator, dac/model/discriminator.py:101-170 -- is such a conv along the concatenated axis: K1 = k taps along
   * frequency, dilation2 = the row pitch.  Runs on the generic fp32-MFMA tile. */
  int32_t K1, dilation2;
  /* Optional (0 / 1 = off): causal ConvTranspose1d with ALL `row_phases` = stride output phases computed by one workgroup.
   * The GEMM's output rows are (channel, phase) pairs, phase fastest, 128 / row_phases channels per 128-row tile
   * (weights from fac_pack_convtr_w_rows, C_out_pad = its padded row count, n_phase = 1, K = 2, pad_left = 1, y_tstride = 1,
   * T_out = input length); the epilogue interleaves the phases through LDS and stores CONTIGUOUS runs of y (B, C_out,
   * T_out * row_phases).  (The polyphase launch, n_phase = stride, lets every phase store with stride `stride`: partial
   * cache lines, measured 2.9x the algorithmic HBM traffic.) */
  int32_t row_phases;
  /* Optional "P8" operands: an activation tensor as three bf16 planes hi / mid / lo with hi + mid + lo == value exactly
   * (round-to-nearest splits), each plane laid out [b][c / 8][t][8 channels] (16 bytes per (8-channel group, time step); C a
   * multiple of 8; planes `*_plane_bytes` apart; every plane is CLOSED BY ONE ZERO UNIT of 16 bytes at offset B * C/8 * T * 16,
   * which consumers read for columns outside the signal, so a plane takes (B * C/8 * T + 1) * 16 bytes).  That is the operand format of the split-bf16 kernels' LDS stages, so a
   * consumer copies it (16 B per lane, no arithmetic beside the matrix instructions) instead of loading fp32 and splitting it:
   * in-kernel splitting costs every split kernel 15 - 50 % of its matrix-pipe time, because vector ALU instructions of the
   * staging waves and the MFMAs of the same SIMD do not overlap.
   *   x_p8  : the INPUT in P8 (then `x` may be NULL); taken by the K = 3 / 5 / 7 split kernel (C_in % 16 == 0) and by the 1- / 2-tap
   *           split GEMM kernel (plain, strided and all-phases transposed launches; C_in % 8 == 0), where both operands then move by
   *           LDS-DMA.
   *   y2_p8 : the pre-activated second output snake(y, alpha_y2) written in P8 instead of (or beside) fp32 `y2`. */
  const void* x_p8;
  int64_t x_p8_plane_bytes;
  void* y2_p8;
  int64_t y2_p8_plane_bytes;
  /* Optional (0 = off): launches the streaming kernels take (weights resident in LDS, inputs straight from global memory) may
   * split BOTH operands into three bf16 planes inside the kernel (weights once per workgroup, inputs per load) and run on the
   * bf16 matrix pipe with fp32-grade results instead of fp32 MFMAs: k = 1 with C_in == C_out in {64, 96, 128, 192, 256, 384} and
   * many columns; with fp32 weights `w` of fac_pack_convtr_w_rows also the all-phases ConvTranspose1d with stride 2 (row_phases
   * = 2), and with `w` of fac_pack_conv_w the k = 4 stride-2 conv, each with C_in * taps <= 384 (conv1d_pw_split.hip).  Needs only
   * `w`; ignored elsewhere.  The host side sets it wherever it hands `w_split` to the other layers. */
  int32_t pw_split;
} fac_conv_desc;

int fac_conv1d_fwd(const fac_conv_desc* d, fac_stream_t stream);
/* x (B, C, T) fp32 [-> snake(x, alpha) when alpha != NULL] -> the P8 planes described at fac_conv_desc.x_p8
 * (out: 3 planes of (B * (C / 8) * T + 1) * 16 bytes one after the other, the zero unit included).  C % 8 == 0. */
int fac_to_p8(const float* x, const float* alpha, void* out, int B, int C, int T, fac_stream_t stream);
/* Weights (C_out, C_in, K) [* scale per output channel, as fac_pack_conv_w] -> three bf16 planes hi/mid/lo with
 * hi + mid + lo == w exactly, laid out per (64-channel tile, 16-input-channel stage) for LDS-DMA.
 * `out` must hold fac_conv_w_split_bytes(C_out, C_in, K) bytes. */
int64_t fac_conv_w_split_bytes(int C_out, int C_in, int K);
int fac_pack_conv_w_split(const float* v, const float* scale, void* out, int C_out, int C_in, int K,
                          fac_stream_t stream);
/* Split weights of the few-output-channel 9- / 3-tap convs (conv1d_bsplit2.hip: C_out <= 32, K1 = 9 stride 1 / 2 or K1 = 3 stride
 * 1, plain or two-level taps K = K2 * K1 -- the (3, 9) / (3, 3) Conv2d stacks of dac/model/discriminator.py:101-170): v (C_out,
 * C_in, K) [* scale per C_out] as bf16 planes per (32-channel tile, 8 virtual channels).  Passed as fac_conv_desc.w_split. */
int64_t fac_conv_w_split2_bytes(int C_out, int C_in, int K, int K1);
int fac_pack_conv_w_split2(const float* v, const float* scale, void* out, int C_out, int C_in, int K, int K1, fac_stream_t stream);
/* Split weights of the 1- and 2-tap convs (conv1d_gemm_split.hip): rows r < R (output channels, or the (channel, phase) rows of
 * fac_pack_convtr_w_rows), element (r, ci, k) read at v[r * row_stride + ci * ci_stride + k * k_stride] [* row_scale[r]], written
 * as three bf16 planes per (128-row tile, 32-channel chunk) in the kernel's swizzled LDS image (one flat LDS-DMA copy per
 * stage).  C_in is padded to a multiple of 32 with zeros.  Passed as fac_conv_desc.w_split of a K = 1 / K = 2 launch.
 * in_stride S > 1: a strided conv with S < K <= 2 S taps (the encoder's k = 2 s downsampling convs, the period discriminators'
 * (5, 1) stride-3 convs): stored as 2 taps over S phase sub-signals x[S u + p] per 32-channel chunk; the launch passes
 * stride = S, K = the conv's tap count. */
int64_t fac_gemm_w_split_bytes(int R, int C_in, int K, int in_stride);
int fac_pack_gemm_w_split(const float* v, int64_t row_stride, int64_t ci_stride, int64_t k_stride, const float* row_scale, void* out,
                          int R, int C_in, int K, int in_stride, fac_stream_t stream);
/* Which kernel instantiation fac_conv1d_fwd picks for this descriptor: returns its id (>= 0) and
 * writes a printable name; lets a profiler attribute per-launch timings without re-deriving
 * the tile-selection rule.  Ids: 0-6, 8 MFMA tile shapes, 7 fused ResidualUnit, 9 VALU kernel for C_out <= 2, 10 split-reduction
 * kernel (<= 640 columns), 11 split-bf16 kernel (k = 5 / 7), 12 store-stream kernel for C_in = 1, 13 channel-split kernel for
 * C_out <= 2 over few tiles, 14 streaming k = 1 kernel (weights resident in LDS, C <= 384), 15 split-bf16 GEMM kernel (k = 1 / 2),
 * 16 split-bf16 kernel for few output channels (conv1d_bsplit2.hip). */
int fac_conv1d_variant(const fac_conv_desc* d, char* name, int name_len);

/* Standalone Snake  y = x + sin(alpha*x)^2 / (alpha + 1e-9)  (dac/nn/layers.py:18-33) for the
 * places where it cannot be fused; alpha (C). In-place allowed. */
int fac_snake_fwd(const float* x, const float* alpha, float* y, int B, int C, int T,
                  fac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K5  SLSTM (dac/model/encodec.py:272-288): nn.LSTM(H, H, L) over time with zero initial
 * state, gates ordered i,f,g,o, then + x (skip).
 * Work buffers are channel-major with the batch innermost and padded to a multiple of 32:
 *   BP = fac_pad32(B);  xT/yT: (H, T, BP);  pre: (4H, T, BP);  c: (H, BP).
 * (each channel's (t, b) plane is contiguous, so W_ih x + b is one GEMM with C = H, "T" = T*BP)
 * ---------------------------------------------------------------------------------------- */

/* (B, H, T) -> (H, T, BP), zero-filling batch columns B..BP-1. */
int fac_lstm_to_time_major(const float* x, float* xT, int B, int H, int T, fac_stream_t stream);
/* out(B,H,T) = yT(H,T,BP) transposed back + skip(B,H,T) (skip may be NULL); if alpha (H) is given the
 * Snake that follows the SLSTM in the model (dac/model/dac.py:95,112) is applied on the way out. */
int fac_lstm_from_time_major(const float* yT, const float* skip, const float* alpha, float* out, int B, int H, int T,
                             fac_stream_t stream);
/* W_hh (4H, H) -> layout streamed by the recurrent kernel (same element count). */
int fac_pack_lstm_whh(const float* w_hh, float* packed, int H, fac_stream_t stream);
/* One layer's recurrence: for t in 0..T-1:  gates = pre[t] + W_hh h_{t-1};  c,h update;
 * yT[:, t] = h_t.  pre already holds W_ih x_t + b_ih + b_hh.  c is scratch of 3*H*BP floats
 * (cell state + two copies of h in MFMA fragment order).
 * H must be a multiple of 64. */
int fac_lstm_layer_fwd(const float* pre, const float* whh_packed, float* yT, float* c, int T,
                       int H, int BP, fac_stream_t stream);
/* Same recurrence continued from a carried state (streaming inference): `c` is the scratch
 * of an earlier call on the same stream of frames, `step0` the number of steps already taken
 * (0 = zero initial state, identical to fac_lstm_layer_fwd). */
int fac_lstm_layer_fwd_from(const float* pre, const float* whh_packed, float* yT, float* c, int T,
                            int H, int BP, int64_t step0, fac_stream_t stream);
/* Training-mode recurrence: additionally stores the activated gates (4H, T, BP) and the cell states (H, T, BP)
 * that back-propagation through time needs (both NULL: identical to fac_lstm_layer_fwd_from). */
int fac_lstm_layer_fwd_train(const float* pre, const float* whh_packed, float* yT, float* c, float* gates_save,
                             float* c_save, int T, int H, int BP, int64_t step0, fac_stream_t stream);
/* Elementwise part of one BPTT step: dh = dy_t + rec (rec = W_hh^T dgates_{t+1}, NULL at the last step), gate
 * derivatives from the saved activations -> dgates_t (rows of stride rs), carried dc (H, BP) (first = last time step). */
int fac_lstm_gate_bwd(const float* dy_t, const float* rec, const float* gates_t, const float* c_t, const float* c_prev,
                      float* dc, float* dgates_t, int H, int BP, int64_t rs, int first, fac_stream_t stream);
/* Back-propagation through time of one layer in ONE call: for t = T-1 .. 0 the recurrent product W_hh^T dgates_{t+1} (weights from
 * fac_pack_lstm_whh_t, same element count as W_hh) and the gate derivatives.  dyT (H, T, BP): gradient w.r.t. the layer's h sequence;
 * gates (4H, T, BP) / cs (H, T, BP): what fac_lstm_layer_fwd_train saved; dgates (4H, T, BP): out.  scratch: 13 * H * BP floats. */
int fac_pack_lstm_whh_t(const float* w_hh, float* packed, int H, fac_stream_t stream);
int fac_lstm_layer_bwd(const float* dyT, const float* whh_t_packed, const float* gates, const float* cs, float* dgates, float* scratch,
                       int T, int H, int BP, fac_stream_t stream);

/* The same recurrence and its BPTT as ONE launch per layer (lstm_persist.hip): the H/8 workgroups stay resident for all T steps
 * with their slice of W_hh in registers and exchange h_t (dgates_t) through device-wide flags.  Covers zero-initial-state layers
 * with fac_lstm_persist_ok(H, B) != 0 (H in {512, 1024, 1536}, B <= 32, H/8 <= CUs, the grids of the forward and backward kernels co-resident by the runtime's occupancy figure, no earlier timeout); B = number of real batch columns, only the
 * column blocks ceil(B/16)*16 are computed and written (the caller zero-fills the padded columns of yT / saves / dgates).
 * whh16: fac_pack_lstm_whh16(W_hh, out (4H*H floats), H, transposed = 0 forward / 1 BPTT).  With NC = 16 * ceil(B/16): hfrag is
 * T * H * NC floats of scratch, fac_lstm_layer_bwd_persist's scratch (4 + 4 * T) * H * NC floats (one fresh exchange region per
 * step).  One resident layer per stream at a time (stream order guarantees it). */
int fac_lstm_persist_ok(int H, int B);
/* 1 when `stream` has (or can still get) one of the process's 64 exchange-flag slots; 0 -> run the per-step kernels on it.
 * The resident launches of one process are serialised on the device (an event chain across streams) and refuse to launch when
 * the runtime's occupancy figure says the H/8 workgroups would not be co-resident. */
int fac_lstm_persist_stream_ok(fac_stream_t stream);
int fac_pack_lstm_whh16(const float* w_hh, float* packed, int H, int transposed, fac_stream_t stream);
int fac_lstm_layer_fwd_persist(const float* pre, const float* whh16, float* hfrag, float* yT, float* gates_save, float* c_save,
                               int T, int H, int B, int BP, fac_stream_t stream);
int fac_lstm_layer_bwd_persist(const float* dyT, const float* whh16t, const float* gates, const float* cs, float* dgates,
                               float* scratch, int T, int H, int B, int BP, fac_stream_t stream);
/* Forward recurrence of a layer in one launch with W_hh . h on the bf16 matrix pipe (fp32-grade three-way bf16 split of both
 * operands, six products, fp32 accumulation -- the arithmetic of the k = 7 conv kernel): what the fp32 resident kernel cannot do
 * at 17 .. 32 batch columns, where its step is its fp32 MFMA time.  Inference only (no saved gates).  fac_lstm_persist_split_ok(H, B)
 * != 0 for H in {512, 1024, 1536}, B <= 32, H/8 <= CUs.  wsplit: fac_pack_lstm_whh_split(W_hh (4H, H), out (4H*H*6 bytes), H);
 * hsplit: T * H * 32 * 6 bytes of scratch (h_t crosses the device as three bf16 planes in MFMA B-fragment order, one fresh region
 * per step); all 32 columns of yT (H, T, BP) are computed.  Same stream / co-residency rules as the fp32 resident kernels.
 * Replaces the per-step launches of dac/model/encodec.py:272-288 at the benchmark batch. */
int fac_lstm_persist_split_ok(int H, int B);
/* Resident-kernel waits of this process that gave up after 4 s without progress (a workgroup of the grid never ran: the device
 * was shared, or a launch was not co-resident).  Such a launch finishes with meaningless results instead of trapping the
 * context; from then on fac_lstm_persist_ok / fac_lstm_persist_split_ok answer 0 (callers use the per-step kernels).  Reads a
 * host-mapped counter: no synchronisation. */
int fac_lstm_persist_timeouts(void);
/* Tells the current device where that counter lives (one one-thread kernel on `stream`, once per device; refuses inside a stream
 * capture).  Returns 1 when the device is armed.  An un-armed device answers fac_lstm_persist_ok / _split_ok = 0: a launch whose
 * waits may give up must be able to report it.  The launch entry points arm by themselves when they can. */
int fac_lstm_persist_arm(fac_stream_t stream);
/* Device-side hand-over of "a resident wait of this device has given up": dst[0] = 1.0f / 0.0f, ordered on `stream` (no host
 * synchronisation).  The optimiser puts the word behind its per-parameter flags, so it crosses the ranks with them, and then
 * fac_mask_flags_if(flags, n, poison): flags[0..n) = 0 when *poison > 0 -- fac_adamw_step_masked then steps nothing, on every
 * rank, instead of applying a gradient that came out of a bailed-out recurrence (optimizers.py:72-108 has no counterpart: the
 * reference's cuDNN LSTM cannot time out). */
int fac_lstm_abort_flag(float* dst, fac_stream_t stream);
int fac_mask_flags_if(float* flags, int n, const float* poison, fac_stream_t stream);
int fac_pack_lstm_whh_split(const float* w_hh, void* packed, int H, fac_stream_t stream);
int fac_lstm_layer_fwd_persist_split(const float* pre, const void* wsplit, void* hsplit, float* yT, int T, int H, int B, int BP,
                                     fac_stream_t stream);
/* dx = dy * (1 - y^2) */
int fac_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, fac_stream_t stream);
/* Backward of the spectral losses w.r.t. the estimate: da (+)= scale * d|a - b|/da (mode 0) or
 * scale * d|log10 max(a,eps) - log10 max(b,eps)|/da (mode 1); spec_power and stft_frames adjoints. */
int fac_pair_bwd(const float* a, const float* b, float* da, int64_t n, int mode, float eps, float scale, int accumulate,
                 fac_stream_t stream);
int fac_spec_power_bwd(const float* spec, const float* dout, float* dspec, int B, int F, int n_frames, int power,
                       fac_stream_t stream);
int fac_stft_frames_bwd(const float* dframes, float* dwave, int B, int T, int n_win, int n_frames, int hop, int pad,
                        int n_off, fac_stream_t stream);
/* Quantizer backward (dac/nn/quantize.py:55-67).  fac_vq_latent_bwd: d_ze = d_zst (straight-through, may be NULL) +
 * wc[b] * 2/(8T) * (z_e - codebook[idx]) (commitment term) and / or z_st = z_e + (codebook[idx] - z_e) (the out_proj
 * input the weight gradient needs).  fac_vq_codebook_grad: dcb[k] (+)= sum over positions that chose k of
 * wb[b] * 2/(8T) * (codebook[k] - z_e) (deterministic gather, one workgroup per code). */
int fac_vq_latent_bwd(const float* z_e, const float* codebook, const int64_t* codes, int64_t codes_bs, const float* d_zst,
                      const float* wc, float* d_ze, float* z_st, int B, int T, fac_stream_t stream);
int fac_vq_codebook_grad(const float* z_e, const float* codebook, const int64_t* codes, int64_t codes_bs, const float* wb,
                         float* dcb, int B, int T, int Kc, int accumulate, fac_stream_t stream);
/* Backward of fac_layernorm_c_affine: dx (B,C,T), dstyle (B, 2C) = [dgamma | dbeta]; stats: scratch of 2*B*T floats. */
int fac_layernorm_c_affine_bwd(const float* x, const float* style, const float* dout, float* dx, float* dstyle, float* stats,
                               int B, int C, int T, fac_stream_t stream);
/* out[b][i] = a[b][i] * w[b] + sign * c[b][i] over B rows of `per` elements (w and c may be NULL). */
int fac_rows_fma(const float* a, const float* w, const float* c, float* out, int B, int64_t per, float sign,
                 fac_stream_t stream);
/* Optimiser step on flat arenas (optimizers.py:72-108, train.py:362-374).  fac_grad_norm_clip: norm_out[0] = ||g||_2,
 * norm_out[1] = min(1, max_norm / (norm + 1e-6)) (torch clip_grad_norm_), scratch: 1024 floats.  fac_adamw_step: torch
 * AdamW with decoupled weight decay and bias correction for `step` (1-based); clip: norm_out of the call above (the
 * gradient is scaled by clip[1]) or NULL. */
int fac_grad_norm_clip(const float* g, int64_t n, float max_norm, float* scratch, float* norm_out, fac_stream_t stream);
int fac_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t step, const float* clip, fac_stream_t stream);
/* The same step with the set of stepped parameters decided on the device: the arena holds n_params parameters, parameter j =
 * elements [offsets[j], offsets[j+1]) (offsets: n_params + 1 int64 device values, offsets[0] = 0, offsets[n_params] = n);
 * flags[j] > 0 <=> parameter j received a gradient on some rank (the flags ride at the tail of the gradient arena through
 * the data-parallel all-reduce, so no host round trip decides this); such a parameter's step count steps[j] (int32, device)
 * is advanced and it is updated with the bias correction of ITS count (torch keeps one count per parameter); the others are
 * left untouched -- no weight decay, no moment update -- as torch.optim.AdamW skips `grad is None` (optimizers.py:72-108
 * under accelerate's DDP, train.py:362-374).  bc: scratch of 2 * n_params floats. */
int fac_adamw_step_masked(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* offsets, int n_params,
                          const float* flags, int32_t* steps, float* bc, float lr, float beta1, float beta2, float eps,
                          float weight_decay, const float* clip, fac_stream_t stream);
/* dst[j][0..n[j]) = src[j][0..n[j]) for `count` fp32 tensors (host arrays of device pointers and element counts), a few launches
 * for all of them: folds the gradient tensors autograd produced into the optimiser's flat gradient arena (the arena is what
 * train.py:362-374's clip_grad_norm_ + optimizer.step() read). */
int fac_gather_copy(const void* const* src, void* const* dst, const int64_t* n, int count, fac_stream_t stream);
/* Training path of the FA-quantizer's side branches (modules/wavenet.py gate, modules/style_encoder.py Mish / GLU /
 * masked mean, modules/attentions.py attention): elementwise backward kernels, and attention with the probability
 * matrix P (B, H, T, T) kept in HBM so that dropout on it and the softmax backward are row kernels.
 * q, k, v, o are (B, H*dk, T); mask (B, T) float 0/1 or NULL. */
int fac_gate_bwd(const float* a, const float* d, float* da, int B, int C, int T, fac_stream_t stream);
int fac_mish_fwd(const float* x, float* y, int64_t n, fac_stream_t stream);
int fac_mish_bwd(const float* x, const float* d, float* dx, int64_t n, fac_stream_t stream);
int fac_glu_bwd(const float* a, const float* d, float* da, int B, int C, int T, fac_stream_t stream);
int fac_mul_scaled(const float* a, const float* b, float* out, float scale, int64_t n, fac_stream_t stream);
int fac_masked_mean_bwd(const float* dout, const float* mask, float* dx, int B, int C, int T, fac_stream_t stream);
int fac_attention_probs(const float* q, const float* k, const float* mask, float* P, int B, int H, int dk, int T,
                        fac_stream_t stream);
int fac_attention_pv(const float* P, const float* v, float* o, int B, int H, int dk, int T, fac_stream_t stream);
int fac_attention_bwd_pv(const float* P_used, const float* v, const float* dO, float* dv, float* dP, int B, int H, int dk,
                         int T, fac_stream_t stream);
int fac_attention_bwd_qk(const float* P, float* dP, const float* q, const float* k, const float* mask, float* dq, float* dk_,
                         int B, int H, int dk, int T, fac_stream_t stream);
/* Backward of fac_aa_snakebeta_fwd: dx (B,C,T), dalpha / dbeta (C) w.r.t. the log-scale parameters;
 * scratch: 2 * B * C * ceil(T/256) floats. */
int fac_aa_snakebeta_bwd(const float* x, const float* alpha_log, const float* beta_log, const float* filter12, const float* dy,
                         float* dx, float* dalpha, float* dbeta, float* scratch, int B, int C, int T, fac_stream_t stream);
/* Discriminator path (dac/model/discriminator.py): layout / elementwise kernels; the convolutions run on
 * fac_conv1d_fwd (MPD: 1-D along the folded time axis, period as batch; MRD: 1-D along frequency over three stacked
 * time rows).  A non-NULL dy / non-zero `backward` selects the adjoint.
 *   leaky_relu: out = x > 0 ? x : slope*x   (backward: out = x > 0 ? dy : slope*dy); with pitch > 0 only positions
 *               t % pitch < valid of every length-T row are kept, the rest set to zero (row-concatenated MPD signals)
 *   period_fold: (B, T) -> B*period rows of `pitch` columns laid one after another (L data columns, then zeros),
 *               reflect-extended on the right to L*period samples
 *   zero_insert: (rows, T) -> (rows, (T-1)*stride + 1), zeros between samples (strided conv data gradient)
 *   row_stack3: (rows = B*T, C, F) -> (rows, 3C, F): time rows t-1, t, t+1 stacked (zero outside a clip)
 *   spec_to_rows: one frequency band [f0, f0+Fb) of spec (B, 2*Ft, T) = [re|im] -> (B*T, 2, Fb)
 *   pad_reflect: (B, T) -> (B, pad_l + T + pad_r)
 *   disc_preprocess: z = 0.8 (x - mean)/(max|x - mean| + 1e-9) per clip; stats: 4*B floats kept for the backward */
/* rows_per_group > 0: besides the column mask, rows (position / pitch) with row % rows_per_group >= valid_rows are zero
 * (the all-zero separator rows between clips of the row-concatenated spectrogram layout). */
int fac_leaky_relu(const float* x, const float* dy, float* out, int64_t n, float slope, int T, int pitch, int valid,
                   int rows_per_group, int valid_rows, fac_stream_t stream);
int fac_period_fold(const float* x, float* out, int B, int T, int period, int L, int pitch, int backward, fac_stream_t stream);
int fac_zero_insert(const float* dy, float* up, int64_t rows, int T, int stride, fac_stream_t stream);
int fac_row_stack3(const float* x, float* out, int64_t rows, int T, int C, int F, int backward, fac_stream_t stream);
/* Multi-resolution discriminator in the row-concatenated layout: spec (B, 2*Ft, T) [re | im] band [f0, f0 + Fb) ->
 * cat (2, B*(T+1)*pitch): row r = b*(T+1) + t holds the band's Fb bins followed by zeros; row t = T of every clip is an
 * all-zero separator (the time padding of the (3, k) convs).  backward: the adjoint, accumulating disjoint bands into a
 * zero-initialised dspec. */
int fac_spec_to_cat(const float* src, float* dst, int B, int Ft, int T, int f0, int Fb, int pitch, int backward, fac_stream_t stream);
int fac_spec_to_rows(const float* src, float* dst, int B, int Ft, int T, int f0, int Fb, int backward, fac_stream_t stream);
int fac_pad_reflect(const float* x, float* out, int B, int T, int pad_l, int pad_r, fac_stream_t stream);
int fac_disc_preprocess(const float* x, const float* dz, float* out, float* stats, int B, int T, fac_stream_t stream);
/* F.cross_entropy (mean) over rows: logits (N, C), labels (N) int64.  loss (1 float) and / or dlogits = (softmax -
 * onehot) * grad_scale; scratch: N floats. */
int fac_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits, float* scratch, int64_t N,
                      int C, float grad_scale, fac_stream_t stream);
/* FocalLoss on the mean cross entropy (losses.py:264-276; train.py:153 gamma = 2): out2[0] = (1 - exp(-ce))^gamma * ce,
 * out2[1] = d out2[0] / d ce.  ce, out2: device scalars. */
int fac_focal_scalar(const float* ce, float* out2, float gamma, fac_stream_t stream);
/* Random-crop batching of train.py:188-212 on the device: dst[b][c][t] = src[b][c][start[b] * scale + t] (0 outside);
 * src (B, C, T_src), dst (B, C, T_dst), start (B) int64 device values in units of `scale` samples. */
int fac_crop_rows(const float* src, float* dst, const int64_t* start, int B, int C, int64_t T_src, int T_dst, int scale,
                  fac_stream_t stream);
/* Left-context buffer of a streaming causal conv: every row of buf (rows x cap) holds
 * [hist columns of history | n_prev columns appended last time]; moves the last `hist` columns to the
 * front (skipped when n_prev == 0) and appends src (rows x n_new, dense) behind them.  hist <= 2048. */
int fac_stream_push(float* buf, const float* src, int64_t rows, int64_t cap, int hist, int n_prev, int n_new,
                    fac_stream_t stream);

/* Weights of a conv's data-gradient conv for the split-bf16 kernels, materialised in one pass: out (C_in, C_out, K)[ci][co][k] =
 * v (C_out, C_in, K)[co][ci][K-1-k] * scale[co] (scale: fac_wn_scale or NULL).  Pack the result with fac_pack_conv_w_split* . */
int fac_flip_transpose_w(const float* v, const float* scale, float* out, int C_out, int C_in, int K, fac_stream_t stream);

/* Batched weight preparation (round 6).  The reference's weight_norm recomputes w = g * v / ||v|| in a pre-forward hook of every
 * conv (dac/model/encodec.py:42-51, dac/nn/layers.py:9-14); here that is one small launch per tensor and layout (fac_wn_scale, then
 * fac_pack_*): ~290 launches of a configs[1] forward, ~1 700 of a train step.  Between fac_prep_begin() and fac_prep_end() the calling
 * thread's fac_wn_scale, fac_pack_conv_w, fac_pack_convtr_w, fac_pack_convtr_w_rows, fac_flip_transpose_w, fac_pack_conv_w_split,
 * fac_pack_gemm_w_split, fac_pack_conv_w_split2 and fac_pack_conv_w_bwd calls are RECORDED (arguments checked, nothing launched) under
 * the phase set by fac_prep_set_phase (0 at begin; a job that reads another job's output needs a higher phase).  fac_prep_end() uploads
 * the job tables and returns a plan id (>= 0; negative = error code).  fac_prep_replay() re-runs every job of the plan from the CURRENT
 * contents of its input tensors -- one launch per (phase, kernel family), phases in order -- with the arithmetic of the single launches
 * (same device functions: results are bit-identical).  The pointers recorded must stay valid for the life of the plan.  fac_prep_free()
 * requires that no replay of the plan is still executing. */
int fac_prep_begin(void);
int fac_prep_set_phase(int phase);
int fac_prep_abort(void);
int fac_prep_end(void);
int fac_prep_replay(int plan, fac_stream_t stream);
int fac_prep_info(int plan, int* n_jobs, int* n_launches);
int fac_prep_free(int plan);

/* ------------------------------------------------------------------------------------------
 * Backward of the conv stack (first kernels of the training step; autograd semantics of
 * dac/model/encodec.py SConv1d / SConvTranspose1d, dac/nn/layers.py snake, weight_norm).
 *
 * bwd_data of a stride-1 conv: run fac_conv1d_fwd over dy with the weights packed by fac_pack_conv_w_bwd
 * (taps flipped, channels transposed; C_out becomes the input-channel axis: rows up to fac_cin_pad(C_out),
 * columns C_in_pad = pad32(C_in), zero-filled by the caller), pad_left = (K-1)*dilation, zero padding,
 * T_out = padded input length; then fac_pad_fold_bwd maps the gradient of the padded signal back to x
 * (reflection: a padded position's gradient is added to the sample it mirrors).  Strided convs (K = 2*stride):
 * the transposed-conv launch on the forward weights (fac_pack_convtr_w), then the same fold.
 * ---------------------------------------------------------------------------------------- */
int fac_pack_conv_w_bwd(const float* v, const float* scale, float* packed, int C_out, int C_in, int K, int C_in_pad,
                        fac_stream_t stream);
int fac_pad_fold_bwd(const float* dxpad, float* dx, int B, int C, int T, int Tp, int pad_left, int pad_mode,
                     fac_stream_t stream);
/* fac_pad_fold_bwd in place for reflect padding, edges only: adds every mirrored padded position of dxpad (B, C, Tp) onto the sample
 * it mirrors (dac/model/encodec.py:96-113 backward); afterwards dx[b][c][j] IS dxpad[b][c][pad_left + j].  Needs T > pad_left +
 * pad_right.  Zero padding needs no call at all (the window is the gradient). */
int fac_pad_fold_edges(float* dxpad, int B, int C, int T, int Tp, int pad_left, fac_stream_t stream);
/* dW (C_out, C_in, K) = sum over (b, t) of dy[b][co][t] * xpad[b][ci][t*stride + k*dilation]; ws: scratch of
 * fac_conv1d_bwd_weight_ws_bytes(...) bytes (partial sums per (b, t) range, added in a fixed order). */
int64_t fac_conv1d_bwd_weight_ws_bytes(int B, int C_in, int C_out, int T_out, int K);
int fac_conv1d_bwd_weight(const float* x, const float* dy, float* dw, void* ws, int64_t ws_bytes, int B, int C_in,
                          int T_in, int C_out, int T_out, int K, int stride, int dilation, int pad_left, int pad_mode,
                          int K1, int dilation2, fac_stream_t stream);
/* The k = 1, stride-1 case with few channels and long signals (the ResidualUnit tails of dac/model/dac.py:25-42 at C = 64 / 96 / 192:
 * dW[co][ci] = sum over (b, t) of dy[b][co][t] * x[b][ci][t]) straight from the fp32 tensors on the fp32 matrix pipe
 * (conv1d_wgrad_k1.hip): both tensors are read once, no operand planes.  The query returns the workspace size (partial sums of the
 * workgroups, added in a fixed order) or -1 when the shape does not run here (channel counts: multiples of 32 in [64, 192];
 * T >= 4096, T % 4 == 0). */
int64_t fac_conv1d_bwd_weight_k1_ws_bytes(int B, int C_in, int C_out, int T);
int fac_conv1d_bwd_weight_k1(const float* x, const float* dy, float* dw, float* db, void* ws, int64_t ws_bytes, int B, int C_in, int C_out,
                             int T, fac_stream_t stream);   /* db: (C_out) bias gradient sum over (b, t) of dy, or NULL */
/* The same kernel for stride-1 convs with few input channels and taps (first layers: the encoder's 1 -> 64 k = 7 conv,
 * dac/model/dac.py:82; the multi-resolution discriminator's 2 -> 32 (3, 9) convs, dac/model/discriminator.py:101-130): the C_in * K
 * columns of dW are shifted views of the input rows.  xpad (B, C_in, Tx) is the conv's PADDED input (the caller pads: left padding,
 * right padding, then zeros up to Tx >= fac_conv1d_bwd_weight_taps_tx);
 * dW (C_out, C_in, K)[co][ci][k] = sum over (b, t < T_out) of dy[b][co][t] * xpad[b][ci][t + (k / K1) * dilation2 + (k % K1) * dilation]
 * (K1 = K: plain taps).  C_out = 32 or 64, C_in * K <= 64, T_out >= 4096; the query returns -1 otherwise.  db as above. */
int64_t fac_conv1d_bwd_weight_taps_tx(int T_out, int K, int K1, int dilation, int dilation2);
int64_t fac_conv1d_bwd_weight_taps_ws_bytes(int B, int C_in, int C_out, int T_out, int K, int K1, int dilation, int dilation2);
int fac_conv1d_bwd_weight_taps(const float* xpad, const float* dy, float* dw, float* db, void* ws, int64_t ws_bytes, int B, int C_in, int Tx,
                               int C_out, int T_out, int K, int K1, int dilation, int dilation2, fac_stream_t stream);
/* The same gradient on the bf16 matrix pipe with fp32-grade operand splitting (conv1d_wgrad_split.hip; same arguments
 * and result layout as fac_conv1d_bwd_weight, error vs fp64 no larger than the fp32 MFMA's).  The workspace query
 * returns -1 for a (K, stride, dilation) the kernel does not cover: use fac_conv1d_bwd_weight then.  K1 / dilation2:
 * two-level taps as in fac_conv_desc (0 = plain); dW stays (C_out, C_in, K) with k = k2 * K1 + k1.  The workspace also
 * holds the pre-split bf16 planes of both operands. */
int64_t fac_conv1d_bwd_weight_split_ws_bytes(int B, int C_in, int T_in, int C_out, int T_out, int K, int stride, int dilation,
                                             int K1, int dilation2);
int fac_conv1d_bwd_weight_split(const float* x, const float* dy, float* dw, void* ws, int64_t ws_bytes, int B, int C_in, int T_in,
                                int C_out, int T_out, int K, int stride, int dilation, int pad_left, int pad_mode, int K1,
                                int dilation2, fac_stream_t stream);
/* The same launch with the bias gradient db[co] = sum over (b, t) of dy folded into the dy operand's split pass (no separate
 * pass over dy); k-major shapes only: C_in * K2 >= 16 and B * C_out <= 65535 (query: fac_conv1d_bwd_weight_split_db_ok). */
int fac_conv1d_bwd_weight_split_db_ok(int B, int C_in, int T_in, int C_out, int T_out, int K, int stride, int dilation, int K1,
                                      int dilation2);
int fac_conv1d_bwd_weight_split_db(const float* x, const float* dy, float* dw, float* db, void* ws, int64_t ws_bytes, int B, int C_in,
                                   int T_in, int C_out, int T_out, int K, int stride, int dilation, int pad_left, int pad_mode, int K1,
                                   int dilation2, fac_stream_t stream);
/* w = g*v/||v|| per row (n_rows x row_len): dv, dg from dW. */
int fac_weight_norm_bwd(const float* v, const float* g, const float* dw, float* dv, float* dg, int n_rows, int row_len,
                        fac_stream_t stream);
/* y = x + sin^2(alpha x)/(alpha + 1e-9): dx (B,C,T) and dalpha (C).  scratch: 32*C floats (partial sums of the
 * two-stage, fixed-order channel reductions). */
int fac_snake_bwd(const float* x, const float* alpha, const float* dy, float* dx, float* dalpha, float* scratch, int B, int C,
                  int T, fac_stream_t stream);
/* The same with its neighbours fused in: dx = add + dy * dsnake/dx (add: the second gradient of a tensor with two consumers, the
 * ResidualUnit skip of dac/model/dac.py:38-42 -- NULL: none), dbias[c] = sum over (b, t) of dx (the bias gradient of the conv
 * that produced x -- NULL: skipped).  scratch: 64*C floats. */
int fac_snake_bwd_fused(const float* x, const float* alpha, const float* dy, const float* add, float* dx, float* dalpha,
                        float* dbias, float* scratch, int B, int C, int T, fac_stream_t stream);
/* The same with the rows of dy `dy_row_stride` (>= T) elements apart: dy is then typically the window [pad_left, pad_left + T) of the
 * padded gradient rows a data-gradient conv wrote, after fac_pad_fold_edges -- no separate un-padding pass over the tensor. */
int fac_snake_bwd_fused_rs(const float* x, const float* alpha, const float* dy, long long dy_row_stride, const float* add, float* dx,
                           float* dalpha, float* dbias, float* scratch, int B, int C, int T, fac_stream_t stream);
/* db[c] = sum over (b, t) of dy; scratch: 32*C floats. */
int fac_bias_grad(const float* dy, float* db, float* scratch, int B, int C, int T, fac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K7  factorized VQ step (dac/nn/quantize.py:34-94 VectorQuantize.forward + the residual
 * bookkeeping of ResidualVectorQuantize.forward :173-193; search identical to
 * quantize/fvq.py:101-116):
 *   z_e = W_in r + b_in (1x1 conv D->8); e = z_e/max(||z_e||,1e-12); c~ = cb/max(||cb||,1e-12);
 *   idx = first argmax_k -( (||e||^2 - 2 e.c~_k) + ||c~_k||^2 );  z_q = cb[idx] (raw rows);
 *   z_st = z_e + (z_q - z_e);  out = W_out z_st + b_out;
 *   zq_acc += out * mask[b];  residual -= out;
 *   loss_part[b][tile] = sum_{d,t in tile} (z_e - z_q)^2   (commitment == codebook value).
 * ---------------------------------------------------------------------------------------- */
typedef struct fac_vq_desc {
  float* residual;        /* (B, D, T) in/out: r <- r - out ; may be NULL to skip the update */
  const float* z_in;      /* (B, D, T) the vectors to quantize (may alias residual) */
  float* zq_acc;          /* (B, D, T) in/out: += out*mask ; NULL to skip */
  float* zq_out;          /* (B, D, T) out (this quantizer's projected output) ; NULL to skip */
  const float* w_in;      /* packed (D, 1, 32) from fac_pack_conv_w (C_out = 8 -> pad 32) */
  const float* b_in;      /* (8) */
  const float* codebook;  /* (Kc, 8) raw */
  const float* w_out;     /* (D, 8) out_proj weight_v rows ([c][d], torch layout (D,8,1)) */
  const float* w_out_scale; /* (D) weight-norm scale g/||v|| from fac_wn_scale, or NULL (== 1) */
  const float* b_out;     /* (D) */
  const float* mask;      /* (B) multiplies out in zq_acc (quantizer dropout), NULL == 1 */
  int64_t* codes;         /* (B, T) at codes[b*codes_bs + t] */
  float* z_e;             /* (B, 8, T) projected latents, or NULL */
  float* loss_part;       /* (B, n_tiles) partial sums of (z_e-z_q)^2, n_tiles = fac_vq_loss_tiles(T) */
  int64_t codes_bs;
  int32_t B, D, T, Kc;
} fac_vq_desc;

int fac_vq_fwd(const fac_vq_desc* d, fac_stream_t stream);
/* Number of time tiles fac_vq_fwd cuts T frames into (= the row length of loss_part; 16 frames per workgroup). */
int fac_vq_loss_tiles(int T);

/* Nearest-code search only, on already-projected latents (N, 8) row-major -> idx (N) int64.
 * The isolated kernel of dac/nn/quantize.py:78-94 / quantize/fvq.py:101-116. */
int fac_vq_search(const float* latents, const float* codebook, int64_t* idx, int64_t N, int Kc,
                  fac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Small fused elementwise / reduction ops of the quantizer (modules/quantize.py:375-454).
 * ---------------------------------------------------------------------------------------- */

/* acts = tanh((a+g)[:, :C]) * sigmoid((a+g)[:, C:])  (modules/commons.py:113-120), a (B,2C,T) -> (B,C,T);
 * g: optional per-clip conditioning (row b at g + b*g_bs, 2C values broadcast over time:
 * modules/wavenet.py:146-155 with gin_channels), or NULL. */
int fac_gate_tanh_sigmoid(const float* a, const float* g, int64_t g_bs, float* out, int B, int C, int T,
                          fac_stream_t stream);
/* Code-embedding sum of the Redecoder (modules/redecoder.py:35-45): out (B, E, T) (+)= sum over n_tab
 * tables (n_tab, V, E) of table_i[codes[b, code_row0 + i, t]]; codes (B, n_codes, T) int64. */
int fac_embed_sum(const int64_t* codes, const float* tables, float* out, int B, int n_tab, int n_codes,
                  int code_row0, int V, int E, int T, int accumulate, fac_stream_t stream);
/* GLU residual: out = res + a[:, :C] * sigmoid(a[:, C:])  (modules/style_encoder.py:26-31) */
int fac_glu_residual(const float* a, const float* res, float* out, int B, int C, int T,
                     fac_stream_t stream);
/* out = a + b (same shape, n elements);  out = a - b - c */
int fac_add(const float* a, const float* b, float* out, int64_t n, fac_stream_t stream);
int fac_sub2(const float* a, const float* b, const float* c, float* out, int64_t n,
             fac_stream_t stream);
/* x[b, c, t] *= mask[b, t]  in place (mask is float 0/1). */
int fac_mul_mask(float* x, const float* mask, int B, int C, int T, fac_stream_t stream);
/* WaveNet skip accumulation (modules/wavenet.py:160-165):
 *   rs (B, 2C, T): x = (x + rs[:, :C]) ; out += rs[:, C:]   (last layer: rs (B,C,T): out += rs) */
int fac_wn_res_skip(const float* rs, float* x, float* out, int B, int C, int T, int last,
                    fac_stream_t stream);
/* 2-head style self-attention core (modules/attentions.py:168-199) on q,k,v (B, H*dk, T):
 * scores = (q/sqrt(dk))^T k, masked_fill(mask==0,-1e4), softmax over keys, out = p v. */
int fac_attention(const float* q, const float* k, const float* v, const float* mask, float* out,
                  int B, int n_heads, int dk, int T, fac_stream_t stream);
/* masked temporal average pool (modules/style_encoder.py:83-91): out[b,c] = sum_t x / sum_t mask */
int fac_masked_mean(const float* x, const float* mask, float* out, int B, int C, int T,
                    fac_stream_t stream);
/* Timbre-conditioned norm (modules/quantize.py:444-449): LayerNorm over C (eps 1e-5, no affine)
 * per (b,t), then * gamma[b,c] + beta[b,c];  style (B, 2C) = [gamma | beta]. */
int fac_layernorm_c_affine(const float* x, const float* style, float* out, int B, int C, int T,
                           fac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K8 / K11  STFT framing + spectral losses (torchaudio MelSpectrogram in
 * modules/quantize.py:228-242; audiotools STFT/mel in dac/nn/loss.py:203-228,294-327).
 * The DFT itself runs through fac_conv1d_fwd as a GEMM against a windowed cos/sin basis.
 * ---------------------------------------------------------------------------------------- */

/* frames[b][n][f] = wave_reflectpad[b][f*hop + n + n_off - pad],  n < n_win, f < n_frames
 * (centre=True framing: pad = n_fft/2, reflect; n_off = offset of the first non-zero window
 * tap inside the n_fft frame). wave (B, T). */
int fac_stft_frames(const float* wave, float* frames, int B, int T, int n_win, int n_frames,
                    int hop, int pad, int n_off, fac_stream_t stream);
/* spec (B, 2F, n_frames) rows [0,F) = Re, [F,2F) = Im  ->  out (B, F, n_frames):
 * power == 2: re^2+im^2 ; power == 1: sqrt(re^2+im^2). */
int fac_spec_power(const float* spec, float* out, int B, int F, int n_frames, int power,
                   fac_stream_t stream);
/* Deterministic two-stage reduction:  out[0] = scale * sum_i f(a_i, b_i)
 *   mode 0: |a-b| ; mode 1: |log10(max(a,eps)) - log10(max(b,eps))| (pow folded in by caller);
 *   mode 2: (a-b)^2.  scratch >= 1024 floats. */
int fac_reduce_pair(const float* a, const float* b, float* out, float* scratch, int64_t n,
                    int mode, float eps, float scale, int accumulate, fac_stream_t stream);

/* K13  anti-aliased SnakeBeta activation of the predictor heads (alias_free_torch/act.py:24-29:
 * 2x Kaiser-sinc upsample -> SnakeBeta with log-scale alpha/beta, modules/quantize.py:77-90 -> 2x
 * low-pass downsample), fused into one pass.  x, y (B, C, T); alpha_log, beta_log (C); filter12 = the
 * 12-tap filter both resamplers share (alias_free_torch/filter.py:27-58). */
int fac_aa_snakebeta_fwd(const float* x, const float* alpha_log, const float* beta_log,
                         const float* filter12, float* y, int B, int C, int T, fac_stream_t stream);

/* losses.py:84:  out[0] (+)= scale * sum_{b,t} sqrt( mean_m (log(|a|+eps) - log(|b|+eps))^2 ),
 * a, b (B, M, T); scratch >= 1024 floats; deterministic two-stage reduction. */
int fac_logdiff_rms(const float* a, const float* b, float* out, float* scratch, int B, int M, int T,
                    float eps, float scale, int accumulate, fac_stream_t stream);

/* Backward of fac_logdiff_rms w.r.t. its SECOND operand (the estimate of losses.py:65-89 reconstruction_loss):
 * db (+)= scale * d/db sum_{b,t} sqrt( mean_m (log(|a|+eps) - log(|b|+eps))^2 ). */
int fac_logdiff_rms_bwd(const float* a, const float* b, float* db, int B, int M, int T, float eps, float scale,
                        int accumulate, fac_stream_t stream);

static inline int fac_pad32(int n) { return (n + 31) & ~31; }
/* packed weights carry zero rows up to a multiple of 48 input channels (lcm of the kernel's
 * channels-per-stage choices), so a partially filled last stage multiplies zeros */
static inline int fac_cin_pad(int c) { return ((c + 47) / 48) * 48; }

/* ------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange as an explicit RCCL collective over a flat fp32 arena (what
 * DistributedDataParallel does behind train.py:110-111 / :289 / :361 -- bucketed all-reduce of the gradients, mean over the ranks).
 * facodec_amd/optim.py exchanges its arenas through torch.distributed by default (backend "nccl" is RCCL on ROCm) and through this
 * entry point with FAC_NATIVE_RCCL=1; a caller without torch binds it directly.  RCCL is resolved at run time (the librccl.so.1
 * already in the process, else dlopen): the library has no link-time dependency on it.
 *   fac_rccl_unique_id     one rank creates the 128-byte id and ships it to the others by any means (the Python side: the
 *                          torch.distributed store);
 *   fac_rccl_comm_init     collective over the ranks; binds the calling thread's current device;
 *   fac_allreduce_arena    in place over `count` floats on `stream` (asynchronous; average != 0: mean over the ranks, ncclAvg);
 *                          calls on one communicator must be issued in the same order on every rank.
 * ---------------------------------------------------------------------------------------- */
int fac_rccl_available(void);
int fac_rccl_unique_id(void* id128);
int fac_rccl_comm_init(void** comm, const void* id128, int nranks, int rank);
int fac_rccl_comm_destroy(void* comm);
int fac_allreduce_arena(void* comm, float* arena, int64_t count, int average, fac_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FACODEC_HIP_H */
