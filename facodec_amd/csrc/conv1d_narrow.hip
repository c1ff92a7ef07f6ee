// Convs with one or two output channels (the decoder's final 96 -> 1, k = 7 + tanh: dac/model/dac.py:158-159).
// An MFMA tile would waste 31/32 of its rows, and the layer is HBM-bound anyway (reads 590 MB at B = 32,
// writes 6 MB): plain VALU kernel, one workgroup per (batch, 1024-step tile), input rows staged through LDS
// eight channels at a time, four consecutive outputs per thread (a 10-value sliding window per channel).
// Rows are padded to a multiple of 4 floats so that a thread's window starts 16-byte aligned: for dilation 1 and K <= 9 the
// window is three ds_read_b128 (lanes 16 B apart: conflict-free); element-wise reads at a 16-byte lane stride are 4-way bank
// conflicts, which made the 96 -> 1 layer LDS-bound at 1.6 TB/s of input.
// Two-level taps (fac_conv_desc.K1: the (3, k) Conv2d layers of the multi-resolution discriminator over a row-concatenated
// signal) run as K / K1 VIRTUAL channels per real one, each the same input row read k2 * dilation2 columns further on (ConvArgs.K2v):
// the 32 -> 1 output conv and the 32 -> 2 data gradient of the first layer then are plain 96-channel, 3- / 9-tap convs here
// instead of 32-row MFMA tiles with 1 or 2 live rows (round 6: 213 -> ~90 us per launch for the first layer's data gradient).
#include "conv1d_mfma.h"

namespace fac {

constexpr int NARROW_TT = 1024;
constexpr int NARROW_CIC = 8;

template <int CO>
__global__ __launch_bounds__(256) void conv1d_narrow_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.KV, dil = a.dil;                  // taps per (virtual) channel
  const int CV = a.CV, K2v = a.K2v;
  const bool two = K2v > 1;                         // two-level taps: zero padding only (fac_conv1d_fwd)
  const int halo = (K - 1) * dil;
  const int XW = (NARROW_TT + halo + 3) & ~3;
  float* xs = sm;                                   // [CIC][XW]
  float* ws = sm + NARROW_CIC * XW;                 // [CIC][K][CO]
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * NARROW_TT;
  const float* xg = a.x + (long long)b * a.x_bs;
  const int tid = threadIdx.x;
  float acc[CO][4];
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;

  // Staging slots: slot (r, s) of this thread is column s * 256 + tid of channel row r -- the column -> input index map
  // (padding / reflection) is the same for every row and chunk and is resolved once.  The loads of chunk c + 1 are issued
  // before chunk c is multiplied (register double buffer), so HBM latency hides behind the FMAs of the same workgroup.
  constexpr int SPR = (NARROW_TT + 64 + 255) / 256;        // slots per row (halo <= 64 columns)
  int s_idx[SPR];
#pragma unroll
  for (int sl = 0; sl < SPR; ++sl) {
    const int c = sl * 256 + tid;
    int idx = -1;
    if (c < XW) {
      const int tin = t0 - a.pad_left + c;
      if (two) idx = tin;                          // the virtual channel's shift comes on top: bounds are checked per row
      else if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
      else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
    }
    s_idx[sl] = idx;
  }
  float v[NARROW_CIC][SPR];
  auto load_chunk = [&](int ci0) {
#pragma unroll
    for (int r = 0; r < NARROW_CIC; ++r) {
      const int vc = ci0 + r;                      // uniform
      const int ci = two ? vc / K2v : vc;
      const int off = two ? (vc - ci * K2v) * a.dil2 : 0;
      const float* xrow = xg + (long long)ci * a.x_cs;
#pragma unroll
      for (int sl = 0; sl < SPR; ++sl) {
        int idx = s_idx[sl];
        bool ok = idx >= 0;
        if (two) {
          idx += off;
          ok = sl * 256 + tid < XW && idx >= 0 && idx < a.T_in;
        }
        v[r][sl] = (vc < CV && ok) ? xrow[idx] : 0.f;
      }
    }
  };
  load_chunk(0);

  for (int ci0 = 0; ci0 < CV; ci0 += NARROW_CIC) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NARROW_CIC; ++r) {
      const int vc = ci0 + r;
      const bool sn = a.alpha_in != nullptr && vc < CV;
      const float al = sn ? a.alpha_in[two ? vc / K2v : vc] : 0.f;
      const float inv = sn ? snake_inv(al) : 0.f;
#pragma unroll
      for (int sl = 0; sl < SPR; ++sl) {
        const int c = sl * 256 + tid;
        if (c < XW) xs[r * XW + c] = sn ? snake_apply(v[r][sl], al, inv) : v[r][sl];
      }
    }
    for (int i = tid; i < NARROW_CIC * K * CO; i += 256) {
      const int r = i / (K * CO), rem = i - r * (K * CO);
      const int k = rem / CO, c = rem - k * CO;
      const int vc = ci0 + r;                      // row (vc * KV + k) of the packed weights IS row (ci * K + k2 * K1 + k)
      ws[i] = vc < CV ? a.w[((long long)vc * K + k) * a.C_out_pad + c] : 0.f;
    }
    __syncthreads();
    if (ci0 + NARROW_CIC < CV) load_chunk(ci0 + NARROW_CIC);
    if (dil == 1 && K <= 9) {
#pragma unroll 2
      for (int r = 0; r < NARROW_CIC; ++r) {
        const float4* xr = reinterpret_cast<const float4*>(xs + r * XW + tid * 4);
        const float* wr = ws + r * K * CO;
        const float4 q0 = xr[0], q1 = xr[1], q2 = xr[2];
        const float win[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if (k >= K) break;
#pragma unroll
          for (int c = 0; c < CO; ++c) {
            const float w = wr[k * CO + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] = fmaf(w, win[k + j], acc[c][j]);
          }
        }
      }
      continue;
    }
#pragma unroll 2
    for (int r = 0; r < NARROW_CIC; ++r) {
      const float* xr = xs + r * XW + tid * 4;
      const float* wr = ws + r * K * CO;
      for (int k = 0; k < K; ++k) {
        const float x0 = xr[k * dil], x1 = xr[k * dil + 1], x2 = xr[k * dil + 2], x3 = xr[k * dil + 3];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
          const float w = wr[k * CO + c];
          acc[c][0] = fmaf(w, x0, acc[c][0]);
          acc[c][1] = fmaf(w, x1, acc[c][1]);
          acc[c][2] = fmaf(w, x2, acc[c][2]);
          acc[c][3] = fmaf(w, x3, acc[c][3]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    if (c >= a.C_out) continue;
    const float bs = a.bias ? a.bias[c] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + tid * 4 + j;
      if (t >= a.T_out) continue;
      float v = acc[c][j] + bs;
      if (a.alpha_out) v = snake_apply(v, a.alpha_out[c], snake_inv(a.alpha_out[c]));
      if (a.act != FAC_ACT_NONE) v = apply_act_slow(v, a.act);
      a.y[(long long)b * a.y_bs + (long long)c * a.y_cs + t] = v;
    }
  }
}

// One input channel, many output channels (the encoder's first conv, 1 -> 64, k = 7: dac/model/dac.py:84): 6 MB in, 2 x 393 MB
// out at B = 32 -- a store stream.  One workgroup per (batch, 1024-step tile); the input window lives in registers (three
// aligned ds_read_b128 per thread), the taps of one output channel come through the scalar cache (uniform addresses), and each
// thread writes 16 bytes of y and of the pre-activated copy y2 per channel (consecutive threads: contiguous 4 KB rows).
constexpr int CIN1_KMAX = 9;

__global__ __launch_bounds__(256) void conv1d_cin1_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[NARROW_TT + 16];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * NARROW_TT;
  const int tid = threadIdx.x;
  const int K = a.K;
  const float* xg = a.x + (long long)b * a.x_bs;
  for (int i = tid; i < NARROW_TT + 16; i += 256) {
    const int tin = t0 - a.pad_left + i;
    int idx;
    if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
    else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
    xs[i] = (i < NARROW_TT + K - 1 && idx >= 0) ? xg[idx] : 0.f;
  }
  __syncthreads();
  const int t = t0 + 4 * tid;
  if (t >= a.T_out) return;
  const float4* xr = reinterpret_cast<const float4*>(xs + 4 * tid);
  const float4 q0 = xr[0], q1 = xr[1], q2 = xr[2];
  const float win[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
  float* yg = a.y ? a.y + (long long)b * a.y_bs : nullptr;
  float* y2g = a.y2 ? a.y2 + (long long)b * a.y_bs : nullptr;
  const bool full = t + 3 < a.T_out && (a.y_cs & 3) == 0 && (a.y_bs & 3) == 0 &&
                    (!yg || (reinterpret_cast<unsigned long long>(a.y) & 15) == 0) &&
                    (!y2g || (reinterpret_cast<unsigned long long>(a.y2) & 15) == 0);
  for (int co = 0; co < a.C_out; ++co) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CIN1_KMAX; ++k) {
      if (k >= K) break;
      const float w = a.w[(long long)k * a.C_out_pad + co];       // packed (cin_pad, K, C_out_pad), channel 0
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaf(w, win[k + j], v[j]);
    }
    const float bs = a.bias ? a.bias[co] : 0.f;
    const float al = a.alpha_out ? a.alpha_out[co] : 0.f;
    const float inv = a.alpha_out ? snake_inv(al) : 0.f;
    float w2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j] + bs;
      if (a.alpha_out) x = snake_apply(x, al, inv);
      if (a.act != FAC_ACT_NONE) x = apply_act_slow(x, a.act);
      v[j] = x;
    }
    if (y2g) {
      const float a2 = a.alpha2[co], i2 = snake_inv(a2);
#pragma unroll
      for (int j = 0; j < 4; ++j) w2[j] = snake_apply(v[j], a2, i2);
    }
    const long long o = (long long)co * a.y_cs + t;
    if (full) {
      if (yg) *reinterpret_cast<float4*>(yg + o) = make_float4(v[0], v[1], v[2], v[3]);
      if (y2g) *reinterpret_cast<float4*>(y2g + o) = make_float4(w2[0], w2[1], w2[2], w2[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (t + j >= a.T_out) continue;
        if (yg) yg[o + j] = v[j];
        if (y2g) y2g[o + j] = w2[j];
      }
    }
  }
}

bool conv_cin1_ok(const ConvArgs& a) {
  return a.C_in == 1 && a.C_out > 2 && a.K <= CIN1_KMAX && a.stride == 1 && a.dil == 1 && a.n_phase == 1 && a.phase_shift == 0 &&
         a.y_tstride == 1 && !a.alpha_in && !a.res && !a.w1 && !a.w_batched && a.B <= 65535 && !conv_two_level(a) &&
         // one workgroup per (batch, 1024-step tile) walks every output channel: needs a chip's worth of tiles (the period
         // discriminators' 1 -> 1024 data gradient over one row-concatenated signal has ~20 and belongs on the MFMA tile)
         (long long)a.B * ((a.T_out + NARROW_TT - 1) / NARROW_TT) >= 256 && a.C_out <= 256;
}

int conv_dispatch_cin1(ConvArgs& a, hipStream_t s) {
  dim3 grid((a.T_out + NARROW_TT - 1) / NARROW_TT, a.B);
  hipLaunchKernelGGL(conv1d_cin1_kernel, grid, dim3(256), 0, s, a);
  return check_launch("conv1d_cin1");
}

// One or two output channels over FEW (batch, 1024-step) tiles but many input channels (the period discriminators' 1024 -> 1
// output conv over one row-concatenated signal, dac/model/discriminator.py:35): 40 MB of input for 60 MFLOP, and only ~10 tiles
// to spread over 256 CUs -- so the reduction over input channels is split as well: workgroup (t tile, channel chunk, batch)
// writes the partial sums of its 64 channels, a second pass adds the chunks in fixed order (deterministic) and applies bias /
// Snake / activation.  (The 32 x 256 MFMA tile did this layer in 250 us -- 39 workgroups each walking all 1024 channels.)
constexpr int THIN_TT = 256;
constexpr int THIN_CC = 64;
constexpr int THIN_KMAX = 9;

template <int CO>
__global__ __launch_bounds__(THIN_TT) void conv1d_thin_part_kernel(ConvArgs a, float* __restrict__ part) {
  const int t = blockIdx.x * THIN_TT + threadIdx.x;
  const int z = blockIdx.y, b = blockIdx.z;
  if (t >= a.T_out) return;
  const float* xg = a.x + (long long)b * a.x_bs;
  int idx[THIN_KMAX];
#pragma unroll
  for (int k = 0; k < THIN_KMAX; ++k) {
    const int tin = t - a.pad_left + k * a.dil;
    int i = -1;
    if (k < a.K) {
      if (a.pad_mode == FAC_PAD_REFLECT) i = reflect_index(tin, a.T_in, a.T_ext);
      else i = (tin >= 0 && tin < a.T_in) ? tin : -1;
    }
    idx[k] = i;
  }
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  const int c_end = min(a.C_in, (z + 1) * THIN_CC);
  for (int ci = z * THIN_CC; ci < c_end; ++ci) {
    const float* xrow = xg + (long long)ci * a.x_cs;
    const float al = a.alpha_in ? a.alpha_in[ci] : 0.f;
    const float inv = a.alpha_in ? snake_inv(al) : 0.f;
#pragma unroll
    for (int k = 0; k < THIN_KMAX; ++k) {
      if (k >= a.K) break;
      float x = idx[k] >= 0 ? xrow[idx[k]] : 0.f;
      if (a.alpha_in) x = snake_apply(x, al, inv);
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = fmaf(a.w[((long long)ci * a.K + k) * a.C_out_pad + c], x, acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < CO; ++c)
    if (c < a.C_out) part[(((long long)z * a.B + b) * CO + c) * a.T_out + t] = acc[c];
}

template <int CO>
__global__ __launch_bounds__(256) void conv1d_thin_sum_kernel(ConvArgs a, const float* __restrict__ part, int n_chunks) {
  const long long n = (long long)a.B * a.C_out * a.T_out;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int t = (int)(i % a.T_out);
    const int c = (int)((i / a.T_out) % a.C_out);
    const int b = (int)(i / ((long long)a.T_out * a.C_out));
    float v = 0.f;
    for (int z = 0; z < n_chunks; ++z) v += part[(((long long)z * a.B + b) * CO + c) * a.T_out + t];
    v += a.bias ? a.bias[c] : 0.f;
    if (a.alpha_out) v = snake_apply(v, a.alpha_out[c], snake_inv(a.alpha_out[c]));
    if (a.act != FAC_ACT_NONE) v = apply_act_slow(v, a.act);
    a.y[(long long)b * a.y_bs + (long long)c * a.y_cs + t] = v;
  }
}

// Round 6: the same kernel pair for up to EIGHT output channels (1 <= K <= 2 taps; C_out 3 .. 8): the data gradient of the quantizers'
// out-projections (1024 -> 8 over 16 x 160 frames, dac/nn/quantize.py:46-53 backward) ran on the 32 x 256 MFMA tile -- 16 workgroups
// each walking all 1 024 channels, 225 us for 42 MFLOP and 10.5 MB (0.2 TFLOP/s; six such launches per train step).
static int thin_co(int C_out) { return C_out <= 1 ? 1 : C_out <= 2 ? 2 : C_out <= 4 ? 4 : 8; }

bool conv_thin_ok(const ConvArgs& a, const void* ws, long long ws_bytes) {
  if (!((a.C_out <= 2 || (a.C_out <= 8 && a.K <= 2 && !a.alpha_in)) && a.C_in >= 2 * THIN_CC && a.K <= THIN_KMAX && a.stride == 1 && a.n_phase == 1 && a.phase_shift == 0 &&
        a.y_tstride == 1 && !a.res && !a.y2 && !a.w1 && !a.w_batched && a.y && ws && a.B <= 65535 && !conv_two_level(a)))
    return false;
  const long long chunks = (a.C_in + THIN_CC - 1) / THIN_CC;
  return chunks <= 65535 && ws_bytes >= chunks * a.B * thin_co(a.C_out) * a.T_out * (long long)sizeof(float);
}

int conv_dispatch_thin(ConvArgs& a, void* ws, hipStream_t s) {
  const int chunks = (a.C_in + THIN_CC - 1) / THIN_CC;
  float* part = reinterpret_cast<float*>(ws);
  dim3 grid((a.T_out + THIN_TT - 1) / THIN_TT, chunks, a.B);
  const long long n = (long long)a.B * a.C_out * a.T_out;
  const int rb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  switch (thin_co(a.C_out)) {
    case 1:
      hipLaunchKernelGGL(conv1d_thin_part_kernel<1>, grid, dim3(THIN_TT), 0, s, a, part);
      hipLaunchKernelGGL(conv1d_thin_sum_kernel<1>, dim3(rb), dim3(256), 0, s, a, part, chunks);
      break;
    case 2:
      hipLaunchKernelGGL(conv1d_thin_part_kernel<2>, grid, dim3(THIN_TT), 0, s, a, part);
      hipLaunchKernelGGL(conv1d_thin_sum_kernel<2>, dim3(rb), dim3(256), 0, s, a, part, chunks);
      break;
    case 4:
      hipLaunchKernelGGL(conv1d_thin_part_kernel<4>, grid, dim3(THIN_TT), 0, s, a, part);
      hipLaunchKernelGGL(conv1d_thin_sum_kernel<4>, dim3(rb), dim3(256), 0, s, a, part, chunks);
      break;
    default:
      hipLaunchKernelGGL(conv1d_thin_part_kernel<8>, grid, dim3(THIN_TT), 0, s, a, part);
      hipLaunchKernelGGL(conv1d_thin_sum_kernel<8>, dim3(rb), dim3(256), 0, s, a, part, chunks);
  }
  return check_launch("conv1d_thin");
}

int conv_dispatch_narrow(ConvArgs& a, hipStream_t s) {
  const int XW = (NARROW_TT + (a.KV - 1) * a.dil + 3) & ~3;
  const size_t lds = ((size_t)NARROW_CIC * XW + (size_t)NARROW_CIC * a.KV * 2) * sizeof(float);
  if (lds > 64 * 1024 || (a.KV - 1) * a.dil > 64) {
    set_error("conv1d(narrow): receptive field too wide (K=%d dil=%d)", a.KV, a.dil);
    return FAC_ERR_ARG;
  }
  dim3 grid((a.T_out + NARROW_TT - 1) / NARROW_TT, a.B);
  if (a.C_out == 1) hipLaunchKernelGGL(conv1d_narrow_kernel<1>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(conv1d_narrow_kernel<2>, grid, dim3(256), lds, s, a);
  return check_launch("conv1d_narrow");
}

}  // namespace fac
