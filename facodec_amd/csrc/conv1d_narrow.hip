// Convs with one or two output channels (the decoder's final 96 -> 1, k = 7 + tanh: dac/model/dac.py:158-159).
// An MFMA tile would waste 31/32 of its rows, and the layer is HBM-bound anyway (reads 590 MB at B = 32,
// writes 6 MB): plain VALU kernel, one workgroup per (batch, 1024-step tile), input rows staged through LDS
// eight channels at a time, four consecutive outputs per thread (a 10-value sliding window per channel).
#include "conv1d_mfma.h"

namespace fac {

constexpr int NARROW_TT = 1024;
constexpr int NARROW_CIC = 8;

template <int CO>
__global__ __launch_bounds__(256) void conv1d_narrow_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.K, dil = a.dil;
  const int halo = (K - 1) * dil;
  const int XW = NARROW_TT + halo;
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

  // The (row, column) -> input index map of this thread's staging slots is the same for every channel
  // chunk: resolve padding / reflection once, then each chunk is NSLOT independent loads issued together.
  constexpr int NSLOT = (NARROW_CIC * (NARROW_TT + 64) + 255) / 256;   // halo <= 64 columns
  int s_row[NSLOT], s_idx[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int i = tid + 256 * j;
    const int r = i / XW, c = i - r * XW;
    int idx = -1;
    if (i < NARROW_CIC * XW) {
      const int tin = t0 - a.pad_left + c;
      if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
      else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
    }
    s_row[j] = i < NARROW_CIC * XW ? r : -1;
    s_idx[j] = idx;
  }

  for (int ci0 = 0; ci0 < a.C_in; ci0 += NARROW_CIC) {
    __syncthreads();
    float v[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int ci = ci0 + s_row[j];
      v[j] = (s_row[j] >= 0 && s_idx[j] >= 0 && ci < a.C_in) ? xg[(long long)ci * a.x_cs + s_idx[j]] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      if (s_row[j] < 0) continue;
      float x = v[j];
      const int ci = ci0 + s_row[j];
      if (a.alpha_in && ci < a.C_in) x = snake_apply(x, a.alpha_in[ci], snake_inv(a.alpha_in[ci]));
      xs[tid + 256 * j] = x;
    }
    for (int i = tid; i < NARROW_CIC * K * CO; i += 256) {
      const int r = i / (K * CO), rem = i - r * (K * CO);
      const int k = rem / CO, c = rem - k * CO;
      const int ci = ci0 + r;
      ws[i] = ci < a.C_in ? a.w[((long long)ci * K + k) * a.C_out_pad + c] : 0.f;
    }
    __syncthreads();
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

int conv_dispatch_narrow(ConvArgs& a, hipStream_t s) {
  const int XW = NARROW_TT + (a.K - 1) * a.dil;
  const size_t lds = ((size_t)NARROW_CIC * XW + (size_t)NARROW_CIC * a.K * 2) * sizeof(float);
  if (lds > 64 * 1024 || (a.K - 1) * a.dil > 64) {
    set_error("conv1d(narrow): receptive field too wide (K=%d dil=%d)", a.K, a.dil);
    return FAC_ERR_ARG;
  }
  dim3 grid((a.T_out + NARROW_TT - 1) / NARROW_TT, a.B);
  if (a.C_out == 1) hipLaunchKernelGGL(conv1d_narrow_kernel<1>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(conv1d_narrow_kernel<2>, grid, dim3(256), lds, s, a);
  return check_launch("conv1d_narrow");
}

}  // namespace fac
