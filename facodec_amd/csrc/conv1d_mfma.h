// K1-K4: Conv1d / polyphase ConvTranspose1d as an implicit GEMM on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain), with the Snake prologue,
// bias, Snake / tanh epilogue and residual add fused in.
//
// Replaces (reference, /root/reference): SConv1d.forward dac/model/encodec.py:212-228,
// SConvTranspose1d.forward :248-270, snake dac/nn/layers.py:18-24, ResidualUnit dac/model/dac.py:25-42.
//
// GEMM view per batch element:  Y[co, t] = sum_{ci,k} Wp[ci][k][co] * Xp[ci, t*stride + k*dil - pad]
//   M = C_out (A operand: packed weights, co fastest -> conflict-free ds_read_b32)
//   N = T_out (B operand: the LDS-staged receptive-field tile, time fastest)
//   K = C_in*K taps, consumed two input channels at a time (the 32x32x2 k-pair = lanes 0-31 / 32-63)
//
// One workgroup (4 waves) owns a CO_TILE x T_TILE output tile of one (batch, phase) and walks C_in
// in chunks of `cic` channels.  Per chunk:
//   * the weight slab [cic][K][CO_TILE] goes HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B/lane,
//     no VGPR round trip), issued BEFORE the MFMA block of the previous chunk;
//   * the input slab [cic][XW] (receptive field incl. halo, reflect/zero padded) is loaded into
//     registers before that MFMA block and written to LDS after it, with Snake applied on the
//     way (once per staged element) -- so HBM/L2 latency hides under the matrix work;
//   * LDS is double-buffered: one barrier per chunk.
// The tap loop is a compile-time unroll (template KT) so every LDS read has an immediate offset
// and the compiler can run the ds_reads ahead of the MFMAs.
#pragma once
#include "common.h"

namespace fac {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int CONV_XMAX = 14;  // staged input dwords held in registers per thread

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* alpha_in;
  const float* alpha_out;
  const float* res;
  float* y;
  long long x_bs, x_cs, y_bs, y_cs, w_bs;
  int B, C_in, T_in, T_ext, C_out, C_out_pad, T_out;
  int K, stride, dil, pad_left, pad_mode;
  int n_phase, y_tstride, act, w_batched;
  int cic;  // input channels per LDS stage (multiple of 2*UC)
  int XW;   // staged input width = (T_TILE-1)*stride + (K-1)*dil + 1
  int XB;   // 64-wide column blocks per staged row = ceil(XW/64)
  int XQ, XR;  // 4 / XB, 4 % XB: (row, block) advance of one wave per staging iteration
};

template <int KT>
struct ConvUnroll {
  static constexpr int UC = (KT == 1) ? 4 : ((KT == 2 || KT == 3) ? 2 : 1);  // channel pairs per unrolled body
};

template <int MB, int NB, int WM, int WN, int KT>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvArgs a) {
  constexpr int CO_TILE = 32 * MB * WM;
  constexpr int T_TILE = 32 * NB * WN;
  constexpr int CO4 = CO_TILE / 4;
  constexpr int UC = ConvUnroll<KT>::UC;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int kq = lane >> 5;
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int t0 = blockIdx.x * T_TILE;
  const int co0 = blockIdx.y * CO_TILE;
  const int b = blockIdx.z / a.n_phase;
  const int phase = blockIdx.z - b * a.n_phase;

  const int K = KT > 0 ? KT : a.K;
  const int cic = a.cic, XW = a.XW, XB = a.XB;
  const int w_stage = cic * K * CO_TILE;  // floats
  const int x_stage = cic * XW;
  float* Wbuf = smem;                 // [2][cic][K][CO_TILE]
  float* Xbuf = smem + 2 * w_stage;   // [2][cic][XW]

  const float* xg = a.x + (long long)b * a.x_bs;
  const float* wg = a.w + (long long)phase * a.C_in * K * a.C_out_pad +
                    (a.w_batched ? (long long)b * a.w_bs : 0ll);
  const int tin0 = t0 * a.stride - a.pad_left;
  const int n_chunks = (a.C_in + cic - 1) / cic;
  const int w_rows_total = a.C_in * K;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // ---- weight slab by LDS-DMA: flat float4 index q -> (row, col4); 64 lanes = 1 KiB contiguous in LDS.
  // Rows past C_in*K and columns past C_out_pad are CLAMPED to valid weights (finite values): the
  // former meet zero-filled input rows, the latter only feed output rows that are never stored.
  auto issue_w = [&](int chunk, int buf) {
    const int n4 = cic * K * CO4;
    const int row_base = chunk * cic * K;
    float* dst0 = Wbuf + buf * w_stage;
    for (int i = wave; i * 64 < n4; i += 4) {
      const int q = i * 64 + lane;
      if (q < n4) {
        const int row = q / CO4;
        const int c4 = q - row * CO4;
        int grow = row_base + row;
        grow = grow < w_rows_total ? grow : w_rows_total - 1;
        int co = co0 + 4 * c4;
        co = co < a.C_out_pad ? co : a.C_out_pad - 4;
        const float* src = wg + (long long)grow * a.C_out_pad + co;
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst0 + i * 256), 16, 0, 0);
      }
    }
  };

  // ---- input slab: iteration `it` (wave-uniform) covers row it/XB, columns (it%XB)*64 + lane
  float xr[CONV_XMAX];
  const int it_r0 = wave / XB, it_c0 = wave - it_r0 * XB;   // this wave's first (row, column block)
  auto load_x = [&](int chunk) {
    const int ci0 = chunk * cic;
    int r = it_r0, cb = it_c0;
#pragma unroll
    for (int j = 0; j < CONV_XMAX; ++j) {
      float v = 0.f;
      const int c = cb * 64 + lane;
      const int ci = ci0 + r;
      if (r < cic && c < XW && ci < a.C_in) {
        const int tin = tin0 + c;
        int idx;
        if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
        else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
        if (idx >= 0) v = xg[(long long)ci * a.x_cs + idx];
      }
      xr[j] = v;
      r += a.XQ;
      cb += a.XR;
      if (cb >= XB) { cb -= XB; ++r; }
    }
  };
  auto store_x = [&](int chunk, int buf) {
    const int ci0 = chunk * cic;
    float* dst = Xbuf + buf * x_stage;
    int r = it_r0, cb = it_c0;
#pragma unroll
    for (int j = 0; j < CONV_XMAX; ++j) {
      const int c = cb * 64 + lane;
      if (r < cic && c < XW) {
        float v = xr[j];
        if (a.alpha_in != nullptr) {
          const int ci = ci0 + r;
          const float al = a.alpha_in[ci < a.C_in ? ci : a.C_in - 1];
          v = snake_apply(v, al, snake_inv(al));   // snake(0) == 0 keeps the zero fill
        }
        dst[r * XW + c] = v;
      }
      r += a.XQ;
      cb += a.XR;
      if (cb >= XB) { cb -= XB; ++r; }
    }
  };

  issue_w(0, 0);
  load_x(0);
  store_x(0, 0);
  __syncthreads();

  const int a_off = wm * (MB * 32) + l31;                 // column inside the weight row
  const int b_off = (wn * (NB * 32) + l31) * a.stride;    // time offset inside the input row
  const int wrow_stride = K * CO_TILE;                    // floats per input channel in Wbuf
  const int dil = a.dil;
  const int nstride = 32 * a.stride;

  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const int buf = chunk & 1;
    const bool has_next = chunk + 1 < n_chunks;
    if (has_next) {
      issue_w(chunk + 1, buf ^ 1);
      load_x(chunk + 1);
    }
    const float* Wb = Wbuf + buf * w_stage + a_off + kq * wrow_stride;
    const float* Xb = Xbuf + buf * x_stage + b_off + kq * XW;
    if constexpr (KT > 0) {
      for (int c2 = 0; c2 < cic; c2 += 2 * UC) {
#pragma unroll
        for (int u = 0; u < UC; ++u) {
          const float* wp = Wb + (c2 + 2 * u) * wrow_stride;
          const float* xp = Xb + (c2 + 2 * u) * XW;
#pragma unroll
          for (int kk = 0; kk < KT; ++kk) {
            float av[MB], bv[NB];
#pragma unroll
            for (int m = 0; m < MB; ++m) av[m] = wp[kk * CO_TILE + m * 32];
#pragma unroll
            for (int n = 0; n < NB; ++n) bv[n] = xp[kk * dil + n * nstride];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
              for (int n = 0; n < NB; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
          }
        }
      }
    } else {
      for (int c2 = 0; c2 < cic; c2 += 2) {
        const float* wp = Wb + c2 * wrow_stride;
        const float* xp = Xb + c2 * XW;
#pragma unroll 2
        for (int kk = 0; kk < K; ++kk) {
          float av[MB], bv[NB];
#pragma unroll
          for (int m = 0; m < MB; ++m) av[m] = wp[kk * CO_TILE + m * 32];
#pragma unroll
          for (int n = 0; n < NB; ++n) bv[n] = xp[kk * dil + n * nstride];
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
    if (has_next) store_x(chunk + 1, buf ^ 1);
    __syncthreads();   // also drains the LDS-DMA of the next weight slab (vmcnt(0))
  }

  // ---- epilogue: C/D layout col = lane&31 (time), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (co)
  float* yg = a.y + (long long)b * a.y_bs;
  const float* rg = a.res ? a.res + (long long)b * a.y_bs : nullptr;
#pragma unroll
  for (int m = 0; m < MB; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * (MB * 32) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
      if (co >= a.C_out) continue;
      const float bsv = a.bias ? a.bias[co] : 0.f;
      float al = 0.f, inv = 0.f;
      if (a.alpha_out) {
        al = a.alpha_out[co];
        inv = snake_inv(al);
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int t = t0 + wn * (NB * 32) + n * 32 + l31;
        if (t >= a.T_out) continue;
        float v = acc[m][n][r] + bsv;
        if (a.alpha_out) v = snake_apply(v, al, inv);
        if (a.act == FAC_ACT_TANH) v = tanhf(v);
        else if (a.act == FAC_ACT_MISH) {
          // x * tanh(softplus(x)); softplus with torch's threshold 20
          float sp = v > 20.f ? v : log1pf(expf(v));
          v = v * tanhf(sp);
        } else if (a.act == FAC_ACT_LOG_MEL) {
          v = (logf(1e-5f + v) + 4.0f) / 4.0f;
        }
        const long long o = (long long)co * a.y_cs + (long long)t * a.y_tstride + phase;
        if (rg) v += rg[o];
        yg[o] = v;
      }
    }
  }
}

// Picks channels-per-stage and launches one instantiation.
template <int MB, int NB, int WM, int WN, int KT>
int launch_cfg(ConvArgs& a, hipStream_t s) {
  constexpr int CO_TILE = 32 * MB * WM;
  constexpr int T_TILE = 32 * NB * WN;
  constexpr int UC = ConvUnroll<KT>::UC;
  constexpr int STEP = 2 * UC;
  a.XW = (T_TILE - 1) * a.stride + (a.K - 1) * a.dil + 1;
  a.XB = (a.XW + 63) / 64;
  a.XQ = 4 / a.XB;
  a.XR = 4 % a.XB;
  const int per_ci = a.K * CO_TILE + a.XW;       // floats per staged input channel
  int lim = 9216 / per_ci;                       // ~36 KB per stage -> 2 stages x 2 workgroups per CU
  const int lim_regs = (4 * CONV_XMAX) / a.XB;   // staged inputs must fit CONV_XMAX registers/thread
  if (lim > lim_regs) lim = lim_regs;
  if (lim > 32) lim = 32;
  lim = (lim / STEP) * STEP;
  if (lim < STEP) lim = STEP;
  const int cin_r = ((a.C_in + STEP - 1) / STEP) * STEP;
  int cic = lim < cin_r ? lim : cin_r;
  // prefer a chunk size that divides the (rounded) channel count: no half-empty last chunk
  for (int c = cic; c >= STEP && c * 2 > cic; c -= STEP)
    if (cin_r % c == 0) { cic = c; break; }
  a.cic = cic;
  if (cic * a.XB > 4 * CONV_XMAX) {
    set_error("conv1d: receptive field too wide to stage (K=%d stride=%d dil=%d)", a.K, a.stride, a.dil);
    return FAC_ERR_ARG;
  }
  const size_t lds = (size_t)2 * cic * per_ci * sizeof(float);
  if (lds > 160 * 1024) {
    set_error("conv1d: tile needs %zu B of LDS (K=%d stride=%d dil=%d)", lds, a.K, a.stride, a.dil);
    return FAC_ERR_ARG;
  }
  auto kern = conv1d_mfma_kernel<MB, NB, WM, WN, KT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid((a.T_out + T_TILE - 1) / T_TILE, (a.C_out + CO_TILE - 1) / CO_TILE, a.B * a.n_phase);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  return check_launch("conv1d_mfma");
}

// One dispatcher per tile shape (each in its own translation unit so they build in parallel).
int conv_dispatch_128x128(ConvArgs& a, hipStream_t s);
int conv_dispatch_96x128(ConvArgs& a, hipStream_t s);
int conv_dispatch_64x128(ConvArgs& a, hipStream_t s);
int conv_dispatch_32x256(ConvArgs& a, hipStream_t s);
int conv_dispatch_128x32(ConvArgs& a, hipStream_t s);

}  // namespace fac
