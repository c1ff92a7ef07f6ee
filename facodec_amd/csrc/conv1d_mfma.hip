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
// One workgroup (4 waves, 256 threads) owns a CO_TILE x T_TILE output tile of one (batch, phase)
// and walks C_in in chunks of `cic` channels; each chunk's weight slab [cic][K][CO_TILE] and
// input slab [cic][XW] are staged in LDS (double-buffered, one barrier per chunk).  Snake is
// applied once per staged input element, on its way into LDS.
#include "common.h"

namespace fac {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
  int cic;  // input channels per LDS stage (even)
  int XW;   // staged input width = (T_TILE-1)*stride + (K-1)*dil + 1
};

template <int MB, int NB, int WM, int WN>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvArgs a) {
  constexpr int CO_TILE = 32 * MB * WM;
  constexpr int T_TILE = 32 * NB * WN;
  constexpr int CO4 = CO_TILE / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int kq = lane >> 5;
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int t0 = blockIdx.x * T_TILE;
  const int co0 = blockIdx.y * CO_TILE;
  const int b = blockIdx.z / a.n_phase;
  const int phase = blockIdx.z - b * a.n_phase;

  const int K = a.K, cic = a.cic, XW = a.XW;
  const int w_stage = cic * K * CO_TILE;  // floats
  const int x_stage = cic * XW;
  float* Wbuf = smem;                 // [2][cic][K][CO_TILE]
  float* Xbuf = smem + 2 * w_stage;   // [2][cic][XW]

  const float* xg = a.x + (long long)b * a.x_bs;
  const float* wg = a.w + (long long)phase * a.C_in * K * a.C_out_pad +
                    (a.w_batched ? (long long)b * a.w_bs : 0ll);
  const int tin0 = t0 * a.stride - a.pad_left;
  const int n_chunks = (a.C_in + cic - 1) / cic;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  auto stage = [&](int chunk, int buf) {
    const int ci0 = chunk * cic;
    // ---- weights: rows (ci,k) of CO_TILE contiguous floats, float4 per thread
    {
      float4* dst = reinterpret_cast<float4*>(Wbuf + buf * w_stage);
      const int n4 = cic * K * CO4;
      const long long row_base = (long long)ci0 * K;
      const int rows_valid = (a.C_in - ci0) * K;  // rows beyond C_in are zero
      for (int i = tid; i < n4; i += 256) {
        const int row = i / CO4;
        const int q = i - row * CO4;
        const int co = co0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows_valid && co < a.C_out_pad)
          v = *reinterpret_cast<const float4*>(wg + (row_base + row) * a.C_out_pad + co);
        dst[i] = v;
      }
    }
    // ---- inputs: one row per wave at a time, lanes along time; snake on the way in
    {
      float* dst = Xbuf + buf * x_stage;
      for (int r = wave; r < cic; r += 4) {
        const int ci = ci0 + r;
        const bool cvalid = ci < a.C_in;
        const float* xrow = xg + (long long)ci * a.x_cs;
        float al = 0.f, inv = 0.f;
        if (a.alpha_in != nullptr && cvalid) {
          al = a.alpha_in[ci];
          inv = snake_inv(al);
        }
        for (int c = lane; c < XW; c += 64) {
          const int tin = tin0 + c;
          float v = 0.f;
          if (cvalid) {
            int idx;
            if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
            else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
            if (idx >= 0) {
              v = xrow[idx];
              if (a.alpha_in != nullptr) v = snake_apply(v, al, inv);
            }
          }
          dst[r * XW + c] = v;
        }
      }
    }
  };

  stage(0, 0);
  __syncthreads();

  const int a_off = wm * (MB * 32) + l31;                 // column inside the weight row
  const int b_off = (wn * (NB * 32) + l31) * a.stride;    // time offset inside the input row
  const int wrow_stride = K * CO_TILE;                    // floats per input channel in Wbuf

  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const int buf = chunk & 1;
    if (chunk + 1 < n_chunks) stage(chunk + 1, buf ^ 1);
    const float* Wb = Wbuf + buf * w_stage + a_off + kq * wrow_stride;
    const float* Xb = Xbuf + buf * x_stage + b_off + kq * XW;
    for (int kk = 0; kk < K; ++kk) {
      const float* wp = Wb + kk * CO_TILE;
      const float* xp = Xb + kk * a.dil;
#pragma unroll 4
      for (int c2 = 0; c2 < cic; c2 += 2) {
        float av[MB], bv[NB];
#pragma unroll
        for (int m = 0; m < MB; ++m) av[m] = wp[c2 * wrow_stride + m * 32];
#pragma unroll
        for (int n = 0; n < NB; ++n) bv[n] = xp[c2 * XW + n * 32 * a.stride];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
      }
    }
    __syncthreads();
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

template <int MB, int NB, int WM, int WN>
static int launch_cfg(ConvArgs& a, hipStream_t s) {
  constexpr int CO_TILE = 32 * MB * WM;
  constexpr int T_TILE = 32 * NB * WN;
  a.XW = (T_TILE - 1) * a.stride + (a.K - 1) * a.dil + 1;
  // channels per stage: keep one stage (weights + inputs) around 36 KB so two stages x two
  // workgroups fit the 160 KB LDS of a CU.
  const int per_ci = (a.K * CO_TILE + a.XW) * 4;
  int cic = (36 * 1024) / per_ci;
  cic &= ~1;
  if (cic < 2) cic = 2;
  if (cic > 32) cic = 32;
  int cin_even = (a.C_in + 1) & ~1;
  if (cic > cin_even) cic = cin_even;
  a.cic = cic;
  const size_t lds = (size_t)2 * cic * per_ci;
  if (lds > 160 * 1024) {
    set_error("conv1d: tile needs %zu B of LDS (K=%d stride=%d dil=%d)", lds, a.K, a.stride, a.dil);
    return FAC_ERR_ARG;
  }
  auto kern = conv1d_mfma_kernel<MB, NB, WM, WN>;
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

// Tile selection: M tile by output channels, narrow-N tile for short sequences (LSTM batches).
static int select_variant(const fac_conv_desc* d) {
  const int co = d->C_out;
  if (d->T_out <= 32) return 0;
  if (co <= 32) return 1;
  if (co <= 64) return 2;
  if (co % 128 != 0 && co % 96 == 0) return 3;
  return 4;
}

}  // namespace fac

extern "C" int fac_conv1d_fwd(const fac_conv_desc* d, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(d && d->x && d->w && d->y, "conv1d: null pointer");
  FAC_REQUIRE(d->B > 0 && d->C_in > 0 && d->C_out > 0 && d->T_in > 0 && d->T_out > 0,
              "conv1d: bad shape B=%d C_in=%d C_out=%d T_in=%d T_out=%d", d->B, d->C_in, d->C_out,
              d->T_in, d->T_out);
  FAC_REQUIRE(d->K >= 1 && d->stride >= 1 && d->dilation >= 1 && d->pad_left >= 0,
              "conv1d: bad K/stride/dilation/pad");
  FAC_REQUIRE(d->C_out_pad % 32 == 0 && d->C_out_pad >= d->C_out, "conv1d: C_out_pad must be a multiple of 32");
  FAC_REQUIRE(d->n_phase >= 1 && d->y_tstride >= 1, "conv1d: bad phase config");
  FAC_REQUIRE((long long)d->B * d->n_phase <= 65535, "conv1d: B*n_phase too large for grid.z");
  ConvArgs a;
  a.x = d->x; a.w = d->w; a.bias = d->bias; a.alpha_in = d->alpha_in; a.alpha_out = d->alpha_out;
  a.res = d->res; a.y = d->y;
  a.x_bs = d->x_bs; a.x_cs = d->x_cs; a.y_bs = d->y_bs; a.y_cs = d->y_cs; a.w_bs = d->w_bs;
  a.B = d->B; a.C_in = d->C_in; a.T_in = d->T_in; a.C_out = d->C_out; a.C_out_pad = d->C_out_pad;
  a.T_out = d->T_out; a.K = d->K; a.stride = d->stride; a.dil = d->dilation;
  a.pad_left = d->pad_left; a.pad_mode = d->pad_mode; a.n_phase = d->n_phase;
  a.y_tstride = d->y_tstride; a.act = d->act; a.w_batched = d->w_batched;
  // length of pad1d's temporary zero extension (only differs from T_in for inputs shorter than the pad)
  {
    long long last = (long long)(d->T_out - 1) * d->stride + (long long)(d->K - 1) * d->dilation - d->pad_left;
    int pad_right = last >= d->T_in ? (int)(last - d->T_in + 1) : 0;
    int max_pad = d->pad_left > pad_right ? d->pad_left : pad_right;
    a.T_ext = d->T_in > max_pad ? d->T_in : max_pad + 1;
  }
  hipStream_t s = (hipStream_t)stream;
  switch (select_variant(d)) {
    case 0: return launch_cfg<1, 1, 4, 1>(a, s);   // 128 x 32
    case 1: return launch_cfg<1, 2, 1, 4>(a, s);   // 32 x 256
    case 2: return launch_cfg<2, 1, 1, 4>(a, s);   // 64 x 128
    case 3: return launch_cfg<3, 1, 1, 4>(a, s);   // 96 x 128
    default: return launch_cfg<2, 2, 2, 2>(a, s);  // 128 x 128
  }
}

extern "C" int fac_conv1d_variant(const fac_conv_desc* d, char* name, int name_len) {
  using namespace fac;
  FAC_REQUIRE(d, "conv1d_variant: null descriptor");
  static const char* names[] = {"conv1d_mfma_kernel<1,1,4,1> 128x32", "conv1d_mfma_kernel<1,2,1,4> 32x256",
                                "conv1d_mfma_kernel<2,1,1,4> 64x128", "conv1d_mfma_kernel<3,1,1,4> 96x128",
                                "conv1d_mfma_kernel<2,2,2,2> 128x128"};
  const int v = select_variant(d);
  if (name && name_len > 0) snprintf(name, name_len, "%s", names[v]);
  return v;
}
