// Backward of the anti-aliased SnakeBeta activation of the predictor heads (alias_free_torch/act.py:24-29 around
// modules/quantize.py:77-90; forward: fac_aa_snakebeta_fwd in misc.hip):
//   xp = replicate-pad(x, 5, 5);  u[m] = 2 sum_ip xp[ip] f[m + 15 - 2 ip]          (m in [0, 2T))
//   a[m] = u + sin^2(u e^alpha) / (e^beta + 1e-9)
//   ap = replicate-pad(a, 5, 6);  y[t] = sum_j ap[2t + j] f[j]
// One wave per (b, c) row, 128-step chunks: recomputes u for the chunk (+ halo), pulls dy back through the decimating
// filter and the padding, through the activation (du, and this tile's share of d alpha / d beta), then through the
// interpolating filter and the input padding.  d alpha / d beta partials per (b, c) row are summed per channel by a
// second kernel in a fixed order.
#include "common.h"
#include "../../include/facodec_hip.h"

namespace fac {

constexpr int AB_TT = 128;          // output steps per chunk of one wave
constexpr int AB_ROWS = 4;          // (b, c) rows per workgroup: one per wave

// Round 6 rewrite.  The predictor heads run this at T = 160 frames on (16, 1024, 160) tensors: the round-3 kernel gave every
// (b, c) row a 256-thread workgroup (16 384 workgroups with 37 % idle lanes, a 12-way branchy tap loop per position, an
// 8-barrier tree reduction) and ran at 110 us = 0.29 TB/s for 31 MB (profiles/r06_pmc_train_before.json).  Now ONE WAVE owns a
// row and walks it in chunks of 128 steps: the taps that exist are enumerated directly (6 per position, parity-selected; the
// replicate-padded edges add their extra taps only at m = 0 / 2T - 1 and t = 0 / T - 1), the parameter partials are reduced
// with wave shuffles in a fixed order, and four rows share a workgroup (uniform trip counts, so the phase barriers are cheap).
__global__ __launch_bounds__(256) void aa_snakebeta_bwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha_log,
                                                               const float* __restrict__ beta_log, const float* __restrict__ filt,
                                                               const float* __restrict__ dy, float* __restrict__ dx,
                                                               float* __restrict__ part, int C, int T, int n_rows) {
  constexpr int XW = AB_TT + 16;            // x / dy window: [t0 - 8, t0 + TT + 8)
  constexpr int UW = 2 * AB_TT + 16;        // du window: [2 t0 - 8, 2 t0 + 2 TT + 8)
  __shared__ float xs_[AB_ROWS][XW];
  __shared__ float dys_[AB_ROWS][XW];
  __shared__ float dus_[AB_ROWS][UW];
  __shared__ float f[12];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x * AB_ROWS + wave;
  const bool live = row < n_rows;
  const int bc = live ? row : n_rows - 1;
  const int c = bc % C;
  float* xs = xs_[wave];
  float* dys = dys_[wave];
  float* dus = dus_[wave];
  const float* xr = x + (long long)bc * T;
  const float* dr = dy + (long long)bc * T;
  if (tid < 12) f[tid] = filt[tid];
  const float ea = expf(alpha_log[c]);
  const float eb = expf(beta_log[c]);
  const float inv = 1.0f / (eb + 1e-9f);
  float s_a = 0.f, s_b = 0.f;
  for (int t0 = 0; t0 < T; t0 += AB_TT) {
    __syncthreads();                                   // the previous chunk's readers are done (and f[] is there)
    for (int i = lane; i < XW; i += 64) {
      const int t = t0 - 8 + i;
      xs[i] = xr[t < 0 ? 0 : (t > T - 1 ? T - 1 : t)];
      dys[i] = (t >= 0 && t < T) ? dr[t] : 0.f;
    }
    __syncthreads();
    for (int i = lane; i < UW; i += 64) {
      const int m = 2 * t0 - 8 + i;
      float du = 0.f;
      if (m >= 0 && m < 2 * T && i >= 2 && i < UW - 2) {      // (the outer two positions of the window are never read)
        // u[m] = 2 sum_q xp[ip_lo + q] f[k0 - 2 q]: ip_lo = (m + 5) >> 1, k0 = 11 (m even) / 10 (m odd), xp[ip] = x[clamp(ip - 5)]
        const int ip_lo = (m + 5) >> 1;
        const int k0 = 11 - (m & 1);
        float u = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          int t = ip_lo + q - 5;
          t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
          u = fmaf(xs[t - (t0 - 8)], f[k0 - 2 * q], u);
        }
        u *= 2.0f;
        // da[m] = sum over the padded positions n that read a[m] of dap[n];  dap[n] = sum_{j == n mod 2} dy[(n - j) / 2] f[j]
        auto dap = [&](int n) {
          const int j0 = n & 1;
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            const int tp = (n - j0) / 2 - q;             // j = j0 + 2 q
            const int li = tp - (t0 - 8);
            if (li >= 0 && li < XW) v = fmaf(dys[li], f[j0 + 2 * q], v);      // outside the window: dy is zero or not needed here
          }
          return v;
        };
        float da = dap(m + 5);
        if (m == 0)
          for (int n = 0; n < 5; ++n) da += dap(n);
        if (m == 2 * T - 1)
          for (int n = 2 * T + 5; n <= 2 * T + 10; ++n) da += dap(n);
        const float ue = u * ea;
        float sn, cs;
        sincosf(ue, &sn, &cs);
        du = da * (1.0f + 2.0f * sn * cs * ea * inv);
        if (m >= 2 * t0 && m < 2 * t0 + 2 * AB_TT) {      // this chunk's own positions: counted once
          s_a += da * 2.0f * sn * cs * ue * inv;
          s_b += da * sn * sn * (-inv * inv) * eb;
        }
      }
      dus[i] = du;
    }
    __syncthreads();
    for (int tl = lane; tl < AB_TT; tl += 64) {
      const int t = t0 + tl;
      if (t >= T) break;
      // dx[t] = 2 sum over the padded positions ip that read x[t] of sum_k du[2 ip - 15 + k] f[k]
      auto dxp = [&](int ip) {
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const int li = 2 * ip - 15 + k - (2 * t0 - 8);
          if (li >= 0 && li < UW) g = fmaf(dus[li], f[k], g);        // du outside [0, 2T) is stored as zero
        }
        return g;
      };
      float g = dxp(t + 5);
      if (t == 0)
        for (int ip = 0; ip < 5; ++ip) g += dxp(ip);
      if (t == T - 1)
        for (int ip = T + 5; ip <= T + 9; ++ip) g += dxp(ip);
      if (live) dx[(long long)bc * T + t] = 2.0f * g;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s_a += __shfl_xor(s_a, o);
    s_b += __shfl_xor(s_b, o);
  }
  if (lane == 0 && live) {
    part[(long long)bc * 2] = s_a;
    part[(long long)bc * 2 + 1] = s_b;
  }
}

// dalpha[c] = sum over (b, tiles) of the partials, fixed order
__global__ void aa_param_reduce_kernel(const float* __restrict__ part, float* __restrict__ dalpha, float* __restrict__ dbeta,
                                       int B, int C, int n_tiles) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sa = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < n_tiles; ++i) {
      sa += part[(((long long)b * C + c) * n_tiles + i) * 2];
      sb += part[(((long long)b * C + c) * n_tiles + i) * 2 + 1];
    }
  dalpha[c] = sa;
  dbeta[c] = sb;
}

}  // namespace fac

extern "C" int fac_aa_snakebeta_bwd(const float* x, const float* alpha_log, const float* beta_log, const float* filter12,
                                    const float* dy, float* dx, float* dalpha, float* dbeta, float* scratch, int B, int C,
                                    int T, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(x && alpha_log && beta_log && filter12 && dy && dx && dalpha && dbeta && scratch && B > 0 && C > 0 && T > 1,
              "aa_snakebeta_bwd: bad arguments");
  FAC_REQUIRE((long long)B * C <= 65535, "aa_snakebeta_bwd: B*C too large");
  const int n_tiles = 1;                    // one partial pair per (b, c) row
  hipLaunchKernelGGL(aa_snakebeta_bwd_kernel, dim3((B * C + AB_ROWS - 1) / AB_ROWS), dim3(256), 0, (hipStream_t)stream, x, alpha_log,
                     beta_log, filter12, dy, dx, scratch, C, T, B * C);
  hipLaunchKernelGGL(aa_param_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, dalpha, dbeta, B, C,
                     n_tiles);
  return check_launch("aa_snakebeta_bwd");
}
