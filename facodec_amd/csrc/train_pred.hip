// Backward of the anti-aliased SnakeBeta activation of the predictor heads (alias_free_torch/act.py:24-29 around
// modules/quantize.py:77-90; forward: fac_aa_snakebeta_fwd in misc.hip):
//   xp = replicate-pad(x, 5, 5);  u[m] = 2 sum_ip xp[ip] f[m + 15 - 2 ip]          (m in [0, 2T))
//   a[m] = u + sin^2(u e^alpha) / (e^beta + 1e-9)
//   ap = replicate-pad(a, 5, 6);  y[t] = sum_j ap[2t + j] f[j]
// One workgroup per (b, c, 256-step tile): recomputes u for the tile (+ halo), pulls dy back through the decimating
// filter and the padding, through the activation (du, and this tile's share of d alpha / d beta), then through the
// interpolating filter and the input padding.  d alpha / d beta partials per (b*c, tile) are summed per channel by a
// second kernel in a fixed order.
#include "common.h"
#include "../../include/facodec_hip.h"

namespace fac {

constexpr int AB_TT = 256;

__global__ __launch_bounds__(256) void aa_snakebeta_bwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha_log,
                                                               const float* __restrict__ beta_log, const float* __restrict__ filt,
                                                               const float* __restrict__ dy, float* __restrict__ dx,
                                                               float* __restrict__ part, int C, int T, int n_tiles) {
  __shared__ float xs[AB_TT + 16];          // x[t0-8 .. t0+TT+7] (clamped)
  __shared__ float dys[AB_TT + 16];         // dy[t0-8 .. t0+TT+7] (zero outside)
  __shared__ float dus[2 * AB_TT + 24];     // du[2 t0 - 8 .. 2 t0 + 2 TT + 15] (zero outside [0, 2T))
  __shared__ float f[12];
  __shared__ float red[2][256];
  const int bc = blockIdx.y, c = bc % C, t0 = blockIdx.x * AB_TT, tid = threadIdx.x;
  const float* xr = x + (long long)bc * T;
  const float* dr = dy + (long long)bc * T;
  if (tid < 12) f[tid] = filt[tid];
  for (int i = tid; i < AB_TT + 16; i += 256) {
    const int t = t0 - 8 + i;
    xs[i] = xr[t < 0 ? 0 : (t > T - 1 ? T - 1 : t)];
    dys[i] = (t >= 0 && t < T) ? dr[t] : 0.f;
  }
  __syncthreads();
  const float ea = expf(alpha_log[c]);
  const float eb = expf(beta_log[c]);
  const float inv = 1.0f / (eb + 1e-9f);
  float s_a = 0.f, s_b = 0.f;
  for (int i = tid; i < 2 * AB_TT + 24; i += 256) {
    const int m = 2 * t0 - 8 + i;
    float du = 0.f;
    if (m >= 0 && m < 2 * T) {
      // u[m] as in the forward
      float u = 0.f;
      const int ip_lo = (m + 4 + 1) >> 1;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int ip = ip_lo + q, k = m + 15 - 2 * ip;
        if (k >= 0 && k <= 11) {
          int t = ip - 5;
          t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
          u = fmaf(xs[t - (t0 - 8)], f[k], u);
        }
      }
      u *= 2.0f;
      // da[m] = sum over padded positions n that read a[m]:  n = m + 5, plus the replicated edges
      int n_lo = m + 5, n_hi = m + 5;
      if (m == 0) n_lo = 0;
      if (m == 2 * T - 1) n_hi = 2 * T + 10;
      float da = 0.f;
      for (int n = n_lo; n <= n_hi; ++n) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const int tw = n - j;                  // = 2 t'
          if (tw >= 0 && !(tw & 1)) {
            const int tp = tw >> 1;
            if (tp < T) {
              const int li = tp - (t0 - 8);
              da = fmaf((li >= 0 && li < AB_TT + 16) ? dys[li] : dr[tp], f[j], da);
            }
          }
        }
      }
      const float ue = u * ea;
      const float sn = sinf(ue), cs = cosf(ue);
      du = da * (1.0f + 2.0f * sn * cs * ea * inv);
      if (m >= 2 * t0 && m < 2 * t0 + 2 * AB_TT) {      // this tile's own positions: counted once
        s_a += da * 2.0f * sn * cs * ue * inv;
        s_b += da * sn * sn * (-inv * inv) * eb;
      }
    }
    dus[i] = du;
  }
  red[0][tid] = s_a;
  red[1][tid] = s_b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    part[((long long)bc * n_tiles + blockIdx.x) * 2] = red[0][0];
    part[((long long)bc * n_tiles + blockIdx.x) * 2 + 1] = red[1][0];
  }
  const int t = t0 + tid;
  if (t < T) {
    int ip_lo = t + 5, ip_hi = t + 5;
    if (t == 0) ip_lo = 0;
    if (t == T - 1) ip_hi = T + 9;
    float g = 0.f;
    for (int ip = ip_lo; ip <= ip_hi; ++ip) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int m = 2 * ip - 15 + k;
        if (m >= 0 && m < 2 * T) {
          const int li = m - (2 * t0 - 8);
          if (li >= 0 && li < 2 * AB_TT + 24) g = fmaf(dus[li], f[k], g);
        }
      }
    }
    dx[(long long)bc * T + t] = 2.0f * g;
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
  const int n_tiles = (T + AB_TT - 1) / AB_TT;
  hipLaunchKernelGGL(aa_snakebeta_bwd_kernel, dim3(n_tiles, B * C), dim3(256), 0, (hipStream_t)stream, x, alpha_log, beta_log,
                     filter12, dy, dx, scratch, C, T, n_tiles);
  hipLaunchKernelGGL(aa_param_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, dalpha, dbeta, B, C,
                     n_tiles);
  return check_launch("aa_snakebeta_bwd");
}
