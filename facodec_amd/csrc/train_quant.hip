// Backward (and training-mode forward where the inference kernel fuses too much) of the FA-quantizer's side branches:
// WaveNet gate (modules/commons.py:113-120), StyleEncoder's Mish / Conv1dGLU / masked mean (modules/style_encoder.py)
// and the 2-head self-attention (modules/attentions.py:168-199) with the attention matrix materialised (T <= a few
// hundred frames: (B, heads, T, T) is small) so that dropout on it and its softmax backward are plain row kernels.
#include "common.h"
#include "../../include/facodec_hip.h"

namespace fac {

#define GRID_STRIDE(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// acts = tanh(a1) * sigmoid(a2), a = [a1 | a2] (B, 2C, T):  da1 = d * sig * (1 - th^2), da2 = d * th * sig * (1 - sig)
__global__ void gate_bwd_kernel(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ da, int C, int T, long long n) {
  const long long ct = (long long)C * T;
  GRID_STRIDE(i, n) {
    const long long b = i / ct, r = i - b * ct;
    const float a1 = a[b * 2 * ct + r], a2 = a[b * 2 * ct + ct + r];
    const float th = tanhf(a1), sg = 1.f / (1.f + expf(-a2));
    da[b * 2 * ct + r] = d[i] * sg * (1.f - th * th);
    da[b * 2 * ct + ct + r] = d[i] * th * sg * (1.f - sg);
  }
}

// y = x * tanh(softplus(x))
__global__ void mish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ d, float* __restrict__ dx, long long n) {
  GRID_STRIDE(i, n) {
    const float xv = x[i];
    const float sp = xv > 20.f ? xv : log1pf(expf(xv));
    const float th = tanhf(sp), sg = 1.f / (1.f + expf(-xv));
    dx[i] = d[i] * (th + xv * (1.f - th * th) * sg);
  }
}

__global__ void mish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  GRID_STRIDE(i, n) y[i] = apply_act_slow(x[i], FAC_ACT_MISH);
}

// y = res + a1 * sigmoid(a2) (a (B, 2C, T)): dres = d (caller), da1 = d * sig, da2 = d * a1 * sig (1 - sig)
__global__ void glu_bwd_kernel(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ da, int C, int T, long long n) {
  const long long ct = (long long)C * T;
  GRID_STRIDE(i, n) {
    const long long b = i / ct, r = i - b * ct;
    const float a1 = a[b * 2 * ct + r], a2 = a[b * 2 * ct + ct + r];
    const float sg = 1.f / (1.f + expf(-a2));
    da[b * 2 * ct + r] = d[i] * sg;
    da[b * 2 * ct + ct + r] = d[i] * a1 * sg * (1.f - sg);
  }
}

// out = a * b * scale (elementwise; dropout masks, gradient products)
__global__ void mul_scaled_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, float scale, long long n) {
  GRID_STRIDE(i, n) out[i] = a[i] * b[i] * scale;
}

// d x[b][c][t] = dout[b][c] / len_b   (masked mean: x.sum(2) / mask.sum(2))
__global__ void masked_mean_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ mask, float* __restrict__ dx, int C, int T, long long n) {
  GRID_STRIDE(i, n) {
    const long long bc = i / T;
    const int b = (int)(bc / C);
    float len = (float)T;
    if (mask) {
      len = 0.f;
      for (int t = 0; t < T; ++t) len += mask[(long long)b * T + t];
    }
    dx[i] = dout[bc] / len;
  }
}

// ---- attention with the probability matrix in HBM.  q, k, v, o: (B, H*dk, T); P: (B, H, T, T); mask (B, T) or null.
__global__ __launch_bounds__(256) void attn_probs_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ mask,
                                                         float* __restrict__ P, int H, int dk, int T) {
  extern __shared__ float sc[];    // [T]
  __shared__ float red[256];
  const int t1 = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const float* qb = q + ((long long)b * H + h) * dk * T;
  const float* kb = k + ((long long)b * H + h) * dk * T;
  const float scale = 1.0f / sqrtf((float)dk);
  const float m1 = mask ? mask[(long long)b * T + t1] : 1.f;
  float mx = -INFINITY;
  for (int t2 = tid; t2 < T; t2 += 256) {
    float s = 0.f;
    for (int d = 0; d < dk; ++d) s = fmaf(qb[(long long)d * T + t1] * scale, kb[(long long)d * T + t2], s);
    if (mask && m1 * mask[(long long)b * T + t2] == 0.f) s = -1e4f;
    sc[t2] = s;
    mx = fmaxf(mx, s);
  }
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int t2 = tid; t2 < T; t2 += 256) { const float e = expf(sc[t2] - mx); sc[t2] = e; sum += e; }
  red[tid] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  sum = red[0];
  float* pr = P + (((long long)b * H + h) * T + t1) * T;
  for (int t2 = tid; t2 < T; t2 += 256) pr[t2] = sc[t2] / sum;
}

// The five small matrix products of the attention (o = P v, dv, dP, dq, dk) as ONE tiled kernel (round 6; the per-output
// kernels of rounds 3 - 5 walked P / dS rows T floats apart from neighbouring lanes: 0.62 ms each for 0.1 GFLOP at T = 188):
//   C[n][m] = [scale *] sum_k A(m, k) B(n, k),  k ascending, one fmaf per term -- the naive kernels' order, so the same bits.
// Each operand is either k-contiguous (element (r, k) at r * T + k) or row-contiguous (at k * T + r); a 64 (m) x 16 (n) tile per
// workgroup, k in chunks of 32 through LDS (coalesced loads along whichever index is contiguous), lane = m, wave = 4 of the n.
struct AttnGemm {
  const float* A;
  const float* B;
  float* C;
  long long a_bs, b_bs, c_bs;   // batch strides (one (clip, head) per blockIdx.z)
  int M, N, K, T;               // T: the row pitch of both operands
  float scale;
  int use_scale;
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void attn_gemm_kernel(AttnGemm g) {
  __shared__ float As[32][65];
  __shared__ float Bs[16][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 16;
  const float* A = g.A + (long long)blockIdx.z * g.a_bs;
  const float* B = g.B + (long long)blockIdx.z * g.b_bs;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < g.K; k0 += 32) {
    const int kc = g.K - k0 < 32 ? g.K - k0 : 32;
    if (A_KC) {
      const int kk = tid & 31, mm = tid >> 5;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int m = mm + 8 * p;
        As[kk][m] = (m0 + m < g.M && kk < kc) ? A[(long long)(m0 + m) * g.T + k0 + kk] : 0.f;
      }
    } else {
      const int mm = tid & 63, kk = tid >> 6;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int k = kk + 4 * p;
        As[k][mm] = (m0 + mm < g.M && k < kc) ? A[(long long)(k0 + k) * g.T + m0 + mm] : 0.f;
      }
    }
    if (B_KC) {
      const int kk = tid & 31, nn = tid >> 5;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int n = nn + 8 * p;
        Bs[n][kk] = (n0 + n < g.N && kk < kc) ? B[(long long)(n0 + n) * g.T + k0 + kk] : 0.f;
      }
    } else {
      const int nn = tid & 15, kk = tid >> 4;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int k = kk + 16 * p;
        Bs[nn][k] = (n0 + nn < g.N && k < kc) ? B[(long long)(k0 + k) * g.T + n0 + nn] : 0.f;
      }
    }
    __syncthreads();
    for (int kk = 0; kk < kc; ++kk) {
      const float a = As[kk][lane];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(a, Bs[4 * wave + j][kk], acc[j]);
    }
    __syncthreads();
  }
  const int m = m0 + lane;
  if (m >= g.M) return;
  float* C = g.C + (long long)blockIdx.z * g.c_bs;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + 4 * wave + j;
    if (n < g.N) C[(long long)n * g.M + m] = g.use_scale ? acc[j] * g.scale : acc[j];
  }
}

template <bool A_KC, bool B_KC>
static void attn_gemm(const float* A, const float* B, float* C, long long a_bs, long long b_bs, long long c_bs, int M, int N, int K, int T,
                      int batch, float scale, int use_scale, hipStream_t s) {
  AttnGemm g{A, B, C, a_bs, b_bs, c_bs, M, N, K, T, scale, use_scale};
  hipLaunchKernelGGL((attn_gemm_kernel<A_KC, B_KC>), dim3((M + 63) / 64, (N + 15) / 16, batch), dim3(256), 0, s, g);
}

// softmax backward per row: dS = P * (dP - sum(dP * P)); masked entries carry no gradient.  In place on dP.
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, const float* __restrict__ mask,
                                                               int H, int T) {
  __shared__ float red[256];
  const long long row = blockIdx.x;        // (b*H + h)*T + t1
  const int tid = threadIdx.x;
  const int t1 = (int)(row % T), b = (int)(row / ((long long)H * T));
  const float* pr = P + row * T;
  float* dr = dP + row * T;
  float s = 0.f;
  for (int t2 = tid; t2 < T; t2 += 256) s = fmaf(dr[t2], pr[t2], s);
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  s = red[0];
  const float m1 = mask ? mask[(long long)b * T + t1] : 1.f;
  for (int t2 = tid; t2 < T; t2 += 256) {
    float g = pr[t2] * (dr[t2] - s);
    if (mask && m1 * mask[(long long)b * T + t2] == 0.f) g = 0.f;
    dr[t2] = g;
  }
}

static inline int grid_for(long long n) { return (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535); }

}  // namespace fac

#define LAUNCH1(kern, n, ...) hipLaunchKernelGGL(fac::kern, dim3(fac::grid_for(n)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int fac_gate_bwd(const float* a, const float* d, float* da, int B, int C, int T, fac_stream_t stream) {
  FAC_REQUIRE(a && d && da && B > 0 && C > 0 && T > 0, "gate_bwd: bad arguments");
  const long long n = (long long)B * C * T;
  LAUNCH1(gate_bwd_kernel, n, a, d, da, C, T, n);
  return fac::check_launch("gate_bwd");
}

extern "C" int fac_mish_bwd(const float* x, const float* d, float* dx, int64_t n, fac_stream_t stream) {
  FAC_REQUIRE(x && d && dx && n > 0, "mish_bwd: bad arguments");
  LAUNCH1(mish_bwd_kernel, n, x, d, dx, (long long)n);
  return fac::check_launch("mish_bwd");
}

extern "C" int fac_mish_fwd(const float* x, float* y, int64_t n, fac_stream_t stream) {
  FAC_REQUIRE(x && y && n > 0, "mish_fwd: bad arguments");
  LAUNCH1(mish_fwd_kernel, n, x, y, (long long)n);
  return fac::check_launch("mish_fwd");
}

extern "C" int fac_glu_bwd(const float* a, const float* d, float* da, int B, int C, int T, fac_stream_t stream) {
  FAC_REQUIRE(a && d && da && B > 0 && C > 0 && T > 0, "glu_bwd: bad arguments");
  const long long n = (long long)B * C * T;
  LAUNCH1(glu_bwd_kernel, n, a, d, da, C, T, n);
  return fac::check_launch("glu_bwd");
}

extern "C" int fac_mul_scaled(const float* a, const float* b, float* out, float scale, int64_t n, fac_stream_t stream) {
  FAC_REQUIRE(a && b && out && n > 0, "mul_scaled: bad arguments");
  LAUNCH1(mul_scaled_kernel, n, a, b, out, scale, (long long)n);
  return fac::check_launch("mul_scaled");
}

extern "C" int fac_masked_mean_bwd(const float* dout, const float* mask, float* dx, int B, int C, int T, fac_stream_t stream) {
  FAC_REQUIRE(dout && dx && B > 0 && C > 0 && T > 0, "masked_mean_bwd: bad arguments");
  const long long n = (long long)B * C * T;
  LAUNCH1(masked_mean_bwd_kernel, n, dout, mask, dx, C, T, n);
  return fac::check_launch("masked_mean_bwd");
}

extern "C" int fac_attention_probs(const float* q, const float* k, const float* mask, float* P, int B, int H, int dk, int T,
                                   fac_stream_t stream) {
  FAC_REQUIRE(q && k && P && B > 0 && H > 0 && dk > 0 && T > 0 && T <= 16384, "attention_probs: bad arguments");
  hipLaunchKernelGGL(fac::attn_probs_kernel, dim3(T, H, B), dim3(256), (size_t)T * sizeof(float), (hipStream_t)stream, q, k, mask, P, H, dk, T);
  return fac::check_launch("attention_probs");
}

extern "C" int fac_attention_pv(const float* P, const float* v, float* o, int B, int H, int dk, int T, fac_stream_t stream) {
  FAC_REQUIRE(P && v && o && B > 0 && H > 0 && dk > 0 && T > 0, "attention_pv: bad arguments");
  FAC_REQUIRE((long long)B * H <= 65535, "attention_pv: B * heads too large");
  // o[d][t1] = sum_t2 P[t1][t2] v[d][t2]:  m = t1, n = d, k = t2
  fac::attn_gemm<true, true>(P, v, o, (long long)T * T, (long long)dk * T, (long long)dk * T, T, dk, T, T, B * H, 1.f, 0, (hipStream_t)stream);
  return fac::check_launch("attention_pv");
}

/* Backward of o = P_used v with P_used = P (or P * dropout mask, applied by the caller to dP):
 * step 1: dv from P_used and dO; dP from dO and v.  step 2 (after the caller multiplied dP by the dropout mask):
 * in-place softmax backward with the un-dropped P, then dq, dk. */
extern "C" int fac_attention_bwd_pv(const float* P_used, const float* v, const float* dO, float* dv, float* dP, int B, int H,
                                    int dk, int T, fac_stream_t stream) {
  FAC_REQUIRE(P_used && v && dO && dv && dP && B > 0 && H > 0 && dk > 0 && T > 0, "attention_bwd_pv: bad arguments");
  FAC_REQUIRE((long long)B * H <= 65535, "attention_bwd_pv: B * heads too large");
  const long long tt = (long long)T * T, dt = (long long)dk * T;
  // dv[d][t2] = sum_t1 P[t1][t2] dO[d][t1]:  m = t2, n = d, k = t1 (P row-contiguous in m)
  fac::attn_gemm<false, true>(P_used, dO, dv, tt, dt, dt, T, dk, T, T, B * H, 1.f, 0, (hipStream_t)stream);
  // dP[t1][t2] = sum_d dO[d][t1] v[d][t2]:  m = t2, n = t1, k = d (both operands row-contiguous)
  fac::attn_gemm<false, false>(v, dO, dP, dt, dt, tt, T, T, dk, T, B * H, 1.f, 0, (hipStream_t)stream);
  return fac::check_launch("attention_bwd_pv");
}

extern "C" int fac_attention_bwd_qk(const float* P, float* dP, const float* q, const float* k, const float* mask, float* dq,
                                    float* dk_, int B, int H, int dk, int T, fac_stream_t stream) {
  FAC_REQUIRE(P && dP && q && k && dq && dk_ && B > 0 && H > 0 && dk > 0 && T > 0, "attention_bwd_qk: bad arguments");
  hipLaunchKernelGGL(fac::attn_softmax_bwd_kernel, dim3((unsigned)((long long)B * H * T)), dim3(256), 0, (hipStream_t)stream, P, dP, mask, H, T);
  FAC_REQUIRE((long long)B * H <= 65535, "attention_bwd_qk: B * heads too large");
  const long long tt = (long long)T * T, dt = (long long)dk * T;
  const float scale = 1.0f / sqrtf((float)dk);
  // dq[d][t1] = scale * sum_t2 dS[t1][t2] k[d][t2];  dk[d][t2] = scale * sum_t1 dS[t1][t2] q[d][t1]
  fac::attn_gemm<true, true>(dP, k, dq, tt, dt, dt, T, dk, T, T, B * H, scale, 1, (hipStream_t)stream);
  fac::attn_gemm<false, true>(dP, q, dk_, tt, dt, dt, T, dk, T, T, B * H, scale, 1, (hipStream_t)stream);
  return fac::check_launch("attention_bwd_qk");
}
