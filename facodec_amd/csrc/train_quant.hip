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

// o[b][h*dk+d][t1] = sum_t2 P[t1][t2] v[d][t2]
__global__ void attn_pv_kernel(const float* __restrict__ P, const float* __restrict__ v, float* __restrict__ o, int H, int dk, int T, long long n) {
  GRID_STRIDE(i, n) {
    const int t1 = (int)(i % T);
    const long long r = i / T;            // (b*H + h)*dk + d
    const long long bh = r / dk;
    const float* pr = P + (bh * T + t1) * T;
    const float* vr = v + r * T;
    float s = 0.f;
    for (int t2 = 0; t2 < T; ++t2) s = fmaf(pr[t2], vr[t2], s);
    o[i] = s;
  }
}

// dV[d][t2] = sum_t1 P[t1][t2] dO[d][t1]
__global__ void attn_dv_kernel(const float* __restrict__ P, const float* __restrict__ dO, float* __restrict__ dv, int H, int dk, int T, long long n) {
  GRID_STRIDE(i, n) {
    const int t2 = (int)(i % T);
    const long long r = i / T;
    const long long bh = r / dk;
    const float* pb = P + bh * T * T + t2;
    const float* dr = dO + r * T;
    float s = 0.f;
    for (int t1 = 0; t1 < T; ++t1) s = fmaf(pb[(long long)t1 * T], dr[t1], s);
    dv[i] = s;
  }
}

// dP[t1][t2] = sum_d dO[d][t1] v[d][t2]
__global__ void attn_dp_kernel(const float* __restrict__ dO, const float* __restrict__ v, float* __restrict__ dP, int H, int dk, int T, long long n) {
  GRID_STRIDE(i, n) {
    const int t2 = (int)(i % T);
    const long long r = i / T;
    const int t1 = (int)(r % T);
    const long long bh = r / T;
    const float* dob = dO + bh * dk * T + t1;
    const float* vb = v + bh * dk * T + t2;
    float s = 0.f;
    for (int d = 0; d < dk; ++d) s = fmaf(dob[(long long)d * T], vb[(long long)d * T], s);
    dP[i] = s;
  }
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

// dQ[d][t1] = scale * sum_t2 dS[t1][t2] k[d][t2];  dK[d][t2] = scale * sum_t1 dS[t1][t2] q[d][t1]
__global__ void attn_dq_kernel(const float* __restrict__ dS, const float* __restrict__ k, float* __restrict__ dq, int H, int dk, int T, long long n) {
  const float scale = 1.0f / sqrtf((float)dk);
  GRID_STRIDE(i, n) {
    const int t1 = (int)(i % T);
    const long long r = i / T;
    const long long bh = r / dk;
    const float* sr = dS + (bh * T + t1) * T;
    const float* kr = k + r * T;
    float s = 0.f;
    for (int t2 = 0; t2 < T; ++t2) s = fmaf(sr[t2], kr[t2], s);
    dq[i] = s * scale;
  }
}

__global__ void attn_dk_kernel(const float* __restrict__ dS, const float* __restrict__ q, float* __restrict__ dk_, int H, int dk, int T, long long n) {
  const float scale = 1.0f / sqrtf((float)dk);
  GRID_STRIDE(i, n) {
    const int t2 = (int)(i % T);
    const long long r = i / T;
    const long long bh = r / dk;
    const float* sb = dS + bh * T * T + t2;
    const float* qr = q + r * T;
    float s = 0.f;
    for (int t1 = 0; t1 < T; ++t1) s = fmaf(sb[(long long)t1 * T], qr[t1], s);
    dk_[i] = s * scale;
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
  const long long n = (long long)B * H * dk * T;
  LAUNCH1(attn_pv_kernel, n, P, v, o, H, dk, T, n);
  return fac::check_launch("attention_pv");
}

/* Backward of o = P_used v with P_used = P (or P * dropout mask, applied by the caller to dP):
 * step 1: dv from P_used and dO; dP from dO and v.  step 2 (after the caller multiplied dP by the dropout mask):
 * in-place softmax backward with the un-dropped P, then dq, dk. */
extern "C" int fac_attention_bwd_pv(const float* P_used, const float* v, const float* dO, float* dv, float* dP, int B, int H,
                                    int dk, int T, fac_stream_t stream) {
  FAC_REQUIRE(P_used && v && dO && dv && dP && B > 0 && H > 0 && dk > 0 && T > 0, "attention_bwd_pv: bad arguments");
  const long long n = (long long)B * H * dk * T, np = (long long)B * H * T * T;
  LAUNCH1(attn_dv_kernel, n, P_used, dO, dv, H, dk, T, n);
  LAUNCH1(attn_dp_kernel, np, dO, v, dP, H, dk, T, np);
  return fac::check_launch("attention_bwd_pv");
}

extern "C" int fac_attention_bwd_qk(const float* P, float* dP, const float* q, const float* k, const float* mask, float* dq,
                                    float* dk_, int B, int H, int dk, int T, fac_stream_t stream) {
  FAC_REQUIRE(P && dP && q && k && dq && dk_ && B > 0 && H > 0 && dk > 0 && T > 0, "attention_bwd_qk: bad arguments");
  hipLaunchKernelGGL(fac::attn_softmax_bwd_kernel, dim3((unsigned)((long long)B * H * T)), dim3(256), 0, (hipStream_t)stream, P, dP, mask, H, T);
  const long long n = (long long)B * H * dk * T;
  LAUNCH1(attn_dq_kernel, n, dP, k, dq, H, dk, T, n);
  LAUNCH1(attn_dk_kernel, n, dP, q, dk_, H, dk, T, n);
  return fac::check_launch("attention_bwd_qk");
}
