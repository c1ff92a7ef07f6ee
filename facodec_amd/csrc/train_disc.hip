// Layout / elementwise kernels of the discriminator path (dac/model/discriminator.py; train.py:280-312).  The
// discriminators' convolutions themselves run on the 1-D conv kernels: an MPD (k,1) Conv2d is a 1-D conv along the
// folded time axis with the period as extra batch; an MRD (3,k) Conv2d is a 1-D conv along frequency over the three
// neighbouring time rows stacked into the channel axis.
#include "common.h"
#include "../../include/facodec_hip.h"

namespace fac {

#define GRID_STRIDE(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
static inline int grid_for(long long n) { return (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535); }

// Optional position mask for row-concatenated signals (MPD): only positions t with t % pitch < valid inside each
// channel row of length T carry data; the rest are the zero gaps that stand in for the convs' padding.
// MRD (rows_per_group > 0): additionally the rows r = position / pitch with r % rows_per_group >= valid_rows are zero (the
// separator row after each clip's frames).
__device__ __forceinline__ bool leaky_pos_ok(long long i, int T, int pitch, int valid, int rpg, int vrows) {
  if (pitch == 0) return true;
  const int pos = (int)(i % T);
  const int row = pos / pitch;
  if (pos - row * pitch >= valid) return false;
  return rpg == 0 || row % rpg < vrows;
}
__global__ void leaky_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float slope, int T, int pitch, int valid, int rpg,
                                 int vrows, long long n) {
  GRID_STRIDE(i, n) {
    const float v = x[i];
    y[i] = leaky_pos_ok(i, T, pitch, valid, rpg, vrows) ? (v > 0.f ? v : v * slope) : 0.f;
  }
}
__global__ void leaky_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, float slope, int T, int pitch,
                                 int valid, int rpg, int vrows, long long n) {
  GRID_STRIDE(i, n) {
    dx[i] = leaky_pos_ok(i, T, pitch, valid, rpg, vrows) ? (x[i] > 0.f ? dy[i] : dy[i] * slope) : 0.f;
  }
}

// Four consecutive positions per thread (16-byte accesses, one position decode per quad).  Needs T and the pitch to be
// multiples of 4 (a quad never straddles two rows) and 16-byte aligned tensors; the scalar kernels above take the rest.
template <bool BWD>
__global__ void leaky_quad_kernel(const float4* __restrict__ x, const float4* __restrict__ dy, float4* __restrict__ out, float slope,
                                  int T, int pitch, int valid, int rpg, int vrows, long long n4) {
  GRID_STRIDE(q, n4) {
    const float4 xv = x[q];
    const float4 gv = BWD ? dy[q] : xv;
    int left = 4;                                   // how many of the quad's positions carry data
    if (pitch != 0) {
      const long long i = 4 * q;
      const int pos = (i >> 31) == 0 ? (int)((unsigned)i % (unsigned)T) : (int)(i % T);
      const int row = pos / pitch;
      const int c = pos - row * pitch;
      left = (rpg == 0 || row % rpg < vrows) ? valid - c : 0;
    }
    float4 o;
    o.x = left > 0 ? (xv.x > 0.f ? gv.x : gv.x * slope) : 0.f;
    o.y = left > 1 ? (xv.y > 0.f ? gv.y : gv.y * slope) : 0.f;
    o.z = left > 2 ? (xv.z > 0.f ? gv.z : gv.z * slope) : 0.f;
    o.w = left > 3 ? (xv.w > 0.f ? gv.w : gv.w * slope) : 0.f;
    out[q] = o;
  }
}

// The same for any pitch / row length (round 6): four consecutive elements per lane with 16-byte accesses; the position of the
// quad's first element is resolved once (one 32-bit modulo, one division) and walked forward from there -- the element-wise
// kernels above cost a 64-bit modulo and two divisions per element and ran at 3.2 - 3.5 TB/s (profiles/r06_pmc_train_before.json).
template <bool BWD>
__global__ void leaky_quad_any_kernel(const float4* __restrict__ x, const float4* __restrict__ dy, float4* __restrict__ out, float slope,
                                      int T, int pitch, int valid, int rpg, int vrows, long long n4) {
  GRID_STRIDE(q, n4) {
    const float4 xv = x[q];
    const float4 gv = BWD ? dy[q] : xv;
    const long long i = 4 * q;
    int pos = (i >> 31) == 0 ? (int)((unsigned)i % (unsigned)T) : (int)(i % T);
    int row = pos / pitch;
    int c = pos - row * pitch;
    int rr = rpg ? row % rpg : 0;
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = c < valid && (rpg == 0 || rr < vrows);
      o[j] = ok ? (xs[j] > 0.f ? gs[j] : gs[j] * slope) : 0.f;
      ++c; ++pos;
      if (pos == T) { pos = 0; c = 0; rr = 0; }
      else if (c == pitch) { c = 0; if (++rr == rpg) rr = 0; }
    }
    out[q] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// out[(b*p + j)*pitch + l] = xr[l*p + j] for l < L (zero for L <= l < pitch), xr = x reflect-extended on the right to
// L*p samples (MPD.pad_to_period + rearrange); rows (b, j) laid one after another with `pitch` columns each
__global__ void period_fold_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int p, int L, int pitch, long long n) {
  GRID_STRIDE(i, n) {
    const int l = (int)(i % pitch);
    const long long r = i / pitch;
    const int j = (int)(r % p);
    const long long b = r / p;
    int s = l * p + j;
    if (s >= T) s = 2 * (T - 1) - s;
    out[i] = l < L ? x[b * T + s] : 0.f;
  }
}
__global__ void period_fold_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int T, int p, int L, int pitch, long long n) {
  GRID_STRIDE(i, n) {
    const int s = (int)(i % T);
    const long long b = i / T;
    float g = dout[(b * p + s % p) * pitch + s / p];
    const int s2 = 2 * (T - 1) - s;            // the reflected copy of sample s, if it lies in the padded tail
    if (s2 >= T && s2 < L * p) g += dout[(b * p + s2 % p) * pitch + s2 / p];
    dx[i] = g;
  }
}

// dy (rows, T) -> up (rows, (T-1)*s + 1) with zeros between samples (data gradient of a strided conv as a stride-1 conv)
__global__ void zero_insert_kernel(const float* __restrict__ dy, float* __restrict__ up, int T, int s, int Tu, long long n) {
  GRID_STRIDE(i, n) {
    const int u = (int)(i % Tu);
    const long long r = i / Tu;
    up[i] = (u % s == 0) ? dy[r * T + u / s] : 0.f;
  }
}

// rows = (b, t) of a (B*T, C, F) tensor: stk[(b,t)][dt*C + c][f] = x[(b, t+dt-1)][c][f] (zero outside 0 <= t+dt-1 < T)
__global__ void row_stack3_kernel(const float* __restrict__ x, float* __restrict__ stk, int T, int C, int F, long long n) {
  const long long cf = (long long)C * F;
  GRID_STRIDE(i, n) {
    const long long row = i / (3 * cf);
    const long long r = i - row * 3 * cf;
    const int dt = (int)(r / cf);
    const long long rest = r - dt * cf;
    const int t = (int)(row % T) + dt - 1;
    stk[i] = (t >= 0 && t < T) ? x[(row + dt - 1) * cf + rest] : 0.f;
  }
}
__global__ void row_stack3_bwd_kernel(const float* __restrict__ dstk, float* __restrict__ dx, int T, int C, int F, long long n) {
  const long long cf = (long long)C * F;
  GRID_STRIDE(i, n) {
    const long long row = i / cf;
    const long long rest = i - row * cf;
    const int t = (int)(row % T);
    float g = 0.f;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const int ts = t - dt + 1;               // the stacked row that read x[row] through its tap dt
      if (ts >= 0 && ts < T) g += dstk[((row - dt + 1) * 3 + dt) * cf + rest];
    }
    dx[i] = g;
  }
}

// spec (B, 2*Ft, T) = [re | im] rows -> rows[(b*T + t)][c][f] = spec[b][c*Ft + f0 + f][t], f < Fb (one frequency band)
__global__ void spec_to_rows_kernel(const float* __restrict__ spec, float* __restrict__ rows, int Ft, int T, int f0, int Fb, long long n) {
  GRID_STRIDE(i, n) {
    const int f = (int)(i % Fb);
    const long long r = i / Fb;
    const int c = (int)(r & 1);
    const long long bt = r >> 1;
    const long long b = bt / T;
    const int t = (int)(bt - b * T);
    rows[i] = spec[(b * 2 * Ft + (long long)c * Ft + f0 + f) * T + t];
  }
}
// adjoint, accumulating the band into dspec (bands are disjoint: plain stores into a zero-initialised buffer)
__global__ void spec_to_rows_bwd_kernel(const float* __restrict__ drows, float* __restrict__ dspec, int Ft, int T, int f0, int Fb, long long n) {
  GRID_STRIDE(i, n) {
    const int f = (int)(i % Fb);
    const long long r = i / Fb;
    const int c = (int)(r & 1);
    const long long bt = r >> 1;
    const long long b = bt / T;
    const int t = (int)(bt - b * T);
    dspec[(b * 2 * Ft + (long long)c * Ft + f0 + f) * T + t] = drows[i];
  }
}

// Row-concatenated band of the spectrogram: cat[c][(b*(T+1) + t)*pitch + f] = spec[b][c*Ft + f0 + f][t] for t < T, f < Fb; zero in
// the gap columns and in the separator row t = T of every clip.
__global__ void spec_to_cat_kernel(const float* __restrict__ spec, float* __restrict__ cat, int Ft, int T, int f0, int Fb, int pitch,
                                   long long per_c, long long n) {
  GRID_STRIDE(i, n) {
    const int c = (int)(i / per_c);
    const long long o = i - c * per_c;
    const int f = (int)(o % pitch);
    const long long r = o / pitch;
    const int t = (int)(r % (T + 1));
    const long long b = r / (T + 1);
    cat[i] = (f < Fb && t < T) ? spec[(b * 2 * Ft + (long long)c * Ft + f0 + f) * T + t] : 0.f;
  }
}
// adjoint: one thread per (b, c, f, t) of the band
__global__ void spec_to_cat_bwd_kernel(const float* __restrict__ dcat, float* __restrict__ dspec, int Ft, int T, int f0, int Fb, int pitch,
                                       long long per_c, long long n) {
  GRID_STRIDE(i, n) {
    const int t = (int)(i % T);
    long long r = i / T;
    const int f = (int)(r % Fb);
    r /= Fb;
    const int c = (int)(r & 1);
    const long long b = r >> 1;
    dspec[(b * 2 * Ft + (long long)c * Ft + f0 + f) * T + t] = dcat[c * per_c + (b * (T + 1) + t) * pitch + f];
  }
}

// out[b][i] = x[b][reflect(i - pad_l)], i < T + pad_l + pad_r
__global__ void pad_reflect_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int pad_l, int Tp, long long n) {
  GRID_STRIDE(i, n) {
    const int u = (int)(i % Tp);
    const long long b = i / Tp;
    int s = u - pad_l;
    if (s < 0) s = -s;
    if (s >= T) s = 2 * (T - 1) - s;
    out[i] = x[b * T + s];
  }
}

// Discriminator.preprocess: z = 0.8 (x - mean) / (max|x - mean| + 1e-9); stats[b] = (mean, max, argmax, sign)
__global__ __launch_bounds__(256) void disc_pre_fwd_kernel(const float* __restrict__ x, float* __restrict__ z, float* __restrict__ stats, int T) {
  __shared__ float red[256];
  __shared__ int redi[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (long long)b * T;
  float s = 0.f;
  for (int t = tid; t < T; t += 256) s += xr[t];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const float mean = red[0] / (float)T;
  __syncthreads();
  float m = -1.f;
  int mi = 0;
  for (int t = tid; t < T; t += 256) { const float a = fabsf(xr[t] - mean); if (a > m) { m = a; mi = t; } }
  red[tid] = m;
  redi[tid] = mi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o && (red[tid + o] > red[tid] || (red[tid + o] == red[tid] && redi[tid + o] < redi[tid]))) { red[tid] = red[tid + o]; redi[tid] = redi[tid + o]; }
    __syncthreads();
  }
  const float mx = red[0];
  const int am = redi[0];
  const float c = 0.8f / (mx + 1e-9f);
  for (int t = tid; t < T; t += 256) z[(long long)b * T + t] = (xr[t] - mean) * c;
  if (tid == 0) { stats[b * 4] = mean; stats[b * 4 + 1] = mx; stats[b * 4 + 2] = (float)am; stats[b * 4 + 3] = (xr[am] - mean) >= 0.f ? 1.f : -1.f; }
}

__global__ __launch_bounds__(256) void disc_pre_bwd_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ dz, float* __restrict__ dx, int T) {
  __shared__ float red[2][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (long long)b * T;
  const float* dr = dz + (long long)b * T;
  const float mean = stats[b * 4], mx = stats[b * 4 + 1], sgn = stats[b * 4 + 3];
  const int am = (int)stats[b * 4 + 2];
  const float c = 0.8f / (mx + 1e-9f);
  float s1 = 0.f, s2 = 0.f;                      // sum dz * y0, sum dz
  for (int t = tid; t < T; t += 256) { s1 = fmaf(dr[t], xr[t] - mean, s1); s2 += dr[t]; }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; } __syncthreads(); }
  // dy0_j = c dz_j - [j = am] sgn * 0.8 / (mx+eps)^2 * S1;   dx = dy0 - mean(dy0)
  const float spike = sgn * (0.8f / ((mx + 1e-9f) * (mx + 1e-9f))) * red[0][0];
  const float mean_dy0 = (c * red[1][0] - spike) / (float)T;
  for (int t = tid; t < T; t += 256) dx[(long long)b * T + t] = c * dr[t] - (t == am ? spike : 0.f) - mean_dy0;
}

}  // namespace fac

#define L1(kern, n, ...) hipLaunchKernelGGL(fac::kern, dim3(fac::grid_for(n)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int fac_leaky_relu(const float* x, const float* dy, float* out, int64_t n, float slope, int T, int pitch, int valid,
                              int rows_per_group, int valid_rows, fac_stream_t stream) {
  FAC_REQUIRE(x && out && n > 0 && (pitch == 0 || (T > 0 && valid > 0 && valid <= pitch)), "leaky_relu: bad arguments");
  FAC_REQUIRE(rows_per_group == 0 || (pitch > 0 && valid_rows > 0 && valid_rows <= rows_per_group), "leaky_relu: bad row groups");
  const bool quad = (n & 3) == 0 && (pitch == 0 || ((T & 3) == 0 && (pitch & 3) == 0)) &&
                    ((reinterpret_cast<unsigned long long>(x) | reinterpret_cast<unsigned long long>(out) |
                      reinterpret_cast<unsigned long long>(dy)) & 15) == 0;
  if (quad) {
    const long long n4 = n / 4;
    const float4 *x4 = reinterpret_cast<const float4*>(x), *d4 = reinterpret_cast<const float4*>(dy);
    float4* o4 = reinterpret_cast<float4*>(out);
    if (dy) L1(leaky_quad_kernel<true>, n4, x4, d4, o4, slope, T > 0 ? T : 1, pitch, valid, rows_per_group, valid_rows, n4);
    else L1(leaky_quad_kernel<false>, n4, x4, d4, o4, slope, T > 0 ? T : 1, pitch, valid, rows_per_group, valid_rows, n4);
  } else if (pitch != 0 && (n & 3) == 0 && ((reinterpret_cast<unsigned long long>(x) | reinterpret_cast<unsigned long long>(out) |
                                               reinterpret_cast<unsigned long long>(dy)) & 15) == 0) {
    const long long n4 = n / 4;
    const float4 *x4 = reinterpret_cast<const float4*>(x), *d4 = reinterpret_cast<const float4*>(dy);
    float4* o4 = reinterpret_cast<float4*>(out);
    if (dy) L1(leaky_quad_any_kernel<true>, n4, x4, d4, o4, slope, T, pitch, valid, rows_per_group, valid_rows, n4);
    else L1(leaky_quad_any_kernel<false>, n4, x4, d4, o4, slope, T, pitch, valid, rows_per_group, valid_rows, n4);
  } else if (dy) L1(leaky_bwd_kernel, n, x, dy, out, slope, T > 0 ? T : 1, pitch, valid, rows_per_group, valid_rows, (long long)n);
  else L1(leaky_fwd_kernel, n, x, out, slope, T > 0 ? T : 1, pitch, valid, rows_per_group, valid_rows, (long long)n);
  return fac::check_launch("leaky_relu");
}

extern "C" int fac_period_fold(const float* x, float* out, int B, int T, int period, int L, int pitch, int backward,
                               fac_stream_t stream) {
  FAC_REQUIRE(x && out && B > 0 && T > 1 && period > 0 && (long long)L * period >= T && (long long)L * period <= 2ll * T - 1 &&
                  pitch >= L,
              "period_fold: bad arguments");
  if (backward) { const long long n = (long long)B * T; L1(period_fold_bwd_kernel, n, x, out, T, period, L, pitch, n); }
  else { const long long n = (long long)B * period * pitch; L1(period_fold_kernel, n, x, out, T, period, L, pitch, n); }
  return fac::check_launch("period_fold");
}

extern "C" int fac_zero_insert(const float* dy, float* up, int64_t rows, int T, int stride, fac_stream_t stream) {
  FAC_REQUIRE(dy && up && rows > 0 && T > 0 && stride > 0, "zero_insert: bad arguments");
  const int Tu = (T - 1) * stride + 1;
  const long long n = (long long)rows * Tu;
  L1(zero_insert_kernel, n, dy, up, T, stride, Tu, n);
  return fac::check_launch("zero_insert");
}

extern "C" int fac_row_stack3(const float* x, float* out, int64_t rows, int T, int C, int F, int backward, fac_stream_t stream) {
  FAC_REQUIRE(x && out && rows > 0 && T > 0 && rows % T == 0 && C > 0 && F > 0, "row_stack3: bad arguments");
  if (backward) { const long long n = (long long)rows * C * F; L1(row_stack3_bwd_kernel, n, x, out, T, C, F, n); }
  else { const long long n = (long long)rows * 3 * C * F; L1(row_stack3_kernel, n, x, out, T, C, F, n); }
  return fac::check_launch("row_stack3");
}

extern "C" int fac_spec_to_rows(const float* src, float* dst, int B, int Ft, int T, int f0, int Fb, int backward, fac_stream_t stream) {
  FAC_REQUIRE(src && dst && B > 0 && Ft > 0 && T > 0 && f0 >= 0 && Fb > 0 && f0 + Fb <= Ft, "spec_to_rows: bad arguments");
  const long long n = (long long)B * T * 2 * Fb;
  if (backward) L1(spec_to_rows_bwd_kernel, n, src, dst, Ft, T, f0, Fb, n);
  else L1(spec_to_rows_kernel, n, src, dst, Ft, T, f0, Fb, n);
  return fac::check_launch("spec_to_rows");
}

extern "C" int fac_spec_to_cat(const float* src, float* dst, int B, int Ft, int T, int f0, int Fb, int pitch, int backward,
                               fac_stream_t stream) {
  FAC_REQUIRE(src && dst && B > 0 && Ft > 0 && T > 0 && f0 >= 0 && Fb > 0 && f0 + Fb <= Ft && pitch >= Fb, "spec_to_cat: bad arguments");
  const long long per_c = (long long)B * (T + 1) * pitch;
  if (backward) { const long long n = (long long)B * 2 * Fb * T; L1(spec_to_cat_bwd_kernel, n, src, dst, Ft, T, f0, Fb, pitch, per_c, n); }
  else { const long long n = 2 * per_c; L1(spec_to_cat_kernel, n, src, dst, Ft, T, f0, Fb, pitch, per_c, n); }
  return fac::check_launch("spec_to_cat");
}

extern "C" int fac_pad_reflect(const float* x, float* out, int B, int T, int pad_l, int pad_r, fac_stream_t stream) {
  FAC_REQUIRE(x && out && B > 0 && T > 1 && pad_l >= 0 && pad_r >= 0 && pad_l < T && pad_r < T, "pad_reflect: bad arguments");
  const int Tp = T + pad_l + pad_r;
  const long long n = (long long)B * Tp;
  L1(pad_reflect_kernel, n, x, out, T, pad_l, Tp, n);
  return fac::check_launch("pad_reflect");
}

extern "C" int fac_disc_preprocess(const float* x, const float* dz, float* out, float* stats, int B, int T, fac_stream_t stream) {
  FAC_REQUIRE(x && out && stats && B > 0 && T > 0, "disc_preprocess: bad arguments");
  if (dz) hipLaunchKernelGGL(fac::disc_pre_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, stats, dz, out, T);
  else hipLaunchKernelGGL(fac::disc_pre_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, out, stats, T);
  return fac::check_launch("disc_preprocess");
}
