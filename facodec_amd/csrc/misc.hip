// Small HBM-bound fused ops of the quantizer front-end and the spectral losses, plus the
// library's error plumbing.  Each kernel cites the reference op it replaces.
#include "common.h"
#include <string.h>

namespace fac {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int EW_THREADS = 256;
inline int ew_grid(long long n, int per_thread = 1) {
  long long g = (n + (long long)EW_THREADS * per_thread - 1) / ((long long)EW_THREADS * per_thread);
  if (g > 8192) g = 8192;  // grid-stride beyond 256 CUs x 32
  if (g < 1) g = 1;
  return (int)g;
}

// dac/nn/layers.py:18-33
__global__ void snake_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                             float* __restrict__ y, int C, int T, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / T) % C);
    const float al = alpha[c];
    y[i] = snake_apply(x[i], al, snake_inv(al));
  }
}

// Row-wise form for long rows: workgroup = (time chunk, row (b, c)); alpha and its reciprocal once per workgroup, no
// per-element index division (the flat kernel above spends more on `(i / T) % C` and the division than on the sine).
__global__ __launch_bounds__(256) void snake_rows_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                         float* __restrict__ y, int C, int T) {
  const long long row = blockIdx.y;
  const float al = alpha[(int)(row % C)];
  const float inv = snake_inv(al);
  const float* xr = x + row * T;
  float* yr = y + row * T;
  for (int t = blockIdx.x * 1024 + threadIdx.x; t < T && t < (blockIdx.x + 1) * 1024; t += 256) yr[t] = snake_apply(xr[t], al, inv);
}

// modules/commons.py:113-120; g (B, 2C) is the per-clip conditioning row added before the gate
// (WN with gin_channels: g_l broadcast over time, modules/wavenet.py:146-155), or NULL.
__global__ void gate_kernel(const float* __restrict__ a, const float* __restrict__ g,
                            float* __restrict__ out, int C, int T, long long g_bs, long long n) {
  const long long ct = (long long)C * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / ct, r = i - b * ct;
    float ta = a[b * 2 * ct + r];
    float sa = a[b * 2 * ct + ct + r];
    if (g) {
      const int c = (int)(r / T);
      ta = __fadd_rn(ta, g[b * g_bs + c]);
      sa = __fadd_rn(sa, g[b * g_bs + C + c]);
    }
    out[i] = __fmul_rn(tanhf(ta), sigmoid_f(sa));
  }
}

// Redecoder input (modules/redecoder.py:35-45): x[b, :, t] = sum_i table_i[codes[b, i, t], :], written
// channel-major (B, E, T) for the conv stack.  tables (n_tab, V, E) row-major; codes (B, n_codes, T) int64,
// table i reads code row code_row0 + i.
__global__ void embed_sum_kernel(const long long* __restrict__ codes, const float* __restrict__ tables,
                                 float* __restrict__ out, int n_tab, int n_codes, int code_row0, int V,
                                 int E, int T, long long n, int accumulate) {
  const long long et = (long long)E * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / et, r = i - b * et;
    const int e = (int)(r / T), t = (int)(r - (long long)e * T);
    float s = accumulate ? out[i] : 0.f;
    for (int k = 0; k < n_tab; ++k) {
      const long long id = codes[(b * n_codes + code_row0 + k) * T + t];
      s = __fadd_rn(s, tables[((long long)k * V + id) * E + e]);
    }
    out[i] = s;
  }
}

// modules/style_encoder.py:26-31 (dropout is identity in eval)
__global__ void glu_res_kernel(const float* __restrict__ a, const float* __restrict__ res,
                               float* __restrict__ out, int C, int T, long long n) {
  const long long ct = (long long)C * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / ct, r = i - b * ct;
    const float x1 = a[b * 2 * ct + r];
    const float x2 = a[b * 2 * ct + ct + r];
    out[i] = __fadd_rn(res[i], __fmul_rn(x1, sigmoid_f(x2)));
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                           float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = __fadd_rn(a[i], b[i]);
}

// modules/quantize.py:410  residual_feature = x - z_p - z_c
__global__ void sub2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                            const float* __restrict__ c, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = __fsub_rn(__fsub_rn(a[i], b[i]), c[i]);
}

__global__ void mul_mask_kernel(float* __restrict__ x, const float* __restrict__ mask, int C, int T,
                                long long n) {
  const long long ct = (long long)C * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / ct;
    const int t = (int)(i % T);
    x[i] = __fmul_rn(x[i], mask[b * T + t]);
  }
}

// streaming left-context buffer: one workgroup per row; the tail is read into
// registers before any lane overwrites the front, so overlapping source / destination are safe
__global__ __launch_bounds__(256) void stream_push_kernel(float* __restrict__ buf, const float* __restrict__ src,
                                                          long long cap, int hist, int n_prev, int n_new) {
  float* row = buf + (long long)blockIdx.x * cap;
  if (n_prev > 0 && hist > 0) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = threadIdx.x + 256 * i;
      r[i] = idx < hist ? row[n_prev + idx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = threadIdx.x + 256 * i;
      if (idx < hist) row[idx] = r[i];
    }
  }
  const float* s = src + (long long)blockIdx.x * n_new;
  for (int j = threadIdx.x; j < n_new; j += 256) row[hist + j] = s[j];
}

// modules/wavenet.py:159-165
__global__ void wn_res_skip_kernel(const float* __restrict__ rs, float* __restrict__ x,
                                   float* __restrict__ out, int C, int T, int last, long long n) {
  const long long ct = (long long)C * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    if (last) {
      out[i] = __fadd_rn(out[i], rs[i]);
    } else {
      const long long b = i / ct, r = i - b * ct;
      x[i] = __fadd_rn(x[i], rs[b * 2 * ct + r]);
      out[i] = __fadd_rn(out[i], rs[b * 2 * ct + ct + r]);
    }
  }
}

// modules/attentions.py:168-199 for the StyleEncoder's 2-head self-attention: one workgroup per
// (b, head, 16-query tile); K/V rows are streamed from L2, scores live in LDS.  ~50 MFLOP per
// clip -- not worth MFMA plumbing.
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q,
                                                        const float* __restrict__ k,
                                                        const float* __restrict__ v,
                                                        const float* __restrict__ mask,
                                                        float* __restrict__ out, int n_heads,
                                                        int dk, int T, int stage_v) {
  extern __shared__ float sm[];
  constexpr int QT = 16;
  float* qs = sm;            // [dk][QT]   (pre-scaled queries)
  float* sc = qs + dk * QT;  // [QT][T]
  const int tq0 = blockIdx.x * QT;
  const int h = blockIdx.y;
  const int b = blockIdx.z;
  const long long base = ((long long)b * n_heads + h) * dk * T;
  const float* qg = q + base;
  const float* kg = k + base;
  const float* vg = v + base;
  const float scale = sqrtf((float)dk);
  for (int i = threadIdx.x; i < dk * QT; i += 256) {
    const int d = i / QT, j = i - d * QT;
    const int tq = tq0 + j;
    qs[i] = tq < T ? __fdiv_rn(qg[(long long)d * T + tq], scale) : 0.f;
  }
  __syncthreads();
  // scores[j][tk] = sum_d qs[d][j] * k[d][tk]
  for (int tk = threadIdx.x; tk < T; tk += 256) {
    float acc[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) acc[j] = 0.f;
    for (int d = 0; d < dk; ++d) {
      const float kv = kg[(long long)d * T + tk];
#pragma unroll
      for (int j = 0; j < QT; ++j) acc[j] = fmaf(qs[d * QT + j], kv, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      float s = acc[j];
      const int tq = tq0 + j;
      if (mask && tq < T) {
        const float m = mask[(long long)b * T + tq] * mask[(long long)b * T + tk];
        if (m == 0.f) s = -1e4f;
      }
      sc[j * T + tk] = s;
    }
  }
  __syncthreads();
  // softmax over tk: one wave handles 4 query rows
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < QT; j += 4) {
      float mx = -INFINITY;
      for (int tk = lane; tk < T; tk += 64) mx = fmaxf(mx, sc[j * T + tk]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float sum = 0.f;
      for (int tk = lane; tk < T; tk += 64) {
        const float e = expf(sc[j * T + tk] - mx);
        sc[j * T + tk] = e;
        sum += e;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      for (int tk = lane; tk < T; tk += 64) sc[j * T + tk] = __fdiv_rn(sc[j * T + tk], sum);
    }
  }
  __syncthreads();
  // out[d][tq] = sum_tk p[tq][tk] * v[d][tk], tk ascending.  Round 6: the 16 rows of V a pass of 256 (d, query) pairs needs go through
  // LDS first (coalesced rows); before, every lane walked its own row of V in global memory, 188 dependent strided loads per output:
  // 0.94 ms per B = 32 call for 0.6 GFLOP.  Same products in the same order: same bits.
  // (stage_v == 0: T too long for the second LDS tile -- the rows are read in place, as before.)
  float* vs = sc + QT * T;   // [16][T + 1]
  for (int i0 = 0; i0 < dk * QT; i0 += 256) {
    const int d0 = i0 / QT;
    if (stage_v) {
      if (i0) __syncthreads();
      for (int e = threadIdx.x; e < 16 * T; e += 256) {
        const int r = e / T, tk = e - r * T;
        vs[r * (T + 1) + tk] = d0 + r < dk ? vg[(long long)(d0 + r) * T + tk] : 0.f;
      }
      __syncthreads();
    }
    const int i = i0 + threadIdx.x;
    const int d = i / QT, j = i - d * QT;
    const int tq = tq0 + j;
    if (d < dk && tq < T) {
      const float* vr = stage_v ? vs + (d - d0) * (T + 1) : vg + (long long)d * T;
      const float* pr = sc + j * T;
      float acc = 0.f;
      for (int tk = 0; tk < T; ++tk) acc = fmaf(pr[tk], vr[tk], acc);
      out[base + (long long)d * T + tq] = acc;
    }
  }
}

// modules/style_encoder.py:83-91
__global__ __launch_bounds__(64) void masked_mean_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ mask,
                                                         float* __restrict__ out, int C, int T) {
  const int c = blockIdx.x, b = blockIdx.y;
  const float* xr = x + ((long long)b * C + c) * T;
  float s = 0.f, m = 0.f;
  for (int t = threadIdx.x; t < T; t += 64) {
    s += xr[t];
    m += mask ? mask[(long long)b * T + t] : 1.0f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_down(s, o, 64);
    m += __shfl_down(m, o, 64);
  }
  if (threadIdx.x == 0) out[(long long)b * C + c] = __fdiv_rn(s, m);
}

// modules/quantize.py:444-449: LayerNorm over channels (biased variance, eps 1e-5) then
// * gamma + beta with [gamma|beta] = timbre_linear(timbre).  One workgroup per (b, 64-step tile);
// lanes along time so the (B,C,T) reads stay coalesced; the 4 waves split C.
__global__ __launch_bounds__(256) void layernorm_c_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ style,
                                                          float* __restrict__ out, int C, int T) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int t = blockIdx.x * 64 + lane;
  const bool tv = t < T;
  const float* xb = x + (long long)b * C * T + (tv ? t : T - 1);
  // Round 6: each chain (channels wave, wave + 4, ...) keeps its order of additions, but its loads go out 16 at a time -- the loop
  // was C / 4 dependent global round trips per lane (130 - 145 us for a 10 MB tensor).  Channels past C add 0.f: same bits.
  constexpr int U = 16;
  float s = 0.f;
  for (int c0 = wave; c0 < C; c0 += 4 * U) {
    float buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = c0 + 4 * u < C ? xb[(long long)(c0 + 4 * u) * T] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += tv ? buf[u] : 0.f;
  }
  red[0][wave][lane] = s;
  __syncthreads();
  const float mean = ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane])) / (float)C;
  float vs = 0.f;
  for (int c0 = wave; c0 < C; c0 += 4 * U) {
    float buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = c0 + 4 * u < C ? xb[(long long)(c0 + 4 * u) * T] : mean;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float d = tv ? buf[u] - mean : 0.f;
      vs = fmaf(d, d, vs);
    }
  }
  red[1][wave][lane] = vs;
  __syncthreads();
  const float var = ((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane])) / (float)C;
  const float rstd = __fdiv_rn(1.0f, sqrtf(var + 1e-5f));
  if (!tv) return;
  const float* gm = style + (long long)b * 2 * C;
  float* ob = out + (long long)b * C * T + t;
#pragma unroll 8
  for (int c = wave; c < C; c += 4) {
    const float nv = __fmul_rn(xb[(long long)c * T] - mean, rstd);
    ob[(long long)c * T] = __fadd_rn(__fmul_rn(nv, gm[c]), gm[C + c]);
  }
}

// T <= 8 (streaming hops): the (C, T) block of one clip is contiguous -- stage it in LDS with coalesced
// loads and run the SAME per-(wave, lane) arithmetic as layernorm_c_kernel on it (bit-identical results,
// without C/4 dependent global round trips per lane).
__global__ __launch_bounds__(256) void layernorm_c_small_t_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ style,
                                                                  float* __restrict__ out, int C, int T) {
  extern __shared__ float xs[];   // [C][T]
  __shared__ float red[2][4][8];
  __shared__ float stat[2][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* xb = x + (long long)b * C * T;
  for (int i = tid; i < C * T; i += 256) xs[i] = xb[i];
  __syncthreads();
  const bool tv = lane < T;
  float s = 0.f;
  if (tv) {
#pragma unroll 16
    for (int c = wave; c < C; c += 4) s += xs[c * T + lane];
    red[0][wave][lane] = s;
  }
  __syncthreads();
  float mean = 0.f, rstd = 0.f;
  if (tv) {
    mean = ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane])) / (float)C;
    float vs = 0.f;
#pragma unroll 16
    for (int c = wave; c < C; c += 4) {
      const float d = xs[c * T + lane] - mean;
      vs = fmaf(d, d, vs);
    }
    red[1][wave][lane] = vs;
  }
  __syncthreads();
  if (tv) {
    const float var = ((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane])) / (float)C;
    rstd = __fdiv_rn(1.0f, sqrtf(var + 1e-5f));
    if (wave == 0) {
      stat[0][lane] = mean;
      stat[1][lane] = rstd;
    }
  }
  __syncthreads();
  // normalise + affine with every lane busy (per element the same two roundings as layernorm_c_kernel)
  const float* gm = style + (long long)b * 2 * C;
  float* ob = out + (long long)b * C * T;
  for (int i = tid; i < C * T; i += 256) {
    const int c = i / T, tt = i - c * T;
    const float nv = __fmul_rn(xs[i] - stat[0][tt], stat[1][tt]);
    ob[i] = __fadd_rn(__fmul_rn(nv, gm[c]), gm[C + c]);
  }
}

// centre=True STFT framing (torch.stft pad_mode='reflect'): frames[b][n][f].
__global__ void stft_frames_kernel(const float* __restrict__ wave, float* __restrict__ frames, int T,
                                   int n_win, int n_frames, int hop, int pad, int n_off,
                                   long long n) {
  const long long per_b = (long long)n_win * n_frames;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / per_b;
    const long long r = i - b * per_b;
    const int nn = (int)(r / n_frames);
    const int f = (int)(r - (long long)nn * n_frames);
    int t = f * hop + nn + n_off - pad;
    if (t < 0) t = -t;
    if (t >= T) t = 2 * (T - 1) - t;
    frames[i] = (t >= 0 && t < T) ? wave[b * T + t] : 0.f;
  }
}

__global__ void spec_power_kernel(const float* __restrict__ spec, float* __restrict__ out, int F,
                                  int n_frames, int power, long long n) {
  const long long per_b = (long long)F * n_frames;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / per_b;
    const long long r = i - b * per_b;
    const float re = spec[b * 2 * per_b + r];
    const float im = spec[b * 2 * per_b + per_b + r];
    const float p = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
    out[i] = power == 2 ? p : sqrtf(p);
  }
}

// ---- backward of the spectral losses (d loss / d estimate) ------------------------------------------------
// d/da of scale * |a - b| (mode 0) or scale * |log10 max(a, eps) - log10 max(b, eps)| (mode 1)
__global__ void pair_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ da,
                                long long n, int mode, float eps, float scale, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float av = a[i], bv = b[i];
    float g;
    if (mode == 0) {
      g = av > bv ? 1.f : (av < bv ? -1.f : 0.f);
    } else if (mode == 2) {
      g = 2.f * (av - bv);
    } else if (mode == 3) {
      const float d = av - bv;
      g = fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
    } else {
      const float d = log10f(fmaxf(av, eps)) - log10f(fmaxf(bv, eps));
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      g = av > eps ? sgn / (av * 2.302585092994046f) : 0.f;
    }
    g *= scale;
    da[i] = accumulate ? da[i] + g : g;
  }
}

// spec (B, 2F, frames) = [re | im]; out = |z| (power 1) or |z|^2 (power 2): dspec from dout
__global__ void spec_power_bwd_kernel(const float* __restrict__ spec, const float* __restrict__ dout,
                                      float* __restrict__ dspec, int F, int n_frames, int power, long long n) {
  const long long per_b = (long long)F * n_frames;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / per_b, r = i - b * per_b;
    const float re = spec[b * 2 * per_b + r], im = spec[b * 2 * per_b + per_b + r];
    float cr, cim;
    if (power == 2) {
      cr = 2.f * re; cim = 2.f * im;
    } else {
      const float m = sqrtf(re * re + im * im);
      cr = m > 0.f ? re / m : 0.f; cim = m > 0.f ? im / m : 0.f;
    }
    dspec[b * 2 * per_b + r] = dout[i] * cr;
    dspec[b * 2 * per_b + per_b + r] = dout[i] * cim;
  }
}

// adjoint of stft_frames_kernel: every sample gathers the frame entries that read it, directly or through
// the reflection at either end (deterministic: no atomics)
__global__ void stft_frames_bwd_kernel(const float* __restrict__ dframes, float* __restrict__ dwave, int T, int n_win,
                                       int n_frames, int hop, int pad, int n_off, long long n) {
  const long long per_b = (long long)n_win * n_frames;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / T;
    const int t = (int)(i - b * T);
    const float* df = dframes + b * per_b;
    float g = 0.f;
    // padded-signal positions u = f*hop + nn + n_off - pad that resolve to sample t
    int us[3];
    int nu = 0;
    us[nu++] = t;
    if (t >= 1) us[nu++] = -t;                      // left reflection:  u < 0  -> -u
    if (t <= T - 2) us[nu++] = 2 * (T - 1) - t;     // right reflection: u >= T -> 2(T-1) - u
    for (int q = 0; q < nu; ++q) {
      const int w = us[q] + pad - n_off;            // = f*hop + nn
      int f_hi = w / hop;
      if (w < 0) continue;
      if (f_hi > n_frames - 1) f_hi = n_frames - 1;
      int f_lo = (w - n_win + 1 + hop - 1) / hop;
      if (w - n_win + 1 < 0) f_lo = 0;
      for (int f = f_lo; f <= f_hi; ++f) {
        const int nn = w - f * hop;
        if (nn >= 0 && nn < n_win) g += df[(long long)nn * n_frames + f];
      }
    }
    dwave[i] = g;
  }
}

__device__ __forceinline__ float pair_term(float a, float b, int mode, float eps) {
  if (mode == 0) return fabsf(a - b);
  if (mode == 1) return fabsf(log10f(fmaxf(a, eps)) - log10f(fmaxf(b, eps)));
  const float d = a - b;
  if (mode == 3) { const float ad = fabsf(d); return ad < 1.f ? 0.5f * d * d : ad - 0.5f; }   // smooth L1, beta = 1
  return d * d;
}

__global__ __launch_bounds__(256) void reduce_pair_stage1(const float* __restrict__ a,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ scratch, long long n,
                                                          int mode, float eps) {
  __shared__ float part[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    s += pair_term(a[i], b[i], mode, eps);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) scratch[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void reduce_pair_stage2(const float* __restrict__ scratch,
                                                          float* __restrict__ out, int nblk,
                                                          float scale, int accumulate) {
  __shared__ float part[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += scratch[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = ((part[0] + part[1]) + (part[2] + part[3])) * scale;
    out[0] = accumulate ? out[0] + tot : tot;
  }
}

// K13: anti-aliased SnakeBeta (alias_free_torch/act.py:24-29 around modules/quantize.py:77-90), fused:
//   u[m]  = 2 * sum_i xp[i] f[m + 15 - 2 i]          (replicate-pad 5, conv_transpose stride 2, crop 15/15;
//                                                     alias_free_torch/resample.py:28-37)
//   a[m]  = u[m] + sin(u[m] e^alpha)^2 / (e^beta + 1e-9)
//   y[t]  = sum_j ap[2 t + j] f[j]                    (replicate-pad 5/6, stride-2 depthwise conv; filter.py:89-96)
// One workgroup per (b, c, 256-step tile): x tile + halo in LDS, the 2x-rate activations in LDS, then the
// decimating filter.  HBM-bound: reads x once, writes y once.
__global__ __launch_bounds__(256) void aa_snakebeta_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ alpha_log,
                                                           const float* __restrict__ beta_log,
                                                           const float* __restrict__ filt,
                                                           float* __restrict__ y, int C, int T) {
  constexpr int TT = 256;
  __shared__ float xs[TT + 16];        // x[t0-8 .. t0+TT+7], index clamped (replicate padding)
  __shared__ float as[2 * TT + 16];    // a[2 t0 - 5 .. 2 t0 + 2 TT + 6]
  __shared__ float f[12];
  const int bc = blockIdx.y;
  const int c = bc % C;
  const int t0 = blockIdx.x * TT;
  const float* xr = x + (long long)bc * T;
  if (threadIdx.x < 12) f[threadIdx.x] = filt[threadIdx.x];
  for (int i = threadIdx.x; i < TT + 16; i += 256) {
    int t = t0 - 8 + i;
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    xs[i] = xr[t];
  }
  __syncthreads();
  const float ea = expf(alpha_log[c]);
  const float inv = __fdiv_rn(1.0f, __fadd_rn(expf(beta_log[c]), 1e-9f));
  for (int i = threadIdx.x; i < 2 * TT + 12; i += 256) {
    int m = 2 * t0 - 5 + i;                       // position in the 2x-rate signal, clamped like F.pad(replicate)
    m = m < 0 ? 0 : (m > 2 * T - 1 ? 2 * T - 1 : m);
    // u[m] = 2 * sum over padded index ip with 0 <= m + 15 - 2 ip <= 11; xp[ip] = x[clamp(ip - 5)]
    float u = 0.f;
    const int ip_lo = (m + 4 + 1) >> 1;           // ceil((m + 4) / 2)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int ip = ip_lo + q;
      const int k = m + 15 - 2 * ip;
      if (k >= 0 && k <= 11) {
        int t = ip - 5;
        t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
        u = fmaf(xs[t - (t0 - 8)], f[k], u);
      }
    }
    u = __fmul_rn(2.0f, u);
    const float sn = sin_sq(__fmul_rn(u, ea));
    as[i] = __fadd_rn(u, __fmul_rn(inv, sn));
  }
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t < T) {
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) o = fmaf(as[2 * threadIdx.x + j], f[j], o);
    y[(long long)bc * T + t] = o;
  }
}

// losses.py:84  l2 = mean_{b,t} sqrt( mean_m ( log(a+eps) - log(b+eps) )^2 );  a, b (B, M, T).
// One thread per (b, t) column (coalesced along t), partial sums per block into scratch.
__global__ __launch_bounds__(256) void logdiff_rms_stage1(const float* __restrict__ a,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ scratch, int M, int T,
                                                          long long n_cols, float eps) {
  __shared__ float part[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_cols; i += (long long)gridDim.x * 256) {
    const long long bb = i / T;
    const int t = (int)(i - bb * T);
    const float* pa = a + bb * M * T + t;
    const float* pb = b + bb * M * T + t;
    float q = 0.f;
    for (int m = 0; m < M; ++m) {
      const float d = logf(fabsf(pa[(long long)m * T]) + eps) - logf(fabsf(pb[(long long)m * T]) + eps);
      q = fmaf(d, d, q);
    }
    s += sqrtf(q / (float)M);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) scratch[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// Backward of logdiff_rms w.r.t. b: db[m] (+)= scale * d/db_m sqrt(mean_m d^2), d_m = log(|a_m|+eps) - log(|b_m|+eps):
// -(d_m / (M r)) * sign(b_m) / (|b_m| + eps), r = sqrt(mean d^2) (0 where r == 0).  One thread per (b, t) column.
__global__ __launch_bounds__(256) void logdiff_rms_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ db, int M, int T, long long n_cols, float eps,
                                                              float scale, int accumulate) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_cols; i += (long long)gridDim.x * 256) {
    const long long bb = i / T;
    const int t = (int)(i - bb * T);
    const float* pa = a + bb * M * T + t;
    const float* pb = b + bb * M * T + t;
    float* pd = db + bb * M * T + t;
    float q = 0.f;
    for (int m = 0; m < M; ++m) {
      const float d = logf(fabsf(pa[(long long)m * T]) + eps) - logf(fabsf(pb[(long long)m * T]) + eps);
      q = fmaf(d, d, q);
    }
    const float r = sqrtf(q / (float)M);
    const float k = r > 0.f ? scale / ((float)M * r) : 0.f;
    for (int m = 0; m < M; ++m) {
      const float bv = pb[(long long)m * T];
      const float d = logf(fabsf(pa[(long long)m * T]) + eps) - logf(fabsf(bv) + eps);
      const float sg = bv > 0.f ? 1.f : (bv < 0.f ? -1.f : 0.f);
      const float g = -k * d * sg / (fabsf(bv) + eps);
      pd[(long long)m * T] = accumulate ? pd[(long long)m * T] + g : g;
    }
  }
}

}  // namespace fac

using namespace fac;

extern "C" int fac_version(void) { return 3; }   // round 3: fac_conv_desc.row_phases, fac_adamw_step_masked, fac_pack_convtr_w_rows
extern "C" const char* fac_last_error(void) { return fac::g_err; }

#define EW_LAUNCH(kern, n, ...)                                                         \
  hipLaunchKernelGGL(kern, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int fac_snake_fwd(const float* x, const float* alpha, float* y, int B, int C, int T,
                             fac_stream_t stream) {
  FAC_REQUIRE(x && alpha && y && B > 0 && C > 0 && T > 0, "snake_fwd: bad arguments");
  const long long n = (long long)B * C * T;
  if (T >= 1024 && (long long)B * C <= 65535) {
    hipLaunchKernelGGL(snake_rows_kernel, dim3((T + 1023) / 1024, B * C), dim3(256), 0, (hipStream_t)stream, x, alpha, y, C, T);
  } else {
    EW_LAUNCH(snake_kernel, n, x, alpha, y, C, T, n);
  }
  return check_launch("snake_fwd");
}

extern "C" int fac_gate_tanh_sigmoid(const float* a, const float* g, int64_t g_bs, float* out, int B,
                                     int C, int T, fac_stream_t stream) {
  FAC_REQUIRE(a && out && B > 0 && C > 0 && T > 0, "gate_tanh_sigmoid: bad arguments");
  const long long n = (long long)B * C * T;
  EW_LAUNCH(gate_kernel, n, a, g, out, C, T, (long long)g_bs, n);
  return check_launch("gate_tanh_sigmoid");
}

extern "C" int fac_embed_sum(const int64_t* codes, const float* tables, float* out, int B, int n_tab,
                             int n_codes, int code_row0, int V, int E, int T, int accumulate,
                             fac_stream_t stream) {
  FAC_REQUIRE(codes && tables && out && B > 0 && n_tab >= 0 && V > 0 && E > 0 && T > 0 &&
                  code_row0 >= 0 && code_row0 + n_tab <= n_codes, "embed_sum: bad arguments");
  const long long n = (long long)B * E * T;
  EW_LAUNCH(embed_sum_kernel, n, (const long long*)codes, tables, out, n_tab, n_codes, code_row0, V, E, T, n,
            accumulate);
  return check_launch("embed_sum");
}

extern "C" int fac_glu_residual(const float* a, const float* res, float* out, int B, int C, int T,
                                fac_stream_t stream) {
  FAC_REQUIRE(a && res && out && B > 0 && C > 0 && T > 0, "glu_residual: bad arguments");
  const long long n = (long long)B * C * T;
  EW_LAUNCH(glu_res_kernel, n, a, res, out, C, T, n);
  return check_launch("glu_residual");
}

extern "C" int fac_add(const float* a, const float* b, float* out, int64_t n, fac_stream_t stream) {
  FAC_REQUIRE(a && b && out && n > 0, "add: bad arguments");
  EW_LAUNCH(add_kernel, n, a, b, out, (long long)n);
  return check_launch("add");
}

extern "C" int fac_sub2(const float* a, const float* b, const float* c, float* out, int64_t n,
                        fac_stream_t stream) {
  FAC_REQUIRE(a && b && c && out && n > 0, "sub2: bad arguments");
  EW_LAUNCH(sub2_kernel, n, a, b, c, out, (long long)n);
  return check_launch("sub2");
}

extern "C" int fac_mul_mask(float* x, const float* mask, int B, int C, int T, fac_stream_t stream) {
  FAC_REQUIRE(x && mask && B > 0 && C > 0 && T > 0, "mul_mask: bad arguments");
  const long long n = (long long)B * C * T;
  EW_LAUNCH(mul_mask_kernel, n, x, mask, C, T, n);
  return check_launch("mul_mask");
}

extern "C" int fac_stream_push(float* buf, const float* src, int64_t rows, int64_t cap, int hist, int n_prev,
                               int n_new, fac_stream_t stream) {
  FAC_REQUIRE(buf && src && rows > 0 && hist >= 0 && hist <= 2048 && n_prev >= 0 && n_new > 0 &&
                  cap >= (int64_t)hist + n_new && cap >= (int64_t)hist + n_prev,
              "stream_push: bad arguments (rows=%lld cap=%lld hist=%d n_prev=%d n_new=%d)", (long long)rows,
              (long long)cap, hist, n_prev, n_new);
  hipLaunchKernelGGL(stream_push_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, buf, src,
                     (long long)cap, hist, n_prev, n_new);
  return check_launch("stream_push");
}

extern "C" int fac_wn_res_skip(const float* rs, float* x, float* out, int B, int C, int T, int last,
                               fac_stream_t stream) {
  FAC_REQUIRE(rs && out && (last || x) && B > 0 && C > 0 && T > 0, "wn_res_skip: bad arguments");
  const long long n = (long long)B * C * T;
  EW_LAUNCH(wn_res_skip_kernel, n, rs, x, out, C, T, last, n);
  return check_launch("wn_res_skip");
}

extern "C" int fac_attention(const float* q, const float* k, const float* v, const float* mask,
                             float* out, int B, int n_heads, int dk, int T, fac_stream_t stream) {
  FAC_REQUIRE(q && k && v && out && B > 0 && n_heads > 0 && dk > 0 && T > 0, "attention: bad arguments");
  size_t lds = ((size_t)dk * 16 + (size_t)16 * T) * sizeof(float);
  FAC_REQUIRE(lds <= 160 * 1024, "attention: T=%d too long for the LDS score tile", T);
  const size_t lds_v = lds + (size_t)16 * (T + 1) * sizeof(float);       // + 16 rows of V per pass
  const int stage_v = lds_v <= 160 * 1024 ? 1 : 0;
  if (stage_v) lds = lds_v;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid((T + 15) / 16, n_heads, B);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), lds, (hipStream_t)stream, q, k, v, mask, out,
                     n_heads, dk, T, stage_v);
  return check_launch("attention");
}

extern "C" int fac_masked_mean(const float* x, const float* mask, float* out, int B, int C, int T,
                               fac_stream_t stream) {
  FAC_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "masked_mean: bad arguments");
  hipLaunchKernelGGL(masked_mean_kernel, dim3(C, B), dim3(64), 0, (hipStream_t)stream, x, mask, out, C, T);
  return check_launch("masked_mean");
}

extern "C" int fac_layernorm_c_affine(const float* x, const float* style, float* out, int B, int C,
                                      int T, fac_stream_t stream) {
  FAC_REQUIRE(x && style && out && B > 0 && C > 0 && T > 0, "layernorm_c_affine: bad arguments");
  if (T <= 8 && (size_t)C * T * 4 <= 64 * 1024) {
    hipLaunchKernelGGL(layernorm_c_small_t_kernel, dim3(B), dim3(256), (size_t)C * T * 4, (hipStream_t)stream, x,
                       style, out, C, T);
    return check_launch("layernorm_c_affine(small T)");
  }
  hipLaunchKernelGGL(layernorm_c_kernel, dim3((T + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x,
                     style, out, C, T);
  return check_launch("layernorm_c_affine");
}

extern "C" int fac_stft_frames(const float* wave, float* frames, int B, int T, int n_win,
                               int n_frames, int hop, int pad, int n_off, fac_stream_t stream) {
  FAC_REQUIRE(wave && frames && B > 0 && T > 0 && n_win > 0 && n_frames > 0 && hop > 0,
              "stft_frames: bad arguments");
  FAC_REQUIRE(pad < T, "stft_frames: reflect pad %d needs a signal longer than %d samples", pad, T);
  const long long n = (long long)B * n_win * n_frames;
  EW_LAUNCH(stft_frames_kernel, n, wave, frames, T, n_win, n_frames, hop, pad, n_off, n);
  return check_launch("stft_frames");
}

extern "C" int fac_spec_power(const float* spec, float* out, int B, int F, int n_frames, int power,
                              fac_stream_t stream) {
  FAC_REQUIRE(spec && out && B > 0 && F > 0 && n_frames > 0 && (power == 1 || power == 2),
              "spec_power: bad arguments");
  const long long n = (long long)B * F * n_frames;
  EW_LAUNCH(spec_power_kernel, n, spec, out, F, n_frames, power, n);
  return check_launch("spec_power");
}

extern "C" int fac_reduce_pair(const float* a, const float* b, float* out, float* scratch,
                               int64_t n, int mode, float eps, float scale, int accumulate,
                               fac_stream_t stream) {
  FAC_REQUIRE(a && b && out && scratch && n > 0 && mode >= 0 && mode <= 3, "reduce_pair: bad arguments");
  long long g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(reduce_pair_stage1, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a, b,
                     scratch, (long long)n, mode, eps);
  hipLaunchKernelGGL(reduce_pair_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, out,
                     (int)g, scale, accumulate);
  return check_launch("reduce_pair");
}

extern "C" int fac_logdiff_rms(const float* a, const float* b, float* out, float* scratch, int B, int M,
                               int T, float eps, float scale, int accumulate, fac_stream_t stream) {
  FAC_REQUIRE(a && b && out && scratch && B > 0 && M > 0 && T > 0, "logdiff_rms: bad arguments");
  const long long n_cols = (long long)B * T;
  long long g = (n_cols + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(logdiff_rms_stage1, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a, b, scratch, M, T,
                     n_cols, eps);
  hipLaunchKernelGGL(reduce_pair_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, out, (int)g, scale,
                     accumulate);
  return check_launch("logdiff_rms");
}

extern "C" int fac_logdiff_rms_bwd(const float* a, const float* b, float* db, int B, int M, int T, float eps, float scale,
                                   int accumulate, fac_stream_t stream) {
  FAC_REQUIRE(a && b && db && B > 0 && M > 0 && T > 0, "logdiff_rms_bwd: bad arguments");
  const long long n_cols = (long long)B * T;
  long long g = (n_cols + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(logdiff_rms_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a, b, db, M, T, n_cols, eps, scale,
                     accumulate);
  return check_launch("logdiff_rms_bwd");
}

extern "C" int fac_aa_snakebeta_fwd(const float* x, const float* alpha_log, const float* beta_log,
                                    const float* filter12, float* y, int B, int C, int T,
                                    fac_stream_t stream) {
  FAC_REQUIRE(x && alpha_log && beta_log && filter12 && y && B > 0 && C > 0 && T > 0,
              "aa_snakebeta_fwd: bad arguments");
  FAC_REQUIRE((long long)B * C <= 65535, "aa_snakebeta_fwd: B*C too large");
  hipLaunchKernelGGL(aa_snakebeta_kernel, dim3((T + 255) / 256, B * C), dim3(256), 0, (hipStream_t)stream, x,
                     alpha_log, beta_log, filter12, y, C, T);
  return check_launch("aa_snakebeta_fwd");
}

extern "C" int fac_pair_bwd(const float* a, const float* b, float* da, int64_t n, int mode, float eps, float scale,
                            int accumulate, fac_stream_t stream) {
  FAC_REQUIRE(a && b && da && n > 0 && mode >= 0 && mode <= 3, "pair_bwd: bad arguments");
  EW_LAUNCH(pair_bwd_kernel, n, a, b, da, (long long)n, mode, eps, scale, accumulate);
  return check_launch("pair_bwd");
}

extern "C" int fac_spec_power_bwd(const float* spec, const float* dout, float* dspec, int B, int F, int n_frames, int power,
                                  fac_stream_t stream) {
  FAC_REQUIRE(spec && dout && dspec && B > 0 && F > 0 && n_frames > 0 && (power == 1 || power == 2), "spec_power_bwd: bad arguments");
  const long long n = (long long)B * F * n_frames;
  EW_LAUNCH(spec_power_bwd_kernel, n, spec, dout, dspec, F, n_frames, power, n);
  return check_launch("spec_power_bwd");
}

extern "C" int fac_stft_frames_bwd(const float* dframes, float* dwave, int B, int T, int n_win, int n_frames, int hop, int pad,
                                   int n_off, fac_stream_t stream) {
  FAC_REQUIRE(dframes && dwave && B > 0 && T > 1 && n_win > 0 && n_frames > 0 && hop > 0 && pad < T, "stft_frames_bwd: bad arguments");
  const long long n = (long long)B * T;
  EW_LAUNCH(stft_frames_bwd_kernel, n, dframes, dwave, T, n_win, n_frames, hop, pad, n_off, n);
  return check_launch("stft_frames_bwd");
}


// ---------------------------------------------------------------------------------------- fp32 (B, C, T) -> P8 planes
// One thread = one (clip, 8-channel group, time step): 8 coalesced fp32 loads (rows T apart), optional Snake, the exact
// round-to-nearest three-way bf16 split, three 16-byte stores (one per plane).  HBM-bound: 4 B read + 6 B written per element.
namespace fac {
typedef __bf16 p8_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void to_p8_kernel(const float* __restrict__ x, const float* __restrict__ alpha, p8_bf16x8* __restrict__ out,
                                                    int C8, int T, long long n, long long plane_units) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long long bg = i / T;                 // b * C8 + g
    const int g = (int)(bg % C8);
    const float* row = x + (bg * 8) * T + t;
    p8_bf16x8 h, m, l;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = row[(long long)k * T];
      if (alpha != nullptr) {
        const float al = alpha[g * 8 + k];
        v = snake_apply(v, al, snake_inv(al));
      }
      const __bf16 a = (__bf16)v;
      const float r1 = v - (float)a;
      const __bf16 b2 = (__bf16)r1;
      const __bf16 c = (__bf16)(r1 - (float)b2);
      h[k] = a; m[k] = b2; l[k] = c;
    }
    out[i] = h;
    out[plane_units + i] = m;
    out[2 * plane_units + i] = l;
  }
  if (blockIdx.x == 0 && threadIdx.x < 3) {     // the zero unit that closes every plane (what consumers read for padding columns)
    p8_bf16x8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (__bf16)0.f;
    out[(long long)threadIdx.x * plane_units + n] = z;
  }
}
}  // namespace fac

extern "C" int fac_to_p8(const float* x, const float* alpha, void* out, int B, int C, int T, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(x && out && B > 0 && C > 0 && C % 8 == 0 && T > 0, "to_p8: bad arguments (C must be a multiple of 8)");
  const long long n = (long long)B * (C / 8) * T;
  const int blocks = (int)((n + 255) / 256 < 262144 ? (n + 255) / 256 : 262144);
  hipLaunchKernelGGL(to_p8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, alpha, reinterpret_cast<p8_bf16x8*>(out), C / 8, T, n, n + 1);
  return check_launch("to_p8");
}
