// Backward pieces of the conv stack (first kernels of the training step, BASELINE.json configs[2]).
//
//   bwd_data  of a stride-1 conv   = the forward kernel on the tap-flipped, channel-transposed weights
//                                    (fac_pack_conv_w_bwd) over dy, giving the gradient of the PADDED input,
//                                    followed by fac_pad_fold_bwd (reflection sends a padded position's gradient
//                                    back to the sample it mirrors: dac/model/encodec.py:96-113 pad1d);
//             of a strided conv    = the polyphase transposed-conv launch on the forward weights (same (C_out,
//                                    C_in, K) tensor = conv_transpose's (in, out, K)), then the same fold;
//             of a transposed conv = the strided forward conv on its own weights;
//   bwd_weight                     = conv1d_wgrad_kernel below: dW[co][ci][k] = sum_{b,t} dy[b][co][t] *
//                                    xpad[b][ci][t*s + k*d]  as an fp32-MFMA GEMM whose contraction runs over
//                                    time; (b, t) ranges are split across workgroups and the partial dW are added
//                                    in split order by a second kernel (deterministic);
//   weight-norm, Snake and bias backward are small reductions.
// Reference semantics: torch autograd through dac/model/encodec.py SConv1d / dac/nn/layers.py snake (checked
// against autograd of the CPU oracle in tests/test_gpu_parity.py).
#include "conv1d_mfma.h"
#include "prep_batch.h"
#include <stdlib.h>

namespace fac {

// packed[(co*K + k')*CP + ci] = v[co][ci][K-1-k'] * scale[co]   (rows co < C_out; the buffer is zero-filled by
// the caller up to fac_cin_pad(C_out) rows and CP = pad32(C_in) columns)
__device__ __forceinline__ void pack_conv_bwd_body(const float* __restrict__ v, const float* __restrict__ scale,
                                                   float* __restrict__ packed, int C_out, int C_in, int K, int CP, long long n,
                                                   int vb, int vg) {
  // n covers the whole padded buffer (cin_pad(C_out) rows x K x CP columns): padding rows / columns are written as zeros
  for (long long i = (long long)vb * 256 + threadIdx.x; i < n; i += (long long)vg * 256) {
    const int ci = (int)(i % CP);
    const long long r = i / CP;
    const int kp = (int)(r % K);
    const int co = (int)(r / K);
    float w = 0.f;
    if (co < C_out && ci < C_in) {
      w = v[((long long)co * C_in + ci) * K + (K - 1 - kp)];
      if (scale) w = __fmul_rn(w, scale[co]);
    }
    packed[i] = w;
  }
}

__global__ __launch_bounds__(256) void pack_conv_bwd_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                            float* __restrict__ packed, int C_out, int C_in, int K, int CP, long long n) {
  pack_conv_bwd_body(v, scale, packed, C_out, C_in, K, CP, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void bwd_batch_kernel(const PrepJob* __restrict__ jobs, const int* __restrict__ first, int njobs) {
  const int j = prep_find_job(first, njobs, blockIdx.x);
  const PrepJob& J = jobs[j];
  pack_conv_bwd_body(static_cast<const float*>(J.a), static_cast<const float*>(J.b), static_cast<float*>(J.out), J.i[0], J.i[1], J.i[2],
                     J.i[3], J.n, blockIdx.x - first[j], J.nblocks);
}

int prep_launch_bwd(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s) {
  hipLaunchKernelGGL(bwd_batch_kernel, dim3(total), dim3(256), 0, s, jobs, first, njobs);
  return check_launch("bwd_batch");
}

// dx[b][c][j] = dxpad[pad_left + j] (+ the gradients of the padded positions that mirror sample j)
__global__ void pad_fold_bwd_kernel(const float* __restrict__ dxpad, float* __restrict__ dx, int T, int Tp,
                                    int pad_left, int mode, long long n) {
  const int pad_right = Tp - pad_left - T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % T);
    const long long row = i / T;
    const float* p = dxpad + row * Tp;
    float g = p[pad_left + j];
    if (mode == FAC_PAD_REFLECT) {
      if (j >= 1 && j <= pad_left) g += p[pad_left - j];                       // xpad[pad_left - j] = x[j]
      const int m = T - 1 - j;                                                 // xpad[pad_left + T-1 + m] = x[T-1-m]
      if (m >= 1 && m <= pad_right) g += p[pad_left + T - 1 + m];
    }
    dx[i] = g;
  }
}

// The same fold IN PLACE, touching only the edges: afterwards dxpad[row][pad_left + j] is dx[row][j], i.e. the gradient is the
// window [pad_left, pad_left + T) of every padded row and a consumer that takes a row stride reads it where it lies (round 6:
// fac_pad_fold_bwd re-read and re-wrote the whole (B, C, T) tensor -- 8 bytes per element -- to add at most 54 mirrored samples per
// row).  One thread per (row, mirrored position); reflect padding only (zero padding has nothing to add).
__global__ void pad_fold_edges_kernel(float* __restrict__ dxpad, int T, int Tp, int pad_left, long long rows) {
  const int pad_right = Tp - pad_left - T;
  const int per_row = pad_left + pad_right;
  const long long n = rows * per_row;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / per_row;
    const int e = (int)(i - row * per_row);
    float* p = dxpad + row * Tp;
    if (e < pad_left) {
      const int j = e + 1;                                   // xpad[pad_left - j] mirrors x[j], j = 1 .. pad_left
      p[pad_left + j] += p[pad_left - j];
    } else {
      const int m = e - pad_left + 1;                        // xpad[pad_left + T - 1 + m] mirrors x[T - 1 - m], m = 1 .. pad_right
      p[pad_left + T - 1 - m] += p[pad_left + T - 1 + m];
    }
  }
}

struct WgArgs {
  const float* x;      // (B, C_in, T_in)
  const float* dy;     // (B, C_out, T_out)
  float* part;         // [S][C_out][C_in][K]
  long long x_bs, x_cs, dy_bs, dy_cs;
  int B, C_in, T_in, T_ext, C_out, T_out, K, stride, dil, pad_left, pad_mode;
  int K2, dil2;        // two-level taps as K2 virtual channels per real one (C_in, K here are the virtual count and K1)
  int CIT;             // input channels per column tile (CIT*K <= 128)
  int XWl;             // staged input columns per time tile
  int n_tt;            // 64-step time tiles per clip
  int tiles_per_split;
};

constexpr int WG_TT = 64;

__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(WgArgs a) {
  extern __shared__ float sm[];
  float* dyl = sm;                          // [64][WG_TT + 1]
  float* xl = sm + 64 * (WG_TT + 1);        // [CIT][XWl]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kk = lane >> 5;
  const int co0 = blockIdx.x * 64;
  const int ci0 = blockIdx.y * a.CIT;
  const int z = blockIdx.z;
  const int cb = wave & 1, jb0 = (wave >> 1) * 2;
  const int ncol = min(a.CIT, a.C_in - ci0) * a.K;

  // this lane's B-operand columns: J = jb*32 + l31 -> (local channel, tap)
  int xoff[2];
  bool jok[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int J = (jb0 + q) * 32 + l31;
    jok[q] = J < ncol;
    const int cl = jok[q] ? J / a.K : 0, k = jok[q] ? J - cl * a.K : 0;
    xoff[q] = cl * a.XWl + k * a.dil;
  }
  f32x16 acc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int tile_lo = z * a.tiles_per_split;
  const int tile_hi = min(a.B * a.n_tt, tile_lo + a.tiles_per_split);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int b = tile / a.n_tt, t0 = (tile - b * a.n_tt) * WG_TT;
    const float* dyb = a.dy + (long long)b * a.dy_bs;
    const float* xb = a.x + (long long)b * a.x_bs;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r * 4 + wave, t = t0 + lane;
      const int co = co0 + row;
      dyl[row * (WG_TT + 1) + lane] = (co < a.C_out && t < a.T_out) ? dyb[(long long)co * a.dy_cs + t] : 0.f;
    }
    const int tin0 = t0 * a.stride - a.pad_left;
    for (int idx = tid; idx < a.CIT * a.XWl; idx += 256) {
      const int cl = idx / a.XWl, c = idx - cl * a.XWl;
      const int v = ci0 + cl;
      const int ci = v / a.K2, k2 = v - ci * a.K2;
      int tin = tin0 + c + k2 * a.dil2;
      if (a.pad_mode == FAC_PAD_REFLECT) tin = reflect_index(tin, a.T_in, a.T_ext);
      xl[idx] = (v < a.C_in && tin >= 0 && tin < a.T_in) ? xb[(long long)ci * a.x_cs + tin] : 0.f;
    }
    __syncthreads();
    const float* ap = dyl + (cb * 32 + l31) * (WG_TT + 1) + kk;
#pragma unroll 8
    for (int tt = 0; tt < WG_TT; tt += 2) {
      const float av = ap[tt];
      const int xt = (tt + kk) * a.stride;
      const float b0 = jok[0] ? xl[xoff[0] + xt] : 0.f;
      const float b1 = jok[1] ? xl[xoff[1] + xt] : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
    }
    __syncthreads();
  }
  // partial dW of this split: row = output channel, columns (ci, k) contiguous
  float* pz = a.part + (long long)z * a.C_out * a.C_in * a.K;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int J = (jb0 + q) * 32 + l31;
    if (J >= ncol) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (co < a.C_out) pz[((long long)co * a.C_in + ci0) * a.K + J] = acc[q][r];
    }
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = part[i];
    for (int z = 1; z < S; ++z) s += part[(long long)z * n + i];
    dw[i] = s;
  }
}

// w = g v/||v|| per row: dg = <dW, v>/||v||;  dv = g/||v|| (dW - v <dW, v>/||v||^2)   (one workgroup per row)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                              const float* __restrict__ dw, float* __restrict__ dv,
                                                              float* __restrict__ dg, int m) {
  __shared__ float red[2][256];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* vr = v + (long long)row * m;
  const float* dr = dw + (long long)row * m;
  float s_vv = 0.f, s_dv = 0.f;
  for (int i = tid; i < m; i += 256) {
    s_vv = fmaf(vr[i], vr[i], s_vv);
    s_dv = fmaf(dr[i], vr[i], s_dv);
  }
  red[0][tid] = s_vv;
  red[1][tid] = s_dv;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
    __syncthreads();
  }
  const float nrm = sqrtf(red[0][0]), dot = red[1][0];
  const float gn = g[row] / nrm;
  if (tid == 0) dg[row] = dot / nrm;
  const float c = dot / red[0][0];
  for (int i = tid; i < m; i += 256) dv[(long long)row * m + i] = gn * (dr[i] - vr[i] * c);
}

// y = x + sin^2(a x)/(a + 1e-9):  dy/dx = 1 + a sin(2 a x)/(a + 1e-9);  dy/da = (x sin(2 a x)(a+eps) - sin^2(a x))/(a+eps)^2
// grid (C, NS): workgroup (c, s) covers a slice of the (b, t) positions of channel c and leaves one partial sum of
// d alpha; channel_reduce_kernel adds the NS partials in order (deterministic).  Same scheme for the bias gradient.
constexpr int RED_NS = 32;

__global__ __launch_bounds__(256) void snake_bwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                        const float* __restrict__ dy, float* __restrict__ dx,
                                                        float* __restrict__ part, int B, int C, int T) {
  __shared__ float red[256];
  const int c = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const float al = alpha[c], ae = al + 1e-9f;
  const long long n = (long long)B * T;
  const int ns = gridDim.y;                     // slices in use (<= RED_NS), see fac_bias_grad
  const long long per = (n + ns - 1) / ns;
  const long long lo = sl * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f;
  if (lo < hi) {                                    // clip by clip: no per-element index division
    const int b_lo = (int)(lo / T), b_hi = (int)((hi - 1) / T);
    for (int b = b_lo; b <= b_hi; ++b) {
      const long long base = (long long)b * T;
      const int t0 = (int)(lo > base ? lo - base : 0);
      const int t1 = (int)(hi < base + T ? hi - base : T);
      const long long ro = ((long long)b * C + c) * T;
      // the slice's elements keep the order p = lo + tid, lo + tid + 256, ... of the flat walk (same partial sums)
      const int first = (int)((256 - ((base + t0 - lo) % 256) + tid) % 256);
      for (int t = t0 + first; t < t1; t += 256) {
        const long long o = ro + t;
        const float xv = x[o], g = dy[o];
        const float ax = al * xv;
        const float sn = sinf(ax), cs = cosf(ax);
        const float s2 = 2.f * sn * cs;
        dx[o] = g * (1.f + al * s2 / ae);
        s += g * (xv * s2 * ae - sn * sn) / (ae * ae);
      }
    }
  }
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) part[c * RED_NS + sl] = red[0];
  if (sl == 0 && tid > 0 && tid < RED_NS && tid >= ns) part[c * RED_NS + tid] = 0.f;
}

// The same backward with the neighbours fused in: dx = add + dy * dsnake/dx (the other gradient of a tensor with two consumers --
// ResidualUnit skip + Snake -- arrives as `add`, no separate fan-in add), and the bias gradient of the conv that PRODUCED x
// (db[c] = sum over (b, t) of dx) as a second partial sum (part2), so no extra pass over dx.
__global__ __launch_bounds__(256) void snake_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                              const float* __restrict__ dy, const float* __restrict__ add,
                                                              float* __restrict__ dx, float* __restrict__ part,
                                                              float* __restrict__ part2, int B, int C, int T, long long dy_rs) {
  __shared__ float red[2][256];
  const int c = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const float al = alpha[c], ae = al + 1e-9f;
  const long long n = (long long)B * T;
  const int ns = gridDim.y;                     // slices in use (<= RED_NS), see fac_bias_grad
  const long long per = (n + ns - 1) / ns;
  const long long lo = sl * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f, sb = 0.f;
  if (lo < hi) {
    const int b_lo = (int)(lo / T), b_hi = (int)((hi - 1) / T);
    for (int b = b_lo; b <= b_hi; ++b) {
      const long long base = (long long)b * T;
      const int t0 = (int)(lo > base ? lo - base : 0);
      const int t1 = (int)(hi < base + T ? hi - base : T);
      const long long ro = ((long long)b * C + c) * T;
      const float* dyr = dy + ((long long)b * C + c) * dy_rs;      // dy rows may sit in a wider buffer (fac_snake_bwd_fused_rs)
      const int first = (int)((256 - ((base + t0 - lo) % 256) + tid) % 256);
      for (int t = t0 + first; t < t1; t += 256) {
        const long long o = ro + t;
        const float xv = x[o], g = dyr[t];
        // (libm on purpose: a hand-rolled Cody-Waite sin^2 / sin 2x pair measured SLOWER here, 378 vs 285 us at (16, 192, 24000))
        const float ax = al * xv;
        const float sn = sinf(ax), cs = cosf(ax);
        const float s2 = 2.f * sn * cs;
        float d = g * (1.f + al * s2 / ae);
        if (add) d += add[o];
        dx[o] = d;
        s += g * (xv * s2 * ae - sn * sn) / (ae * ae);
        sb += d;
      }
    }
  }
  red[0][tid] = s;
  red[1][tid] = sb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    part[c * RED_NS + sl] = red[0][0];
    if (part2) part2[c * RED_NS + sl] = red[1][0];
  }
  if (sl == 0 && tid > 0 && tid < RED_NS && tid >= ns) {      // unused slices of this channel's rows
    part[c * RED_NS + tid] = 0.f;
    if (part2) part2[c * RED_NS + tid] = 0.f;
  }
}

// The same kernel on 16-byte accesses (round 6): T % 4 == 0, x / add / dx rows 16-byte aligned, dy rows 8-byte aligned (a window of
// padded rows starts an even number of samples into its row).  A lane owns four consecutive steps: four independent sin / cos
// evaluations and 40 bytes of loads in flight per trip instead of one and 12 -- the scalar kernel ran at 4.1 TB/s against the
// 6 - 7 of the LeakyReLU kernels on the same tensors (profiles/r06_hbm_kernels.json).  Slices are cut in quads; partial sums are
// per-lane chains over its quads, then the same fixed-order tree.
__global__ __launch_bounds__(256) void snake_bwd_fused_v4_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                                 const float* __restrict__ dy, const float* __restrict__ add,
                                                                 float* __restrict__ dx, float* __restrict__ part,
                                                                 float* __restrict__ part2, int B, int C, int T, long long dy_rs) {
  __shared__ float red[2][256];
  const int c = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const float al = alpha[c], ae = al + 1e-9f;
  const int T4 = T >> 2;
  const long long n = (long long)B * T4;                       // quads of this channel
  const int ns = gridDim.y;                     // slices in use (<= RED_NS), see fac_bias_grad
  const long long per = (n + ns - 1) / ns;
  const long long lo = sl * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f, sb = 0.f;
  if (lo < hi) {
    const int b_lo = (int)(lo / T4), b_hi = (int)((hi - 1) / T4);
    for (int b = b_lo; b <= b_hi; ++b) {
      const long long base = (long long)b * T4;
      const int q0 = (int)(lo > base ? lo - base : 0);
      const int q1 = (int)(hi < base + T4 ? hi - base : T4);
      const long long ro = ((long long)b * C + c) * T;
      const float* dyr = dy + ((long long)b * C + c) * dy_rs;
      for (int q = q0 + tid; q < q1; q += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(x + ro + 4 * q);
        const float2 g0 = *reinterpret_cast<const float2*>(dyr + 4 * q), g1 = *reinterpret_cast<const float2*>(dyr + 4 * q + 2);
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
        if (add) av = *reinterpret_cast<const float4*>(add + ro + 4 * q);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {g0.x, g0.y, g1.x, g1.y}, as[4] = {av.x, av.y, av.z, av.w};
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float ax = al * xs[j];
          const float sn = sinf(ax), cs = cosf(ax);
          const float s2 = 2.f * sn * cs;
          d[j] = gs[j] * (1.f + al * s2 / ae) + as[j];
          s += gs[j] * (xs[j] * s2 * ae - sn * sn) / (ae * ae);
          sb += d[j];
        }
        *reinterpret_cast<float4*>(dx + ro + 4 * q) = make_float4(d[0], d[1], d[2], d[3]);
      }
    }
  }
  red[0][tid] = s;
  red[1][tid] = sb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    part[c * RED_NS + sl] = red[0][0];
    if (part2) part2[c * RED_NS + sl] = red[1][0];
  }
  if (sl == 0 && tid > 0 && tid < RED_NS && tid >= ns) {      // unused slices of this channel's rows
    part[c * RED_NS + tid] = 0.f;
    if (part2) part2[c * RED_NS + tid] = 0.f;
  }
}

// db[c] = sum over (b, t) of dy: workgroup (c, slice) sums its contiguous share of the flattened (b, t) range -- walked clip by
// clip (no per-element division), 16 bytes per lane where the rows allow it, four independent partial sums per thread -- and a
// second pass adds the RED_NS slices in fixed order.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ part, int B, int C, int T) {
  __shared__ float red[256];
  const int c = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const int ns = gridDim.y;                     // slices in use (<= RED_NS): small tensors get few, the rest of the row is zero-filled
  const long long n = (long long)B * T;
  const long long per = (n + ns - 1) / ns;
  const long long lo = sl * per, hi = lo + per < n ? lo + per : n;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (lo < hi) {
    const int b_lo = (int)(lo / T), b_hi = (int)((hi - 1) / T);
    const bool vec = (T & 3) == 0 && (reinterpret_cast<unsigned long long>(dy) & 15) == 0;
    for (int b = b_lo; b <= b_hi; ++b) {
      const long long base = (long long)b * T;
      int t0 = (int)(lo > base ? lo - base : 0);
      const int t1 = (int)(hi < base + T ? hi - base : T);
      const float* row = dy + ((long long)b * C + c) * T;
      if (vec) {
        const int a0 = (t0 + 3) & ~3, a1 = t1 & ~3;              // aligned interior [a0, a1), scalar edges
        if (a0 < a1) {
          for (int t = t0 + tid; t < a0; t += 256) s0 += row[t];
          // four 16-byte loads in flight per lane (round 6: with one, the kernel ran at 1.4 TB/s -- latency-, not bandwidth-bound)
          int t = a0 + 4 * tid;
          for (; t + 3072 < a1; t += 4096) {
            const float4 v0 = *reinterpret_cast<const float4*>(row + t);
            const float4 v1 = *reinterpret_cast<const float4*>(row + t + 1024);
            const float4 v2 = *reinterpret_cast<const float4*>(row + t + 2048);
            const float4 v3 = *reinterpret_cast<const float4*>(row + t + 3072);
            s0 += (v0.x + v1.x) + (v2.x + v3.x); s1 += (v0.y + v1.y) + (v2.y + v3.y);
            s2 += (v0.z + v1.z) + (v2.z + v3.z); s3 += (v0.w + v1.w) + (v2.w + v3.w);
          }
          for (; t < a1; t += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(row + t);
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
          }
          t0 = a1;
        }
      }
      for (int t = t0 + tid; t < t1; t += 256) s0 += row[t];
    }
  }
  red[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) part[c * RED_NS + sl] = red[0];
  if (sl == 0 && tid > 0 && tid < RED_NS && tid >= ns) part[c * RED_NS + tid] = 0.f;      // unused slices of this channel's row
}

__global__ void channel_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int i = 0; i < RED_NS; ++i) s += part[c * RED_NS + i];
  out[c] = s;
}

static int wgrad_geometry(int B, int C_in, int C_out, int T_out, int K, int* cit, int* splits, int* n_tt, int* per) {
  if (K > 128) return -1;
  *cit = 128 / K;
  *n_tt = (T_out + WG_TT - 1) / WG_TT;
  const long long tiles = (long long)B * *n_tt;
  const long long wgs = (long long)((C_out + 63) / 64) * ((C_in + *cit - 1) / *cit);
  // few-channel layers have only a handful of (co, ci) tiles: split their long (b, t) range across many more
  // workgroups (the partial buffers stay small exactly there); bounded by 512 MB of partials
  long long S = (2048 + wgs - 1) / wgs;
  if (S > tiles) S = tiles;
  if (S > 1024) S = 1024;
  const long long per_split_bytes = (long long)C_out * C_in * K * 4;
  if (S * per_split_bytes > (512ll << 20)) S = (512ll << 20) / per_split_bytes;
  if (S < 1) S = 1;
  *per = (int)((tiles + S - 1) / S);
  *splits = (int)((tiles + *per - 1) / *per);
  return 0;
}

}  // namespace fac

extern "C" int fac_pack_conv_w_bwd(const float* v, const float* scale, float* packed, int C_out, int C_in, int K,
                                   int C_in_pad, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(v && packed && C_out > 0 && C_in > 0 && K > 0 && C_in_pad >= C_in && C_in_pad % 32 == 0,
              "pack_conv_w_bwd: bad arguments");
  const long long n = (long long)cin_pad_dev(C_out) * K * C_in_pad;     // the whole padded buffer
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  if (prep_recording()) {
    PrepJob j{}; j.a = v; j.b = scale; j.out = packed; j.kind = PK_CONV_BWD; j.nblocks = blocks; j.n = n;
    j.i[0] = C_out; j.i[1] = C_in; j.i[2] = K; j.i[3] = C_in_pad;
    return prep_record(PU_BWD, j);
  }
  hipLaunchKernelGGL(pack_conv_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, scale, packed, C_out, C_in,
                     K, C_in_pad, n);
  return check_launch("pack_conv_w_bwd");
}

extern "C" int fac_pad_fold_bwd(const float* dxpad, float* dx, int B, int C, int T, int Tp, int pad_left, int pad_mode,
                                fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(dxpad && dx && B > 0 && C > 0 && T > 0 && pad_left >= 0 && Tp >= pad_left + T, "pad_fold_bwd: bad arguments");
  FAC_REQUIRE(pad_mode != FAC_PAD_REFLECT || (T > pad_left && T > Tp - pad_left - T),
              "pad_fold_bwd: reflect padding needs a signal longer than the pad");
  const long long n = (long long)B * C * T;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(pad_fold_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dxpad, dx, T, Tp, pad_left,
                     pad_mode, n);
  return check_launch("pad_fold_bwd");
}

extern "C" int fac_pad_fold_edges(float* dxpad, int B, int C, int T, int Tp, int pad_left, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(dxpad && B > 0 && C > 0 && T > 0 && pad_left >= 0 && Tp >= pad_left + T, "pad_fold_edges: bad arguments");
  const int pad_right = Tp - pad_left - T;
  // a sample must not be both a mirror target of the left edge and of the right edge, and the mirrored ranges must lie inside
  // the signal: T > pad_left + pad_right keeps the two edge regions disjoint (each thread then owns its target)
  FAC_REQUIRE(T > pad_left + pad_right, "pad_fold_edges: signal shorter than its padding (use fac_pad_fold_bwd)");
  if (pad_left + pad_right == 0) return FAC_OK;
  const long long n = (long long)B * C * (pad_left + pad_right);
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(pad_fold_edges_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dxpad, T, Tp, pad_left, (long long)B * C);
  return check_launch("pad_fold_edges");
}

extern "C" int64_t fac_conv1d_bwd_weight_ws_bytes(int B, int C_in, int C_out, int T_out, int K) {
  int cit, S, n_tt, per;
  if (fac::wgrad_geometry(B, C_in, C_out, T_out, K, &cit, &S, &n_tt, &per)) return -1;
  return (int64_t)S * C_out * C_in * K * 4;
}

extern "C" int fac_conv1d_bwd_weight(const float* x, const float* dy, float* dw, void* ws, int64_t ws_bytes, int B,
                                     int C_in, int T_in, int C_out, int T_out, int K, int stride, int dilation,
                                     int pad_left, int pad_mode, int K1, int dilation2, fac_stream_t stream) {
  using namespace fac;
  if (K1 <= 0 || K1 > K) K1 = K;
  FAC_REQUIRE(K % K1 == 0 && (K1 == K || dilation2 > 0), "conv1d_bwd_weight: bad two-level taps");
  const int K2 = K / K1, K_total = K, C_in_real = C_in;
  K = K1;                  // the kernel sees K2 virtual input channels per real one, K1 taps each
  C_in = C_in_real * K2;
  FAC_REQUIRE(x && dy && dw && ws && B > 0 && C_in > 0 && C_out > 0 && T_in > 0 && T_out > 0 && K > 0 && stride > 0 &&
                  dilation > 0 && pad_left >= 0,
              "conv1d_bwd_weight: bad arguments");
  WgArgs a;
  int S;
  FAC_REQUIRE(wgrad_geometry(B, C_in, C_out, T_out, K, &a.CIT, &S, &a.n_tt, &a.tiles_per_split) == 0,
              "conv1d_bwd_weight: K=%d too large", K);
  FAC_REQUIRE(ws_bytes >= (int64_t)S * C_out * C_in * K * 4, "conv1d_bwd_weight: workspace too small");
  a.x = x; a.dy = dy; a.part = reinterpret_cast<float*>(ws);
  a.x_bs = (long long)C_in_real * T_in; a.x_cs = T_in; a.dy_bs = (long long)C_out * T_out; a.dy_cs = T_out;
  a.B = B; a.C_in = C_in; a.T_in = T_in; a.C_out = C_out; a.T_out = T_out; a.K = K; a.stride = stride; a.dil = dilation;
  a.pad_left = pad_left; a.pad_mode = pad_mode; a.K2 = K2; a.dil2 = K2 > 1 ? dilation2 : 0;
  (void)K_total;
  {
    long long last = (long long)(T_out - 1) * stride + (long long)(K2 - 1) * a.dil2 + (long long)(K - 1) * dilation - pad_left;
    int pad_right = last >= T_in ? (int)(last - T_in + 1) : 0;
    int max_pad = pad_left > pad_right ? pad_left : pad_right;
    a.T_ext = T_in > max_pad ? T_in : max_pad + 1;
  }
  a.XWl = (WG_TT - 1) * stride + (K - 1) * dilation + 1;
  const size_t lds = ((size_t)64 * (WG_TT + 1) + (size_t)a.CIT * a.XWl) * sizeof(float);
  FAC_REQUIRE(lds <= 160 * 1024, "conv1d_bwd_weight: tile needs %zu B of LDS", lds);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_wgrad_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid((C_out + 63) / 64, (C_in + a.CIT - 1) / a.CIT, S);
  hipLaunchKernelGGL(conv1d_wgrad_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
  const long long n = (long long)C_out * C_in * K;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a.part, dw, S, n);
  return check_launch("conv1d_bwd_weight");
}

extern "C" int fac_weight_norm_bwd(const float* v, const float* g, const float* dw, float* dv, float* dg, int n_rows,
                                   int row_len, fac_stream_t stream) {
  FAC_REQUIRE(v && g && dw && dv && dg && n_rows > 0 && row_len > 0, "weight_norm_bwd: bad arguments");
  hipLaunchKernelGGL(fac::weight_norm_bwd_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, v, g, dw, dv, dg, row_len);
  return fac::check_launch("weight_norm_bwd");
}

// slices of the (b, t) range per channel for the reduction kernels above: one per ~8 K work items, at most RED_NS
static unsigned red_slices(long long items) {
  long long ns = (items + 8191) / 8192;
  return (unsigned)(ns < 1 ? 1 : (ns > fac::RED_NS ? fac::RED_NS : ns));
}

extern "C" int fac_snake_bwd(const float* x, const float* alpha, const float* dy, float* dx, float* dalpha, float* scratch,
                             int B, int C, int T, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(x && alpha && dy && dx && dalpha && scratch && B > 0 && C > 0 && T > 0, "snake_bwd: bad arguments");
  hipLaunchKernelGGL(snake_bwd_kernel, dim3(C, red_slices((long long)B * T)), dim3(256), 0, (hipStream_t)stream, x, alpha, dy, dx, scratch, B, C, T);
  hipLaunchKernelGGL(channel_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, dalpha, C);
  return check_launch("snake_bwd");
}

extern "C" int fac_snake_bwd_fused_rs(const float* x, const float* alpha, const float* dy, long long dy_row_stride, const float* add, float* dx,
                                      float* dalpha, float* dbias, float* scratch, int B, int C, int T, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(x && alpha && dy && dx && dalpha && scratch && B > 0 && C > 0 && T > 0 && dy_row_stride >= T, "snake_bwd_fused: bad arguments");
  float* part2 = dbias ? scratch + (long long)RED_NS * C : nullptr;
  static const bool v4_on = !(getenv("FAC_SNAKE_BWD_V4") && getenv("FAC_SNAKE_BWD_V4")[0] == '0');
  auto al = [](const void* p, unsigned m) { return (reinterpret_cast<unsigned long long>(p) & m) == 0; };
  if (v4_on && (T & 3) == 0 && (dy_row_stride & 1) == 0 && al(x, 15) && al(dx, 15) && (!add || al(add, 15)) && al(dy, 7))
    hipLaunchKernelGGL(snake_bwd_fused_v4_kernel, dim3(C, red_slices((long long)B * T / 4)), dim3(256), 0, (hipStream_t)stream, x, alpha, dy, add,
                       dx, scratch, part2, B, C, T, dy_row_stride);
  else
    hipLaunchKernelGGL(snake_bwd_fused_kernel, dim3(C, red_slices((long long)B * T)), dim3(256), 0, (hipStream_t)stream, x, alpha, dy, add, dx,
                       scratch, part2, B, C, T, dy_row_stride);
  hipLaunchKernelGGL(channel_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, dalpha, C);
  if (dbias)
    hipLaunchKernelGGL(channel_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, part2, dbias, C);
  return check_launch("snake_bwd_fused");
}

extern "C" int fac_snake_bwd_fused(const float* x, const float* alpha, const float* dy, const float* add, float* dx, float* dalpha,
                                   float* dbias, float* scratch, int B, int C, int T, fac_stream_t stream) {
  return fac_snake_bwd_fused_rs(x, alpha, dy, T, add, dx, dalpha, dbias, scratch, B, C, T, stream);
}

extern "C" int fac_bias_grad(const float* dy, float* db, float* scratch, int B, int C, int T, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(dy && db && scratch && B > 0 && C > 0 && T > 0, "bias_grad: bad arguments");
  // one slice per ~8 K elements of a channel (round 6: (C, 32) workgroups for a (16, 1024, 160) tensor were 32 768 workgroups of
  // 256 threads for 2.6 M elements -- 50 us; 43 such launches per train step)
  hipLaunchKernelGGL(bias_grad_kernel, dim3(C, red_slices((long long)B * T)), dim3(256), 0, (hipStream_t)stream, dy, scratch, B, C, T);
  hipLaunchKernelGGL(channel_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, db, C);
  return check_launch("bias_grad");
}

// ------------------------------------------------------------------------------------------------------------
// Quantizer backward (dac/nn/quantize.py:55-67): z_st = z_e + (codebook[idx] - z_e) feeds out_proj (straight-through:
// d z_e += d z_st); commitment mse(z_e, z_q.detach()) and codebook mse(z_q, z_e.detach()) are means over the (8, T)
// plane per sample, weighted per sample by wc[b] / wb[b] (quantizer-dropout mask, 1/B and the loss weight folded in).
namespace fac {

constexpr int VQB_CD = 8;

__global__ void vq_latent_bwd_kernel(const float* __restrict__ z_e, const float* __restrict__ cb,
                                     const long long* __restrict__ codes, long long codes_bs, const float* __restrict__ d_zst,
                                     const float* __restrict__ wc, float* __restrict__ d_ze, float* __restrict__ z_st, int T,
                                     long long n) {
  const float inv = 2.0f / (float)(VQB_CD * T);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long long r = i / T;
    const int d = (int)(r % VQB_CD);
    const int b = (int)(r / VQB_CD);
    const float ze = z_e[i];
    const float zq = cb[codes[(long long)b * codes_bs + t] * VQB_CD + d];
    if (d_ze) d_ze[i] = (d_zst ? d_zst[i] : 0.f) + wc[b] * inv * (ze - zq);
    if (z_st) z_st[i] = __fadd_rn(ze, __fsub_rn(zq, ze));
  }
}

// one workgroup per code: gathers every position that chose it (deterministic, no atomics)
__global__ __launch_bounds__(256) void vq_codebook_grad_kernel(const float* __restrict__ z_e, const float* __restrict__ cb,
                                                               const long long* __restrict__ codes, long long codes_bs,
                                                               const float* __restrict__ wb, float* __restrict__ dcb, int B,
                                                               int T, int accumulate) {
  __shared__ float red[VQB_CD][256];
  const int k = blockIdx.x, tid = threadIdx.x;
  const float inv = 2.0f / (float)(VQB_CD * T);
  float s[VQB_CD];
#pragma unroll
  for (int d = 0; d < VQB_CD; ++d) s[d] = 0.f;
  for (long long p = tid; p < (long long)B * T; p += 256) {
    const int b = (int)(p / T), t = (int)(p - (long long)b * T);
    if (codes[(long long)b * codes_bs + t] != k) continue;
    const float w = wb[b] * inv;
#pragma unroll
    for (int d = 0; d < VQB_CD; ++d) s[d] += w * (cb[k * VQB_CD + d] - z_e[((long long)b * VQB_CD + d) * T + t]);
  }
#pragma unroll
  for (int d = 0; d < VQB_CD; ++d) red[d][tid] = s[d];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o)
#pragma unroll
      for (int d = 0; d < VQB_CD; ++d) red[d][tid] += red[d][tid + o];
    __syncthreads();
  }
  if (tid < VQB_CD) dcb[k * VQB_CD + tid] = (accumulate ? dcb[k * VQB_CD + tid] : 0.f) + red[tid][0];
}

// LayerNorm over channels + per-clip affine (modules/quantize.py:444-449): out = xhat * gamma_b + beta_b.
// Column kernel: dx; row kernel: dgamma / dbeta (sums over time, one workgroup per (b, c)).
__global__ __launch_bounds__(256) void layernorm_c_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ style,
                                                                const float* __restrict__ dout, float* __restrict__ dx,
                                                                float* __restrict__ stats, int C, int T) {
  __shared__ float red[4][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 64 + lane;
  const bool tv = t < T;
  const float* xb = x + (long long)b * C * T + (tv ? t : T - 1);
  const float* db = dout + (long long)b * C * T + (tv ? t : T - 1);
  const float* gm = style + (long long)b * 2 * C;
  // Round 6: loads go out 16 at a time per chain (they were C / 4 dependent round trips per lane: 288 us for 10 MB), and the
  // column statistics the row kernel needs are written from here (a separate single-wave kernel recomputed them in 217 us).
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
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  if (tv && wave == 0) {
    stats[((long long)b * T + t) * 2] = mean;
    stats[((long long)b * T + t) * 2 + 1] = rstd;
  }
  float s1 = 0.f, s2 = 0.f;          // sum of dxhat, sum of dxhat * xhat
  for (int c0 = wave; c0 < C; c0 += 4 * U) {
    float bx[U], bd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = c0 + 4 * u < C;
      bx[u] = in ? xb[(long long)(c0 + 4 * u) * T] : mean;
      bd[u] = in ? db[(long long)(c0 + 4 * u) * T] * gm[c0 + 4 * u] : 0.f;
    }
    if (tv) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float xh = (bx[u] - mean) * rstd;
        s1 += bd[u];
        s2 = fmaf(bd[u], xh, s2);
      }
    }
  }
  red[2][wave][lane] = s1;
  red[3][wave][lane] = s2;
  __syncthreads();
  const float m1 = ((red[2][0][lane] + red[2][1][lane]) + (red[2][2][lane] + red[2][3][lane])) / (float)C;
  const float m2 = ((red[3][0][lane] + red[3][1][lane]) + (red[3][2][lane] + red[3][3][lane])) / (float)C;
  if (!tv) return;
  float* ob = dx + (long long)b * C * T + t;
#pragma unroll 8
  for (int c = wave; c < C; c += 4) {
    const float xh = (xb[(long long)c * T] - mean) * rstd;
    const float dxh = db[(long long)c * T] * gm[c];
    ob[(long long)c * T] = rstd * (dxh - m1 - xh * m2);
  }
}

// dgamma / dbeta: sums over time, one workgroup per (b, c); the column statistics come from layernorm_c_bwd_x_kernel.
__global__ __launch_bounds__(64) void layernorm_c_bwd_style_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                   const float* __restrict__ dout, float* __restrict__ dstyle,
                                                                   int C, int T) {
  const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const float* xr = x + ((long long)b * C + c) * T;
  const float* dr = dout + ((long long)b * C + c) * T;
  float sg = 0.f, sb = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float mean = stats[((long long)b * T + t) * 2], rstd = stats[((long long)b * T + t) * 2 + 1];
    const float g = dr[t];
    sg = fmaf(g, (xr[t] - mean) * rstd, sg);
    sb += g;
  }
  for (int o = 32; o > 0; o >>= 1) {
    sg += __shfl_down(sg, o, 64);
    sb += __shfl_down(sb, o, 64);
  }
  if (lane == 0) {
    dstyle[(long long)b * 2 * C + c] = sg;
    dstyle[(long long)b * 2 * C + C + c] = sb;
  }
}

}  // namespace fac

extern "C" int fac_vq_latent_bwd(const float* z_e, const float* codebook, const int64_t* codes, int64_t codes_bs,
                                 const float* d_zst, const float* wc, float* d_ze, float* z_st, int B, int T,
                                 fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(z_e && codebook && codes && (d_ze || z_st) && (!d_ze || wc) && B > 0 && T > 0, "vq_latent_bwd: bad arguments");
  const long long n = (long long)B * VQB_CD * T;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(vq_latent_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z_e, codebook,
                     (const long long*)codes, (long long)codes_bs, d_zst, wc, d_ze, z_st, T, n);
  return check_launch("vq_latent_bwd");
}

extern "C" int fac_vq_codebook_grad(const float* z_e, const float* codebook, const int64_t* codes, int64_t codes_bs,
                                    const float* wb, float* dcb, int B, int T, int Kc, int accumulate, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(z_e && codebook && codes && wb && dcb && B > 0 && T > 0 && Kc > 0, "vq_codebook_grad: bad arguments");
  hipLaunchKernelGGL(vq_codebook_grad_kernel, dim3(Kc), dim3(256), 0, (hipStream_t)stream, z_e, codebook,
                     (const long long*)codes, (long long)codes_bs, wb, dcb, B, T, accumulate);
  return check_launch("vq_codebook_grad");
}

extern "C" int fac_layernorm_c_affine_bwd(const float* x, const float* style, const float* dout, float* dx, float* dstyle,
                                          float* stats, int B, int C, int T, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(x && style && dout && dx && dstyle && stats && B > 0 && C > 0 && T > 0, "layernorm_c_affine_bwd: bad arguments");
  hipLaunchKernelGGL(layernorm_c_bwd_x_kernel, dim3((T + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, style, dout, dx, stats, C, T);
  hipLaunchKernelGGL(layernorm_c_bwd_style_kernel, dim3(C, B), dim3(64), 0, (hipStream_t)stream, x, stats, dout, dstyle, C, T);
  return check_launch("layernorm_c_affine_bwd");
}
