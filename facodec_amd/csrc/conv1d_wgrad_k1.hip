// Weight gradient of the k = 1 ResidualUnit tails with few channels (C = 64 / 96 / 192 at T = 48 000 / 24 000: dac/model/dac.py:25-42,
// backward of train.py:361) straight from the fp32 tensors, on the fp32 matrix pipe:
//     dW[co][ci] = sum over (b, t) of dy[b][co][t] * x[b][ci][t]
// 6 - 28 GFLOP over two tensors of 200 - 300 MB: on the split kernel (conv1d_wgrad_split.hip) such a launch is its two operand-split
// passes and the plane reads -- 32 bytes per element pair against the 8 of the fp32 tensors (0.37 / 0.53 / 0.60 ms at B = 16) -- while
// its matrix work would fit the fp32 pipe inside the time the two tensors take to stream in ONCE (v_mfma_f32_32x32x2_f32, 157 TFLOP/s:
// 0.04 - 0.2 ms).  So: no planes, no LDS staging, no workspace beyond the partial sums.
//
// Operand fragments of v_mfma_f32_32x32x2_f32 are one float per lane: row (lane % 32), contraction slot (lane / 32).  The contraction
// here is over time and a sum does not care which time step sits in which slot as long as BOTH operands agree: half-wave h of a
// 32-step window takes steps [16 h, 16 h + 16), so a lane loads 16 CONSECUTIVE floats of its row (64 bytes, four 16-byte loads: a
// wave instruction covers 32 full 128-byte lines) and MFMA j of the window multiplies register j of both operands.
//
// A wave owns a QUADRANT of RB x CB blocks of 32 x 32 (2 x 2: C = 64 / 128; 3 x 3: C = 96 / 192): RB + CB fragment loads per window for
// RB x CB blocks of matrix work, accumulators in registers (one wave per SIMD: the 512-entry register file is the wave's), the next
// window's fragments requested before the current window's MFMAs (two register sets).  A fragment load touches 32 cache lines (one
// per row, 16 bytes of each) and the texture path takes them a line at a time, so loads per MFMA are what bounds this kernel: with
// one column block per wave (first form: every dy fragment fetched by three or six waves) it ran at a third of the fp32 pipe
// (0.25 / 0.46 ms at C = 96 / 192); quadrants: 0.205 / 0.292 ms (62 % of the pipe at C = 192), C = 64 0.125 ms (3.1 TB/s).
// A workgroup is four waves: the quadrants of the matrix ("roles"; one, two or four of them), WPR = 4 / roles times over; the waves
// of a role take alternate windows; workgroup g takes the window units g, g + S, g + 2 S, ... so that the workgroups running side by
// side read adjacent pieces of the same rows.  Every wave stores its own partial sums; [S x WPR][C_out][C_in] partial matrices are
// added by a second launch in a fixed order (eight interleaved chains).  Deterministic; fp32 products and sums (no operand split:
// the error against fp64 is the fp32 kernels').
#include "common.h"

namespace fac {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WK1_WAVES = 4;      // one per SIMD: a wave holds a whole RB x CB quadrant (up to 144 accumulator + 192 fragment registers)

struct Wk1Args {
  const float* x;
  const float* dy;
  float* part;             // [S * WPR][C_out * C_in (+ C_out row sums of dy when with_db)]
  int with_db;
  int B, C_in, C_out, T;
  int wins_per_clip;       // ceil(T / 32)
  int n_win;               // B * wins_per_clip
  int n_cg;                // column groups of CB blocks (roles = row groups x column groups)
  int roles;
  int wpr;                 // waves per role
};

template <int RB, int CB>
__global__ __launch_bounds__(64 * WK1_WAVES) void wgrad_k1_kernel(Wk1Args a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = wave % a.roles, slot = wave / a.roles;
  const int rg = role / a.n_cg, cg = role - rg * a.n_cg;
  const int r = lane & 31, h = lane >> 5;
  // Workgroup g takes the window units g, g + S, g + 2 S, ... (a unit = WPR consecutive windows, one per wave of a role): the
  // workgroups that run side by side read adjacent pieces of the same rows at about the same time.
  const int unit_stride = gridDim.x * a.wpr;
  if (wave >= a.roles * a.wpr) return;

  f32x16 acc[RB][CB];
#pragma unroll
  for (int m = 0; m < RB; ++m)
#pragma unroll
    for (int n = 0; n < CB; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  float4 A[2][RB][4], Bv[2][CB][4];
  auto load = [&](int w, float4 (&Ad)[RB][4], float4 (&Bd)[CB][4]) {
    const int b = w / a.wins_per_clip;
    const int t = (w - b * a.wins_per_clip) * 32 + 16 * h;
    const float* pd = a.dy + ((long long)b * a.C_out + rg * RB * 32 + r) * a.T + t;
    const float* px = a.x + ((long long)b * a.C_in + cg * CB * 32 + r) * a.T + t;
    if (t + 16 <= a.T) {
      // piece q of every fragment, then piece q + 1: the four accesses to a 128-byte line are RB + CB instructions apart (the fill is
      // back before the next one asks: C = 192 0.33 -> 0.30 ms against fragment-by-fragment order)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int m = 0; m < RB; ++m) Ad[m][q] = *reinterpret_cast<const float4*>(pd + (long long)m * 32 * a.T + 4 * q);
#pragma unroll
        for (int n = 0; n < CB; ++n) Bd[n][q] = *reinterpret_cast<const float4*>(px + (long long)n * 32 * a.T + 4 * q);
      }
    } else {                                             // the last window of a clip: steps beyond T contribute zeros
      auto guarded = [&](const float* p, float4 (&d)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[q].x = t + 4 * q + 0 < a.T ? p[4 * q + 0] : 0.f;
          d[q].y = t + 4 * q + 1 < a.T ? p[4 * q + 1] : 0.f;
          d[q].z = t + 4 * q + 2 < a.T ? p[4 * q + 2] : 0.f;
          d[q].w = t + 4 * q + 3 < a.T ? p[4 * q + 3] : 0.f;
        }
      };
#pragma unroll
      for (int m = 0; m < RB; ++m) guarded(pd + (long long)m * 32 * a.T, Ad[m]);
#pragma unroll
      for (int n = 0; n < CB; ++n) guarded(px + (long long)n * 32 * a.T, Bd[n]);
    }
  };
  auto mma = [&](const float4 (&Ac)[RB][4], const float4 (&Bc)[CB][4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int m = 0; m < RB; ++m)
#pragma unroll
        for (int n = 0; n < CB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[m][q].x, Bc[n][q].x, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < RB; ++m)
#pragma unroll
        for (int n = 0; n < CB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[m][q].y, Bc[n][q].y, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < RB; ++m)
#pragma unroll
        for (int n = 0; n < CB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[m][q].z, Bc[n][q].z, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < RB; ++m)
#pragma unroll
        for (int n = 0; n < CB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[m][q].w, Bc[n][q].w, acc[m][n], 0, 0, 0);
    }
  };

  // bias gradient (train.py:361's db = sum over (b, t) of dy): the column-group-0 waves add up the dy fragments they hold anyway
  float rs[RB];
#pragma unroll
  for (int m = 0; m < RB; ++m) rs[m] = 0.f;
  const bool sums = a.with_db && cg == 0;
  auto rowsum = [&](const float4 (&Ac)[RB][4]) {
#pragma unroll
    for (int m = 0; m < RB; ++m) {
      const float s0 = (Ac[m][0].x + Ac[m][0].y) + (Ac[m][0].z + Ac[m][0].w), s1 = (Ac[m][1].x + Ac[m][1].y) + (Ac[m][1].z + Ac[m][1].w);
      const float s2 = (Ac[m][2].x + Ac[m][2].y) + (Ac[m][2].z + Ac[m][2].w), s3 = (Ac[m][3].x + Ac[m][3].y) + (Ac[m][3].z + Ac[m][3].w);
      rs[m] += (s0 + s1) + (s2 + s3);
    }
  };
  int w = blockIdx.x * a.wpr + slot;
  if (w < a.n_win) load(w, A[0], Bv[0]);
  for (; w < a.n_win; w += 2 * unit_stride) {            // two windows per trip: the register sets alternate statically
    const int w1 = w + unit_stride, w2 = w + 2 * unit_stride;
    if (w1 < a.n_win) load(w1, A[1], Bv[1]);
    if (sums) rowsum(A[0]);
    mma(A[0], Bv[0]);
    if (w1 < a.n_win) {
      if (sums) rowsum(A[1]);                            // (before set 0 is requested again: the sums read registers, not memory)
      if (w2 < a.n_win) load(w2, A[0], Bv[0]);
      mma(A[1], Bv[1]);
    }
  }

  // this wave's quadrant of partial-sum matrix (workgroup, slot); the other quadrants come from the other roles of the slot
  const long long pn = (long long)a.C_out * a.C_in + (a.with_db ? a.C_out : 0);
  float* pz = a.part + ((long long)blockIdx.x * a.wpr + slot) * pn;
  if (sums) {
#pragma unroll
    for (int m = 0; m < RB; ++m) {
      const float tot = rs[m] + __shfl_xor(rs[m], 32, 64);             // the two halves of a row's window
      if (h == 0) pz[(long long)a.C_out * a.C_in + (rg * RB + m) * 32 + r] = tot;
    }
  }
#pragma unroll
  for (int m = 0; m < RB; ++m)
#pragma unroll
    for (int n = 0; n < CB; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = (rg * RB + m) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        pz[(long long)co * a.C_in + (cg * CB + n) * 32 + r] = acc[m][n][i];
      }
}

// dw[i] = sum over the S workgroups' partial sums, eight interleaved chains in a fixed order
__global__ __launch_bounds__(256) void wgrad_k1_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                              int S, long long n, long long n_dw) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* p = part + i;
    int z = 0;
    for (; z + 8 <= S; z += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] += p[(long long)(z + j) * n];
    }
    for (int j = 0; z + j < S; ++j) c[j] += p[(long long)(z + j) * n];
    const float v = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    if (i < n_dw) dw[i] = v; else db[i - n_dw] = v;
  }
}

// Geometry; returns the number of workgroups S (0: the shape does not run here).  Few channels, long signals: both channel counts
// multiples of 64 or of 96 (quadrants of 2 x 2 or 3 x 3 blocks), at most four quadrants, 16-byte aligned rows.
static int wk1_geometry(int B, int C_in, int C_out, int T, Wk1Args* a, int* Qo) {
  if (B <= 0 || T < 4096 || T % 4 != 0 || C_in > 192 || C_out > 192 || C_in < 64 || C_out < 64) return 0;
  int Q = 0;
  if (C_in % 96 == 0 && C_out % 96 == 0) Q = 3;
  else if (C_in % 64 == 0 && C_out % 64 == 0) Q = 2;
  if (Q == 0) return 0;
  const int n_rg = C_out / (32 * Q), n_cg = C_in / (32 * Q);
  const int roles = n_rg * n_cg;
  if (roles > WK1_WAVES) return 0;
  a->B = B; a->C_in = C_in; a->C_out = C_out; a->T = T;
  a->wins_per_clip = (T + 31) / 32;
  a->n_win = B * a->wins_per_clip;
  a->n_cg = n_cg; a->roles = roles; a->wpr = WK1_WAVES / roles;
  // one workgroup per CU (one wave per SIMD: the register file is the wave's), at least 8 windows per wave
  long long S = 256;
  const long long max_s = a->n_win / (8ll * a->wpr);
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  *Qo = Q;
  return (int)S;
}

}  // namespace fac

extern "C" int64_t fac_conv1d_bwd_weight_k1_ws_bytes(int B, int C_in, int C_out, int T) {
  using namespace fac;
  Wk1Args a;
  int Q;
  const int S = wk1_geometry(B, C_in, C_out, T, &a, &Q);
  return S > 0 ? (int64_t)S * a.wpr * ((int64_t)C_out * C_in + C_out) * 4 : -1;
}

extern "C" int fac_conv1d_bwd_weight_k1(const float* x, const float* dy, float* dw, float* db, void* ws, int64_t ws_bytes, int B,
                                        int C_in, int C_out, int T, fac_stream_t stream) {
  using namespace fac;
  Wk1Args a;
  int Q;
  const int S = wk1_geometry(B, C_in, C_out, T, &a, &Q);
  FAC_REQUIRE(x && dy && dw && ws && S > 0, "conv1d_bwd_weight_k1: shape not supported (query fac_conv1d_bwd_weight_k1_ws_bytes)");
  FAC_REQUIRE(ws_bytes >= (int64_t)S * a.wpr * ((int64_t)C_out * C_in + C_out) * 4, "conv1d_bwd_weight_k1: workspace too small");
  a.x = x; a.dy = dy; a.part = reinterpret_cast<float*>(ws); a.with_db = db != nullptr ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (Q == 3) hipLaunchKernelGGL((wgrad_k1_kernel<3, 3>), dim3(S), dim3(64 * WK1_WAVES), 0, st, a);
  else hipLaunchKernelGGL((wgrad_k1_kernel<2, 2>), dim3(S), dim3(64 * WK1_WAVES), 0, st, a);
  const long long n_dw = (long long)C_out * C_in, n = n_dw + (db != nullptr ? C_out : 0);
  hipLaunchKernelGGL(wgrad_k1_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, dw, db, S * a.wpr, n, n_dw);
  return check_launch("conv1d_bwd_weight_k1");
}
