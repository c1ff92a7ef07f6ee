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
  int B, C_in, C_out, T;   // C_in: columns of dW = (virtual) input rows; T: time steps of dy
  int Cx, Tx;              // real channels and row length of the x buffer
  int K, K1, dil, dil2;    // column v = (ci, k) = (v / K, v % K) reads x[ci][t + (k / K1) * dil2 + (k % K1) * dil]  (k = 1 conv: K = K1 = 1)
  int max_off;             // largest such offset
  int wins_per_clip;       // ceil(T / 32)
  int n_win;               // B * wins_per_clip
  int n_cg;                // column groups of CB blocks (roles = row groups x column groups)
  int roles;
  int wpr;                 // waves per role
};

template <int RB, int CB, bool VIRT>
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

  // column block n of this lane: virtual row v -> (channel, tap offset) of the x buffer (columns past the end repeat the last one
  // and are never stored)
  struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };      // 16-byte load from a 4-byte aligned address (tap offsets)
  long long xo[VIRT ? CB : 1];
  int xoff[VIRT ? CB : 1];
  if constexpr (VIRT) {
#pragma unroll
    for (int n = 0; n < CB; ++n) {
      int v = (cg * CB + n) * 32 + r;
      v = v < a.C_in ? v : a.C_in - 1;
      const int ci = v / a.K, k = v - ci * a.K;
      xoff[n] = (k / a.K1) * a.dil2 + (k % a.K1) * a.dil;
      xo[n] = (long long)ci * a.Tx + xoff[n];
    }
  }
  float4 A[2][RB][4], Bv[2][CB][4];
  auto load = [&](int w, float4 (&Ad)[RB][4], float4 (&Bd)[CB][4]) {
    const int b = w / a.wins_per_clip;
    const int t = (w - b * a.wins_per_clip) * 32 + 16 * h;
    const float* pd = a.dy + ((long long)b * a.C_out + rg * RB * 32 + r) * a.T + t;
    const float* px = VIRT ? a.x + (long long)b * a.Cx * a.Tx + t : a.x + ((long long)b * a.C_in + cg * CB * 32 + r) * a.T + t;
    if (t + 16 <= a.T && (!VIRT || t + 16 + a.max_off <= a.Tx)) {
      // piece q of every fragment, then piece q + 1: the four accesses to a 128-byte line are RB + CB instructions apart (the fill is
      // back before the next one asks: C = 192 0.33 -> 0.30 ms against fragment-by-fragment order)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int m = 0; m < RB; ++m) Ad[m][q] = *reinterpret_cast<const float4*>(pd + (long long)m * 32 * a.T + 4 * q);
#pragma unroll
        for (int n = 0; n < CB; ++n) {
          if constexpr (VIRT) {
            const f4u u = *reinterpret_cast<const f4u*>(px + xo[n] + 4 * q);
            Bd[n][q] = make_float4(u.x, u.y, u.z, u.w);
          } else {
            Bd[n][q] = *reinterpret_cast<const float4*>(px + (long long)n * 32 * a.T + 4 * q);
          }
        }
      }
    } else {                                             // the last window of a clip: steps beyond T contribute zeros
      auto guarded = [&](const float* p, int lim, float4 (&d)[4]) {          // lim: valid elements from p on
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[q].x = 4 * q + 0 < lim ? p[4 * q + 0] : 0.f;
          d[q].y = 4 * q + 1 < lim ? p[4 * q + 1] : 0.f;
          d[q].z = 4 * q + 2 < lim ? p[4 * q + 2] : 0.f;
          d[q].w = 4 * q + 3 < lim ? p[4 * q + 3] : 0.f;
        }
      };
#pragma unroll
      for (int m = 0; m < RB; ++m) guarded(pd + (long long)m * 32 * a.T, a.T - t, Ad[m]);
#pragma unroll
      for (int n = 0; n < CB; ++n) {
        if constexpr (VIRT) guarded(px + xo[n], a.Tx - t - xoff[n], Bd[n]);
        else guarded(px + (long long)n * 32 * a.T, a.T - t, Bd[n]);
      }
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
  auto rowsum = [&](const float4 (&Ac)[RB][4]) {         // (a running sum: no temporaries next to 340 live registers)
#pragma unroll
    for (int m = 0; m < RB; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) rs[m] += (Ac[m][q].x + Ac[m][q].y) + (Ac[m][q].z + Ac[m][q].w);
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
        const int col = (cg * CB + n) * 32 + r;
        if (!VIRT || col < a.C_in) pz[(long long)co * a.C_in + col] = acc[m][n][i];
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
  a->Cx = C_in; a->Tx = T; a->K = 1; a->K1 = 1; a->dil = 1; a->dil2 = 0; a->max_off = 0;
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

// The same kernel for stride-1 convs with FEW INPUT CHANNELS and taps (the first layers: 1 -> 64 k = 7 of the encoder, 2 -> 32 (3, 9) of
// the multi-resolution discriminator): the columns of dW are the C_in * K virtual rows (ci, k), each a shifted view of one row of the
// (padded) input -- lane v of a column block loads its 16 steps from x[ci][t + offset(k)], a 4-byte aligned 16-byte load of a row
// that sits in the L2s.  C_out = 32 or 64, C_in * K <= 64.  Returns S (0: not here) and the quadrant (RB, CB).
static int wk1_taps_geometry(int B, int C_in, int Tx, int C_out, int T_out, int K, int K1, int dil, int dil2, Wk1Args* a, int* RBo, int* CBo) {
  if (K1 <= 0 || K1 > K) K1 = K;
  if (B <= 0 || T_out < 4096 || C_in <= 0 || K < 1 || K % K1 != 0 || dil < 1 || (K1 < K && dil2 < 1) || C_in * K > 64 || (C_out != 32 && C_out != 64)) return 0;
  const int max_off = (K / K1 - 1) * (K1 < K ? dil2 : 0) + (K1 - 1) * dil;
  const int wins = (T_out + 31) / 32;
  if (Tx < wins * 32 + max_off) return 0;                 // every window reads inside its row
  a->B = B; a->C_in = C_in * K; a->C_out = C_out; a->T = T_out;
  a->Cx = C_in; a->Tx = Tx; a->K = K; a->K1 = K1; a->dil = dil; a->dil2 = K1 < K ? dil2 : 0; a->max_off = max_off;
  a->wins_per_clip = wins;
  a->n_win = B * wins;
  a->n_cg = 1; a->roles = 1; a->wpr = WK1_WAVES;
  long long S = 256;
  const long long max_s = a->n_win / (8ll * a->wpr);
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  *RBo = C_out / 32;
  *CBo = (C_in * K + 31) / 32;
  return (int)S;
}

}  // namespace fac

// Row length the padded input of fac_conv1d_bwd_weight_taps must have (zeros behind the last position a valid output reads).
extern "C" int64_t fac_conv1d_bwd_weight_taps_tx(int T_out, int K, int K1, int dilation, int dilation2) {
  if (K1 <= 0 || K1 > K) K1 = K;
  if (K < 1 || T_out < 1 || K % K1 != 0) return -1;
  const int64_t max_off = (int64_t)(K / K1 - 1) * (K1 < K ? dilation2 : 0) + (int64_t)(K1 - 1) * dilation;
  return (((int64_t)T_out + 31) / 32) * 32 + max_off;
}

extern "C" int64_t fac_conv1d_bwd_weight_taps_ws_bytes(int B, int C_in, int C_out, int T_out, int K, int K1, int dilation, int dilation2) {
  using namespace fac;
  Wk1Args a;
  int RB, CB;
  const int64_t tx = fac_conv1d_bwd_weight_taps_tx(T_out, K, K1, dilation, dilation2);
  if (tx < 0 || tx > 0x7fffffff) return -1;
  const int S = wk1_taps_geometry(B, C_in, (int)tx, C_out, T_out, K, K1, dilation, dilation2, &a, &RB, &CB);
  return S > 0 ? (int64_t)S * a.wpr * ((int64_t)C_out * C_in * K + C_out) * 4 : -1;
}

// xpad (B, C_in, Tx): the conv's PADDED input (position p of the output's receptive field origin at p = t), Tx >= fac_..._taps_tx.
// dW (C_out, C_in, K)[co][ci][k] = sum over (b, t < T_out) of dy[b][co][t] * xpad[b][ci][t + (k / K1) * dilation2 + (k % K1) * dilation].
extern "C" int fac_conv1d_bwd_weight_taps(const float* xpad, const float* dy, float* dw, float* db, void* ws, int64_t ws_bytes, int B,
                                          int C_in, int Tx, int C_out, int T_out, int K, int K1, int dilation, int dilation2,
                                          fac_stream_t stream) {
  using namespace fac;
  Wk1Args a;
  int RB, CB;
  const int S = wk1_taps_geometry(B, C_in, Tx, C_out, T_out, K, K1, dilation, dilation2, &a, &RB, &CB);
  FAC_REQUIRE(xpad && dy && dw && ws && S > 0, "conv1d_bwd_weight_taps: shape not supported (query fac_conv1d_bwd_weight_taps_ws_bytes / _tx)");
  const int64_t n_dw = (int64_t)C_out * C_in * K;
  FAC_REQUIRE(ws_bytes >= (int64_t)S * a.wpr * (n_dw + C_out) * 4, "conv1d_bwd_weight_taps: workspace too small");
  a.x = xpad; a.dy = dy; a.part = reinterpret_cast<float*>(ws); a.with_db = db != nullptr ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(S), blk(64 * WK1_WAVES);
  if (RB == 1 && CB == 1) hipLaunchKernelGGL((wgrad_k1_kernel<1, 1, true>), g, blk, 0, st, a);
  else if (RB == 1) hipLaunchKernelGGL((wgrad_k1_kernel<1, 2, true>), g, blk, 0, st, a);
  else if (CB == 1) hipLaunchKernelGGL((wgrad_k1_kernel<2, 1, true>), g, blk, 0, st, a);
  else hipLaunchKernelGGL((wgrad_k1_kernel<2, 2, true>), g, blk, 0, st, a);
  const long long n = n_dw + (db != nullptr ? C_out : 0);
  hipLaunchKernelGGL(wgrad_k1_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, dw, db, S * a.wpr, n, (long long)n_dw);
  return check_launch("conv1d_bwd_weight_taps");
}

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
  if (Q == 3) hipLaunchKernelGGL((wgrad_k1_kernel<3, 3, false>), dim3(S), dim3(64 * WK1_WAVES), 0, st, a);
  else hipLaunchKernelGGL((wgrad_k1_kernel<2, 2, false>), dim3(S), dim3(64 * WK1_WAVES), 0, st, a);
  const long long n_dw = (long long)C_out * C_in, n = n_dw + (db != nullptr ? C_out : 0);
  hipLaunchKernelGGL(wgrad_k1_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, dw, db, S * a.wpr, n, n_dw);
  return check_launch("conv1d_bwd_weight_k1");
}
