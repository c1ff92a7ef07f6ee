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
// A wave owns RB row blocks (32 output channels each) x ONE column block (32 input channels); a workgroup holds one wave per (row
// group, column block) "role" of its column-block set, WPR times over: waves of the same role take alternate windows of the
// workgroup's range of (clip, window) pairs and are added in a fixed order through LDS at the end; the workgroups' partial sums
// [S][C_out][C_in] are added by a second launch in a fixed order (eight interleaved chains).  Deterministic; fp32 products and sums
// (no operand split: the error against fp64 is the fp32 kernels').  The next window's fragments are requested before the current
// window's 16 RB MFMAs are issued (two register sets).  [Second form: twelve waves per workgroup, three per SIMD, ONE register set --
// the waves of a SIMD cover each other's loads, every SIMD carries the same matrix work (six waves left two SIMDs with half of it),
// and all roles of a C = 192 layer fit one workgroup, so dy is read once: 0.25 -> MEASURED_96 ms at C = 96, 0.43 -> MEASURED_192 at C = 192.]
#include "common.h"

namespace fac {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WK1_WAVES = 12;

struct Wk1Args {
  const float* x;
  const float* dy;
  float* part;             // [S][C_out][C_in]
  int B, C_in, C_out, T;
  int wins_per_clip;       // ceil(T / 32)
  int n_win;               // B * wins_per_clip
  int win_per_wg;          // windows per workgroup (a multiple of WPR)
  int n_rg;                // row groups of RB blocks
  int cb_per_wg;           // column blocks per workgroup (grid.y sets of them)
  int wpr;                 // waves per role
};

template <int RB>
__global__ __launch_bounds__(768) void wgrad_k1_kernel(Wk1Args a) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [wave][RB][16][64] for the end-of-range exchange
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int roles = a.n_rg * a.cb_per_wg;
  const int role = wave % roles, slot = wave / roles;
  const int rg = role / a.cb_per_wg;
  const int cb = blockIdx.y * a.cb_per_wg + role % a.cb_per_wg;
  const int r = lane & 31, h = lane >> 5;
  // Workgroup g takes the window units g, g + S, g + 2 S, ... (a unit = WPR consecutive windows, one per wave of a role): the
  // workgroups that run side by side read ADJACENT 128-byte pieces of the same rows at about the same time -- DRAM pages and TLB
  // entries are shared across the chip -- where contiguous ranges per workgroup made 49 000 concurrent 128-byte streams.
  const int unit_stride = gridDim.x * a.wpr;

  f32x16 acc[RB];
#pragma unroll
  for (int m = 0; m < RB; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;

  float4 A[RB][4], Bv[4];
  auto load = [&](int w, float4 (&Ad)[RB][4], float4 (&Bd)[4]) {
    const int b = w / a.wins_per_clip;
    const int t = (w - b * a.wins_per_clip) * 32 + 16 * h;
    const float* px = a.x + ((long long)b * a.C_in + cb * 32 + r) * a.T + t;
    if (t + 16 <= a.T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Bd[q] = *reinterpret_cast<const float4*>(px + 4 * q);
#pragma unroll
      for (int m = 0; m < RB; ++m) {
        const float* pd = a.dy + ((long long)b * a.C_out + (rg * RB + m) * 32 + r) * a.T + t;
#pragma unroll
        for (int q = 0; q < 4; ++q) Ad[m][q] = *reinterpret_cast<const float4*>(pd + 4 * q);
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
      guarded(px, Bd);
#pragma unroll
      for (int m = 0; m < RB; ++m) guarded(a.dy + ((long long)b * a.C_out + (rg * RB + m) * 32 + r) * a.T + t, Ad[m]);
    }
  };
  auto mma = [&](const float4 (&Ac)[RB][4], const float4 (&Bc)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int m = 0; m < RB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[m][q].x, Bc[q].x, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < RB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[m][q].y, Bc[q].y, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < RB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[m][q].z, Bc[q].z, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < RB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[m][q].w, Bc[q].w, acc[m], 0, 0, 0);
    }
  };

  const bool active = wave < roles * a.wpr;            // (8 roles: four of the twelve waves have nothing to do)
  // One register set: the three waves of a SIMD cover each other's loads (12 waves x 16 KB in flight per CU).
  for (int w = active ? blockIdx.x * a.wpr + slot : a.n_win; w < a.n_win; w += unit_stride) {
    load(w, A, Bv);
    mma(A, Bv);
  }

  // waves of one role: slot 0 adds slots 1 .. WPR - 1 in that order, then stores the workgroup's partial sums
  if (active && slot != 0) {
#pragma unroll
    for (int m = 0; m < RB; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) red[((wave * RB + m) * 16 + i) * 64 + lane] = acc[m][i];
  }
  __syncthreads();
  if (!active || slot != 0) return;
  for (int s = 1; s < a.wpr; ++s) {
    const int other = s * roles + role;
#pragma unroll
    for (int m = 0; m < RB; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] += red[((other * RB + m) * 16 + i) * 64 + lane];
  }
  float* pz = a.part + (long long)blockIdx.x * a.C_out * a.C_in;
#pragma unroll
  for (int m = 0; m < RB; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = (rg * RB + m) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      pz[(long long)co * a.C_in + cb * 32 + r] = acc[m][i];
    }
}

// dw[i] = sum over the S workgroups' partial sums, eight interleaved chains in a fixed order
__global__ __launch_bounds__(256) void wgrad_k1_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* p = part + i;
    int z = 0;
    for (; z + 8 <= S; z += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] += p[(long long)(z + j) * n];
    }
    for (int j = 0; z + j < S; ++j) c[j] += p[(long long)(z + j) * n];
    dw[i] = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
  }
}

template <int RB> __global__ void wgrad_k1_kernel(Wk1Args a);

static int wk1_occupancy(int RB) {       // resident workgroups per CU (registers / LDS decide: 1 or 2)
  static int occ[4] = {0, 0, 0, 0};
  if (occ[RB] == 0) {
    int n = 0;
    const size_t lds = (size_t)WK1_WAVES * RB * 16 * 64 * 4;
    const hipError_t e = RB == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_k1_kernel<3>, 64 * WK1_WAVES, lds)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_k1_kernel<2>, 64 * WK1_WAVES, lds);
    occ[RB] = (e == hipSuccess && n >= 1) ? (n > 4 ? 4 : n) : 1;
  }
  return occ[RB];
}

// Geometry; returns the number of workgroup ranges S (0: the shape does not run here).  Few channels, long signals: both channel
// counts multiples of 32, at most 6 roles per workgroup, rows dividing into groups of 2 or 3 blocks, 16-byte aligned rows.
static int wk1_geometry(int B, int C_in, int C_out, int T, Wk1Args* a, int* RBo, int* gy) {
  if (B <= 0 || T < 4096 || T % 4 != 0 || C_in % 32 != 0 || C_out % 32 != 0 || C_in > 192 || C_out > 192 || C_in < 64 || C_out < 64) return 0;
  const int nrb = C_out / 32, ncb = C_in / 32;
  const int RB = nrb % 3 == 0 ? 3 : (nrb % 2 == 0 ? 2 : 0);
  if (RB == 0) return 0;
  const int n_rg = nrb / RB;
  if (n_rg > 2) return 0;
  int cbw = 0;
  for (int c = 6; c >= 1; --c)
    if (ncb % c == 0 && n_rg * c <= WK1_WAVES) { cbw = c; break; }
  if (cbw == 0) return 0;
  const int roles = n_rg * cbw, wpr = WK1_WAVES / roles;
  a->B = B; a->C_in = C_in; a->C_out = C_out; a->T = T;
  a->wins_per_clip = (T + 31) / 32;
  a->n_win = B * a->wins_per_clip;
  a->n_rg = n_rg; a->cb_per_wg = cbw; a->wpr = wpr;
  *gy = ncb / cbw;
  // as many workgroups as are resident at once (one round: no tail), at least 8 windows per wave
  long long S = (long long)256 * wk1_occupancy(RB) / *gy;
  const long long max_s = a->n_win / (8ll * wpr);
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  long long per = (a->n_win + S - 1) / S;
  per = (per + wpr - 1) / wpr * wpr;
  a->win_per_wg = (int)per;
  S = (a->n_win + per - 1) / per;
  *RBo = RB;
  return (int)S;
}

}  // namespace fac

extern "C" int64_t fac_conv1d_bwd_weight_k1_ws_bytes(int B, int C_in, int C_out, int T) {
  using namespace fac;
  Wk1Args a;
  int RB, gy;
  const int S = wk1_geometry(B, C_in, C_out, T, &a, &RB, &gy);
  return S > 0 ? (int64_t)S * C_out * C_in * 4 : -1;
}

extern "C" int fac_conv1d_bwd_weight_k1(const float* x, const float* dy, float* dw, void* ws, int64_t ws_bytes, int B, int C_in,
                                        int C_out, int T, fac_stream_t stream) {
  using namespace fac;
  Wk1Args a;
  int RB, gy;
  const int S = wk1_geometry(B, C_in, C_out, T, &a, &RB, &gy);
  FAC_REQUIRE(x && dy && dw && ws && S > 0, "conv1d_bwd_weight_k1: shape not supported (query fac_conv1d_bwd_weight_k1_ws_bytes)");
  FAC_REQUIRE(ws_bytes >= (int64_t)S * C_out * C_in * 4, "conv1d_bwd_weight_k1: workspace too small");
  a.x = x; a.dy = dy; a.part = reinterpret_cast<float*>(ws);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)WK1_WAVES * RB * 16 * 64 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_k1_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_k1_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (RB == 3) hipLaunchKernelGGL(wgrad_k1_kernel<3>, dim3(S, gy), dim3(64 * WK1_WAVES), lds, st, a);
  else hipLaunchKernelGGL(wgrad_k1_kernel<2>, dim3(S, gy), dim3(64 * WK1_WAVES), lds, st, a);
  const long long n = (long long)C_out * C_in;
  hipLaunchKernelGGL(wgrad_k1_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, dw, S, n);
  return check_launch("conv1d_bwd_weight_k1");
}
