// Skinny conv: few output columns (B * T_out <= 640), many weights -- the streaming-inference regime
// (1-2 frames per hop) and every per-clip Linear of the quantizer.  The op is a GEMV-like
// weight stream: the tiled kernel would put one workgroup per 128 output channels and let it walk all of
// C_in * K alone (measured 60-270 us per launch for ~10 MB of weights = a few % of HBM bandwidth).
//
// Here the reduction dimension is split across workgroups as well:
//   grid = (co tiles of 128) x (S slices of the (ci-pair, tap) rows) x (32-column blocks * phases)
//   * one float4 per lane fetches W[ci][k][co0 + 4*i .. +3] (i = lane & 31): a half-wave reads 512
//     contiguous bytes of the packed (C_in, K, C_out) weights, and the four components feed four MFMAs whose
//     output row i stands for channel co0 + 4*i + m -- no LDS, no transposition, weights read once per
//     column block, 32 KB of loads in flight per workgroup;
//   * the B operand (x[ci][t*stride + k*dil - pad]) is a few KB, served by L1/L2;
//   * S > 1: partial tiles go to the caller's workspace and a second small kernel adds them in slice order
//     (deterministic, unlike atomics on the outputs) and runs the epilogue.  (A single-kernel version with a
//     ticket counter + __threadfence per workgroup was measured 5x slower: every fence is an L2 write-back.  Round 4 tried it
//     again without fences -- write-through (sc1) partial tiles, `s_waitcnt vmcnt(0)`, a relaxed per-tile ticket, the last
//     workgroup of a tile reading the S slices back with sc1 loads: bit-identical, but the streaming hop went from p50 1.24 to
//     1.69 ms: one workgroup pulling S x 16 KB past the L2 costs more than the dependent launch it saves.  Two launches it is.)
//   * the epilogue is the tiled kernel's: bias, Snake, activation, residual, pre-activated second output.
#include "conv1d_mfma.h"

namespace fac {

constexpr int SK_CO = 128;           // output channels per tile
constexpr int SK_MAX_COLS = 640;
constexpr int SK_MAX_S = 32;

struct SkinnyGeom {
  int S;            // slices of the reduction
  int rows;         // (ci-pair, tap) rows in total
  int rows_per_slice;
  int n_cb;         // 32-column blocks
  int co_tiles;
};

// element e of a 128 x 32 tile (e = m*1024 + r*64 + lane: accumulator m, MFMA register r, lane) -> output
__device__ __forceinline__ void skinny_emit(const ConvArgs& a, int e, float v, int co0, int cb, int phase, int ncol) {
  const int ln = e & 63, ri = (e >> 6) & 15, m = e >> 10;
  const int co = co0 + 4 * ((ri & 3) + 8 * (ri >> 2) + 4 * (ln >> 5)) + m;
  const int c2 = cb * 32 + (ln & 31);
  if (co >= a.C_out || c2 >= ncol) return;
  const int b2 = c2 / a.T_out, t2 = c2 - b2 * a.T_out;
  v += a.bias ? a.bias[co] : 0.f;
  if (a.alpha_out) { const float al = a.alpha_out[co]; v = snake_apply(v, al, snake_inv(al)); }
  if (a.act != FAC_ACT_NONE) v = apply_act_slow(v, a.act);
  const long long o = (long long)b2 * a.y_bs + (long long)co * a.y_cs + (long long)t2 * a.y_tstride + phase;
  if (a.res) v += a.res[o];
  if (a.y) a.y[o] = v;
  if (a.y2) { const float al2 = a.alpha2[co]; a.y2[o] = snake_apply(v, al2, snake_inv(al2)); }
}

__global__ __launch_bounds__(256) void conv1d_skinny_kernel(ConvArgs a, SkinnyGeom g, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][4 m][16 r][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kk = lane >> 5;
  const int co0 = blockIdx.x * SK_CO;
  const int slice = blockIdx.y;
  const int cb = blockIdx.z % g.n_cb, phase = blockIdx.z / g.n_cb;
  const int K = a.K, CP = a.C_out_pad;
  const int ncol = a.B * a.T_out;

  // this lane's output column (b, t) as seen by the B operand
  const int col = cb * 32 + l31;
  const bool col_ok = col < ncol;
  const int b = col_ok ? col / a.T_out : 0;
  const int t = col_ok ? col - b * a.T_out : 0;
  const float* xb = a.x + (long long)b * a.x_bs;
  const int tin_base = t * a.stride - a.pad_left;
  const bool reflect = a.pad_mode == FAC_PAD_REFLECT;

  const float* wg = a.w + (long long)phase * cin_pad_dev(a.C_in) * K * CP + co0 + 4 * l31;
  const bool w_ok = co0 + 4 * l31 < CP;

  // rows of this wave: contiguous quarter of the slice
  const int r_lo = slice * g.rows_per_slice;
  const int r_hi = min(g.rows, r_lo + g.rows_per_slice);
  const int per_wave = (r_hi - r_lo + 3) / 4;
  int r = r_lo + wave * per_wave;
  const int r_end = min(r_hi, r + per_wave);

  f32x16 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;

  int jp = r / K, k = r - jp * K;
  constexpr int U = 8;
  while (r < r_end) {
    float4 av[U];
    float bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = r + u < r_end;
      const int ci = 2 * jp + kk;
      av[u] = (ok && w_ok) ? *reinterpret_cast<const float4*>(wg + ((long long)ci * K + k) * CP)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
      int tin = tin_base + k * a.dil;
      if (reflect) tin = reflect_index(tin, a.T_in, a.T_ext);
      bv[u] = (ok && col_ok && ci < a.C_in && tin >= 0 && tin < a.T_in) ? xb[(long long)ci * a.x_cs + tin] : 0.f;
      if (++k == K) { k = 0; ++jp; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u], acc[3], 0, 0, 0);
    }
    r += U;
  }

  // ---- waves -> one tile (fixed order)
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) red[((wave * 4 + m) * 16 + i) * 64 + lane] = acc[m][i];
  __syncthreads();
  const int tile = blockIdx.z * gridDim.x + blockIdx.x;
  float4* mine = reinterpret_cast<float4*>(part) + ((long long)tile * g.S + slice) * 1024;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e4 = tid + 256 * j;
    const float4 p0 = reinterpret_cast<const float4*>(red)[e4], p1 = reinterpret_cast<const float4*>(red)[1024 + e4];
    const float4 p2 = reinterpret_cast<const float4*>(red)[2048 + e4], p3 = reinterpret_cast<const float4*>(red)[3072 + e4];
    float4 sum;
    sum.x = ((p0.x + p1.x) + p2.x) + p3.x;
    sum.y = ((p0.y + p1.y) + p2.y) + p3.y;
    sum.z = ((p0.z + p1.z) + p2.z) + p3.z;
    sum.w = ((p0.w + p1.w) + p2.w) + p3.w;
    if (g.S > 1) {
      mine[e4] = sum;
    } else {
      skinny_emit(a, 4 * e4 + 0, sum.x, co0, cb, phase, ncol);
      skinny_emit(a, 4 * e4 + 1, sum.y, co0, cb, phase, ncol);
      skinny_emit(a, 4 * e4 + 2, sum.z, co0, cb, phase, ncol);
      skinny_emit(a, 4 * e4 + 3, sum.w, co0, cb, phase, ncol);
    }
  }
}

// S partial tiles -> output: one workgroup per quarter tile, every thread issues its S float4 loads at
// once (a single memory round trip) and adds them in slice order.
__global__ __launch_bounds__(256) void conv1d_skinny_reduce_kernel(ConvArgs a, SkinnyGeom g,
                                                                   const float* __restrict__ part) {
  const int tile = blockIdx.x >> 2, quarter = blockIdx.x & 3;
  const int zi = tile / g.co_tiles;
  const int co0 = (tile - zi * g.co_tiles) * SK_CO;
  const int cb = zi % g.n_cb, phase = zi / g.n_cb;
  const int ncol = a.B * a.T_out;
  const int e4 = quarter * 256 + threadIdx.x;
  const float4* base = reinterpret_cast<const float4*>(part) + (long long)tile * g.S * 1024 + e4;
  float4 pv[SK_MAX_S];
#pragma unroll
  for (int s = 0; s < SK_MAX_S; ++s) pv[s] = s < g.S ? base[(long long)s * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sum = pv[0];
#pragma unroll
  for (int s = 1; s < SK_MAX_S; ++s) {
    if (s < g.S) { sum.x += pv[s].x; sum.y += pv[s].y; sum.z += pv[s].z; sum.w += pv[s].w; }
  }
  skinny_emit(a, 4 * e4 + 0, sum.x, co0, cb, phase, ncol);
  skinny_emit(a, 4 * e4 + 1, sum.y, co0, cb, phase, ncol);
  skinny_emit(a, 4 * e4 + 2, sum.z, co0, cb, phase, ncol);
  skinny_emit(a, 4 * e4 + 3, sum.w, co0, cb, phase, ncol);
}

static SkinnyGeom skinny_geom(const ConvArgs& a, int* n_tiles) {
  SkinnyGeom g;
  g.rows = (cin_pad_dev(a.C_in) / 2) * a.K;
  const int ncol = a.B * a.T_out;
  g.n_cb = (ncol + 31) / 32;
  g.co_tiles = (a.C_out + SK_CO - 1) / SK_CO;
  const int tiles = g.co_tiles * g.n_cb * a.n_phase;
  int S = (512 + tiles - 1) / tiles;                      // ~2 workgroups per CU
  if (tiles >= 128) S = 1;                                // enough tiles already: skip the reduce kernel
  const int max_s = g.rows / 32 > 0 ? g.rows / 32 : 1;    // >= 8 rows per wave
  if (S > max_s) S = max_s;
  if (S > SK_MAX_S) S = SK_MAX_S;
  if (S < 1) S = 1;
  g.rows_per_slice = (g.rows + S - 1) / S;
  g.S = (g.rows + g.rows_per_slice - 1) / g.rows_per_slice;
  *n_tiles = tiles;
  return g;
}

bool conv_skinny_ok(const ConvArgs& a, const void* ws, long long ws_bytes) {
  if (!ws || a.alpha_in || a.w1 || a.w_batched || a.phase_shift != 0) return false;
  const long long ncol = (long long)a.B * a.T_out;
  if (ncol > SK_MAX_COLS) return false;
  const long long rows = (long long)(cin_pad_dev(a.C_in) / 2) * a.K;
  if (rows < 24) return false;                          // too little to split: the tiled kernel is fine
  int tiles;
  const SkinnyGeom g = skinny_geom(a, &tiles);
  return ws_bytes >= (long long)tiles * g.S * 16384;
}

int conv_dispatch_skinny(ConvArgs& a, void* ws, long long ws_bytes, hipStream_t s) {
  int tiles;
  const SkinnyGeom g = skinny_geom(a, &tiles);
  (void)ws_bytes;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_skinny_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_set = true;
  }
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(conv1d_skinny_kernel, dim3(g.co_tiles, g.S, g.n_cb * a.n_phase), dim3(256), 65536, s, a, g, part);
  if (g.S > 1)
    hipLaunchKernelGGL(conv1d_skinny_reduce_kernel, dim3(tiles * 4), dim3(256), 0, s, a, g, part);
  return check_launch("conv1d_skinny");
}

}  // namespace fac
