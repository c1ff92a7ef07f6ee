// Skinny conv: few output columns (B * T_out <= 640), many weights -- the streaming-inference regime
// (1-2 frames per hop) and every per-clip Linear of the quantizer.  The op is a GEMV-like
// weight stream: the tiled kernel would put one workgroup per 128 output channels and let it walk all of
// C_in * K alone (measured 60-270 us per launch for ~10 MB of weights = a few % of HBM bandwidth).
//
// Here the reduction dimension is split across workgroups as well:
//   grid = (co tiles of 32) x (S slices of the (ci-pair, tap) rows) x (32-column blocks * phases)
//   * the MFMA A fragment is read straight from the packed (C_in, K, C_out) weights: lane = output channel,
//     a half-wave reads 128 contiguous bytes per (ci, tap) row -- no LDS, weights touched once per column block;
//   * the B operand (x[ci][t*stride + k*dil - pad]) is a few KB, served by L1/L2;
//   * partial 32x32 tiles go to the caller's workspace; the LAST workgroup to arrive at a tile (ticket
//     counter) adds the S partials in slice order -- deterministic, unlike atomics on the outputs -- and runs
//     the same epilogue as the tiled kernel (bias, Snake, activation, residual, pre-activated second output).
#include "conv1d_mfma.h"

namespace fac {

constexpr int SK_CO = 32;            // output channels per tile
constexpr int SK_COUNTERS = 16384;   // ticket counters at the head of the workspace
constexpr int SK_MAX_COLS = 640;
constexpr int SK_MAX_S = 32;

struct SkinnyGeom {
  int S;            // slices of the reduction
  int rows;         // (ci-pair, tap) rows in total
  int rows_per_slice;
  int n_cb;         // 32-column blocks
};

__global__ __launch_bounds__(256) void conv1d_skinny_kernel(ConvArgs a, SkinnyGeom g, float* __restrict__ part,
                                                            unsigned* __restrict__ counters) {
  __shared__ __attribute__((aligned(16))) float red[4][1024];   // [wave][16 r][64 lanes]
  __shared__ unsigned ticket;
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

  const float* wg = a.w + (long long)phase * cin_pad_dev(a.C_in) * K * CP + co0 + l31;
  const bool w_ok = co0 + l31 < CP;

  // rows of this wave: contiguous quarter of the slice
  const int r_lo = slice * g.rows_per_slice;
  const int r_hi = min(g.rows, r_lo + g.rows_per_slice);
  const int per_wave = (r_hi - r_lo + 3) / 4;
  int r = r_lo + wave * per_wave;
  const int r_end = min(r_hi, r + per_wave);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  int jp = r / K, k = r - jp * K;
  constexpr int U = 8;
  while (r < r_end) {
    float av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = r + u < r_end;
      const int ci = 2 * jp + kk;
      av[u] = (ok && w_ok) ? wg[((long long)ci * K + k) * CP] : 0.f;
      int tin = tin_base + k * a.dil;
      if (reflect) tin = reflect_index(tin, a.T_in, a.T_ext);
      bv[u] = (ok && col_ok && ci < a.C_in && tin >= 0 && tin < a.T_in) ? xb[(long long)ci * a.x_cs + tin] : 0.f;
      if (++k == K) { k = 0; ++jp; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    r += U;
  }

  // ---- waves -> one partial tile (fixed order)
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i * 64 + lane] = acc[i];
  __syncthreads();
  const int tile = blockIdx.z * gridDim.x + blockIdx.x;
  float4 sum;
  {
    const float4 p0 = reinterpret_cast<const float4*>(red[0])[tid], p1 = reinterpret_cast<const float4*>(red[1])[tid];
    const float4 p2 = reinterpret_cast<const float4*>(red[2])[tid], p3 = reinterpret_cast<const float4*>(red[3])[tid];
    sum.x = ((p0.x + p1.x) + p2.x) + p3.x;
    sum.y = ((p0.y + p1.y) + p2.y) + p3.y;
    sum.z = ((p0.z + p1.z) + p2.z) + p3.z;
    sum.w = ((p0.w + p1.w) + p2.w) + p3.w;
  }
  if (g.S > 1) {
    float4* base = reinterpret_cast<float4*>(part) + (long long)tile * g.S * 256;
    base[(long long)slice * 256 + tid] = sum;
    __threadfence();
    __syncthreads();
    if (tid == 0) ticket = atomicAdd(&counters[tile], 1u);
    __syncthreads();
    if (ticket != (unsigned)(g.S - 1)) return;
    __threadfence();
    // last arrival: S float4 loads per thread, all in flight at once, summed in slice order
    float4 pv[SK_MAX_S];
#pragma unroll
    for (int s = 0; s < SK_MAX_S; ++s)
      pv[s] = s < g.S ? base[(long long)s * 256 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    sum = pv[0];
#pragma unroll
    for (int s = 1; s < SK_MAX_S; ++s) {
      if (s < g.S) { sum.x += pv[s].x; sum.y += pv[s].y; sum.z += pv[s].z; sum.w += pv[s].w; }
    }
    if (tid == 0) counters[tile] = 0;   // ready for the next launch on this stream
  }

  // ---- epilogue (same order of operations as the tiled kernel's emit_block)
  const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = 4 * tid + q;
    const int ln = e & 63, ri = e >> 6;
    const int co = co0 + (ri & 3) + 8 * (ri >> 2) + 4 * (ln >> 5);
    const int c2 = cb * 32 + (ln & 31);
    if (co >= a.C_out || c2 >= ncol) continue;
    const int b2 = c2 / a.T_out, t2 = c2 - b2 * a.T_out;
    float v = sv[q] + (a.bias ? a.bias[co] : 0.f);
    if (a.alpha_out) { const float al = a.alpha_out[co]; v = snake_apply(v, al, snake_inv(al)); }
    if (a.act != FAC_ACT_NONE) v = apply_act_slow(v, a.act);
    const long long o = (long long)b2 * a.y_bs + (long long)co * a.y_cs + (long long)t2 * a.y_tstride + phase;
    if (a.res) v += a.res[o];
    if (a.y) a.y[o] = v;
    if (a.y2) { const float al2 = a.alpha2[co]; a.y2[o] = snake_apply(v, al2, snake_inv(al2)); }
  }
}

static SkinnyGeom skinny_geom(const ConvArgs& a, int* n_tiles) {
  SkinnyGeom g;
  g.rows = (cin_pad_dev(a.C_in) / 2) * a.K;
  const int ncol = a.B * a.T_out;
  g.n_cb = (ncol + 31) / 32;
  const int co_tiles = (a.C_out + SK_CO - 1) / SK_CO;
  const int tiles = co_tiles * g.n_cb * a.n_phase;
  int S = (512 + tiles - 1) / tiles;                      // ~2 workgroups per CU
  const int max_s = g.rows / 16 > 0 ? g.rows / 16 : 1;    // >= 4 rows per wave
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
  if (rows < 96) return false;                          // too little to split: the tiled kernel is fine
  int tiles;
  const SkinnyGeom g = skinny_geom(a, &tiles);
  return tiles <= SK_COUNTERS && ws_bytes >= (long long)SK_COUNTERS * 4 + (long long)tiles * g.S * 4096;
}

int conv_dispatch_skinny(ConvArgs& a, void* ws, long long ws_bytes, hipStream_t s) {
  int tiles;
  const SkinnyGeom g = skinny_geom(a, &tiles);
  const int co_tiles = (a.C_out + SK_CO - 1) / SK_CO;
  (void)ws_bytes;
  unsigned* counters = reinterpret_cast<unsigned*>(ws);
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + SK_COUNTERS * 4);
  hipLaunchKernelGGL(conv1d_skinny_kernel, dim3(co_tiles, g.S, g.n_cb * a.n_phase), dim3(256), 0, s, a, g, part,
                     counters);
  return check_launch("conv1d_skinny");
}

}  // namespace fac
