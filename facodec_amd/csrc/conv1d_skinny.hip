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
//   * the epilogue is the tiled kernel's: bias, Snake, activation, residual, pre-activated second output -- plus the two
//     WaveNet epilogues of the streaming hop (round 6: a hop is ~170 dependent launches on its longer chain, so every
//     elementwise launch folded into a reduction kernel is ~6 us of the 1.2 ms): FAC_ACT_GATE (tanh x sigmoid of the two
//     channel halves, modules/commons.py:113-120) and FAC_ACT_WN_RES_SKIP (modules/wavenet.py:159-165).
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
  if (a.act == FAC_ACT_WN_RES_SKIP) {        // first half: x += rs (res = x, may alias y); second half: out (= y2) += rs
    const int half = a.C_out >> 1;
    if (co < half) {
      const long long o = (long long)b2 * a.y_bs + (long long)co * a.y_cs + t2;
      a.y[o] = __fadd_rn(a.res[o], v);
    } else {
      const long long o = (long long)b2 * a.y_bs + (long long)(co - half) * a.y_cs + t2;
      a.y2[o] = __fadd_rn(a.y2[o], v);
    }
    return;
  }
  if (a.alpha_out) { const float al = a.alpha_out[co]; v = snake_apply(v, al, snake_inv(al)); }
  if (a.act != FAC_ACT_NONE) v = apply_act_slow(v, a.act);
  const long long o = (long long)b2 * a.y_bs + (long long)co * a.y_cs + (long long)t2 * a.y_tstride + phase;
  if (a.res) v += a.res[o];
  if (a.y) a.y[o] = v;
  if (a.y2) { const float al2 = a.alpha2[co]; a.y2[o] = snake_apply(v, al2, snake_inv(al2)); }
}

// U = weight rows a wave has in flight per round trip (one float4 + one B value per lane each)
template <int U>
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

// FAC_ACT_GATE: channel co of the first half of the output channels and co + C_out / 2 live in two tiles; one thread adds the
// S partial sums of both (slice order, as above) and writes tanh(a) * sigmoid(b) -- gate_kernel's arithmetic (misc.hip) on the
// values the plain reduction would have written.  C_out / 2 is a multiple of the 128-row tile, S > 1.
__global__ __launch_bounds__(256) void conv1d_skinny_reduce_gate_kernel(ConvArgs a, SkinnyGeom g, const float* __restrict__ part) {
  const int half_tiles = g.co_tiles >> 1;
  const int pair = blockIdx.x >> 2, quarter = blockIdx.x & 3;
  const int zi = pair / half_tiles;
  const int ct = pair - zi * half_tiles;
  const int co0 = ct * SK_CO;
  const int cb = zi % g.n_cb;
  const int ncol = a.B * a.T_out;
  const int e4 = quarter * 256 + threadIdx.x;
  const int half = a.C_out >> 1;
  float4 sums[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tile = zi * g.co_tiles + ct + h * half_tiles;
    const float4* base = reinterpret_cast<const float4*>(part) + (long long)tile * g.S * 1024 + e4;
    float4 pv[SK_MAX_S];
#pragma unroll
    for (int s = 0; s < SK_MAX_S; ++s) pv[s] = s < g.S ? base[(long long)s * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sum = pv[0];
#pragma unroll
    for (int s = 1; s < SK_MAX_S; ++s) {
      if (s < g.S) { sum.x += pv[s].x; sum.y += pv[s].y; sum.z += pv[s].z; sum.w += pv[s].w; }
    }
    sums[h] = sum;
  }
  const float va[4] = {sums[0].x, sums[0].y, sums[0].z, sums[0].w}, vb[4] = {sums[1].x, sums[1].y, sums[1].z, sums[1].w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = 4 * e4 + j;
    const int ln = e & 63, ri = (e >> 6) & 15, m = e >> 10;
    const int co = co0 + 4 * ((ri & 3) + 8 * (ri >> 2) + 4 * (ln >> 5)) + m;
    const int c2 = cb * 32 + (ln & 31);
    if (co >= half || c2 >= ncol) continue;
    const int b2 = c2 / a.T_out, t2 = c2 - b2 * a.T_out;
    const float ta = va[j] + (a.bias ? a.bias[co] : 0.f), sa = vb[j] + (a.bias ? a.bias[co + half] : 0.f);
    a.y[(long long)b2 * a.y_bs + (long long)co * a.y_cs + t2] = __fmul_rn(tanhf(ta), sigmoid_f(sa));
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// One to four output columns and a small weight matrix (the frame-rate layers of a streaming hop at B = 1: the prosody WaveNet,
// the mel projections -- tools/tune/hop_layers.py): ONE launch.  The two-kernel scheme above costs such a conv 10 - 13 us of the
// hop's graph whatever its size (dependent launch + the partial tiles' round trip through memory), a single small kernel ~6 us.
// Output channels are split across workgroups (8 per workgroup: two float4 of the packed (row, co) weights), every workgroup walks
// ALL (ci, tap) rows, one row per thread and step, VALU FMAs against the <= 4 column values; the 256 threads' 8 x 4 partial sums
// meet in LDS in a fixed order (32-thread groups, then the 8 groups).  The layouts the 32-column MFMA tile wastes (1 - 2 live
// columns of 32) do not exist here.  Epilogue as above, FAC_ACT_GATE included (the workgroup then owns channels c .. c + 3 of BOTH halves).
// CO = 8 output channels per workgroup (two float4 per weight row), or 4 where 8 would leave fewer than 256 workgroups walking
// more than ~100 KB of weights each (one workgroup streams at ~1/100 of the chip's bandwidth).
template <int CO>
__global__ __launch_bounds__(256) void conv1d_gemv_kernel(ConvArgs a) {
  constexpr int GV_CO = CO;
  __shared__ float red[256 * 33];
  __shared__ float fin[8][32];
  __shared__ float tot[32];
  const int tid = threadIdx.x;
  const int K = a.K, CP = a.C_out_pad;
  const int ncol = a.B * a.T_out;
  const bool gate = a.act == FAC_ACT_GATE;
  const int half = a.C_out >> 1;
  // workgroups i, i + 8, i + 16 ... share an XCD (and its L2): give them NEIGHBOURING channel groups, so that the 128-byte lines of a
  // weight row (four groups of 32 bytes) are fetched into one L2 instead of four
  const int nb = gridDim.x, per = nb >> 3;
  const int wg = (nb & 7) == 0 ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
  const int coA = gate ? 4 * wg : GV_CO * wg;
  const int coB = gate ? half + 4 * wg : coA + 4;
  const bool okA = coA < CP, okB = (CO == 8 || gate) && coB < CP;
  const int rows = a.C_in * K;
  const bool reflect = a.pad_mode == FAC_PAD_REFLECT;
  const int phase = blockIdx.y;        // polyphase ConvTranspose1d: weights of phase p, outputs at t * y_tstride + p
  const float* wbase = a.w + (long long)phase * cin_pad_dev(a.C_in) * K * CP;
  // column c -> (b, first input index)
  int xoff[4], tin0[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int cc = c < ncol ? c : 0;
    const int b = cc / a.T_out, t = cc - b * a.T_out;
    xoff[c] = b;                     // batch index (the offset itself may exceed 32 bits)
    tin0[c] = t * a.stride - a.pad_left;
  }
  float acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = 0.f;
#pragma unroll 4
  for (int r = tid; r < rows; r += 256) {
    const int ci = r / K, k = r - ci * K;
    const float* wr = wbase + (long long)r * CP;
    const float4 wa = okA ? *reinterpret_cast<const float4*>(wr + coA) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 wb = okB ? *reinterpret_cast<const float4*>(wr + coB) : make_float4(0.f, 0.f, 0.f, 0.f);
    float xv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int tin = tin0[c] + k * a.dil;
      if (reflect) tin = reflect_index(tin, a.T_in, a.T_ext);
      xv[c] = (c < ncol && tin >= 0 && tin < a.T_in) ? a.x[(long long)xoff[c] * a.x_bs + (long long)ci * a.x_cs + tin] : 0.f;
    }
    const float w8[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = fmaf(w8[i], xv[c], acc[i][c]);
  }
  // ---- 256 threads x 32 values -> 32 values, fixed order
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) red[tid * 33 + i * 4 + c] = acc[i][c];
  __syncthreads();
  {
    const int v = tid & 31, grp = tid >> 5;
    float sum = 0.f;
    for (int j = 0; j < 32; ++j) sum += red[(grp * 32 + j) * 33 + v];
    fin[grp][v] = sum;
  }
  __syncthreads();
  if (tid < 32) {
    float sum = fin[0][tid];
#pragma unroll
    for (int gq = 1; gq < 8; ++gq) sum += fin[gq][tid];
    const int i = tid >> 2;
    const int co = i < 4 ? coA + i : coB + (i - 4);
    tot[tid] = sum + ((a.bias && co < a.C_out && (i < 4 || okB)) ? a.bias[co] : 0.f);
  }
  __syncthreads();
  if (tid >= 32) return;
  const int i = tid >> 2, c = tid & 3;
  if (c >= ncol || (i >= 4 && !okB)) return;
  const int co = i < 4 ? coA + i : coB + (i - 4);
  if (co >= a.C_out) return;
  const int b2 = c / a.T_out, t2 = c - b2 * a.T_out;
  float v = tot[tid];
  if (gate) {
    if (i >= 4) return;
    a.y[(long long)b2 * a.y_bs + (long long)co * a.y_cs + t2] = __fmul_rn(tanhf(v), sigmoid_f(tot[tid + 16]));
    return;
  }
  const long long o = (long long)b2 * a.y_bs + (long long)co * a.y_cs + (long long)t2 * a.y_tstride + phase;
  if (a.act == FAC_ACT_WN_RES_SKIP) {
    if (co < half) {
      const long long o = (long long)b2 * a.y_bs + (long long)co * a.y_cs + t2;
      a.y[o] = __fadd_rn(a.res[o], v);
    } else {
      const long long o = (long long)b2 * a.y_bs + (long long)(co - half) * a.y_cs + t2;
      a.y2[o] = __fadd_rn(a.y2[o], v);
    }
    return;
  }
  if (a.alpha_out) { const float al = a.alpha_out[co]; v = snake_apply(v, al, snake_inv(al)); }
  if (a.act != FAC_ACT_NONE) v = apply_act_slow(v, a.act);
  if (a.res) v += a.res[o];
  if (a.y) a.y[o] = v;
  if (a.y2) { const float al2 = a.alpha2[co]; a.y2[o] = snake_apply(v, al2, snake_inv(al2)); }
}

static int skinny_env(const char* name, int dflt);
// 8 or 4 channels per workgroup, or 0: not a shape for this kernel
static int gemv_co(const ConvArgs& a) {
  const bool on = skinny_env("FAC_GEMV", 1) != 0;            // (read per launch: tools/tune/stream_ab_inproc.py flips it between sessions)
  static const long long max_wg = (long long)skinny_env("FAC_GEMV_MAX_WG_KB", 100) * 1024;   // weights one workgroup may walk
  if (!on || (long long)a.B * a.T_out > 4 || a.phase_shift != 0) return 0;
  if (a.act == FAC_ACT_GATE || a.act == FAC_ACT_WN_RES_SKIP) {
    if (a.n_phase != 1 || a.y_tstride != 1) return 0;
    if (a.act == FAC_ACT_GATE) return a.C_out % 8 == 0 && (long long)a.C_in * a.K * 32 <= max_wg ? 8 : 0;
  }
  const long long row_bytes8 = (long long)a.C_in * a.K * 32;
  if (row_bytes8 <= max_wg * 3 / 4 || (long long)((a.C_out + 7) / 8) * a.n_phase >= 256)
    return row_bytes8 <= max_wg ? 8 : (row_bytes8 / 2 <= max_wg ? 4 : 0);
  return row_bytes8 / 2 <= max_wg ? 4 : 0;
}
static bool gemv_ok(const ConvArgs& a) { return gemv_co(a) != 0; }

// tuning knobs (tools/tune/skinny_probe.py): workgroups aimed at, fewest (ci-pair, tap) rows per slice, rows in flight per wave
static int skinny_env(const char* name, int dflt) {   // (declared above for gemv_ok)
  const char* v = getenv(name);
  return v && v[0] ? atoi(v) : dflt;
}
static const int kSkinnyWgs = skinny_env("FAC_SKINNY_WGS", 512);
static const int kSkinnyMinRows = skinny_env("FAC_SKINNY_MIN_ROWS", 32);
static const int kSkinnyU = skinny_env("FAC_SKINNY_U", 8);

static SkinnyGeom skinny_geom(const ConvArgs& a, int* n_tiles) {
  SkinnyGeom g;
  g.rows = (cin_pad_dev(a.C_in) / 2) * a.K;
  const int ncol = a.B * a.T_out;
  g.n_cb = (ncol + 31) / 32;
  g.co_tiles = (a.C_out + SK_CO - 1) / SK_CO;
  const int tiles = g.co_tiles * g.n_cb * a.n_phase;
  int S = (kSkinnyWgs + tiles - 1) / tiles;               // ~2 workgroups per CU
  if (tiles >= 128) S = 1;                                // enough tiles already: skip the reduce kernel
  // >= 8 rows per wave; >= 4 for the short reductions (k = 1 with <= 512 channels: otherwise one workgroup per tile, a handful of
  // workgroups on the chip and the whole epilogue in them -- 17 -> 10 us for the 64- / 96-channel ResidualUnit tails of a hop)
  const int min_rows = g.rows <= 256 ? (kSkinnyMinRows < 16 ? kSkinnyMinRows : 16) : kSkinnyMinRows;
  const int max_s = g.rows / min_rows > 0 ? g.rows / min_rows : 1;
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
  if (a.act == FAC_ACT_GATE && (((g.S < 2 || a.C_out % (2 * SK_CO) != 0) && !gemv_ok(a)) || a.n_phase != 1 || a.y_tstride != 1 ||
                                a.alpha_out || a.res || a.y2 || !a.y)) return false;
  if (a.act == FAC_ACT_WN_RES_SKIP && (a.C_out % 2 != 0 || a.n_phase != 1 || a.y_tstride != 1 || a.alpha_out || !a.res || !a.y2 || !a.y))
    return false;
  return ws_bytes >= (long long)tiles * g.S * 16384;
}

int conv_dispatch_skinny(ConvArgs& a, void* ws, long long ws_bytes, hipStream_t s) {
  if (const int co = gemv_co(a)) {
    const int wgs = a.act == FAC_ACT_GATE ? (a.C_out / 2 + 3) / 4 : (a.C_out + co - 1) / co;
    if (co == 8) hipLaunchKernelGGL(conv1d_gemv_kernel<8>, dim3(wgs, a.n_phase), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(conv1d_gemv_kernel<4>, dim3(wgs, a.n_phase), dim3(256), 0, s, a);
    return check_launch("conv1d_gemv");
  }
  int tiles;
  const SkinnyGeom g = skinny_geom(a, &tiles);
  (void)ws_bytes;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_skinny_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_skinny_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_skinny_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_set = true;
  }
  float* part = reinterpret_cast<float*>(ws);
  const dim3 grid(g.co_tiles, g.S, g.n_cb * a.n_phase);
  const int per_wave = (g.rows_per_slice + 3) / 4;
  if (kSkinnyU >= 32 && per_wave > 16)
    hipLaunchKernelGGL(conv1d_skinny_kernel<32>, grid, dim3(256), 65536, s, a, g, part);
  else if (kSkinnyU >= 16 && per_wave > 8)
    hipLaunchKernelGGL(conv1d_skinny_kernel<16>, grid, dim3(256), 65536, s, a, g, part);
  else
    hipLaunchKernelGGL(conv1d_skinny_kernel<8>, grid, dim3(256), 65536, s, a, g, part);
  if (a.act == FAC_ACT_GATE)
    hipLaunchKernelGGL(conv1d_skinny_reduce_gate_kernel, dim3(tiles * 2), dim3(256), 0, s, a, g, part);
  else if (g.S > 1)
    hipLaunchKernelGGL(conv1d_skinny_reduce_kernel, dim3(tiles * 4), dim3(256), 0, s, a, g, part);
  return check_launch("conv1d_skinny");
}

}  // namespace fac
