// Weight gradient of the conv stack on the bf16 matrix pipe, fp32-exact (the same 3-way operand split as
// conv1d_bsplit.hip: six v_mfma_f32_32x32x16_bf16 per K = 16 step, fp32 accumulate, smallest terms first).
//
//   dW[co][ci][k] = sum_{b,t} dy[b][co][t] * xpad[b][ci][t*stride + k*dil]        (torch autograd of F.conv1d behind
//                                                                                  dac/model/encodec.py:212-228)
// GEMM view: M = C_out (A = dy), N = (ci, k) columns (B = shifted x), contraction over time.  The MFMA contracts 16
// consecutive time steps, so a B fragment is 8 consecutive bf16 of one input row starting at element t + k*dil -- an
// arbitrary, mostly unaligned offset.  Measured on MI355X (tools/microbench/lds_unaligned_probe.hip): ds_read_b128 at
// any address that is not 16-byte aligned costs 8x; so the staged input rows are kept in up to FOUR copies shifted by
// 0..3 elements and a fragment is read as two 8-byte-aligned ds_read_b64 from the copy (shift mod 4) -- the LDS cycles
// of one b128, no VALU work in the MFMA waves, any (K, dilation, stride) through per-lane base offsets computed once.
// Strided convs (K = 2*stride and the discriminators' k = 5 / stride 3) stage the input phase-major
// (row (ci, phase)[u] = x[ci][u*stride + phase]) so that consecutive output steps are consecutive elements again.
//
// Workgroup = 8 waves on a 128 (co) x 128 (ci,k columns) tile of dW for one slice of the (b, t) range:
//   waves 4-7 stage 32 time steps per stage: fp32 loads -> split3 -> A [plane][128 co][32 t], B [copy][plane][row][..];
//   waves 0-3 each own 64 x 64 (2 x 2 MFMA blocks), two K = 16 steps per stage; LDS double-buffered.
// Partial tiles of the S slices are added in slice order by wgrad_reduce_kernel (deterministic, as before).
#include "conv1d_mfma.h"

namespace fac {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int WS_CO = 128;          // output channels per tile (A rows)
constexpr int WS_NC = 128;          // (ci, k) columns per tile
constexpr int WS_TT = 32;           // time steps per stage
constexpr int WS_APB = 80;          // A row pitch in bytes: 32 bf16 + 16 B pad -> conflict-free 16-lane b128 groups
constexpr int WS_A_PLANE = WS_CO * WS_APB;
constexpr int WS_A_STAGE = 3 * WS_A_PLANE;
constexpr int WS_ITEMS = 4;         // staging work items per lane and operand (256 staging lanes)

struct WsArgs {
  const float* x;      // (B, C_in, T_in)
  const float* dy;     // (B, C_out, T_out)
  float* part;         // [S][C_out][C_in][K]
  long long x_bs, dy_bs;
  int x_cs, dy_cs;
  int B, C_in, T_in, T_ext, C_out, T_out, K, stride, dil, pad_left, pad_mode;
  int cit;             // input channels per column tile (cit * K <= 128)
  int R;               // staged input rows per stage = cit * stride (row = (channel, phase))
  int NCP;             // shifted copies of the staged rows (1..4)
  int nq;              // 4-element quads per staged row
  int XPB;             // staged row pitch in bytes (multiple of 8)
  int n_tt;            // 32-step time tiles per clip
  int tiles_per_split;
};

struct __attribute__((packed, aligned(4))) F4u { float v[4]; };   // 4-byte-aligned 16-byte global load

__device__ __forceinline__ void split3w(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

__global__ __launch_bounds__(512, 2) void conv1d_wgrad_split_kernel(WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co0 = blockIdx.x * WS_CO;
  const int ci0 = blockIdx.y * a.cit;
  const int z = blockIdx.z;
  const int PSB = a.R * a.XPB;                  // bytes of one plane of one copy
  const int B_STAGE = a.NCP * 3 * PSB;
  const int STAGE = (WS_A_STAGE + B_STAGE + 15) & ~15;
  const int tile_lo = z * a.tiles_per_split;
  const int tile_hi = min(a.B * a.n_tt, tile_lo + a.tiles_per_split);
  const int n_chunks = tile_hi - tile_lo;
  const int s = a.stride;

  if (wave >= 4) {
    // ======================================================================= staging waves
    const int sl = tid - 256;
    __builtin_amdgcn_s_setprio(3);
    // B work items (row, quad): constant over the stages
    int b_xoff[WS_ITEMS], b_u0[WS_ITEMS], b_lds[WS_ITEMS], b_ph[WS_ITEMS];
    const int n_items = a.R * a.nq;
#pragma unroll
    for (int j = 0; j < WS_ITEMS; ++j) {
      const int id = sl + 256 * j;
      const bool ok = id < n_items;
      const int row = ok ? id / a.nq : 0;
      const int q = id - row * a.nq;
      const int cl = row / s, ph = row - cl * s;
      const int ci = ci0 + cl;
      b_xoff[j] = (ok && ci < a.C_in) ? ci * a.x_cs : -1;
      b_ph[j] = ph;
      b_u0[j] = 4 * q;
      b_lds[j] = row * a.XPB + q * 8;
    }
    const int n_need = 3 + a.NCP;          // consecutive staged elements one item touches (4 + NCP - 1)

    auto load_tile = [&](int chunk, float (&ra)[WS_ITEMS][4], float (&rb)[WS_ITEMS][8]) {
      const int tile = tile_lo + chunk;
      const int b = tile / a.n_tt;
      const int t0 = (tile - b * a.n_tt) * WS_TT;
      const float* dyb = a.dy + (long long)b * a.dy_bs;
      const float* xb = a.x + (long long)b * a.x_bs;
#pragma unroll
      for (int j = 0; j < WS_ITEMS; ++j) {          // A: dy[co0 + row][t0 + 4q .. +3]
        const int id = sl + 256 * j;
        const int row = id >> 3, q = id & 7;
        const int co = co0 + row, t = t0 + 4 * q;
        if (co < a.C_out && t + 3 < a.T_out) {
          const F4u v = *reinterpret_cast<const F4u*>(dyb + co * a.dy_cs + t);
#pragma unroll
          for (int i = 0; i < 4; ++i) ra[j][i] = v.v[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) ra[j][i] = (co < a.C_out && t + i < a.T_out) ? dyb[co * a.dy_cs + t + i] : 0.f;
        }
      }
      const int tb = t0 * s - a.pad_left;            // input position of staged element u = 0, phase 0
#pragma unroll
      for (int j = 0; j < WS_ITEMS; ++j) {          // B: row (ci, phase), elements u0 .. u0 + n_need - 1
        if (b_xoff[j] < 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) rb[j][i] = 0.f;
          continue;
        }
        const float* xr = xb + b_xoff[j];
        const int p0 = tb + b_u0[j] * s + b_ph[j];
        if (s == 1 && p0 >= 0 && p0 + 7 < a.T_in) {
          const F4u v0 = *reinterpret_cast<const F4u*>(xr + p0);
          const F4u v1 = *reinterpret_cast<const F4u*>(xr + p0 + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { rb[j][i] = v0.v[i]; rb[j][4 + i] = v1.v[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float v = 0.f;
            if (i < n_need) {
              const int tin = p0 + i * s;
              int idx;
              if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
              else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
              if (idx >= 0) v = xr[idx];
            }
            rb[j][i] = v;
          }
        }
      }
    };

    auto write_tile = [&](int buf, const float (&ra)[WS_ITEMS][4], const float (&rb)[WS_ITEMS][8]) {
      unsigned char* Ab = sm + buf * STAGE;
      unsigned char* Bb = Ab + WS_A_STAGE;
#pragma unroll
      for (int j = 0; j < WS_ITEMS; ++j) {
        const int id = sl + 256 * j;
        const int row = id >> 3, q = id & 7;
        bf16x4 h, m, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __bf16 p0, p1, p2;
          split3w(ra[j][i], p0, p1, p2);
          h[i] = p0; m[i] = p1; l[i] = p2;
        }
        unsigned char* d = Ab + row * WS_APB + q * 8;
        *reinterpret_cast<bf16x4*>(d) = h;
        *reinterpret_cast<bf16x4*>(d + WS_A_PLANE) = m;
        *reinterpret_cast<bf16x4*>(d + 2 * WS_A_PLANE) = l;
      }
#pragma unroll
      for (int j = 0; j < WS_ITEMS; ++j) {
        if (sl + 256 * j >= n_items) continue;
        __bf16 ph_[7], pm_[7], pl_[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) split3w(rb[j][i], ph_[i], pm_[i], pl_[i]);
        unsigned char* d = Bb + b_lds[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r >= a.NCP) break;
          bf16x4 h, m, l;
#pragma unroll
          for (int i = 0; i < 4; ++i) { h[i] = ph_[r + i]; m[i] = pm_[r + i]; l[i] = pl_[r + i]; }
          unsigned char* dr = d + (r * 3) * PSB;
          *reinterpret_cast<bf16x4*>(dr) = h;
          *reinterpret_cast<bf16x4*>(dr + PSB) = m;
          *reinterpret_cast<bf16x4*>(dr + 2 * PSB) = l;
        }
      }
    };

    // loads of tile c+2 are issued one stage before they are split and written (register double buffer)
    float a0[WS_ITEMS][4], b0[WS_ITEMS][8], a1[WS_ITEMS][4], b1[WS_ITEMS][8];
    load_tile(0, a0, b0);
    if (n_chunks > 1) load_tile(1, a1, b1);
    write_tile(0, a0, b0);
    __syncthreads();
    for (int chunk = 0; chunk < n_chunks; chunk += 2) {
      if (chunk + 1 < n_chunks) {
        if (chunk + 2 < n_chunks) load_tile(chunk + 2, a0, b0);
        write_tile(1, a1, b1);
      }
      __syncthreads();
      if (chunk + 1 >= n_chunks) break;
      if (chunk + 2 < n_chunks) {
        if (chunk + 3 < n_chunks) load_tile(chunk + 3, a1, b1);
        write_tile(0, a0, b0);
      }
      __syncthreads();
    }
    return;
  }

  // ========================================================================= MFMA waves
  const int l31 = lane & 31, kq = lane >> 5;
  const int mh = wave >> 1, nh = wave & 1;
  const int ncol = min(a.cit, a.C_in - ci0) * a.K;
  int boff[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    int J = nh * 64 + n * 32 + l31;
    if (J >= a.cit * a.K) J = 0;                 // padding columns: read column 0, results are never stored
    const int cl = J / a.K, k = J - cl * a.K;
    const int kd = k * a.dil;
    const int shift = kd / s, ph = kd - shift * s;
    const int r = shift & 3;                     // NCP < 4 only when every shift is < NCP
    boff[n] = WS_A_STAGE + (r * 3) * PSB + (cl * s + ph) * a.XPB + (shift - r) * 2 + kq * 16;
  }
  const int aoff = (mh * 64 + l31) * WS_APB + kq * 16;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  __syncthreads();   // tile 0 staged
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char* st = sm + (chunk & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < WS_TT / 16; ++ks) {
      bf16x8 A[2][3], Bf[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          A[m][p] = *reinterpret_cast<const bf16x8*>(st + aoff + p * WS_A_PLANE + m * 32 * WS_APB + ks * 32);
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const unsigned char* bp = st + boff[n] + p * PSB + ks * 32;
          const bf16x4 lo = *reinterpret_cast<const bf16x4*>(bp);
          const bf16x4 hi = *reinterpret_cast<const bf16x4*>(bp + 8);
          Bf[n][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      __builtin_amdgcn_sched_barrier(0);
      // smallest terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi (planes 0 = hi, 1 = mid, 2 = lo); the term
      // loop is outside the block loops so that consecutive MFMAs write different accumulators
      constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][TA[q]], Bf[n][TB[q]], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // partial dW of this slice: row = output channel, columns (ci, k) contiguous
  float* pz = a.part + (long long)z * a.C_out * a.C_in * a.K;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int J = nh * 64 + n * 32 + l31;
      if (J >= ncol) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mh * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        if (co < a.C_out) pz[((long long)co * a.C_in + ci0) * a.K + J] = acc[m][n][r];
      }
    }
}

__global__ void wgrad_split_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float acc = part[i];
    for (int z = 1; z < S; ++z) acc += part[(long long)z * n + i];
    dw[i] = acc;
  }
}

// Geometry shared by the workspace query and the launch.  Returns 0 when the shape runs on the split kernel.
static int ws_geometry(int B, int C_in, int T_in, int C_out, int T_out, int K, int stride, int dil, WsArgs* a, int* splits,
                       size_t* lds) {
  if (K < 1 || K > WS_NC || stride < 1 || dil < 1) return -1;
  a->cit = WS_NC / K;
  a->R = a->cit * stride;
  const int max_shift = ((K - 1) * dil) / stride;
  a->NCP = max_shift + 1 < 4 ? max_shift + 1 : 4;
  const int XW = WS_TT + max_shift;
  a->nq = (XW + 3) / 4;
  a->XPB = 8 * a->nq + 8;
  if ((long long)a->R * a->nq > 256 * WS_ITEMS) return -1;                       // staging items per stage
  const size_t stage = ((size_t)WS_A_STAGE + (size_t)a->NCP * 3 * a->R * a->XPB + 15) & ~(size_t)15;
  *lds = 2 * stage;
  if (*lds > 160 * 1024) return -1;
  if ((long long)C_in * T_in >= (1ll << 31) || (long long)C_out * T_out >= (1ll << 31)) return -1;
  a->n_tt = (T_out + WS_TT - 1) / WS_TT;
  const long long tiles = (long long)B * a->n_tt;
  const long long wgs = (long long)((C_out + WS_CO - 1) / WS_CO) * ((C_in + a->cit - 1) / a->cit);
  // one workgroup per CU (LDS): aim at ~4 waves of workgroups over the 256 CUs; bounded by 512 MB of partials
  long long S = (1024 + wgs - 1) / wgs;
  if (S > tiles) S = tiles;
  if (S > 2048) S = 2048;
  const long long per_split_bytes = (long long)C_out * C_in * K * 4;
  if (S * per_split_bytes > (512ll << 20)) S = (512ll << 20) / per_split_bytes;
  if (S < 1) S = 1;
  a->tiles_per_split = (int)((tiles + S - 1) / S);
  *splits = (int)((tiles + a->tiles_per_split - 1) / a->tiles_per_split);
  return 0;
}

}  // namespace fac

extern "C" int64_t fac_conv1d_bwd_weight_split_ws_bytes(int B, int C_in, int T_in, int C_out, int T_out, int K, int stride,
                                                        int dilation) {
  fac::WsArgs a;
  int S;
  size_t lds;
  if (fac::ws_geometry(B, C_in, T_in, C_out, T_out, K, stride, dilation, &a, &S, &lds)) return -1;
  return (int64_t)S * C_out * C_in * K * 4;
}

extern "C" int fac_conv1d_bwd_weight_split(const float* x, const float* dy, float* dw, void* ws, int64_t ws_bytes, int B,
                                           int C_in, int T_in, int C_out, int T_out, int K, int stride, int dilation,
                                           int pad_left, int pad_mode, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(x && dy && dw && ws && B > 0 && C_in > 0 && C_out > 0 && T_in > 0 && T_out > 0 && K > 0 && stride > 0 &&
                  dilation > 0 && pad_left >= 0,
              "conv1d_bwd_weight_split: bad arguments");
  WsArgs a;
  int S;
  size_t lds;
  FAC_REQUIRE(ws_geometry(B, C_in, T_in, C_out, T_out, K, stride, dilation, &a, &S, &lds) == 0,
              "conv1d_bwd_weight_split: shape not supported (K=%d stride=%d dilation=%d)", K, stride, dilation);
  FAC_REQUIRE(ws_bytes >= (int64_t)S * C_out * C_in * K * 4, "conv1d_bwd_weight_split: workspace too small");
  a.x = x; a.dy = dy; a.part = reinterpret_cast<float*>(ws);
  a.x_bs = (long long)C_in * T_in; a.x_cs = T_in; a.dy_bs = (long long)C_out * T_out; a.dy_cs = T_out;
  a.B = B; a.C_in = C_in; a.T_in = T_in; a.C_out = C_out; a.T_out = T_out; a.K = K; a.stride = stride; a.dil = dilation;
  a.pad_left = pad_left; a.pad_mode = pad_mode;
  {
    long long last = (long long)(T_out - 1) * stride + (long long)(K - 1) * dilation - pad_left;
    int pad_right = last >= T_in ? (int)(last - T_in + 1) : 0;
    int max_pad = pad_left > pad_right ? pad_left : pad_right;
    a.T_ext = T_in > max_pad ? T_in : max_pad + 1;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_wgrad_split_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid((C_out + WS_CO - 1) / WS_CO, (C_in + a.cit - 1) / a.cit, S);
  hipLaunchKernelGGL(conv1d_wgrad_split_kernel, grid, dim3(512), lds, (hipStream_t)stream, a);
  const long long n = (long long)C_out * C_in * K;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a.part, dw, S, n);
  return check_launch("conv1d_bwd_weight_split");
}
