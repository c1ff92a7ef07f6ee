// Weight gradient of the conv stack on the bf16 matrix pipe, fp32-grade (the same 3-way operand split as
// conv1d_bsplit.hip: six v_mfma_f32_32x32x16_bf16 per K = 16 step, fp32 accumulate, smallest terms first).
//
//   dW[co][ci][k] = sum_{b,t} dy[b][co][t] * xpad[b][ci][t*stride + k*dil]        (torch autograd of F.conv1d behind
//                                                                                  dac/model/encodec.py:212-228)
// GEMM view: M = C_out (A = dy), N = (ci, k) columns (B = shifted x), contraction over time.
//
// Two launches (+ the deterministic slice reduction):
//  1. split_planes_kernel, once per operand: fp32 -> three bf16 planes in HBM (hi, mid, lo; x = hi + mid + lo exactly), with
//     the conv's padding MATERIALISED (reflect / zero, dac/model/encodec.py:96-113) and, for strided convs, the time axis
//     de-interleaved phase-major (row (ci, phase)[u] = xpad[ci][u*stride + phase]) so that consecutive output steps are
//     consecutive elements.  dy gets a zero tail up to a multiple of 32 steps.  HBM-bound: 4 B read + 6 B written per
//     element.  (Measured before this split existed: with the fp32 -> bf16 splitting done by the GEMM's own staging waves,
//     ~6 VALU instructions per MFMA shared each SIMD's issue port with the matrix pipe and held the kernel at 100-110
//     TFLOP/s-equivalent -- 137 with the arithmetic removed, 147 with no staging at all; every dy tile was also re-split by
//     every column tile, 43 times at C = 768.)
//  2. conv1d_wgrad_planes_kernel: workgroup = 8 waves on a 128 (co) x 128 ((ci,k) columns) tile of dW for one slice of the
//     (b, t) range.  Waves 4-7 only COPY 32 time steps per stage from the planes into LDS (inline-asm loads in a register
//     ring D tiles deep with explicit s_waitcnt -- hipcc's own placement waited for the youngest loads); waves 0-3 each own
//     64 x 64 (2 x 2 MFMA blocks), two K = 16 steps per stage; LDS double-buffered.
//     The MFMA contracts 16 consecutive time steps, so a B fragment is 8 consecutive bf16 of one staged row starting at
//     element t + (k*dil)/stride -- mostly unaligned, and ds_read_b128 at an address that is not 16-byte aligned costs 8x
//     on this part (tools/microbench/lds_unaligned_probe.hip).  Hence up to FOUR copies of the staged rows shifted by 0..3
//     elements: a fragment is two 8-byte-aligned ds_read_b64 from copy (shift mod 4) -- the LDS cycles of one b128, no
//     VALU in the MFMA waves, any (K, dilation, stride) through per-lane base offsets computed once.
// Partial tiles of the S slices are added in slice order by wgrad_split_reduce_kernel (deterministic).
#include "conv1d_mfma.h"
#include <stdlib.h>
#include <type_traits>

namespace fac {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WS_CO = 128;          // output channels per tile (A rows)
constexpr int WS_NC = 128;          // (ci, k) columns per tile
constexpr int WS_TT = 32;           // time steps per stage
constexpr int WS_APB = 80;          // A row pitch in bytes: 32 bf16 + 16 B pad -> conflict-free 16-lane b128 groups
constexpr int WS_A_PLANE = WS_CO * WS_APB;
constexpr int WS_A_STAGE = 3 * WS_A_PLANE;
constexpr int WS_APIECES = 6;       // 16-byte A pieces per staging lane and stage (3 planes x 128 rows x 4 / 256)

struct WsArgs {
  const unsigned char* ap;   // dy planes  [3][B*C_out][UA] bf16
  const unsigned char* bp;   // x planes   [3][B*C_in*stride][UB] bf16 (padded, phase-major)
  float* part;               // [S][C_out][C_in][K]
  long long a_plane_bytes, b_plane_bytes;
  int UA, UB;                // row lengths (elements) of the plane tensors
  int B, C_in, C_out, K, stride, dil;
  // Two-level taps (tap k = k2*K1 + k1 at offset k2*dil2 + k1*dil, dil2 % stride == 0) are handled as K2 "virtual input
  // channels" per real one: virtual channel v = ci*K2 + k2 has K1 taps and reads row ci shifted by k2*dil2/stride staged
  // elements; the dW column order (ci, k2, k1) is unchanged.  Here K = K1 and C_in = C_in_real * K2.
  int K2, dil2s;
  int cit;             // (virtual) input channels per column tile (cit * K <= 128)
  int R;               // staged input rows per stage = cit * stride (row = (channel, phase))
  int NCP;             // shifted copies of the staged rows (1..4)
  int nq;              // 4-element quads per staged row
  int XPB;             // staged row pitch in bytes (multiple of 8)
  int n_tt;            // 32-step time tiles per clip
  int tiles_per_split;
};

__device__ __forceinline__ void split3w(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// dst[p][row * s + ph][u] = plane p of pad(src[row])[u * s + ph - pad_left]  (0 beyond the padded signal), u < U (U % 8 == 0).
// One thread = 8 consecutive u of one destination row: three 16-byte stores.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, long long rows,
                                                           int T, int T_ext, int T_pad, int s, int U, int pad_left, int pad_mode,
                                                           long long plane_bytes) {
  const int u8 = U >> 3;
  const long long n = rows * s * u8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int uq = (int)(i % u8);
    const long long drow = i / u8;
    const long long row = drow / s;
    const int ph = (int)(drow - row * s);
    const float* xr = src + row * T;
    const int p0 = (uq * 8) * s + ph - pad_left;
    float v[8];
    if (s == 1 && p0 >= 0 && p0 + 7 < T) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = xr[p0 + j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pos = p0 + j * s;
        int idx = -1;
        if (pos + pad_left < T_pad) {                 // inside the padded signal (the tail beyond it is zero fill)
          if (pad_mode == FAC_PAD_REFLECT) idx = reflect_index(pos, T, T_ext);
          else idx = (pos >= 0 && pos < T) ? pos : -1;
        }
        v[j] = idx >= 0 ? xr[idx] : 0.f;
      }
    }
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 a, b, c;
      split3w(v[j], a, b, c);
      h[j] = a; m[j] = b; l[j] = c;
    }
    unsigned char* d = dst + (drow * U + uq * 8) * 2;
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + plane_bytes) = m;
    *reinterpret_cast<bf16x8*>(d + 2 * plane_bytes) = l;
  }
}

// The dy operand's split pass with the bias gradient folded in (db[co] = sum over (b, t) of dy: one more pass over dy otherwise):
// grid (chunks of 2048 steps, rows = B * C_out); a workgroup splits its chunk of one row and leaves the chunk's sum in
// part[row][chunk]; bias_from_rowsums_kernel adds them per output channel in a fixed order (clip-major, then chunk).
__global__ __launch_bounds__(256) void split_planes_rowsum_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                                                  float* __restrict__ part, int T, int U, int n_chunks,
                                                                  long long plane_bytes) {
  __shared__ float red[256];
  const long long row = blockIdx.y;
  const int uq = blockIdx.x * blockDim.x + threadIdx.x;     // 8-step piece of the row (blockDim.x = 256, or 64 for rows of <= 512 steps)
  const float* xr = src + row * T;
  float sum = 0.f;
  if (uq * 8 < U) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pos = uq * 8 + j;
      v[j] = pos < T ? xr[pos] : 0.f;
    }
    sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 a, b, c;
      split3w(v[j], a, b, c);
      h[j] = a; m[j] = b; l[j] = c;
    }
    unsigned char* d = dst + (row * U + uq * 8) * 2;
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + plane_bytes) = m;
    *reinterpret_cast<bf16x8*>(d + 2 * plane_bytes) = l;
  }
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[row * n_chunks + blockIdx.x] = red[0];
}

__global__ void bias_from_rowsums_kernel(const float* __restrict__ part, float* __restrict__ db, int B, int C_out, int n_chunks) {
  const int co = blockIdx.x * blockDim.x + threadIdx.x;
  if (co >= C_out) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* p = part + ((long long)b * C_out + co) * n_chunks;
    for (int c = 0; c < n_chunks; ++c) s += p[c];
  }
  db[co] = s;
}

// NB: 8-byte B pieces per staging lane and stage (ceil(NCP * 3 * R * nq / 256));  D: depth of the register ring.
template <int NB, int D>
__global__ __launch_bounds__(512, 2) void conv1d_wgrad_planes_kernel(WsArgs a) {
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
    // ======================================================================= staging waves: planes -> LDS, copies only
    const int sl = tid - 256;
    __builtin_amdgcn_s_setprio(3);
    // Per-lane constants: byte offset of each piece relative to the tile's uniform base pointer, and its LDS address.
    unsigned a_off[WS_APIECES];
    int a_lds[WS_APIECES];
#pragma unroll
    for (int j = 0; j < WS_APIECES; ++j) {
      const int id = sl + 256 * j;
      const int plane = id >> 9, rem = id & 511;
      const int row = rem >> 2, pc = rem & 3;
      const int co = co0 + row < a.C_out ? co0 + row : a.C_out - 1;     // rows past C_out are computed but never stored
      a_off[j] = (unsigned)(plane * a.a_plane_bytes + ((long long)co * a.UA + 8 * pc) * 2);
      a_lds[j] = plane * WS_A_PLANE + row * WS_APB + pc * 16;
    }
    unsigned b_off[NB];
    int b_lds[NB];
    const int n_b = a.NCP * 3 * a.R * a.nq;
    const int rows_valid = min(a.cit, a.C_in - ci0) * s;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int id = sl + 256 * j;
      const bool ok = id < n_b;
      if (!ok) id = 0;
      const int q = id % a.nq;
      int t = id / a.nq;
      int row = t % a.R;
      t /= a.R;
      const int p = t % 3, r = t / 3;
      const int srow = row < rows_valid ? row : 0;                       // rows of channels past C_in: never stored columns
      const int v = ci0 + srow / s, ph = srow % s;                       // virtual channel -> (real channel, k2)
      const int ci_real = v / a.K2, k2 = v - ci_real * a.K2;
      b_off[j] = (unsigned)(p * a.b_plane_bytes + ((long long)(ci_real * s + ph) * a.UB + k2 * a.dil2s + 4 * q + r) * 2);
      b_lds[j] = ok ? WS_A_STAGE + (r * 3 + p) * PSB + row * a.XPB + q * 8 : -1;
    }

    // Exactly LPT loads per tile, all inline asm (invisible to hipcc's s_waitcnt pass, which otherwise waits for the
    // YOUNGEST loads at the tile-dependent joins); after `s_waitcnt vmcnt((D - 1) * LPT)` the tile issued D - 1 tiles ago
    // has landed (loads return in order).  No copies between a load and its wait: the asm writes the ring registers.
    constexpr int LPT = WS_APIECES + NB;
    constexpr int WAITN = (D - 1) * LPT < 63 ? (D - 1) * LPT : 63;
    auto load_tile = [&](int chunk, f32x4 (&ra)[WS_APIECES], f32x2 (&rb)[NB]) {
      const int tile = tile_lo + (chunk < n_chunks ? chunk : n_chunks - 1);     // past the end: reload the last tile (keeps LPT)
      const int b = tile / a.n_tt;
      const int t0 = (tile - b * a.n_tt) * WS_TT;
      const unsigned char* ab = a.ap + ((long long)b * a.C_out * a.UA + t0) * 2;            // uniform
      const unsigned char* bb = a.bp + ((long long)b * (a.C_in / a.K2) * s * a.UB + t0) * 2;   // uniform
#pragma unroll
      for (int j = 0; j < WS_APIECES; ++j)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[j]) : "v"(a_off[j]), "s"(ab) : "memory");
#pragma unroll
      for (int j = 0; j < NB; ++j)
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(rb[j]) : "v"(b_off[j]), "s"(bb) : "memory");
    };
    auto wait_tile = [&](f32x4 (&ra)[WS_APIECES], f32x2 (&rb)[NB]) {
      asm volatile("s_waitcnt vmcnt(%6)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5])
                   : "n"(WAITN) : "memory");
#pragma unroll
      for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(rb[j]) : : "memory");
    };
    auto write_tile = [&](int buf, const f32x4 (&ra)[WS_APIECES], const f32x2 (&rb)[NB]) {
      unsigned char* st = sm + buf * STAGE;
#pragma unroll
      for (int j = 0; j < WS_APIECES; ++j) *reinterpret_cast<f32x4*>(st + a_lds[j]) = ra[j];
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (b_lds[j] >= 0) *reinterpret_cast<f32x2*>(st + b_lds[j]) = rb[j];
    };

    // Ring of D tiles in registers: tile t lives in slot t % D.  Slot c (the MFMA waves multiply tile c): wait for and write
    // tile c + 1 into the other LDS stage, then reuse its registers for the loads of tile c + 1 + D.  Exactly one load_tile
    // per slot (clamped past the end), so the in-flight count behind any tile is always D - 1 tiles.
    f32x4 ra[D][WS_APIECES];
    f32x2 rb[D][NB];
#pragma unroll
    for (int i = 0; i < D; ++i) load_tile(i, ra[i], rb[i]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wait_tile(ra[0], rb[0]);
    write_tile(0, ra[0], rb[0]);
    load_tile(D, ra[0], rb[0]);
    __syncthreads();
    constexpr int U = (D % 2 == 0) ? D : 2 * D;      // unroll: static ring slot and LDS stage per position
    for (int base = 0; base < n_chunks; base += U) {
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int c = base + i;
        if (c < n_chunks) {
          if (c + 1 < n_chunks) {
            wait_tile(ra[(i + 1) % D], rb[(i + 1) % D]);
            write_tile((i + 1) & 1, ra[(i + 1) % D], rb[(i + 1) % D]);
            load_tile(c + 1 + D, ra[(i + 1) % D], rb[(i + 1) % D]);
          }
          __syncthreads();
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

  auto ld_frags = [&](const unsigned char* st, int ks, bf16x8 (&A)[2][3], bf16x8 (&Bf)[2][3]) {
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
  };

  __syncthreads();   // tile 0 staged
  bf16x8 A[2][2][3], Bf[2][2][3];                // fragments of step ks + 1 are requested before the MFMAs of step ks
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char* st = sm + (chunk & 1) * STAGE;
    ld_frags(st, 0, A[0], Bf[0]);
#pragma unroll
    for (int ks = 0; ks < WS_TT / 16; ++ks) {
      if (ks + 1 < WS_TT / 16) ld_frags(st, ks + 1, A[(ks + 1) & 1], Bf[(ks + 1) & 1]);
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
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks & 1][m][TA[q]], Bf[ks & 1][n][TB[q]], acc[m][n], 0, 0, 0);
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

// dW[i] = sum over the S slices in a FIXED order: eight interleaved chains (slice z goes to chain z % 8, each chain in
// increasing z), then the chains pairwise -- deterministic like a plain loop, but with eight loads in flight per lane
// instead of one dependent add per HBM round trip (the plain loop cost 18 ms per training step over 324 launches).
__global__ __launch_bounds__(256) void wgrad_split_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
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

// Geometry shared by the workspace query and the launch.  Returns 0 when the shape runs on the split kernel.
static int ws_geometry(int B, int C_in_real, int T_in, int C_out, int T_out, int K_total, int stride, int dil, int K1, int dil2,
                       WsArgs* a, int* splits, size_t* lds) {
  if (K1 <= 0 || K1 > K_total) K1 = K_total;
  if (K_total % K1 != 0) return -1;
  const int K2 = K_total / K1, K = K1, C_in = C_in_real * K2;       // virtual channels (see WsArgs)
  if (K2 > 1 && (dil2 <= 0 || dil2 % stride != 0)) return -1;
  a->K2 = K2;
  a->dil2s = K2 > 1 ? dil2 / stride : 0;
  if (K < 1 || K > WS_NC || stride < 1 || dil < 1) return -1;
  a->cit = WS_NC / K;
  a->R = a->cit * stride;
  const int max_shift = ((K - 1) * dil) / stride;
  a->NCP = max_shift + 1 < 4 ? max_shift + 1 : 4;
  const int XW = WS_TT + max_shift;
  a->nq = (XW + 3) / 4;
  if ((long long)a->NCP * 3 * a->R * a->nq > 256 * 19) return -1;               // 8-byte B pieces per stage (NB <= 19)
  // Row pitch: 8 * nq bytes of data + 8..64 bytes of padding, chosen to minimise LDS bank conflicts of the B fragment
  // reads (ds_read_b64: 32 lanes per cycle over 64 dword banks; lane J reads row (J / K, phase) of copy shift % 4).
  int best_pitch = -1;
  long best_score = -1;
  for (int p = 1; p <= 8; ++p) {
    const int pitch = 8 * a->nq + 8 * p;
    const size_t st = ((size_t)WS_A_STAGE + (size_t)a->NCP * 3 * a->R * pitch + 15) & ~(size_t)15;
    if (2 * st > 160 * 1024) break;
    long score = 0;
    for (int blk = 0; blk < 4; ++blk) {               // the four 32-column blocks of the tile
      int cnt[64] = {0};
      for (int l = 0; l < 32; ++l) {
        int J = blk * 32 + l;
        if (J >= a->cit * K) J = 0;
        const int cl = J / K, k = J - cl * K, kd = k * dil, shift = kd / stride, ph = kd - shift * stride, r = shift & 3;
        const long addr = (long)(r * 3) * a->R * pitch + (long)(cl * stride + ph) * pitch + (shift - r) * 2;
        cnt[(addr / 4) & 63]++;
        cnt[(addr / 4 + 1) & 63]++;
      }
      int mx = 0;
      for (int i = 0; i < 64; ++i) mx = cnt[i] > mx ? cnt[i] : mx;
      score += mx;
    }
    if (best_score < 0 || score < best_score) { best_score = score; best_pitch = pitch; }
  }
  if (best_pitch < 0) return -1;
  a->XPB = best_pitch;
  const size_t stage = ((size_t)WS_A_STAGE + (size_t)a->NCP * 3 * a->R * a->XPB + 15) & ~(size_t)15;
  *lds = 2 * stage;
  a->n_tt = (T_out + WS_TT - 1) / WS_TT;
  a->UA = a->n_tt * WS_TT;
  a->UB = (a->n_tt * WS_TT + max_shift + (K2 - 1) * a->dil2s + 8 + 7) & ~7;
  a->a_plane_bytes = (long long)B * C_out * a->UA * 2;
  a->b_plane_bytes = (long long)B * C_in_real * stride * a->UB * 2;
  // per-lane plane offsets are 32-bit; per-clip bases are 64-bit
  if (3 * a->a_plane_bytes >= (1ll << 32) || 3 * a->b_plane_bytes >= (1ll << 32)) return -1;
  const long long tiles = (long long)B * a->n_tt;
  const long long wgs = (long long)((C_out + WS_CO - 1) / WS_CO) * ((C_in + a->cit - 1) / a->cit);
  // One workgroup per CU (LDS) and equal-length workgroups: the launch runs in ceil(wgs * S / 256) rounds.  Pick the slice
  // count (around 4 rounds; more slices = more partial-sum traffic) that wastes the least of the last round; bounded by
  // 512 MB of partials.
  const long long per_split_bytes = (long long)C_out * C_in * K * 4;
  long long s_max = (2048 + wgs - 1) / wgs;
  if (s_max > tiles) s_max = tiles;
  if (s_max > 2048) s_max = 2048;
  if (s_max * per_split_bytes > (512ll << 20)) s_max = (512ll << 20) / per_split_bytes;
  if (s_max < 1) s_max = 1;
  long long s_min = (768 + wgs - 1) / wgs;
  if (s_min > s_max) s_min = s_max;
  long long S = s_min;
  double best = -1.0;
  for (long long c = s_min; c <= s_max; ++c) {
    const long long per = (tiles + c - 1) / c, real = (tiles + per - 1) / per;     // slices that actually get tiles
    const long long total = wgs * real, rounds = (total + 255) / 256;
    // time ~ rounds * tiles per slice (+ ~3 tiles of fixed cost per workgroup)
    const double cost = (double)rounds * (double)(per + 3);
    const double eff = (double)(wgs * tiles) / (256.0 * cost);
    if (eff > best + 1e-9) { best = eff; S = c; }
  }
  a->tiles_per_split = (int)((tiles + S - 1) / S);
  *splits = (int)((tiles + a->tiles_per_split - 1) / a->tiles_per_split);
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// k-major variant (round 3): the B operand staged the way the A operand is.
//
// In the kernel above a 32-lane B fragment covers ~4.6 input channels x K taps, i.e. 32 DIFFERENT (row, shift) pairs: the
// ds_read_b64 pairs hit the LDS banks irregularly whatever the row pitch (measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.42-0.51, profiles/r02_pmc_mfma_wgrad.json), and the shifted copies cost up to 4x the staged bytes.  Here the 128 tile
// columns are four BLOCKS of 32, block = (tap k, group of 32 consecutive (virtual) input channels): all 32 lanes of a B
// fragment read the SAME time window of 32 different rows.  So each block's rows are staged once, already shifted by the
// block's own tap offset (the shift moves into the global-load address, any alignment), as 32 rows x 32 steps x 3 planes with
// the A operand's 80-byte pitch: a B fragment is ONE aligned, conflict-free ds_read_b128, exactly like an A fragment.
// No shifted copies, no constraint on K * dilation (the d = 9 layers used to fall back to the fp32 kernel), the same
// staging code for both operands (12 x 16 bytes per staging lane and stage).
// Blocks are numbered gb = group * K + k; partial sums are written block-wise ([slice][co][gb][32], 128-byte runs) and the
// reduction kernel scatters them into dW (C_out, C_in, K) -- the scattered side is the 1x dW, not the S x partials.
// Staging (round 3, second step): the planes are pure copies, and a CU pulls only ~15.6 B/clk from L2 into VGPRs against
// 45-52 B/clk by LDS-DMA (tools/microbench/cu_stream_probe.hip).  A stage of this kernel is 48 KB per 48 MFMAs per wave
// (1536 matrix-pipe cycles) = 32 B/clk: through registers the MFMA waves wait for the copies half of the time (measured MFMA
// busy 36-41 % on either variant).  So both operands move by global_load_lds (16 B per lane, no VGPR, no ds_write), three
// LDS stages deep (loads of stage c + 2 in flight while stage c is multiplied; explicit s_waitcnt vmcnt on the in-order load
// queue, raw s_barrier).  LDS-DMA writes 64 lanes x 16 B contiguously, so rows cannot carry padding: a row is 64 bytes =
// four 16-byte pieces, and piece p of row r lives in slot p ^ ((r >> 2) & 3) -- the 16 lanes of one ds_read_b128 phase
// (rows r .. r + 15, same piece) then cover all 16 four-bank groups (conflict-free), and the DMA side only has to pick, per
// lane, WHICH global piece it fetches for its fixed slot.
// (The B sources are only 2-byte aligned for odd tap shifts: global_load_lds takes them, tests/test_wgrad_split.py.)
constexpr int WK_ROWB = 64;                        // bytes per staged row: 32 bf16
constexpr int WK_PLANE = 128 * WK_ROWB;            // one plane of one operand (8 KB)
constexpr int WK_OPND = 3 * WK_PLANE;              // one operand (24 KB = 24 DMA blocks of 1 KB)
constexpr int WK_STAGE = 2 * WK_OPND;              // 48 KB
constexpr int WK_NST = 3;                          // LDS stages
constexpr int WK_PIECES = 6;                       // 16-byte pieces per staging lane, operand and stage

struct WkArgs {
  const unsigned char* ap;   // dy planes [3][B*C_out][UA]
  const unsigned char* bp;   // x planes  [3][B*C_in_real*stride][UB]
  float* part;               // [S][C_out][NBk][32]
  long long a_plane_bytes, b_plane_bytes;
  int UA, UB;
  int B, C_in_real, CV, C_out, K, stride, dil, K2, dil2s;
  int NBk;                   // blocks = ceil(CV / 32) * K
  int n_tt, tiles_per_split;
  int MT;                    // 128-row tiles per workgroup (1 or 2)
  int gx, gy, gz, per_xcd;   // logical grid (row tiles, column-block quads, (b, t) slices) and workgroups per XCD (0: plain 3-D grid)
  int narrow_rows;           // bit 0: row tiles with <= 96 real output channels run the column-split wave layouts (FAC_WGRAD_NARROW=0: off);
                             // bit 1: their clamped duplicate rows are not staged (FAC_WGRAD_SKIP_DUP=0: staged as before)
};

__device__ __forceinline__ void wk_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// MT = row tiles of 128 per workgroup.  MT = 1: 4 MFMA waves (64 x 64 each) + 4 DMA waves, three LDS stages of 48 KB.
// MT = 2 (round 4, C_out > 128): 256 x 128 tile, EIGHT MFMA waves -- two per SIMD, so one wave's fragment reads and barrier skew hide
// under the other's MFMAs -- + 4 DMA waves, two stages of 72 KB.  What the stage costs the matrix pipe is its LDS WRITE traffic
// (DESIGN 9.2), and a 128 x 128 tile writes 48 KB per 48 MFMAs of a wave; 256 x 128 writes 72 KB per 96 MFMAs of a SIMD: a quarter
// less per MFMA, and half the barriers.
template <int MT, bool KSP = false>
__global__ __launch_bounds__((4 * MT + 4) * 64, MT == 1 ? 2 : 1) void conv1d_wgrad_kmajor_kernel(WkArgs a) {
  static_assert(!KSP || MT == 1, "the k-split wave layout is a 128 x 128 tile");
  constexpr int NMW = 4 * MT;                     // MFMA waves
  constexpr int A_PLANE = MT * WK_PLANE;          // one plane of the dy operand
  constexpr int A_OPND = 3 * A_PLANE;
  constexpr int STAGE = A_OPND + WK_OPND;
  constexpr int NST = MT == 1 ? 3 : 2;
  constexpr int PA = WK_PIECES * MT, PB = WK_PIECES;      // 16-byte pieces per staging lane, operand and stage
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order (round 5): the launch is 1-D; workgroup id i runs on XCD i % 8 (each XCD has its own L2), and XCD k takes
  // the contiguous range [k * per_xcd, (k + 1) * per_xcd) of the logical order (row tile fastest, then column blocks, slice
  // slowest).  Workgroups of one (b, t) slice read the SAME rows of both operand planes at the same time -- all row tiles share
  // the x planes, all column blocks the dy planes -- so a slice's workgroups now meet in one L2 instead of eight.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.per_xcd > 0) {
    const int logical = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= a.per_xcd || logical >= a.gx * a.gy * a.gz) return;
    bx = logical % a.gx;
    const int r = logical / a.gx;
    by = r % a.gy;
    bz = r / a.gy;
  }
  const int co0 = bx * 128 * MT;
  const int gb0 = by * 4;
  const int z = bz;
  const int tile_lo = z * a.tiles_per_split;
  const int tile_hi = min(a.B * a.n_tt, tile_lo + a.tiles_per_split);
  const int n_chunks = tile_hi - tile_lo;
  const int s = a.stride;

  if (wave >= NMW) {
    // ===================================================================== staging waves: planes -> LDS by LDS-DMA
    const int lw = wave - NMW;
    __builtin_amdgcn_s_setprio(3);
    unsigned a_off[PA], b_off[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int blk = j * 4 + lw;                 // 1 KB block of the operand: 16 rows of one plane
      const int plane = blk / (8 * MT);
      const int row = (blk - plane * 8 * MT) * 16 + (lane >> 2);
      const int piece = (lane & 3) ^ ((row >> 2) & 3);      // the global piece that belongs in this lane's slot
      const int co = co0 + row < a.C_out ? co0 + row : a.C_out - 1;        // rows past C_out: computed, never stored
      a_off[j] = (unsigned)(plane * a.a_plane_bytes + ((long long)co * a.UA + 8 * piece) * 2);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int blk = j * 4 + lw;
      const int plane = blk >> 3;
      const int row = (blk & 7) * 16 + (lane >> 2);
      const int piece = (lane & 3) ^ ((row >> 2) & 3);
      int gb = gb0 + (row >> 5);
      gb = gb < a.NBk ? gb : a.NBk - 1;                                    // blocks past the end: never stored
      const int g = gb / a.K, k = gb - g * a.K;
      int v = g * 32 + (row & 31);
      v = v < a.CV ? v : a.CV - 1;                                         // channels past the end: never stored
      const int ci = v / a.K2, k2 = v - ci * a.K2;
      const int kd = k * a.dil, shift = kd / s, ph = kd - shift * s;
      b_off[j] = (unsigned)(plane * a.b_plane_bytes + ((long long)(ci * s + ph) * a.UB + shift + k2 * a.dil2s + 8 * piece) * 2);
    }
    constexpr int LPT = PA + PB;                  // DMA instructions per lane and stage
    // Round 6: a row tile whose upper row blocks are clamped duplicates (the column-split wave layouts below read rows
    // [0, 32 / 64 / 96) only) does not stage them: 18 / 12 / 6 of the stage's 48 KB stay in the L2s.  These launches are bound by the
    // traffic into LDS (32 real rows: 12 MFMAs per wave and 48 KB stage), not by the matrix pipe.  Block j of this wave is 16 rows
    // of one plane; which blocks it skips is wave-uniform, and so is the number n_a it issues per stage (0, 3 or 6).
    int need_rows = 128 * MT;
    if (MT == 1 && (a.narrow_rows & 2)) {
      const int real = a.C_out - co0;
      if (KSP) need_rows = real <= 64 ? 64 : (real <= 96 ? 96 : 128);        // (the k-split layout reads at least two row blocks)
      else need_rows = real <= 32 ? 32 : (real <= 64 ? 64 : (real <= 96 ? 96 : 128));
    }
    bool need_a[PA];
    int n_a = 0;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int blk = j * 4 + lw;
      const int plane = blk / (8 * MT);
      need_a[j] = (blk - plane * 8 * MT) * 16 < need_rows;
      n_a += need_a[j] ? 1 : 0;
    }
    auto issue = [&](int chunk, int buf) {
      const int tile = tile_lo + (chunk < n_chunks ? chunk : n_chunks - 1);     // past the end: reload the last tile (keeps LPT)
      const int b = tile / a.n_tt;
      const int t0 = (tile - b * a.n_tt) * WS_TT;
      const unsigned char* ab = a.ap + ((long long)b * a.C_out * a.UA + t0) * 2;                 // uniform
      const unsigned char* bb = a.bp + ((long long)b * a.C_in_real * s * a.UB + t0) * 2;        // uniform
      unsigned char* st = sm + buf * STAGE;
#pragma unroll
      for (int j = 0; j < PA; ++j)
        if (need_a[j]) __builtin_amdgcn_global_load_lds((glb_void_t*)(ab + a_off[j]), (lds_void_t*)(st + (j * 4 + lw) * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < PB; ++j)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(bb + b_off[j]), (lds_void_t*)(st + A_OPND + (j * 4 + lw) * 1024), 16, 0, 0);
    };
    // everything but the youngest stage's loads has landed (loads return in order)
    auto landed = [&](bool younger_in_flight) {
      if (!younger_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (n_a == PA) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(LPT) : "memory");
      else if (n_a == 3) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PB + 3) : "memory");
      else if (n_a == 0) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    issue(0, 0);
    if (NST == 3 && n_chunks > 1) issue(1, 1);
    landed(NST == 3 && n_chunks > 1);
    wk_barrier();                                 // stage 0 visible to the MFMA waves
    for (int base = 0; base < n_chunks; base += NST) {
#pragma unroll
      for (int i = 0; i < NST; ++i) {             // static LDS stage
        const int c = base + i;
        if (c < n_chunks) {
          // stage (c + NST - 1) % NST was read during iteration c - 1, which every wave has left
          if (c + NST - 1 < n_chunks) issue(c + NST - 1, (i + NST - 1) % NST);
          if (c + 1 < n_chunks) landed(NST == 3 && c + 2 < n_chunks);
          wk_barrier();
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  if constexpr (KSP) {
    // ===================================================== MFMA waves, k-split layout (round 5): 64 rows x ALL 128 columns each,
    // wave (mh, kh) contracts only time steps [16 kh, 16 kh + 16) of every 32-step stage.  A stage then costs a wave
    // 6 + 12 = 18 fragment reads for its 48 MFMAs where the 64 x 64 layout needs 24 (12 per k step): a quarter less LDS read
    // traffic on a loop that is LDS-bandwidth-bound.  The two k halves of a 64-row block meet once, at the end,
    // through LDS (fixed order: kh = 0 + kh = 1).
    // Products in the order (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi) -- the three 2^-16 terms first, as before, but
    // with the two lo fragments FIRST: they are dead after two products, and their registers (plus one spare set) take the next
    // stage's (A lo, B lo, A hi, B hi) fragments while the last four products run.  The stage barrier sits right behind the point
    // where every fragment of the stage is in registers, i.e. in FRONT of those four products: the next stage's first two products
    // never wait for LDS.
    //
    // Round 6 (second form of the narrow-rows idea below): a row tile with at most 64 / 96 REAL output channels -- every C = 64 layer,
    // the second row tile of the C = 192 layers, the C = 96 layers -- splits its waves over the COLUMNS instead: wave (nh, kh) takes
    // the MB = 2 / 3 real 32-row blocks x columns [64 nh, 64 nh + 64) x its k half, 24 / 36 MFMAs per stage instead of 48 of which
    // half / a quarter multiplied clamped duplicate rows.  Same staging, same barriers, same product order per accumulator, same
    // (kh = 0) + (kh = 1) exchange: every dW element is the same sum in the same order as in the row-split layout.
    auto ksp_body = [&](auto mb_tag, auto nb_tag, auto cs_tag) {
    constexpr int MB = decltype(mb_tag)::value, NB = decltype(nb_tag)::value;
    constexpr bool CS = decltype(cs_tag)::value;     // waves split over columns (true) or over rows (false)
    const int l31 = lane & 31, kq = lane >> 5;
    const int wh = wave >> 1, kh = wave & 1;
    const int sw = (l31 >> 2) & 3;
    const int po = ((kh * 2 + kq) ^ sw) * 16;
    const int aoff = ((CS ? 0 : wh * 64) + l31) * WK_ROWB + po;
    const int boff = A_OPND + ((CS ? wh * 64 : 0) + l31) * WK_ROWB + po;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    // fragment sets: Ahi / Bhi alternate between two register sets (the current stage's stay live to its last product);
    // lo and mid fragments are re-used in place
    bf16x8 Ahi[2][MB], Bhi[2][NB], Amid[MB], Bmid[NB], Alo[MB], Blo[NB];
    auto rd_a = [&](const unsigned char* st, int plane, bf16x8 (&d)[MB]) {
#pragma unroll
      for (int m = 0; m < MB; ++m) d[m] = *reinterpret_cast<const bf16x8*>(st + aoff + plane * A_PLANE + m * 32 * WK_ROWB);
    };
    auto rd_b = [&](const unsigned char* st, int plane, bf16x8 (&d)[NB]) {
#pragma unroll
      for (int n = 0; n < NB; ++n) d[n] = *reinterpret_cast<const bf16x8*>(st + boff + plane * WK_PLANE + n * 32 * WK_ROWB);
    };
    auto mm = [&](const bf16x8 (&x)[MB], const bf16x8 (&y)[NB]) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[m], y[n], acc[m][n], 0, 0, 0);
    };
    wk_barrier();   // stage 0 staged
    rd_a(sm, 2, Alo); rd_b(sm, 0, Bhi[0]); rd_a(sm, 0, Ahi[0]); rd_b(sm, 2, Blo);
    // chunk c lives in LDS stage c % 3 and in fragment set c & 1: the pattern repeats every 6 chunks
    for (int base = 0; base < n_chunks; base += 6) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int chunk = base + i;
        if (chunk < n_chunks) {
          const unsigned char* st = sm + (i % 3) * STAGE;
          const unsigned char* stn = sm + ((i + 1) % 3) * STAGE;
          const int cur = i & 1, nxt = cur ^ 1;
          rd_a(st, 1, Amid); rd_b(st, 1, Bmid);
          __builtin_amdgcn_sched_barrier(0);
          mm(Alo, Bhi[cur]);                       // (lo, hi)
          mm(Ahi[cur], Blo);                       // (hi, lo)
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0), as a builtin so that the compiler's own wait-count bookkeeping
                                                   // sees it: every fragment of this stage is in registers
          wk_barrier();                            // this stage may be overwritten; the next one has landed
          // (behind the last chunk this reads a stage nobody filled: valid LDS, values never used -- no branch, no join)
          rd_a(stn, 2, Alo); rd_b(stn, 0, Bhi[nxt]); rd_a(stn, 0, Ahi[nxt]); rd_b(stn, 2, Blo);
          __builtin_amdgcn_sched_barrier(0);
          mm(Amid, Bmid);                          // (mid, mid)
          mm(Amid, Bhi[cur]);                      // (mid, hi)
          mm(Ahi[cur], Bmid);                      // (hi, mid)
          mm(Ahi[cur], Bhi[cur]);                  // (hi, hi)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // the two k halves of each wave pair: kh = 1 parks its sums in LDS (all three stages are free: the DMA waves issue nothing
    // past the last chunk and every MFMA wave is past its last fragment read), kh = 0 adds them and stores the partial dW
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float* ex = reinterpret_cast<float*>(sm) + wh * (8 * 16 * 64);       // [m][n][r][lane] per wave pair: at most 32 KB
    if (kh == 1) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) ex[((m * NB + n) * 16 + r) * 64 + lane] = acc[m][n][r];
    }
    __syncthreads();
    if (kh == 1) return;
    float* pz = a.part + (long long)z * a.C_out * a.NBk * 32;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int gb = gb0 + (CS ? wh * 2 : 0) + n;
        if (gb >= a.NBk) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (CS ? 0 : wh * 64) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
          const float v = acc[m][n][r] + ex[((m * NB + n) * 16 + r) * 64 + lane];
          if (co < a.C_out) pz[((long long)co * a.NBk + gb) * 32 + l31] = v;
        }
      }
    };
    using i2 = std::integral_constant<int, 2>;
    using i3 = std::integral_constant<int, 3>;
    using i4 = std::integral_constant<int, 4>;
    const int real_rows = a.C_out - co0;
    if (a.narrow_rows && real_rows <= 64) ksp_body(i2(), i2(), std::true_type());
    else if (a.narrow_rows && real_rows <= 96) ksp_body(i3(), i2(), std::true_type());
    else ksp_body(i2(), i4(), std::false_type());
    return;
  }

  // ========================================================================= MFMA waves: 64 x 64 each (2 x 2 blocks)
  const int l31 = lane & 31, kq = lane >> 5;
  const int mh = wave >> 1, nh = wave & 1;
  const int sw = (l31 >> 2) & 3;
  const int aoff = (mh * 64 + l31) * WK_ROWB;
  const int boff = A_OPND + (nh * 64 + l31) * WK_ROWB;
  int poff[WS_TT / 16];
#pragma unroll
  for (int ks = 0; ks < WS_TT / 16; ++ks) poff[ks] = ((ks * 2 + kq) ^ sw) * 16;

  if constexpr (MT == 1) {
    // ---- at most 32 real output channels in this row tile (round 6: the multi-resolution discriminator's 32-channel stacks and
    // the 1-channel output convs -- 75 + of the step's 227 weight gradients): three quarters of the 128 x 128 tile's MFMAs
    // multiplied padding rows.  The four MFMA waves share the ONE real 32-row block and take 32 columns each (wave w: columns
    // [32 w, 32 w + 32) of the tile): 24 MFMAs per 16-step slab instead of 96; the staging waves and the barrier protocol are
    // untouched.  Two accumulators per wave (products 0 / 2 / 4 and 1 / 3 / 5) so that consecutive MFMAs do not depend on each other.
    if (a.narrow_rows && a.C_out - co0 <= 32) {
      const int aoff1 = l31 * WK_ROWB;
      const int boff1 = A_OPND + (wave * 32 + l31) * WK_ROWB;
      f32x16 acc_a, acc_b;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_a[r] = 0.f; acc_b[r] = 0.f; }
      bf16x8 A1[2][3], B1[2][3];
      auto ld1 = [&](const unsigned char* st, int ks, bf16x8 (&Ad)[3], bf16x8 (&Bd)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          Ad[p] = *reinterpret_cast<const bf16x8*>(st + aoff1 + p * A_PLANE + poff[ks]);
          Bd[p] = *reinterpret_cast<const bf16x8*>(st + boff1 + p * WK_PLANE + poff[ks]);
        }
      };
      wk_barrier();   // stage 0 staged
      for (int base = 0; base < n_chunks; base += NST) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int chunk = base + i;
          if (chunk < n_chunks) {
            const unsigned char* st = sm + i * STAGE;
            ld1(st, 0, A1[0], B1[0]);
#pragma unroll
            for (int ks = 0; ks < WS_TT / 16; ++ks) {
              if (ks + 1 < WS_TT / 16) ld1(st, ks + 1, A1[(ks + 1) & 1], B1[(ks + 1) & 1]);
              __builtin_amdgcn_sched_barrier(0);
              constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};     // smallest terms first
#pragma unroll
              for (int q = 0; q < 6; q += 2) {
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[ks & 1][TA[q]], B1[ks & 1][TB[q]], acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[ks & 1][TA[q + 1]], B1[ks & 1][TB[q + 1]], acc_b, 0, 0, 0);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            wk_barrier();
          }
        }
      }
      float* pz1 = a.part + (long long)z * a.C_out * a.NBk * 32;
      const int gb = gb0 + wave;
      if (gb < a.NBk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kq;
          if (co < a.C_out) pz1[((long long)co * a.NBk + gb) * 32 + l31] = acc_a[r] + acc_b[r];
        }
      }
      return;
    }
    // ---- 33 .. 96 real output channels (C = 64 / 96 layers, the second row tile of the C = 192 layers; round 6): the same column
    // split with R = 2 / 3 row blocks per wave -- wave w: R x 32 rows x columns [32 w, 32 w + 32), 6 R MFMAs per 16-step slab
    // instead of 24 of which a half / a quarter multiplied clamped duplicate rows.  One accumulator per (row block, column block)
    // and the product order of the 64 x 64 layout: the same sums in the same order, bit for bit.
    auto colsplit = [&](auto r_tag) {
      constexpr int R = decltype(r_tag)::value;
      const int aoff1 = l31 * WK_ROWB;
      const int boff1 = A_OPND + (wave * 32 + l31) * WK_ROWB;
      f32x16 accr[R];
#pragma unroll
      for (int m = 0; m < R; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) accr[m][r] = 0.f;
      bf16x8 A1[2][R][3], B1[2][3];
      auto ld1 = [&](const unsigned char* st, int ks, bf16x8 (&Ad)[R][3], bf16x8 (&Bd)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int m = 0; m < R; ++m) Ad[m][p] = *reinterpret_cast<const bf16x8*>(st + aoff1 + p * A_PLANE + m * 32 * WK_ROWB + poff[ks]);
          Bd[p] = *reinterpret_cast<const bf16x8*>(st + boff1 + p * WK_PLANE + poff[ks]);
        }
      };
      wk_barrier();   // stage 0 staged
      for (int base = 0; base < n_chunks; base += NST) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int chunk = base + i;
          if (chunk < n_chunks) {
            const unsigned char* st = sm + i * STAGE;
            ld1(st, 0, A1[0], B1[0]);
#pragma unroll
            for (int ks = 0; ks < WS_TT / 16; ++ks) {
              if (ks + 1 < WS_TT / 16) ld1(st, ks + 1, A1[(ks + 1) & 1], B1[(ks + 1) & 1]);
              __builtin_amdgcn_sched_barrier(0);
              constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};     // smallest terms first
#pragma unroll
              for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int m = 0; m < R; ++m)
                  accr[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[ks & 1][m][TA[q]], B1[ks & 1][TB[q]], accr[m], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
            wk_barrier();
          }
        }
      }
      float* pzr = a.part + (long long)z * a.C_out * a.NBk * 32;
      const int gb = gb0 + wave;
      if (gb < a.NBk) {
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
            if (co < a.C_out) pzr[((long long)co * a.NBk + gb) * 32 + l31] = accr[m][r];
          }
      }
    };
    if (a.narrow_rows && a.C_out - co0 <= 64) {
      colsplit(std::integral_constant<int, 2>());
      return;
    }
    if (a.narrow_rows && a.C_out - co0 <= 96) {
      colsplit(std::integral_constant<int, 3>());
      return;
    }
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  auto ld_a = [&](const unsigned char* st, int ks, bf16x8 (&A)[2][3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        A[m][p] = *reinterpret_cast<const bf16x8*>(st + aoff + p * A_PLANE + m * 32 * WK_ROWB + poff[ks]);
  };
  auto ld_b = [&](const unsigned char* st, int ks, bf16x8 (&Bf)[2][3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        Bf[n][p] = *reinterpret_cast<const bf16x8*>(st + boff + p * WK_PLANE + n * 32 * WK_ROWB + poff[ks]);
  };

  wk_barrier();   // stage 0 staged
  // MT = 1: both operands' fragments are requested a step ahead (two register sets).  MT = 2 (three waves per SIMD: 168 registers):
  // only the dy fragments are; the x fragments are requested at the start of their step and the sibling MFMA wave covers the wait.
  constexpr int NBS = MT == 1 ? 2 : 1;
  bf16x8 A[2][2][3], Bf[NBS][2][3];
  for (int base = 0; base < n_chunks; base += NST) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int chunk = base + i;
      if (chunk < n_chunks) {
        const unsigned char* st = sm + i * STAGE;
        ld_a(st, 0, A[0]);
        if (NBS == 2) ld_b(st, 0, Bf[0]);
#pragma unroll
        for (int ks = 0; ks < WS_TT / 16; ++ks) {
          if (NBS == 1) ld_b(st, ks, Bf[0]);
          if (ks + 1 < WS_TT / 16) {
            ld_a(st, ks + 1, A[(ks + 1) & 1]);
            if (NBS == 2) ld_b(st, ks + 1, Bf[(ks + 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
          constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};     // smallest terms first (see above)
#pragma unroll
          for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks & 1][m][TA[q]], Bf[NBS == 2 ? (ks & 1) : 0][n][TB[q]], acc[m][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        wk_barrier();
      }
    }
  }

  // partial dW of this slice, block-wise: [co][gb][32 channels] -- 128-byte runs per (row, block)
  float* pz = a.part + (long long)z * a.C_out * a.NBk * 32;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int gb = gb0 + nh * 2 + n;
      if (gb >= a.NBk) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mh * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        if (co < a.C_out) pz[((long long)co * a.NBk + gb) * 32 + l31] = acc[m][n][r];
      }
    }
}

// dW[co][v * K + k] = sum over the S slices (eight interleaved chains in a fixed order, as wgrad_split_reduce_kernel) of
// part[z][co][g * K + k][v % 32], g = v / 32: one thread per partial element -- coalesced reads of the S x larger side.
__global__ __launch_bounds__(256) void wgrad_kmajor_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int C_out,
                                                                  int NBk, int K, int CV) {
  const long long n = (long long)C_out * NBk * 32;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int cl = (int)(i & 31);
    const long long r = i >> 5;
    const int gb = (int)(r % NBk);
    const int co = (int)(r / NBk);
    const int g = gb / K, k = gb - g * K;
    const int v = g * 32 + cl;
    if (v >= CV) continue;
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* p = part + i;
    int z = 0;
    for (; z + 8 <= S; z += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] += p[(long long)(z + j) * n];
    }
    for (int j = 0; z + j < S; ++j) c[j] += p[(long long)(z + j) * n];
    dw[((long long)co * CV + v) * K + k] = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
  }
}

// Geometry of the k-major kernel; returns 0 when the shape runs on it.  Used for (virtual) channel counts >= 16: below that a
// block of 32 lanes would hold one or two useful columns (the 1- and 2-channel input layers), where the kernel above is better.
static int wk_geometry(int B, int C_in_real, int T_in, int C_out, int T_out, int K_total, int stride, int dil, int K1, int dil2,
                       WkArgs* a, int* splits) {
  static const bool on = !(getenv("FAC_WGRAD_KMAJOR") && getenv("FAC_WGRAD_KMAJOR")[0] == '0');
  if (!on) return -1;
  if (K1 <= 0 || K1 > K_total) K1 = K_total;
  if (K_total % K1 != 0) return -1;
  const int K2 = K_total / K1, K = K1, CV = C_in_real * K2;
  if (K2 > 1 && (dil2 <= 0 || dil2 % stride != 0)) return -1;
  if (CV < 16 || K < 1 || stride < 1 || dil < 1) return -1;
  a->K2 = K2;
  a->dil2s = K2 > 1 ? dil2 / stride : 0;
  a->K = K; a->stride = stride; a->dil = dil; a->B = B; a->C_in_real = C_in_real; a->CV = CV; a->C_out = C_out;
  a->NBk = ((CV + 31) / 32) * K;
  const int max_shift = ((K - 1) * dil) / stride;
  a->n_tt = (T_out + WS_TT - 1) / WS_TT;
  a->UA = a->n_tt * WS_TT;
  a->UB = (a->n_tt * WS_TT + max_shift + (K2 - 1) * a->dil2s + 8 + 7) & ~7;
  a->a_plane_bytes = (long long)B * C_out * a->UA * 2;
  a->b_plane_bytes = (long long)B * C_in_real * stride * a->UB * 2;
  if (3 * a->a_plane_bytes >= (1ll << 32) || 3 * a->b_plane_bytes >= (1ll << 32)) return -1;     // 32-bit per-lane offsets
  const long long tiles = (long long)B * a->n_tt;
  // measured SLOWER than the 128-row tile on every layer (C = 512 k7: 145 vs 163 TFLOP/s-eq, LSTM W_ih: 121 vs 135,
  // profiles/r04_wgrad_wide_tile.log): two LDS stages expose the DMA latency that three stages hide, and that outweighs the
  // quarter fewer LDS writes per MFMA.  Opt-in (FAC_WGRAD_WIDE=1).
  static const bool wide_on = getenv("FAC_WGRAD_WIDE") && getenv("FAC_WGRAD_WIDE")[0] == '1';
  a->MT = (wide_on && C_out > 128) ? 2 : 1;
  const long long wgs = (long long)((C_out + 128 * a->MT - 1) / (128 * a->MT)) * ((a->NBk + 3) / 4);
  const long long per_split_bytes = (long long)C_out * a->NBk * 32 * 4;
  long long s_max = (2048 + wgs - 1) / wgs;
  if (s_max > tiles) s_max = tiles;
  if (s_max > 2048) s_max = 2048;
  if (s_max * per_split_bytes > (512ll << 20)) s_max = (512ll << 20) / per_split_bytes;
  if (s_max < 1) s_max = 1;
  long long s_min = (768 + wgs - 1) / wgs;
  if (s_min > s_max) s_min = s_max;
  long long S = s_min;
  double best = -1.0;
  for (long long c = s_min; c <= s_max; ++c) {        // same cost model as ws_geometry: one workgroup per CU, equal lengths
    const long long per = (tiles + c - 1) / c, real = (tiles + per - 1) / per;
    const long long total = wgs * real, rounds = (total + 255) / 256;
    const double cost = (double)rounds * (double)(per + 3);
    const double eff = (double)(wgs * tiles) / (256.0 * cost);
    if (eff > best + 1e-9) { best = eff; S = c; }
  }
  a->tiles_per_split = (int)((tiles + S - 1) / S);
  *splits = (int)((tiles + a->tiles_per_split - 1) / a->tiles_per_split);
  return 0;
}

template <int NB, int D>
static void ws_launch(const WsArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
  auto kern = conv1d_wgrad_planes_kernel<NB, D>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a);
}

static long long ws_align(long long v) { return (v + 255) & ~255ll; }

}  // namespace fac

// workspace = [partials | dy planes | x planes]
extern "C" int64_t fac_conv1d_bwd_weight_split_ws_bytes(int B, int C_in, int T_in, int C_out, int T_out, int K, int stride,
                                                        int dilation, int K1, int dilation2) {
  {
    fac::WkArgs k;
    int S;
    if (fac::wk_geometry(B, C_in, T_in, C_out, T_out, K, stride, dilation, K1, dilation2, &k, &S) == 0)
      return fac::ws_align((int64_t)S * C_out * k.NBk * 32 * 4) + fac::ws_align(3 * k.a_plane_bytes) + fac::ws_align(3 * k.b_plane_bytes) +
             fac::ws_align((int64_t)B * C_out * ((k.UA + 2047) / 2048) * 4);
  }
  fac::WsArgs a;
  int S;
  size_t lds;
  if (fac::ws_geometry(B, C_in, T_in, C_out, T_out, K, stride, dilation, K1, dilation2, &a, &S, &lds)) return -1;
  return fac::ws_align((int64_t)S * C_out * C_in * K * 4) + fac::ws_align(3 * a.a_plane_bytes) + fac::ws_align(3 * a.b_plane_bytes) +
         fac::ws_align((int64_t)B * C_out * ((a.UA + 2047) / 2048) * 4);
}

static int bwd_weight_split_impl(const float* x, const float* dy, float* dw, float* db, void* ws, int64_t ws_bytes, int B, int C_in,
                                 int T_in, int C_out, int T_out, int K, int stride, int dilation, int pad_left, int pad_mode, int K1,
                                 int dilation2, fac_stream_t stream);

extern "C" int fac_conv1d_bwd_weight_split_db_ok(int B, int C_in, int T_in, int C_out, int T_out, int K, int stride, int dilation, int K1,
                                                 int dilation2) {
  fac::WkArgs k;
  int S;
  return fac::wk_geometry(B, C_in, T_in, C_out, T_out, K, stride, dilation, K1, dilation2, &k, &S) == 0 && (long long)B * C_out <= 65535;
}

extern "C" int fac_conv1d_bwd_weight_split(const float* x, const float* dy, float* dw, void* ws, int64_t ws_bytes, int B,
                                           int C_in, int T_in, int C_out, int T_out, int K, int stride, int dilation,
                                           int pad_left, int pad_mode, int K1, int dilation2, fac_stream_t stream) {
  return bwd_weight_split_impl(x, dy, dw, nullptr, ws, ws_bytes, B, C_in, T_in, C_out, T_out, K, stride, dilation, pad_left, pad_mode, K1,
                               dilation2, stream);
}

extern "C" int fac_conv1d_bwd_weight_split_db(const float* x, const float* dy, float* dw, float* db, void* ws, int64_t ws_bytes, int B,
                                              int C_in, int T_in, int C_out, int T_out, int K, int stride, int dilation,
                                              int pad_left, int pad_mode, int K1, int dilation2, fac_stream_t stream) {
  return bwd_weight_split_impl(x, dy, dw, db, ws, ws_bytes, B, C_in, T_in, C_out, T_out, K, stride, dilation, pad_left, pad_mode, K1,
                               dilation2, stream);
}

static int bwd_weight_split_impl(const float* x, const float* dy, float* dw, float* db, void* ws, int64_t ws_bytes, int B, int C_in,
                                 int T_in, int C_out, int T_out, int K, int stride, int dilation, int pad_left, int pad_mode, int K1,
                                 int dilation2, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(x && dy && dw && ws && B > 0 && C_in > 0 && C_out > 0 && T_in > 0 && T_out > 0 && K > 0 && stride > 0 &&
                  dilation > 0 && pad_left >= 0,
              "conv1d_bwd_weight_split: bad arguments");
  {
    WkArgs k;
    int S;
    if (wk_geometry(B, C_in, T_in, C_out, T_out, K, stride, dilation, K1, dilation2, &k, &S) == 0) {
      const long long part_bytes = ws_align((long long)S * C_out * k.NBk * 32 * 4);
      FAC_REQUIRE(ws_bytes >= part_bytes + ws_align(3 * k.a_plane_bytes) + ws_align(3 * k.b_plane_bytes) +
                                  ws_align((long long)B * C_out * ((k.UA + 2047) / 2048) * 4),
                  "conv1d_bwd_weight_split: workspace too small");
      unsigned char* wsb = reinterpret_cast<unsigned char*>(ws);
      k.part = reinterpret_cast<float*>(ws);
      unsigned char* ap = wsb + part_bytes;
      unsigned char* bp = ap + ws_align(3 * k.a_plane_bytes);
      k.ap = ap; k.bp = bp;
      const hipStream_t st = (hipStream_t)stream;
      const int K2 = k.K2, K1e = k.K;
      long long last = (long long)(T_out - 1) * stride + (long long)(K2 - 1) * (K2 > 1 ? dilation2 : 0) + (long long)(K1e - 1) * dilation - pad_left;
      const int pad_right = last >= T_in ? (int)(last - T_in + 1) : 0;
      const int max_pad = pad_left > pad_right ? pad_left : pad_right;
      const int T_ext = T_in > max_pad ? T_in : max_pad + 1;
      const int T_pad = (int)(pad_left + (last + 1 > T_in ? last + 1 : T_in));
      const long long na = (long long)B * C_out * (k.UA / 8), nb = (long long)B * C_in * stride * (k.UB / 8);
      const int ga = (int)((na + 255) / 256 < 65535 * 16 ? (na + 255) / 256 : 65535 * 16);
      const int gb = (int)((nb + 255) / 256 < 65535 * 16 ? (nb + 255) / 256 : 65535 * 16);
      if (db != nullptr && (long long)B * C_out <= 65535) {
        const int n_chunks = (k.UA + 2047) / 2048;
        float* rs = reinterpret_cast<float*>(bp + ws_align(3 * k.b_plane_bytes));
        // short rows (the 160-frame layers: 20 pieces of 8 steps per row) get one wave per row instead of 256 threads of which 236 idle
        hipLaunchKernelGGL(split_planes_rowsum_kernel, dim3(n_chunks, B * C_out), dim3(k.UA <= 512 ? 64 : 256), 0, st, dy, ap, rs, T_out, k.UA,
                           n_chunks, k.a_plane_bytes);
        hipLaunchKernelGGL(bias_from_rowsums_kernel, dim3((C_out + 255) / 256), dim3(256), 0, st, rs, db, B, C_out, n_chunks);
        db = nullptr;
      } else {
        hipLaunchKernelGGL(split_planes_kernel, dim3(ga), dim3(256), 0, st, dy, ap, (long long)B * C_out, T_out, T_out, T_out, 1, k.UA, 0,
                           FAC_PAD_ZERO, k.a_plane_bytes);
      }
      FAC_REQUIRE(db == nullptr, "conv1d_bwd_weight_split_db: too many rows for the fused bias gradient (B * C_out > 65535)");
      hipLaunchKernelGGL(split_planes_kernel, dim3(gb), dim3(256), 0, st, x, bp, (long long)B * C_in, T_in, T_ext, T_pad, stride, k.UB,
                         pad_left, pad_mode, k.b_plane_bytes);
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_wgrad_kmajor_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_wgrad_kmajor_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_wgrad_kmajor_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr_set = true;
      }
      dim3 grid((C_out + 128 * k.MT - 1) / (128 * k.MT), (k.NBk + 3) / 4, S);
      // Measured policy (profiles/r05_wgrad_xcd_ksplit.log, same box, B = 16 training shapes):
      //  * XCD-aware order when the launch has at least 5 (b, t) slices: +4 .. +20 % on the ResidualUnit / strided / transposed
      //    layers (each XCD then works on whole slices); with 2 - 4 slices (LSTM input projections, the 1024 -> 1536 conv at
      //    T = 160) an XCD gets a fraction of a slice and the plain order measured 3 - 9 % faster;
      //  * k-split wave layout for the k = 7 stride-1 layers whose slice is at most one round of workgroups: +2 .. +5 % there,
      //    -2 .. -8 % on 1-tap / strided / wide layers (their loops are short: the end-of-loop exchange shows).
      static const int xcd_env = [] { const char* e = getenv("FAC_WGRAD_XCD"); return e == nullptr ? -1 : (e[0] != '0' ? 1 : 0); }();
      static const int ksp_env = [] { const char* e = getenv("FAC_WGRAD_KSPLIT"); return e == nullptr ? -1 : (e[0] != '0' ? 1 : 0); }();
      const bool xcd_order = xcd_env >= 0 ? xcd_env == 1 : S >= 5;
      const bool ksplit = k.MT == 1 && (ksp_env >= 0 ? ksp_env == 1
                                                     : (k.K == 7 && k.K2 == 1 && stride == 1 && (long long)grid.x * grid.y <= 256 && S >= 4));
      static const bool narrow_on = !(getenv("FAC_WGRAD_NARROW") && getenv("FAC_WGRAD_NARROW")[0] == '0');
      static const bool skip_dup = !(getenv("FAC_WGRAD_SKIP_DUP") && getenv("FAC_WGRAD_SKIP_DUP")[0] == '0');
      k.narrow_rows = narrow_on ? (skip_dup ? 3 : 1) : 0;      // bit 0: column-split layouts; bit 1: their duplicate rows are not staged
      k.gx = (int)grid.x; k.gy = (int)grid.y; k.gz = (int)grid.z; k.per_xcd = 0;
      if (xcd_order) {
        const long long total = (long long)grid.x * grid.y * grid.z;
        k.per_xcd = (int)((total + 7) / 8);
        grid = dim3((unsigned)(8 * k.per_xcd), 1, 1);
      }
      if (k.MT == 2) hipLaunchKernelGGL(conv1d_wgrad_kmajor_kernel<2>, grid, dim3(768), (size_t)2 * (3 * 2 * WK_PLANE + WK_OPND), st, k);
      else if (ksplit) hipLaunchKernelGGL((conv1d_wgrad_kmajor_kernel<1, true>), grid, dim3(512), (size_t)WK_NST * WK_STAGE, st, k);
      else hipLaunchKernelGGL(conv1d_wgrad_kmajor_kernel<1>, grid, dim3(512), (size_t)WK_NST * WK_STAGE, st, k);
      const long long n = (long long)C_out * k.NBk * 32;
      const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
      hipLaunchKernelGGL(wgrad_kmajor_reduce_kernel, dim3(blocks), dim3(256), 0, st, k.part, dw, S, C_out, k.NBk, k.K, k.CV);
      return check_launch("conv1d_bwd_weight_split(k-major)");
    }
  }
  FAC_REQUIRE(db == nullptr, "conv1d_bwd_weight_split_db: the fused bias gradient needs the k-major kernel's shapes (>= 16 input channels)");
  WsArgs a;
  int S;
  size_t lds;
  FAC_REQUIRE(ws_geometry(B, C_in, T_in, C_out, T_out, K, stride, dilation, K1, dilation2, &a, &S, &lds) == 0,
              "conv1d_bwd_weight_split: shape not supported (K=%d stride=%d dilation=%d)", K, stride, dilation);
  const long long part_bytes = ws_align((long long)S * C_out * C_in * K * 4);
  FAC_REQUIRE(ws_bytes >= part_bytes + ws_align(3 * a.a_plane_bytes) + ws_align(3 * a.b_plane_bytes),
              "conv1d_bwd_weight_split: workspace too small");
  unsigned char* wsb = reinterpret_cast<unsigned char*>(ws);
  a.part = reinterpret_cast<float*>(ws);
  unsigned char* ap = wsb + part_bytes;
  unsigned char* bp = ap + ws_align(3 * a.a_plane_bytes);
  a.ap = ap; a.bp = bp;
  a.B = B; a.C_in = C_in * a.K2; a.C_out = C_out; a.K = K / a.K2; a.stride = stride; a.dil = dilation;
  const hipStream_t st = (hipStream_t)stream;
  int T_ext;
  long long last = (long long)(T_out - 1) * stride + (long long)(a.K2 - 1) * (a.K2 > 1 ? dilation2 : 0) + (long long)(a.K - 1) * dilation - pad_left;
  {
    int pad_right = last >= T_in ? (int)(last - T_in + 1) : 0;
    int max_pad = pad_left > pad_right ? pad_left : pad_right;
    T_ext = T_in > max_pad ? T_in : max_pad + 1;
  }
  const int T_pad = (int)(pad_left + (last + 1 > T_in ? last + 1 : T_in));       // padded positions that hold (mapped) samples
  {  // dy -> planes with a zero tail;  x -> padded, phase-major planes
    const long long na = (long long)B * C_out * (a.UA / 8), nb = (long long)B * C_in * stride * (a.UB / 8);
    const int ga = (int)((na + 255) / 256 < 65535 * 16 ? (na + 255) / 256 : 65535 * 16);
    const int gb = (int)((nb + 255) / 256 < 65535 * 16 ? (nb + 255) / 256 : 65535 * 16);
    hipLaunchKernelGGL(split_planes_kernel, dim3(ga), dim3(256), 0, st, dy, ap, (long long)B * C_out, T_out, T_out, T_out, 1, a.UA, 0,
                       FAC_PAD_ZERO, a.a_plane_bytes);
    hipLaunchKernelGGL(split_planes_kernel, dim3(gb), dim3(256), 0, st, x, bp, (long long)B * C_in, T_in, T_ext, T_pad, stride, a.UB,
                       pad_left, pad_mode, a.b_plane_bytes);
  }
  dim3 grid((C_out + WS_CO - 1) / WS_CO, (a.C_in + a.cit - 1) / a.cit, S);
  const int nb8 = (a.NCP * 3 * a.R * a.nq + 255) / 256;
  if (nb8 <= 10) ws_launch<10, 3>(a, grid, lds, st);
  else if (nb8 <= 14) ws_launch<14, 3>(a, grid, lds, st);
  else ws_launch<19, 2>(a, grid, lds, st);
  const long long n = (long long)C_out * C_in * K;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.part, dw, S, n);
  return check_launch("conv1d_bwd_weight_split");
}
