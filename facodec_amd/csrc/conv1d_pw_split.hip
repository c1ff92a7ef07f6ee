// The streaming k = 1 kernel of conv1d_pw.hip on the bf16 matrix pipe (round 6): the ResidualUnit tails at C = 64 .. 384
// (dac/model/dac.py:33-42) with both operands split into three bf16 planes (hi + mid + lo == value exactly; six products,
// smallest first, fp32 accumulate -- the "fp32-grade" arithmetic of conv1d_bsplit.hip / conv1d_gemm_split.hip).
// Same skeleton as conv1d_pw.hip -- weights resident in LDS for the life of the workgroup, a wave owns 32 columns x 32 MBW
// output channels at a time, B operand straight from global memory through a register ring, epilogue in the C/D layout with
// bias / Snake / residual / y / y2 -- but:
//   * the weights are split ONCE in the prologue into the A-fragment order of v_mfma_f32_32x32x16_bf16
//     ([K step of 16][32-row block][plane][lane][8 bf16]: one ds_read_b128 per fragment, conflict-free), 6 bytes per weight:
//     96 output channels x 192 inputs = 108 KB, so C = 192 runs as two slices of 96 and C = 256 / 384 as four / six slices of 64
//     (the slices of one column-block set sit on one XCD: the input rows come from HBM once, from that XCD's L2 afterwards);
//   * a K step is 16 input channels: lane (kq, l31) loads x[16 s + 8 kq + j][t0 + l31], j = 0..7 (two full 128-byte lines per
//     instruction, as before), splits the eight values in registers and feeds 6 x MBW MFMAs of 32 cycles -- 2.7 x fewer
//     matrix-pipe cycles than the 8 x MBW fp32 MFMAs of 64 cycles for the same 16 channels (C = 192: 288 fp32 MFMAs per block
//     and wave were 37 us per round of the chip against 33 us of HBM time).
// Measured at B = 32 (profiles/r06_pws_ab.log, fp32 kernel -> this one): C = 64 0.37 -> 0.32 ms, 96 0.63 -> 0.47, 128 0.46 -> 0.34,
// 192 0.87 -> 0.57, 256 0.30 -> 0.21, 384 0.60 -> 0.40; error against fp64 as small as the fp32 kernel's (2 - 5e-7 of the largest value).
#include "conv1d_mfma.h"

namespace fac {

typedef __bf16 pws_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void pws_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// MBW: accumulator blocks per wave; NSPLIT: waves sharing a column block (C_out slice = 32 * MBW * NSPLIT); WAVES per workgroup;
// D: K steps of 16 input channels in flight per wave (C_in % (16 D) == 0)
template <int MBW, int NSPLIT, int WAVES, int D>
__global__ __launch_bounds__(WAVES * 64) void conv1d_pws_kernel(ConvArgs a, int nblk, int n_items) {
  constexpr int CO = 32 * MBW * NSPLIT;
  constexpr int NB = CO / 32;                                  // 32-row blocks of the slice
  extern __shared__ __attribute__((aligned(16))) unsigned char pws_sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kq = lane >> 5;
  const int S16 = a.C_in >> 4;
  const int n_slices = a.C_out / CO;
  const int slice = (blockIdx.x >> 3) % n_slices;
  const int wg = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * n_slices));
  const int n_wg = gridDim.x / n_slices;
  const int co_base = slice * CO;
  float* prm_all = reinterpret_cast<float*>(pws_sm + (size_t)S16 * NB * 3 * 1024);
  {
    const int n_frag = S16 * NB * 64;                          // one (K step, block, lane) = 8 weights = three 16-byte stores
    for (int i = tid; i < n_frag; i += WAVES * 64) {
      const int ln = i & 63, rest = i >> 6;
      const int mb = rest % NB, s = rest / NB;
      const int co = co_base + 32 * mb + (ln & 31);
      const int ci0 = 16 * s + 8 * (ln >> 5);
      pws_bf16x8 h, m, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 a0, a1, a2;
        pws_split3(a.w[(long long)(ci0 + j) * a.C_out_pad + co], a0, a1, a2);
        h[j] = a0; m[j] = a1; l[j] = a2;
      }
      unsigned char* dst = pws_sm + ((size_t)(s * NB + mb) * 3) * 1024 + ln * 16;
      *reinterpret_cast<pws_bf16x8*>(dst) = h;
      *reinterpret_cast<pws_bf16x8*>(dst + 1024) = m;
      *reinterpret_cast<pws_bf16x8*>(dst + 2048) = l;
    }
    for (int i = tid; i < CO; i += WAVES * 64) {
      const int co = co_base + i;
      prm_all[i] = a.bias ? a.bias[co] : 0.f;
      prm_all[CO + i] = a.alpha_out ? a.alpha_out[co] : 0.f;
      prm_all[2 * CO + i] = a.alpha_out ? snake_inv(a.alpha_out[co]) : 0.f;
      prm_all[3 * CO + i] = a.y2 ? a.alpha2[co] : 0.f;
      prm_all[4 * CO + i] = a.y2 ? snake_inv(a.alpha2[co]) : 0.f;
    }
  }
  __syncthreads();
  const int half = NSPLIT == 2 ? (wave & 1) : 0;
  const int co0 = half * 32 * MBW;
  const int mb0 = half * MBW;                                  // first 32-row block of this wave
  const float* prm = prm_all + co0 + 4 * kq;
  const long long xs = a.x_cs;
  const unsigned char* Al = pws_sm + (size_t)mb0 * 3 * 1024 + lane * 16;   // fragment (s, m, p): Al[((s NB + m) 3 + p) 1024]
  const int stride_items = n_wg * (WAVES / NSPLIT);

  auto item_ptr = [&](int it, long long& yoff, bool& ok) -> const float* {
    const int b = it / nblk;
    const int t = (it - b * nblk) * 32 + l31;
    ok = t < a.T_out;
    const int tc = ok ? t : a.T_out - 1;
    yoff = (long long)b * a.y_bs + (long long)(co_base + co0 + 4 * kq) * a.y_cs + tc;
    return a.x + (long long)b * a.x_bs + (long long)(8 * kq) * a.x_cs + tc;
  };

  int item = wg * (WAVES / NSPLIT) + (wave / NSPLIT);
  if (item >= n_items) return;
  long long yoff;
  bool ok;
  const float* xp = item_ptr(item, yoff, ok);
  float xr[D][8];                                              // D K steps (16 D input channels) in flight
#pragma unroll
  for (int q = 0; q < D; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[q][j] = xp[(long long)(16 * q + j) * xs];

  for (;;) {
    f32x16 acc[MBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float* rp = a.res ? a.res + yoff : nullptr;
    float rv[16];                                              // ONE residual block in flight (registers: 3 waves per SIMD)
    auto ld_res = [&](int m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = rp ? rp[(long long)(m * 32 + (r & 3) + 8 * (r >> 2)) * a.y_cs] : 0.f;
    };
    const int nxt = item + stride_items;
    const bool more = nxt < n_items;
    long long yoff_n = yoff;
    bool ok_n = ok;
    const float* xn = more ? item_ptr(nxt, yoff_n, ok_n) : xp;

    for (int s0 = 0; s0 < S16; s0 += D) {
      const bool last = s0 + D >= S16;
      if (last) ld_res(0);                                     // the first residual block rides along with the last trip
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const int s = s0 + q;
        pws_bf16x8 B[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          __bf16 b0, b1, b2;
          pws_split3(xr[q][j], b0, b1, b2);
          B[0][j] = b0; B[1][j] = b1; B[2][j] = b2;
        }
        // refill this slot with K step s + D (the next block's first rows during the last trip)
        const float* src = last ? xn + (long long)(16 * q) * xs : xp + (long long)(16 * (s + D)) * xs;
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[q][j] = src[(long long)j * xs];
        const unsigned char* as = Al + (size_t)s * NB * 3 * 1024;
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
          const pws_bf16x8 A0 = *reinterpret_cast<const pws_bf16x8*>(as + (m * 3 + 0) * 1024);
          const pws_bf16x8 A1 = *reinterpret_cast<const pws_bf16x8*>(as + (m * 3 + 1) * 1024);
          const pws_bf16x8 A2 = *reinterpret_cast<const pws_bf16x8*>(as + (m * 3 + 2) * 1024);
          // smallest terms first (as conv1d_gemm_split.hip): mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[1], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B[0], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B[2], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[0], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B[1], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B[0], acc[m], 0, 0, 0);
        }
      }
    }

    float* yp = a.y ? a.y + yoff : nullptr;
    float* y2p = a.y2 ? a.y2 + yoff : nullptr;
#pragma unroll
    for (int m = 0; m < MBW; ++m) {
      // values first (they consume the residual block), then the next block's residual loads, then this block's stores: the
      // loads are in front of the stores in the memory queue, so waiting for them does not wait for the stores to drain
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2);
        float v = acc[m][r] + prm[row];
        if (a.alpha_out) v = snake_apply(v, prm[CO + row], prm[2 * CO + row]);
        if (a.act != FAC_ACT_NONE) v = apply_act_slow(v, a.act);
        acc[m][r] = v + rv[r];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (m + 1 < MBW) ld_res(m + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m * 32 + (r & 3) + 8 * (r >> 2);
          const float v = acc[m][r];
          if (yp) yp[(long long)row * a.y_cs] = v;
          if (y2p) y2p[(long long)row * a.y_cs] = snake_apply(v, prm[3 * CO + row], prm[4 * CO + row]);
        }
      }
    }
    if (!more) break;
    item = nxt;
    xp = xn;
    yoff = yoff_n;
    ok = ok_n;
  }
}

// ---- the same streaming skeleton for two small-channel layers with TAPS (round 6): the causal ConvTranspose1d 192 -> 96 of the
// decoder's last block (stride 2: two taps, all output phases as rows -- weights of fac_pack_convtr_w_rows) and the encoder's first
// downsampling conv 64 -> 128 (k = 4, stride 2 -- weights of fac_pack_conv_w), dac/model/dac.py:45-66,107-128, plus the data
// gradients of each (which are the other one's shape).  On the tiled split GEMM kernel these ran at 72 / 79 TFLOP/s-eq for
// 1.2 - 1.8 GB of traffic (1.56 / 0.64 ms at B = 32: six 32-channel stages per 128 x 128 tile, then a 128 KB epilogue).  Here the
// contraction index is the VIRTUAL channel v = ci * KT + k (exactly the row order of both packed weight layouts), X_v[u] =
// x[ci][u * S + k - pad_left]; 64 output rows per workgroup slice (C_in * KT <= 384 virtual channels: 147 KB of planes), the
// slices of one column-block set on one XCD as above.  RP = 2: rows are (channel, phase) pairs, phase fastest -- the two phases
// of a channel sit in neighbouring accumulator registers of one lane, so the interleave is an 8-byte store.
template <int KT, int S, int RP, int WAVES, int D>
__global__ __launch_bounds__(WAVES * 64) void conv1d_pwt_kernel(ConvArgs a, int nblk, int n_items) {
  constexpr int MBW = 2, CO = 64, NB = 2;
  constexpr int CPS = 16 / KT;                                 // real channels per K step
  constexpr int CPL = 8 / KT;                                  // real channels per lane and K step
  extern __shared__ __attribute__((aligned(16))) unsigned char pws_sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kq = lane >> 5;
  const int S16 = (a.C_in * KT) >> 4;
  const int n_slices = (a.C_out * RP) / CO;
  const int slice = (blockIdx.x >> 3) % n_slices;
  const int wg = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * n_slices));
  const int n_wg = gridDim.x / n_slices;
  const int row_base = slice * CO;
  float* prm_all = reinterpret_cast<float*>(pws_sm + (size_t)S16 * NB * 3 * 1024);
  {
    const int n_frag = S16 * NB * 64;
    for (int i = tid; i < n_frag; i += WAVES * 64) {
      const int ln = i & 63, rest = i >> 6;
      const int mb = rest % NB, s = rest / NB;
      const int row = row_base + 32 * mb + (ln & 31);
      const int v0 = 16 * s + 8 * (ln >> 5);
      pws_bf16x8 h, m, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 a0, a1, a2;
        pws_split3(a.w[(long long)(v0 + j) * a.C_out_pad + row], a0, a1, a2);
        h[j] = a0; m[j] = a1; l[j] = a2;
      }
      unsigned char* dst = pws_sm + ((size_t)(s * NB + mb) * 3) * 1024 + ln * 16;
      *reinterpret_cast<pws_bf16x8*>(dst) = h;
      *reinterpret_cast<pws_bf16x8*>(dst + 1024) = m;
      *reinterpret_cast<pws_bf16x8*>(dst + 2048) = l;
    }
    for (int i = tid; i < CO; i += WAVES * 64) {
      const int ch = (row_base + i) / RP;
      prm_all[i] = a.bias ? a.bias[ch] : 0.f;
      prm_all[CO + i] = a.y2 ? a.alpha2[ch] : 0.f;
      prm_all[2 * CO + i] = a.y2 ? snake_inv(a.alpha2[ch]) : 0.f;
    }
  }
  __syncthreads();
  const float* prm = prm_all + 4 * kq;
  const long long xs = a.x_cs;
  const unsigned char* Al = pws_sm + lane * 16;
  const int stride_items = n_wg * WAVES;

  // per item: the KT column offsets of this lane's output column (reflected / clamped) and 1 / 0 factors for zero padding
  struct Cols { int off[KT]; float f[KT]; };
  auto item_ptr = [&](int it, long long& yoff, bool& ok, Cols& c) -> const float* {
    const int b = it / nblk;
    const int u = (it - b * nblk) * 32 + l31;
    ok = u < a.T_out;
    const int uc = ok ? u : a.T_out - 1;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int t = uc * S + k - a.pad_left;
      int j;
      if (a.pad_mode == FAC_PAD_REFLECT) j = reflect_index(t, a.T_in, a.T_ext);
      else j = (t >= 0 && t < a.T_in) ? t : -1;
      c.off[k] = j < 0 ? 0 : j;
      c.f[k] = j < 0 ? 0.f : 1.f;
    }
    // row (row_base + 4 kq + local row) of the output; RP == 2: channel (row / 2), columns 2 u + phase
    yoff = (long long)b * a.y_bs + (long long)((row_base + 4 * kq) / RP) * a.y_cs + (long long)uc * RP;
    return a.x + (long long)b * a.x_bs + (long long)(CPL * kq) * xs;
  };

  int item = wg * WAVES + wave;
  if (item >= n_items) return;
  long long yoff;
  bool ok;
  Cols cc;
  const float* xp = item_ptr(item, yoff, ok, cc);
  float xr[D][8];
#pragma unroll
  for (int q = 0; q < D; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[q][j] = xp[(long long)(CPS * q + j / KT) * xs + cc.off[j % KT]];

  for (;;) {
    f32x16 acc[MBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const int nxt = item + stride_items;
    const bool more = nxt < n_items;
    long long yoff_n = yoff;
    bool ok_n = ok;
    Cols cn = cc;
    const float* xn = more ? item_ptr(nxt, yoff_n, ok_n, cn) : xp;

    for (int s0 = 0; s0 < S16; s0 += D) {
      const bool last = s0 + D >= S16;
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const int s = s0 + q;
        pws_bf16x8 B[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          __bf16 b0, b1, b2;
          pws_split3(xr[q][j] * cc.f[j % KT], b0, b1, b2);
          B[0][j] = b0; B[1][j] = b1; B[2][j] = b2;
        }
        if (last) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xr[q][j] = xn[(long long)(CPS * q + j / KT) * xs + cn.off[j % KT]];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) xr[q][j] = xp[(long long)(CPS * (s + D) + j / KT) * xs + cc.off[j % KT]];
        }
        const unsigned char* as = Al + (size_t)s * NB * 3 * 1024;
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
          const pws_bf16x8 A0 = *reinterpret_cast<const pws_bf16x8*>(as + (m * 3 + 0) * 1024);
          const pws_bf16x8 A1 = *reinterpret_cast<const pws_bf16x8*>(as + (m * 3 + 1) * 1024);
          const pws_bf16x8 A2 = *reinterpret_cast<const pws_bf16x8*>(as + (m * 3 + 2) * 1024);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[1], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B[0], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B[2], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[0], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B[1], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B[0], acc[m], 0, 0, 0);
        }
      }
    }

    if (ok) {
      float* yp = a.y ? a.y + yoff : nullptr;
      float* y2p = a.y2 ? a.y2 + yoff : nullptr;
#pragma unroll
      for (int m = 0; m < MBW; ++m) {
        if (RP == 2) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2);   // even: phases 0 / 1 of channel (row_base + 4 kq + row) / 2
            const float v0 = acc[m][r] + prm[row], v1 = acc[m][r + 1] + prm[row + 1];
            const long long o = (long long)(row >> 1) * a.y_cs;
            if (yp) *reinterpret_cast<float2*>(yp + o) = make_float2(v0, v1);
            if (y2p)
              *reinterpret_cast<float2*>(y2p + o) = make_float2(snake_apply(v0, prm[CO + row], prm[2 * CO + row]),
                                                                snake_apply(v1, prm[CO + row + 1], prm[2 * CO + row + 1]));
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2);
            const float v = acc[m][r] + prm[row];
            if (yp) yp[(long long)row * a.y_cs] = v;
            if (y2p) y2p[(long long)row * a.y_cs] = snake_apply(v, prm[CO + row], prm[2 * CO + row]);
          }
        }
      }
    }
    if (!more) break;
    item = nxt;
    xp = xn;
    yoff = yoff_n;
    ok = ok_n;
    cc = cn;
  }
}

// C -> output channels per weight slice (the slice's three planes, 6 bytes per weight, must fit the LDS)
static int pws_slice_channels(int C) {
  switch (C) {
    case 64: case 96: case 128: return C;
    case 192: return 96;
    case 256: case 384: return 64;
    default: return 0;
  }
}

bool conv_pws_ok(const ConvArgs& a) {
  static const bool on = !(getenv("FAC_PW_SPLIT") && getenv("FAC_PW_SPLIT")[0] == '0');
  if (!on || !(a.K == 1 && a.stride == 1 && a.n_phase == 1 && a.phase_shift == 0 && a.y_tstride == 1 && a.pad_left == 0 && !a.alpha_in &&
               !a.w1 && !a.w_batched && !conv_two_level(a) && a.T_in >= a.T_out && a.C_in == a.C_out && !a.x_p8))
    return false;
  const int co = pws_slice_channels(a.C_out);
  if (!co || a.C_out_pad != a.C_out) return false;
  // enough column blocks that every wave slot of a 256-CU chip walks at least two of them
  const long long items = (long long)a.B * ((a.T_out + 31) / 32);
  const int nsplit = co == 128 ? 2 : 1;
  const long long slots = (256 / (a.C_out / co)) * (12 / nsplit);
  return items >= 2 * slots;
}

template <int MBW, int NSPLIT, int WAVES, int D>
static int pws_launch(ConvArgs& a, hipStream_t s) {
  const int nblk = (a.T_out + 31) / 32;
  const long long n_items = (long long)a.B * nblk;
  if (n_items > 0x7fffffffll) {
    set_error("conv1d(pointwise, split): too many column blocks (%lld)", n_items);
    return FAC_ERR_ARG;
  }
  constexpr int CO = 32 * MBW * NSPLIT;
  const size_t lds = (size_t)(a.C_in / 16) * (CO / 32) * 3 * 1024 + 5 * CO * sizeof(float);
  auto kern = conv1d_pws_kernel<MBW, NSPLIT, WAVES, D>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  constexpr int per_wg = WAVES / NSPLIT;
  const int n_slices = a.C_out / CO;
  long long groups = 256 / (8 * n_slices);
  const long long need = (n_items + 8 * per_wg - 1) / (8 * per_wg);
  if (groups > need) groups = need;
  if (groups < 1) groups = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * 8 * n_slices)), dim3(WAVES * 64), lds, s, a, nblk, (int)n_items);
  return check_launch("conv1d_pw_split");
}

// 12 waves (three per SIMD, <= 168 registers; 16 spill, 8 lose 10 - 30 %) and 4 / 3 K steps in flight per wave (2 / 3 accumulator
// blocks), measured on the forward's six layer shapes: profiles/r06_pws_ab.log
int conv_dispatch_pws(ConvArgs& a, hipStream_t s) {
  switch (pws_slice_channels(a.C_out)) {
    case 64: return pws_launch<2, 1, 12, 4>(a, s);      // C = 64; C = 256 / 384 in slices of 64
    case 128: return pws_launch<2, 2, 12, 4>(a, s);
    default: return pws_launch<3, 1, 12, 3>(a, s);      // C = 96; C = 192 in two slices of 96
  }
}

// the two tap shapes of conv1d_pwt_kernel: 1 = all-phases ConvTranspose1d with stride 2 (K = 2, row_phases = 2), 2 = k = 4 stride-2 conv
static int pwt_shape(const ConvArgs& a) {
  if (a.K == 2 && a.stride == 1 && a.rp == 2 && a.pad_left == 1 && a.pad_mode == FAC_PAD_ZERO) return 1;
  if (a.K == 4 && a.stride == 2 && a.rp == 1) return 2;
  return 0;
}

bool conv_pwt_ok(const ConvArgs& a) {
  static const bool on = !(getenv("FAC_PW_TAPS") && getenv("FAC_PW_TAPS")[0] == '0') &&
                         !(getenv("FAC_PW_SPLIT") && getenv("FAC_PW_SPLIT")[0] == '0');
  if (!on || !pwt_shape(a) || !a.w || !a.x) return false;
  if (!(a.dil == 1 && a.n_phase == 1 && a.phase_shift == 0 && a.y_tstride == 1 && !a.alpha_in && !a.alpha_out && a.act == FAC_ACT_NONE &&
        !a.res && !a.w1 && !a.w_batched && !conv_two_level(a) && !a.x_p8 && !a.y2_p8))
    return false;
  const int kv = a.C_in * a.K, rows = a.C_out * a.rp;
  if (kv % 64 != 0 || kv > 384 || rows % 64 != 0 || rows > a.C_out_pad) return false;
  if (a.rp == 2 && ((a.y_cs | a.y_bs) & 1)) return false;      // 8-byte stores
  const long long items = (long long)a.B * ((a.T_out + 31) / 32);
  return items >= 2 * (256 / (rows / 64)) * 12;
}

template <int KT, int S, int RP, int D>
static int pwt_launch(ConvArgs& a, hipStream_t s) {
  constexpr int WAVES = 12;
  const int nblk = (a.T_out + 31) / 32;
  const long long n_items = (long long)a.B * nblk;
  if (n_items > 0x7fffffffll) {
    set_error("conv1d(streaming, taps): too many column blocks (%lld)", n_items);
    return FAC_ERR_ARG;
  }
  const size_t lds = (size_t)(a.C_in * KT / 16) * 2 * 3 * 1024 + 3 * 64 * sizeof(float);
  auto kern = conv1d_pwt_kernel<KT, S, RP, WAVES, D>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const int n_slices = a.C_out * RP / 64;
  long long groups = 256 / (8 * n_slices);
  const long long need = (n_items + 8 * WAVES - 1) / (8 * WAVES);
  if (groups > need) groups = need;
  if (groups < 1) groups = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * 8 * n_slices)), dim3(WAVES * 64), lds, s, a, nblk, (int)n_items);
  return check_launch("conv1d_pw_taps");
}

int conv_dispatch_pwt(ConvArgs& a, hipStream_t s) {
  if (pwt_shape(a) == 1) return pwt_launch<2, 1, 2, 2>(a, s);
  return pwt_launch<4, 2, 1, 2>(a, s);
}

}  // namespace fac
