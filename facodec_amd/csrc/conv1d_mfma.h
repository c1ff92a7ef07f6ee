// K1-K4: Conv1d / polyphase ConvTranspose1d as an implicit GEMM on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain), with the Snake prologue,
// bias, Snake / tanh epilogue and residual add fused in.
//
// Replaces (reference, /root/reference): SConv1d.forward dac/model/encodec.py:212-228,
// SConvTranspose1d.forward :248-270, snake dac/nn/layers.py:18-24, ResidualUnit dac/model/dac.py:25-42.
//
// GEMM view per batch element:  Y[co, t] = sum_{ci,k} Wp[ci][k][co] * Xp[ci, t*stride + k*dil - pad]
//   M = C_out (A operand: packed weights, co fastest -> conflict-free ds_read_b32)
//   N = T_out (B operand: the LDS-staged receptive-field tile, time fastest)
//   K = C_in*K taps, consumed two input channels at a time (the 32x32x2 k-pair = lanes 0-31 / 32-63)
//
// One workgroup = 8 waves owns a CO_TILE x T_TILE output tile of one (batch, phase) and walks C_in
// in chunks of `cic` channels.  The waves are specialised (a wave64 is in-order: its own VALU / VMEM
// work would stall its MFMA stream, while a DIFFERENT wave's does not -- separate pipes per SIMD):
//   * waves 0-3 ("MFMA waves") only read fragments from LDS and issue matrix instructions;
//   * waves 4-7 ("staging waves") fill the other LDS stage meanwhile: the weight slab
//     [cic][K][CO_TILE] by LDS-DMA (global_load_lds, 16 B/lane, no VGPR round trip) and the input
//     slab [cic][XW] (receptive field incl. halo, reflect/zero padded) through registers, with
//     Snake applied on the way in (once per staged element);
//   * LDS is double-buffered: one workgroup barrier per chunk.
// The tap loop is a compile-time unroll (template KT) so every LDS read has an immediate offset
// and the compiler can run the ds_reads ahead of the MFMAs.
#pragma once
#include "common.h"

namespace fac {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int CONV_XMAX = 6;   // staged input float4s held in registers per staging thread

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* alpha_in;
  const float* alpha_out;
  const float* res;
  float* y;
  float* y2;            // optional pre-activated copy: snake(y, alpha2)
  const float* alpha2;
  const float* w1;      // fused ResidualUnit: packed 1x1 weights (C, 1, C) applied to snake(conv + bias)
  const float* bias1;
  long long x_bs, x_cs, y_bs, y_cs, w_bs;
  int B, C_in, T_in, T_ext, C_out, C_out_pad, T_out;
  int K, stride, dil, pad_left, pad_mode;
  int n_phase, y_tstride, act, w_batched;
  int phase_shift;   // left trim of a non-causal transposed conv (0 = causal)
  int K1, dil2;      // two-level taps: tap k = k2*K1 + k1 at offset k2*dil2 + k1*dil (K1 == K: plain conv)
  // Two-level taps run as K2v = K / K1 VIRTUAL input channels per real one: virtual channel v = ci*K2v + k2 is row ci read
  // k2*dil2 columns further on, with KV = K1 taps.  The packed weights need no change: row (ci*K + k2*K1 + k1) of
  // (C_in_pad, K, C_out_pad) IS row (v*K1 + k1).  The stage then holds only the K1-tap receptive field of each virtual
  // channel instead of the whole K2-row span (3.2x fewer staged columns for the (3, 9) convs of the spectrogram
  // discriminator at pitch 272), and the tap loop is the compile-time one.  Plain convs: K2v = 1, KV = K, CV = C_in.
  int K2v, KV, CV;
  int cic;  // input channels per LDS stage (multiple of 2*UC)
  int XW;   // staged input width = (T_TILE-1)*stride + (K-1)*dil + 1
  int XB;   // 64-wide column blocks per staged row = ceil(XW/64)
  int XQ, XR;  // 4 / XB, 4 % XB: (row, block) advance of one wave per staging iteration
  int n_t_tiles;
  int n_tiles, persist;   // conv1d_bsplit.hip: tiles of the launch; 1 = workgroups walk several tiles as one chunk stream
#if defined(FAC_PROF) || defined(FAC_PROF2)
  unsigned long long* dbg;   // per-workgroup cycle counters (tuning builds only)
#endif
  int x_off;   // columns staged to the left of the receptive field so that the slab starts 16-B aligned
  int gflat;   // conv1d_gemm_split.hip: columns are the flattened (clip, time) index (K = 1)
  int grt;     // conv1d_gemm_split.hip: > 0 = number of row tiles, and the row tile is the FASTEST index of the logical order
  const unsigned char* x_p8;   // input as P8 planes (fac_conv_desc.x_p8) or NULL
  long long x_p8_ps;           // bytes between planes
  unsigned char* y2_p8;        // second output as P8 planes or NULL
  long long y2_p8_ps;
  int rp;      // > 1: output rows are (channel, phase) pairs, phase fastest, CO_TILE / rp channels per tile (all-phases
               // ConvTranspose1d, fac_conv_desc.row_phases); the all-waves epilogue interleaves them into contiguous runs
};

// largest tap offset of a (possibly two-level) conv
__host__ __device__ inline int conv_max_tap_offset(const ConvArgs& a) {
  const int K1 = (a.K1 > 0 && a.K1 < a.K) ? a.K1 : a.K;
  return (a.K / K1 - 1) * a.dil2 + (K1 - 1) * a.dil;
}
__host__ __device__ inline bool conv_two_level(const ConvArgs& a) { return a.K1 > 0 && a.K1 < a.K; }
// fills K2v / KV / CV from K, K1 (call once the descriptor fields are set)
inline void conv_set_virtual(ConvArgs& a) {
  a.KV = conv_two_level(a) ? a.K1 : a.K;
  a.K2v = a.K / a.KV;
  a.CV = a.C_in * a.K2v;
}

template <int KT>
struct ConvUnroll {
  static constexpr int UC = (KT == 1) ? 4 : ((KT == 2 || KT == 3) ? 2 : 1);  // channel pairs per unrolled body
  // input channels per LDS stage when the tap count is a compile-time constant (0: chosen at run time)
  static constexpr int CIC = KT == 1 ? 24 : (KT == 2 || KT == 3) ? 16 : (KT >= 4 && KT <= 7) ? 8 : (KT > 7 ? 4 : 0);
};

// Channels per stage of the compile-time tap loops.  On the 256-column tiles a staged row of a 2- or 3-tap conv is two 64-lane
// float4 blocks wide, and 16 channels would need 32 (row, block) register slots per staging wave where CONV_XMAX gives 24 --
// the launch then fell back to the run-time tap loop (no fragment prefetch ring): 8 channels per stage there.
template <int KT, int T_TILE>
constexpr int conv_cic() {
  return ((KT == 2 || KT == 3) && T_TILE >= 256) ? 8 : ConvUnroll<KT>::CIC;
}

#ifndef FAC_CONV_WPE
#define FAC_CONV_WPE 4
#endif
template <int MB, int NB, int WM, int WN, int KT, bool FUSE = false>
__global__ __launch_bounds__((WM * WN + 4) * 64, (WM * WN == 4 ? FAC_CONV_WPE : 3)) void conv1d_mfma_kernel(ConvArgs a) {
  constexpr int NMW = WM * WN;   // MFMA waves (4 or 8); 4 staging waves follow them
  constexpr int CO_TILE = 32 * MB * WM;
  constexpr int T_TILE = 32 * NB * WN;
  constexpr int CO4 = CO_TILE / 4;
  // 8-MFMA-wave tiles keep one workgroup per CU, so nothing overlaps their epilogue: all 12 waves run it, from an
  // accumulator tile in LDS, 16 bytes per lane along time (see the split kernel, conv1d_bsplit.hip).
  constexpr bool ALLW = (WM * WN == 8) && !FUSE;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // [0, NMW): MFMA waves, then 4 staging waves

  // Work decode.  Workgroups are dispatched round-robin over the 8 XCDs (observed: block i -> XCD
  // i % 8, speed only, never correctness): remap so that each XCD walks a CONTIGUOUS range of the tile
  // list ordered (co-tile slowest, then batch/phase, then time tile).  The workgroups resident on one
  // XCD then share one C_out tile, i.e. stream the same weight slabs through that XCD's private L2.
  int t0, co0, b, phase;
  {
    const int n = gridDim.x;
    const int q8 = n >> 3, r8 = n & 7;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int nt = a.n_t_tiles, nbp = a.B * a.n_phase;
    const int tt = id % nt;
    const int rest = id / nt;
    const int bp = rest % nbp;
    const int ct = rest / nbp;
    t0 = tt * T_TILE;
    co0 = ct * CO_TILE;
    b = bp / a.n_phase;
    phase = bp - b * a.n_phase;
  }
  // Output offset of this polyphase component and its input shift: phases below the left trim of a
  // non-causal transposed conv land one output period later and read x[t], x[t+1] (one tap later).
  // (the staged slab is the same for every phase -- one column wider -- and only the fragment read
  // offset moves, so the slab stays 16-byte aligned for the LDS-DMA)
  int y_off = phase - a.phase_shift, tap_shift = 0;
  if (y_off < 0) {
    y_off += a.n_phase;
    tap_shift = 1;
  }

  const int K = KT > 0 ? KT : a.KV;       // taps per (virtual) input channel
  const int cic = KT > 0 ? conv_cic<KT, T_TILE>() : a.cic;
  const int XW = a.XW, XB = a.XB;
  const int w_stage = cic * K * CO_TILE;  // floats
  const int x_stage = cic * XW;
  float* Wbuf = smem;                 // [2][cic][K][CO_TILE]
  float* Xbuf = smem + 2 * w_stage;   // [2][cic][XW]
  const int n_chunks = (a.CV + cic - 1) / cic;

  if (wave >= NMW) {
    // ===================== staging waves: HBM/L2 -> LDS for chunk c+1 while chunk c is multiplied
    const int lw = wave - NMW;
    // Both roles share each SIMD's VALU issue port, and a pending MFMA of an older / higher-priority
    // wave blocks it: at equal priority the staging waves got ~1 VALU slot per MFMA (measured: 12k
    // cycles to ISSUE one chunk's loads; MFMA waves then idled 47 % of their life at the barrier).
    // The staging stream is short (a few hundred instructions per chunk), so it runs at raised priority
    // and the MFMA waves absorb the few lost slots.
#ifndef FAC_ABL_NOPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const float* xg = a.x + (long long)b * a.x_bs;
    const int w_rows_total = cin_pad_dev(a.C_in) * a.K;   // rows C_in*K.. are zero (fac_pack_conv_w); a.K = K2v * K
    const float* wg = a.w + (long long)phase * w_rows_total * a.C_out_pad +
                      (a.w_batched ? (long long)b * a.w_bs : 0ll);
    const int tin0 = t0 * a.stride - a.pad_left - a.x_off;   // multiple of 4 by construction
    // float4 global loads need 16-B aligned rows: base pointer, batch and channel strides
    const bool vec_ok = ((a.x_cs | a.x_bs) & 3) == 0 && ((reinterpret_cast<unsigned long long>(a.x) & 15) == 0);
    const int r_first = lw / XB, cb_first = lw - r_first * XB;
    // pure-DMA input staging: no Snake prologue, 16-B aligned rows, and the whole slab inside the
    // signal (no reflection / zero padding / ragged tail in this tile)
    const int K2v = a.K2v, dil2 = a.dil2;
    const bool x_dma = a.alpha_in == nullptr && vec_ok && tin0 >= 0 && tin0 + (K2v - 1) * dil2 + XW <= a.T_in &&
                       (K2v == 1 || (dil2 & 3) == 0);

#ifdef FAC_PROF
    unsigned long long lt_issue = 0, lt_wait = 0, lt_store = 0, lt_bar = 0;
#endif
    auto stage = [&](int chunk, int buf) {
#ifdef FAC_PROF
      const unsigned long long q0 = __builtin_readcyclecounter();
#endif
      // weight slab by LDS-DMA: flat float4 index q -> (row, col4); 64 lanes = 1 KiB contiguous in
      // LDS.  The packed buffer carries ZERO rows up to fac_cin_pad(C_in) channels, so a partially
      // filled last stage multiplies zeros whatever the input slab holds there; columns past
      // C_out_pad are clamped to valid weights and only feed output rows that are never stored.
#ifndef FAC_ABL_NOW
      {
        const int n4 = cic * K * CO4;
        const int row_base = chunk * cic * K;
        float* dst0 = Wbuf + buf * w_stage;
        for (int i = lw; i * 64 < n4; i += 4) {
          const int q = i * 64 + lane;
          if (q < n4) {
            const int row = q / CO4;
            const int c4 = q - row * CO4;
            int grow = row_base + row;
            grow = grow < w_rows_total ? grow : w_rows_total - 1;
            int co = co0 + 4 * c4;
            co = co < a.C_out_pad ? co : a.C_out_pad - 4;
            const float* src = wg + (long long)grow * a.C_out_pad + co;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst0 + i * 256), 16, 0, 0);
          }
        }
      }
#endif
#ifdef FAC_ABL_NOX
      return;
#endif
      // input slab: slot j of this wave covers row r, float4 column cb*64 + lane; all loads are
      // issued first, then Snake is applied on the way into LDS (once per staged element).
      // Interior + 16-B aligned rows move as float4; edges (reflection / zero padding / ragged ends)
      // fall back to per-element indexing.
      const int ci0 = chunk * cic;
      const int XW4 = XW >> 2;
      if (x_dma) {
        // Interior tile of an input that needs no Snake / padding: the slab rows go straight
        // HBM/L2 -> LDS by LDS-DMA as well, one 16-B piece per lane -- no VGPRs and (almost) no
        // VALU work next to the MFMAs.  Channels past C_in are clamped (their weights are zero).
        float* dst = Xbuf + buf * x_stage;
        for (int it = lw; it < cic * XB; it += 4) {
          const int r = it / XB;
          const int c4 = (it - r * XB) * 64 + lane;
          if (c4 < XW4) {
            int v = ci0 + r;
            v = v < a.CV ? v : a.CV - 1;
            const int ci = v / K2v, k2 = v - ci * K2v;
            const float* src = xg + (long long)ci * a.x_cs + tin0 + k2 * dil2 + 4 * c4;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst + r * XW + (it - r * XB) * 256), 16, 0, 0);
          }
        }
#ifdef FAC_PROF
        lt_issue += __builtin_readcyclecounter() - q0;
#endif
        return;
      }
      float4 xr[CONV_XMAX];
      float xal[CONV_XMAX];
      {
        int r = r_first, cb = cb_first;
#pragma unroll
        for (int j = 0; j < CONV_XMAX; ++j) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          float al = 0.f;
          const int c4 = cb * 64 + lane;
          const int vch = ci0 + r;
          const int ci = vch / K2v, k2 = vch - ci * K2v;
          if (r < cic && c4 < XW4 && vch < a.CV) {
            const int tin = tin0 + 4 * c4 + k2 * dil2;
            const float* xrow = xg + (long long)ci * a.x_cs;
            if (vec_ok && (tin & 3) == 0 && tin >= 0 && tin + 3 < a.T_in) {
              v = *reinterpret_cast<const float4*>(xrow + tin);
            } else {
              float e[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                int idx;
                if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin + u, a.T_in, a.T_ext);
                else idx = (tin + u >= 0 && tin + u < a.T_in) ? tin + u : -1;
                e[u] = idx >= 0 ? xrow[idx] : 0.f;
              }
              v = make_float4(e[0], e[1], e[2], e[3]);
            }
            if (a.alpha_in != nullptr) al = a.alpha_in[ci];
          }
          xr[j] = v;
          xal[j] = al;
          r += a.XQ;
          cb += a.XR;
          if (cb >= XB) { cb -= XB; ++r; }
        }
      }
#ifdef FAC_PROF
      const unsigned long long q1 = __builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long q2 = __builtin_readcyclecounter();
      lt_issue += q1 - q0;
      lt_wait += q2 - q1;
#endif
      {
        float* dst = Xbuf + buf * x_stage;
        int r = r_first, cb = cb_first;
#pragma unroll
        for (int j = 0; j < CONV_XMAX; ++j) {
          const int c4 = cb * 64 + lane;
          if (r < cic && c4 < XW4) {
            float4 v = xr[j];
            if (a.alpha_in != nullptr) {   // snake(0) == 0 keeps the zero fill
              const float al = xal[j], inv = snake_inv(al);
              v.x = snake_apply(v.x, al, inv);
              v.y = snake_apply(v.y, al, inv);
              v.z = snake_apply(v.z, al, inv);
              v.w = snake_apply(v.w, al, inv);
            }
            *reinterpret_cast<float4*>(dst + r * XW + 4 * c4) = v;
          }
          r += a.XQ;
          cb += a.XR;
          if (cb >= XB) { cb -= XB; ++r; }
        }
      }
#ifdef FAC_PROF
      lt_store += __builtin_readcyclecounter() - q2;
#endif
    };

    stage(0, 0);
    __syncthreads();   // the barrier's release also drains the LDS-DMA (vmcnt(0))
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
#ifndef FAC_ABL_NOSTAGE
      if (chunk + 1 < n_chunks) stage(chunk + 1, (chunk & 1) ^ 1);
#endif
#ifdef FAC_PROF
      const unsigned long long qb = __builtin_readcyclecounter();
#endif
#ifndef FAC_ABL_NOBAR
      __syncthreads();
#endif
#ifdef FAC_PROF
      lt_bar += __builtin_readcyclecounter() - qb;
#endif
    }
    if constexpr (FUSE) {
      // fused ResidualUnit: every stage buffer is free now -- DMA the whole 1x1 weight matrix
      // [C][CO_TILE] over them while the MFMA waves apply bias + Snake to their accumulators
      const int n4 = CO_TILE * CO4;
      for (int i = lw; i * 64 < n4; i += 4) {
        const int q = i * 64 + lane;
        if (q < n4) {
          const int row = q / CO4;
          const int c4 = q - row * CO4;
          const float* src = a.w1 + (long long)row * a.C_out_pad + 4 * c4;
          __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + i * 256), 16, 0, 0);
        }
      }
      __syncthreads();
    }
#ifdef FAC_PROF
    if (a.dbg && lane == 0) {
      unsigned long long* d = a.dbg + (1ll << 21) + ((long long)blockIdx.x * 4 + lw) * 4;
      d[0] = lt_issue; d[1] = lt_wait; d[2] = lt_store; d[3] = lt_bar;
    }
#endif
    if constexpr (!ALLW) return;
    __builtin_amdgcn_s_setprio(0);
  }

#ifdef FAC_PROF
  unsigned long long pf_first = 0, pf_bar = 0, pf_loop = 0;
  const unsigned long long pf_start = __builtin_readcyclecounter();
#endif
  if (wave < NMW) {
  // ========================= MFMA waves
  const int l31 = lane & 31;
  const int kq = lane >> 5;
  const int wm = wave / WN;
  const int wn = wave % WN;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int a_off = wm * (MB * 32) + l31;                 // column inside the weight row
  const int b_off = (wn * (NB * 32) + l31) * a.stride + a.x_off + tap_shift;   // time offset inside the input row
  const int wrow_stride = K * CO_TILE;                    // floats per input channel in Wbuf
  const int dil = a.dil;
  const int nstride = 32 * a.stride;

#ifdef FAC_PROF
  unsigned long long t_start = __builtin_readcyclecounter(), t_bar = 0, t_first = 0;
#endif
  __syncthreads();   // chunk 0 staged
#ifdef FAC_PROF
  t_first = __builtin_readcyclecounter() - t_start;
#endif
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const int buf = chunk & 1;
    const float* Wb = Wbuf + buf * w_stage + a_off + kq * wrow_stride;
    const float* Xb = Xbuf + buf * x_stage + b_off + kq * XW;
    if constexpr (KT > 0) {
      // Straight-line chunk: P = CIC/2*KT (channel pair, tap) positions of MB*NB MFMAs each.  The
      // fragments of position p+2 are requested from LDS before the MFMAs of position p are issued
      // (3-deep register ring, static indices), so ds_read latency never stalls the matrix pipe.
      constexpr int CICc = conv_cic<KT, T_TILE>();
      constexpr int P = CICc / 2 * KT;
      float av[3][MB], bv[3][NB];
      auto ldfrag = [&](int pos, float* avp, float* bvp) {
        const int c2 = (pos / KT) * 2, kk = pos % KT;
#ifdef FAC_ABL_NOLDS
#pragma unroll
        for (int m = 0; m < MB; ++m) avp[m] = (float)(pos + m) * 1e-3f + (float)lane;
#pragma unroll
        for (int n = 0; n < NB; ++n) bvp[n] = (float)(pos - n) * 1e-3f;
#else
#pragma unroll
        for (int m = 0; m < MB; ++m) avp[m] = Wb[c2 * wrow_stride + kk * CO_TILE + m * 32];
#pragma unroll
        for (int n = 0; n < NB; ++n) bvp[n] = Xb[c2 * XW + kk * dil + n * nstride];
#endif
      };
      ldfrag(0, av[0], bv[0]);
      if constexpr (P > 1) ldfrag(1, av[1], bv[1]);
#pragma unroll
      for (int pos = 0; pos < P; ++pos) {
        if (pos + 2 < P) ldfrag(pos + 2, av[(pos + 2) % 3], bv[(pos + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads two positions ahead of their MFMAs
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n)
#ifdef FAC_ABL_NOMFMA
            acc[m][n][pos & 15] += av[pos % 3][m] * bv[pos % 3][n];
#else
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pos % 3][m], bv[pos % 3][n], acc[m][n], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      for (int c2 = 0; c2 < cic; c2 += 2) {
        const float* wq = Wb + c2 * wrow_stride;
        const float* xp = Xb + c2 * XW;
#pragma unroll 2
        for (int kk = 0; kk < K; ++kk) {
          float av[MB], bv[NB];
#pragma unroll
          for (int m = 0; m < MB; ++m) av[m] = wq[kk * CO_TILE + m * 32];
#pragma unroll
          for (int n = 0; n < NB; ++n) bv[n] = xp[kk * dil + n * nstride];
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
#ifdef FAC_PROF
    const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
#ifndef FAC_ABL_NOBAR
    __syncthreads();
#endif
#ifdef FAC_PROF
    t_bar += __builtin_readcyclecounter() - tb0;
#endif
  }
#ifdef FAC_PROF
  const unsigned long long t_loop = __builtin_readcyclecounter() - t_start;
  pf_first = t_first; pf_bar = t_bar; pf_loop = __builtin_readcyclecounter() - pf_start;
#endif

  if constexpr (ALLW) {
    constexpr int EP = T_TILE + 4;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          smem[(wm * (MB * 32) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq) * EP + wn * (NB * 32) + n * 32 + l31] = acc[m][n][r];
  } else {
  // ---- epilogue: C/D layout col = lane&31 (time), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (co)
  float* yg = a.y ? a.y + (long long)b * a.y_bs : nullptr;
  float* y2g = a.y2 ? a.y2 + (long long)b * a.y_bs : nullptr;
  const float* rg = a.res ? a.res + (long long)b * a.y_bs : nullptr;
  // One 32-row block `blk` (block index m of this wave) -> bias, activation, residual, stores.
  // Row groups g = rows 8g..8g+3 (+4 for the upper half-wave).  Every load a group needs (bias, Snake
  // alphas, residuals -- y may alias res, so the compiler will not hoist residual loads above stores
  // by itself) is issued one group AHEAD of its math and stores (2-deep register ring, static
  // indices): one exposed memory round trip per block instead of one per group (measured 30k cycles
  // of epilogue on the k=1 residual layers before).
  auto emit_block = [&](const f32x16 (&blk)[NB], int m, const float* bias, const float* alpha_out, int act) {
    float bsv[2][4], alv[2][4], al2[2][4], rv[2][4][NB];
    auto ld_group = [&](int g, int slot) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = co0 + wm * (MB * 32) + m * 32 + i + 8 * g + 4 * kq;
        const int cc = co < a.C_out ? co : a.C_out - 1;
        bsv[slot][i] = bias ? bias[cc] : 0.f;
        alv[slot][i] = alpha_out ? alpha_out[cc] : 0.f;
        al2[slot][i] = y2g ? a.alpha2[cc] : 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const int t = t0 + wn * (NB * 32) + n * 32 + l31;
          rv[slot][i][n] = (rg && co < a.C_out && t < a.T_out)
                               ? rg[(long long)co * a.y_cs + (long long)t * a.y_tstride + y_off] : 0.f;
        }
      }
    };
    ld_group(0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int slot = g & 1;
      if (g + 1 < 4) ld_group(g + 1, slot ^ 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        const int co = co0 + wm * (MB * 32) + m * 32 + i + 8 * g + 4 * kq;
        if (co >= a.C_out) continue;
        const float al = alv[slot][i];
        const float inv = alpha_out ? snake_inv(al) : 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const int t = t0 + wn * (NB * 32) + n * 32 + l31;
          if (t >= a.T_out) continue;
          float v = blk[n][r] + bsv[slot][i];
          if (alpha_out) v = snake_apply(v, al, inv);
          if (act != FAC_ACT_NONE) v = apply_act_slow(v, act);
          v += rv[slot][i][n];
          const long long o = (long long)co * a.y_cs + (long long)t * a.y_tstride + y_off;
          if (yg) yg[o] = v;
          if (y2g) y2g[o] = snake_apply(v, al2[slot][i], snake_inv(al2[slot][i]));
        }
      }
    }
  };

  if constexpr (!FUSE) {
#pragma unroll
    for (int m = 0; m < MB; ++m) emit_block(acc[m], m, a.bias, a.alpha_out, a.act);
  } else {
    // ---- fused ResidualUnit tail (dac/model/dac.py:33-34,38-42): h = snake(conv7 + b7, alpha2) stays in
    // the accumulators; in the C/D layout register r of a 32x32 tile holds rows R0(r) (lanes 0-31) and
    // R0(r)+4 (lanes 32-63) of column lane&31 -- exactly an MFMA B fragment for the k-pair
    // (R0(r), R0(r)+4).  So the 1x1 conv y = W1 h runs straight out of the registers (A = W1 from LDS),
    // and h never travels to LDS or HBM.  Needs every channel in one wave: tiles with WM == 1.
    static_assert(WM == 1, "fused ResidualUnit needs all channels of a column in one wave");
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float bs[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = m * 32 + i + 8 * g + 4 * kq;
          bs[i] = a.bias ? a.bias[c] : 0.f;
          al[i] = a.alpha_out[c];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float inv = snake_inv(al[i]);
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[m][n][4 * g + i] = snake_apply(acc[m][n][4 * g + i] + bs[i], al[i], inv);
        }
      }
    }
    __syncthreads();   // W1 landed in LDS (staged by the staging waves meanwhile)
    const float* W1s = smem + 4 * kq * CO_TILE + l31;   // [c][CO_TILE]; this lane's k row offset and column
#pragma unroll
    for (int cb = 0; cb < MB; ++cb) {
      f32x16 acc2[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[n][r] = 0.f;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c0r = m * 32 + (r & 3) + 8 * (r >> 2);
          const float av1 = W1s[c0r * CO_TILE + cb * 32];
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc2[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, acc[m][n][r], acc2[n], 0, 0, 0);
        }
      }
      emit_block(acc2, cb, a.bias1, nullptr, FAC_ACT_NONE);
    }
  }
#ifdef FAC_PROF
  if (a.dbg && lane == 0) {
    unsigned long long* d = a.dbg + ((long long)blockIdx.x * 4 + (wave & 3)) * 4;
    d[0] = t_first; d[1] = t_bar; d[2] = t_loop; d[3] = __builtin_readcyclecounter() - t_start;
  }
#endif
  }   // !ALLW epilogue
  }   // MFMA waves

  if constexpr (ALLW) {
    __syncthreads();
    constexpr int EP = T_TILE + 4;
    constexpr int NTH = (NMW + 4) * 64;
    constexpr int QPR = T_TILE / 4;
    float* yg = a.y ? a.y + (long long)b * a.y_bs : nullptr;
    float* y2g = a.y2 ? a.y2 + (long long)b * a.y_bs : nullptr;
    const float* rg = a.res ? a.res + (long long)b * a.y_bs : nullptr;
    const bool vec_ok = a.y_tstride == 1 && (a.y_cs & 3) == 0 && (a.y_bs & 3) == 0 &&
                        (!yg || (reinterpret_cast<unsigned long long>(a.y) & 15) == 0) &&
                        (!y2g || (reinterpret_cast<unsigned long long>(a.y2) & 15) == 0) &&
                        (!rg || (reinterpret_cast<unsigned long long>(a.res) & 15) == 0);
    if (a.rp > 1) {
      // ---- all-phases transposed conv: tile rows = (channel cl, phase p), p fastest; a lane emits four CONSECUTIVE output
      // samples u = t * rp + p of one channel (gathered from rp rows of the accumulator tile), so y / y2 leave as 16-byte
      // pieces of contiguous runs instead of rp interleaved strided streams.
      const int rp = a.rp, cpt = CO_TILE / rp;
      const int QPRp = (T_TILE * rp) / 4;                   // T_TILE * rp outputs per channel in this tile
      const long long u0 = (long long)t0 * rp, T_tot = (long long)a.T_out * rp;
      const int ch0 = (co0 / CO_TILE) * cpt;
      for (int q = tid; q < cpt * QPRp; q += NTH) {
        const int cl = q / QPRp, uq = q - cl * QPRp;
        const int co = ch0 + cl;
        const long long u = u0 + 4 * uq;
        if (co >= a.C_out || u >= T_tot) continue;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ul = 4 * uq + i, tl = ul / rp, p = ul - tl * rp;
          v[i] = smem[(cl * rp + p) * EP + tl];
        }
        const float bs = a.bias ? a.bias[co] : 0.f;
        const float al = a.alpha_out ? a.alpha_out[co] : 0.f;
        const float inv = a.alpha_out ? snake_inv(al) : 0.f;
        const long long o = (long long)co * a.y_cs + u;
        const bool full = vec_ok && u + 3 < T_tot;
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (rg) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rv[i] = u + i < T_tot ? rg[o + i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = v[i] + bs;
          if (a.alpha_out) x = snake_apply(x, al, inv);
          if (a.act != FAC_ACT_NONE) x = apply_act_slow(x, a.act);
          v[i] = x + rv[i];
        }
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (y2g) {
          const float a2 = a.alpha2[co], i2 = snake_inv(a2);
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = snake_apply(v[i], a2, i2);
        }
        if (full) {
          if (yg) *reinterpret_cast<float4*>(yg + o) = make_float4(v[0], v[1], v[2], v[3]);
          if (y2g) *reinterpret_cast<float4*>(y2g + o) = make_float4(w[0], w[1], w[2], w[3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (u + i >= T_tot) continue;
            if (yg) yg[o + i] = v[i];
            if (y2g) y2g[o + i] = w[i];
          }
        }
      }
    } else
    for (int q = tid; q < CO_TILE * QPR; q += NTH) {
      const int row = q / QPR, tq = q - row * QPR;
      const int co = co0 + row, t = t0 + 4 * tq;
      if (co >= a.C_out || t >= a.T_out) continue;
      const float4 av = *reinterpret_cast<const float4*>(smem + row * EP + 4 * tq);
      float v[4] = {av.x, av.y, av.z, av.w};
      const float bs = a.bias ? a.bias[co] : 0.f;
      const float al = a.alpha_out ? a.alpha_out[co] : 0.f;
      const float inv = a.alpha_out ? snake_inv(al) : 0.f;
      const long long o = (long long)co * a.y_cs + (long long)t * a.y_tstride + y_off;
      const bool full = vec_ok && t + 3 < a.T_out;
      float rv[4] = {0.f, 0.f, 0.f, 0.f};
      if (rg) {
        if (full) {
          const float4 r4 = *reinterpret_cast<const float4*>(rg + o);
          rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) rv[i] = t + i < a.T_out ? rg[o + (long long)i * a.y_tstride] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = v[i] + bs;
        if (a.alpha_out) x = snake_apply(x, al, inv);
        if (a.act != FAC_ACT_NONE) x = apply_act_slow(x, a.act);
        v[i] = x + rv[i];
      }
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      if (y2g) {
        const float a2 = a.alpha2[co], i2 = snake_inv(a2);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = snake_apply(v[i], a2, i2);
      }
      if (full) {
        if (yg) *reinterpret_cast<float4*>(yg + o) = make_float4(v[0], v[1], v[2], v[3]);
        if (y2g) *reinterpret_cast<float4*>(y2g + o) = make_float4(w[0], w[1], w[2], w[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (t + i >= a.T_out) continue;
          if (yg) yg[o + (long long)i * a.y_tstride] = v[i];
          if (y2g) y2g[o + (long long)i * a.y_tstride] = w[i];
        }
      }
    }
#ifdef FAC_PROF
    if (a.dbg && lane == 0 && wave < 4) {
      unsigned long long* d = a.dbg + ((long long)blockIdx.x * 4 + wave) * 4;
      d[0] = pf_first; d[1] = pf_bar; d[2] = pf_loop; d[3] = __builtin_readcyclecounter() - pf_start;
    }
#endif
  }
}

#if defined(FAC_PROF) || defined(FAC_PROF2)
extern unsigned long long* g_conv_dbg;
#endif

// Picks channels-per-stage and launches one instantiation.
template <int MB, int NB, int WM, int WN, int KT, bool FUSE = false>
int launch_cfg(ConvArgs& a, hipStream_t s) {
  constexpr int CO_TILE = 32 * MB * WM;
  constexpr int T_TILE = 32 * NB * WN;
  constexpr int UC = ConvUnroll<KT>::UC;
  constexpr int STEP = 2 * UC;
  // staged slab = receptive field of the tile, extended on the left to a 16-byte boundary and on the
  // right to a multiple of 4 columns, so interior slabs move as float4
  a.x_off = ((-a.pad_left) % 4 + 4) % 4;
  a.XW = (((T_TILE - 1) * a.stride + (a.KV - 1) * a.dil + 1 + a.x_off + (a.phase_shift > 0 ? 1 : 0)) + 3) & ~3;
  a.XB = (a.XW / 4 + 63) / 64;   // 64-lane blocks of float4 columns per staged row
  a.XQ = 4 / a.XB;
  a.XR = 4 % a.XB;
  const int per_ci = a.KV * CO_TILE + a.XW;      // floats per staged (virtual) input channel
  int cic;
  if constexpr (KT > 0) {
    cic = conv_cic<KT, T_TILE>();
    // the compile-time stage must fit the register slots and ~half the LDS; otherwise use the
    // run-time-sized generic path (unusual stride / dilation for this tap count)
    if (cic * a.XB > 4 * CONV_XMAX || (size_t)2 * cic * per_ci * sizeof(float) > (WM * WN == 4 ? 80 : 160) * 1024)
      return launch_cfg<MB, NB, WM, WN, 0, FUSE>(a, s);
  } else {
    int lim = 9216 / per_ci;                       // ~36 KB per stage -> 2 stages x 2 workgroups per CU
    const int lim_regs = (4 * CONV_XMAX) / a.XB;   // staged inputs must fit CONV_XMAX registers/thread
    if (lim > lim_regs) lim = lim_regs;
    if (lim > 32) lim = 32;
    lim = (lim / STEP) * STEP;
    if (lim < STEP) lim = STEP;
    const int cin_r = ((a.CV + STEP - 1) / STEP) * STEP;
    cic = lim < cin_r ? lim : cin_r;
    // prefer a chunk size that divides the (rounded) channel count: no half-empty last chunk
    for (int c = cic; c >= STEP && c * 2 > cic; c -= STEP)
      if (cin_r % c == 0) { cic = c; break; }
    if (cic * a.XB > 4 * CONV_XMAX) {
      set_error("conv1d: receptive field too wide to stage (K=%d stride=%d dil=%d)", a.K, a.stride, a.dil);
      return FAC_ERR_ARG;
    }
  }
  a.cic = cic;
  size_t lds = (size_t)2 * cic * per_ci * sizeof(float);
  if constexpr ((WM * WN == 8) && !FUSE) {   // accumulator tile of the all-waves epilogue
    const size_t epi = (size_t)CO_TILE * (T_TILE + 4) * sizeof(float);
    if (lds < epi) lds = epi;
  }
  if (lds > 160 * 1024) {
    set_error("conv1d: tile needs %zu B of LDS (K=%d stride=%d dil=%d)", lds, a.K, a.stride, a.dil);
    return FAC_ERR_ARG;
  }
  auto kern = conv1d_mfma_kernel<MB, NB, WM, WN, KT, FUSE>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  a.n_t_tiles = (a.T_out + T_TILE - 1) / T_TILE;
#ifdef FAC_PROF
  a.dbg = g_conv_dbg;
#endif
  if (a.rp > 1 && !((WM * WN == 8) && !FUSE && CO_TILE == 128)) {
    set_error("conv1d: row_phases needs the 128-row all-waves-epilogue tile");
    return FAC_ERR_ARG;
  }
  const int n_co_tiles = a.rp > 1 ? a.C_out_pad / CO_TILE : (a.C_out + CO_TILE - 1) / CO_TILE;
  const long long n_wg = (long long)a.n_t_tiles * n_co_tiles * a.B * a.n_phase;
  if (n_wg > 0x7fffffffll) {
    set_error("conv1d: too many workgroups (%lld)", n_wg);
    return FAC_ERR_ARG;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3((WM * WN + 4) * 64), lds, s, a);
  return check_launch("conv1d_mfma");
}

// One dispatcher per tile shape (each in its own translation unit so they build in parallel).
int conv_dispatch_128x128(ConvArgs& a, hipStream_t s);
int conv_dispatch_96x128(ConvArgs& a, hipStream_t s);
int conv_dispatch_64x128(ConvArgs& a, hipStream_t s);
int conv_dispatch_32x256(ConvArgs& a, hipStream_t s);
int conv_dispatch_128x32(ConvArgs& a, hipStream_t s);
int conv_dispatch_128x256(ConvArgs& a, hipStream_t s);
int conv_dispatch_96x256(ConvArgs& a, hipStream_t s);
int conv_dispatch_fused_ru(ConvArgs& a, hipStream_t s);
int conv_dispatch_128x160(ConvArgs& a, hipStream_t s);
int conv_dispatch_narrow(ConvArgs& a, hipStream_t s);
bool conv_cin1_ok(const ConvArgs& a);
bool conv_pw_ok(const ConvArgs& a);
int conv_dispatch_pw(ConvArgs& a, hipStream_t s);
bool conv_thin_ok(const ConvArgs& a, const void* ws, long long ws_bytes);
int conv_dispatch_thin(ConvArgs& a, void* ws, hipStream_t s);
int conv_dispatch_cin1(ConvArgs& a, hipStream_t s);
bool conv_bsplit2_ok(const ConvArgs& a);
int conv_dispatch_bsplit2(ConvArgs& a, hipStream_t s);
bool conv_gsplit_ok(const ConvArgs& a);
int conv_dispatch_gsplit(ConvArgs& a, hipStream_t s);
bool conv_bsplit_ok(const ConvArgs& a);
bool conv_bsplit_p8_ok(const ConvArgs& a);   // ... and the launch takes a P8 input (wide shape: C_in >= 64, C_in % 16 == 0)
int conv_dispatch_bsplit(ConvArgs& a, hipStream_t s);
bool conv_pws_ok(const ConvArgs& a);
int conv_dispatch_pws(ConvArgs& a, hipStream_t s);
bool conv_pwt_ok(const ConvArgs& a);        // streaming kernel with taps: stride-2 ConvTranspose1d (all phases) / k = 4 stride-2 conv, few channels
int conv_dispatch_pwt(ConvArgs& a, hipStream_t s);
bool conv_skinny_ok(const ConvArgs& a, const void* ws, long long ws_bytes);
int conv_dispatch_skinny(ConvArgs& a, void* ws, long long ws_bytes, hipStream_t s);

}  // namespace fac
