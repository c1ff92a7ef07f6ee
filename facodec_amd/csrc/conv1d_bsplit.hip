// fp32 convolution on the bf16 matrix pipe, without giving up fp32 accuracy.
//
// Every fp32 value is EXACTLY the sum of three round-to-nearest bf16 terms (8 + 8 + 8 significand bits:
// x = hi + mid + lo), and a product of two bf16 values is exact in fp32.  So w*x = sum of 9 exact cross
// products accumulated in fp32; the three smallest (mid*lo, lo*mid, lo*lo, <= 2^-24 relative) are below the
// rounding of the fp32 accumulation itself and are dropped: SIX v_mfma_f32_32x32x16_bf16 per K = 16 step.
// Measured on MI355X (tools/microbench/bf16_split_probe.hip): the bf16 pipe sustains 2 184 TFLOP/s, the fp32
// pipe 138 TFLOP/s, so six bf16 MFMAs cost 1/2.6 of the fp32 MFMAs they replace; error against fp64 of a
// K = 4 096 dot product: 2.16e-6 (this scheme) vs 2.57e-6 (v_mfma_f32_32x32x2_f32) relative to max|ref|.
// Inputs, outputs, bias, Snake, residuals and the accumulators stay fp32; nothing is stored in bf16 in HBM
// except the (pre-split, lossless) weights.
//
// Tile: 64 output channels x 256 time steps per workgroup, 4 MFMA waves (each 64 x 64 = 2 x 2 MFMA blocks)
// + NSW staging waves.  Stage = G groups of 8 input channels x all K taps = H = G*K "half slots" (8 channels
// of one tap, ordered tap-major); one MFMA contracts K = 16 = two half slots (half-wave 0: slot 2s, half-wave
// 1: slot 2s+1), an odd H is padded with a zero slot.  G = 2, K = 7: 7 MFMA steps per 16 channels, no padding.
//   weights : pre-split in HBM as [co tile][stage][plane][half slot][64 co][8 ci] bf16 -- one contiguous slab
//             per stage, moved by LDS-DMA; an A fragment (8 ci of one co) is one ds_read_b128;
//   inputs  : fp32 (B, C, T) rows -> the staging waves split each value (5 VALU ops) and write
//             [plane][ci group][column][8 ci] -- a B fragment of any tap is one aligned ds_read_b128 at column
//             t + k*dilation (no alignment constraints on dilation, unlike the fp32 slab).  The fp32 loads of
//             stage c+2 are issued one stage before they are split (register double buffer).
// Two stages of LDS (146 KB): one workgroup per CU.  Measured alternatives that fit two workgroups per CU
// (8-channel stages with tap pairs: 80 KB; a single stage of 73 KB with the co-resident workgroup as the
// second pipeline stage) ran at 100-120 TFLOP/s-equivalent against 125-180 for this layout.
#include "conv1d_mfma.h"
#include <type_traits>
#include "inflight_regs.h"

namespace fac {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BS_CO = 64;
constexpr int BS_NSW = 4;   // staging waves
#ifndef FAC_BS_NSW_WIDE
#define FAC_BS_NSW_WIDE 4
#endif
constexpr int BS_NSW_WIDE = FAC_BS_NSW_WIDE;
constexpr int BS_XU = 3;    // (ci group, 64-column block) staging units per staging wave
// Two shapes of the same kernel (the weight layout depends on G, so the choice is a pure function of C_in):
//   wide    (C_in >= BS_WIDE_MIN): G = 2 groups of 8 channels per stage, 4 MFMA waves, tile 64 x 256
//   narrow  (C_in <  BS_WIDE_MIN): G = 1, tap PAIRS per MFMA (7 taps + one zero tap), 8 MFMA waves, tile 64 x 512:
//           few-channel layers have few stages per tile, so a tile twice as long (and two MFMA waves per SIMD)
//           amortises the per-tile prologue / epilogue that one resident workgroup per CU cannot hide.
//   The boundary was 160 channels while the MFMA waves ran the epilogue alone; with the all-waves epilogue the wide shape
//   wins from 64 channels up (C = 128: 140 -> 161, C = 96: 112 -> 126, C = 64: 110 -> 118 TFLOP/s-eq), so narrow is left for
//   the 32- and 48-channel layers (MPD).
#ifndef FAC_BS_WIDE_MIN
#define FAC_BS_WIDE_MIN 64
#endif
constexpr int BS_WIDE_MIN = FAC_BS_WIDE_MIN;
__host__ __device__ constexpr int bs_group(int C_in) { return C_in >= BS_WIDE_MIN ? 2 : 1; }

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

__host__ __device__ constexpr int bs_slots(int K, int G) { return (G * K + 1) & ~1; }

// v (C_out, C_in, K) [* scale per C_out] -> split layout described above.  One thread per (tile, stage, half
// slot, co): 8 input channels -> three 16-byte pieces.
__global__ void pack_conv_split_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                       bf16x8* __restrict__ out, int C_out, int C_in, int K, int G, int H, int n_st,
                                       long long n) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(idx % BS_CO);
    long long r = idx / BS_CO;
    const int hs = (int)(r % H);          // half slot -> (ci group, tap)
    r /= H;
    const int s = (int)(r % n_st);
    const int ct = (int)(r / n_st);
    const int g = hs % G, k = hs / G;     // slot = tap-major, group-minor: both half slots of a step share the tap (G = 2)
    const int cog = ct * BS_CO + co;
    const float sc = (scale != nullptr && cog < C_out) ? scale[cog] : 1.0f;
    bf16x8 h, m, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ci = (s * G + g) * 8 + i;
      float w = 0.f;
      if (cog < C_out && ci < C_in && k < K) {
        w = v[((long long)cog * C_in + ci) * K + k];
        if (scale != nullptr) w = __fmul_rn(w, sc);
      }
      __bf16 a, b2, c;
      split3(w, a, b2, c);
      h[i] = a; m[i] = b2; l[i] = c;
    }
    const long long base = ((long long)ct * n_st + s) * 3;
    out[((base + 0) * H + hs) * BS_CO + co] = h;
    out[((base + 1) * H + hs) * BS_CO + co] = m;
    out[((base + 2) * H + hs) * BS_CO + co] = l;
  }
}

template <int KT, int G, int NMW, int NSW>
__global__ __launch_bounds__((NMW + NSW) * 64, (NMW + NSW) / 4) void conv1d_bsplit_kernel(ConvArgs a) {
  constexpr int MB = 2, NB = 2;
  constexpr int BS_TT = 64 * NMW;                     // time steps per tile
  constexpr int H = bs_slots(KT, G);                  // half slots per stage (even)
  constexpr int W_STAGE = 3 * H * BS_CO * 16;         // bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // [0, NMW): MFMA waves, then the staging waves
  const int XW = a.XW;
  const int X_STAGE = 48 * G * XW;                              // 3 planes x G groups x XW x 16 B
  unsigned char* Wbuf = sm;              // [2][W_STAGE]
  unsigned char* Xbuf = sm + 2 * W_STAGE;   // [2][X_STAGE]

  // XCD-aware work decode (see conv1d_mfma.h): each XCD walks a contiguous range of (co tile, b, t tile)
  int t0, co0, b;
  {
    const int n = gridDim.x;
    const int q8 = n >> 3, r8 = n & 7;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int nt = a.n_t_tiles;
    const int tt = id % nt;
    const int rest = id / nt;
    b = rest % a.B;
    co0 = (rest / a.B) * BS_CO;
    t0 = tt * BS_TT;
  }
  const int n_chunks = (a.C_in + 8 * G - 1) / (8 * G);
  const int dil = a.dil;

#ifdef FAC_PROF
  unsigned long long pf0 = 0, pf1 = 0, pf2 = 0;
#endif
  if (wave >= NMW) {
    // ===================== staging waves
    const int lw = wave - NMW;
    __builtin_amdgcn_s_setprio(FAC_PRIO_STAGE);
    const float* xg = a.x + (long long)b * a.x_bs;
    const int xcs = (int)a.x_cs;     // one clip's rows stay far below 2^31 elements (checked by the dispatcher)
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w) + (long long)(co0 / BS_CO) * n_chunks * W_STAGE;
    const int n_blk = (XW + 63) >> 6;
    // (ci group, 64-column block) units of this wave; the column -> input index map is chunk-invariant
    int u_g[BS_XU], u_c[BS_XU], u_idx[BS_XU];
#pragma unroll
    for (int j = 0; j < BS_XU; ++j) {
      const int u = lw + NSW * j;
      u_g[j] = u % G;
      const int c = (u / G) * 64 + lane;
      u_c[j] = (u < G * n_blk && c < XW) ? c : -1;
      const int tin = t0 - a.pad_left + c;
      int idx;
      if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
      else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
      u_idx[j] = u_c[j] >= 0 ? idx : -1;
    }
    // Weights: one contiguous slab per stage, 16 B per lane by LDS-DMA.  Every staging wave issues exactly ND DMA instructions
    // (the block index is clamped: a wave short of one block re-copies the last block -- same bytes to the same place), so the
    // position of a stage's loads in the wave's in-order load queue is a compile-time constant.
    constexpr int NBLK = W_STAGE / 1024;               // 1 KiB blocks per weight stage
    constexpr int ND = (NBLK + NSW - 1) / NSW;         // DMA instructions per staging wave and stage
    constexpr int NX = BS_XU * 8;                      // input loads per staging wave and stage
    static_assert(W_STAGE % 1024 == 0 && ND + NX <= 63 && NX == 24, "vmcnt is a 6-bit counter; FAC_XREGS24_* list 24 registers per set");
    // (used by the narrow shape; a fully unrolled version with clamped block indices made hipcc spill 443 registers there)
    auto stage_w = [&](int chunk, int buf) {
#if !defined(FAC_ABL_NOSTAGE) && !defined(FAC_ABL_NOSTAGE_W)
      constexpr int N16 = W_STAGE / 16;
      const unsigned char* src = wsrc + (long long)chunk * W_STAGE;
      unsigned char* dst = Wbuf + buf * W_STAGE;
      for (int i = lw; i * 64 < N16; i += NSW) {
        const int q = i * 64 + lane;
        if (q < N16)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(src + (long long)q * 16), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
      }
#endif
    };
    auto write_x = [&](int buf, const float (&xr)[BS_XU][8]) {       // xr: landed samples, padding lanes already zero
#if !defined(FAC_ABL_NOSTAGE) && !defined(FAC_ABL_NOSTAGE_X)
      unsigned char* xd = Xbuf + buf * X_STAGE;
#pragma unroll
      for (int j = 0; j < BS_XU; ++j) {
        if (u_c[j] < 0) continue;
        bf16x8 h, m, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __bf16 p0, p1, p2;
          split3(xr[j][i], p0, p1, p2);
          h[i] = p0; m[i] = p1; l[i] = p2;
        }
        *reinterpret_cast<bf16x8*>(xd + ((0 * G + u_g[j]) * XW + u_c[j]) * 16) = h;
        *reinterpret_cast<bf16x8*>(xd + ((1 * G + u_g[j]) * XW + u_c[j]) * 16) = m;
        *reinterpret_cast<bf16x8*>(xd + ((2 * G + u_g[j]) * XW + u_c[j]) * 16) = l;
      }
#endif
    };
    // Every instruction of the staging waves costs the SIMD's MFMA wave issue time, so the loads are kept to one instruction
    // each: the channel row is a uniform (scalar) base, the column a per-lane 32-bit byte offset resolved once per tile; lanes on
    // padding read a clamped column and are zeroed when the value is taken out of its landing register, and C_in % (8 G) == 0
    // (dispatcher) makes every channel of a stage real.
    if (G == 2 && a.x_p8 != nullptr) {
      // ---- P8 input (fac_conv_desc.x_p8): the producer wrote the three bf16 planes, 8 channels x 16 B per time step -- exactly
      // a column of this kernel's input stage.  A unit (channel group, column) is three 16-byte loads and three ds_write_b128: NO
      // vector-ALU work at all (the fp32 path below spends ~170 VALU instructions per staging wave and stage on the split, and
      // VALU instructions do not overlap the MFMAs of the same SIMD: profiles/r04_bsplit_stage_phases.log).  Everything is
      // requested at the start of the step and written at its end; nothing stays in flight across a barrier.
      typedef float wv4 __attribute__((ext_vector_type(4)));
      const unsigned char* xp = a.x_p8 + (long long)b * (a.C_in / 8) * a.T_in * 16;      // plane 0 of this clip
      const long long grp_bytes = (long long)a.T_in * 16;
      unsigned u_poff[BS_XU];
      bool any_pad = false;
#pragma unroll
      for (int j = 0; j < BS_XU; ++j) {
        u_poff[j] = (unsigned)(u_idx[j] >= 0 ? u_idx[j] : 0) * 16u;
        any_pad = any_pad || __builtin_amdgcn_ballot_w64(u_c[j] >= 0 && u_idx[j] < 0) != 0;
      }
      for (int c = -1; c < n_chunks; ++c) {                          // step(c): chunk c + 1 -> stage (c + 1) & 1
        if (c + 1 < n_chunks) {
          const int buf = (c + 1) & 1;
          wv4 wv[ND], xv[BS_XU][3];
          const unsigned char* src = wsrc + (long long)(c + 1) * W_STAGE;
#pragma unroll
          for (int j = 0; j < ND; ++j) {
            const int bi = min(lw + NSW * j, NBLK - 1);
            wv[j] = *reinterpret_cast<const wv4*>(src + bi * 1024 + lane * 16);
          }
#pragma unroll
          for (int j = 0; j < BS_XU; ++j) {
            const unsigned char* grp = xp + (long long)((c + 1) * G + u_g[j]) * grp_bytes;      // uniform
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) xv[j][pl] = *reinterpret_cast<const wv4*>(grp + pl * a.x_p8_ps + u_poff[j]);
          }
          unsigned char* xd = Xbuf + buf * X_STAGE;
#pragma unroll
          for (int j = 0; j < BS_XU; ++j) {
            if (u_c[j] < 0) continue;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              wv4 v = xv[j][pl];
              if (any_pad && u_idx[j] < 0) v = wv4{0.f, 0.f, 0.f, 0.f};            // edge tiles only (wave-uniform guard)
              *reinterpret_cast<wv4*>(xd + ((pl * G + u_g[j]) * XW + u_c[j]) * 16) = v;
            }
          }
          unsigned char* dst = Wbuf + buf * W_STAGE;
#pragma unroll
          for (int j = 0; j < ND; ++j) {
            const int bi = min(lw + NSW * j, NBLK - 1);
            *reinterpret_cast<wv4*>(dst + bi * 1024 + lane * 16) = wv[j];
          }
        }
        __syncthreads();       // c = -1: chunk 0 staged; later: pairs with the MFMA waves' barrier behind chunk c
      }
    } else if constexpr (G == 2) {
      // ---- wide shape: the inputs of chunk c + 2 stay IN FLIGHT across the barrier -----------------------------------------
      // Rounds 1-3 wrote this as a "register double buffer" of plain C++ loads; hipcc waits for such a load at its first use --
      // and the padding select is a use -- so it placed s_waitcnt vmcnt(0) right behind the last load of the same stage (and again
      // in front of every barrier, together with the weight DMA issued a moment earlier): every stage paid a full memory round
      // trip on the staging waves' critical path, which is what kept the matrix pipe at 48 % busy (found in the ISA in round 4).
      // Loads that stay in flight cannot be compiler-visible values: hipcc does not know that the destination of an inline-asm
      // load is invalid until the matching s_waitcnt, and it did copy such registers at control-flow joins (right at B = 2,
      // wrong codes at B = 32 when the memory system is loaded).  So the landing registers are NAMED PHYSICAL REGISTERS that the
      // compiler never sees as values: set A = v208..v231 (even chunks), set B = v232..v255 (odd chunks), written by
      // `global_load_dword vNNN` and read back -- after `s_waitcnt vmcnt(n)` on the in-order load queue -- by the v_cndmask that
      // zeroes the padding lanes anyway.  hipcc allocates registers from v0 upwards and this kernel needs ~120, far from v208;
      // tools/check_inflight_regs.py (tests/test_isa_inflight.py) verifies on the ISA that nothing else touches v208..v255.
#define BS_LD(n, R)                                                                                                          \
  asm volatile("global_load_dword v" #R ", %0, %1" : : "v"(u_boff[(n) / 8]), "s"(grp[(n) / 8] + (long long)((n) % 8) * xcs) : "memory", "v" #R);
#define BS_RD(n, R) asm volatile("v_cndmask_b32_e64 %0, 0, v" #R ", %1" : "=v"(xr[(n) / 8][(n) % 8]) : "s"(u_mask[(n) / 8]) : "memory");
      unsigned u_boff[BS_XU];
      unsigned long long u_mask[BS_XU];                               // lanes of unit j that hold a real sample
#pragma unroll
      for (int j = 0; j < BS_XU; ++j) {
        u_boff[j] = (unsigned)(u_idx[j] >= 0 ? u_idx[j] : 0) * 4u;
        u_mask[j] = __builtin_amdgcn_ballot_w64(u_idx[j] >= 0);
      }
      auto rows_of = [&](int chunk, const float* (&grp)[BS_XU]) {
#pragma unroll
        for (int j = 0; j < BS_XU; ++j) grp[j] = xg + (long long)((chunk * G + u_g[j]) * 8) * xcs;
      };
      auto load_a = [&](int chunk) {
#if !defined(FAC_ABL_NOSTAGE) && !defined(FAC_ABL_NOSTAGE_X)
        const float* grp[BS_XU];
        rows_of(chunk, grp);
        FAC_XREGS24_A(BS_LD)
#endif
      };
      auto load_b = [&](int chunk) {
#if !defined(FAC_ABL_NOSTAGE) && !defined(FAC_ABL_NOSTAGE_X)
        const float* grp[BS_XU];
        rows_of(chunk, grp);
        FAC_XREGS24_B(BS_LD)
#endif
      };
      auto stage_w_fixed = [&](int chunk, int buf) {       // FAC_BS_W_VGPR=0: exactly ND LDS-DMA instructions per wave
        const unsigned char* src = wsrc + (long long)chunk * W_STAGE;
        unsigned char* dst = Wbuf + buf * W_STAGE;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
          const int i = min(lw + NSW * j, NBLK - 1);
          __builtin_amdgcn_global_load_lds((glb_void_t*)(src + (long long)i * 1024 + lane * 16), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
        }
      };
      (void)stage_w_fixed;
      auto take_a = [&](float (&xr)[BS_XU][8]) { FAC_XREGS24_A(BS_RD) };
      auto take_b = [&](float (&xr)[BS_XU][8]) { FAC_XREGS24_B(BS_RD) };
      // ONE software-pipelined loop from c = -2: step(c) = { weight DMA of chunk c + 1 into stage (c + 1) & 1; wait for the inputs
      // of c + 1 (requested by step(c - 1), older than that DMA: vmcnt(ND)); request the inputs of c + 2; take c + 1 out of its
      // landing registers, split, write; wait for the DMA (vmcnt(NX): the inputs of c + 2 stay in flight); barrier }, each part
      // skipped where its chunk does not exist.  At most ND + NX loads are in flight.
#ifdef FAC_PROF2
      long long pq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
      for (int base = -2; base < n_chunks; base += 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                // c + 1 has parity 1 - i: its stage and its register set
          const int c = base + i;
          if (c >= n_chunks) break;
          const bool has_next = c + 1 >= 0 && c + 1 < n_chunks, has_next2 = c + 2 < n_chunks;
          float xr[BS_XU][8];
#ifndef FAC_BS_W_VGPR
#define FAC_BS_W_VGPR 1
#endif
#if FAC_BS_W_VGPR
          // The weight slab goes through registers (global_load_dwordx4 -> ds_write_b128, requested before the input work of the
          // stage, written behind it -- no value crosses a barrier, plain compiler-allocated registers) instead of LDS-DMA:
          // measured +2 .. 4 % (C = 192 .. 768).  Ablations of round 4 (profiles/r04_bsplit_ablation.log): what the weight
          // stage costs the MFMA waves is its LDS WRITE traffic, whichever way it arrives (a DMA that reads one cache-resident
          // KiB over and over costs the same as the real one), not the L2 fetch.  FAC_BS_W_VGPR=0: LDS-DMA.
          typedef float wv4 __attribute__((ext_vector_type(4)));
          wv4 wv[ND];
#ifdef FAC_PROF2
          const long long q0 = clock64();
          long long q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0, q6 = q0;
#endif
          if (has_next) {
            const unsigned char* src = wsrc + (long long)(c + 1) * W_STAGE;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
              const int bi = min(lw + NSW * j, NBLK - 1);
              asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wv[j]) : "v"((unsigned)(bi * 1024 + lane * 16)), "s"(src) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(ND) : "memory");
#ifdef FAC_PROF2
            q1 = clock64();
#endif
            if (i == 0) take_b(xr); else take_a(xr);
#ifdef FAC_PROF2
            q2 = clock64();
#endif
          }
          if (has_next2) {
            if (i == 0) load_a(c + 2); else load_b(c + 2);
          }
#ifdef FAC_PROF2
          q3 = clock64();
#endif
          if (has_next) {
            write_x(1 - i, xr);
#ifdef FAC_PROF2
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            q4 = clock64();
#endif
            if (has_next2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NX) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef FAC_PROF2
            q5 = clock64();
#endif
            unsigned char* dst = Wbuf + (1 - i) * W_STAGE;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
              asm volatile("" : "+v"(wv[j]) : : "memory");
              const int bi = min(lw + NSW * j, NBLK - 1);
              *reinterpret_cast<wv4*>(dst + bi * 1024 + lane * 16) = wv[j];
            }
          }
          if (c >= -1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef FAC_PROF2
            q6 = clock64();
#endif
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#ifdef FAC_PROF2
            if (c >= 0 && has_next2) {      // steady-state iterations only
              pq[0] += q1 - q0; pq[1] += q2 - q1; pq[2] += q3 - q2; pq[3] += q4 - q3; pq[4] += q5 - q4; pq[5] += q6 - q5;
              pq[6] += clock64() - q6; pq[7] += 1;
            }
#endif
          }
#else
          if (has_next) {
            stage_w_fixed(c + 1, 1 - i);                             // that stage was read during chunk c - 1
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(ND) : "memory");
            if (i == 0) take_b(xr); else take_a(xr);
          }
          if (has_next2) {
            if (i == 0) load_a(c + 2); else load_b(c + 2);
          }
          if (has_next) write_x(1 - i, xr);
          if (c >= -1) {
            if (has_next2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(NX) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // c = -1: chunk 0 staged; later: pairs with the MFMA waves' barrier behind chunk c
            asm volatile("" ::: "memory");
          }
#endif
        }
      }
#ifdef FAC_PROF2
      if (a.dbg && lw == 0 && lane == 0) {
        unsigned long long* d = a.dbg + (long long)blockIdx.x * 16 + 8;
        for (int k = 0; k < 8; ++k) d[k] = (unsigned long long)pq[k];
      }
#endif
#undef BS_LD
#undef BS_RD
    } else {
      // ---- narrow shape (32 / 48 input channels, three waves per SIMD: no registers to spare for named landing sets): plain
      // loads, which hipcc waits for within the stage
      int u_off[BS_XU];
#pragma unroll
      for (int j = 0; j < BS_XU; ++j) u_off[j] = u_idx[j] >= 0 ? u_idx[j] : 0;
      auto load_x = [&](int chunk, float (&xr)[BS_XU][8]) {
#if !defined(FAC_ABL_NOSTAGE) && !defined(FAC_ABL_NOSTAGE_X)
#pragma unroll
        for (int j = 0; j < BS_XU; ++j) {
          const float* grp = xg + (long long)((chunk * G + u_g[j]) * 8) * xcs;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float v = (grp + (long long)i * xcs)[u_off[j]];
            xr[j][i] = u_idx[j] >= 0 ? v : 0.f;
          }
        }
#endif
      };
      float xa[BS_XU][8], xb[BS_XU][8];
      load_x(0, xa);
      stage_w(0, 0);
      if (n_chunks > 1) load_x(1, xb);
      write_x(0, xa);
      __syncthreads();
      for (int chunk = 0; chunk < n_chunks; chunk += 2) {
        if (chunk + 1 < n_chunks) {
          stage_w(chunk + 1, 1);
          if (chunk + 2 < n_chunks) load_x(chunk + 2, xa);
          write_x(1, xb);
        }
        __syncthreads();
        if (chunk + 1 >= n_chunks) break;
        if (chunk + 2 < n_chunks) {
          stage_w(chunk + 2, 0);
          if (chunk + 3 < n_chunks) load_x(chunk + 3, xb);
          write_x(0, xa);
        }
        __syncthreads();
      }
    }
    __builtin_amdgcn_s_setprio(0);
  } else {
  // ========================= MFMA waves
#ifdef FAC_PROF
  const unsigned long long tp0 = wall_clock64();
#endif
  __builtin_amdgcn_s_setprio(FAC_PRIO_MFMA);
#ifdef FAC_PROF2
  const long long ck0 = clock64();
  const unsigned long long wk0 = wall_clock64();
#endif
  const int l31 = lane & 31;
  const int kq = lane >> 5;
  const int n0 = wave * 64;
  f32x16 acc[MB][NB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // Half slot hs = (tap hs / G, group hs % G).  For step st the two half-waves read slots 2st and 2st+1:
  // G = 2: same tap st, groups 0 / 1;  G = 1: taps 2st / 2st+1 -- either way a per-lane base plus a uniform
  // per-step offset.
  const int x_lane = (G == 2 ? kq * XW : kq * dil) * 16;
  const int x_step = (G == 2 ? dil : 2 * dil) * 16;

  __syncthreads();   // chunk 0 staged
#ifdef FAC_PROF
  const unsigned long long tp1 = wall_clock64();
#endif
#ifndef FAC_BS_PIPE
#define FAC_BS_PIPE 1
#endif
  if constexpr (NMW == 4 && FAC_BS_PIPE) {
    // ---- wide shape, round 4: the fragment pipeline runs ACROSS the stage barrier.
    // A stage is H / 2 steps of 24 MFMAs; the fragments of step s + 1 are requested while step s multiplies (two register sets).
    //  * The 12 ds_read_b128 of the next step are interleaved with the MFMAs of the current one (sched_group_barrier: 2 MFMAs,
    //    1 read, ...) instead of being issued in a burst in front of them: a wave issues in order, so the burst kept the matrix
    //    pipe idle for the issue time of twelve LDS instructions once per step.
    //  * The stage barrier sits in front of the LAST step's MFMAs, not behind them: by then every fragment of the stage is in
    //    registers (s_waitcnt lgkmcnt(0)), so the staging waves may overwrite the buffer, and the first fragments of the NEXT stage
    //    -- staged long ago -- are requested right behind the barrier and arrive under the last step's 24 MFMAs.  With the barrier
    //    at the end, every stage started with an exposed LDS round trip and a drained matrix pipe.
    // H / 2 is odd (7, 5, 3 steps), so the register-set parity flips from stage to stage: the loop body covers two stages.
    constexpr int S = H / 2;
    // A tile whose upper 32 rows lie beyond C_out (the second tile of the 96-channel layers: a quarter of their matrix work was
    // spent on zero rows) runs the same pipeline with ONE row block per wave.
    auto pipeline = [&](auto MBc) {
    constexpr int MBv = decltype(MBc)::value;
    bf16x8 A[2][MBv][3], Bf[2][NB][3];
    auto ld = [&](int buf, int st, bf16x8 (&Ad)[MBv][3], bf16x8 (&Bd)[NB][3]) {
      const unsigned char* Wb = Wbuf + buf * W_STAGE + (kq * BS_CO + l31) * 16;          // half slot 2 st + kq
      const unsigned char* Xb = Xbuf + buf * X_STAGE + (n0 + l31) * 16 + x_lane + st * x_step;
      constexpr int PO[3] = {1, 0, 2};   // planes in order of first use: mid, hi, lo
#pragma unroll
      for (int pi = 0; pi < 3; ++pi) {
#pragma unroll
        for (int n = 0; n < NB; ++n) Bd[n][PO[pi]] = *reinterpret_cast<const bf16x8*>(Xb + (PO[pi] * G * XW + n * 32) * 16);
#pragma unroll
        for (int m = 0; m < MBv; ++m) Ad[m][PO[pi]] = *reinterpret_cast<const bf16x8*>(Wb + ((PO[pi] * H + 2 * st) * BS_CO + m * 32) * 16);
      }
    };
    auto mma = [&](const bf16x8 (&Ac)[MBv][3], const bf16x8 (&Bc)[NB][3]) {
      // smallest terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi; the term loop is OUTSIDE the block loops so that
      // consecutive MFMAs write different accumulators
      constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < MBv; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac[m][TA[q]], Bc[n][TB[q]], acc[m][n], 0, 0, 0);
    };
    auto interleave = [&]() {            // 24 MFMAs and 12 LDS reads in the region: M M R  M M R ...
#pragma unroll
      for (int i = 0; i < 3 * (MBv + NB); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, (6 * MBv * NB) / (3 * (MBv + NB)) > 0 ? (6 * MBv * NB) / (3 * (MBv + NB)) : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    };
    ld(0, 0, A[0], Bf[0]);
    for (int base = 0; base < n_chunks; base += 2) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int chunk = base + cc;
        if (chunk >= n_chunks) break;
        const int buf = cc;                                     // chunk & 1 (base is even)
#pragma unroll
        for (int st = 0; st < S; ++st) {
          const int cur = (cc * S + st) & 1;                   // compile-time after unrolling
          if (st + 1 < S) {
            if (cur == 0) ld(buf, st + 1, A[1], Bf[1]); else ld(buf, st + 1, A[0], Bf[0]);
            if (cur == 0) mma(A[0], Bf[0]); else mma(A[1], Bf[1]);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
          } else {
            // last step of the stage: its fragments were requested a step ago; once they are in, the buffer is free
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (chunk + 1 < n_chunks) {
              if (cur == 0) ld(buf ^ 1, 0, A[1], Bf[1]); else ld(buf ^ 1, 0, A[0], Bf[0]);
            }
            if (cur == 0) mma(A[0], Bf[0]); else mma(A[1], Bf[1]);
            if (chunk + 1 < n_chunks) interleave();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    };
    if (co0 + 32 >= a.C_out) pipeline(std::integral_constant<int, 1>{});
    else pipeline(std::integral_constant<int, MB>{});
  } else {
#ifdef FAC_PROF2
  long long pm[3] = {0, 0, 0};
#endif
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
#ifdef FAC_PROF2
    const long long m0 = clock64();
#endif
    const int buf = chunk & 1;
    const unsigned char* Wb = Wbuf + buf * W_STAGE + (kq * BS_CO + l31) * 16;          // half slot 2s + kq
    const unsigned char* Xb = Xbuf + buf * X_STAGE + (n0 + l31) * 16 + x_lane;
    // A fragments are requested one step ahead (register double buffer); B fragments at the start of their
    // step, in the order the six terms consume them (the compiler waits per fragment, and the second workgroup
    // on the CU covers what latency remains) -- keeps the kernel under the 170 VGPRs of 3 waves per SIMD.
#ifndef FAC_BS_BDBL
#define FAC_BS_BDBL 1
#endif
    constexpr bool BDBL = FAC_BS_BDBL && NMW == 4;     // B fragments requested one step ahead as well (two register sets)
    bf16x8 A[2][MB][3], Bf[BDBL ? 2 : 1][NB][3];
    auto ldA = [&](int st, bf16x8 (&Ad)[MB][3]) {
#ifdef FAC_ABL_NOLDS
      if (st > 1) return;
#endif
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int m = 0; m < MB; ++m)
          Ad[m][p] = *reinterpret_cast<const bf16x8*>(Wb + ((p * H + 2 * st) * BS_CO + m * 32) * 16);
    };
    auto ldB = [&](int st, bf16x8 (&Bd)[NB][3]) {
#ifdef FAC_ABL_NOLDS
      if (st > 0) return;
#endif
      const int xo = st * x_step;
      constexpr int PO[3] = {1, 0, 2};   // planes in order of first use: mid, hi, lo
#pragma unroll
      for (int pi = 0; pi < 3; ++pi)
#pragma unroll
        for (int n = 0; n < NB; ++n)
          Bd[n][PO[pi]] = *reinterpret_cast<const bf16x8*>(Xb + xo + (PO[pi] * G * XW + n * 32) * 16);
    };
    // 8 MFMA waves (2 per SIMD, 3 waves per SIMD in all -> 170 VGPRs): the sibling wave hides the LDS latency,
    // so A is fetched at the start of its step too and only one A buffer is kept
    constexpr bool ADBL = NMW == 4;
    if (ADBL) ldA(0, A[0]);
    if (BDBL) ldB(0, Bf[0]);
#pragma unroll
    for (int st = 0; st < H / 2; ++st) {
      if (BDBL) {
        if (st + 1 < H / 2) ldB(st + 1, Bf[(st + 1) & 1]);
      } else {
        ldB(st, Bf[0]);
      }
      if (ADBL) {
        if (st + 1 < H / 2) ldA(st + 1, A[(st + 1) & 1]);
      } else {
        ldA(st, A[0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int s = ADBL ? (st & 1) : 0;
      const int sb = BDBL ? (st & 1) : 0;
      // smallest terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi.  The term loop is OUTSIDE the
      // block loops so that consecutive MFMAs write different accumulators (no back-to-back dependency).
      constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#ifndef FAC_ABL_NOMFMA
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s][m][TA[q]], Bf[sb][n][TB[q]], acc[m][n], 0, 0, 0);
#else
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n][st & 15] += (float)A[s][m][0][0] * (float)Bf[sb][n][1][1] + (float)A[s][m][2][3] + (float)Bf[sb][n][2][5] + (float)A[s][m][1][7] * (float)Bf[sb][n][0][2];
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef FAC_PROF2
    const long long m1 = clock64();
#endif
    __syncthreads();
#ifdef FAC_PROF2
    pm[0] += m1 - m0; pm[1] += clock64() - m1; pm[2] += 1;
#endif
  }
#ifdef FAC_PROF2
  if (a.dbg && wave == 0 && lane == 0) {
    unsigned long long* d = a.dbg + (long long)blockIdx.x * 16;
    d[0] = (unsigned long long)pm[0]; d[1] = (unsigned long long)pm[1]; d[2] = (unsigned long long)pm[2];
  }
#endif

  }
#ifdef FAC_PROF
  const unsigned long long tp2 = wall_clock64();
  pf0 = tp0; pf1 = tp1; pf2 = tp2;
#endif
  __builtin_amdgcn_s_setprio(0);
#ifdef FAC_PROF2
  if (a.dbg && wave == 0 && lane == 0) {      // shader clock (s_memtime) against the constant 100 MHz clock over the main loop
    unsigned long long* d = a.dbg + (long long)blockIdx.x * 16;
    d[3] = (unsigned long long)(clock64() - ck0);
    d[4] = wall_clock64() - wk0;
    d[5] = (unsigned long long)n_chunks;
  }
#endif
  // ---- accumulators -> LDS (both stage buffers are free now): tile[co][t] fp32, row pitch BS_TT + 4 floats.
  // C/D layout of the 32x32 block: register r <-> row (r & 3) + 8 (r >> 2) + 4 kq, column l31.
  {
    float* tile = reinterpret_cast<float*>(sm);
    constexpr int EP = BS_TT + 4;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tile[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq) * EP + n0 + n * 32 + l31] = acc[m][n][r];
  }
  }   // MFMA waves
  __syncthreads();

  // ---- epilogue by ALL waves (the staging waves are idle by now): bias, Snake, activation, residual, y / y2.
  // One lane = 4 consecutive time steps of one output channel; consecutive lanes = consecutive quads of a row, so residual
  // loads and the stores are 16-byte pieces of contiguous 1 KiB runs.  (With the MFMA waves alone -- 64 outputs per lane,
  // scalar 4-byte accesses in C/D order, one workgroup per CU so nothing else to overlap with -- the epilogue cost the
  // C <= 192 layers a third of their time.)
  {
    const float* tile = reinterpret_cast<const float*>(sm);
    constexpr int EP = BS_TT + 4;
    constexpr int NTH = (NMW + NSW) * 64;
    constexpr int QPR = BS_TT / 4;                       // quads per row
    float* yg = a.y ? a.y + (long long)b * a.y_bs : nullptr;
    float* y2g = a.y2 ? a.y2 + (long long)b * a.y_bs : nullptr;
    const float* rg = a.res ? a.res + (long long)b * a.y_bs : nullptr;
    const bool vec_ok = (a.y_cs & 3) == 0 && (a.y_bs & 3) == 0 && (!yg || (reinterpret_cast<unsigned long long>(a.y) & 15) == 0) &&
                        (!y2g || (reinterpret_cast<unsigned long long>(a.y2) & 15) == 0) &&
                        (!rg || (reinterpret_cast<unsigned long long>(a.res) & 15) == 0);
    for (int q = tid; q < BS_CO * QPR; q += NTH) {
      const int row = q / QPR, tq = q - row * QPR;
      const int co = co0 + row, t = t0 + 4 * tq;
      if (co >= a.C_out || t >= a.T_out) continue;
      const float4 av = *reinterpret_cast<const float4*>(tile + row * EP + 4 * tq);
      float v[4] = {av.x, av.y, av.z, av.w};
      const float bs = a.bias ? a.bias[co] : 0.f;
      const float al = a.alpha_out ? a.alpha_out[co] : 0.f;
      const float inv = a.alpha_out ? snake_inv(al) : 0.f;
      const long long o = (long long)co * a.y_cs + t;
      const bool full = vec_ok && t + 3 < a.T_out;
      float rv[4] = {0.f, 0.f, 0.f, 0.f};
      if (rg) {
        if (full) {
          const float4 r4 = *reinterpret_cast<const float4*>(rg + o);
          rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) rv[i] = t + i < a.T_out ? rg[o + i] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = v[i] + bs;
        if (a.alpha_out) x = snake_apply(x, al, inv);
        if (a.act != FAC_ACT_NONE) x = apply_act_slow(x, a.act);
        v[i] = x + rv[i];
      }
      float w[4];
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
          if (yg) yg[o + i] = v[i];
          if (y2g) y2g[o + i] = w[i];
        }
      }
    }
  }
#ifdef FAC_PROF
  if (a.dbg && tid == 0) {
    unsigned long long* d = a.dbg + (long long)blockIdx.x * 8;
    d[0] = pf0; d[1] = pf1; d[2] = pf2; d[3] = wall_clock64(); d[4] = 0; d[5] = 0;
  }
#endif
}

bool conv_bsplit_ok(const ConvArgs& a) {
  if (!((a.K == 7 || a.K == 5 || a.K == 3) && a.stride == 1 && a.n_phase == 1 && a.phase_shift == 0 && a.y_tstride == 1 && !a.alpha_in &&
        !a.w1 && !a.w_batched && (long long)a.B * a.T_out > 640))
    return false;
  const int G = bs_group(a.C_in), tt = G == 2 ? 256 : 512;
  const int kp = G == 2 ? a.K - 1 : ((a.K + 1) & ~1) - 1;                        // G = 1 also reads the zero tap
  return a.C_in % (8 * G) == 0 && G * ((tt + kp * a.dil + 63) / 64) <= BS_NSW * BS_XU &&
         a.x_cs * (long long)a.C_in < (1ll << 31);
}

bool conv_bsplit_p8_ok(const ConvArgs& a) {
  return conv_bsplit_ok(a) && bs_group(a.C_in) == 2 && (long long)a.T_in * 16 < (1ll << 32);
}

template <int KT, int G, int NMW, int NSW>
static int bsplit_launch(ConvArgs& a, hipStream_t s) {
  constexpr int H = bs_slots(KT, G), TT = 64 * NMW;
  a.XW = TT + (H / G - 1) * a.dil;       // G = 1: the padded zero tap still reads (finite) staged columns
  size_t lds = 2 * ((size_t)3 * H * BS_CO * 16 + (size_t)48 * G * a.XW);
  const size_t epi = (size_t)BS_CO * (TT + 4) * sizeof(float);      // the accumulator tile of the all-waves epilogue
  if (lds < epi) lds = epi;
  if (lds > 160 * 1024) {
    set_error("conv1d(bf16 split): tile needs %zu B of LDS (dil=%d)", lds, a.dil);
    return FAC_ERR_ARG;
  }
  auto kern = conv1d_bsplit_kernel<KT, G, NMW, NSW>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  a.n_t_tiles = (a.T_out + TT - 1) / TT;
  const long long n_wg = (long long)a.n_t_tiles * ((a.C_out + BS_CO - 1) / BS_CO) * a.B;
  if (n_wg > 0x7fffffffll) {
    set_error("conv1d: too many workgroups (%lld)", n_wg);
    return FAC_ERR_ARG;
  }
#if defined(FAC_PROF) || defined(FAC_PROF2)
  a.dbg = g_conv_dbg;
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3((NMW + NSW) * 64), lds, s, a);
  return check_launch("conv1d_bsplit");
}

int conv_dispatch_bsplit(ConvArgs& a, hipStream_t s) {
  if (a.K == 5)   // the discriminators' (5,1) convs and their data gradients, the WaveNet / style-encoder k = 5 convs
    return bs_group(a.C_in) == 2 ? bsplit_launch<5, 2, 4, BS_NSW_WIDE>(a, s) : bsplit_launch<5, 1, 8, BS_NSW>(a, s);
  if (a.K == 3)   // the encoder's output conv (1024 -> 1024)
    return bs_group(a.C_in) == 2 ? bsplit_launch<3, 2, 4, BS_NSW_WIDE>(a, s) : bsplit_launch<3, 1, 8, BS_NSW>(a, s);
  return bs_group(a.C_in) == 2 ? bsplit_launch<7, 2, 4, BS_NSW_WIDE>(a, s) : bsplit_launch<7, 1, 8, BS_NSW>(a, s);
}

}  // namespace fac

// tuning aid: resident workgroups per CU the runtime computes for the split kernel at a given LDS size
extern "C" int fac_debug_bsplit_occupancy(int lds_bytes) {
  int n = -1;
  auto kern = fac::conv1d_bsplit_kernel<7, 2, 4, fac::BS_NSW_WIDE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, (4 + fac::BS_NSW) * 64, (size_t)lds_bytes) != hipSuccess) return -1;
  return n;
}

extern "C" int64_t fac_conv_w_split_bytes(int C_out, int C_in, int K) {
  using namespace fac;
  const int G = bs_group(C_in);
  const int64_t n_ct = (C_out + BS_CO - 1) / BS_CO, n_st = (C_in + 8 * G - 1) / (8 * G);
  return n_ct * n_st * 3 * bs_slots(K, G) * BS_CO * 16;
}

extern "C" int fac_pack_conv_w_split(const float* v, const float* scale, void* out, int C_out, int C_in, int K,
                                     fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(v && out && C_out > 0 && C_in > 0 && K > 0, "pack_conv_w_split: bad arguments");
  const int G = bs_group(C_in);
  const int n_ct = (C_out + BS_CO - 1) / BS_CO, n_st = (C_in + 8 * G - 1) / (8 * G), H = bs_slots(K, G);
  const long long n = (long long)n_ct * n_st * H * BS_CO;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(pack_conv_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, scale,
                     reinterpret_cast<bf16x8*>(out), C_out, C_in, K, G, H, n_st, n);
  return check_launch("pack_conv_w_split");
}
