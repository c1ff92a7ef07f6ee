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
#include "prep_batch.h"

namespace fac {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BS_CO = 64;
constexpr int BS_NSW = 4;   // staging waves
#ifndef FAC_BS_NSW_WIDE
#define FAC_BS_NSW_WIDE 4
#endif
constexpr int BS_NSW_WIDE = FAC_BS_NSW_WIDE;
constexpr int BS_XU = 3;    // (ci group, 64-column block) staging units per staging wave
#ifndef FAC_BS_PERSIST
#define FAC_BS_PERSIST 1
#endif
static int conv_device_cus() {
  static int cus[16] = {0};
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (cus[dev] == 0 && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus[dev] = v;
  return cus[dev];
}
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
__device__ __forceinline__ void pack_conv_split_body(const float* __restrict__ v, const float* __restrict__ scale,
                                                     bf16x8* __restrict__ out, int C_out, int C_in, int K, int G, int H, int n_st,
                                                     long long n, int vb, int vg) {
  for (long long idx = (long long)vb * 256 + threadIdx.x; idx < n; idx += (long long)vg * 256) {
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

__global__ __launch_bounds__(256) void pack_conv_split_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                              bf16x8* __restrict__ out, int C_out, int C_in, int K, int G, int H,
                                                              int n_st, long long n) {
  pack_conv_split_body(v, scale, out, C_out, C_in, K, G, H, n_st, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void bsplit_batch_kernel(const PrepJob* __restrict__ jobs, const int* __restrict__ first, int njobs) {
  const int j = prep_find_job(first, njobs, blockIdx.x);
  const PrepJob& J = jobs[j];
  pack_conv_split_body(static_cast<const float*>(J.a), static_cast<const float*>(J.b), static_cast<bf16x8*>(J.out), J.i[0], J.i[1],
                       J.i[2], J.i[3], J.i[4], J.i[5], J.n, blockIdx.x - first[j], J.nblocks);
}

int prep_launch_bsplit(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s) {
  hipLaunchKernelGGL(bsplit_batch_kernel, dim3(total), dim3(256), 0, s, jobs, first, njobs);
  return check_launch("bsplit_batch");
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
  const int STG = W_STAGE + X_STAGE;        // one LDS stage: the weights of a chunk, then its inputs
  unsigned char* Wbuf = sm;                 // stage s at Wbuf + s * STG
  unsigned char* Xbuf = sm + W_STAGE;       // stage s at Xbuf + s * STG
  const int n_chunks = (a.C_in + 8 * G - 1) / (8 * G);
  const int dil = a.dil;

  // XCD-aware work decode (see conv1d_mfma.h): each XCD walks a contiguous range of (co tile, b, t tile).  v = virtual block id.
  const int n_tiles = a.n_tiles;
  // Per-tile code (tile decode, staging parameters, epilogue) reads the launch arguments through a pointer to the kernarg segment
  // that is laundered once per use site and tile: otherwise hipcc hoists every scalar load of the struct out of the tile loop and
  // keeps ~70 SGPRs live across the chunk loops (144 spilled, their v_readlane reloads landing in the staging waves' steps).
  typedef const __attribute__((address_space(4))) ConvArgs* KArgP;
  const KArgP kargs = (KArgP)__builtin_amdgcn_kernarg_segment_ptr();
  auto fresh_args = [&]() {
    KArgP q = kargs;
    asm volatile("" : "+s"(q));
    return q;
  };
  auto decode = [&](KArgP ka, int v, int& t0_, int& co0_, int& b_) {
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
    const int xcd = v & 7, within = v >> 3;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int nt = ka->n_t_tiles;
    const int tt = id % nt;
    const int rest = id / nt;
    const int nb = ka->B;
    b_ = rest % nb;
    co0_ = (rest / nb) * BS_CO;
    t0_ = tt * BS_TT;
  };
  // Round 4: the workgroup walks SEVERAL tiles (v = blockIdx.x, + gridDim.x, ...; one workgroup per CU).  With a.persist (host:
  // wide shape, fp32 inputs, an even number of chunks) the staging waves treat the tiles as ONE chunk stream: while the MFMA waves
  // multiply the last chunk of a tile (stage 1), chunk 0 of the next tile is staged into stage 0 and chunk 1 is requested, so a
  // tile costs its stages plus the epilogue -- not a workgroup launch, a cold prologue (one stage of staging behind a memory round
  // trip) and a drain.  The epilogue's fp32 tile then lives in stage 1 (free behind the last chunk) instead of at the LDS base.
  const bool overlap = a.persist != 0;
  unsigned char* epi_base = sm + (overlap ? STG : 0);
  // The roles run the tile walk as separate instantiations of one generic lambda: the register allocator then sees the staging
  // waves' cross-tile state and the MFMA waves' accumulators / fragment sets as unrelated live ranges (one loop around both roles
  // made the MFMA loop spill).
  auto walk = [&](auto role) {
  constexpr int ROLE = decltype(role)::value;          // 0: MFMA waves, 1: staging waves (fp32 inputs), 2: staging waves (P8 inputs)
  constexpr bool STAGING = ROLE != 0;
  bool first = true;                                  // this tile starts cold (always, without a.persist)
  // staging waves, overlap mode: load offsets / real-sample masks / bases of the CURRENT tile (c*) and of the next one (n*)
  unsigned cboff[BS_XU] = {}, nboff[BS_XU] = {};
  unsigned long long cmask[BS_XU] = {}, nmask[BS_XU] = {};
  const float *cxg = nullptr, *nxg = nullptr;
  const unsigned char *cws = nullptr, *nws = nullptr;
  for (int vb = blockIdx.x; vb < n_tiles; vb += gridDim.x) {
  int t0, co0, b;
  decode(fresh_args(), vb, t0, co0, b);
  const bool has_nt = overlap && vb + (int)gridDim.x < n_tiles;

#ifdef FAC_PROF
  unsigned long long pf0 = 0, pf1 = 0, pf2 = 0;
#endif
  if constexpr (STAGING) {
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
      unsigned char* dst = Wbuf + buf * STG;
      for (int i = lw; i * 64 < N16; i += NSW) {
        const int q = i * 64 + lane;
        if (q < N16)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(src + (long long)q * 16), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
      }
#endif
    };
    auto write_x = [&](int buf, const float (&xr)[BS_XU][8]) {       // xr: landed samples, padding lanes already zero
#if !defined(FAC_ABL_NOSTAGE) && !defined(FAC_ABL_NOSTAGE_X)
      unsigned char* xd = Xbuf + buf * STG;
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
    if constexpr (ROLE == 2) {
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
          unsigned char* xd = Xbuf + buf * STG;
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
          unsigned char* dst = Wbuf + buf * STG;
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
      // per-tile parameters of the loads: byte offset of the lane's column (padding lanes read a clamped column), mask of the lanes
      // that hold a real sample, the clip's rows, the co tile's weight slabs
      auto tile_params = [&](int t0_, int b_, int co0_, unsigned (&boff)[BS_XU], unsigned long long (&mask)[BS_XU], const float*& xgp,
                             const unsigned char*& wsp) {
        const KArgP ka = fresh_args();
        const int pad_left = ka->pad_left, pad_mode = ka->pad_mode, T_in = ka->T_in, T_ext = ka->T_ext;
#pragma unroll
        for (int j = 0; j < BS_XU; ++j) {
          const int tin = t0_ - pad_left + u_c[j];
          int idx;
          if (pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, T_in, T_ext);
          else idx = (tin >= 0 && tin < T_in) ? tin : -1;
          if (u_c[j] < 0) idx = -1;
          boff[j] = (unsigned)(idx >= 0 ? idx : 0) * 4u;
          mask[j] = __builtin_amdgcn_ballot_w64(idx >= 0);
        }
        xgp = ka->x + (long long)b_ * ka->x_bs;
        wsp = reinterpret_cast<const unsigned char*>(ka->w) + (long long)(co0_ / BS_CO) * n_chunks * W_STAGE;
      };
      if (first) tile_params(t0, b, co0, cboff, cmask, cxg, cws);
      if (has_nt) {
        int nt0, nco0, nb;
        decode(fresh_args(), vb + (int)gridDim.x, nt0, nco0, nb);
        tile_params(nt0, nb, nco0, nboff, nmask, nxg, nws);
      }
      auto load_a = [&](const float* xgp, const unsigned (&u_boff)[BS_XU], int chunk) {
#if !defined(FAC_ABL_NOSTAGE) && !defined(FAC_ABL_NOSTAGE_X)
        const float* grp[BS_XU];
#pragma unroll
        for (int j = 0; j < BS_XU; ++j) grp[j] = xgp + (long long)((chunk * G + u_g[j]) * 8) * xcs;
        FAC_XREGS24_A(BS_LD)
#endif
      };
      auto load_b = [&](const float* xgp, const unsigned (&u_boff)[BS_XU], int chunk) {
#if !defined(FAC_ABL_NOSTAGE) && !defined(FAC_ABL_NOSTAGE_X)
        const float* grp[BS_XU];
#pragma unroll
        for (int j = 0; j < BS_XU; ++j) grp[j] = xgp + (long long)((chunk * G + u_g[j]) * 8) * xcs;
        FAC_XREGS24_B(BS_LD)
#endif
      };
      auto take_a = [&](float (&xr)[BS_XU][8], const unsigned long long (&u_mask)[BS_XU]) { FAC_XREGS24_A(BS_RD) };
      auto take_b = [&](float (&xr)[BS_XU][8], const unsigned long long (&u_mask)[BS_XU]) { FAC_XREGS24_B(BS_RD) };
      // ONE software-pipelined loop: step(c) = { request the weights of chunk c + 1 (registers); wait for the inputs of c + 1
      // (requested by step(c - 1), older than those weight loads: vmcnt(ND)) and take them out of their landing registers; request
      // the inputs of c + 2; split + write c + 1 into stage (c + 1) & 1; wait for the weights (vmcnt(NX): the inputs of c + 2 stay
      // in flight) and write them; barrier }, each part skipped where its chunk does not exist.  At most ND + NX loads are in
      // flight.  A cold tile starts at c = -2; in overlap mode chunks n_chunks and n_chunks + 1 are chunks 0 and 1 of the NEXT
      // tile (n_chunks is even: same stage and register-set parity), and a tile that was started that way begins at c = 0.
      // The weight slab goes through registers (global_load_dwordx4 -> ds_write_b128, requested before the input work of the
      // stage, written behind it -- plain compiler-allocated registers, no value crosses a barrier) instead of LDS-DMA: measured
      // +2 .. 4 % (profiles/r04_bsplit_ablation.log: what the weight stage costs is its LDS write traffic either way).
      typedef float wv4 __attribute__((ext_vector_type(4)));
      const int wblk0 = min(lw * ND, NBLK - ND);
      const unsigned lane16 = (unsigned)lane * 16u;
#ifdef FAC_PROF2
      long long pq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BSQ(k) q[k] = clock64();
#else
#define BSQ(k)
#endif
      for (int base = first ? -2 : 0; base < n_chunks; base += 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                // c + 1 has parity 1 - i: its stage and its register set
          const int c = base + i;
          if (c >= n_chunks) break;
          const bool nx1 = c + 1 >= n_chunks, nx2 = c + 2 >= n_chunks;          // the chunk belongs to the next tile
          const bool has_next = c + 1 >= 0 && (!nx1 || has_nt), has_next2 = !nx2 || has_nt;
          float xr[BS_XU][8];
          wv4 wv[ND];
#ifdef FAC_PROF2
          long long q[7];
          for (int k = 0; k < 7; ++k) q[k] = clock64();
#endif
          if (has_next) {
            // this wave's ND consecutive 1 KiB blocks of the slab (the last wave's range is shifted back onto its neighbour's
            // instead of running over the end: the same bytes go to the same place twice) -- one lane offset for all of them, block
            // j as a scalar base on the load side and as an immediate offset on the LDS side: no per-block vector arithmetic
            const unsigned char* src = (nx1 ? nws : cws) + (long long)(nx1 ? c + 1 - n_chunks : c + 1) * W_STAGE + wblk0 * 1024;
#pragma unroll
            for (int j = 0; j < ND; ++j)
              asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wv[j]) : "v"(lane16), "s"(src + j * 1024) : "memory");
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(ND) : "memory");
            BSQ(1)
            unsigned long long u_mask[BS_XU];
#pragma unroll
            for (int j = 0; j < BS_XU; ++j) u_mask[j] = nx1 ? nmask[j] : cmask[j];
            if (i == 0) take_b(xr, u_mask); else take_a(xr, u_mask);
            BSQ(2)
          }
          if (has_next2) {
            unsigned u_boff[BS_XU];
#pragma unroll
            for (int j = 0; j < BS_XU; ++j) u_boff[j] = nx2 ? nboff[j] : cboff[j];
            const float* xgp = nx2 ? nxg : cxg;
            const int chunk = nx2 ? c + 2 - n_chunks : c + 2;
            if (i == 0) load_a(xgp, u_boff, chunk); else load_b(xgp, u_boff, chunk);
          }
          BSQ(3)
          if (has_next) {
            write_x(1 - i, xr);
#ifdef FAC_PROF2
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            BSQ(4)
            if (has_next2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NX) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BSQ(5)
            unsigned char* dst = Wbuf + (1 - i) * STG + wblk0 * 1024 + lane16;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
              asm volatile("" : "+v"(wv[j]) : : "memory");
              *reinterpret_cast<wv4*>(dst + j * 1024) = wv[j];
            }
          }
          if (c >= -1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            BSQ(6)
            __builtin_amdgcn_s_barrier();      // c = -1: chunk 0 staged; later: pairs with the MFMA waves' barrier behind chunk c
            asm volatile("" ::: "memory");
#ifdef FAC_PROF2
            if (c >= 0 && c + 2 < n_chunks) {      // steady-state iterations only
              pq[0] += q[1] - q[0]; pq[1] += q[2] - q[1]; pq[2] += q[3] - q[2]; pq[3] += q[4] - q[3]; pq[4] += q[5] - q[4];
              pq[5] += q[6] - q[5]; pq[6] += clock64() - q[6]; pq[7] += 1;
            }
#endif
          }
        }
      }
      if (has_nt) {       // the next tile's parameters become the current ones
#pragma unroll
        for (int j = 0; j < BS_XU; ++j) {
          cboff[j] = nboff[j];
          cmask[j] = nmask[j];
        }
        cxg = nxg;
        cws = nws;
      }
#ifdef FAC_PROF2
      if (a.dbg && lw == 0 && lane == 0) {
        unsigned long long* d = a.dbg + (long long)blockIdx.x * 16 + 8;
        for (int k = 0; k < 8; ++k) d[k] = (unsigned long long)pq[k];
      }
#endif
#undef BSQ
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

  if (first) __syncthreads();   // chunk 0 staged (a tile started by the previous tile's last step needs no barrier: overlap mode)
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
      const unsigned char* Wb = Wbuf + buf * STG + (kq * BS_CO + l31) * 16;          // half slot 2 st + kq
      const unsigned char* Xb = Xbuf + buf * STG + (n0 + l31) * 16 + x_lane + st * x_step;
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
    const unsigned char* Wb = Wbuf + buf * STG + (kq * BS_CO + l31) * 16;          // half slot 2s + kq
    const unsigned char* Xb = Xbuf + buf * STG + (n0 + l31) * 16 + x_lane;
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
  // ---- accumulators -> LDS (both stage buffers are free now; overlap mode: stage 1 is, stage 0 already holds the next tile's
  // chunk 0): tile[co][t] fp32, row pitch BS_TT + 4 floats.
  // C/D layout of the 32x32 block: register r <-> row (r & 3) + 8 (r >> 2) + 4 kq, column l31.
  {
    float* tile = reinterpret_cast<float*>(epi_base);
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
    const KArgP e = fresh_args();
    const float* tile = reinterpret_cast<const float*>(epi_base);
    constexpr int EP = BS_TT + 4;
    constexpr int NTH = (NMW + NSW) * 64;
    constexpr int QPR = BS_TT / 4;                       // quads per row
    float* yg = e->y ? e->y + (long long)b * e->y_bs : nullptr;
    float* y2g = e->y2 ? e->y2 + (long long)b * e->y_bs : nullptr;
    const float* rg = e->res ? e->res + (long long)b * e->y_bs : nullptr;
    const bool vec_ok = (e->y_cs & 3) == 0 && (e->y_bs & 3) == 0 && (!yg || (reinterpret_cast<unsigned long long>(e->y) & 15) == 0) &&
                        (!y2g || (reinterpret_cast<unsigned long long>(e->y2) & 15) == 0) &&
                        (!rg || (reinterpret_cast<unsigned long long>(e->res) & 15) == 0);
    for (int q = tid; q < BS_CO * QPR; q += NTH) {
      const int row = q / QPR, tq = q - row * QPR;
      const int co = co0 + row, t = t0 + 4 * tq;
      if (co >= e->C_out || t >= e->T_out) continue;
      const float4 av = *reinterpret_cast<const float4*>(tile + row * EP + 4 * tq);
      float v[4] = {av.x, av.y, av.z, av.w};
      const float bs = e->bias ? e->bias[co] : 0.f;
      const float al = e->alpha_out ? e->alpha_out[co] : 0.f;
      const float inv = e->alpha_out ? snake_inv(al) : 0.f;
      const long long o = (long long)co * e->y_cs + t;
      const bool full = vec_ok && t + 3 < e->T_out;
      float rv[4] = {0.f, 0.f, 0.f, 0.f};
      if (rg) {
        if (full) {
          const float4 r4 = *reinterpret_cast<const float4*>(rg + o);
          rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) rv[i] = t + i < e->T_out ? rg[o + i] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = v[i] + bs;
        if (e->alpha_out) x = snake_apply(x, al, inv);
        if (e->act != FAC_ACT_NONE) x = apply_act_slow(x, e->act);
        v[i] = x + rv[i];
      }
      float w[4];
      if (y2g) {
        const float a2 = e->alpha2[co], i2 = snake_inv(a2);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = snake_apply(v[i], a2, i2);
      }
      if (full) {
        if (yg) *reinterpret_cast<float4*>(yg + o) = make_float4(v[0], v[1], v[2], v[3]);
        if (y2g) *reinterpret_cast<float4*>(y2g + o) = make_float4(w[0], w[1], w[2], w[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (t + i >= e->T_out) continue;
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
  first = !overlap;
  if (vb + (int)gridDim.x < n_tiles) __syncthreads();     // the epilogue has read its tile: the stage buffers may be written again
  }   // tiles of this workgroup
  };
  if (wave >= NMW) {
    // P8 inputs get their own instantiation: it keeps nothing in flight across statements, and the ISA check of the named landing
    // registers (tools/check_inflight_regs.py: named_lifetime_violations) then sees no path from a load site into its code
    if constexpr (G == 2) {
      if (a.x_p8 != nullptr) walk(std::integral_constant<int, 2>{});
      else walk(std::integral_constant<int, 1>{});
    } else {
      walk(std::integral_constant<int, 1>{});
    }
  } else {
    walk(std::integral_constant<int, 0>{});
  }
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
  const size_t stg = (size_t)3 * H * BS_CO * 16 + (size_t)48 * G * a.XW;
  const size_t epi = (size_t)BS_CO * (TT + 4) * sizeof(float);      // the accumulator tile of the all-waves epilogue
  a.n_t_tiles = (a.T_out + TT - 1) / TT;
  const long long n_wg = (long long)a.n_t_tiles * ((a.C_out + BS_CO - 1) / BS_CO) * a.B;
  if (n_wg > 0x7fffffffll) {
    set_error("conv1d: too many workgroups (%lld)", n_wg);
    return FAC_ERR_ARG;
  }
  // One workgroup per CU walking several tiles, the next tile's first chunk staged under the last chunk of the current one
  // (kernel header): wide shape with fp32 inputs, an even number of chunks (stage / register-set parity continues across tiles),
  // more tiles than CUs, and the epilogue tile must fit behind stage 0.  FAC_BS_PERSIST=0 restores one tile per workgroup.
  const int n_chunks = (a.C_in + 8 * G - 1) / (8 * G);
  int cus = conv_device_cus() & ~7;
  a.persist = (FAC_BS_PERSIST && G == 2 && NMW == 4 && a.x_p8 == nullptr && n_chunks % 2 == 0 && cus >= 8 && n_wg > cus &&
               stg + (stg > epi ? stg : epi) <= 160 * 1024) ? 1 : 0;
  size_t lds = a.persist ? stg + (stg > epi ? stg : epi) : (2 * stg > epi ? 2 * stg : epi);
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
  a.n_tiles = (int)n_wg;
#if defined(FAC_PROF) || defined(FAC_PROF2)
  a.dbg = g_conv_dbg;
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.persist ? cus : n_wg)), dim3((NMW + NSW) * 64), lds, s, a);
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
  if (prep_recording()) {
    PrepJob j{}; j.a = v; j.b = scale; j.out = out; j.kind = PK_CONV_SPLIT; j.nblocks = blocks; j.n = n;
    j.i[0] = C_out; j.i[1] = C_in; j.i[2] = K; j.i[3] = G; j.i[4] = H; j.i[5] = n_st;
    return prep_record(PU_BSPLIT, j);
  }
  hipLaunchKernelGGL(pack_conv_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, scale,
                     reinterpret_cast<bf16x8*>(out), C_out, C_in, K, G, H, n_st, n);
  return check_launch("pack_conv_w_split");
}
