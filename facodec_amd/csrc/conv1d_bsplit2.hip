// Few-output-channel convs with many taps on the bf16 matrix pipe, fp32-grade (the 3-way operand split of conv1d_bsplit.hip):
// the (3, 9) / (3, 3) Conv2d stacks of the multi-resolution discriminator (dac/model/discriminator.py:101-170) in their
// row-concatenated 1-D form (two-level taps, stride 1 or 2 along frequency, 32 output channels).
//
// On the fp32 matrix pipe these 32-channel layers run at ~90 TFLOP/s (conv1d_mfma_kernel<1,2,1,4,9>, 12 % of the training
// step).  conv1d_bsplit.hip's 64-row tile would idle half of every MFMA on them, and its stage cannot follow a stride or the
// second tap level.  This kernel keeps its operand formats and adds what these layers need:
//   tile   : 32 output channels x 512 columns per workgroup: 4 MFMA waves of 32 x 128 (1 x 4 blocks: one A fragment feeds four
//            B fragments -- 15 ds_read_b128 per 24 MFMAs, where 32 x 64 per wave would need 9 per 12) + 4 staging waves;
//   stage  : 8 VIRTUAL input channels x K1 taps.  Virtual channel v = ci * K2 + k2 is row ci read k2 * dilation2 columns
//            further on (conv1d_mfma.h uses the same numbering): the (K2 x K1) taps become K2 x C_in channels of a K1-tap conv,
//            and the weights (C_out, C_in, K2 * K1) need no re-ordering, they ARE (C_out, C_in * K2, K1);
//   taps   : two per MFMA (half-wave 0 tap 2s, half-wave 1 tap 2s + 1; an odd K1 is padded with a zero-weight tap);
//   stride : the staged columns are de-interleaved by input phase (column c at position (c % S) * XWh + c / S), so the B
//            fragment of tap k for 32 consecutive outputs is 32 consecutive 16-byte pieces of phase k % S -- aligned and
//            bank-conflict-free like the stride-1 case;
//   inputs : fp32 rows, split by the staging waves (as conv1d_bsplit.hip); zero padding only (the discriminator's layout
//            carries its padding as zero gaps);
//   epilogue by all eight waves from an fp32 tile in LDS (bias, activation, residual), 16-byte stores.
#include "conv1d_mfma.h"
#include "inflight_regs.h"
#include "prep_batch.h"

namespace fac {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B2_CO = 32;                // output channels per tile
constexpr int B2_NB = 4;                 // 32-column blocks per MFMA wave
constexpr int B2_TT = 32 * B2_NB * 4;    // 512 columns per tile

__device__ __forceinline__ void b2_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

__host__ __device__ constexpr int b2_slots(int K1) { return (K1 + 1) & ~1; }

// v (C_out, CV, K1) [* scale per C_out] (CV = C_in * K2 virtual channels, contiguous (K2, K1) taps per real channel) ->
// [co tile of 32][chunk of 8 cv][plane][tap slot][32 co][8 cv] bf16.  One thread per (tile, chunk, slot, co).
__device__ __forceinline__ void pack_conv_split2_body(const float* __restrict__ v, const float* __restrict__ scale,
                                                      bf16x8* __restrict__ out, int C_out, int CV, int K1, int H, int n_ch, long long n,
                                                      int vb, int vg) {
  for (long long idx = (long long)vb * 256 + threadIdx.x; idx < n; idx += (long long)vg * 256) {
    const int co = (int)(idx % B2_CO);
    long long r = idx / B2_CO;
    const int k = (int)(r % H);
    r /= H;
    const int ch = (int)(r % n_ch);
    const int ct = (int)(r / n_ch);
    const int cog = ct * B2_CO + co;
    const float sc = (scale != nullptr && cog < C_out) ? scale[cog] : 1.0f;
    bf16x8 h, m, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int cv = ch * 8 + i;
      float w = 0.f;
      if (cog < C_out && cv < CV && k < K1) {
        w = v[((long long)cog * CV + cv) * K1 + k];
        if (scale != nullptr) w = __fmul_rn(w, sc);
      }
      __bf16 a, b2, c;
      b2_split3(w, a, b2, c);
      h[i] = a; m[i] = b2; l[i] = c;
    }
    const long long base = ((long long)ct * n_ch + ch) * 3;
    out[((base + 0) * H + k) * B2_CO + co] = h;
    out[((base + 1) * H + k) * B2_CO + co] = m;
    out[((base + 2) * H + k) * B2_CO + co] = l;
  }
}

__global__ __launch_bounds__(256) void pack_conv_split2_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                               bf16x8* __restrict__ out, int C_out, int CV, int K1, int H, int n_ch,
                                                               long long n) {
  pack_conv_split2_body(v, scale, out, C_out, CV, K1, H, n_ch, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void bsplit2_batch_kernel(const PrepJob* __restrict__ jobs, const int* __restrict__ first, int njobs) {
  const int j = prep_find_job(first, njobs, blockIdx.x);
  const PrepJob& J = jobs[j];
  pack_conv_split2_body(static_cast<const float*>(J.a), static_cast<const float*>(J.b), static_cast<bf16x8*>(J.out), J.i[0], J.i[1],
                        J.i[2], J.i[3], J.i[4], J.n, blockIdx.x - first[j], J.nblocks);
}

int prep_launch_bsplit2(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s) {
  hipLaunchKernelGGL(bsplit2_batch_kernel, dim3(total), dim3(256), 0, s, jobs, first, njobs);
  return check_launch("bsplit2_batch");
}

// KT: taps per virtual channel (K1);  S: stride (1 or 2);  XU: (64-column block) staging units per staging wave
template <int KT, int S, int XU>
__global__ __launch_bounds__(512, 2) void conv1d_bsplit2_kernel(ConvArgs a) {
  constexpr int H = b2_slots(KT);                   // tap slots per stage (even)
  constexpr int W_STAGE = 3 * H * B2_CO * 16;       // bytes
  constexpr int XWh = B2_TT + (KT - 1 + S - 1) / S; // staged positions per input phase
  constexpr int XWT = S * XWh;                      // positions per plane
  constexpr int X_STAGE = 3 * XWT * 16;
  constexpr int XIN = (B2_TT - 1) * S + KT;         // input columns a tile reads
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // [0, 4): MFMA waves, then the staging waves
  unsigned char* Wbuf = sm;                 // [2][W_STAGE]
  unsigned char* Xbuf = sm + 2 * W_STAGE;   // [2][X_STAGE]

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
    co0 = (rest / a.B) * B2_CO;
    t0 = tt * B2_TT;
  }
  const int n_chunks = (a.CV + 7) / 8;

  if (wave >= 4) {
    // ===================== staging waves
    const int lw = wave - 4;
    __builtin_amdgcn_s_setprio(FAC_PRIO_STAGE);
    const float* xg = a.x + (long long)b * a.x_bs;
    const int xcs = (int)a.x_cs;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w) + (long long)(co0 / B2_CO) * n_chunks * W_STAGE;
    const int K2v = a.K2v, dil2 = a.dil2;
    // units of this wave: window column c = (lw + 4 j) * 64 + lane, input sample tin0 + c (+ k2 * dil2 per channel)
    int u_c[XU], u_pos[XU];
#pragma unroll
    for (int j = 0; j < XU; ++j) {
      const int c = (lw + 4 * j) * 64 + lane;
      u_c[j] = c < XIN ? c : -1;
      u_pos[j] = ((c % S) * XWh + c / S) * 16;
    }
    const int tin0 = t0 * S - a.pad_left;
    // Weights: one contiguous slab per stage, 16 B per lane by LDS-DMA; every staging wave issues exactly ND instructions (a wave
    // short of one block re-copies the last block), so `s_waitcnt vmcnt(n)` on the in-order load queue can name what it waits for.
    constexpr int NBLK = W_STAGE / 1024;
    constexpr int ND = (NBLK + 3) / 4;
    constexpr int NX = XU * 8;
    static_assert(W_STAGE % 1024 == 0 && ND + NX <= 63, "vmcnt is a 6-bit counter");
    auto stage_w = [&](int chunk, int buf) {
      const unsigned char* src = wsrc + (long long)chunk * W_STAGE;
      unsigned char* dst = Wbuf + buf * W_STAGE;
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        const int i = min(lw + 4 * j, NBLK - 1);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + (long long)i * 1024 + lane * 16), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
      }
    };
    // which (virtual channel i of the chunk, unit j) elements are real samples (the others are the conv's zero padding)
    auto in_range = [&](int chunk, int i, int j, int& tin) {
      const int v = chunk * 8 + i;
      const bool vok = v < a.CV;
      const int vv = vok ? v : 0;
      const int k2 = vv % K2v;
      tin = tin0 + k2 * dil2 + u_c[j];
      return vok && u_c[j] >= 0 && tin >= 0 && tin < a.T_in;
    };
    // The inputs of chunk c + 2 stay in flight across the barrier, landing in NAMED physical registers (inflight_regs.h; see
    // conv1d_bsplit.hip for why neither plain C++ loads nor compiler-allocated asm destinations work): element n = 8 j + i of a set
    // = (unit j, virtual channel i of the chunk); scalar row base + per-lane byte offset, one instruction per element; lanes on
    // padding read offset 0 and are zeroed by the v_cndmask that takes the value out of its landing register.
#define B2_LD(n, R)                                                                                                      \
  {                                                                                                                      \
    int tin;                                                                                                             \
    const bool ok = in_range(chunk, (n) % 8, (n) / 8, tin);                                                              \
    asm volatile("global_load_dword v" #R ", %0, %1" : : "v"(ok ? (unsigned)tin * 4u : 0u), "s"(row[(n) % 8]) : "memory", "v" #R); \
  }
#define B2_RD(n, R)                                                                                                      \
  {                                                                                                                      \
    int tin;                                                                                                             \
    const unsigned long long m = __builtin_amdgcn_ballot_w64(in_range(chunk, (n) % 8, (n) / 8, tin));                    \
    asm volatile("v_cndmask_b32_e64 %0, 0, v" #R ", %1" : "=v"(xr[(n) / 8][(n) % 8]) : "s"(m) : "memory");              \
  }
    auto load_x = [&](int chunk, int set) {
      const float* row[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int v = chunk * 8 + i;                              // uniform: (real channel, second-level tap) of virtual channel v
        v = v < a.CV ? v : 0;
        row[i] = xg + (long long)(v / K2v) * xcs;
      }
      if constexpr (XU == 5) {
        if (set == 0) { FAC_XREGS40_A(B2_LD) } else { FAC_XREGS40_B(B2_LD) }
      } else {
        if (set == 0) { FAC_XREGS24_A(B2_LD) } else { FAC_XREGS24_B(B2_LD) }
      }
    };
    auto take_x = [&](int chunk, int set, float (&xr)[XU][8]) {      // after the s_waitcnt that covers the set's loads
      if constexpr (XU == 5) {
        if (set == 0) { FAC_XREGS40_A(B2_RD) } else { FAC_XREGS40_B(B2_RD) }
      } else {
        if (set == 0) { FAC_XREGS24_A(B2_RD) } else { FAC_XREGS24_B(B2_RD) }
      }
    };
    auto write_x = [&](int buf, const float (&xr)[XU][8]) {           // xr: landed samples, padding already zero
      unsigned char* xd = Xbuf + buf * X_STAGE;
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        if (u_c[j] < 0) continue;
        bf16x8 h, m, l;
#if defined(FAC_ABL2_NOSTAGE_X)
        continue;                      // ablation builds only (tools/tune/abl_bsplit2.py): results are wrong, timing is the point
#elif defined(FAC_ABL2_NOSPLIT)
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (__bf16)xr[j][i];
        m = h; l = h;
#else
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __bf16 p0, p1, p2;
          b2_split3(xr[j][i], p0, p1, p2);
          h[i] = p0; m[i] = p1; l[i] = p2;
        }
#endif
        *reinterpret_cast<bf16x8*>(xd + u_pos[j]) = h;
        *reinterpret_cast<bf16x8*>(xd + XWT * 16 + u_pos[j]) = m;
        *reinterpret_cast<bf16x8*>(xd + 2 * XWT * 16 + u_pos[j]) = l;
      }
    };
    // ONE software-pipelined loop from c = -2: step(c) = { weight DMA of chunk c + 1; wait for the inputs of c + 1 (requested by
    // step(c - 1), older than that DMA: vmcnt(ND)); request the inputs of c + 2; take c + 1 out of its landing registers, split,
    // write; wait for the DMA (vmcnt(NX)); barrier }.  Chunk parity = LDS stage = register set.  At most ND + NX loads in flight.
    for (int base = -2; base < n_chunks; base += 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = base + i;
        if (c >= n_chunks) break;
        const bool has_next = c + 1 >= 0 && c + 1 < n_chunks, has_next2 = c + 2 < n_chunks;
        float xr[XU][8];
        if (has_next) {
          stage_w(c + 1, 1 - i);
          asm volatile("s_waitcnt vmcnt(%0)" : : "n"(ND) : "memory");
          take_x(c + 1, 1 - i, xr);
        }
        if (has_next2) load_x(c + 2, i);
        if (has_next) write_x(1 - i, xr);
        if (c >= -1) {
          if (has_next2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(NX) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
    }
#undef B2_LD
#undef B2_RD
    __builtin_amdgcn_s_setprio(0);
  } else {
    // ========================= MFMA waves: 32 rows x 128 columns each
    __builtin_amdgcn_s_setprio(FAC_PRIO_MFMA);
    const int l31 = lane & 31;
    const int kq = lane >> 5;
    const int n0 = wave * (32 * B2_NB);
    f32x16 acc[B2_NB];
#pragma unroll
    for (int n = 0; n < B2_NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // step st: this half-wave's tap k = 2 st + kq (the padded tap of an odd K1 has zero weights: any staged position will do)
    int xk[H / 2];
#pragma unroll
    for (int st = 0; st < H / 2; ++st) {
      int k = 2 * st + kq;
      k = k < KT ? k : KT - 1;
      xk[st] = ((k % S) * XWh + k / S + n0 + l31) * 16;
    }

    __syncthreads();   // chunk 0 staged
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
      const int buf = chunk & 1;
      const unsigned char* Wb = Wbuf + buf * W_STAGE + (kq * B2_CO + l31) * 16;          // tap slot 2 st + kq
      const unsigned char* Xb = Xbuf + buf * X_STAGE;
      bf16x8 A[2][3], Bf[B2_NB][3];
      auto ldA = [&](int st, bf16x8 (&Ad)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) Ad[p] = *reinterpret_cast<const bf16x8*>(Wb + ((p * H + 2 * st) * B2_CO) * 16);
      };
      auto ldB = [&](int st) {
        constexpr int PO[3] = {1, 0, 2};   // planes in order of first use: mid, hi, lo
#pragma unroll
        for (int pi = 0; pi < 3; ++pi)
#pragma unroll
          for (int n = 0; n < B2_NB; ++n)
            Bf[n][PO[pi]] = *reinterpret_cast<const bf16x8*>(Xb + xk[st] + (PO[pi] * XWT + n * 32) * 16);
      };
      ldA(0, A[0]);
#pragma unroll
      for (int st = 0; st < H / 2; ++st) {
        ldB(st);
        if (st + 1 < H / 2) ldA(st + 1, A[(st + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};     // smallest terms first
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int n = 0; n < B2_NB; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[st & 1][TA[q]], Bf[n][TB[q]], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
    // accumulators -> fp32 tile in LDS (both stage buffers are free now)
    float* tile = reinterpret_cast<float*>(sm);
    constexpr int EP = B2_TT + 4;
#pragma unroll
    for (int n = 0; n < B2_NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * kq) * EP + n0 + n * 32 + l31] = acc[n][r];
  }
  __syncthreads();

  // ---- epilogue by all eight waves: one lane = 4 consecutive time steps of one output channel
  {
    const float* tile = reinterpret_cast<const float*>(sm);
    constexpr int EP = B2_TT + 4;
    constexpr int QPR = B2_TT / 4;
    float* yg = a.y ? a.y + (long long)b * a.y_bs : nullptr;
    float* y2g = a.y2 ? a.y2 + (long long)b * a.y_bs : nullptr;
    const float* rg = a.res ? a.res + (long long)b * a.y_bs : nullptr;
    const bool vec_ok = (a.y_cs & 3) == 0 && (a.y_bs & 3) == 0 && (!yg || (reinterpret_cast<unsigned long long>(a.y) & 15) == 0) &&
                        (!y2g || (reinterpret_cast<unsigned long long>(a.y2) & 15) == 0) &&
                        (!rg || (reinterpret_cast<unsigned long long>(a.res) & 15) == 0);
    for (int q = tid; q < B2_CO * QPR; q += 512) {
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
          if (yg) yg[o + i] = v[i];
          if (y2g) y2g[o + i] = w[i];
        }
      }
    }
  }
}

// (K1, stride) pairs built: (9, 1), (9, 2), (3, 1).  Zero padding, no Snake prologue, at most 32 output channels.
bool conv_bsplit2_ok(const ConvArgs& a) {
  static const bool on = !(getenv("FAC_BSPLIT2") && getenv("FAC_BSPLIT2")[0] == '0');
  if (!on) return false;
  if (!(a.dil == 1 && a.n_phase == 1 && a.phase_shift == 0 && a.y_tstride == 1 && a.rp == 1 && !a.alpha_in && !a.w1 && !a.w_batched &&
        a.pad_mode == FAC_PAD_ZERO))
    return false;
  const bool shape = (a.KV == 9 && (a.stride == 1 || a.stride == 2)) || (a.KV == 3 && a.stride == 1);
  if (!shape || a.C_out > B2_CO || a.C_out < 8) return false;
  if ((long long)a.B * a.T_out < 4096) return false;
  return a.x_cs * (long long)a.C_in < (1ll << 31) && (long long)a.T_in + (long long)a.K2v * a.dil2 < (1ll << 30);
}

template <int KT, int S, int XU>
static int bsplit2_launch(ConvArgs& a, hipStream_t s) {
  constexpr int H = b2_slots(KT);
  constexpr int XWh = B2_TT + (KT - 1 + S - 1) / S;
  size_t lds = 2 * ((size_t)3 * H * B2_CO * 16 + (size_t)3 * S * XWh * 16);
  const size_t epi = (size_t)B2_CO * (B2_TT + 4) * sizeof(float);
  if (lds < epi) lds = epi;
  static_assert(((B2_TT - 1) * S + KT + 63) / 64 <= 4 * XU, "staging units do not cover the input window");
  auto kern = conv1d_bsplit2_kernel<KT, S, XU>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (lds > 160 * 1024) {
    set_error("conv1d(bf16 split, 32-row tile): %zu B of LDS", lds);
    return FAC_ERR_ARG;
  }
  a.n_t_tiles = (a.T_out + B2_TT - 1) / B2_TT;
  const long long n_wg = (long long)a.n_t_tiles * ((a.C_out + B2_CO - 1) / B2_CO) * a.B;
  if (n_wg > 0x7fffffffll) {
    set_error("conv1d: too many workgroups (%lld)", n_wg);
    return FAC_ERR_ARG;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), lds, s, a);
  return check_launch("conv1d_bsplit2");
}

int conv_dispatch_bsplit2(ConvArgs& a, hipStream_t s) {
  if (a.KV == 9) return a.stride == 2 ? bsplit2_launch<9, 2, 5>(a, s) : bsplit2_launch<9, 1, 3>(a, s);
  return bsplit2_launch<3, 1, 3>(a, s);
}

}  // namespace fac

extern "C" int64_t fac_conv_w_split2_bytes(int C_out, int C_in, int K, int K1) {
  using namespace fac;
  if (K1 <= 0 || K1 > K) K1 = K;
  const int64_t CV = (int64_t)C_in * (K / K1);
  const int64_t n_ct = (C_out + B2_CO - 1) / B2_CO, n_ch = (CV + 7) / 8;
  return n_ct * n_ch * 3 * b2_slots(K1) * B2_CO * 16;
}

extern "C" int fac_pack_conv_w_split2(const float* v, const float* scale, void* out, int C_out, int C_in, int K, int K1,
                                      fac_stream_t stream) {
  using namespace fac;
  if (K1 <= 0 || K1 > K) K1 = K;
  FAC_REQUIRE(v && out && C_out > 0 && C_in > 0 && K > 0 && K % K1 == 0, "pack_conv_w_split2: bad arguments");
  const int CV = C_in * (K / K1);
  const int n_ct = (C_out + B2_CO - 1) / B2_CO, n_ch = (CV + 7) / 8, H = b2_slots(K1);
  const long long n = (long long)n_ct * n_ch * H * B2_CO;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  if (prep_recording()) {
    PrepJob j{}; j.a = v; j.b = scale; j.out = out; j.kind = PK_CONV_SPLIT2; j.nblocks = blocks; j.n = n;
    j.i[0] = C_out; j.i[1] = CV; j.i[2] = K1; j.i[3] = H; j.i[4] = n_ch;
    return prep_record(PU_BSPLIT2, j);
  }
  hipLaunchKernelGGL(pack_conv_split2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, scale,
                     reinterpret_cast<bf16x8*>(out), C_out, CV, K1, H, n_ch, n);
  return check_launch("pack_conv_w_split2");
}
