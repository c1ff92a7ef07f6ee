// K5p: the SLSTM recurrence (dac/model/encodec.py:272-288, nn.LSTM gate order i,f,g,o, zero initial state) and its
// back-propagation through time as ONE launch per layer.
//
// lstm.hip launches one kernel per time step; every step then re-streams its 128-192 KB slice of W_hh per workgroup
// and pays a kernel boundary.  Here the H/8 workgroups of a layer (192 at H = 1536, 128 at H = 1024: fewer than the
// 256 CUs, one workgroup of 16 waves per CU) stay resident for the whole sequence:
//   * each lane keeps its 2 * H/64 weights of W_hh in registers for all T steps (the slice of a workgroup is exactly
//     the register file of its waves: 48 VGPRs at H = 1536) -- W_hh is read from memory once per layer, not once per step;
//   * the batch is handled in 16-column blocks on v_mfma_f32_16x16x4_f32 (B = 16 is one block; the per-step kernel's
//     32x32x2 tile spends half its MFMA time on the padded columns 16..31);
//   * the cell state lives in a register of the thread that owns the (unit, column);
//   * steps are separated by a device-wide flag exchange instead of a kernel boundary: workgroup i writes h_t with
//     write-through (agent-scope, sc1) stores into a FRESH region per step, waits for their acknowledgement and stores
//     "finished step t" into flags[i]; a waiter reads all flags with ONE wave-wide sc1 load per poll (no read-modify-write,
//     nothing serialises on a counter) and then reads h_t with plain loads -- no XCD can hold an older copy of lines that
//     did not exist before.  No buffer_wbl2 / buffer_inv on the step path: the memory-model fences of a textbook grid
//     barrier measured 25-92 us per step on this part (exchange modes below).  The flags hold epoch + step;
//     the epoch advances by T per launch (last workgroup out), so a replayed hipGraph needs no host-side reset.  A
//     waiter that sees no progress for ~4 s traps (loud launch failure) instead of hanging the queue.
// Requirements (fac_lstm_persist_ok): H a multiple of 256 with H/64 in {8, 16, 24}, B <= 32, H/8 <= CUs of the device,
// zero initial state.  Everything else (streaming sessions that carry state, larger batches) stays on lstm.hip.
//
// Operand layouts (v_mfma_f32_16x16x4_f32: A lane l = row l%16, k l/16; B lane l = k l/16, col l%16; D lane l, reg r =
// row 4*(l/16)+r, col l%16).  Wave w of a workgroup contracts k in [w*H/16, (w+1)*H/16) in KS = H/64 MFMA steps:
//   weights  packed[(blk*16 + w)*(KS/2) + j][lane] float4 = {A(s=2j, rb=0), A(2j, 1), A(2j+1, 0), A(2j+1, 1)},
//            A(s, rb) = M[row(blk, rb*16 + l%16)][k0(blk) + w*H/16 + 4*s + l/16]            (fac_pack_lstm_whh16)
//   state    frag[(cb*16 + w)*(KS/4) + j4][lane] float4 component jj = X[k = w*H/16 + 4*(4*j4+jj) + l/16][cb*16 + l%16]
//   k <-> hidden unit: permuted inside every group of 16 units (k_of_unit) so that a workgroup's 8 units x 16 columns are
//   four whole 128-byte lines of the state buffer; the weights are packed with the same permutation (unit_of_k).
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>
#include <unordered_map>

#include "common.h"

namespace fac {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int LSTM_SYNC_SLOTS = 64;
constexpr int LSTM_MAX_WG = 256;
struct alignas(256) LstmSync {
  unsigned epoch;                 // flags of finished launches are <= epoch
  unsigned done;                  // workgroups that left the current launch
  unsigned pad[62];
  unsigned flags[LSTM_MAX_WG];    // flags[i] = epoch + n: workgroup i finished exchange point n of this launch
  unsigned flags2[LSTM_MAX_WG];   // second exchange point of a BPTT step
};
__device__ LstmSync g_lstm_sync[LSTM_SYNC_SLOTS];

constexpr long long LSTM_SPIN_LIMIT = 400000000ll;   // wall_clock64 ticks (100 MHz): 4 s without progress -> give up (no trap)

// A wait that sees no progress for LSTM_SPIN_LIMIT gives up WITHOUT trapping (round 5; a trap kills the whole context, and under
// eight ranks one slow rank would become seven hung RCCL collectives): it raises g_lstm_abort, which every other waiter of the
// device notices on its next slow-path check, counts the event in a host-mapped word the HOST reads without synchronising
// (fac_lstm_persist_timeouts; fac_lstm_persist_ok / _split_ok answer 0 from then on, so callers fall back to the per-step
// kernels of lstm.hip), and the launch runs to its end with meaningless results.
__device__ unsigned g_lstm_abort;
__device__ unsigned* g_lstm_host_timeouts;

// wave 0: wait until flags[first .. first+count) have all reached `target`; then the workgroup passes a barrier (mode 0
// only: and every wave takes an agent-scope acquire that drops stale L1 / L2 lines of the exchanged buffers).
// Exchange modes.  0: cache maintenance (buffer_wbl2 / buffer_inv) around plain accesses.  1: every exchanged word moves with
// agent-scope (sc1) stores and loads, ordered by s_waitcnt only -- no L2 write-back / invalidate per step, but every
// workgroup's copy of h_t comes from the memory side.  2: sc1 stores into a FRESH region per step, each workgroup writing
// whole cache lines of its own; the consumers read with plain loads: no XCD can hold an older copy of a line that did
// not exist before, so the first reader of an XCD misses to memory and the other 23 workgroups of that XCD hit its L2.
__device__ __forceinline__ void wait_flags(const unsigned* flags, int first, int count, unsigned target, int mode) {
  if ((threadIdx.x >> 6) == 0) {
    const int lane = threadIdx.x & 63;
    const long long t0 = wall_clock64();
    for (unsigned it = 1;; ++it) {
      bool behind;
      if (count <= 64) {
        const int i = lane < count ? lane : count - 1;
        const unsigned f = __hip_atomic_load(flags + first + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        behind = (int)(f - target) < 0;
      } else {   // all LSTM_MAX_WG flags in ONE wave-wide agent-scope load (first == 0; the array is 16-byte aligned)
        u32x4 f;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(flags + 4 * lane) : "memory");
        behind = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) behind |= (4 * lane + c < count) && (int)(f[c] - target) < 0;
      }
      if (__ballot(behind) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if ((it & 1023u) == 0) {
        if (__hip_atomic_load(&g_lstm_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (wall_clock64() - t0 > LSTM_SPIN_LIMIT) {
          if (lane == 0) {
            __hip_atomic_store(&g_lstm_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned* h = g_lstm_host_timeouts;
            if (h) __hip_atomic_fetch_add(h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          break;
        }
      }
    }
  }
  __syncthreads();
  if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ f32x4 load_agent_x4(const f32x4* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the stores of this workgroup (issued by threads tid < n_store, a multiple of 64) become visible device-wide, then flag
__device__ __forceinline__ void publish_flag(unsigned* flag, unsigned value, int n_store, int mode) {
  if (mode == 0) {
    if ((int)threadIdx.x < n_store) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through (sc1) stores of this wave are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ unsigned launch_epoch(LstmSync* sync, unsigned* s_base) {
  if (threadIdx.x == 0) *s_base = __hip_atomic_load(&sync->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return *s_base;
}

// last workgroup out advances the epoch past every flag value of this launch
__device__ __forceinline__ void leave_launch(LstmSync* sync, unsigned base, unsigned advance, unsigned nwg) {
  if (threadIdx.x == 0) {
    const unsigned d = __hip_atomic_fetch_add(&sync->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (d == nwg - 1) {
      __hip_atomic_store(&sync->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync->epoch, base + advance, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Contraction index <-> hidden unit.  Within every 16 units, unit 8p + e sits at k = 4*(e>>1) + 2p + (e&1): the 8 units a
// workgroup produces per step then fill, for all 16 columns, the float4s of 32 consecutive lanes of one fragment block --
// 512 contiguous bytes = four whole 128-byte lines that no other workgroup writes (mode 2 relies on this).
__host__ __device__ __forceinline__ int k_of_unit(int unit) {
  const int u16 = unit & 15, p = u16 >> 3, e = u16 & 7;
  return (unit & ~15) + ((e >> 1) << 2) + 2 * p + (e & 1);
}
__host__ __device__ __forceinline__ int unit_of_k(int k) {
  const int r = k & 15, jj = r >> 2, kq = r & 3;
  return (k & ~15) + 8 * (kq >> 1) + 2 * jj + (kq & 1);
}

// position of X[unit][col] in the fragment-ordered exchange buffer (see the header of this file)
__device__ __forceinline__ long long frag_index(int unit, int col, int H, int KS) {
  const int k = k_of_unit(unit);
  const int kw = H >> 4;                 // k per wave
  const int w = k / kw, rem = k - w * kw;
  const int s = rem >> 2, kq = rem & 3;
  const int j4 = s >> 2, jj = s & 3;
  const int cb = col >> 4, c16 = col & 15;
  return ((((long long)(cb * 16 + w) * (KS >> 2) + j4) * 64 + kq * 16 + c16) << 2) + jj;
}

// One wave's share of  D(32 x NCB*16) += A(32 x H/16) * X(H/16 x NCB*16)  from its register-resident weights; the 16
// partial results of the workgroup meet in `red`.
template <int KS, int NCB>
__device__ __forceinline__ void wave_product(const float4 (&a4)[KS / 2], const float* frag, int wave, int lane,
                                             float (*red)[32][NCB * 16 + 1], int mode) {
  f32x4 acc[2][NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[0][cb][r] = 0.f;
      acc[1][cb][r] = 0.f;
    }
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const f32x4* bp = reinterpret_cast<const f32x4*>(frag) + ((long long)(cb * 16 + wave) * (KS / 4)) * 64 + lane;
    f32x4 b4[KS / 4];
    if (mode != 1) {
#pragma unroll
      for (int j = 0; j < KS / 4; ++j) b4[j] = bp[(long long)j * 64];
    } else {
#pragma unroll
      for (int j = 0; j < KS / 4; ++j) b4[j] = load_agent_x4(bp + (long long)j * 64);
#pragma unroll
      for (int j = 0; j < KS / 4; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b4[j])::"memory");
    }
#pragma unroll
    for (int j = 0; j < KS / 4; ++j) {
      const float bv[4] = {b4[j][0], b4[j][1], b4[j][2], b4[j][3]};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int s = 4 * j + jj;
        const float4 a = a4[s >> 1];
        const float a0 = (s & 1) ? a.z : a.x;
        const float a1 = (s & 1) ? a.w : a.y;
        acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[jj], acc[0][cb], 0, 0, 0);
        acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[jj], acc[1][cb], 0, 0, 0);
      }
    }
  }
  const int c16 = lane & 15, r0 = 4 * (lane >> 4);
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][rb * 16 + r0 + r][cb * 16 + c16] = acc[rb][cb][r];
}

// ---------------------------------------------------------------------------------------------------------- forward
// grid = H/8 workgroups of 1024 threads; workgroup ub owns hidden units ub*8 .. +8 (32 gate rows q*8+u).
template <int KS, int NCB>
__global__ __launch_bounds__(1024) void lstm_fwd_persist_kernel(const float* __restrict__ pre,     // (4H, T, BP)
                                                                const float* __restrict__ whh16,   // fac_pack_lstm_whh16(.., 0)
                                                                float* hfrag,                      // T x H*NC, fragment order
                                                                float* __restrict__ yT,            // (H, T, BP)
                                                                float* __restrict__ save_g,        // (4H, T, BP) or null
                                                                float* __restrict__ save_c,        // (H, T, BP) or null
                                                                int slot, int T, int H, int BP, int mode) {
  constexpr int NC = NCB * 16;
  __shared__ float red[16][32][NC + 1];
  __shared__ unsigned s_base;
  LstmSync* sync = &g_lstm_sync[slot];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ub = blockIdx.x, nwg = gridDim.x;
  const long long rs = (long long)T * BP;

  float4 a4[KS / 2];
  {
    const float4* ap = reinterpret_cast<const float4*>(whh16) + ((long long)(ub * 16 + wave) * (KS / 2)) * 64 + lane;
#pragma unroll
    for (int j = 0; j < KS / 2; ++j) a4[j] = ap[(long long)j * 64];
  }
  const unsigned base = launch_epoch(sync, &s_base);

  const bool gate_thread = tid < 8 * NC;
  const int u = tid / NC, col = tid - u * NC;
  const int unit = ub * 8 + u;
  float pre_v[4] = {0.f, 0.f, 0.f, 0.f};
  float c_reg = 0.f;
  if (gate_thread) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pre_v[q] = pre[(long long)(q * H + unit) * rs + col];
  }
  const long long hpos = gate_thread ? frag_index(unit, col, H, KS) : 0;
  const long long hbuf = (long long)H * NC;

  for (int t = 0; t < T; ++t) {
    float gate[4] = {pre_v[0], pre_v[1], pre_v[2], pre_v[3]};
    if (t > 0) {
      wait_flags(sync->flags, 0, nwg, base + (unsigned)t, mode);  // every workgroup has published h_{t-1}
      wave_product<KS, NCB>(a4, hfrag + (mode == 2 ? t - 1 : (t - 1) & 1) * hbuf, wave, lane, red, mode);
      __syncthreads();
      if (gate_thread) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < 16; ++w) s += red[w][q * 8 + u][col];
          gate[q] += s;
        }
      }
    }
    if (gate_thread) {
      const float ig = sigmoid_f(gate[0]);
      const float fg = sigmoid_f(gate[1]);
      const float gg = tanhf(gate[2]);
      const float og = sigmoid_f(gate[3]);
      const float c_new = __fadd_rn(__fmul_rn(fg, c_reg), __fmul_rn(ig, gg));
      c_reg = c_new;
      const float hv = __fmul_rn(og, tanhf(c_new));
      if (t + 1 < T) store_agent(hfrag + (mode == 2 ? t : t & 1) * hbuf + hpos, hv);   // first: its write-through is the critical path
      const long long o = (long long)unit * rs + (long long)t * BP + col;
      yT[o] = hv;
      if (save_g != nullptr) {
        save_g[o] = ig;
        save_g[(long long)H * rs + o] = fg;
        save_g[2ll * H * rs + o] = gg;
        save_g[3ll * H * rs + o] = og;
        save_c[o] = c_new;
      }
    }
    if (t + 1 < T) {
      publish_flag(sync->flags + ub, base + (unsigned)t + 1u, 8 * NC, mode);
      if (gate_thread) {      // next step's gate pre-activations arrive while the flags are polled
#pragma unroll
        for (int q = 0; q < 4; ++q) pre_v[q] = pre[(long long)(q * H + unit) * rs + (long long)(t + 1) * BP + col];
      }
    }
  }
  leave_launch(sync, base, (unsigned)T, (unsigned)nwg);
}

// ------------------------------------------------------------------------------------------ forward, bf16 x 3 operands
// The same resident recurrence with W_hh . h on the bf16 matrix pipe, fp32-grade in the sense of conv1d_bsplit.hip: every
// weight and every state value is the sum of three round-to-nearest bf16 terms, six of the nine cross products are kept,
// accumulation in fp32.  At 32 batch columns the fp32 kernel above is bound by its v_mfma_f32_16x16x4_f32 time (32 x 32 x H
// MACs per workgroup and step at 128 MAC / clk: 5.1 us at H = 1536); six v_mfma_f32_32x32x16_bf16 per 16 k do the same
// product in 2.2 us.
//   grid      H/8 workgroups of 512 threads (2 waves per SIMD: up to 256 VGPRs each); workgroup ub owns units ub*8 .. +8
//   weights   wave w contracts k in [w*H/8, (w+1)*H/8) in NK = H/128 MFMA steps; its A fragments (rows = gate*8 + unit,
//             3 planes) stay in 12*NK VGPRs for the whole sequence (144 at H = 1536): fac_pack_lstm_whh_split
//   state     h_t crosses the device ALREADY split, in B-fragment order: hs[t][plane][k/16][lane] 16-byte pieces, lane =
//             32*((k%16)/8) + column, piece = 8 consecutive k.  The 8 units of a workgroup are exactly one piece per column:
//             per plane it writes 32 lanes x 16 B = 512 contiguous bytes = four whole 128-byte lines of a FRESH region per
//             step that nobody else writes (exchange mode 2 above; the only mode of this kernel), by write-through stores.
//             Consumers read their 3*NK pieces per lane with plain 16-byte loads through a 4-step register window.
typedef __bf16 ls_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ls_f32x16 __attribute__((ext_vector_type(16)));
constexpr int LS_NW = 8;

__device__ __forceinline__ void ls_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// 16-byte piece at (uniform base, lane offset): request only; the value is valid behind ls_wait_pieces
__device__ __forceinline__ ls_bf16x8 ls_load_piece(const ls_bf16x8* base, unsigned lane_off) {
  ls_bf16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(lane_off), "s"(base) : "memory");
  return v;
}
__device__ __forceinline__ void ls_wait_pieces(ls_bf16x8 (&d)[3], int behind) {
  switch (behind) {   // constant after unrolling
    case 0: asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])::"memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])::"memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])::"memory"); break;
    default: asm volatile("s_waitcnt vmcnt(9)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])::"memory"); break;
  }
}

// W_hh (4H, H) -> packed[((ub*8 + w)*NK + s)*3 + plane][lane]: A fragment of step s of wave w (lane: row l%32 = gate*8 + unit,
// k = (w*NK + s)*16 + 8*(l/32) .. +8)
__global__ __launch_bounds__(256) void pack_whh_split_kernel(const float* __restrict__ w, ls_bf16x8* __restrict__ out, int H) {
  const int NK = H / (16 * LS_NW);
  const long long n = (long long)(H / 8) * LS_NW * NK * 64;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int s = (int)(r % NK);
    r /= NK;
    const int wv = (int)(r % LS_NW);
    const int ub = (int)(r / LS_NW);
    const int l31 = lane & 31, kq = lane >> 5;
    const int row = (l31 >> 3) * H + ub * 8 + (l31 & 7);
    const int k0 = (wv * NK + s) * 16 + 8 * kq;
    ls_bf16x8 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __bf16 a, b, c;
      ls_split3(w[(long long)row * H + k0 + i], a, b, c);
      ph[i] = a; pm[i] = b; pl[i] = c;
    }
    const long long o = (r * NK + s) * 3;      // r = ub*8 + wv
    out[(o + 0) * 64 + lane] = ph;
    out[(o + 1) * 64 + lane] = pm;
    out[(o + 2) * 64 + lane] = pl;
  }
}

template <int NK>
__global__ __launch_bounds__(LS_NW * 64) void lstm_fwd_persist_split_kernel(const float* __restrict__ pre,          // (4H, T, BP)
                                                                           const ls_bf16x8* __restrict__ wsplit,   // pack_whh_split_kernel
                                                                           ls_bf16x8* hs,                          // T x 3 x H/16 x 64 pieces
                                                                           float* __restrict__ yT,                 // (H, T, BP)
                                                                           int slot, int T, int H, int BP) {
  constexpr int WIN = NK < 4 ? NK : 4;
  __shared__ float red[LS_NW][32][33];
  __shared__ __attribute__((aligned(16))) __bf16 piece[3][32][8];
  __shared__ unsigned s_base;
  LstmSync* sync = &g_lstm_sync[slot];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kq = lane >> 5;
  const unsigned lane16 = (unsigned)lane << 4;
  const int ub = blockIdx.x, nwg = gridDim.x;
  const long long rs = (long long)T * BP;
  const int nS = H >> 4;                         // 16-k steps of the whole contraction
  const long long hregion = 3ll * nS * 64;       // pieces per time step

  ls_bf16x8 W[NK][3];
  {
    const ls_bf16x8* wp = wsplit + ((long long)(ub * LS_NW + wave) * NK * 3) * 64 + lane;
#pragma unroll
    for (int s = 0; s < NK; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p) W[s][p] = wp[(long long)(s * 3 + p) * 64];
  }
  const unsigned base = launch_epoch(sync, &s_base);

  const bool gate_thread = tid < 256;            // (unit u, column col) of the 8 x 32 outputs of a step
  const int u = tid >> 5, col = tid & 31;
  const int unit = ub * 8 + u;
  float pre_v[4] = {0.f, 0.f, 0.f, 0.f};
  float c_reg = 0.f;
  if (gate_thread) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pre_v[q] = pre[(long long)(q * H + unit) * rs + col];
  }
  // the three pieces of this workgroup per column: threads 0 .. 95 = (plane, column)
  const int sp = tid >> 5, sc = tid & 31;
  const long long spos = ((long long)sp * nS + (ub >> 1)) * 64 + 32 * (ub & 1) + sc;

  for (int t = 0; t < T; ++t) {
    float gate[4] = {pre_v[0], pre_v[1], pre_v[2], pre_v[3]};
    if (t > 0) {
      wait_flags(sync->flags, 0, nwg, base + (unsigned)t, 2);   // every workgroup has published h_{t-1}
      {
        // The pieces of step s are requested WIN steps ahead.  hipcc sinks plain loads down to their first use (one memory round
        // trip per step), so the loads and their waits are explicit: in-order return, s_waitcnt vmcnt(n) with n = the loads issued
        // behind the ones needed (tools/check_inflight_regs.py replays the queue on the ISA: tests/test_isa_inflight.py).
        const ls_bf16x8* hp;                                                                       // uniform: scalar base
        {
          const unsigned long long a = reinterpret_cast<unsigned long long>(hs + (long long)(t - 1) * hregion + (long long)(wave * NK) * 64);
          const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
          hp = reinterpret_cast<const ls_bf16x8*>(((unsigned long long)hi << 32) | lo);
        }
        ls_bf16x8 hb[WIN][3];
        ls_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < WIN; ++s)
#pragma unroll
          for (int p = 0; p < 3; ++p) hb[s][p] = ls_load_piece(hp + ((long long)p * nS + s) * 64, lane16);
        // smallest terms first, as in conv1d_bsplit.hip: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi
        constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < NK; ++s) {
          constexpr int AHEAD = WIN - 1;
          ls_wait_pieces(hb[s % WIN], 3 * (NK - 1 - s < AHEAD ? NK - 1 - s : AHEAD));
#pragma unroll
          for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][TA[q]], hb[s % WIN][TB[q]], acc, 0, 0, 0);
          if (s + WIN < NK) {
#pragma unroll
            for (int p = 0; p < 3; ++p) hb[s % WIN][p] = ls_load_piece(hp + ((long long)p * nS + s + WIN) * 64, lane16);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kq][l31] = acc[r];
      }
      __syncthreads();
      if (gate_thread) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < LS_NW; ++w) sum += red[w][q * 8 + u][col];
          gate[q] += sum;
        }
      }
    }
    float hv = 0.f;
    if (gate_thread) {
      const float ig = sigmoid_f(gate[0]);
      const float fg = sigmoid_f(gate[1]);
      const float gg = tanhf(gate[2]);
      const float og = sigmoid_f(gate[3]);
      const float c_new = __fadd_rn(__fmul_rn(fg, c_reg), __fmul_rn(ig, gg));
      c_reg = c_new;
      hv = __fmul_rn(og, tanhf(c_new));
      if (t + 1 < T) {
        __bf16 a, b, c;
        ls_split3(hv, a, b, c);
        piece[0][col][u] = a;
        piece[1][col][u] = b;
        piece[2][col][u] = c;
      }
    }
    if (t + 1 < T) {
      __syncthreads();
      if (tid < 96) {    // first: the write-through of the pieces is the critical path
        const ls_bf16x8 v = *reinterpret_cast<const ls_bf16x8*>(&piece[sp][sc][0]);
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(hs + (long long)t * hregion + spos), "v"(v) : "memory");
      }
    }
    if (gate_thread) yT[(long long)unit * rs + (long long)t * BP + col] = hv;
    if (t + 1 < T) {
      publish_flag(sync->flags + ub, base + (unsigned)t + 1u, 128, 2);
      if (gate_thread) {      // next step's gate pre-activations arrive while the flags are polled
#pragma unroll
        for (int q = 0; q < 4; ++q) pre_v[q] = pre[(long long)(q * H + unit) * rs + (long long)(t + 1) * BP + col];
      }
    }
  }
  leave_launch(sync, base, (unsigned)T, (unsigned)nwg);
}

// --------------------------------------------------------------------------------------------------------- backward
// grid = H/8 workgroups: workgroup (ub, q) = blockIdx.x / 4, blockIdx.x % 4 computes, for the 32 hidden units
// ub*32 .. +32, the quarter-q part of  W_hh^T dgates_{t+1}  (contraction over the H gate rows q*H .. q*H+H) and then
// the gate derivatives of the 8 units ub*32 + q*8 .. +8.  Two exchange points per step: the four quarter sums of a
// unit block meet (4 flags), then dgates_t is published to everyone.
template <int KS, int NCB>
__global__ __launch_bounds__(1024) void lstm_bwd_persist_kernel(const float* __restrict__ dyT,     // (H, T, BP)
                                                                const float* __restrict__ whh16t,  // fac_pack_lstm_whh16(.., 1)
                                                                const float* __restrict__ gates,   // (4H, T, BP) saved
                                                                const float* __restrict__ cs,      // (H, T, BP) saved
                                                                float* __restrict__ dgates,        // (4H, T, BP) out
                                                                float* partial,                    // (4, H, NC)
                                                                float* dgfrag,                     // T x 4 x H*NC, fragment order
                                                                int slot, int T, int H, int BP, int mode) {
  constexpr int NC = NCB * 16;
  __shared__ float red[16][32][NC + 1];
  __shared__ unsigned s_base;
  LstmSync* sync = &g_lstm_sync[slot];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x, nwg = gridDim.x;
  const int ub = wg >> 2, q = wg & 3;
  const long long rs = (long long)T * BP;
  const long long hbuf = (long long)H * NC;

  float4 a4[KS / 2];
  {
    const float4* ap = reinterpret_cast<const float4*>(whh16t) + ((long long)(wg * 16 + wave) * (KS / 2)) * 64 + lane;
#pragma unroll
    for (int j = 0; j < KS / 2; ++j) a4[j] = ap[(long long)j * 64];
  }
  const unsigned base = launch_epoch(sync, &s_base);

  const bool gate_thread = tid < 8 * NC;
  const int u = tid / NC, col = tid - u * NC;
  const int unit = ub * 32 + q * 8 + u;
  const long long hpos = gate_thread ? frag_index(unit, col, H, KS) : 0;
  float dc_reg = 0.f;

  for (int n = 0; n < T; ++n) {
    const int t = T - 1 - n;
    // the saved activations do not depend on the exchange: fetch them first
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, ct = 0.f, cp = 0.f, dh = 0.f;
    const long long o = (long long)unit * rs + (long long)t * BP + col;
    if (gate_thread) {
      ig = gates[o];
      fg = gates[(long long)H * rs + o];
      gg = gates[2ll * H * rs + o];
      og = gates[3ll * H * rs + o];
      ct = cs[o];
      cp = t > 0 ? cs[o - BP] : 0.f;
      dh = dyT[o];
    }
    if (n > 0) {
      wait_flags(sync->flags2, 0, nwg, base + (unsigned)n, mode);   // dgates_{t+1} of every unit is published
      wave_product<KS, NCB>(a4, dgfrag + (mode == 2 ? n - 1 : (n - 1) & 1) * 4 * hbuf + q * hbuf, wave, lane, red, mode);
      __syncthreads();
      for (int e = tid; e < 32 * NC; e += 1024) {
        const int row = e / NC, c = e - row * NC;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w][row][c];
        store_agent(partial + (long long)q * hbuf + (long long)(ub * 32 + row) * NC + c, s);
      }
      publish_flag(sync->flags + wg, base + (unsigned)n, 1024, mode);
      wait_flags(sync->flags, ub * 4, 4, base + (unsigned)n, mode);   // the four quarter sums of this unit block
      if (gate_thread) {
        const float* pp = partial + (long long)unit * NC + col;
        dh += ((load_agent(pp) + load_agent(pp + hbuf)) + (load_agent(pp + 2 * hbuf) + load_agent(pp + 3 * hbuf)));
      }
    }
    if (gate_thread) {
      const float tc = tanhf(ct);
      const float d_o = dh * tc;
      const float dcv = dh * og * (1.f - tc * tc) + dc_reg;
      dc_reg = dcv * fg;
      float dg[4];
      dg[0] = dcv * gg * ig * (1.f - ig);
      dg[1] = dcv * cp * fg * (1.f - fg);
      dg[2] = dcv * ig * (1.f - gg * gg);
      dg[3] = d_o * og * (1.f - og);
      if (n + 1 < T) {      // first: their write-through is the critical path
#pragma unroll
        for (int g = 0; g < 4; ++g) store_agent(dgfrag + (mode == 2 ? n : n & 1) * 4 * hbuf + g * hbuf + hpos, dg[g]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) dgates[(long long)g * H * rs + o] = dg[g];
    }
    if (n + 1 < T) publish_flag(sync->flags2 + wg, base + (unsigned)n + 1u, 8 * NC, mode);
  }
  leave_launch(sync, base, (unsigned)T, (unsigned)nwg);
}

// W_hh (4H, H) -> register fragments of the persistent kernels.  transposed = 0 (forward): block = 8 hidden units,
// row r32 = gate*8 + u of W_hh, k = hidden index.  transposed = 1 (BPTT): block = (32 hidden units, gate quarter q),
// row r32 = unit within the block, k = the H gate rows of quarter q:  A = W_hh[q*H + k][unit].
__global__ __launch_bounds__(256) void pack_whh16_kernel(const float* __restrict__ w, float* __restrict__ out, int H, int transposed) {
  const long long n = (long long)4 * H * H;
  const int KS = H >> 6, kw = H >> 4;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256) {
    const int comp = (int)(o & 3);
    const int lane = (int)((o >> 2) & 63);
    long long rest = o >> 8;
    const int j = (int)(rest % (KS >> 1));
    rest /= (KS >> 1);
    const int wv = (int)(rest & 15);
    const int blk = (int)(rest >> 4);
    const int s = 2 * j + (comp >> 1), rb = comp & 1;
    const int r32 = rb * 16 + (lane & 15);
    const int k = wv * kw + 4 * s + (lane >> 4);
    float v;
    if (!transposed) {
      const int gate = r32 >> 3, uu = r32 & 7;
      v = w[(long long)(gate * H + blk * 8 + uu) * H + unit_of_k(k)];
    } else {
      const int ubk = blk >> 2, qq = blk & 3;
      v = w[(long long)(qq * H + unit_of_k(k)) * H + ubk * 32 + r32];
    }
    out[o] = v;
  }
}

// Exchange-flag slots are handed out per (device, stream) for the life of the process; a process that has used them all
// gets -1 and the caller runs the per-step kernels (lstm.hip) on that stream instead (fac_lstm_persist_stream_ok).
static int lstm_sync_slot(hipStream_t stream) {
  static std::mutex mu;
  static std::unordered_map<unsigned long long, int> slots;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long key = (reinterpret_cast<unsigned long long>(stream) << 4) ^ (unsigned long long)dev;
  std::lock_guard<std::mutex> lock(mu);
  auto it = slots.find(key);
  if (it != slots.end()) return it->second;
  if ((int)slots.size() >= LSTM_SYNC_SLOTS) return -1;
  const int s = (int)slots.size();
  slots.emplace(key, s);
  return s;
}

constexpr int LSTM_MAX_DEV = 16;

static int device_cus() {
  static int cus[LSTM_MAX_DEV] = {0};
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LSTM_MAX_DEV) return 0;
  if (cus[dev] == 0 && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus[dev] = v;
  return cus[dev];
}

// The resident kernels spin on flags written by the OTHER workgroups of their grid, so the whole grid must be resident at once
// (ADVICE r3).  Two guards, both on the host:
//  * the runtime's own occupancy figure for the kernel x the device's CUs must cover the grid (a workgroup is 1024 threads + ~35 KB
//    of LDS, i.e. one per CU; a build that needs more registers would silently stop fitting);
//  * resident launches of one process are SERIALISED on the device, whatever streams or threads they come from: every launch
//    waits for the event recorded behind the previous one (two half-resident grids would wait for each other's CUs for ever).
//    Launches issued during stream capture skip the event (an un-captured event cannot be waited for inside a capture): a graph's
//    own launches are ordered by the graph, and replaying such a graph next to other resident launches is the caller's to order.
//  Several PROCESSES sharing one device cannot be ordered from here: facodec_amd.ops.lstm_persist_ok refuses the resident path
//  when ranks share a device.
static bool grid_fits(const void* kern, int grid, int threads = 1024) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0) != hipSuccess) return false;
  return (long long)per_cu * device_cus() >= grid;
}

struct PersistOrder {
  std::mutex mu;
  hipEvent_t ev[LSTM_MAX_DEV] = {};
  bool have[LSTM_MAX_DEV] = {};
};
static PersistOrder g_order;

// Holds the process-wide mutex ACROSS "wait for the previous resident launch", the launch itself and "record behind it"
// (as three critical sections two host threads -- ctypes releases the GIL, autograd runs backward on its own thread --
// could both wait on the same old event and then launch unordered: the two-half-resident-grids deadlock).
struct ResidentLaunch {
  std::unique_lock<std::mutex> lock;
  hipStream_t stream;
  int dev = -1;
  explicit ResidentLaunch(hipStream_t st) : lock(g_order.mu), stream(st) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= LSTM_MAX_DEV) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
    dev = d;
    if (g_order.have[dev]) (void)hipStreamWaitEvent(stream, g_order.ev[dev], 0);
  }
  ~ResidentLaunch() {
    if (dev < 0) return;
    if (!g_order.ev[dev] && hipEventCreateWithFlags(&g_order.ev[dev], hipEventDisableTiming) != hipSuccess) return;
    g_order.have[dev] = hipEventRecord(g_order.ev[dev], stream) == hipSuccess;
  }
};

// Host-mapped counter of waits that gave up (g_lstm_host_timeouts on every device of this process points at it).
struct TimeoutWord {
  std::mutex mu;
  unsigned* host = nullptr;
  bool armed[LSTM_MAX_DEV] = {};
};
static TimeoutWord g_timeouts;

static unsigned persist_timeouts() {
  std::lock_guard<std::mutex> lock(g_timeouts.mu);
  return g_timeouts.host ? *reinterpret_cast<volatile unsigned*>(g_timeouts.host) : 0u;
}

__global__ void lstm_arm_timeout_word_kernel(unsigned* p) { g_lstm_host_timeouts = p; }

// Called in front of every resident launch (under ResidentLaunch's lock): the first launch on a device tells the device where to
// count its timeouts -- by a one-thread kernel on the same stream (no synchronising call, nothing allocated inside a stream
// capture: a launch issued during capture before the word exists simply runs un-armed, the abort word still ends its waits).
static void arm_timeout_word(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LSTM_MAX_DEV) return;
  std::lock_guard<std::mutex> lock(g_timeouts.mu);
  if (g_timeouts.armed[dev]) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
  if (!g_timeouts.host) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return;
    g_timeouts.host = static_cast<unsigned*>(p);
    *g_timeouts.host = 0u;
  }
  void* dptr = nullptr;
  if (hipHostGetDevicePointer(&dptr, g_timeouts.host, 0) != hipSuccess) return;
  hipLaunchKernelGGL(lstm_arm_timeout_word_kernel, dim3(1), dim3(1), 0, stream, static_cast<unsigned*>(dptr));
  g_timeouts.armed[dev] = hipGetLastError() == hipSuccess;
}

static bool timeout_word_armed() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LSTM_MAX_DEV) return false;
  std::lock_guard<std::mutex> lock(g_timeouts.mu);
  return g_timeouts.armed[dev];
}

// dst[0] = 1 if a resident wait of this device has given up since the process started, else 0 (device-side read of the abort
// word, ordered on `stream` behind the launches whose outcome it reports; no host synchronisation).
__global__ void lstm_abort_flag_kernel(float* dst) { dst[0] = __hip_atomic_load(&g_lstm_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1.f : 0.f; }
// flags[0 .. n) = 0 when *poison > 0 (after the flag exchange: ANY rank's abort) -- the masked AdamW step then steps nothing.
__global__ void mask_flags_if_kernel(float* flags, int n, const float* poison) {
  if (*poison > 0.f)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) flags[i] = 0.f;
}

// FAC_LSTM_EXCHANGE = fence (mode 0) | sc1 (mode 1) | fresh (mode 2, default)
static int exchange_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("FAC_LSTM_EXCHANGE");
    mode = e == nullptr ? 2 : (e[0] == 'f' && e[1] == 'e') ? 0 : e[0] == 's' ? 1 : 2;
  }
  return mode;
}

static bool persist_shape_ok(int H, int B) {
  if (H <= 0 || H % 256 != 0 || B <= 0 || B > 32) return false;
  const int ks = H / 64;
  if (ks != 8 && ks != 16 && ks != 24) return false;
  const int wgs = H / 8;
  return wgs <= LSTM_MAX_WG && wgs <= device_cus();
}

static bool persist_split_shape_ok(int H, int B) {
  if (B <= 0 || B > 32 || (H != 512 && H != 1024 && H != 1536)) return false;
  const int wgs = H / 8;
  return wgs <= LSTM_MAX_WG && wgs <= device_cus();
}

using FwdKern = void (*)(const float*, const float*, float*, float*, float*, float*, int, int, int, int, int);
using BwdKern = void (*)(const float*, const float*, const float*, const float*, float*, float*, float*, int, int, int, int, int);
using SplitKern = void (*)(const float*, const ls_bf16x8*, ls_bf16x8*, float*, int, int, int, int);

static FwdKern fwd_kernel_for(int H, int B) {
  switch ((H / 64) * 10 + (B + 15) / 16) {
    case 81: return lstm_fwd_persist_kernel<8, 1>;
    case 82: return lstm_fwd_persist_kernel<8, 2>;
    case 161: return lstm_fwd_persist_kernel<16, 1>;
    case 162: return lstm_fwd_persist_kernel<16, 2>;
    case 241: return lstm_fwd_persist_kernel<24, 1>;
    case 242: return lstm_fwd_persist_kernel<24, 2>;
  }
  return nullptr;
}

static BwdKern bwd_kernel_for(int H, int B) {
  switch ((H / 64) * 10 + (B + 15) / 16) {
    case 81: return lstm_bwd_persist_kernel<8, 1>;
    case 82: return lstm_bwd_persist_kernel<8, 2>;
    case 161: return lstm_bwd_persist_kernel<16, 1>;
    case 162: return lstm_bwd_persist_kernel<16, 2>;
    case 241: return lstm_bwd_persist_kernel<24, 1>;
    case 242: return lstm_bwd_persist_kernel<24, 2>;
  }
  return nullptr;
}

static SplitKern split_kernel_for(int H) {
  switch (H) {
    case 512: return lstm_fwd_persist_split_kernel<4>;
    case 1024: return lstm_fwd_persist_split_kernel<8>;
    case 1536: return lstm_fwd_persist_split_kernel<12>;
  }
  return nullptr;
}

// the occupancy answer per (device, kernel) is asked once
static bool grid_fits_cached(const void* kern, int grid, int threads) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, bool> cache;
  if (kern == nullptr) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(dev, kern);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const bool ok = grid_fits(kern, grid, threads);
  cache[key] = ok;
  return ok;
}

// (a launch whose waits can bail out must be able to SAY so -- an un-armed counter word, e.g. because the first resident
// launch of the process was issued inside a stream capture or hipHostMalloc failed, means "not usable": fac_lstm_persist_arm.)
// What fac_lstm_persist_ok / _split_ok answer: the shape is inside the kernels, the whole grid of the forward AND the backward
// kernel is co-resident by the runtime's own occupancy figure (a shape or device that fails it used to be a hard
// FAC_REQUIRE error in the launch instead of a fall-back), and no resident wait of this process has ever timed out.
static bool persist_usable(int H, int B) {
  return persist_shape_ok(H, B) && persist_timeouts() == 0u && timeout_word_armed() &&
         grid_fits_cached(reinterpret_cast<const void*>(fwd_kernel_for(H, B)), H / 8, 1024) &&
         grid_fits_cached(reinterpret_cast<const void*>(bwd_kernel_for(H, B)), H / 8, 1024);
}

static bool persist_split_usable(int H, int B) {
  return persist_split_shape_ok(H, B) && persist_timeouts() == 0u && timeout_word_armed() &&
         grid_fits_cached(reinterpret_cast<const void*>(split_kernel_for(H)), H / 8, LS_NW * 64);
}

}  // namespace fac

extern "C" int fac_lstm_persist_split_ok(int H, int B) { return fac::persist_split_usable(H, B) ? 1 : 0; }

extern "C" int fac_lstm_persist_timeouts(void) { return (int)fac::persist_timeouts(); }

extern "C" int fac_lstm_persist_arm(fac_stream_t stream) {
  fac::arm_timeout_word((hipStream_t)stream);
  return fac::timeout_word_armed() ? 1 : 0;
}

extern "C" int fac_lstm_abort_flag(float* dst, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(dst != nullptr, "lstm_abort_flag: null pointer");
  hipLaunchKernelGGL(lstm_abort_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
  return check_launch("lstm_abort_flag");
}

extern "C" int fac_mask_flags_if(float* flags, int n, const float* poison, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(flags && poison && n > 0, "mask_flags_if: bad arguments");
  hipLaunchKernelGGL(mask_flags_if_kernel, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0, (hipStream_t)stream, flags, n, poison);
  return check_launch("mask_flags_if");
}

extern "C" int fac_pack_lstm_whh_split(const float* w_hh, void* packed, int H, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(w_hh && packed && H > 0 && H % 128 == 0, "pack_lstm_whh_split: H must be a multiple of 128");
  hipLaunchKernelGGL(pack_whh_split_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, w_hh, reinterpret_cast<ls_bf16x8*>(packed), H);
  return check_launch("pack_lstm_whh_split");
}

extern "C" int fac_lstm_layer_fwd_persist_split(const float* pre, const void* wsplit, void* hsplit, float* yT, int T, int H, int B, int BP,
                                                fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(pre && wsplit && hsplit && yT, "lstm_layer_fwd_persist_split: null pointer");
  FAC_REQUIRE(T > 0 && BP >= 32 && BP >= B && BP % 32 == 0, "lstm_layer_fwd_persist_split: bad T / BP");
  FAC_REQUIRE(persist_split_shape_ok(H, B), "lstm_layer_fwd_persist_split: H=%d B=%d is outside the kernel (fac_lstm_persist_split_ok)", H, B);
  const int slot = lstm_sync_slot((hipStream_t)stream);
  FAC_REQUIRE(slot >= 0, "lstm_layer_fwd_persist_split: more than %d streams in use", LSTM_SYNC_SLOTS);
  SplitKern kern = split_kernel_for(H);
  FAC_REQUIRE(kern != nullptr, "lstm_layer_fwd_persist_split: no kernel for H=%d", H);
  arm_timeout_word((hipStream_t)stream);        // (no-op once armed; an un-armed word makes the launch unusable)
  FAC_REQUIRE(persist_split_usable(H, B), "lstm_layer_fwd_persist_split: %d workgroups are not co-resident on this device, or an earlier "
              "resident launch timed out (%u): ask fac_lstm_persist_split_ok first and fall back to fac_lstm_layer_fwd", H / 8, persist_timeouts());
  int rc;
  {
    ResidentLaunch order((hipStream_t)stream);
    arm_timeout_word((hipStream_t)stream);
    hipLaunchKernelGGL(kern, dim3(H / 8), dim3(LS_NW * 64), 0, (hipStream_t)stream, pre, reinterpret_cast<const ls_bf16x8*>(wsplit),
                       reinterpret_cast<ls_bf16x8*>(hsplit), yT, slot, T, H, BP);
    rc = check_launch("lstm_layer_fwd_persist_split");
  }
  return rc;
}


extern "C" int fac_lstm_persist_ok(int H, int B) { return fac::persist_usable(H, B) ? 1 : 0; }

extern "C" int fac_lstm_persist_stream_ok(fac_stream_t stream) { return fac::lstm_sync_slot((hipStream_t)stream) >= 0 ? 1 : 0; }

extern "C" int fac_pack_lstm_whh16(const float* w_hh, float* packed, int H, int transposed, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(w_hh && packed && H > 0 && H % 256 == 0, "pack_lstm_whh16: H must be a multiple of 256");
  hipLaunchKernelGGL(pack_whh16_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, w_hh, packed, H, transposed ? 1 : 0);
  return check_launch("pack_lstm_whh16");
}

extern "C" int fac_lstm_layer_fwd_persist(const float* pre, const float* whh16, float* hfrag, float* yT, float* gates_save,
                                          float* c_save, int T, int H, int B, int BP, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(pre && whh16 && hfrag && yT, "lstm_layer_fwd_persist: null pointer");
  FAC_REQUIRE((gates_save == nullptr) == (c_save == nullptr), "lstm_layer_fwd_persist: gates_save and c_save go together");
  FAC_REQUIRE(T > 0 && BP >= B && BP % 32 == 0, "lstm_layer_fwd_persist: bad T / BP");
  FAC_REQUIRE(persist_shape_ok(H, B), "lstm_layer_fwd_persist: H=%d B=%d is outside the resident kernel (fac_lstm_persist_ok)", H, B);
  const int slot = lstm_sync_slot((hipStream_t)stream);
  FAC_REQUIRE(slot >= 0, "lstm_layer_fwd_persist: more than %d streams in use", LSTM_SYNC_SLOTS);
  FwdKern kern = fwd_kernel_for(H, B);
  FAC_REQUIRE(kern != nullptr, "lstm_layer_fwd_persist: no kernel for H=%d", H);
  arm_timeout_word((hipStream_t)stream);        // (no-op once armed; an un-armed word makes the launch unusable)
  FAC_REQUIRE(persist_usable(H, B), "lstm_layer_fwd_persist: %d workgroups are not co-resident on this device, or an earlier resident "
              "launch timed out (%u): ask fac_lstm_persist_ok first and fall back to fac_lstm_layer_fwd", H / 8, persist_timeouts());
  int rc;
  {
    ResidentLaunch order((hipStream_t)stream);
    arm_timeout_word((hipStream_t)stream);
    hipLaunchKernelGGL(kern, dim3(H / 8), dim3(1024), 0, (hipStream_t)stream, pre, whh16, hfrag, yT, gates_save, c_save, slot, T, H, BP,
                       exchange_mode());
    rc = check_launch("lstm_layer_fwd_persist");
  }
  return rc;
}

extern "C" int fac_lstm_layer_bwd_persist(const float* dyT, const float* whh16t, const float* gates, const float* cs, float* dgates,
                                          float* scratch, int T, int H, int B, int BP, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(dyT && whh16t && gates && cs && dgates && scratch, "lstm_layer_bwd_persist: null pointer");
  FAC_REQUIRE(T > 0 && BP >= B && BP % 32 == 0, "lstm_layer_bwd_persist: bad T / BP");
  FAC_REQUIRE(persist_shape_ok(H, B), "lstm_layer_bwd_persist: H=%d B=%d is outside the resident kernel (fac_lstm_persist_ok)", H, B);
  const int slot = lstm_sync_slot((hipStream_t)stream);
  FAC_REQUIRE(slot >= 0, "lstm_layer_bwd_persist: more than %d streams in use", LSTM_SYNC_SLOTS);
  const int ncb = (B + 15) / 16;
  const long long hbuf = (long long)H * ncb * 16;
  float* partial = scratch;              // scratch = [partial 4*H*NC | dgates fragments T * 4*H*NC]
  float* dgfrag = scratch + 4 * hbuf;
  BwdKern kern = bwd_kernel_for(H, B);
  FAC_REQUIRE(kern != nullptr, "lstm_layer_bwd_persist: no kernel for H=%d", H);
  arm_timeout_word((hipStream_t)stream);        // (no-op once armed; an un-armed word makes the launch unusable)
  FAC_REQUIRE(persist_usable(H, B), "lstm_layer_bwd_persist: %d workgroups are not co-resident on this device, or an earlier resident "
              "launch timed out (%u)", H / 8, persist_timeouts());
  int rc;
  {
    ResidentLaunch order((hipStream_t)stream);
    arm_timeout_word((hipStream_t)stream);
    hipLaunchKernelGGL(kern, dim3(H / 8), dim3(1024), 0, (hipStream_t)stream, dyT, whh16t, gates, cs, dgates, partial, dgfrag, slot, T, H, BP,
                       exchange_mode());
    rc = check_launch("lstm_layer_bwd_persist");
  }
  return rc;
}
