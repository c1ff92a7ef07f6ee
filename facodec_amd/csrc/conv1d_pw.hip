// Pointwise (k = 1) convs of the ResidualUnits at C <= 192 (dac/model/dac.py:33-42: y = x + conv1(snake(conv7(snake(x))))) as a
// STREAMING kernel.  At B = 32 these layers move four 590 MB tensors (input, residual, y, pre-activated copy y2) for
// 28-57 GFLOP: an elementwise pass over the same four tensors takes 0.48 ms (5.1 TB/s, tools/microbench/cu_stream_probe.hip),
// the fp32 MFMAs 0.2-0.4 ms, the tiled kernel (stage buffers, chunk barriers, one 133 KB workgroup per CU whose read phase
// and write phase alternate) 0.75-1.1 ms.  Here nothing is staged and nothing synchronises after the prologue:
//   * the whole weight matrix W[ci][co] (<= 147 KB) sits in LDS for the life of the workgroup: A fragments are ds_read_b32,
//     requested one K step ahead (spelled out: left alone the compiler reads them right before their MFMAs);
//   * a wave owns 32 time steps x MBW x 32 output channels of one clip at a time and walks a list of such blocks; its B
//     fragments come straight from global memory -- lane (k half kq, column l31) loads x[2s + kq][t0 + l31], two full
//     128-byte lines per wave instruction -- through a 16-deep register ring that runs on into the next block's rows;
//   * the epilogue works in the C/D register layout (each store instruction writes two full 128-byte lines): bias, residual
//     (requested one block ahead), y and y2 = snake(y).
// What sets the speed is how many waves keep requests in flight: s_waitcnt counts loads AND stores, so behind its ~100-200
// epilogue stores a wave's next loads wait for its own stores to drain at the chip's HBM rate -- unavoidable per wave, harmless
// only if enough other waves keep the memory system busy (measured at C = 192: one wave per SIMD 1.14 ms, two 0.87 ms, and
// 0.57 ms with the stores removed; spreading the epilogue over the next block's K loop inside ONE wave does not help, MFMAs
// and VALU of the same wave issue in order).  So the accumulators are kept small -- at most 3 blocks (48 registers) per wave,
// wider layers split their output channels over two waves that read the same input rows (the second read hits L1/L2) --
// and 16 waves (4 per SIMD, <= 128 registers) share the weights of one workgroup.
// The input is read from HBM once (the tiled kernel reads it once per C_out tile), sums run over ci in the same order as the
// tiled kernel and the epilogue applies (acc + bias) -> Snake -> activation -> + residual in the same order: identical results.
#include "conv1d_mfma.h"

namespace fac {

constexpr int PW_WAVES = 16;       // four waves per SIMD
constexpr int PW_D = 16;           // B-fragment ring depth (K steps in flight; C_in % 32 == 0)

// MBW: accumulator blocks per wave; NSPLIT: waves sharing a column block (C_out = 32 * MBW * NSPLIT)
template <int MBW, int NSPLIT>
__global__ __launch_bounds__(PW_WAVES * 64, 4) void conv1d_pw_kernel(ConvArgs a, int nblk, int n_items) {
  constexpr int CO = 32 * MBW * NSPLIT;                       // output channels of this workgroup's weight slice
  extern __shared__ __attribute__((aligned(16))) float Ws[];   // [C_in][CO] weights, then [5][CO]: bias, alpha_out, 1/alpha_out, alpha_y2, 1/alpha_y2
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kq = lane >> 5;
  // Layers wider than the LDS holds (C = 256, 384) are cut into C_out slices of CO channels, one slice per workgroup.  The
  // workgroups of one slice group are 8 apart in launch order, i.e. on the same XCD (blocks are dealt round-robin over the
  // 8 XCDs), and walk the same column blocks: the input rows they all read come from that XCD's L2 after the first.
  const int n_slices = a.C_out / CO;
  const int slice = (blockIdx.x >> 3) % n_slices;
  const int wg = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * n_slices));     // index among the workgroups of this slice
  const int n_wg = gridDim.x / n_slices;
  const int co_base = slice * CO;
  {
    constexpr int CO4 = CO / 4;
    const int n4 = a.C_in * CO4;
    for (int i = tid; i < n4; i += PW_WAVES * 64) {
      const int ci = i / CO4, c4 = i - ci * CO4;
      reinterpret_cast<float4*>(Ws)[i] = *reinterpret_cast<const float4*>(a.w + (long long)ci * a.C_out_pad + co_base + 4 * c4);
    }
    float* prm = Ws + a.C_in * CO;
    for (int i = tid; i < CO; i += PW_WAVES * 64) {
      const int co = co_base + i;
      prm[i] = a.bias ? a.bias[co] : 0.f;
      prm[CO + i] = a.alpha_out ? a.alpha_out[co] : 0.f;
      prm[2 * CO + i] = a.alpha_out ? snake_inv(a.alpha_out[co]) : 0.f;
      prm[3 * CO + i] = a.y2 ? a.alpha2[co] : 0.f;
      prm[4 * CO + i] = a.y2 ? snake_inv(a.alpha2[co]) : 0.f;
    }
  }
  __syncthreads();
  // the waves of one column block are neighbours (wave = 2 q + half): the second read of an input row hits L1
  const int half = NSPLIT == 2 ? (wave & 1) : 0;
  const int co0 = half * 32 * MBW;                             // first output channel of this wave
  const float* prm = Ws + a.C_in * CO + co0 + 4 * kq;          // parameters of row (co0 + row + 4 kq): prm[row]
  const int S = a.C_in >> 1;                                   // K steps of the 32x32x2 MFMA
  const long long xs2 = 2 * a.x_cs;
  const float* Wl = Ws + kq * CO + co0 + l31;                  // A fragment of step s, block m: Wl[2 s CO + 32 m]
  const int stride_items = n_wg * (PW_WAVES / NSPLIT);

  auto item_ptr = [&](int it, long long& yoff, bool& ok) -> const float* {
    const int b = it / nblk;
    const int t = (it - b * nblk) * 32 + l31;
    ok = t < a.T_out;
    const int tc = ok ? t : a.T_out - 1;                       // loads are unconditional (clamped), only stores are predicated
    yoff = (long long)b * a.y_bs + (long long)(co_base + co0 + 4 * kq) * a.y_cs + tc;
    return a.x + (long long)b * a.x_bs + (long long)kq * a.x_cs + tc;
  };

  int item = wg * (PW_WAVES / NSPLIT) + (wave / NSPLIT);
  if (item >= n_items) return;
  long long yoff;
  bool ok;
  const float* xp = item_ptr(item, yoff, ok);
  float xr[PW_D];
#pragma unroll
  for (int j = 0; j < PW_D; ++j) xr[j] = xp[(long long)j * xs2];

  for (;;) {
    f32x16 acc[MBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float* rp = a.res ? a.res + yoff : nullptr;
    float rv[2][16];
    auto ld_res = [&](int m, float (&dst)[16]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r] = rp ? rp[(long long)(m * 32 + (r & 3) + 8 * (r >> 2)) * a.y_cs] : 0.f;
    };
    // the ring runs on into the next block's rows during the last trip
    const int nxt = item + stride_items;
    const bool more = nxt < n_items;
    long long yoff_n = yoff;
    bool ok_n = ok;
    const float* xn = more ? item_ptr(nxt, yoff_n, ok_n) : xp;

    float av[2][MBW];
    const float* wl = Wl;
#pragma unroll
    for (int m = 0; m < MBW; ++m) av[0][m] = wl[32 * m];
    wl += 2 * CO;
    const float* pn = xp + (long long)PW_D * xs2;              // row the next refill reads
    for (int s0 = 0; s0 < S; s0 += PW_D) {
      if (s0 + PW_D >= S) {
        ld_res(0, rv[0]);                                      // the first residual block rides along with the last trip
        pn = xn;
      }
#pragma unroll
      for (int j = 0; j < PW_D; ++j) {
#pragma unroll
        for (int m = 0; m < MBW; ++m) av[(j + 1) & 1][m] = wl[32 * m];   // (the step after the last reads parameter rows: unused)
        wl += 2 * CO;
        __builtin_amdgcn_sched_barrier(0);
        const float bv = xr[j];
#pragma unroll
        for (int m = 0; m < MBW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][m], bv, acc[m], 0, 0, 0);
        xr[j] = *pn;
        pn += xs2;
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue in the C/D layout: register r of block m <-> row 32 m + (r & 3) + 8 (r >> 2) (+ co0 + 4 kq)
    float* yp = a.y ? a.y + yoff : nullptr;
    float* y2p = a.y2 ? a.y2 + yoff : nullptr;
#pragma unroll
    for (int m = 0; m < MBW; ++m) {
      if (m + 1 < MBW) ld_res(m + 1, rv[(m + 1) & 1]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2);
        float v = acc[m][r] + prm[row];
        if (a.alpha_out) v = snake_apply(v, prm[CO + row], prm[2 * CO + row]);
        if (a.act != FAC_ACT_NONE) v = apply_act_slow(v, a.act);
        v += rv[m & 1][r];
        if (ok) {
#ifdef FAC_PW_NT_STORES
          if (yp) __builtin_nontemporal_store(v, yp + (long long)row * a.y_cs);
          if (y2p) __builtin_nontemporal_store(snake_apply(v, prm[3 * CO + row], prm[4 * CO + row]), y2p + (long long)row * a.y_cs);
#else
          if (yp) yp[(long long)row * a.y_cs] = v;
          if (y2p) y2p[(long long)row * a.y_cs] = snake_apply(v, prm[3 * CO + row], prm[4 * CO + row]);
#endif
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

// C_out -> (channels per weight slice, slices): the slice must fit the LDS next to nothing else, and a wave needs >= 2
// MFMAs per input row it loads
static int pw_slice_channels(int C_in, int C_out) {
  if (C_in != C_out) return (C_out == 64 || C_out == 96 || C_out == 128 || C_out == 192) ? C_out : 0;
  switch (C_out) {
    case 64: case 96: case 128: case 192: return C_out;
    case 256: return 128;
    case 384: return 96;
    default: return 0;
  }
}

bool conv_pw_ok(const ConvArgs& a) {
  if (!(a.K == 1 && a.stride == 1 && a.n_phase == 1 && a.phase_shift == 0 && a.y_tstride == 1 && a.pad_left == 0 && !a.alpha_in &&
        !a.w1 && !a.w_batched && !conv_two_level(a) && a.T_in >= a.T_out))
    return false;
  const int co = pw_slice_channels(a.C_in, a.C_out);
  if (!co || a.C_out_pad != a.C_out) return false;
  if (a.C_in % (2 * PW_D) != 0 || a.C_in < 2 * PW_D) return false;
  if (((size_t)a.C_in * co + 5 * co) * sizeof(float) > 160 * 1024) return false;
  // enough column blocks that every wave slot of a 256-CU chip walks at least two of them
  const long long items = (long long)a.B * ((a.T_out + 31) / 32);
  const int nsplit = (co == 96 || co == 64) ? 1 : 2;                  // see conv_dispatch_pw
  const long long slots = (256 / (a.C_out / co)) * (PW_WAVES / nsplit);
  return items >= 2 * slots && (reinterpret_cast<unsigned long long>(a.w) & 15) == 0;
}

template <int MBW, int NSPLIT>
static int pw_launch(ConvArgs& a, hipStream_t s) {
  const int nblk = (a.T_out + 31) / 32;
  const long long n_items = (long long)a.B * nblk;
  if (n_items > 0x7fffffffll) {
    set_error("conv1d(pointwise): too many column blocks (%lld)", n_items);
    return FAC_ERR_ARG;
  }
  constexpr int CO = 32 * MBW * NSPLIT;
  const size_t lds = ((size_t)a.C_in * CO + 5 * CO) * sizeof(float);
  auto kern = conv1d_pw_kernel<MBW, NSPLIT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  constexpr int per_wg = PW_WAVES / NSPLIT;             // column blocks a workgroup works on at a time
  const int n_slices = a.C_out / CO;
  // one persistent workgroup per CU, in groups of 8 x n_slices (8 XCDs x the slices of one set of column blocks)
  long long groups = n_cu / (8 * n_slices);
  const long long need = (n_items + 8 * per_wg - 1) / (8 * per_wg);
  if (groups > need) groups = need;
  if (groups < 1) groups = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * 8 * n_slices)), dim3(PW_WAVES * 64), lds, s, a, nblk, (int)n_items);
  return check_launch("conv1d_pw");
}

int conv_dispatch_pw(ConvArgs& a, hipStream_t s) {
  switch (pw_slice_channels(a.C_in, a.C_out)) {
    case 64: return pw_launch<2, 1>(a, s);      // C = 64
    case 96: return pw_launch<3, 1>(a, s);      // C = 96; C = 384 in four slices
    case 128: return pw_launch<2, 2>(a, s);     // C = 128; C = 256 in two slices
    default: return pw_launch<3, 2>(a, s);      // C = 192
  }
}

}  // namespace fac
