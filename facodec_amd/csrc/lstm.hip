// K5: SLSTM recurrence (reference: dac/model/encodec.py:272-288; nn.LSTM gate order i,f,g,o,
// zero initial state).  The input projection W_ih x_t + b runs as one big GEMM through
// fac_conv1d_fwd; this file holds the strictly sequential part.
//
// Work buffers are channel-major with the batch innermost: x / y are (H, T, BP), the gate
// pre-activations (4H, T, BP), BP = batch padded to 32.  A time step's slab is then rows of 32
// contiguous batch values (the MFMA B operand k = hidden index, n = batch is a coalesced row read),
// y[:, t] doubles as the h_t state of step t+1, and -- the reason for this order -- each channel's
// (t, b) plane is contiguous, so the input projection is ONE full-width GEMM for the conv kernel
// (C = H, "time" = T*BP) instead of T narrow ones.
//
// One launch per time step.  Workgroup = 8 hidden units x 4 gates = one 32-row MFMA block for a
// 32-wide batch block; its 16 (or 8) waves split the K = H reduction, stream their slice of W_hh with
// fully coalesced float4 loads (weights are read exactly once per step; the per-XCD L2 / MALL
// keeps them on chip between steps), reduce through LDS and apply the cell update in place.
#include "common.h"

namespace fac {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// NW waves split the K = H reduction; GPW = k-groups (of 8) per wave, 0 = run-time loop.  With GPW
// known at compile time every weight / state load of the step is issued before the first MFMA, so
// the step pays ONE memory round trip (the step is latency- and weight-bandwidth-bound: M = 32).
template <int NW, int GPW>
__global__ __launch_bounds__(NW * 64) void lstm_step_kernel(const float* __restrict__ pre_t,   // (4H, BP)
                                                            const float* __restrict__ whh,     // packed
                                                            const float* __restrict__ h_prev,  // fragment-packed h_{t-1} or null
                                                            float* __restrict__ h_next,        // fragment-packed h_t
                                                            float* __restrict__ c,             // (H, BP)
                                                            float* __restrict__ y_t,           // (H, BP) rows of stride rs
                                                            float* __restrict__ save_g,        // training: activated gates of this step (4H rows, stride rs) or null
                                                            float* __restrict__ save_c,        // training: c_t (H rows, stride rs) or null
                                                            int H, int BP, long long rs) {   // rs: row stride of pre / y
  __shared__ float red[NW][32][33];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int kq = lane >> 5;
  const int ublk = blockIdx.x;
  const int col0 = blockIdx.y * 32;

  // the gate pre-activations of this workgroup's 8 units do not depend on h: fetch them first
  float pre_v[4] = {0.f, 0.f, 0.f, 0.f};
  float c_old = 0.f;
  if (tid < 256) {
    const int u = tid >> 5, col = tid & 31;
    const int unit = ublk * 8 + u;
#pragma unroll
    for (int q = 0; q < 4; ++q) pre_v[q] = pre_t[(long long)(q * H + unit) * rs + col0 + col];
    if (h_prev != nullptr) c_old = c[(long long)unit * BP + col0 + col];
  }

  if (h_prev != nullptr) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kgs = H / 8;
    const int per_wave = GPW > 0 ? GPW : kgs / NW;
    const int kg0 = wave * per_wave;
    const float4* ap = reinterpret_cast<const float4*>(whh) + ((long long)ublk * kgs + kg0) * 64 + lane;
    // h_{t-1} in MFMA-B fragment order [col block][k group of 8][kq][32 batch][4 = k pair index]: one
    // coalesced float4 per lane and group, exactly like the weights (4x fewer VMEM instructions than
    // dword reads of a row-major state -- the step is VMEM-issue bound at M = 32)
    const float4* bp = reinterpret_cast<const float4*>(h_prev) + ((long long)blockIdx.y * kgs + kg0) * 64 + lane;
    if constexpr (GPW > 0) {
      float4 a4[GPW], b4[GPW];
#pragma unroll
      for (int g = 0; g < GPW; ++g) {
        a4[g] = ap[(long long)g * 64];
        b4[g] = bp[(long long)g * 64];
      }
#pragma unroll
      for (int g = 0; g < GPW; ++g) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].x, b4[g].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].y, b4[g].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].z, b4[g].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].w, b4[g].w, acc, 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int g = 0; g < per_wave; ++g) {
        const float4 a4 = ap[(long long)g * 64];
        const float4 b4 = bp[(long long)g * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kq;
      red[wave][row][l31] = acc[r];
    }
  }
  __syncthreads();
  if (tid < 256) {
    const int u = tid >> 5;
    const int col = tid & 31;
    const int unit = ublk * 8 + u;
    float gate[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float s = 0.f;
      if (h_prev != nullptr) {
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w][q * 8 + u][col];
      }
      gate[q] = pre_v[q] + s;
    }
    const float ig = sigmoid_f(gate[0]);
    const float fg = sigmoid_f(gate[1]);
    const float gg = tanhf(gate[2]);
    const float og = sigmoid_f(gate[3]);
    const long long o = (long long)unit * BP + col0 + col;
    const float c_new = __fadd_rn(__fmul_rn(fg, c_old), __fmul_rn(ig, gg));
    c[o] = c_new;
    const float hv = __fmul_rn(og, tanhf(c_new));
    y_t[(long long)unit * rs + col0 + col] = hv;
    if (save_g != nullptr) {   // what BPTT needs (nn.LSTM keeps the same in its reserve space)
      save_g[(long long)(0 * H + unit) * rs + col0 + col] = ig;
      save_g[(long long)(1 * H + unit) * rs + col0 + col] = fg;
      save_g[(long long)(2 * H + unit) * rs + col0 + col] = gg;
      save_g[(long long)(3 * H + unit) * rs + col0 + col] = og;
      save_c[(long long)unit * rs + col0 + col] = c_new;
    }
    // same value in fragment order for the next step: unit = 8*kg + 2*jj + kq
    const int kgn = unit >> 3, w8 = unit & 7;
    h_next[((((long long)blockIdx.y * (H >> 3) + kgn) * 2 + (w8 & 1)) * 32 + col) * 4 + (w8 >> 1)] = hv;
  }
}

// One BPTT step of an LSTM layer (elementwise part): dh = dy_t + W_hh^T dgates_{t+1} (the product `rec`, computed by
// the conv kernel, or null at the last step); gate derivatives from the saved activations; carries dc.
__global__ void lstm_gate_bwd_kernel(const float* __restrict__ dy_t, const float* __restrict__ rec,
                                     const float* __restrict__ gates_t, const float* __restrict__ c_t,
                                     const float* __restrict__ c_prev, float* __restrict__ dc,
                                     float* __restrict__ dgates_t, int H, int BP, long long rs, int first) {
  const long long n = (long long)H * BP;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int unit = (int)(i / BP), col = (int)(i - (long long)unit * BP);
    const long long o = (long long)unit * rs + col;
    const float ig = gates_t[o], fg = gates_t[(long long)H * rs + o], gg = gates_t[2ll * H * rs + o], og = gates_t[3ll * H * rs + o];
    const float ct = c_t[o], cp = c_prev ? c_prev[o] : 0.f;
    const float tc = tanhf(ct);
    float dh = dy_t[o];
    if (rec) dh += rec[i];
    const float d_o = dh * tc;
    const float dcv = dh * og * (1.f - tc * tc) + (first ? 0.f : dc[i]);
    dc[i] = dcv * fg;
    dgates_t[o] = dcv * gg * ig * (1.f - ig);
    dgates_t[(long long)H * rs + o] = dcv * cp * fg * (1.f - fg);
    dgates_t[2ll * H * rs + o] = dcv * ig * (1.f - gg * gg);
    dgates_t[3ll * H * rs + o] = d_o * og * (1.f - og);
  }
}

// ---- back-propagation through time, one layer, driven from the host side of the library (fac_lstm_layer_bwd): per step
//   rec partials:  W_hh^T dgates_{t+1}, as FOUR partial products (one per gate quarter of the 4H contraction) so that 4 x H/32
//                  workgroups (192 at H = 1536) stream W_hh^T exactly like the forward step streams W_hh -- fragment-packed
//                  weights and dgates, every load of the step issued before the first MFMA, 16 waves splitting K, LDS reduce;
//   gate kernel:   dh = dy_t + sum of the four partials, gate derivatives, dc carry; writes dgates_t row-major (what the
//                  weight-gradient GEMMs read) AND in MFMA-fragment order (what the next step's partial products read).
// Two launches per step issued from one C call per layer (the Python loop with three ctypes launches per step was host-bound:
// 1 920 dependent launches per layer pair), 4 x fewer VMEM instructions than the split-reduction conv kernel it replaces.
template <int NW, int GPW>
__global__ __launch_bounds__(NW * 64) void lstm_rec_bwd_kernel(const float* __restrict__ dg_frag,   // fragment-packed dgates_{t+1} (4H x BP)
                                                               const float* __restrict__ whh_t,     // fac_pack_lstm_whh_t
                                                               float* __restrict__ partial,         // (4, H, BP)
                                                               int H, int BP) {
  __shared__ float red[NW][32][33];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int kq = lane >> 5;
  const int ub = blockIdx.x, q = blockIdx.y, cb = blockIdx.z;
  const int kgs = H / 8;                       // k-groups per quarter
  const int per_wave = GPW > 0 ? GPW : kgs / NW;
  const int kg0 = wave * per_wave;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float4* ap = reinterpret_cast<const float4*>(whh_t) + (((long long)q * (H / 32) + ub) * kgs + kg0) * 64 + lane;
  const float4* bp = reinterpret_cast<const float4*>(dg_frag) + ((long long)cb * (4 * kgs) + (long long)q * kgs + kg0) * 64 + lane;
  if constexpr (GPW > 0) {
    float4 a4[GPW], b4[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
      a4[g] = ap[(long long)g * 64];
      b4[g] = bp[(long long)g * 64];
    }
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].x, b4[g].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].y, b4[g].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].z, b4[g].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].w, b4[g].w, acc, 0, 0, 0);
    }
  } else {
#pragma unroll 4
    for (int g = 0; g < per_wave; ++g) {
      const float4 a4 = ap[(long long)g * 64];
      const float4 b4 = bp[(long long)g * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kq;
    red[wave][row][l31] = acc[r];
  }
  __syncthreads();
  for (int o = tid; o < 32 * 32; o += NW * 64) {
    const int row = o >> 5, col = o & 31;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w][row][col];
    partial[((long long)q * H + ub * 32 + row) * BP + cb * 32 + col] = s;
  }
}

__global__ void lstm_gate_bwd2_kernel(const float* __restrict__ dy_t, const float* __restrict__ partial,   // (4, H, BP) or null
                                      const float* __restrict__ gates_t, const float* __restrict__ c_t,
                                      const float* __restrict__ c_prev, float* __restrict__ dc,
                                      float* __restrict__ dgates_t, float* __restrict__ dg_frag, int H, int BP, long long rs, int first) {
  const long long n = (long long)H * BP;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int unit = (int)(i / BP), col = (int)(i - (long long)unit * BP);
    const long long o = (long long)unit * rs + col;
    const float ig = gates_t[o], fg = gates_t[(long long)H * rs + o], gg = gates_t[2ll * H * rs + o], og = gates_t[3ll * H * rs + o];
    const float ct = c_t[o], cp = c_prev ? c_prev[o] : 0.f;
    const float tc = tanhf(ct);
    float dh = dy_t[o];
    if (partial) dh += ((partial[i] + partial[n + i]) + (partial[2 * n + i] + partial[3 * n + i]));
    const float d_o = dh * tc;
    const float dcv = dh * og * (1.f - tc * tc) + (first ? 0.f : dc[i]);
    dc[i] = dcv * fg;
    float dg[4];
    dg[0] = dcv * gg * ig * (1.f - ig);
    dg[1] = dcv * cp * fg * (1.f - fg);
    dg[2] = dcv * ig * (1.f - gg * gg);
    dg[3] = d_o * og * (1.f - og);
    const int cbk = col >> 5, c31 = col & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dgates_t[(long long)q * H * rs + o] = dg[q];
      const int r = q * H + unit;          // row of the (4H x BP) matrix; fragment order as the forward's h state
      dg_frag[((((long long)cbk * (4 * H >> 3) + (r >> 3)) * 2 + (r & 1)) * 32 + c31) * 4 + ((r & 7) >> 1)] = dg[q];
    }
  }
}

// W_hh (4H, H) -> transposed, fragment-packed per (gate quarter q, block of 32 hidden units): out[blk = q * H/32 + ub][kg][kq][32 units][4]
// holds W_hh[q*H + kg*8 + 2*jj + kq][ub*32 + i]  (jj = float4 component) -- the A operand of lstm_rec_bwd_kernel.
__global__ __launch_bounds__(256) void pack_whh_t_kernel(const float* __restrict__ w, float* __restrict__ out, int H) {
  const long long n = (long long)4 * H * H;
  const int kgs = H / 8, nub = H / 32;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256) {
    const int jj = o & 3;
    const int i = (o >> 2) & 31;
    const int kq = (o >> 7) & 1;
    const long long rest = o >> 8;
    const int kg = (int)(rest % kgs);
    const int blk = (int)(rest / kgs);
    const int q = blk / nub, ub = blk - q * nub;
    const long long row = (long long)q * H + kg * 8 + 2 * jj + kq;
    out[o] = w[row * H + ub * 32 + i];
  }
}

__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * (1.f - y[i] * y[i]);
}

// (B, H, T) -> (H, T, BP): per hidden unit, transpose the (b, t) plane through a 32x33 tile.
__global__ __launch_bounds__(256) void to_time_major_kernel(const float* __restrict__ x,
                                                            float* __restrict__ xT, int B, int H,
                                                            int T, int BP) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, h = blockIdx.y, b0 = blockIdx.z * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = b0 + ty + 8 * j, t = t0 + tx;
    tile[ty + 8 * j][tx] = (b < B && t < T) ? x[((long long)b * H + h) * T + t] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t0 + ty + 8 * j, b = b0 + tx;
    if (t < T) xT[((long long)h * T + t) * BP + b] = tile[tx][ty + 8 * j];
  }
}

__global__ __launch_bounds__(256) void from_time_major_kernel(const float* __restrict__ yT,
                                                              const float* __restrict__ skip,
                                                              const float* __restrict__ alpha,
                                                              float* __restrict__ out, int B, int H,
                                                              int T, int BP) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, h = blockIdx.y, b0 = blockIdx.z * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t0 + ty + 8 * j, b = b0 + tx;
    tile[ty + 8 * j][tx] = (t < T) ? yT[((long long)h * T + t) * BP + b] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = b0 + ty + 8 * j, t = t0 + tx;
    if (b < B && t < T) {
      const long long o = ((long long)b * H + h) * T + t;
      float v = tile[tx][ty + 8 * j];
      if (skip) v = __fadd_rn(v, skip[o]);
      if (alpha) v = snake_apply(v, alpha[h], snake_inv(alpha[h]));
      out[o] = v;
    }
  }
}

}  // namespace fac

extern "C" int fac_lstm_to_time_major(const float* x, float* xT, int B, int H, int T,
                                      fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(x && xT && B > 0 && H > 0 && T > 0, "lstm_to_time_major: bad arguments");
  const int BP = fac_pad32(B);
  dim3 grid((T + 31) / 32, H, BP / 32);
  hipLaunchKernelGGL(to_time_major_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, xT, B, H, T, BP);
  return check_launch("lstm_to_time_major");
}

extern "C" int fac_lstm_from_time_major(const float* yT, const float* skip, const float* alpha,
                                        float* out, int B, int H, int T, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(yT && out && B > 0 && H > 0 && T > 0, "lstm_from_time_major: bad arguments");
  const int BP = fac_pad32(B);
  dim3 grid((T + 31) / 32, H, BP / 32);
  hipLaunchKernelGGL(from_time_major_kernel, grid, dim3(256), 0, (hipStream_t)stream, yT, skip, alpha, out, B, H, T, BP);
  return check_launch("lstm_from_time_major");
}

extern "C" int fac_lstm_layer_fwd(const float* pre, const float* whh_packed, float* yT, float* c,
                                  int T, int H, int BP, fac_stream_t stream) {
  return fac_lstm_layer_fwd_from(pre, whh_packed, yT, c, T, H, BP, 0, stream);
}

extern "C" int fac_lstm_layer_fwd_from(const float* pre, const float* whh_packed, float* yT, float* c,
                                       int T, int H, int BP, int64_t step0, fac_stream_t stream) {
  return fac_lstm_layer_fwd_train(pre, whh_packed, yT, c, nullptr, nullptr, T, H, BP, step0, stream);
}

extern "C" int fac_lstm_layer_fwd_train(const float* pre, const float* whh_packed, float* yT, float* c, float* gates_save,
                                        float* c_save, int T, int H, int BP, int64_t step0, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE((gates_save == nullptr) == (c_save == nullptr), "lstm_layer_fwd: gates_save and c_save go together");
  FAC_REQUIRE(step0 >= 0, "lstm_layer_fwd: negative step0");
  FAC_REQUIRE(pre && whh_packed && yT && c, "lstm_layer_fwd: null pointer");
  FAC_REQUIRE(T > 0 && H > 0 && H % 64 == 0, "lstm_layer_fwd: H=%d must be a multiple of 64", H);
  FAC_REQUIRE(BP > 0 && BP % 32 == 0, "lstm_layer_fwd: BP=%d must be a multiple of 32", BP);
  dim3 grid(H / 8, BP / 32);
  const long long rs = (long long)T * BP;   // channel-major work buffers: row (= hidden unit) stride
  // 16 waves when the k-groups divide evenly, with the per-wave trip count fixed at compile time
  // for the shipped sizes (H = 1024 -> 8, H = 1536 -> 12 groups per wave)
  void (*kern)(const float*, const float*, const float*, float*, float*, float*, float*, float*, int, int, long long);
  int threads;
  const int kgs = H / 8;
  if (kgs % 16 == 0) {
    threads = 1024;
    switch (kgs / 16) {
      case 12: kern = lstm_step_kernel<16, 12>; break;
      case 8: kern = lstm_step_kernel<16, 8>; break;
      case 4: kern = lstm_step_kernel<16, 4>; break;
      case 2: kern = lstm_step_kernel<16, 2>; break;
      case 1: kern = lstm_step_kernel<16, 1>; break;
      default: kern = lstm_step_kernel<16, 0>; break;
    }
  } else {
    threads = 512;
    kern = lstm_step_kernel<8, 0>;
  }
  for (int t = 0; t < T; ++t) {
    // scratch = [c | h ping | h pong], H*BP floats each
    float* hp[2] = {c + (long long)H * BP, c + 2ll * H * BP};
    const long long g = step0 + t;   // global step index: the two h buffers alternate on it
    const float* h_prev = g == 0 ? nullptr : hp[(g - 1) & 1];
    hipLaunchKernelGGL(kern, grid, dim3(threads), 0, (hipStream_t)stream, pre + (long long)t * BP,
                       whh_packed, h_prev, hp[g & 1], c, yT + (long long)t * BP,
                       gates_save ? gates_save + (long long)t * BP : nullptr, c_save ? c_save + (long long)t * BP : nullptr, H,
                       BP, rs);
  }
  return check_launch("lstm_layer_fwd");
}

extern "C" int fac_lstm_gate_bwd(const float* dy_t, const float* rec, const float* gates_t, const float* c_t,
                                 const float* c_prev, float* dc, float* dgates_t, int H, int BP, int64_t rs, int first,
                                 fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(dy_t && gates_t && c_t && dc && dgates_t && H > 0 && BP > 0, "lstm_gate_bwd: bad arguments");
  const long long n = (long long)H * BP;
  const int blocks = (int)((n + 255) / 256);
  hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy_t, rec, gates_t, c_t, c_prev,
                     dc, dgates_t, H, BP, (long long)rs, first);
  return check_launch("lstm_gate_bwd");
}

extern "C" int fac_pack_lstm_whh_t(const float* w_hh, float* packed, int H, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(w_hh && packed && H > 0 && H % 64 == 0, "pack_lstm_whh_t: H must be a multiple of 64");
  hipLaunchKernelGGL(pack_whh_t_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, w_hh, packed, H);
  return check_launch("pack_lstm_whh_t");
}

extern "C" int fac_lstm_layer_bwd(const float* dyT, const float* whh_t_packed, const float* gates, const float* cs, float* dgates,
                                  float* scratch, int T, int H, int BP, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(dyT && whh_t_packed && gates && cs && dgates && scratch, "lstm_layer_bwd: null pointer");
  FAC_REQUIRE(T > 0 && H > 0 && H % 64 == 0 && BP > 0 && BP % 32 == 0, "lstm_layer_bwd: H must be a multiple of 64, BP of 32");
  const long long rs = (long long)T * BP, n = (long long)H * BP;
  // scratch = [dc (H*BP) | partial (4*H*BP) | dg_frag ping (4*H*BP) | dg_frag pong (4*H*BP)]
  float* dc = scratch;
  float* partial = scratch + n;
  float* frag[2] = {scratch + 5 * n, scratch + 9 * n};
  const int kgs = H / 8;
  void (*rec)(const float*, const float*, float*, int, int);
  int threads = 1024;
  if (kgs % 16 == 0) {
    switch (kgs / 16) {
      case 12: rec = lstm_rec_bwd_kernel<16, 12>; break;
      case 8: rec = lstm_rec_bwd_kernel<16, 8>; break;
      default: rec = lstm_rec_bwd_kernel<16, 0>; break;
    }
  } else {
    threads = 512;
    rec = lstm_rec_bwd_kernel<8, 0>;
  }
  const int gblocks = (int)((n + 255) / 256);
  for (int t = T - 1; t >= 0; --t) {
    const bool first = t == T - 1;
    if (!first)
      hipLaunchKernelGGL(rec, dim3(H / 32, 4, BP / 32), dim3(threads), 0, (hipStream_t)stream, frag[(t + 1) & 1], whh_t_packed, partial, H, BP);
    hipLaunchKernelGGL(lstm_gate_bwd2_kernel, dim3(gblocks), dim3(256), 0, (hipStream_t)stream, dyT + (long long)t * BP,
                       first ? nullptr : partial, gates + (long long)t * BP, cs + (long long)t * BP,
                       t > 0 ? cs + (long long)(t - 1) * BP : nullptr, dc, dgates + (long long)t * BP, frag[t & 1], H, BP, rs,
                       first ? 1 : 0);
  }
  return check_launch("lstm_layer_bwd");
}

extern "C" int fac_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(y && dy && dx && n > 0, "tanh_bwd: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, dy, dx, (long long)n);
  return check_launch("tanh_bwd");
}
