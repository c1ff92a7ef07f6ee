// K6: weight-norm materialisation and packing into the layouts the MFMA kernels stream.
// Replaces torch.nn.utils.weight_norm's per-forward  w = v * (g / ||v||)
// (reference: dac/model/encodec.py:42-51 apply_parametrization_norm, dac/nn/layers.py:9-14).
// HBM-bound: every weight element is read once and written once, coalesced on both sides
// (the transposes go through a 32x33 LDS tile).
#include "common.h"
#include "prep_batch.h"

namespace fac {

// scale[i] = g[i] / sqrt(sum v[i,:]^2); one workgroup per slice.
// (bodies take the virtual workgroup index vb of a grid of vg: the single launches pass blockIdx / gridDim, the batch kernel of
// prep_batch.h its job-relative index)
__device__ __forceinline__ void wn_scale_body(const float* __restrict__ v, const float* __restrict__ g,
                                              float* __restrict__ scale, int slice_len, int vb, float* part) {
  const int i = vb;
  if (g == nullptr) {
    if (threadIdx.x == 0) scale[i] = 1.0f;
    return;
  }
  const float* p = v + (long long)i * slice_len;
  float s = 0.f;
  for (int j = threadIdx.x; j < slice_len; j += 256) {
    float x = p[j];
    s = fmaf(x, x, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = (part[0] + part[1]) + (part[2] + part[3]);
    scale[i] = __fdiv_rn(g[i], sqrtf(tot));
  }
}

__global__ __launch_bounds__(256) void wn_scale_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                       float* __restrict__ scale, int slice_len) {
  __shared__ float part[4];
  wn_scale_body(v, g, scale, slice_len, blockIdx.x, part);
}

// v (C_out, R) with R = C_in*K  ->  packed (R, C_out_pad); tiles of 32 x 32 through LDS.
__device__ __forceinline__ void pack_conv_body(const float* __restrict__ v, const float* __restrict__ scale,
                                               float* __restrict__ out, int C_out, int R, int C_out_pad, int R_pad,
                                               int bx, int by, float (*tile)[33]) {
  const int r0 = bx * 32;
  const int c0 = by * 32;
  const int tx = threadIdx.x & 31;
  const int ty = threadIdx.x >> 5;  // 0..7
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = c0 + ty + 8 * j;
    const int r = r0 + tx;
    float val = 0.f;
    if (co < C_out && r < R) {
      val = v[(long long)co * R + r];
      if (scale) val = __fmul_rn(val, scale[co]);
    }
    tile[ty + 8 * j][tx] = val;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j;
    const int co = c0 + tx;
    if (r < R_pad && co < C_out_pad) out[(long long)r * C_out_pad + co] = tile[tx][ty + 8 * j];   // rows R..R_pad: zeros
  }
}

__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                        float* __restrict__ out, int C_out, int R, int C_out_pad, int R_pad) {
  __shared__ float tile[32][33];
  pack_conv_body(v, scale, out, C_out, R, C_out_pad, R_pad, blockIdx.x, blockIdx.y, tile);
}

// ConvTranspose1d v (C_in, C_out, K=2s), scale per C_in  ->  packed[p][ci][j][co]
//   = v[ci][co][p + s*(1-j)] * scale[ci]. One workgroup per (ci, 32-co block); K is small.
__device__ __forceinline__ void pack_convtr_body(const float* __restrict__ v, const float* __restrict__ scale,
                                                 float* __restrict__ out, int C_in, int C_out, int s, int C_out_pad,
                                                 int bx, int by, float* tl) {   // tl: [32][K+1]
  const int K = 2 * s;
  const int ci = bx;                               // up to cin_pad(C_in): the padding channels get zero rows
  const int c0 = by * 32;
  const bool real = ci < C_in;
  const float sc = (scale && real) ? scale[ci] : 1.0f;
  const float* src = v + ((long long)(real ? ci : 0) * C_out + c0) * K;
  const int n = 32 * K;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int co = i / K, k = i - co * K;
    float val = 0.f;
    if (real && c0 + co < C_out) val = __fmul_rn(src[i], sc);
    tl[co * (K + 1) + k] = val;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int co = i & 31;
    const int pj = i >> 5;      // 0..K-1 = p*2 + j
    const int p = pj >> 1, j = pj & 1;
    const int k = p + s * (1 - j);
    if (c0 + co < C_out_pad)
      out[(((long long)p * cin_pad_dev(C_in) + ci) * 2 + j) * C_out_pad + c0 + co] = tl[co * (K + 1) + k];
  }
}

__global__ __launch_bounds__(256) void pack_convtr_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                          float* __restrict__ out, int C_in, int C_out, int s, int C_out_pad) {
  extern __shared__ float tl[];  // [32][K+1]
  pack_convtr_body(v, scale, out, C_in, C_out, s, C_out_pad, blockIdx.x, blockIdx.y, tl);
}

// ConvTranspose1d v (C_in, C_out, K = 2s) -> packed[ci][j][row] for the all-phases launch: row = tile*128 + (co % cpt)*s + p,
// cpt = 128 / s channels per tile (see fac_pack_convtr_w_rows).  One thread per output element, rows fastest.
__device__ __forceinline__ void pack_convtr_rows_body(const float* __restrict__ v, const float* __restrict__ scale,
                                                      float* __restrict__ out, int C_in, int C_out, int s, int R_pad,
                                                      long long n, int vb, int vg) {
  const int cpt = 128 / s;
  for (long long i = (long long)vb * 256 + threadIdx.x; i < n; i += (long long)vg * 256) {
    const int row = (int)(i % R_pad);
    const long long r2 = i / R_pad;
    const int j = (int)(r2 & 1);
    const int ci = (int)(r2 >> 1);
    const int tile = row >> 7, rl = row & 127;
    const int cl = rl / s, p = rl - cl * s;
    const int co = tile * cpt + cl;
    float val = 0.f;
    if (ci < C_in && cl < cpt && co < C_out) {
      val = v[((long long)ci * C_out + co) * (2 * s) + p + s * (1 - j)];
      if (scale) val = __fmul_rn(val, scale[ci]);
    }
    out[i] = val;
  }
}

__global__ __launch_bounds__(256) void pack_convtr_rows_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                               float* __restrict__ out, int C_in, int C_out, int s, int R_pad,
                                                               long long n) {
  pack_convtr_rows_body(v, scale, out, C_in, C_out, s, R_pad, n, blockIdx.x, gridDim.x);
}

// Weights of the data-gradient conv in one pass: out (C_in, C_out, K)[ci][co][k] = v (C_out, C_in, K)[co][ci][K-1-k] * scale[co]
// (weight norm applied, channels swapped, taps flipped -- torch's rows_fma + permute + flip + contiguous as one launch).
__device__ __forceinline__ void flip_transpose_w_body(const float* __restrict__ v, const float* __restrict__ scale,
                                                      float* __restrict__ out, int C_out, int C_in, int K, long long n, int vb, int vg) {
  for (long long i = (long long)vb * 256 + threadIdx.x; i < n; i += (long long)vg * 256) {
    const int k = (int)(i % K);
    const long long r = i / K;
    const int co = (int)(r % C_out);
    const int ci = (int)(r / C_out);
    float w = v[((long long)co * C_in + ci) * K + (K - 1 - k)];
    if (scale) w = __fmul_rn(w, scale[co]);
    out[i] = w;
  }
}

__global__ __launch_bounds__(256) void flip_transpose_w_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                               float* __restrict__ out, int C_out, int C_in, int K, long long n) {
  flip_transpose_w_body(v, scale, out, C_out, C_in, K, n, blockIdx.x, gridDim.x);
}

// One workgroup of the batch (prep_batch.h): the job it belongs to, then that job's body with the job-relative workgroup index.
__global__ __launch_bounds__(256) void pack_batch_kernel(const PrepJob* __restrict__ jobs, const int* __restrict__ first, int njobs) {
  __shared__ float sm[32 * 33];
  const int j = prep_find_job(first, njobs, blockIdx.x);
  const PrepJob& J = jobs[j];
  const int vb = blockIdx.x - first[j];
  const float* a = static_cast<const float*>(J.a);
  const float* b = static_cast<const float*>(J.b);
  float* o = static_cast<float*>(J.out);
  switch (J.kind) {
    case PK_WN_SCALE: wn_scale_body(a, b, o, J.i[0], vb, sm); break;
    case PK_CONV: pack_conv_body(a, b, o, J.i[0], J.i[1], J.i[2], J.i[3], vb % J.i[4], vb / J.i[4], reinterpret_cast<float (*)[33]>(sm)); break;
    case PK_CONVTR: pack_convtr_body(a, b, o, J.i[0], J.i[1], J.i[2], J.i[3], vb % J.i[4], vb / J.i[4], sm); break;
    case PK_CONVTR_ROWS: pack_convtr_rows_body(a, b, o, J.i[0], J.i[1], J.i[2], J.i[3], J.n, vb, J.nblocks); break;
    case PK_FLIP_T: flip_transpose_w_body(a, b, o, J.i[0], J.i[1], J.i[2], J.n, vb, J.nblocks); break;
    default: break;
  }
}

int prep_launch_pack(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s) {
  hipLaunchKernelGGL(pack_batch_kernel, dim3(total), dim3(256), 0, s, jobs, first, njobs);
  return check_launch("pack_batch");
}

// W_hh (4H, H) -> packed[ublk][kg][kq][i][4]: for unit block ublk (8 hidden units) the 32 gate
// rows i = gate*8 + u, k = kg*8 + 2*jj + kq  (jj = 0..3 is the float4 component).
__global__ __launch_bounds__(256) void pack_whh_kernel(const float* __restrict__ w,
                                                       float* __restrict__ out, int H) {
  const long long n = (long long)4 * H * H;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n;
       o += (long long)gridDim.x * 256) {
    const int jj = o & 3;
    const int i = (o >> 2) & 31;
    const int kq = (o >> 7) & 1;
    const long long rest = o >> 8;
    const int kgs = H / 8;
    const int kg = rest % kgs;
    const int ublk = rest / kgs;
    const int gate = i >> 3, u = i & 7;
    const int row = gate * H + ublk * 8 + u;
    const int k = kg * 8 + 2 * jj + kq;
    out[o] = w[(long long)row * H + k];
  }
}

}  // namespace fac

extern "C" int fac_wn_scale(const float* v, const float* g, float* scale, int n_slices,
                            int slice_len, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(scale && n_slices > 0 && slice_len > 0 && (v || !g), "wn_scale: bad arguments");
  if (prep_recording()) {
    PrepJob j{}; j.a = v; j.b = g; j.out = scale; j.kind = PK_WN_SCALE; j.nblocks = n_slices; j.i[0] = slice_len;
    return prep_record(PU_PACK, j);
  }
  hipLaunchKernelGGL(wn_scale_kernel, dim3(n_slices), dim3(256), 0, (hipStream_t)stream, v, g,
                     scale, slice_len);
  return check_launch("wn_scale");
}

extern "C" int fac_pack_conv_w(const float* v, const float* scale, float* packed, int C_out,
                               int C_in, int K, int C_out_pad, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(v && packed && C_out > 0 && C_in > 0 && K > 0, "pack_conv_w: bad arguments");
  FAC_REQUIRE(C_out_pad % 32 == 0 && C_out_pad >= C_out, "pack_conv_w: C_out_pad must be a multiple of 32");
  const int R = C_in * K;
  const int R_pad = cin_pad_dev(C_in) * K;              // the zero rows of the padding channels are written too
  dim3 grid((R_pad + 31) / 32, C_out_pad / 32);
  if (prep_recording()) {
    PrepJob j{}; j.a = v; j.b = scale; j.out = packed; j.kind = PK_CONV; j.nblocks = (int)(grid.x * grid.y);
    j.i[0] = C_out; j.i[1] = R; j.i[2] = C_out_pad; j.i[3] = R_pad; j.i[4] = (int)grid.x;
    return prep_record(PU_PACK, j);
  }
  hipLaunchKernelGGL(pack_conv_kernel, grid, dim3(256), 0, (hipStream_t)stream, v, scale, packed,
                     C_out, R, C_out_pad, R_pad);
  return check_launch("pack_conv_w");
}

extern "C" int fac_pack_convtr_w(const float* v, const float* scale, float* packed, int C_in,
                                 int C_out, int stride, int C_out_pad, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(v && packed && C_out > 0 && C_in > 0 && stride > 0, "pack_convtr_w: bad arguments");
  FAC_REQUIRE(C_out_pad % 32 == 0 && C_out_pad >= C_out, "pack_convtr_w: C_out_pad must be a multiple of 32");
  dim3 grid(cin_pad_dev(C_in), C_out_pad / 32);
  const size_t lds = (size_t)32 * (2 * stride + 1) * sizeof(float);
  if (prep_recording()) {
    FAC_REQUIRE(2 * stride <= 32, "pack_convtr_w: a recorded launch takes strides up to 16");
    PrepJob j{}; j.a = v; j.b = scale; j.out = packed; j.kind = PK_CONVTR; j.nblocks = (int)(grid.x * grid.y);
    j.i[0] = C_in; j.i[1] = C_out; j.i[2] = stride; j.i[3] = C_out_pad; j.i[4] = (int)grid.x;
    return prep_record(PU_PACK, j);
  }
  hipLaunchKernelGGL(pack_convtr_kernel, grid, dim3(256), lds, (hipStream_t)stream, v, scale,
                     packed, C_in, C_out, stride, C_out_pad);
  return check_launch("pack_convtr_w");
}

extern "C" int fac_pack_convtr_w_rows(const float* v, const float* scale, float* packed, int C_in, int C_out, int stride,
                                      fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(v && packed && C_out > 0 && C_in > 0 && stride > 0 && stride <= 128, "pack_convtr_w_rows: bad arguments");
  const int R_pad = fac_convtr_rows(C_out, stride);
  const long long n = (long long)cin_pad_dev(C_in) * 2 * R_pad;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  if (prep_recording()) {
    PrepJob j{}; j.a = v; j.b = scale; j.out = packed; j.kind = PK_CONVTR_ROWS; j.nblocks = blocks; j.n = n;
    j.i[0] = C_in; j.i[1] = C_out; j.i[2] = stride; j.i[3] = R_pad;
    return prep_record(PU_PACK, j);
  }
  hipLaunchKernelGGL(pack_convtr_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, scale, packed, C_in, C_out,
                     stride, R_pad, n);
  return check_launch("pack_convtr_w_rows");
}

extern "C" int fac_flip_transpose_w(const float* v, const float* scale, float* out, int C_out, int C_in, int K, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(v && out && C_out > 0 && C_in > 0 && K > 0, "flip_transpose_w: bad arguments");
  const long long n = (long long)C_out * C_in * K;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  if (prep_recording()) {
    PrepJob j{}; j.a = v; j.b = scale; j.out = out; j.kind = PK_FLIP_T; j.nblocks = blocks; j.n = n;
    j.i[0] = C_out; j.i[1] = C_in; j.i[2] = K;
    return prep_record(PU_PACK, j);
  }
  hipLaunchKernelGGL(flip_transpose_w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, scale, out, C_out, C_in, K, n);
  return check_launch("flip_transpose_w");
}

extern "C" int fac_pack_lstm_whh(const float* w_hh, float* packed, int H, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(w_hh && packed && H > 0 && H % 8 == 0, "pack_lstm_whh: H must be a multiple of 8");
  hipLaunchKernelGGL(pack_whh_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, w_hh, packed, H);
  return check_launch("pack_lstm_whh");
}
