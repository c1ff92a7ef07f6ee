// Small elementwise helpers of the training path (gradient bookkeeping).
#include "common.h"
#include "../../include/facodec_hip.h"

namespace fac {

// out[b][i] = a[b][i] * w[b] + sign * c[b][i]   (w, c optional)
__global__ void rows_fma_kernel(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ c,
                                float* __restrict__ out, long long per, float sign, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = a[i];
    if (w) v *= w[i / per];
    if (c) v += sign * c[i];
    out[i] = v;
  }
}

}  // namespace fac

extern "C" int fac_rows_fma(const float* a, const float* w, const float* c, float* out, int B, int64_t per, float sign,
                            fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(a && out && B > 0 && per > 0, "rows_fma: bad arguments");
  const long long n = (long long)B * per;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(rows_fma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, w, c, out, (long long)per, sign, n);
  return check_launch("rows_fma");
}

// Softmax cross-entropy over rows (N, C) with int64 labels (F.cross_entropy, mean reduction): one workgroup per row.
// loss_row[n] = logsumexp(x_n) - x_n[label];  dlogits = (softmax - onehot) * scale  (scale = upstream / N).
namespace fac {

__global__ __launch_bounds__(256) void ce_row_kernel(const float* __restrict__ x, const long long* __restrict__ label,
                                                     float* __restrict__ loss_row, float* __restrict__ dx, int Cn, float scale) {
  __shared__ float red[256];
  const long long n = blockIdx.x;
  const int tid = threadIdx.x;
  const float* xr = x + n * Cn;
  float mx = -INFINITY;
  for (int c = tid; c < Cn; c += 256) mx = fmaxf(mx, xr[c]);
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
  mx = red[0];
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < Cn; c += 256) s += expf(xr[c] - mx);
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  s = red[0];
  const long long lb = label[n];
  if (tid == 0 && loss_row) loss_row[n] = logf(s) + mx - xr[lb];
  if (dx) {
    float* dr = dx + n * Cn;
    for (int c = tid; c < Cn; c += 256) dr[c] = (expf(xr[c] - mx) / s - (c == lb ? 1.f : 0.f)) * scale;
  }
}

__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ v, float* __restrict__ out, long long n, float scale) {
  __shared__ float red[256];
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) s += v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

}  // namespace fac

extern "C" int fac_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits, float* scratch,
                                 int64_t N, int C, float grad_scale, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(logits && labels && scratch && (loss || dlogits) && N > 0 && C > 0, "cross_entropy: bad arguments");
  hipLaunchKernelGGL(ce_row_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)labels,
                     loss ? scratch : nullptr, dlogits, C, grad_scale);
  if (loss) hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, loss, (long long)N, 1.0f / (float)N);
  return check_launch("cross_entropy");
}

// Focal modulation of a MEAN cross entropy (losses.py:264-276, train.py:153 gamma = 2): ce -> (1 - exp(-ce))^2 * ce.
// io[0] = ce in; out[0] = loss, out[1] = d loss / d ce  (both on the device: no host round trip in the step).
namespace fac {
__global__ void focal_scalar_kernel(const float* __restrict__ ce, float* __restrict__ out, float gamma) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float c = ce[0];
    const float p = expf(-c);
    const float q = 1.f - p;
    const float qg = powf(q, gamma);
    out[0] = qg * c;
    out[1] = qg + (q > 0.f ? gamma * powf(q, gamma - 1.f) * p * c : 0.f);
  }
}

// Random-crop batching of train.py:188-212 on the device: dst[b][c][t] = src[b][c][start[b] * scale + t],
// src (B, C, T_src), dst (B, C, T_dst), start (B) int64 in units of `scale` samples (frames -> hop 300 for the waves).
__global__ void crop_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, const long long* __restrict__ start,
                                 int C, long long T_src, int T_dst, int scale, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T_dst);
    const long long bc = i / T_dst;
    const long long b = bc / C;
    const long long s = start[b] * scale + t;
    dst[i] = (s >= 0 && s < T_src) ? src[bc * T_src + s] : 0.f;
  }
}
}  // namespace fac

extern "C" int fac_focal_scalar(const float* ce, float* out2, float gamma, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(ce && out2 && gamma >= 0.f, "focal_scalar: bad arguments");
  hipLaunchKernelGGL(focal_scalar_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ce, out2, gamma);
  return check_launch("focal_scalar");
}

extern "C" int fac_crop_rows(const float* src, float* dst, const int64_t* start, int B, int C, int64_t T_src, int T_dst, int scale,
                             fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(src && dst && start && B > 0 && C > 0 && T_src > 0 && T_dst > 0 && scale > 0, "crop_rows: bad arguments");
  const long long n = (long long)B * C * T_dst;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(crop_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, (const long long*)start, C,
                     (long long)T_src, T_dst, scale, n);
  return check_launch("crop_rows");
}
