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
