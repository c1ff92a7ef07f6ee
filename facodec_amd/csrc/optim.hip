// Optimiser step of the training path on flat parameter / gradient arenas (reference: optimizers.py:72-108 builds one
// torch AdamW(lr, betas (0.9, 0.98), eps 1e-9, weight_decay 0.1) + ExponentialLR per model key; train.py:362-374 clips
// each key's gradient norm to 1000 and steps).  One launch per key instead of ~10 ATen kernels per tensor.
#include "common.h"
#include "../../include/facodec_hip.h"

namespace fac {

constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
  __shared__ float red[256];
  // 16-byte loads, four independent sums per lane (round 6: the scalar loop with one dependent fmaf chain ran at 1.5 TB/s); the
  // order is fixed by (n, grid), so the norm is reproducible run to run
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long long n4 = (reinterpret_cast<unsigned long long>(g) & 15) == 0 ? n >> 2 : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 q = g4[i];
    s0 = fmaf(q.x, q.x, s0); s1 = fmaf(q.y, q.y, s1); s2 = fmaf(q.z, q.z, s2); s3 = fmaf(q.w, q.w, s3);
  }
  for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s0 = fmaf(g[i], g[i], s0);
  const float s = (s0 + s1) + (s2 + s3);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// norm_out[0] = ||g||, norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_)
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int n_part, float max_norm,
                                                          float* __restrict__ norm_out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_part; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(red[0]);
    norm_out[0] = nrm;
    const float c = max_norm / (nrm + 1e-6f);
    norm_out[1] = c < 1.f ? c : 1.f;
  }
}

// torch.optim.AdamW (decoupled weight decay): p *= 1 - lr*wd; m, v EMAs; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                             const float* __restrict__ clip) {
  const float gs = clip ? clip[1] : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi;
  }
}

// ---- masked step: which parameters take part is decided ON THE DEVICE from per-parameter flags that travelled with the
// gradient arena through the data-parallel all-reduce (flag > 0 on any rank -> every rank steps the parameter, like torch
// AdamW under DDP steps every parameter whose .grad is set).  No host read of the flags, one launch per key.
// prepare: steps[j] += 1 and the bias corrections of parameter j where flags[j] > 0; bc[2j] = 0 marks "skip".
__global__ void adamw_prepare_kernel(const float* __restrict__ flags, int* __restrict__ steps, float* __restrict__ bc, int P,
                                     float b1, float b2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= P) return;
  if (flags[j] > 0.f) {
    const int s = ++steps[j];
    bc[2 * j] = (float)(1.0 - pow((double)b1, (double)s));
    bc[2 * j + 1] = (float)sqrt(1.0 - pow((double)b2, (double)s));
  } else {
    bc[2 * j] = 0.f;
    bc[2 * j + 1] = 1.f;
  }
}

constexpr int ADAMW_CHUNK = 2048;   // elements per workgroup trip

__global__ __launch_bounds__(256) void adamw_masked_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, long long n, const long long* __restrict__ off,
                                                           int P, const float* __restrict__ bc, float lr, float b1, float b2,
                                                           float eps, float wd, const float* __restrict__ clip) {
  const float gs = clip ? clip[1] : 1.f;
  const long long n_chunks = (n + ADAMW_CHUNK - 1) / ADAMW_CHUNK;
  const bool aligned16 = ((reinterpret_cast<unsigned long long>(p) | reinterpret_cast<unsigned long long>(g) |
                           reinterpret_cast<unsigned long long>(m) | reinterpret_cast<unsigned long long>(v)) & 15) == 0;
  for (long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const long long base = c * ADAMW_CHUNK;
    // parameter that owns the chunk's first element: largest j with off[j] <= base (uniform over the workgroup)
    int lo = 0, hi = P - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (off[mid] <= base) lo = mid; else hi = mid - 1;
    }
    int j = lo;
    long long next = off[j + 1];
    float bc1 = bc[2 * j], bc2s = bc[2 * j + 1];
    if (next >= base + ADAMW_CHUNK && base + ADAMW_CHUNK <= n && aligned16) {
      // the whole chunk belongs to parameter j (all but ~1 500 of the ~70 000 chunks): 16-byte accesses, the same arithmetic per
      // element (round 6: the 4-byte form ran at 2.2 TB/s of its 28 B per element)
      if (bc1 == 0.f) continue;
#pragma unroll
      for (int k = 0; k < ADAMW_CHUNK / 1024; ++k) {
        const long long i = base + k * 1024 + 4 * threadIdx.x;
        const float4 g4 = *reinterpret_cast<const float4*>(g + i);
        float4 p4 = *reinterpret_cast<const float4*>(p + i);
        float4 m4 = *reinterpret_cast<const float4*>(m + i);
        float4 v4 = *reinterpret_cast<const float4*>(v + i);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float pv[4] = {p4.x, p4.y, p4.z, p4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gi = gv[e] * gs;
          float pi = pv[e] * (1.f - lr * wd);
          const float mi = b1 * mv[e] + (1.f - b1) * gi;
          const float vi = b2 * vv[e] + (1.f - b2) * gi * gi;
          mv[e] = mi;
          vv[e] = vi;
          pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
          pv[e] = pi;
        }
        *reinterpret_cast<float4*>(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < ADAMW_CHUNK / 256; ++k) {
      const long long i = base + k * 256 + threadIdx.x;
      if (i >= n) break;
      while (i >= next) {
        ++j;
        next = off[j + 1];
        bc1 = bc[2 * j];
        bc2s = bc[2 * j + 1];
      }
      if (bc1 == 0.f) continue;
      const float gi = g[i] * gs;
      float pi = p[i] * (1.f - lr * wd);
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
      p[i] = pi;
    }
  }
}

// Many tensors -> their slices of an arena in a few launches: the gradients autograd left as separate tensors, folded into the
// optimiser's gradient arena (FlatAdamW._rebind).  torch._foreach_copy_ did this with ~17 us of host time per tensor and one
// hipMemcpyAsync for every second tensor (622 per step, round 6), at the one point of the step where the device has nothing queued.
constexpr int GC_MAX = 112;              // entries per launch (the table travels as a kernel argument: < 4 KB)
constexpr int GC_CHUNK = 256 * 4 * 8;    // elements per workgroup
struct GatherTable {
  const float* src[GC_MAX];
  float* dst[GC_MAX];
  unsigned n[GC_MAX];
  unsigned block0[GC_MAX + 1];           // first workgroup of entry j (prefix sum)
  int count;
};

__global__ __launch_bounds__(256) void gather_copy_kernel(GatherTable t) {
  int lo = 0, hi = t.count;              // entry of this workgroup: last j with block0[j] <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.block0[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const float* __restrict__ src = t.src[lo];
  float* __restrict__ dst = t.dst[lo];
  const unsigned n = t.n[lo];
  const unsigned base = (blockIdx.x - t.block0[lo]) * (unsigned)GC_CHUNK;
  const unsigned end = base + GC_CHUNK < n ? base + GC_CHUNK : n;
  if ((((unsigned long long)src | (unsigned long long)dst) & 15ull) == 0) {
    const unsigned end4 = base + ((end - base) & ~3u);
    for (unsigned i = base + 4 * threadIdx.x; i < end4; i += 1024)
      *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
    for (unsigned i = end4 + threadIdx.x; i < end; i += 256) dst[i] = src[i];
  } else {
    for (unsigned i = base + threadIdx.x; i < end; i += 256) dst[i] = src[i];
  }
}

}  // namespace fac

extern "C" int fac_gather_copy(const void* const* src, void* const* dst, const int64_t* n, int count, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(count >= 0 && (count == 0 || (src && dst && n)), "gather_copy: bad arguments");
  for (int j0 = 0; j0 < count; j0 += GC_MAX) {
    GatherTable t;
    t.count = count - j0 < GC_MAX ? count - j0 : GC_MAX;
    unsigned blocks = 0;
    for (int j = 0; j < t.count; ++j) {
      FAC_REQUIRE(src[j0 + j] && dst[j0 + j] && n[j0 + j] > 0 && n[j0 + j] < (1ll << 32), "gather_copy: bad entry %d", j0 + j);
      t.src[j] = reinterpret_cast<const float*>(src[j0 + j]);
      t.dst[j] = reinterpret_cast<float*>(dst[j0 + j]);
      t.n[j] = (unsigned)n[j0 + j];
      t.block0[j] = blocks;
      blocks += (unsigned)((n[j0 + j] + GC_CHUNK - 1) / GC_CHUNK);
    }
    t.block0[t.count] = blocks;
    hipLaunchKernelGGL(gather_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
  }
  return check_launch("gather_copy");
}

extern "C" int fac_adamw_step_masked(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* offsets,
                                     int n_params, const float* flags, int32_t* steps, float* bc, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, const float* clip, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(p && g && m && v && offsets && flags && steps && bc && n > 0 && n_params > 0, "adamw_step_masked: bad arguments");
  hipLaunchKernelGGL(adamw_prepare_kernel, dim3((n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream, flags, steps, bc,
                     n_params, beta1, beta2);
  const long long n_chunks = (n + ADAMW_CHUNK - 1) / ADAMW_CHUNK;
  const int blocks = (int)(n_chunks < 16384 ? n_chunks : 16384);
  hipLaunchKernelGGL(adamw_masked_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                     reinterpret_cast<const long long*>(offsets), n_params, bc, lr, beta1, beta2, eps, weight_decay, clip);
  return check_launch("adamw_step_masked");
}

extern "C" int fac_grad_norm_clip(const float* g, int64_t n, float max_norm, float* scratch, float* norm_out, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(g && scratch && norm_out && n > 0, "grad_norm_clip: bad arguments");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, (long long)n, scratch);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, SUMSQ_BLOCKS, max_norm, norm_out);
  return check_launch("grad_norm_clip");
}

extern "C" int fac_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t step, const float* clip, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2_sqrt, clip);
  return check_launch("adamw_step");
}
