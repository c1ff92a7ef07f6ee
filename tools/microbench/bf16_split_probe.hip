// Probe (not part of the library): can fp32 convolutions run on the bf16 matrix pipe without losing fp32
// accuracy?  x = hi + mid + lo with three round-to-nearest bf16 terms is EXACT for fp32 x (8+8+8 significand
// bits); a*b = sum of the 9 cross products, each exact in fp32; dropping the three smallest (mid*lo, lo*mid,
// lo*lo, <= 2^-24 relative) leaves 6 MFMAs.  Measures (1) the sustained rate of v_mfma_f32_32x32x16_bf16
// against v_mfma_f32_32x32x2_f32 and (2) the error of a K = 4096 dot-product tile against fp64 for
// fp32-MFMA / 6-term / 9-term.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rate_bf16(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <bool RANDOM>
__global__ __launch_bounds__(1024) void rate_bf16_lds(float* out, int iters) {
  extern __shared__ float pin[];
  bf16x8 a[6], b[6];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int q = 0; q < 6; ++q)
    for (int i = 0; i < 8; ++i) {
      if (RANDOM) {   // operands with random mantissas / signs: the data-dependent power of real activations
        h = h * 1664525u + 1013904223u; a[q][i] = (__bf16)(((int)(h >> 8) & 0xffff) / 32768.0f - 1.0f);
        h = h * 1664525u + 1013904223u; b[q][i] = (__bf16)(((int)(h >> 8) & 0xffff) / 32768.0f - 1.0f);
      } else {
        a[q][i] = (__bf16)(float)(threadIdx.x + i + q); b[q][i] = (__bf16)(float)(i + 1 - q);
      }
    }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters / 6; ++it) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {   // distinct operands per MFMA, 4 accumulators: the conv kernel's issue pattern
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[q], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[5 - q], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[5 - q], b[q], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[5 - q], b[5 - q], c3, 0, 0, 0);
    }
  }
  if (threadIdx.x == 0) pin[0] = c0[0];
  out[(blockIdx.x * 256 + threadIdx.x) % (2048 * 256)] = c0[0] + c1[1] + c2[2] + c3[3] + pin[0];
}

__global__ __launch_bounds__(256) void rate_f32(float* out, int iters) {
  float a = threadIdx.x, b = 1.5f;
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// one wave: C(32x32) = A(32xK) * B(Kx32); A row-major (32,K), B row-major (K,32)
__global__ __launch_bounds__(64) void tile_gemm(const float* A, const float* B, float* Cf32, float* C6, float* C9, int K) {
  const int lane = threadIdx.x, l31 = lane & 31, kq = lane >> 5;
  f32x16 cf = {0}, c6 = {0}, c9 = {0};
  for (int k0 = 0; k0 < K; k0 += 2) cf = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k0 + kq], B[(k0 + kq) * 32 + l31], cf, 0, 0, 0);
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int i = 0; i < 8; ++i) {
      __bf16 h, m, l;
      split3(A[l31 * K + k0 + 8 * kq + i], h, m, l); ah[i] = h; am[i] = m; al[i] = l;
      split3(B[(k0 + 8 * kq + i) * 32 + l31], h, m, l); bh[i] = h; bm[i] = m; bl[i] = l;
    }
    // small terms first
    c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c6, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, c9, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, c9, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, c9, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c9, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c9, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c9, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c9, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c9, 0, 0, 0);
    c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c9, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kq;
    Cf32[row * 32 + l31] = cf[r]; C6[row * 32 + l31] = c6[r]; C9[row * 32 + l31] = c9[r];
  }
}

int main() {
  float* d;
  const int blocks = 256 * 8;
  CK(hipMalloc(&d, blocks * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(rate_f32, dim3(blocks), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(rate_bf16, dim3(blocks), dim3(256), 0, 0, d, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flop = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * (which == 0 ? 2 : 16);
      printf("%s: %.3f ms  %.1f TFLOP/s\n", which == 0 ? "mfma_f32_32x32x2_f32 " : "mfma_f32_32x32x16_bf16", ms, flop / ms / 1e9);
    }
  }
  // one MFMA wave per SIMD vs two: does a single wave keep the bf16 pipe busy?  (100 KB of LDS pins one
  // workgroup per CU; 256 threads = 1 wave per SIMD, 512 = 2)
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rate_bf16_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rate_bf16_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  for (int random = 0; random < 2; ++random)
    for (int threads : {256, 512}) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (random) hipLaunchKernelGGL(rate_bf16_lds<true>, dim3(256 * 4), dim3(threads), 100 * 1024, 0, d, iters * 4);
        else hipLaunchKernelGGL(rate_bf16_lds<false>, dim3(256 * 4), dim3(threads), 100 * 1024, 0, d, iters * 4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)256 * 4 * (threads / 64) * (iters * 4 / 6 * 6) * 4 * 2.0 * 32 * 32 * 16;
        printf("bf16 MFMA, 1 WG/CU, %d waves/SIMD, %s operands: %.3f ms  %.1f TFLOP/s\n", threads / 256,
               random ? "random" : "constant", ms, flop / ms / 1e9);
      }
    }
  // numerics
  for (int K : {256, 4096}) {
    std::vector<float> A(32 * K), B(K * 32);
    srand(1);
    auto rnd = []() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(sqrt(-2 * log(u)) * cos(6.283185307179586 * v)); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 3 * 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tile_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, dC + 1024, dC + 2048, K);
    std::vector<float> C(3 * 1024);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double err[4] = {0, 0, 0, 0}, nrm = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0; float f = 0;
      for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; f = fmaf(A[i * K + k], B[k * 32 + j], f); }
      nrm = fmax(nrm, fabs(ref));
      for (int v = 0; v < 3; ++v) err[v] = fmax(err[v], fabs(C[v * 1024 + i * 32 + j] - ref));
      err[3] = fmax(err[3], fabs((double)f - ref));
    }
    printf("K=%d max|err| / max|ref|:  fp32 MFMA %.3e   bf16x3 6-term %.3e   bf16x3 9-term %.3e   sequential fp32 fma (CPU) %.3e\n",
           K, err[0] / nrm, err[1] / nrm, err[2] / nrm, err[3] / nrm);
  }
  return 0;
}
