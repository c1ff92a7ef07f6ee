// Shared helpers for the libfacodec_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/facodec_hip.h"

namespace fac {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return FAC_ERR_LAUNCH;
  }
  return FAC_OK;
}

#define FAC_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::fac::set_error(__VA_ARGS__);    \
      return FAC_ERR_ARG;               \
    }                                   \
  } while (0)

// Device-side twin of fac_cin_pad (include/facodec_hip.h): packed weights have zero rows up to a
// multiple of 48 input channels.
__host__ __device__ constexpr int cin_pad_dev(int c) { return ((c + 47) / 48) * 48; }

// sin(y)^2 to ~1 ulp of sin: Cody-Waite reduction by pi/2 (3 constants, exact products via fma for
// |k| < 2^13) + the Cephes single-precision minimax polynomials on [-pi/4, pi/4]; the quadrant only
// decides WHICH polynomial is squared (sign drops out).  Arguments beyond +-4096 take libm's sinf.
// The rare paths are kept OUT of line on purpose: libm's sinf / tanhf / log1pf bodies are hundreds of
// instructions each, and inlining them at every Snake site made the staging loop of the conv kernel
// overflow the instruction cache (staging then ran at ~20k cycles per chunk -- profiles/ ablation).
__device__ __attribute__((noinline)) float sin_sq_slow(float y) {
  const float s = sinf(y);
  return __fmul_rn(s, s);
}

__device__ __forceinline__ float sin_sq(float y) {
  if (__builtin_expect(fabsf(y) > 4096.0f, 0)) return sin_sq_slow(y);
  const float k = rintf(y * 0.63661977236758134308f);
  float r = fmaf(k, -1.5703125f, y);
  r = fmaf(k, -4.837512969970703125e-4f, r);
  r = fmaf(k, -7.54978995489188216e-8f, r);
  const float z = r * r;
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  const float sn = fmaf(ps * z, r, r);
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  const float cs = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));
  const float v = (((int)k) & 1) ? cs : sn;
  return __fmul_rn(v, v);
}

// Snake activation, dac/nn/layers.py:18-24:  x + (alpha + 1e-9)^-1 * sin(alpha*x)^2.
// inv = 1/(alpha+1e-9) is computed with a true division; the multiply and the add stay separate
// roundings (no fma contraction) like the reference expression.
__device__ __forceinline__ float snake_inv(float alpha) { return __fdiv_rn(1.0f, __fadd_rn(alpha, 1e-9f)); }
__device__ __forceinline__ float snake_apply(float x, float alpha, float inv) {
  return __fadd_rn(x, __fmul_rn(inv, sin_sq(__fmul_rn(alpha, x))));
}

// Epilogue activations other than Snake (cold: once per output element of a few small layers).
__device__ __attribute__((noinline)) float apply_act_slow(float v, int act) {
  if (act == FAC_ACT_TANH) return tanhf(v);
  if (act == FAC_ACT_MISH) {
    // x * tanh(softplus(x)); softplus with torch's threshold 20 (modules/style_encoder.py:6-10)
    const float sp = v > 20.f ? v : log1pf(expf(v));
    return v * tanhf(sp);
  }
  if (act == FAC_ACT_LOG_MEL) return (logf(1e-5f + v) + 4.0f) / 4.0f;   // modules/quantize.py:241
  return v;
}

__device__ __forceinline__ float sigmoid_f(float x) { return __fdiv_rn(1.0f, 1.0f + expf(-x)); }

// Index into x (length T) of position `t` of the reflect-padded signal, or -1 where the value
// is zero.  Text = max(T, max_pad+1) is the length of pad1d's temporary zero extension
// (dac/model/encodec.py:96-113): for T > pad it is plain reflection (-j -> j, T-1+j -> T-1-j).
__device__ __forceinline__ int reflect_index(int t, int T, int Text) {
  int j;
  if (t < 0) j = -t;
  else if (t < Text) j = t;
  else j = 2 * (Text - 1) - t;
  return (j >= 0 && j < T) ? j : -1;
}

}  // namespace fac

// Issue priorities (s_setprio) of the two wave roles of the split-bf16 kernels (conv1d_bsplit / bsplit2 / gemm_split / wgrad k-major):
// MFMA waves and staging waves share a SIMD; the arbiter picks the higher priority when both have an instruction ready.
#ifndef FAC_PRIO_STAGE
#define FAC_PRIO_STAGE 3
#endif
#ifndef FAC_PRIO_MFMA
#define FAC_PRIO_MFMA 0
#endif

