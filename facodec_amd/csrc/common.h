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

// Snake activation, dac/nn/layers.py:18-24:  x + (alpha + 1e-9)^-1 * sin(alpha*x)^2.
// inv = 1/(alpha+1e-9) is computed once per channel with a true division; the multiply and
// the add stay separate roundings (no fma contraction) like the reference expression.
__device__ __forceinline__ float snake_inv(float alpha) { return __fdiv_rn(1.0f, __fadd_rn(alpha, 1e-9f)); }
__device__ __forceinline__ float snake_apply(float x, float alpha, float inv) {
  float s = sinf(__fmul_rn(alpha, x));
  return __fadd_rn(x, __fmul_rn(inv, __fmul_rn(s, s)));
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
