// Probe (not part of the library): cost of ds_read_b128 on gfx950 when the per-lane address is not 16-byte aligned,
// for the access patterns a time-contracting weight-gradient GEMM needs (B fragment = 8 consecutive bf16 time steps of
// one (input channel, tap) column at element offset t + k*dilation).  Reports clocks per ds_read_b128 per wave, one
// wave per SIMD (4 waves per workgroup) all reading, 1 workgroup per CU.
//   build: hipcc --offload-arch=gfx950 -O3 lds_unaligned_probe.hip -o lds_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U4 { u32x4 v; };

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int PITCH = 144;   // bytes per row: 64 bf16 + 8 pad (conflict-free for aligned fragment reads)
constexpr int NREAD = 32;

// mode: 0 aligned rows (lane&31 = row)            1..7: same + (mode * 2) bytes
//       10 + d: 7-tap columns, J = lane&31 -> (row J / 7, tap J % 7), offset tap * d elements   (d = 1, 3, 9)
//       20: dword-aligned but not 16-B aligned (+4)    21: +8
__global__ __launch_bounds__(256) void probe(unsigned* out, long long* clk, int mode, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, kq = lane >> 5;
  int base;
  if (mode < 10) base = l31 * PITCH + kq * 16 + mode * 2;
  else if (mode < 20) { const int d = mode - 10; base = (l31 / 7) * PITCH + ((l31 % 7) * d + kq * 8) * 2; }
  else base = l31 * PITCH + kq * 16 + (mode == 20 ? 4 : 8);
  u32x4 acc = {0, 0, 0, 0};
  const long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < NREAD; ++r) {
      const U4* p = reinterpret_cast<const U4*>(sm + base + r * 32 * PITCH % (96 * 1024) + (it & 1) * 32);
      u32x4 v = p->v;
      acc ^= v;
    }
  }
  long long c1 = clock64();
  (void)t0;
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

int main() {
  unsigned* out;
  long long* clk;
  const int nwg = 256;
  CK(hipMalloc(&out, nwg * 256 * 4));
  CK(hipMalloc(&clk, nwg * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int iters = 2000;
  const int modes[] = {0, 1, 2, 3, 4, 20, 21, 11, 13, 19};
  for (int m : modes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 112 * 1024, 0, out, clk, m, iters);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    long long h[nwg];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < nwg; ++i) s += (double)h[i];
    printf("mode %2d: %.2f counter ticks, %.3f ns per ds_read_b128 per wave (4 waves/CU reading; kernel %.3f ms)\n", m,
           s / nwg / ((double)iters * NREAD), 1e6 * ms / ((double)iters * NREAD), ms);
  }
  return 0;
}
