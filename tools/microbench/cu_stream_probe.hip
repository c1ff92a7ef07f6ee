// Probe (not part of the library): how fast ONE workgroup per CU can pull bytes in, by path and by number of waves:
//   mode 0: global_load_dwordx4 -> VGPR        mode 1: LDS-DMA (global_load_lds, 16 B per lane)
// from a working set that is L2-resident (`small`: every CU re-reads the same 1 MiB) or streams from HBM (`big`: 4 GiB,
// each CU its own range).  Reports bytes / clock / CU at the measured wall time (2.4 GHz nominal) and the chip-wide TB/s.
// Answers whether a CU's load path (outstanding-miss capacity x latency) caps the staging rate of the conv kernels.
//   build: hipcc --offload-arch=gfx950 -O3 cu_stream_probe.hip -o cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;

template <int U>
__global__ void probe(const float4* __restrict__ src, float* out, long long per_cu_f4, long long span_f4, int mode, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, nth = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63;
  const float4* base = src + (long long)blockIdx.x * per_cu_f4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  long long pos = tid;
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = base[((pos + (long long)u * nth) & (span_f4 - 1))];
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].w; }
    } else if (mode == 2) {
      // conv-like: this CU's tile `it / 6` covers 1 KiB of each of 192 rows (pitch 96 KB); a trip moves U x waves rows
      const int nw = nth >> 6;
      const long long tile = (long long)blockIdx.x * 64 + it / (192 / (U * nw) > 0 ? 192 / (U * nw) : 1);
      const int rbase = (it % (192 / (U * nw) > 0 ? 192 / (U * nw) : 1)) * U * nw;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long row = rbase + u * nw + wave;
        const float4* p = src + (((tile / 94) * 192 + row) * 6000 + (tile % 94) * 64 + lane);
        __builtin_amdgcn_global_load_lds((glb_void_t*)p, (lds_void_t*)(sm + (wave * U + u) * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4* p = base + ((pos + (long long)u * nth) & (span_f4 - 1));
        __builtin_amdgcn_global_load_lds((glb_void_t*)p, (lds_void_t*)(sm + (wave * U + u) * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    }
    pos += (long long)U * nth;
  }
  if (mode == 1) acc.x = reinterpret_cast<float*>(sm)[tid];
  out[blockIdx.x * 1024 + tid] = acc.x + acc.y;
}

// Mixed traffic: out1 = a + b, out2 = a * b (two read streams, two write streams, float4 per lane, grid-stride) -- what an
// ideal elementwise pass over the 1x1 conv's four tensors (input, residual, y, y2) would reach.
__global__ void rw_probe(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o1, float4* __restrict__ o2,
                         long long n, int nread, int nwrite) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float4 x = a[i];
    float4 y = nread > 1 ? b[i] : x;
    float4 s = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    if (nwrite > 0) o1[i] = s;
    if (nwrite > 1) o2[i] = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
    if (nwrite == 0 && s.x == 123.456f) o1[i] = s;
  }
}

int main() {
  {
    const long long n = (590ll << 20) / 16;      // 590 MB per stream, the C = 192, T = 24 000, B = 32 tensors
    float4 *a, *b, *o1, *o2;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&o1, n * 16)); CK(hipMalloc(&o2, n * 16));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
    for (int cfg = 0; cfg < 5; ++cfg) {
      const int nr = cfg == 0 ? 1 : cfg == 1 ? 2 : cfg == 2 ? 1 : cfg == 3 ? 2 : 2;
      const int nw = cfg == 0 ? 0 : cfg == 1 ? 0 : cfg == 2 ? 1 : cfg == 3 ? 1 : 2;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rw_probe, dim3(256 * 16), dim3(256), 0, 0, a, b, o1, o2, n, nr, nw);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
      }
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("elementwise %d read + %d write streams of 590 MB: %7.3f ms  %5.2f TB/s\n", nr, nw, ms, (double)(nr + nw) * n * 16 / ms / 1e9);
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(o1)); CK(hipFree(o2));
  }
  const long long big = 4ll << 30;
  float4* src;
  float* out;
  CK(hipMalloc(&src, big));
  CK(hipMemset(src, 0, big));
  CK(hipMalloc(&out, 256 * 1024 * 4));
  auto k = probe<8>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  for (int set = 0; set < 2; ++set)
    for (int mode = 0; mode < 2; ++mode)
      for (int nw : {4, 8, 12, 16}) {
        const long long per_cu = set == 0 ? 0 : (big / 16) / 256;          // small: all CUs read the same range
        const long long span = set == 0 ? (1ll << 20) / 16 : (big / 16) / 256;
        const int nth = nw * 64;
        const long long bytes_per_cu = set == 0 ? (64ll << 20) : (16ll << 20);
        const int iters = (int)(bytes_per_cu / (8ll * nth * 16));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(k, dim3(256), dim3(nth), 128 * 1024, 0, src, out, per_cu, span, mode, iters);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double tot = (double)iters * 8 * nth * 16 * 256;
        printf("%s  %s  waves/CU %2d : %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", set == 0 ? "L2-resident 1 MiB" : "HBM stream       ",
               mode == 0 ? "global_load->VGPR" : "LDS-DMA          ", nw, ms, tot / ms / 1e9, tot / 256 / (ms * 1e-3 * 2.4e9));
      }
  for (int nw : {4, 8}) {
    // 256 CUs x 64 tiles x 192 KiB, tiles walk 32 x 192 rows of 24000 floats (the k = 1 conv's x at C = 192)
    const int nth = nw * 64, iters = 64 * (192 / (8 * nw));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(256), dim3(nth), 128 * 1024, 0, src, out, 0ll, 1ll, 2, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double tot = 256.0 * 64 * 192 * 1024;
    printf("conv-like rows (192 x 1 KiB, pitch 96 KB) LDS-DMA waves/CU %2d : %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU\n", nw, ms, tot / ms / 1e9,
           tot / 256 / (ms * 1e-3 * 2.4e9));
  }
  return 0;
}
