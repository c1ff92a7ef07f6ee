// Probe (not part of the library): which 64-byte-row XOR swizzles are bank-conflict-free on gfx950 for the two access shapes of
// conv1d_gemm_split.hip -- ds_write_b128 by staging lanes that walk consecutive rows (one 16-byte piece g per wave), and
// ds_read_b128 of MFMA fragments (lane = row % 32, piece = lane / 32).  Row r keeps piece p at slot p ^ f(r), f(r) = bit A of r
// | bit B of r << 1 (A = B = -1: no swizzle).  Prints ns per instruction per wave; run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace
// for the conflict ratio of each variant (kernel names carry A, B).
//   build: hipcc --offload-arch=gfx950 -O3 lds_bank_probe.hip -o lds_bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int A, int B>
__device__ __forceinline__ int swz(int r) {
  if (A < 0) return 0;
  return ((r >> A) & 1) | (((r >> B) & 1) << 1);
}

template <int A, int B>
__global__ __launch_bounds__(256) void wr_probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane, g = wave;
  const unsigned addr = (unsigned)(size_t)(sm) + row * 64 + ((g ^ swz<A, B>(row)) * 16);
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int it = 0; it < iters; ++it) {
    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:4096\n\tds_write_b128 %0, %1 offset:8192\n\tds_write_b128 %0, %1 offset:12288"
                 :: "v"(addr), "v"(v) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = reinterpret_cast<float*>(sm)[5];
}

template <int A, int B>
__global__ __launch_bounds__(256) void rd_probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384 / 4; i += 256) reinterpret_cast<float*>(sm)[i] = (float)i;
  __syncthreads();
  const int row = lane & 31, piece = (lane >> 5) + 2 * (wave & 1);
  const unsigned addr = (unsigned)(size_t)(sm) + row * 64 + ((piece ^ swz<A, B>(row)) * 16);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x4 a, b, c, d;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
    acc += a + b + c + d;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == -1.f) out[threadIdx.x] = acc[0];
}

template <int A, int B>
void run(float* out, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms[2];
  for (int k = 0; k < 2; ++k) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (k == 0) hipLaunchKernelGGL((wr_probe<A, B>), dim3(256), dim3(256), 16384, 0, out, iters);
      else hipLaunchKernelGGL((rd_probe<A, B>), dim3(256), dim3(256), 16384, 0, out, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[k], e0, e1));
    }
  }
  // 4 instructions per iteration per wave, 4 waves per workgroup share one CU's LDS
  printf("f = bit%d | bit%d<<1 : ds_write_b128 %.2f ns / instr / CU    ds_read_b128 %.2f ns / instr / CU\n", A, B,
         ms[0] * 1e6 / (iters * 16.0), ms[1] * 1e6 / (iters * 16.0));
}

int main() {
  float* out;
  CK(hipMalloc(&out, 4096));
  const int iters = 20000;
  run<-1, -1>(out, iters);
  run<2, 3>(out, iters);   // what conv1d_gemm_split.hip / conv1d_wgrad_split.hip use
  run<1, 2>(out, iters);
  run<0, 1>(out, iters);
  run<1, 3>(out, iters);
  run<1, 4>(out, iters);
  run<2, 4>(out, iters);
  run<3, 4>(out, iters);
  run<0, 2>(out, iters);
  run<0, 3>(out, iters);
  run<3, 2>(out, iters);
  run<3, 1>(out, iters);
  CK(hipDeviceSynchronize());
  return 0;
}
