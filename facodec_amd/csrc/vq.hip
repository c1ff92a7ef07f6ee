// K7: factorized vector quantization step.
// Reference: VectorQuantize.forward / decode_latents dac/nn/quantize.py:34-94, the residual loop of
// ResidualVectorQuantize.forward :173-193; the search is textually the same in the (dead) named
// variant quantize/fvq.py:101-116.
//
// HBM-bound (reads the (B,D,T) latent once, writes residual / accumulated output once); the
// 1024 x 8 codebook is normalised into LDS (32 KB + 4 KB of squared norms) by every workgroup,
// the search is a per-lane scan (lane = time step, the four waves split the codebook) with a
// strict '>' first-index arg-max and a fixed-order cross-wave combine, so ties resolve to the
// lowest index exactly like torch.max on CPU.
//
// Arithmetic follows the reference expression order:
//   e = z_e / max(sqrt(sum z_e^2), 1e-12)               (F.normalize, true division)
//   dist_k = (sum e^2 - (2e).c~_k) + sum c~_k^2 ; idx = first argmax(-dist)
//   z_q = raw codebook row ; z_st = z_e + (z_q - z_e) ; out = W_out z_st + b_out
#include "common.h"

namespace fac {

constexpr int VQ_TT = 64;   // time steps per workgroup
constexpr int VQ_CD = 8;    // codebook_dim (modules/commons.py:303)

struct VqArgs {
  float* residual;
  const float* z_in;
  float* zq_acc;
  float* zq_out;
  const float* w_in;
  const float* b_in;
  const float* codebook;
  const float* w_out;
  const float* w_out_scale;
  const float* b_out;
  const float* mask;
  long long* codes;
  float* z_e;
  float* loss_part;
  long long codes_bs;
  int B, D, T, Kc;
};

__device__ __forceinline__ float row_norm_sq(const float* v) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < VQ_CD; ++d) s = __fadd_rn(s, __fmul_rn(v[d], v[d]));
  return s;
}

// Normalise the codebook into LDS: cbn[k][8] and cc[k] = sum cbn[k]^2.
__device__ __forceinline__ void load_codebook(const float* __restrict__ cb, float* cbn, float* cc,
                                              int Kc, int tid, int nthreads) {
  for (int k = tid; k < Kc; k += nthreads) {
    float v[VQ_CD];
    const float4 lo = *reinterpret_cast<const float4*>(cb + (long long)k * VQ_CD);
    const float4 hi = *reinterpret_cast<const float4*>(cb + (long long)k * VQ_CD + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
    v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    const float nrm = fmaxf(sqrtf(row_norm_sq(v)), 1e-12f);
#pragma unroll
    for (int d = 0; d < VQ_CD; ++d) v[d] = __fdiv_rn(v[d], nrm);
#pragma unroll
    for (int d = 0; d < VQ_CD; ++d) cbn[k * VQ_CD + d] = v[d];
    cc[k] = row_norm_sq(v);
  }
}

// Scan codes [k_begin, k_end) for the normalised query e; returns best (-dist) and index.
__device__ __forceinline__ void scan_codes(const float* e, float ee, const float* cbn,
                                           const float* cc, int k_begin, int k_end, float& best,
                                           int& best_k) {
  float e2[VQ_CD];
#pragma unroll
  for (int d = 0; d < VQ_CD; ++d) e2[d] = __fmul_rn(2.0f, e[d]);
  best = -INFINITY;
  best_k = k_begin;
  for (int k = k_begin; k < k_end; ++k) {
    const float4 lo = *reinterpret_cast<const float4*>(cbn + k * VQ_CD);
    const float4 hi = *reinterpret_cast<const float4*>(cbn + k * VQ_CD + 4);
    float dot = __fmul_rn(e2[0], lo.x);
    dot = fmaf(e2[1], lo.y, dot);
    dot = fmaf(e2[2], lo.z, dot);
    dot = fmaf(e2[3], lo.w, dot);
    dot = fmaf(e2[4], hi.x, dot);
    dot = fmaf(e2[5], hi.y, dot);
    dot = fmaf(e2[6], hi.z, dot);
    dot = fmaf(e2[7], hi.w, dot);
    const float dist = __fadd_rn(__fsub_rn(ee, dot), cc[k]);
    const float neg = -dist;
    if (neg > best) {
      best = neg;
      best_k = k;
    }
  }
}

// 16 time steps per workgroup (round 5; 64 before: 96 workgroups at B = 32 x 160 frames, 5/8 of the chip idle and every wave a
// chain of 256 dependent-looking loads).  256 threads = 16 time steps x 16 groups:
//   in-proj   wave = channel quarter (the SAME four sequential FMA chains per (t, d) as before -- z_e is bit-identical), lane =
//             (d pair, t): one x load (64 B per 16 lanes) + one 8-byte weight read from LDS + 2 FMAs per channel, 32 channels of
//             loads in flight per wave;
//   search    group g scans codes [64 g, 64 g + 64) with strict '>', then a fixed ascending combine: first maximum wins;
//   out-proj  group g owns channels g, g + 16, ...: the same per-element expressions as before.
// The 1-D grid is decoded so that an XCD gets a contiguous range of (clip, tile) pairs: the tiles of a clip share the 128-byte
// lines of its rows, and consecutive workgroup ids land on different XCDs (each with its own L2).
constexpr int VT = 16;      // time steps per workgroup of vq_fwd_kernel
constexpr int VG = 16;      // thread groups per time step

__global__ __launch_bounds__(256) void vq_fwd_kernel(VqArgs a, int n_tiles, int per_xcd) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cbn = sm;                              // [Kc][8]
  float* cc = cbn + a.Kc * VQ_CD;               // [Kc]
  float* part = cc + ((a.Kc + 3) & ~3);         // [4][8][16]; everything behind cc stays 16-byte aligned for any codebook size
  float* zes = part + 4 * VQ_CD * VT;           // [8][16]  z_e
  float* bestv = zes + VQ_CD * VT;              // [16 groups][16]
  int* bestk = reinterpret_cast<int*>(bestv + VG * VT);  // [16][16]
  float* wsm = reinterpret_cast<float*>(bestk + VG * VT);   // [D][8] in-proj weights

  const int tid = threadIdx.x, tl = tid & (VT - 1), g = tid >> 4, wave = tid >> 6;
  // XCD-aware decode: workgroup id i runs on XCD i % 8
  const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || logical >= a.B * n_tiles) return;
  const int b = logical / n_tiles;
  const int tile = logical - b * n_tiles;
  const int t = tile * VT + tl;
  const bool tv = t < a.T;
  const long long bofs = (long long)b * a.D * a.T;

  load_codebook(a.codebook, cbn, cc, a.Kc, tid, 256);
  for (int c = tid; c < a.D; c += 256) {
    const float4 lo = *reinterpret_cast<const float4*>(a.w_in + (long long)c * 32);
    const float4 hi = *reinterpret_cast<const float4*>(a.w_in + (long long)c * 32 + 4);
    *reinterpret_cast<float4*>(wsm + c * VQ_CD) = lo;
    *reinterpret_cast<float4*>(wsm + c * VQ_CD + 4) = hi;
  }
  __syncthreads();

  // ---- in_proj: wave = quarter of the D input channels, thread = (t, two of the eight output dims)
  {
    const int dq = g & 3;
    const int cper = (a.D + 3) / 4;
    const int c_begin = wave * cper;
    const int c_end = min(a.D, c_begin + cper);
    const float* zp = a.z_in + bofs + (tv ? t : 0);
    float s0 = 0.f, s1 = 0.f;
    constexpr int UN = 32;
    int c = c_begin;
    for (; c + UN <= c_end; c += UN) {
      float xv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) xv[u] = zp[(long long)(c + u) * a.T];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const float2 w = *reinterpret_cast<const float2*>(wsm + (c + u) * VQ_CD + 2 * dq);
        const float x = tv ? xv[u] : 0.f;
        s0 = fmaf(w.x, x, s0);
        s1 = fmaf(w.y, x, s1);
      }
    }
    for (; c < c_end; ++c) {
      const float2 w = *reinterpret_cast<const float2*>(wsm + c * VQ_CD + 2 * dq);
      const float x = tv ? zp[(long long)c * a.T] : 0.f;
      s0 = fmaf(w.x, x, s0);
      s1 = fmaf(w.y, x, s1);
    }
    part[(wave * VQ_CD + 2 * dq) * VT + tl] = s0;
    part[(wave * VQ_CD + 2 * dq + 1) * VT + tl] = s1;
  }
  __syncthreads();
  if (tid < VQ_CD * VT) {
    const int d = tid >> 4;
    float v = (part[(0 * VQ_CD + d) * VT + tl] + part[(1 * VQ_CD + d) * VT + tl]) +
              (part[(2 * VQ_CD + d) * VT + tl] + part[(3 * VQ_CD + d) * VT + tl]);
    v = __fadd_rn(v, a.b_in[d]);
    zes[d * VT + tl] = v;
    if (a.z_e && tv) a.z_e[((long long)b * VQ_CD + d) * a.T + t] = v;
  }
  __syncthreads();

  // ---- normalise + search (every group normalises its time step's query; cheap)
  float ze[VQ_CD], e[VQ_CD];
#pragma unroll
  for (int d = 0; d < VQ_CD; ++d) ze[d] = zes[d * VT + tl];
  {
    const float nrm = fmaxf(sqrtf(row_norm_sq(ze)), 1e-12f);
#pragma unroll
    for (int d = 0; d < VQ_CD; ++d) e[d] = __fdiv_rn(ze[d], nrm);
  }
  const float ee = row_norm_sq(e);
  {
    const int kper = (a.Kc + VG - 1) / VG;
    const int k_begin = min(a.Kc, g * kper);
    const int k_end = min(a.Kc, k_begin + kper);
    float bv;
    int bk;
    scan_codes(e, ee, cbn, cc, k_begin, k_end, bv, bk);
    bestv[g * VT + tl] = bv;              // an empty range leaves -inf: never chosen by the strict '>' below
    bestk[g * VT + tl] = bk;
  }
  __syncthreads();
  int idx = bestk[tl];
  {
    float bv = bestv[tl];
#pragma unroll
    for (int w = 1; w < VG; ++w) {
      const float v = bestv[w * VT + tl];
      if (v > bv) {
        bv = v;
        idx = bestk[w * VT + tl];
      }
    }
  }

  // ---- gather raw code, straight-through value, loss partial
  float zst[VQ_CD];
  float lsum = 0.f;
  {
    const float* cr = a.codebook + (long long)idx * VQ_CD;
#pragma unroll
    for (int d = 0; d < VQ_CD; ++d) {
      const float zq = cr[d];
      const float df = __fsub_rn(ze[d], zq);
      lsum = __fadd_rn(lsum, __fmul_rn(df, df));
      zst[d] = __fadd_rn(ze[d], __fsub_rn(zq, ze[d]));
    }
  }
  if (g == 0) {                           // lanes 0..15 of wave 0
    if (tv) a.codes[(long long)b * a.codes_bs + t] = idx;
    float l = tv ? lsum : 0.f;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) l += __shfl_down(l, o, 16);
    if (tl == 0 && a.loss_part) a.loss_part[(long long)b * n_tiles + tile] = l;
  }

  // ---- out_proj + residual bookkeeping: group g handles channels g, g+16, ...  The loads of 8 channels
  // are issued together before their math (the accumulator / residual may alias the input, so the
  // compiler would otherwise serialise one memory round trip per channel).
  if (tv) {
    const float mk = a.mask ? a.mask[b] : 1.0f;
    constexpr int UB = 8;
    for (int c0 = g; c0 < a.D; c0 += VG * UB) {
      float zin[UB], zacc[UB], bo[UB], sc[UB], wv[UB][VQ_CD];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int c = c0 + VG * u;
        const bool cv = c < a.D;
        const int cc = cv ? c : a.D - 1;
        const long long off = bofs + (long long)cc * a.T + t;
        zin[u] = a.residual ? a.z_in[off] : 0.f;
        zacc[u] = a.zq_acc ? a.zq_acc[off] : 0.f;
        bo[u] = a.b_out[cc];
        sc[u] = a.w_out_scale ? a.w_out_scale[cc] : 1.0f;
        const float4 lo = *reinterpret_cast<const float4*>(a.w_out + (long long)cc * VQ_CD);
        const float4 hi = *reinterpret_cast<const float4*>(a.w_out + (long long)cc * VQ_CD + 4);
        wv[u][0] = lo.x; wv[u][1] = lo.y; wv[u][2] = lo.z; wv[u][3] = lo.w;
        wv[u][4] = hi.x; wv[u][5] = hi.y; wv[u][6] = hi.z; wv[u][7] = hi.w;
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int c = c0 + VG * u;
        if (c >= a.D) continue;
        float o = __fmul_rn(__fmul_rn(wv[u][0], sc[u]), zst[0]);
#pragma unroll
        for (int d = 1; d < VQ_CD; ++d) o = fmaf(__fmul_rn(wv[u][d], sc[u]), zst[d], o);
        o = __fadd_rn(o, bo[u]);
        const long long off = bofs + (long long)c * a.T + t;
        if (a.zq_out) a.zq_out[off] = o;
        if (a.zq_acc) a.zq_acc[off] = __fadd_rn(zacc[u], __fmul_rn(o, mk));
        if (a.residual) a.residual[off] = __fsub_rn(zin[u], o);
      }
    }
  }
}

// Few time steps (streaming hops: T = 1-2 frames): the lane-per-time-step kernel above would leave one
// lane per wave busy and pay ~100 dependent memory round trips.  Here one workgroup owns one (b, t) and its
// 256 lanes spread over channels / codes, with the SAME arithmetic per value: the in-proj FMA chains run
// over the same channel quarters in the same order (from LDS), the code scan keeps "first maximum wins",
// the out-proj is per channel -- results are bit-identical to vq_fwd_kernel.
__global__ __launch_bounds__(256) void vq_fwd_small_t_kernel(VqArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cbn = sm;                              // [Kc][8]
  float* cc = cbn + a.Kc * VQ_CD;               // [Kc]
  float* xs = cc + a.Kc;                        // [D]
  float* wsm = xs + a.D;                        // [D][8]
  float* part = wsm + a.D * VQ_CD;              // [4][8]
  float* zes = part + 4 * VQ_CD;                // [8]
  float* bestv = zes + VQ_CD;                   // [256]
  int* bestk = reinterpret_cast<int*>(bestv + 256);
  const int tid = threadIdx.x;
  const int t = blockIdx.x, b = blockIdx.y;
  const long long bofs = (long long)b * a.D * a.T;

  load_codebook(a.codebook, cbn, cc, a.Kc, tid, 256);
  for (int c = tid; c < a.D; c += 256) {
    xs[c] = a.z_in[bofs + (long long)c * a.T + t];
    const float4 lo = *reinterpret_cast<const float4*>(a.w_in + (long long)c * 32);
    const float4 hi = *reinterpret_cast<const float4*>(a.w_in + (long long)c * 32 + 4);
    *reinterpret_cast<float4*>(wsm + c * VQ_CD) = lo;
    *reinterpret_cast<float4*>(wsm + c * VQ_CD + 4) = hi;
  }
  __syncthreads();
  if (tid < 32) {   // (quarter q, dim d): the chain of vq_fwd_kernel's wave q
    const int q = tid >> 3, d = tid & 7;
    const int cper = (a.D + 3) / 4;
    const int c_begin = q * cper, c_end = min(a.D, c_begin + cper);
    float sacc = 0.f;
#pragma unroll 16
    for (int c = c_begin; c < c_end; ++c) sacc = fmaf(wsm[c * VQ_CD + d], xs[c], sacc);
    part[q * VQ_CD + d] = sacc;
  }
  __syncthreads();
  if (tid < VQ_CD) {
    float v = (part[0 * VQ_CD + tid] + part[1 * VQ_CD + tid]) + (part[2 * VQ_CD + tid] + part[3 * VQ_CD + tid]);
    v = __fadd_rn(v, a.b_in[tid]);
    zes[tid] = v;
    if (a.z_e) a.z_e[((long long)b * VQ_CD + tid) * a.T + t] = v;
  }
  __syncthreads();
  float ze[VQ_CD], e[VQ_CD];
#pragma unroll
  for (int d = 0; d < VQ_CD; ++d) ze[d] = zes[d];
  {
    const float nrm = fmaxf(sqrtf(row_norm_sq(ze)), 1e-12f);
#pragma unroll
    for (int d = 0; d < VQ_CD; ++d) e[d] = __fdiv_rn(ze[d], nrm);
  }
  const float ee = row_norm_sq(e);
  {
    const int kper = (a.Kc + 255) / 256;
    const int k_begin = min(a.Kc, tid * kper), k_end = min(a.Kc, k_begin + kper);
    float bv;
    int bk;
    scan_codes(e, ee, cbn, cc, k_begin, k_end, bv, bk);
    bestv[tid] = k_begin < k_end ? bv : -INFINITY;
    bestk[tid] = bk;
  }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {   // ties keep the lower index (first maximum wins)
    if (tid < off) {
      if (bestv[tid + off] > bestv[tid] || (bestv[tid + off] == bestv[tid] && bestk[tid + off] < bestk[tid])) {
        bestv[tid] = bestv[tid + off];
        bestk[tid] = bestk[tid + off];
      }
    }
    __syncthreads();
  }
  const int idx = bestk[0];
  float zst[VQ_CD];
  {
    const float* cr = a.codebook + (long long)idx * VQ_CD;
#pragma unroll
    for (int d = 0; d < VQ_CD; ++d) {
      const float zq = cr[d];
      zst[d] = __fadd_rn(ze[d], __fsub_rn(zq, ze[d]));
    }
  }
  if (tid == 0) a.codes[(long long)b * a.codes_bs + t] = idx;
  const float mk = a.mask ? a.mask[b] : 1.0f;
  for (int c = tid; c < a.D; c += 256) {
    const long long off = bofs + (long long)c * a.T + t;
    const float zin = a.residual ? a.z_in[off] : 0.f;
    const float zacc = a.zq_acc ? a.zq_acc[off] : 0.f;
    const float sc = a.w_out_scale ? a.w_out_scale[c] : 1.0f;
    const float* wr = a.w_out + (long long)c * VQ_CD;
    float o = __fmul_rn(__fmul_rn(wr[0], sc), zst[0]);
#pragma unroll
    for (int d = 1; d < VQ_CD; ++d) o = fmaf(__fmul_rn(wr[d], sc), zst[d], o);
    o = __fadd_rn(o, a.b_out[c]);
    if (a.zq_out) a.zq_out[off] = o;
    if (a.zq_acc) a.zq_acc[off] = __fadd_rn(zacc, __fmul_rn(o, mk));
    if (a.residual) a.residual[off] = __fsub_rn(zin, o);
  }
}

// Search only: latents (N, 8) row-major -> idx.
__global__ __launch_bounds__(256) void vq_search_kernel(const float* __restrict__ lat,
                                                        const float* __restrict__ cb,
                                                        long long* __restrict__ idx_out,
                                                        long long N, int Kc) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cbn = sm;
  float* cc = cbn + Kc * VQ_CD;
  float* bestv = cc + Kc;
  int* bestk = reinterpret_cast<int*>(bestv + 4 * VQ_TT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  load_codebook(cb, cbn, cc, Kc, tid, 256);
  __syncthreads();
  for (long long base = (long long)blockIdx.x * VQ_TT; base < N; base += (long long)gridDim.x * VQ_TT) {
    const long long n = base + lane;
    const bool nv = n < N;
    float ze[VQ_CD], e[VQ_CD];
    if (nv) {
      const float4 lo = *reinterpret_cast<const float4*>(lat + n * VQ_CD);
      const float4 hi = *reinterpret_cast<const float4*>(lat + n * VQ_CD + 4);
      ze[0] = lo.x; ze[1] = lo.y; ze[2] = lo.z; ze[3] = lo.w;
      ze[4] = hi.x; ze[5] = hi.y; ze[6] = hi.z; ze[7] = hi.w;
    } else {
#pragma unroll
      for (int d = 0; d < VQ_CD; ++d) ze[d] = 0.f;
    }
    const float nrm = fmaxf(sqrtf(row_norm_sq(ze)), 1e-12f);
#pragma unroll
    for (int d = 0; d < VQ_CD; ++d) e[d] = __fdiv_rn(ze[d], nrm);
    const float ee = row_norm_sq(e);
    const int kper = (Kc + 3) / 4;
    const int k_begin = wave * kper;
    const int k_end = min(Kc, k_begin + kper);
    float bv;
    int bk;
    scan_codes(e, ee, cbn, cc, k_begin, k_end, bv, bk);
    bestv[wave * VQ_TT + lane] = bv;
    bestk[wave * VQ_TT + lane] = bk;
    __syncthreads();
    if (wave == 0 && nv) {
      int idx = bestk[lane];
      float b0 = bestv[lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float v = bestv[w * VQ_TT + lane];
        if (v > b0) {
          b0 = v;
          idx = bestk[w * VQ_TT + lane];
        }
      }
      idx_out[n] = idx;
    }
    __syncthreads();
  }
}

}  // namespace fac

extern "C" int fac_vq_loss_tiles(int T) { return (T + fac::VT - 1) / fac::VT; }

extern "C" int fac_vq_fwd(const fac_vq_desc* d, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(d && d->z_in && d->w_in && d->b_in && d->codebook && d->w_out && d->b_out && d->codes,
              "vq_fwd: null pointer");
  FAC_REQUIRE(d->B > 0 && d->D > 0 && d->T > 0 && d->Kc > 0, "vq_fwd: bad shape");
  FAC_REQUIRE(d->B <= 65535, "vq_fwd: B too large");
  VqArgs a;
  a.residual = d->residual; a.z_in = d->z_in; a.zq_acc = d->zq_acc; a.zq_out = d->zq_out;
  a.w_in = d->w_in; a.b_in = d->b_in; a.codebook = d->codebook; a.w_out = d->w_out; a.w_out_scale = d->w_out_scale;
  a.b_out = d->b_out; a.mask = d->mask; a.codes = (long long*)d->codes; a.z_e = d->z_e;
  a.loss_part = d->loss_part; a.codes_bs = d->codes_bs;
  a.B = d->B; a.D = d->D; a.T = d->T; a.Kc = d->Kc;
  const size_t lds = ((size_t)d->Kc * VQ_CD + ((d->Kc + 3) & ~3) + 4 * VQ_CD * VT + VQ_CD * VT + 2 * VG * VT + (size_t)d->D * VQ_CD) * 4;
  FAC_REQUIRE(lds <= 160 * 1024, "vq_fwd: codebook of %d entries + %d in-proj rows do not fit LDS", d->Kc, d->D);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_fwd_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (d->T <= 8 && !d->loss_part && d->D <= 4096) {   // streaming hops: one workgroup per (b, t)
    const size_t lds_s = ((size_t)d->Kc * (VQ_CD + 1) + (size_t)d->D * (VQ_CD + 1) + 4 * VQ_CD + VQ_CD + 512) * 4;
    if (lds_s <= 160 * 1024) {
      static bool attr_s = false;
      if (!attr_s) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_fwd_small_t_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_s = true;
      }
      hipLaunchKernelGGL(vq_fwd_small_t_kernel, dim3(d->T, d->B), dim3(256), lds_s, (hipStream_t)stream, a);
      return check_launch("vq_fwd(small T)");
    }
  }
  const int n_tiles = (d->T + VT - 1) / VT;
  const long long total = (long long)d->B * n_tiles;
  FAC_REQUIRE(total <= (1ll << 28), "vq_fwd: too many tiles");
  const int per_xcd = (int)((total + 7) / 8);
  hipLaunchKernelGGL(vq_fwd_kernel, dim3(8 * per_xcd), dim3(256), lds, (hipStream_t)stream, a, n_tiles, per_xcd);
  return check_launch("vq_fwd");
}

extern "C" int fac_vq_search(const float* latents, const float* codebook, int64_t* idx, int64_t N,
                             int Kc, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(latents && codebook && idx && N > 0 && Kc > 0, "vq_search: bad arguments");
  const size_t lds = ((size_t)Kc * (VQ_CD + 1) + 8 * VQ_TT) * 4;
  FAC_REQUIRE(lds <= 160 * 1024, "vq_search: codebook of %d entries does not fit LDS", Kc);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_search_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  long long tiles = (N + VQ_TT - 1) / VQ_TT;
  int grid = (int)(tiles < 2048 ? tiles : 2048);
  hipLaunchKernelGGL(vq_search_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, latents,
                     codebook, (long long*)idx, (long long)N, Kc);
  return check_launch("vq_search");
}
