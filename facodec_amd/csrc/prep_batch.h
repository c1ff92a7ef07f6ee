// Batched weight preparation (round 6): the per-forward weight-norm scales and weight packs of a whole module tree as a few launches.
// The reference re-materialises w = g * v / ||v|| in every forward (dac/model/encodec.py:42-51, torch weight_norm's pre-forward
// hook); so does this library -- one small launch per tensor and layout: ~290 launches (2.7 ms) of a 54 ms forward at configs[1],
// ~1 700 launches of a train step.  Between fac_prep_begin() and fac_prep_end() the calling thread's fac_wn_scale / fac_pack_* /
// fac_flip_transpose_w calls are RECORDED instead of launched (same argument checks, same arithmetic: the batch kernels call the very
// device functions the single launches call); fac_prep_replay(plan) then runs every recorded job again from the tensors' CURRENT
// contents: one launch per (phase, translation unit), phases in order (0: scales, 1: packs that read them, 2: packs of packs).
#pragma once
#include "common.h"

namespace fac {

struct PrepJob {            // 96 bytes; read through the scalar cache (one job per workgroup)
  const void* a;            // v
  const void* b;            // scale / g
  void* out;
  long long n;              // elements of the grid-stride loop (kind-specific)
  long long l[3];
  int kind;
  int nblocks;              // virtual grid of this job
  int i[8];
};

enum PrepUnit { PU_PACK = 0, PU_BSPLIT, PU_GSPLIT, PU_BSPLIT2, PU_BWD, PU_COUNT };

enum PrepKind {
  PK_WN_SCALE = 0, PK_CONV, PK_CONVTR, PK_CONVTR_ROWS, PK_FLIP_T,      // pack.hip
  PK_CONV_SPLIT,                                                        // conv1d_bsplit.hip
  PK_GEMM_SPLIT,                                                        // conv1d_gemm_split.hip
  PK_CONV_SPLIT2,                                                       // conv1d_bsplit2.hip
  PK_CONV_BWD                                                           // conv1d_bwd.hip
};

bool prep_recording();                      // this thread is between fac_prep_begin and fac_prep_end
int prep_record(int unit, const PrepJob& j);

// one per translation unit with recordable kernels: launches its batch kernel over `total` virtual workgroups
int prep_launch_pack(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s);
int prep_launch_bsplit(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s);
int prep_launch_gsplit(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s);
int prep_launch_bsplit2(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s);
int prep_launch_bwd(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s);

inline int prep_blocks(long long n) { return (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535); }

// job of virtual workgroup b: the last j with first[j] <= b (first[0] = 0)
__device__ __forceinline__ int prep_find_job(const int* __restrict__ first, int njobs, int b) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

}  // namespace fac
