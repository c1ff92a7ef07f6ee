/*
 * oracle/vq_search.c -- plain-C restatement of the reference's nearest-code search.
 * TEST INFRASTRUCTURE ONLY (the checker); never linked into or called by the product path.
 *
 * Follows decode_latents, /root/reference/dac/nn/quantize.py:78-94 (textually identical in
 * quantize/fvq.py:101-116):
 *     e  = F.normalize(latents)      row / max(||row||_2, 1e-12)            :83
 *     c  = F.normalize(codebook)                                             :84
 *     d  = e.pow(2).sum(1) - 2*e @ c.T + c.pow(2).sum(1).T                   :87-91
 *     idx = (-d).max(1)[1]           first index on ties (torch CPU)        :92
 * Pinned by tests/test_oracle_c.py against tests/golden/vq_kat.npz (answers of the real reference:
 * duplicate rows, equal-direction rows, zero latent, 262 144-vector sweep; 0 mismatches).
 * Single-threaded scalar code, sequential sums, true IEEE division.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define CD 8

static float norm_sq(const float* v) {
  float s = 0.f;
  for (int d = 0; d < CD; ++d) s = s + v[d] * v[d];
  return s;
}

/* latents (N, 8) row-major, codebook (K, 8) -> idx (N).  Returns 0, or -1 on allocation failure. */
int oracle_vq_search(const float* latents, const float* codebook, int64_t* idx, int64_t N, int K) {
  float* cn = (float*)malloc((size_t)K * (CD + 1) * sizeof(float));
  if (!cn) return -1;
  float* cc = cn + (size_t)K * CD;
  for (int k = 0; k < K; ++k) {
    const float* r = codebook + (size_t)k * CD;
    float n = sqrtf(norm_sq(r));
    if (n < 1e-12f) n = 1e-12f;
    for (int d = 0; d < CD; ++d) cn[(size_t)k * CD + d] = r[d] / n;
    cc[k] = norm_sq(cn + (size_t)k * CD);
  }
  for (int64_t i = 0; i < N; ++i) {
    const float* z = latents + (size_t)i * CD;
    float e[CD];
    float n = sqrtf(norm_sq(z));
    if (n < 1e-12f) n = 1e-12f;
    for (int d = 0; d < CD; ++d) e[d] = z[d] / n;
    const float ee = norm_sq(e);
    float best = -INFINITY;
    int best_k = 0;
    for (int k = 0; k < K; ++k) {
      const float* c = cn + (size_t)k * CD;
      float dot = 0.f;
      for (int d = 0; d < CD; ++d) dot = dot + (2.0f * e[d]) * c[d];
      const float dist = (ee - dot) + cc[k];
      if (-dist > best) {
        best = -dist;
        best_k = k;
      }
    }
    idx[i] = best_k;
  }
  free(cn);
  return 0;
}
