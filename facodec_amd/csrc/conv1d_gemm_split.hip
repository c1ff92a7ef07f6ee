// 1- and 2-tap stride-1 convs as a split-bf16 GEMM (fp32-grade: the 3-way operand split of conv1d_bsplit.hip, six
// v_mfma_f32_32x32x16_bf16 per K = 16 step, fp32 accumulate, smallest terms first).
//
// Which layers: the k = 1 convs with many channels (ResidualUnit tails at C = 512 / 768, the LSTM input projections
// 1024 -> 4096 / 1536 -> 6144 run as ONE GEMM over every (t, b), the quantizer-side and predictor-head 1x1 convs at the
// latent rate) and -- with fac_conv_desc.row_phases -- the causal ConvTranspose1d with all output phases as GEMM rows.
// On the fp32 matrix pipe these sat at 55-85 TFLOP/s (half of that pipe's peak, profiles/r02_bench_line.json) because a 1- or
// 2-tap conv re-uses nothing across taps; conv1d_bsplit.hip's stage shape (8-channel groups x taps) degenerates to one MFMA
// step per barrier for them.  Here the contraction runs over 32 input channels per stage the way the weight-gradient GEMM
// (conv1d_wgrad_split.hip, k-major variant) contracts over 32 time steps:
//   tile 128 rows x 128 columns, 4 MFMA waves (64 x 64 = 2 x 2 blocks each) + 4 staging waves;
//   A (weights): pre-split, pre-swizzled bf16 planes in HBM, one contiguous slab per (row tile, 32-channel chunk):
//                [plane][tap][128 rows][64 B] -> LDS by LDS-DMA (flat copy, no VGPR, no VALU);
//   B (inputs):  fp32 (B, C, T) rows; a staging lane owns (column, 8-channel group) units: 8 coalesced loads (consecutive
//                lanes = consecutive columns), 3-way split, three ds_write_b128 into [plane][column][32 channels] rows --
//                K-contiguous per column, so a B fragment of tap k is one aligned ds_read_b128 of row (column + k);
//   both operands use 64-byte rows with the XOR swizzle of the weight-gradient kernel (slot = piece ^ ((row >> 2) & 3)):
//   fragment reads and staging writes are bank-conflict-free without padding;
//   K = 1: columns are the FLATTENED (clip, time) index (no halo), so 160-frame latents fill 128-column tiles exactly and
//          the LSTM projections are plain GEMMs; three LDS stages;   K = 2: per-clip column tiles, one halo row, two stages;
//   epilogue by all eight waves from an fp32 tile in LDS (bias, Snake / activation, residual, second pre-activated output,
//   row_phases interleave), 16-byte stores of contiguous runs.
#include "conv1d_mfma.h"
#include "prep_batch.h"

namespace fac {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int GS_ROWS = 128;                   // output rows (channels, or (channel, phase) pairs) per tile
constexpr int GS_COLS = 128;                   // output columns per tile
constexpr int GS_CI = 32;                      // input channels per stage
constexpr int GS_RB = 64;                      // bytes per LDS row: 32 bf16
constexpr int GS_APL = GS_ROWS * GS_RB;        // one (plane, tap) of the weights: 8 KB
constexpr int GS_XR = 132;                     // staged input rows per plane (128 + K - 1, rounded)
constexpr int GS_BPL = GS_XR * GS_RB;          // one plane of the inputs

__host__ __device__ constexpr int gs_stage_bytes(int K) { return 3 * K * GS_APL + 3 * GS_BPL; }
__host__ __device__ constexpr int gs_stages(int K) { return K == 1 ? 3 : 2; }

__device__ __forceinline__ void gs_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// XOR swizzle of the INPUT planes (round 6).  The weights keep slot = piece ^ ((row >> 2) & 3): they reach the LDS by DMA and are only
// read.  The inputs are also WRITTEN by ds_write_b128 (fp32 inputs, split in the staging waves), and a 16-byte store is serviced in
// groups of 8 consecutive lanes = 8 consecutive rows over a 128-byte bank window (two rows): with (row >> 2) & 3 rows r and r + 2 of
// a group share their banks -- every staging store took two LDS passes (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.13 - 0.27 on this
// kernel, 0.00 on the others: profiles/r06_pmc_train.json).  f = bit 2 | (bit 1 ^ bit 3) << 1 keeps the fragment reads (16-lane groups
// {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} over a 256-byte window, rows shifted by the tap) conflict-free and makes 8 aligned
// consecutive rows hit 8 different 16-byte slots (exhaustive search over the GF(2)-linear maps of the row bits: tools/tune/lds_swizzle_search.py).
#ifdef FAC_GS_OLD_SWZ   // tuning builds: the swizzle of rounds 3 - 5
__device__ __forceinline__ int gs_xswz(int row) { return (row >> 2) & 3; }
#else
__device__ __forceinline__ int gs_xswz(int row) { return ((row >> 2) & 1) | ((((row >> 1) ^ (row >> 3)) & 1) << 1); }
#endif

__device__ __forceinline__ void gs_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Weights (R rows, C_in, K_total taps) given through strides (element (r, ci, k) at v[r*rs + ci*cs + k*ks]) [* row_scale[r]]
// -> [row tile][chunk][plane][tap][row 128][slot 4][8 bf16], slot = piece ^ ((row >> 2) & 3), piece = (ci % 32) / 8.
// Input stride S > 1 (strided conv, K_total <= 2 S): chunk = (32 real channels, input phase p), p fastest; its K = 2 taps are
// k = p and k = S + p (zero when >= K_total) -- the conv over the phase-p sub-signal x[S u + p].
// One thread per (tile, chunk, tap, row, piece): three 16-byte stores.
__device__ __forceinline__ void pack_gemm_split_body(const float* __restrict__ v, long long rs, long long cs, long long ks,
                                                     const float* __restrict__ row_scale, unsigned char* __restrict__ out,
                                                     int R, int C_in, int K, int n_ch, int S, int K_total, long long n, int vb, int vg) {
  for (long long idx = (long long)vb * 256 + threadIdx.x; idx < n; idx += (long long)vg * 256) {
    const int piece = (int)(idx & 3);
    const int row = (int)((idx >> 2) & 127);
    long long r2 = idx >> 9;
    const int k = (int)(r2 % K);
    r2 /= K;
    const int ch = (int)(r2 % n_ch);
    const int tile = (int)(r2 / n_ch);
    const int rg = tile * GS_ROWS + row;
    const float sc = (row_scale != nullptr && rg < R) ? row_scale[rg] : 1.0f;
    bf16x8 h, m, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ci = (ch / S) * GS_CI + piece * 8 + i;
      const int kk = S * k + ch % S;
      float w = 0.f;
      if (rg < R && ci < C_in && kk < K_total) {
        w = v[(long long)rg * rs + (long long)ci * cs + (long long)kk * ks];
        if (row_scale != nullptr) w = __fmul_rn(w, sc);
      }
      __bf16 a, b, c;
      gs_split3(w, a, b, c);
      h[i] = a; m[i] = b; l[i] = c;
    }
    const int slot = piece ^ ((row >> 2) & 3);
    unsigned char* base = out + ((long long)tile * n_ch + ch) * (3 * K * GS_APL) + (long long)k * GS_APL + row * GS_RB + slot * 16;
    *reinterpret_cast<bf16x8*>(base) = h;
    *reinterpret_cast<bf16x8*>(base + (long long)K * GS_APL) = m;
    *reinterpret_cast<bf16x8*>(base + (long long)2 * K * GS_APL) = l;
  }
}

__global__ __launch_bounds__(256) void pack_gemm_split_kernel(const float* __restrict__ v, long long rs, long long cs, long long ks,
                                                              const float* __restrict__ row_scale, unsigned char* __restrict__ out,
                                                              int R, int C_in, int K, int n_ch, int S, int K_total, long long n) {
  pack_gemm_split_body(v, rs, cs, ks, row_scale, out, R, C_in, K, n_ch, S, K_total, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void gsplit_batch_kernel(const PrepJob* __restrict__ jobs, const int* __restrict__ first, int njobs) {
  const int j = prep_find_job(first, njobs, blockIdx.x);
  const PrepJob& J = jobs[j];
  pack_gemm_split_body(static_cast<const float*>(J.a), J.l[0], J.l[1], J.l[2], static_cast<const float*>(J.b),
                       static_cast<unsigned char*>(J.out), J.i[0], J.i[1], J.i[2], J.i[3], J.i[4], J.i[5], J.n, blockIdx.x - first[j],
                       J.nblocks);
}

int prep_launch_gsplit(const PrepJob* jobs, const int* first, int njobs, int total, hipStream_t s) {
  hipLaunchKernelGGL(gsplit_batch_kernel, dim3(total), dim3(256), 0, s, jobs, first, njobs);
  return check_launch("gsplit_batch");
}

template <int K>
__global__ __launch_bounds__(512, 2) void conv1d_gemm_split_kernel(ConvArgs a) {
  constexpr int NST = gs_stages(K);
  constexpr int STAGE = gs_stage_bytes(K);
  constexpr int A_BYTES = 3 * K * GS_APL;
  constexpr int D = NST - 1;                   // weight slabs are requested D stages ahead
  constexpr int NA = 6 * K;                    // DMA instructions per staging lane and stage (24 K blocks of 1 KB over 4 waves)
  constexpr int GS_NU = K == 1 ? 2 : 3;        // (column, 8-channel group) units per staging lane: 4 groups x (128 + K - 1) columns
  constexpr int NBL = GS_NU * 8;               // register loads per staging lane and stage
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool flat = a.gflat != 0;

  // XCD-aware work decode (see conv1d_mfma.h): each XCD walks a contiguous range of (row tile, clip, column tile)
  int n0, row0, b;
  {
    const int n = gridDim.x;
    const int q8 = n >> 3, r8 = n & 7;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int nt = a.n_t_tiles, nb = flat ? 1 : a.B;
    if (a.grt > 0) {
      // Row tile fastest (round 5): the workgroups that are co-resident on an XCD cover ALL row tiles of a few consecutive
      // column tiles, so every input tile is fetched into that XCD's L2 once and then read by every row tile from there.  With
      // the row tile slowest (below) an XCD sits on one row tile -- its weight slab stays in L2 -- and streams the WHOLE input
      // past it: the input leaves HBM / the Infinity Cache once per row tile (counters: 2.0 GB per transposed-conv launch
      // against 0.6 GB algorithmic, profiles/r04_pmc_traffic.json).  Which side to keep resident is the host's choice.
      const int rt = id % a.grt;
      const int rest = id / a.grt;
      const int tt = rest % nt;
      b = rest / nt;
      row0 = rt * GS_ROWS;
      n0 = tt * GS_COLS;
    } else {
      const int tt = id % nt;
      const int rest = id / nt;
      b = rest % nb;
      row0 = (rest / nb) * GS_ROWS;
      n0 = tt * GS_COLS;
    }
  }
  const int S = (K == 2 && a.stride > 1) ? a.stride : 1;      // input stride: S phase sub-signals as virtual channel chunks
  const int n_chunks = ((a.C_in + GS_CI - 1) / GS_CI) * S;   // a ragged last chunk multiplies the packed weights' zero padding
  const long long n_total = flat ? (long long)a.B * a.T_out : (long long)a.T_out;     // columns of this (clip | whole batch)

  if (wave >= 4) {
    // ===================================================================== staging waves
    const int lw = wave - 4;
    const int sl = tid - 256;
    __builtin_amdgcn_s_setprio(FAC_PRIO_STAGE);
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w) + (long long)(row0 / GS_ROWS) * n_chunks * A_BYTES;
    if (a.x_p8 != nullptr) {
      // ------------------------------------------------------------------- P8 input: BOTH operands by LDS-DMA
      // The input arrives as the bf16 planes themselves, 16 bytes per (8-channel group, time step) (fac_conv_desc.x_p8): that is
      // one slot of a B row, so a staging lane copies it straight into its fixed slot -- lane l of a 1 KiB block owns (row
      // blk * 16 + l / 4, slot l % 4) and fetches the piece the XOR swizzle puts there.  No registers, no vector-ALU work per
      // stage (in the fp32 path below the split costs the staging waves ~100 VALU instructions per stage, which do not overlap
      // the MFMAs of their SIMD: the stage of 48 MFMAs then lasts twice its matrix-pipe time).  Columns outside the signal read
      // the zero unit that closes every plane.  Inputs are requested NST - 1 stages ahead, like the weights.
      constexpr int XROWS = GS_COLS + K - 1;
      constexpr int NBB = (XROWS * GS_RB + 1023) / 1024;          // 1 KiB blocks per input plane (the last one partly beyond XROWS)
      constexpr int NB = (3 * NBB + 3) / 4;                       // input DMA instructions per staging wave and stage
      constexpr int LPT = NA + NB;
      const int C8 = a.C_in / 8;
      const unsigned zero_off = (unsigned)(((long long)a.B * C8 * a.T_in) * 16);      // the plane's trailing zero unit
      // per DMA slot j: (plane, row, piece) of this lane; the unit offset inside a plane for channel group 0 of the chunk
      int b_plane[NB], b_row[NB], b_piece[NB], b_lds[NB];
      unsigned b_off[NB];
      bool b_live[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int id = min(j * 4 + lw, 3 * NBB - 1);
        const int plane = id / NBB, blk = id - plane * NBB;
        const int row = blk * 16 + (lane >> 2);
        b_plane[j] = plane;
        b_row[j] = row;
        b_piece[j] = (lane & 3) ^ gs_xswz(row);
        b_lds[j] = A_BYTES + plane * GS_BPL + blk * 1024;
        b_live[j] = row < GS_XR;                                   // rows past the plane's LDS allocation must not be written
        unsigned off = zero_off;
        if (flat) {
          const long long nn = (long long)n0 + row;
          if (nn < n_total && row < XROWS) {
            const long long bb = nn / a.T_out;
            off = (unsigned)(((bb * C8) * a.T_in + (nn - bb * a.T_out)) * 16);
          }
        } else if (S == 1) {
          const int tin = n0 - a.pad_left + row;
          if (tin >= 0 && tin < a.T_in && row < XROWS) off = (unsigned)((((long long)b * C8) * a.T_in + tin) * 16);
        }
        b_off[j] = off;
      }
      const unsigned grp_bytes = (unsigned)a.T_in * 16u;            // one 8-channel group of one clip
      auto issue = [&](int chunk, int buf) {
        const int ch = chunk < n_chunks ? chunk : n_chunks - 1;     // past the end: re-request the last chunk (keeps LPT constant)
        const unsigned char* asrc = wsrc + (long long)ch * A_BYTES;
        unsigned char* dst = sm + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int blk = j * 4 + lw;
          __builtin_amdgcn_global_load_lds((glb_void_t*)(asrc + blk * 1024 + lane * 16), (lds_void_t*)(dst + blk * 1024), 16, 0, 0);
        }
        const int c32 = ch / S, ph = ch - c32 * S;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          int g8 = c32 * 4 + b_piece[j];
          g8 = g8 < C8 ? g8 : C8 - 1;                               // ragged last chunk: zero weights, any finite input will do
          unsigned off = b_off[j];
          if (S > 1) {       // column r of the window <-> sample (n0 + r) * S + phase - pad_left of the padded signal
            const int tin = (n0 + b_row[j]) * S + ph - a.pad_left;
            int idx;
            if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
            else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
            off = (idx >= 0 && b_row[j] < XROWS) ? (unsigned)((((long long)b * C8) * a.T_in + idx) * 16) : zero_off;
          }
          const unsigned goff = off == zero_off ? zero_off : off + (unsigned)g8 * grp_bytes;
          const unsigned char* bsrc = a.x_p8 + (long long)b_plane[j] * a.x_p8_ps;
          if (b_live[j]) __builtin_amdgcn_global_load_lds((glb_void_t*)(bsrc + goff), (lds_void_t*)(dst + b_lds[j]), 16, 0, 0);
        }
      };
      auto landed = [&](bool younger_in_flight) {                   // everything but the youngest stage's requests has landed
        if (younger_in_flight) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      };
      issue(0, 0);
      if (NST == 3) issue(1, 1);
      if (NST == 3) landed(true); else landed(false);
      gs_barrier();                                                 // stage 0 visible to the MFMA waves
      for (int base = 0; base < n_chunks; base += NST) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int c = base + i;
          if (c < n_chunks) {
            // stage (c + NST - 1) % NST was read during iteration c - 1, which every wave has left
            if (c + NST - 1 < n_chunks) issue(c + NST - 1, (i + NST - 1) % NST);
            if (c + 1 < n_chunks) landed(NST == 3 && c + 2 < n_chunks);
            gs_barrier();
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    // input units of this lane: column c of the staged window (row c of the B planes), 8-channel group g
    long long u_off[GS_NU];
    int u_lds[GS_NU], u_g[GS_NU];
    bool u_ok[GS_NU], u_in[GS_NU];
#pragma unroll
    for (int j = 0; j < GS_NU; ++j) {
      // units 0 .. 511: (group, column) = (unit / 128, unit % 128) -- 8 consecutive lanes are 8 aligned consecutive rows of one
      // group (conflict-free stores, gs_xswz); the K - 1 halo columns of the four groups are units 512 ..
      const int unit = j * 256 + sl;
      const int g = unit < 4 * GS_COLS ? unit >> 7 : (unit - 4 * GS_COLS) / (K > 1 ? K - 1 : 1);
      const int c = unit < 4 * GS_COLS ? unit & (GS_COLS - 1) : GS_COLS + (unit - 4 * GS_COLS) % (K > 1 ? K - 1 : 1);
      u_ok[j] = unit < 4 * (GS_COLS + K - 1);                // a unit that exists in the staged window
      u_lds[j] = c * GS_RB + ((g ^ gs_xswz(c)) * 16);
      long long off = 0;
      bool in = false;
      if (flat) {                                            // K == 1: flattened (clip, time) columns, no halo
        const long long nn = (long long)n0 + c;
        if (nn < n_total) {
          const long long bb = nn / a.T_out;
          off = bb * a.x_bs + (nn - bb * a.T_out);
          in = true;
        }
      } else if (S == 1) {
        const int tin = n0 - a.pad_left + c;
        if (tin >= 0 && tin < a.T_in) {
          off = (long long)b * a.x_bs + tin;
          in = true;
        }
      } else {
        off = c;                                             // strided: resolved per chunk (phase-dependent), see load_b
      }
      u_in[j] = in && u_ok[j];
      u_off[j] = off;                                        // batch / column part of the address; the channel is added per chunk
      u_g[j] = g;
    }
    auto issue_a = [&](int chunk, int buf) {
      const unsigned char* src = wsrc + (long long)chunk * A_BYTES;
      unsigned char* dst = sm + buf * STAGE;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int blk = j * 4 + lw;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + blk * 1024 + lane * 16), (lds_void_t*)(dst + blk * 1024), 16, 0, 0);
      }
    };
    // register loads by inline asm (invisible to hipcc's own s_waitcnt placement, as in conv1d_wgrad_split.hip): exactly
    // NBL loads per call, lanes without a real unit load a clamped address and are zeroed at the split
    auto load_b = [&](int chunk, float (&xr)[GS_NU][8], bool (&uin)[GS_NU]) {
      const int c32 = chunk / S, ph = chunk - c32 * S;
#pragma unroll
      for (int j = 0; j < GS_NU; ++j) {
        long long o = u_off[j];
        bool in = u_in[j];
        if (S > 1) {          // column c of the window <-> sample (n0 + c) * S + phase - pad_left of the padded signal
          const int tin = (n0 + (int)u_off[j]) * S + ph - a.pad_left;
          int idx;
          if (a.pad_mode == FAC_PAD_REFLECT) idx = reflect_index(tin, a.T_in, a.T_ext);
          else idx = (tin >= 0 && tin < a.T_in) ? tin : -1;
          in = u_ok[j] && idx >= 0;
          o = (long long)b * a.x_bs + (idx >= 0 ? idx : 0);
        }
        uin[j] = in;
        const float* p = a.x + (in ? o : 0ll);
        const int ch0 = c32 * GS_CI + u_g[j] * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          // channels past C_in (ragged last chunk): their weights are zero, any finite value will do -- re-read the last channel
          const int ch = ch0 + i < a.C_in ? ch0 + i : a.C_in - 1;
          asm volatile("global_load_dword %0, %1, off" : "=v"(xr[j][i]) : "v"(p + (long long)ch * a.x_cs) : "memory");
        }
      }
    };
    auto write_b = [&](int buf, float (&xr)[GS_NU][8], const bool (&uin)[GS_NU]) {
      unsigned char* xd = sm + buf * STAGE + A_BYTES;
#pragma unroll
      for (int j = 0; j < GS_NU; ++j) {
        if (!u_ok[j]) continue;
        bf16x8 h, m, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __bf16 p0, p1, p2;
          gs_split3(uin[j] ? xr[j][i] : 0.f, p0, p1, p2);
          h[i] = p0; m[i] = p1; l[i] = p2;
        }
        *reinterpret_cast<bf16x8*>(xd + u_lds[j]) = h;
        *reinterpret_cast<bf16x8*>(xd + GS_BPL + u_lds[j]) = m;
        *reinterpret_cast<bf16x8*>(xd + 2 * GS_BPL + u_lds[j]) = l;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // wait until everything but the youngest `after` loads of this wave has landed (loads return in order)
    auto landed = [&](int after, float (&xr)[GS_NU][8]) {
      if (after == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (after == NBL) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NBL) : "memory");
      else if (after == NA) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NA) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NA + NBL) : "memory");
#pragma unroll
      for (int j = 0; j < GS_NU; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(xr[j][i]) : : "memory");
    };

    float xa[GS_NU][8], xb[GS_NU][8];
    bool ia[GS_NU], ib[GS_NU];                    // which units of the register sets hold real samples
    // prologue: stage 0 complete; weight slabs of stages 1 .. D-1 and the inputs of chunk 1 in flight
    issue_a(0, 0);
    load_b(0, xa, ia);
    int after = 0;
    if (D == 2 && n_chunks > 1) { issue_a(1, 1); after += NA; }
    if (n_chunks > 1) { load_b(1, xb, ib); after += NBL; }
    landed(after, xa);
    write_b(0, xa, ia);
    gs_barrier();                                 // stage 0 visible to the MFMA waves
    for (int base = 0; base < n_chunks; base += 6) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {               // static LDS stage (i % NST) and register slot (i % 2)
        const int c = base + i;
        if (c < n_chunks) {
          // at this point in flight: [weights of c + 1 (D == 2)], inputs of c + 1
          int aft = 0;
          if (c + D < n_chunks) {
            issue_a(c + D, (i + D) % NST);        // that stage was read during iteration c - 1, which every wave has left
            if (D == 2) aft += NA;                // D == 1: these are the weights of c + 1 themselves -- must land now
          }
          if (c + 2 < n_chunks) {
            if (i % 2 == 0) load_b(c + 2, xa, ia); else load_b(c + 2, xb, ib);
            aft += NBL;
          }
          if (c + 1 < n_chunks) {
            if (i % 2 == 0) { landed(aft, xb); write_b((i + 1) % NST, xb, ib); }
            else { landed(aft, xa); write_b((i + 1) % NST, xa, ia); }
          }
          gs_barrier();
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_setprio(0);
  } else {
    // ========================================================================= MFMA waves: 64 x 64 each (2 x 2 blocks)
    __builtin_amdgcn_s_setprio(FAC_PRIO_MFMA);
    const int l31 = lane & 31, kq = lane >> 5;
    const int mh = wave >> 1, nh = wave & 1;
    const int swa = (l31 >> 2) & 3;
    const int aoff = (mh * 64 + l31) * GS_RB;
    int apo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) apo[ks] = ((ks * 2 + kq) ^ swa) * 16;
    // B fragment of tap k: row (column + k); its swizzle depends on the row
    int boff[K][2];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int row = nh * 64 + l31 + k;
        boff[k][ks] = A_BYTES + row * GS_RB + (((ks * 2 + kq) ^ gs_xswz(row)) * 16);
      }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // step s = k * 2 + ks of a stage
    auto ld_frags = [&](const unsigned char* st, int step, bf16x8 (&A)[2][3], bf16x8 (&Bf)[2][3]) {
      const int k = step >> 1, ks = step & 1;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          A[m][p] = *reinterpret_cast<const bf16x8*>(st + (p * K + k) * GS_APL + aoff + m * 32 * GS_RB + apo[ks]);
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n)     // rows 32 apart: gs_xswz(row) is unchanged, the swizzle of block n = that of block 0
          Bf[n][p] = *reinterpret_cast<const bf16x8*>(st + boff[k][ks] + p * GS_BPL + n * 32 * GS_RB);
    };

    gs_barrier();   // stage 0 staged
    bf16x8 A[2][2][3], Bf[2][2][3];
    for (int base = 0; base < n_chunks; base += NST) {
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const int chunk = base + i;
        if (chunk < n_chunks) {
          const unsigned char* st = sm + i * STAGE;
          ld_frags(st, 0, A[0], Bf[0]);
#pragma unroll
          for (int step = 0; step < 2 * K; ++step) {
            if (step + 1 < 2 * K) ld_frags(st, step + 1, A[(step + 1) & 1], Bf[(step + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};     // smallest terms first
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
              for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                  acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[step & 1][m][TA[q]], Bf[step & 1][n][TB[q]], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          gs_barrier();
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // accumulators -> fp32 tile in LDS (every stage buffer is free: nothing is in flight after the last barrier)
    float* tile = reinterpret_cast<float*>(sm);
    constexpr int EP = GS_COLS + 4;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tile[(mh * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq) * EP + nh * 64 + n * 32 + l31] = acc[m][n][r];
  }
  gs_barrier();

  // ---- epilogue by all eight waves: one lane = 4 consecutive output samples of one channel
  {
    const float* tile = reinterpret_cast<const float*>(sm);
    constexpr int EP = GS_COLS + 4;
    constexpr int NTH = 512;
    const int rp = a.rp;
    const int cpt = GS_ROWS / rp;                           // channels per tile (rp == 1: 128)
    const int QPR = (GS_COLS * rp) / 4;                     // output quads per channel in this tile
    const int ch0 = (row0 / GS_ROWS) * cpt;
    const long long u_tot = n_total * rp;                   // outputs per (clip | batch) row
    const long long u0 = (long long)n0 * rp;
    const bool al_ok = (a.y_cs & 3) == 0 && (a.y_bs & 3) == 0 && (!a.y || (reinterpret_cast<unsigned long long>(a.y) & 15) == 0) &&
                       (!a.y2 || (reinterpret_cast<unsigned long long>(a.y2) & 15) == 0) &&
                       (!a.res || (reinterpret_cast<unsigned long long>(a.res) & 15) == 0) && (!flat || (a.T_out & 3) == 0);
    for (int q = tid; q < cpt * QPR; q += NTH) {
      const int cl = q / QPR, uq = q - cl * QPR;
      const int co = ch0 + cl;
      const long long u = u0 + 4 * uq;
      if (co >= a.C_out || u >= u_tot) continue;
      float v[4];
      if (rp == 1) {
        const float4 av = *reinterpret_cast<const float4*>(tile + cl * EP + 4 * uq);
        v[0] = av.x; v[1] = av.y; v[2] = av.z; v[3] = av.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ul = 4 * uq + i, tl = ul / rp, p = ul - tl * rp;
          v[i] = tile[(cl * rp + p) * EP + tl];
        }
      }
      // output address of sample u (flat: u -> (clip, time); a quad never straddles clips when T_out % 4 == 0)
      long long o, lim = u_tot - u;
      if (flat) {
        const long long bb = u / a.T_out, tt = u - bb * a.T_out;
        o = bb * a.y_bs + (long long)co * a.y_cs + tt;
        const long long in_clip = a.T_out - tt;
        lim = lim < in_clip ? lim : in_clip;
      } else {
        o = (long long)b * a.y_bs + (long long)co * a.y_cs + u;
      }
      const bool full = al_ok && lim >= 4;
      const float bs = a.bias ? a.bias[co] : 0.f;
      const float al = a.alpha_out ? a.alpha_out[co] : 0.f;
      const float inv = a.alpha_out ? snake_inv(al) : 0.f;
      if (!full && flat && lim < 4) {
        // ragged quad of the flattened layout (T_out % 4 != 0): sample by sample
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long long ui = u + i;
          if (ui >= u_tot) continue;
          const long long bb = ui / a.T_out, tt = ui - bb * a.T_out;
          const long long oi = bb * a.y_bs + (long long)co * a.y_cs + tt;
          float x = v[i] + bs;
          if (a.alpha_out) x = snake_apply(x, al, inv);
          if (a.act != FAC_ACT_NONE) x = apply_act_slow(x, a.act);
          if (a.res) x += a.res[oi];
          if (a.y) a.y[oi] = x;
          if (a.y2) { const float a2 = a.alpha2[co]; a.y2[oi] = snake_apply(x, a2, snake_inv(a2)); }
        }
        continue;
      }
      float rv[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.res) {
        if (full) {
          const float4 r4 = *reinterpret_cast<const float4*>(a.res + o);
          rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) rv[i] = i < lim ? a.res[o + i] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = v[i] + bs;
        if (a.alpha_out) x = snake_apply(x, al, inv);
        if (a.act != FAC_ACT_NONE) x = apply_act_slow(x, a.act);
        v[i] = x + rv[i];
      }
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.y2) {
        const float a2 = a.alpha2[co], i2 = snake_inv(a2);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = snake_apply(v[i], a2, i2);
      }
      if (full) {
        if (a.y) *reinterpret_cast<float4*>(a.y + o) = make_float4(v[0], v[1], v[2], v[3]);
        if (a.y2) *reinterpret_cast<float4*>(a.y2 + o) = make_float4(w[0], w[1], w[2], w[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i >= lim) continue;
          if (a.y) a.y[o + i] = v[i];
          if (a.y2) a.y2[o + i] = w[i];
        }
      }
    }
  }
}

// Shapes the kernel takes.  K = 1: pad_left 0, T_in >= T_out; K = 2: zero padding (pad_left 0 or 1), per-clip tiles.
bool conv_gsplit_ok(const ConvArgs& a) {
  static const bool on = !(getenv("FAC_GEMM_SPLIT") && getenv("FAC_GEMM_SPLIT")[0] == '0');
  if (!on) return false;
  const bool strided = a.stride > 1;          // K <= 2 * stride taps as 2 taps of `stride` phase sub-signals
  if (!(a.dil == 1 && a.n_phase == 1 && a.phase_shift == 0 && a.y_tstride == 1 && !a.alpha_in && !a.w1 && !a.w_batched &&
        !conv_two_level(a)))
    return false;
  if (strided) {
    if (!(a.K > a.stride && a.K <= 2 * a.stride && a.stride <= 16 && a.rp == 1)) return false;
  } else if (!(a.K == 1 || a.K == 2)) {
    return false;
  }
  if (a.C_in < (strided ? 32 : 64)) return false;
  if (!strided && a.K == 1 && (a.pad_left != 0 || a.T_in < a.T_out)) return false;
  if (!strided && a.K == 2 && (a.pad_mode != FAC_PAD_ZERO || a.pad_left > 1)) return false;
  const int rows = a.rp > 1 ? a.C_out_pad : a.C_out;
  if (rows < 64) return false;
  const long long cols = (long long)a.B * a.T_out;
  if (cols < 1024) return false;
  if ((strided || a.K == 2) && a.T_out < 256) return false;   // per-clip tiles: short clips leave half-empty tiles to the fp32 kernel
  return (long long)a.B * a.x_bs < (1ll << 40);
}

template <int K>
static int gsplit_launch(ConvArgs& a, hipStream_t s) {
  auto kern = conv1d_gemm_split_kernel<K>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  size_t lds = (size_t)gs_stages(K) * gs_stage_bytes(K);
  const size_t epi = (size_t)GS_ROWS * (GS_COLS + 4) * sizeof(float);
  if (lds < epi) lds = epi;
  a.gflat = (K == 1) ? 1 : 0;
  const long long n_total = a.gflat ? (long long)a.B * a.T_out : (long long)a.T_out;
  a.n_t_tiles = (int)((n_total + GS_COLS - 1) / GS_COLS);
  const int rows = a.rp > 1 ? a.C_out_pad : a.C_out;
  const int n_rt = (rows + GS_ROWS - 1) / GS_ROWS;
  const long long n_wg = (long long)a.n_t_tiles * n_rt * (a.gflat ? 1 : a.B);
  // order of the tiles inside an XCD's share (see the kernel): FAC_GS_ROW_FAST = 0 / 1 forces, default = by bytes: keep the
  // weights resident when re-reading the input per row tile is the cheaper side, else the input
  {
    static const int env = [] { const char* e = getenv("FAC_GS_ROW_FAST"); return e == nullptr ? -1 : (e[0] != '0' ? 1 : 0); }();
    // bytes that leave the L2s under either order (model): x = the input, W = all split weights, n_ct column tiles.
    //   row tile slowest: the input once per row tile, the weights once                          -> x * n_rt + W
    //   row tile fastest: the input once; the weights stay in an XCD's 4 MB L2 if they fit (8 copies), else they are streamed
    //                     once per group of g column tiles that are co-resident with all n_rt row tiles -> x + W * n_ct / g
    const double x_bytes = 4.0 * a.B * a.C_in * (double)a.T_in;
    const double w_bytes = 6.0 * rows * (double)a.C_in * K * (a.stride > 1 && K == 2 ? a.stride : 1);
    const double n_ct = (double)a.n_t_tiles * (a.gflat ? 1 : a.B);
    const double g = 48.0 / n_rt > 1.0 ? 48.0 / n_rt : 1.0;
    const double slow = x_bytes * n_rt + w_bytes;
    const double fast = x_bytes + (w_bytes <= 3.5 * 1048576.0 ? 8.0 * w_bytes : w_bytes * n_ct / g);
    const bool by_bytes = n_rt > 1 && fast < 0.8 * slow;
    a.grt = (env >= 0 ? env == 1 : by_bytes) ? n_rt : 0;
  }
  if (n_wg > 0x7fffffffll) {
    set_error("conv1d(gemm split): too many workgroups (%lld)", n_wg);
    return FAC_ERR_ARG;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), lds, s, a);
  return check_launch("conv1d_gemm_split");
}

int conv_dispatch_gsplit(ConvArgs& a, hipStream_t s) {
  return (a.K == 1 && a.stride == 1) ? gsplit_launch<1>(a, s) : gsplit_launch<2>(a, s);
}

}  // namespace fac

// K: taps of the conv; in_stride S: 1, or the stride of a strided conv with S < K <= 2 S (stored as 2 taps x S phase chunks)
extern "C" int64_t fac_gemm_w_split_bytes(int R, int C_in, int K, int in_stride) {
  using namespace fac;
  const int S = in_stride > 1 ? in_stride : 1, Kt = S > 1 ? 2 : K;
  const int64_t n_tiles = (R + GS_ROWS - 1) / GS_ROWS, n_ch = (int64_t)((C_in + GS_CI - 1) / GS_CI) * S;
  return n_tiles * n_ch * 3 * Kt * GS_APL;
}

extern "C" int fac_pack_gemm_w_split(const float* v, int64_t row_stride, int64_t ci_stride, int64_t k_stride, const float* row_scale,
                                     void* out, int R, int C_in, int K, int in_stride, fac_stream_t stream) {
  using namespace fac;
  const int S = in_stride > 1 ? in_stride : 1;
  FAC_REQUIRE(v && out && R > 0 && C_in > 0 && K >= 1 && ((S == 1 && K <= 2) || (S > 1 && K > S && K <= 2 * S)),
              "pack_gemm_w_split: bad arguments (K must be 1 or 2, or in (S, 2S] for input stride S)");
  const int Kt = S > 1 ? 2 : K;
  const int n_tiles = (R + GS_ROWS - 1) / GS_ROWS, n_ch = ((C_in + GS_CI - 1) / GS_CI) * S;
  const long long n = (long long)n_tiles * n_ch * Kt * GS_ROWS * 4;
  const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  if (prep_recording()) {
    PrepJob j{}; j.a = v; j.b = row_scale; j.out = out; j.kind = PK_GEMM_SPLIT; j.nblocks = blocks; j.n = n;
    j.l[0] = row_stride; j.l[1] = ci_stride; j.l[2] = k_stride;
    j.i[0] = R; j.i[1] = C_in; j.i[2] = Kt; j.i[3] = n_ch; j.i[4] = S; j.i[5] = K;
    return prep_record(PU_GSPLIT, j);
  }
  hipLaunchKernelGGL(pack_gemm_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, (long long)row_stride,
                     (long long)ci_stride, (long long)k_stride, row_scale, reinterpret_cast<unsigned char*>(out), R, C_in, Kt, n_ch, S, K, n);
  return check_launch("pack_gemm_w_split");
}
