// C-ABI entry points of the MFMA conv (kernel: conv1d_mfma.h; instantiations: conv1d_tile_*.hip).
#include "conv1d_mfma.h"
#include <stdlib.h>

namespace fac {

// Tile selection: M tile by output channels, narrow-N tile for short sequences (LSTM batches).
static int select_variant(const fac_conv_desc* d) {
  const int co = d->C_out;
  if (d->row_phases > 1) return 5;      // (channel, phase) rows: the 128 x 256 tile with the all-waves LDS epilogue
  if (d->T_out <= 32) return 0;
  if (co <= 32) return 1;
  if (co <= 64) return 2;
  // k = 1 convs re-use nothing across taps: per staged byte they do 7x less MFMA work than k = 7 and are
  // LDS-DMA-bound on 128-wide time tiles; long sequences take 256-wide tiles with 8 MFMA waves
  // (measured +15..30 % on the k = 1 layers, neutral on k = 7).
  static const bool k1_wide = !(getenv("FAC_K1_WIDE") && getenv("FAC_K1_WIDE")[0] == '0');
  const bool wide = k1_wide && d->K == 1 && d->n_phase == 1 && d->T_out >= 512;   // (wide tiles measured slower for K = 2)
  // 2 s clips are 160 latent frames: a 160-wide tile wastes nothing where 128 + 32 would waste 37 %
  if (d->T_out > 128 && d->T_out <= 160 && co > 64) return 8;
  if (co % 128 != 0 && co % 96 == 0) return wide ? 6 : 3;
  return wide ? 5 : 4;
}

#if defined(FAC_PROF) || defined(FAC_PROF2)
unsigned long long* g_conv_dbg = nullptr;
#endif

// One or two output channels, plain stride-1 conv with nothing but bias / activation in the epilogue.
// (it parallelises over (batch, 1024-step tile) only: with fewer than ~128 such tiles -- the period discriminators' 1024 -> 1
// output conv over one row-concatenated signal -- the MFMA tile is 10x faster despite wasting 31 of its 32 rows)
static const bool NARROW_TWO_LEVEL = !(getenv("FAC_NARROW_TWO_LEVEL") && getenv("FAC_NARROW_TWO_LEVEL")[0] == '0');
static bool narrow_ok(const fac_conv_desc* d) {
  return d->C_out <= 2 && d->stride == 1 && d->n_phase == 1 && d->phase_shift == 0 && d->y_tstride == 1 && !d->res && !d->y2 &&
         !d->w_batched && d->y && (long long)d->B <= 65535 && (long long)d->B * ((d->T_out + 1023) / 1024) >= 128;
}

}  // namespace fac

#if defined(FAC_PROF) || defined(FAC_PROF2)
extern "C" void fac_debug_set_buffer(void* p) { fac::g_conv_dbg = (unsigned long long*)p; }
#endif

extern "C" int fac_conv1d_fwd(const fac_conv_desc* d, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(d && (d->x || d->x_p8) && (d->w || d->w_split) && (d->y || d->y2 || d->y2_p8), "conv1d: null pointer");
  FAC_REQUIRE(!d->y2 || d->alpha_y2 || d->act == FAC_ACT_WN_RES_SKIP, "conv1d: y2 needs alpha_y2");
  FAC_REQUIRE(d->B > 0 && d->C_in > 0 && d->C_out > 0 && d->T_in > 0 && d->T_out > 0,
              "conv1d: bad shape B=%d C_in=%d C_out=%d T_in=%d T_out=%d", d->B, d->C_in, d->C_out,
              d->T_in, d->T_out);
  FAC_REQUIRE(d->K >= 1 && d->stride >= 1 && d->dilation >= 1 && d->pad_left >= 0,
              "conv1d: bad K/stride/dilation/pad");
  FAC_REQUIRE(d->C_out_pad % 32 == 0 && (d->C_out_pad >= d->C_out || d->row_phases > 1), "conv1d: C_out_pad must be a multiple of 32");
  FAC_REQUIRE(d->n_phase >= 1 && d->y_tstride >= 1, "conv1d: bad phase config");
  FAC_REQUIRE((long long)d->B * d->n_phase <= 65535, "conv1d: B*n_phase too large for grid.z");
  ConvArgs a;
  a.x = d->x; a.w = d->w; a.bias = d->bias; a.alpha_in = d->alpha_in; a.alpha_out = d->alpha_out;
  a.res = d->res; a.y = d->y; a.y2 = d->y2; a.alpha2 = d->alpha_y2;
  a.w1 = d->w_k1; a.bias1 = d->bias_k1;
  a.x_p8 = reinterpret_cast<const unsigned char*>(d->x_p8); a.x_p8_ps = d->x_p8_plane_bytes;
  a.y2_p8 = reinterpret_cast<unsigned char*>(d->y2_p8); a.y2_p8_ps = d->y2_p8_plane_bytes;
  FAC_REQUIRE(!d->y2_p8 || d->alpha_y2, "conv1d: y2_p8 needs alpha_y2");
  if (d->w_k1) {
    FAC_REQUIRE(d->C_in == d->C_out && d->C_out_pad == d->C_out && d->n_phase == 1 && d->stride == 1 &&
                    d->alpha_out && d->act == FAC_ACT_NONE && !d->w_batched,
                "conv1d: fused ResidualUnit needs C_in == C_out (multiple of 32), stride 1, a Snake alpha_out");
  }
  a.x_bs = d->x_bs; a.x_cs = d->x_cs; a.y_bs = d->y_bs; a.y_cs = d->y_cs; a.w_bs = d->w_bs;
  a.B = d->B; a.C_in = d->C_in; a.T_in = d->T_in; a.C_out = d->C_out; a.C_out_pad = d->C_out_pad;
  a.T_out = d->T_out; a.K = d->K; a.stride = d->stride; a.dil = d->dilation;
  a.pad_left = d->pad_left; a.pad_mode = d->pad_mode; a.n_phase = d->n_phase;
  a.y_tstride = d->y_tstride; a.act = d->act; a.w_batched = d->w_batched;
  a.phase_shift = d->phase_shift;
  a.rp = d->row_phases > 1 ? d->row_phases : 1;
  if (a.rp > 1) {
    FAC_REQUIRE(d->n_phase == 1 && d->y_tstride == 1 && d->phase_shift == 0 && d->stride == 1 && a.rp <= 128 && !d->w_k1 &&
                    !d->w_batched && d->C_out_pad == fac_convtr_rows(d->C_out, a.rp) && !(d->K1 > 0 && d->K1 < d->K),
                "conv1d: row_phases needs a plain stride-1 launch on weights from fac_pack_convtr_w_rows");
  }
  a.K1 = d->K1 > 0 ? d->K1 : d->K; a.dil2 = d->dilation2;
  FAC_REQUIRE(a.K1 <= d->K && d->K % a.K1 == 0 && (a.K1 == d->K || d->dilation2 > 0), "conv1d: bad two-level taps (K=%d K1=%d)", d->K, d->K1);
  FAC_REQUIRE(!conv_two_level(a) || (!d->w_k1 && d->n_phase == 1 && d->pad_mode == FAC_PAD_ZERO && !d->w_batched),
              "conv1d: two-level taps need a plain, zero-padded conv");
  conv_set_virtual(a);
  FAC_REQUIRE(d->phase_shift >= 0 && d->phase_shift < d->n_phase + (d->n_phase == 1), "conv1d: bad phase_shift");
  // length of pad1d's temporary zero extension (only differs from T_in for inputs shorter than the pad)
  {
    long long last = (long long)(d->T_out - 1) * d->stride + (long long)conv_max_tap_offset(a) - d->pad_left;
    int pad_right = last >= d->T_in ? (int)(last - d->T_in + 1) : 0;
    int max_pad = d->pad_left > pad_right ? d->pad_left : pad_right;
    a.T_ext = d->T_in > max_pad ? d->T_in : max_pad + 1;
  }
  hipStream_t s = (hipStream_t)stream;
  if (d->x_p8 || d->y2_p8) {      // P8 operands exist only in the kernels listed at fac_conv_desc.x_p8: no silent fp32 detour
    const bool gs = d->w_split && (d->K <= 2 || (d->stride > 1 && d->K <= 2 * d->stride)) && conv_gsplit_ok(a) &&
                    !conv_skinny_ok(a, d->ws, d->ws_bytes) && d->C_in % 8 == 0 && d->x_p8_plane_bytes < (1ll << 32);
    const bool ok = !d->y2_p8 && (gs || (d->w_split && !conv_two_level(a) && conv_bsplit_p8_ok(a) && !conv_cin1_ok(a)));
    FAC_REQUIRE(ok, "conv1d: P8 operands given but the launch does not run on a kernel that takes them (K=%d stride=%d C_in=%d columns=%lld)",
                d->K, d->stride, d->C_in, (long long)d->B * d->T_out);
  }
  if (d->act == FAC_ACT_GATE || d->act == FAC_ACT_WN_RES_SKIP) {    // epilogues of the split-reduction kernel only
    FAC_REQUIRE(d->w && !d->w_k1 && !d->x_p8 && !conv_two_level(a) && conv_skinny_ok(a, d->ws, d->ws_bytes),
                "conv1d: FAC_ACT_GATE / FAC_ACT_WN_RES_SKIP exist only for few-column launches (B * T_out <= 640) with a workspace");
    return conv_dispatch_skinny(a, d->ws, d->ws_bytes, s);
  }
  if (d->w_k1) return conv_dispatch_fused_ru(a, s);
  a.gflat = 0;
  a.grt = 0;
  // few-output-channel 9- / 3-tap convs (two-level taps included) with split weights of fac_pack_conv_w_split2
  if (d->w_split && (a.KV == 9 || a.KV == 3) && d->C_out <= 32 && conv_bsplit2_ok(a)) {
    a.w = reinterpret_cast<const float*>(d->w_split);
    return conv_dispatch_bsplit2(a, s);
  }
  FAC_REQUIRE(!conv_two_level(a) || d->w, "conv1d: two-level taps outside the split kernel's shapes need fp32 weights");
  static const bool pw_on = !(getenv("FAC_PW") && getenv("FAC_PW")[0] == '0');
  // stride-2 layers with few channels (weights resident in LDS as bf16 planes, inputs streamed): conv1d_pw_split.hip
  if (pw_on && d->pw_split && d->w && conv_pwt_ok(a)) return conv_dispatch_pwt(a, s);
  // 1- / 2-tap convs with split weights in the GEMM layout (fac_pack_gemm_w_split): the bf16 matrix pipe, fp32-grade
  if (d->w_split && (d->K <= 2 || (d->stride > 1 && d->K <= 2 * d->stride)) && conv_gsplit_ok(a) &&
      !conv_skinny_ok(a, d->ws, d->ws_bytes)) {
    a.w = reinterpret_cast<const float*>(d->w_split);
    return conv_dispatch_gsplit(a, s);
  }
  if (a.rp > 1) {
    FAC_REQUIRE(d->w != nullptr && d->w != (const float*)d->w_split, "conv1d: row_phases launch outside the split kernel's shapes needs fp32 weights");
    return conv_dispatch_128x256(a, s);
  }
  const bool two_level = conv_two_level(a);
  // K = 3 / 5 / 7 split kernel first (its shapes exclude the few-column launches the split-reduction kernel takes)
  if (d->w_split && !two_level && conv_bsplit_ok(a) && !conv_cin1_ok(a)) {
    a.w = reinterpret_cast<const float*>(d->w_split);
    return conv_dispatch_bsplit(a, s);
  }
  // every kernel below reads fp32 weights: a split-only launch (w == w_split or NULL) must not get here
  FAC_REQUIRE(d->w != nullptr && (const void*)d->w != d->w_split,
              "conv1d: shape does not qualify for a split-bf16 kernel (K=%d stride=%d C_in=%d C_out=%d columns=%lld) and no fp32 "
              "weights were given", d->K, d->stride, d->C_in, d->C_out, (long long)d->B * d->T_out);
  if (!two_level && conv_skinny_ok(a, d->ws, d->ws_bytes)) return conv_dispatch_skinny(a, d->ws, d->ws_bytes, s);
  if (narrow_ok(d) && (!two_level || (NARROW_TWO_LEVEL && (a.KV - 1) * a.dil <= 64))) return conv_dispatch_narrow(a, s);
  if (conv_thin_ok(a, d->ws, d->ws_bytes)) return conv_dispatch_thin(a, d->ws, s);   // C_out <= 2 without enough tiles for narrow
  if (conv_cin1_ok(a)) return conv_dispatch_cin1(a, s);
  if (pw_on && d->pw_split && conv_pw_ok(a) && conv_pws_ok(a)) return conv_dispatch_pws(a, s);   // k = 1 tails at C <= 192 on the bf16 pipe
  if (pw_on && conv_pw_ok(a)) return conv_dispatch_pw(a, s);
  switch (select_variant(d)) {
    case 0: return conv_dispatch_128x32(a, s);
    case 1: return conv_dispatch_32x256(a, s);
    case 2: return conv_dispatch_64x128(a, s);
    case 3: return conv_dispatch_96x128(a, s);
    case 5: return conv_dispatch_128x256(a, s);
    case 6: return conv_dispatch_96x256(a, s);
    case 8: return conv_dispatch_128x160(a, s);
    default: return conv_dispatch_128x128(a, s);
  }
}

extern "C" int fac_conv1d_variant(const fac_conv_desc* d, char* name, int name_len) {
  using namespace fac;
  FAC_REQUIRE(d, "conv1d_variant: null descriptor");
  static const char* names[] = {"conv1d_mfma_kernel<1,1,4,1,K> 128x32", "conv1d_mfma_kernel<1,2,1,4,K> 32x256",
                                "conv1d_mfma_kernel<2,1,1,4,K> 64x128", "conv1d_mfma_kernel<3,1,1,4,K> 96x128",
                                "conv1d_mfma_kernel<2,2,2,2,K> 128x128", "conv1d_mfma_kernel<2,2,2,4,K> 128x256",
                                "conv1d_mfma_kernel<3,1,1,8,K> 96x256", "fused", "conv1d_mfma_kernel<1,5,4,1,K> 128x160"};
  if (d->w_k1) {
    if (name && name_len > 0) snprintf(name, name_len, "conv1d_mfma_kernel<C/32,1,1,4,7,fused RU> Cx128");
    return 7;
  }
  if (d->pw_split && d->w && !(getenv("FAC_PW") && getenv("FAC_PW")[0] == '0')) {
    ConvArgs a{};
    a.K = d->K; a.K1 = d->K1 > 0 ? d->K1 : d->K; a.stride = d->stride; a.dil = d->dilation; a.rp = d->row_phases > 1 ? d->row_phases : 1;
    a.pad_left = d->pad_left; a.pad_mode = d->pad_mode; a.w = d->w; a.x = d->x; a.n_phase = d->n_phase; a.phase_shift = d->phase_shift;
    a.y_tstride = d->y_tstride; a.alpha_in = d->alpha_in; a.alpha_out = d->alpha_out; a.act = d->act; a.res = d->res; a.w1 = d->w_k1;
    a.w_batched = d->w_batched; a.x_p8 = reinterpret_cast<const unsigned char*>(d->x_p8); a.y2_p8 = reinterpret_cast<unsigned char*>(d->y2_p8);
    a.C_in = d->C_in; a.C_out = d->C_out; a.C_out_pad = d->C_out_pad; a.y_cs = d->y_cs; a.y_bs = d->y_bs; a.B = d->B; a.T_out = d->T_out;
    if (conv_pwt_ok(a)) {
      if (name && name_len > 0) snprintf(name, name_len, "conv1d_pwt_kernel<%d taps> (streaming, W planes in LDS, bf16x3 split, fp32-grade)", d->K);
      return 18;
    }
  }
  if (d->w_split && d->C_out <= 32) {
    ConvArgs a{};
    a.K = d->K; a.K1 = d->K1 > 0 ? d->K1 : d->K; a.C_in = d->C_in; a.dil2 = d->dilation2; conv_set_virtual(a);
    a.stride = d->stride; a.dil = d->dilation; a.n_phase = d->n_phase; a.phase_shift = d->phase_shift; a.y_tstride = d->y_tstride;
    a.rp = d->row_phases > 1 ? d->row_phases : 1; a.alpha_in = d->alpha_in; a.w1 = d->w_k1; a.w_batched = d->w_batched;
    a.pad_mode = d->pad_mode; a.C_out = d->C_out; a.B = d->B; a.T_out = d->T_out; a.T_in = d->T_in; a.x_cs = d->x_cs;
    if ((a.KV == 9 || a.KV == 3) && conv_bsplit2_ok(a)) {
      if (name && name_len > 0) snprintf(name, name_len, "conv1d_bsplit2_kernel<%d,%d> 32x512 (bf16x3 split, fp32-grade)", a.KV, a.stride);
      return 16;
    }
  }
  if (d->w_split && (d->K <= 2 || (d->stride > 1 && d->K <= 2 * d->stride))) {
    ConvArgs a{};
    a.K = d->K; a.stride = d->stride; a.dil = d->dilation; a.n_phase = d->n_phase; a.phase_shift = d->phase_shift; a.y_tstride = d->y_tstride;
    a.alpha_in = d->alpha_in; a.w1 = d->w_k1; a.w_batched = d->w_batched; a.C_in = d->C_in; a.C_out = d->C_out; a.C_out_pad = d->C_out_pad;
    a.K1 = d->K; a.pad_left = d->pad_left; a.pad_mode = d->pad_mode; a.T_in = d->T_in; a.T_out = d->T_out; a.B = d->B; a.x_bs = d->x_bs;
    a.rp = d->row_phases > 1 ? d->row_phases : 1; a.x_cs = d->x_cs;
    if (conv_gsplit_ok(a) && !conv_skinny_ok(a, d->ws, d->ws_bytes)) {
      if (name && name_len > 0)
        snprintf(name, name_len, "conv1d_gemm_split_kernel<%d> 128x128 (bf16x3 split GEMM, fp32-grade)", d->stride > 1 ? 2 : d->K);
      return 15;
    }
  }
  if (d->row_phases > 1) {
    if (name && name_len > 0) snprintf(name, name_len, "conv1d_mfma_kernel<2,2,2,4,2> 128x256 (convtr, all phases per tile)");
    return 5;
  }
  {
    ConvArgs a{};
    a.alpha_in = d->alpha_in; a.w1 = d->w_k1; a.w_batched = d->w_batched; a.phase_shift = d->phase_shift;
    a.pad_mode = d->pad_mode; a.pad_left = d->pad_left; a.B = d->B; a.T_out = d->T_out; a.C_in = d->C_in; a.C_out = d->C_out; a.K = d->K;
    a.n_phase = d->n_phase; a.x_cs = d->x_cs; a.x_bs = d->x_bs;
    if (!(d->K1 > 0 && d->K1 < d->K) && conv_skinny_ok(a, d->ws, d->ws_bytes)) {
      if (name && name_len > 0) snprintf(name, name_len, "conv1d_skinny_kernel (split reduction, <=640 columns)");
      return 10;
    }
  }
  {
    const bool two = d->K1 > 0 && d->K1 < d->K;
    if (narrow_ok(d) && (!two || (NARROW_TWO_LEVEL && (d->K1 - 1) * d->dilation <= 64))) {
      if (name && name_len > 0) snprintf(name, name_len, two ? "conv1d_narrow_kernel (VALU, C_out<=2, two-level taps)" : "conv1d_narrow_kernel (VALU, C_out<=2)");
      return 9;
    }
  }
  {
    ConvArgs a{};
    a.C_in = d->C_in; a.C_out = d->C_out; a.K = d->K; a.stride = d->stride; a.n_phase = d->n_phase; a.phase_shift = d->phase_shift;
    a.y_tstride = d->y_tstride; a.res = d->res; a.y2 = d->y2; a.w1 = d->w_k1; a.w_batched = d->w_batched; a.y = d->y; a.B = d->B;
    a.T_out = d->T_out; a.K1 = d->K1 > 0 ? d->K1 : d->K;
    if (conv_thin_ok(a, d->ws, d->ws_bytes)) {
      if (name && name_len > 0) snprintf(name, name_len, "conv1d_thin_kernel (VALU, C_out<=8, split channels)");
      return 13;
    }
  }
  {
    ConvArgs a{};
    a.C_in = d->C_in; a.C_out = d->C_out; a.K = d->K; a.stride = d->stride; a.dil = d->dilation; a.n_phase = d->n_phase;
    a.phase_shift = d->phase_shift; a.y_tstride = d->y_tstride; a.alpha_in = d->alpha_in; a.res = d->res; a.w1 = d->w_k1;
    a.w_batched = d->w_batched; a.B = d->B; a.K1 = d->K1 > 0 ? d->K1 : d->K; a.T_out = d->T_out;
    if (conv_cin1_ok(a)) {
      if (name && name_len > 0) snprintf(name, name_len, "conv1d_cin1_kernel (VALU, C_in=1, store stream)");
      return 12;
    }
    a.K = d->K; a.pad_left = d->pad_left; a.T_in = d->T_in; a.C_out_pad = d->C_out_pad; a.w = d->w;
    static const bool pw_on = !(getenv("FAC_PW") && getenv("FAC_PW")[0] == '0');
    a.x_p8 = reinterpret_cast<const unsigned char*>(d->x_p8);
    if (pw_on && d->pw_split && conv_pw_ok(a) && conv_pws_ok(a)) {
      if (name && name_len > 0) snprintf(name, name_len, "conv1d_pws_kernel (k=1 streaming, W planes in LDS, bf16x3 split, fp32-grade)");
      return 17;
    }
    if (pw_on && conv_pw_ok(a)) {
      if (name && name_len > 0) snprintf(name, name_len, "conv1d_pw_kernel (k=1 streaming, W in LDS)");
      return 14;
    }
  }
  if (d->w_split) {
    ConvArgs a{};
    a.K = d->K; a.stride = d->stride; a.n_phase = d->n_phase; a.phase_shift = d->phase_shift; a.y_tstride = d->y_tstride;
    a.alpha_in = d->alpha_in; a.w1 = d->w_k1; a.w_batched = d->w_batched; a.C_in = d->C_in; a.dil = d->dilation;
    a.B = d->B; a.T_out = d->T_out;
    if (conv_bsplit_ok(a)) {
      if (name && name_len > 0) snprintf(name, name_len, "conv1d_bsplit_kernel<%d> 64x256 (bf16x3 split, fp32-grade)", d->K);
      return 11;
    }
  }
  const int v = select_variant(d);
  if (name && name_len > 0) snprintf(name, name_len, "%s", names[v]);
  return v;
}
