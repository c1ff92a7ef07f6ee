// Fused ResidualUnit instantiations (dac/model/dac.py:25-42): conv k7 (dilated) -> +bias -> Snake ->
// conv k1 -> +bias -> +x  [-> second, pre-activated output] in ONE launch.  One C x 128 tile per
// workgroup with every channel of a column inside one wave (WM = 1), so the 1x1 conv consumes the
// k7 accumulators directly as MFMA B fragments (see conv1d_mfma.h).  C in {64, 96, 128}.
#include "conv1d_mfma.h"

namespace fac {
int conv_dispatch_fused_ru(ConvArgs& a, hipStream_t s) {
  if (a.K != 7) {
    set_error("fused ResidualUnit: only kernel_size 7 is instantiated (got %d)", a.K);
    return FAC_ERR_ARG;
  }
  switch (a.C_out) {
    case 64: return launch_cfg<2, 1, 1, 4, 7, true>(a, s);     // (64 x 256 measured slower: 1.66 vs 1.52 ms)
    case 96: return launch_cfg<3, 1, 1, 4, 7, true>(a, s);
    case 128: return launch_cfg<4, 1, 1, 4, 7, true>(a, s);
    default:
      set_error("fused ResidualUnit: channel count %d not in {64, 96, 128}", a.C_out);
      return FAC_ERR_ARG;
  }
}
}  // namespace fac
