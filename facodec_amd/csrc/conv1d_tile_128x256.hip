// Instantiations of the MFMA conv kernel for the 128x256 (C_out x T) workgroup tile: 8 MFMA waves
// (2 x 4, 64x64 each) + 4 staging waves, one workgroup per CU -- half the weight-slab traffic per FLOP
// of the 128x128 tile, for long sequences.
#include "conv1d_mfma.h"

namespace fac {
int conv_dispatch_128x256(ConvArgs& a, hipStream_t s) {
  switch (a.KV) {
    case 1: return launch_cfg<2,2,2,4, 1>(a, s);
    case 2: return launch_cfg<2,2,2,4, 2>(a, s);      // all-phases ConvTranspose1d (row_phases)
    case 7: return launch_cfg<2,2,2,4, 7>(a, s);
    default: return launch_cfg<2,2,2,4, 0>(a, s);
  }
}
// 96 x 256: 8 MFMA waves of 96 x 32 each (C_out = 96 / 192 layers of the decoder).
int conv_dispatch_96x256(ConvArgs& a, hipStream_t s) {
  switch (a.KV) {
    case 1: return launch_cfg<3,1,1,8, 1>(a, s);
    default: return launch_cfg<3,1,1,8, 0>(a, s);
  }
}
// 128 x 160: four MFMA waves stacked along C_out, each 32 rows x 160 columns -- the latent-rate layers
// of a 2 s clip (160 frames) fit one tile exactly.
int conv_dispatch_128x160(ConvArgs& a, hipStream_t s) {
  switch (a.KV) {
    case 1: return launch_cfg<1,5,4,1, 1>(a, s);
    case 2: return launch_cfg<1,5,4,1, 2>(a, s);
    case 3: return launch_cfg<1,5,4,1, 3>(a, s);
    case 7: return launch_cfg<1,5,4,1, 7>(a, s);
    default: return launch_cfg<1,5,4,1, 0>(a, s);
  }
}
}  // namespace fac
