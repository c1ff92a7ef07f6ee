// Instantiations of the MFMA conv kernel for the 32x256 (C_out x T) workgroup tile.
#include "conv1d_mfma.h"

namespace fac {
int conv_dispatch_32x256(ConvArgs& a, hipStream_t s) {
  switch (a.KV) {
    case 1: return launch_cfg<1,2,1,4, 1>(a, s);
    case 3: return launch_cfg<1,2,1,4, 3>(a, s);      // (3, 3) convs of the spectrogram discriminator (virtual channels)
    case 7: return launch_cfg<1,2,1,4, 7>(a, s);
    case 9: return launch_cfg<1,2,1,4, 9>(a, s);      // its (3, 9) convs
    default: return launch_cfg<1,2,1,4, 0>(a, s);
  }
}
}  // namespace fac
