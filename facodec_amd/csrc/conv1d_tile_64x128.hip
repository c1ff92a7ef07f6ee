// Instantiations of the MFMA conv kernel for the 64x128 (C_out x T) workgroup tile.
#include "conv1d_mfma.h"

namespace fac {
int conv_dispatch_64x128(ConvArgs& a, hipStream_t s) {
  switch (a.KV) {
    case 1: return launch_cfg<2,1,1,4, 1>(a, s);
    case 7: return launch_cfg<2,1,1,4, 7>(a, s);
    default: return launch_cfg<2,1,1,4, 0>(a, s);
  }
}
}  // namespace fac
