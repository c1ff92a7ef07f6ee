// Instantiations of the MFMA conv kernel for the 96x128 (C_out x T) workgroup tile.
#include "conv1d_mfma.h"

namespace fac {
int conv_dispatch_96x128(ConvArgs& a, hipStream_t s) {
  switch (a.KV) {
    case 1: return launch_cfg<3,1,1,4, 1>(a, s);
    case 2: return launch_cfg<3,1,1,4, 2>(a, s);
    case 7: return launch_cfg<3,1,1,4, 7>(a, s);
    default: return launch_cfg<3,1,1,4, 0>(a, s);
  }
}
}  // namespace fac
