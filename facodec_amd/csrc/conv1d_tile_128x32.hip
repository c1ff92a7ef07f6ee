// Instantiations of the MFMA conv kernel for the 128x32 (C_out x T) workgroup tile.
#include "conv1d_mfma.h"

namespace fac {
int conv_dispatch_128x32(ConvArgs& a, hipStream_t s) {
  switch (a.KV) {
    case 1: return launch_cfg<1,1,4,1, 1>(a, s);
    default: return launch_cfg<1,1,4,1, 0>(a, s);
  }
}
}  // namespace fac
