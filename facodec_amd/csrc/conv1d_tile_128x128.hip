// Instantiations of the MFMA conv kernel for the 128x128 (C_out x T) workgroup tile.
#include "conv1d_mfma.h"

namespace fac {
int conv_dispatch_128x128(ConvArgs& a, hipStream_t s) {
  switch (a.KV) {
    case 1: return launch_cfg<2,2,2,2, 1>(a, s);
    case 2: return launch_cfg<2,2,2,2, 2>(a, s);
    case 3: return launch_cfg<2,2,2,2, 3>(a, s);
    case 4: return launch_cfg<2,2,2,2, 4>(a, s);
    case 5: return launch_cfg<2,2,2,2, 5>(a, s);
    case 7: return launch_cfg<2,2,2,2, 7>(a, s);
    case 10: return launch_cfg<2,2,2,2, 10>(a, s);
    case 12: return launch_cfg<2,2,2,2, 12>(a, s);
    default: return launch_cfg<2,2,2,2, 0>(a, s);
  }
}
}  // namespace fac
