// wino_wgrad_kernel<4, 8> (conv_wino_wgrad_kernel.h): blocks of 4 x 8 tiles
#include "conv_wino_wgrad_kernel.h"

namespace vc {
int launch_wino_wgrad_4x8(hipStream_t st, const WinoWgArgs& a) { return launch_wino_wgrad<4, 8>(st, a); }
}  // namespace vc
