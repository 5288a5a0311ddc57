// wino_wgrad_kernel<2, 14> (conv_wino_wgrad_kernel.h): blocks of 2 x 14 tiles
#include "conv_wino_wgrad_kernel.h"

namespace vc {
int launch_wino_wgrad_2x14(hipStream_t st, const WinoWgArgs& a) { return launch_wino_wgrad<2, 14>(st, a); }
}  // namespace vc
