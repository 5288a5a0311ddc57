// wino_wgrad_kernel<4, 7> (conv_wino_wgrad_kernel.h): blocks of 4 x 7 tiles
#include "conv_wino_wgrad_kernel.h"

namespace vc {
int launch_wino_wgrad_4x7(hipStream_t st, const WinoWgArgs& a) { return launch_wino_wgrad<4, 7>(st, a); }
}  // namespace vc
