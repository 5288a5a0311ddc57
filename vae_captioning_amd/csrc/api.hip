// ABI bookkeeping: version, last-error buffer, device check.
#include "common.h"
#include "vaecap.h"

namespace vc {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace vc

extern "C" int vc_abi_version(void) { return VC_ABI_VERSION; }
extern "C" const char* vc_last_error(void) { return vc::last_error_buf(); }

extern "C" int vc_device_check(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return vc::fail(e != hipSuccess ? (int)e : vc::VC_EINVAL, "%s: no HIP device visible", __func__);
    if (device < 0 || device >= n) return vc::fail(vc::VC_EINVAL, "%s: device index out of range", __func__);
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return vc::fail((int)e, "%s: hipGetDeviceProperties failed", __func__);
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return vc::fail(vc::VC_EINVAL, "%s: device is not gfx950 (MI355X): %s", __func__, (long)0) ;
    return 0;
}
