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
    {
        snprintf(vc::last_error_buf(), 512, "%s: device %d is %s, not gfx950 (MI355X)", __func__, device, p.gcnArchName);
        return vc::VC_EINVAL;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// roctx ranges (tracing): named ranges around the phases of a step, visible to `rocprofv3 --marker-trace`.
// the roctx library is bound at run time (no link-time dependency; without it the calls are no-ops that still return 0).
#include <dlfcn.h>

#include <mutex>
namespace vc {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
static Roctx* roctx() {
    static Roctx r;
    static std::once_flag once;
    std::call_once(once, [] {
        // rocprofv3 (rocprofiler-sdk) sees the ranges of ITS roctx library; the roctracer-era libroctx64 is the fallback
        const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1",
                               "libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so.4"};
        void* h = nullptr;
        for (int i = 0; !h && i < 6; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            *(void**)(&r.push) = dlsym(h, "roctxRangePushA");
            *(void**)(&r.pop) = dlsym(h, "roctxRangePop");
        }
    });
    return (r.push && r.pop) ? &r : nullptr;
}
}  // namespace vc

extern "C" int vc_trace_available(void) { return vc::roctx() ? 1 : 0; }
extern "C" int vc_trace_push(const char* name) {
    VC_CHECK_ARG(name, "null name");
    if (vc::Roctx* r = vc::roctx()) r->push(name);
    return 0;
}
extern "C" int vc_trace_pop(void) {
    if (vc::Roctx* r = vc::roctx()) r->pop();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// CRC-32C (polynomial 0x1EDC6F41, reflected 0x82F63B78), slicing-by-8, host only.
namespace vc {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
};
}  // namespace vc

extern "C" int vc_host_crc32c(const void* data, size_t nbytes, uint32_t* crc_inout) {
    VC_CHECK_ARG(crc_inout && (data || nbytes == 0), "null pointer");
    static const vc::Crc32cTables T;
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = ~*crc_inout;
    while (nbytes && ((uintptr_t)p & 7)) { c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xff]; --nbytes; }
    while (nbytes >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = T.t[7][w & 0xff] ^ T.t[6][(w >> 8) & 0xff] ^ T.t[5][(w >> 16) & 0xff] ^ T.t[4][(w >> 24) & 0xff] ^
            T.t[3][(w >> 32) & 0xff] ^ T.t[2][(w >> 40) & 0xff] ^ T.t[1][(w >> 48) & 0xff] ^ T.t[0][(w >> 56) & 0xff];
        p += 8; nbytes -= 8;
    }
    while (nbytes--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xff];
    *crc_inout = ~c;
    return 0;
}
