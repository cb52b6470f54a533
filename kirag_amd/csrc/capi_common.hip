// Error plumbing and device selection shared by every entry point of libkirag_amd.so.
#include "common.hpp"

#include <atomic>
#include <cstring>

namespace kr {

std::string& last_error_ref() {
    static thread_local std::string e;
    return e;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

std::atomic<int> g_force_exact{0};
std::atomic<int> g_byte_prescan{1};
std::atomic<int> g_byte_min_rows{1 << 19};
std::atomic<int> g_eps8_permille{1000};
std::atomic<unsigned long long> g_va_retired_bias{0};
std::atomic<unsigned long long> g_vmm_min_reserve{16ull << 30};

bool is_device_pointer(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // unregistered (pageable) host memory
    return at.type == hipMemoryTypeDevice;
}

int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(KR_ENODEV, "no HIP device available (%s); libkirag_amd has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(KR_EINVAL, "device %d out of range [0,%d)", device, n);
    static thread_local int checked = -1;
    if (checked != device) {
        hipDeviceProp_t p;
        e = hipGetDeviceProperties(&p, device);
        if (e != hipSuccess) return fail(KR_EHIP, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
            return fail(KR_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", device, p.gcnArchName);
        checked = device;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(KR_EHIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
    return 0;
}

}  // namespace kr

extern "C" {
int kr_abi_version(void) { return KR_ABI_VERSION; }
const char* kr_last_error(void) { return kr::last_error_ref().c_str(); }
int kr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int kr_set_option(const char* name, int value) {
    if (name && std::strcmp(name, "force_exact_scores") == 0) { kr::g_force_exact.store(value != 0); return 0; }
    if (name && std::strcmp(name, "byte_prescan") == 0) { kr::g_byte_prescan.store(value != 0); return 0; }
    if (name && std::strcmp(name, "debug_byte_min_rows") == 0) { kr::g_byte_min_rows.store(value < 0 ? (1 << 19) : value); return 0; }
    if (name && std::strcmp(name, "debug_eps8_permille") == 0) { kr::g_eps8_permille.store(value <= 0 ? 1000 : value); return 0; }
    if (name && std::strcmp(name, "debug_va_retired_tib") == 0) { kr::g_va_retired_bias.store((unsigned long long)(value < 0 ? 0 : value) << 40); return 0; }
    if (name && std::strcmp(name, "debug_vmm_min_reserve_mib") == 0) { kr::g_vmm_min_reserve.store(value <= 0 ? (16ull << 30) : ((unsigned long long)value << 20)); return 0; }
    return kr::fail(KR_EINVAL, "unknown option '%s'", name ? name : "(null)");
}
int kr_format_ids(const int64_t* ids, int64_t n, char sep, char* out, int64_t cap, int64_t* written) {
    if (n < 0 || (n > 0 && (!ids || !out)) || !written) return kr::fail(KR_EINVAL, "bad arguments");
    char* p = out;
    char* const end = out + cap;
    for (int64_t i = 0; i < n; ++i) {
        if (end - p < 21) return kr::fail(KR_EINVAL, "kr_format_ids: output buffer of %lld bytes is too small (21 per id always suffice)", (long long)cap);
        if (i) *p++ = sep;
        const int64_t v = ids[i];
        uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;     // |INT64_MIN| as unsigned: no overflow
        char tmp[20];
        int len = 0;
        do { tmp[len++] = (char)('0' + (int)(u % 10ull)); u /= 10ull; } while (u);
        if (v < 0) *p++ = '-';
        while (len) *p++ = tmp[--len];
    }
    *written = (int64_t)(p - out);
    return 0;
}
}
