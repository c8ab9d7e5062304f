// runtime.hip -- error state, device checks, arenas, uploads.
#include "runtime.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[1024] = "";

void f5_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int f5_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* f5_last_error(void) { return g_err; }
extern "C" int f5_version(void) { return F5HIP_VERSION; }

extern "C" int f5_device_count(char* name_out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    int usable = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) != hipSuccess) continue;
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) {
            if (usable == 0 && name_out) snprintf(name_out, 64, "%s", prop.gcnArchName);
            ++usable;
        }
    }
    if (usable == 0 && name_out) name_out[0] = 0;
    return usable;
}

int f5_check_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
        (void)hipGetLastError();
        return f5_fail(F5_ENODEVICE, "no HIP device available: libf5hip has no CPU fallback");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return f5_fail(F5_ENODEVICE, "cannot query HIP device %d", dev);
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return f5_fail(F5_ENODEVICE, "device %d is %s; libf5hip is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    return 0;
}

int f5_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

int DevArena::alloc(void** out, size_t bytes, bool zero) {
    *out = nullptr;
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) / 256 * 256;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return f5_fail(F5_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    if (zero) {
        e = hipMemset(p, 0, bytes);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return f5_fail(F5_EHIP, "hipMemset failed: %s", hipGetErrorString(e));
        }
    }
    ptrs.push_back(p);
    total += bytes;
    *out = p;
    return 0;
}
void DevArena::release() {
    for (void* p : ptrs) (void)hipFree(p);
    ptrs.clear();
    total = 0;
}

uint16_t f5_f32_to_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
size_t f5_elem_size(int precision) { return precision == F5_PREC_BF16 ? 2 : 4; }

int f5_upload_f32(DevArena& a, const float* host, size_t count, float** out) {
    F5_TRY(a.alloc_t(out, count, false));
    F5_HIP(hipMemcpy(*out, host, count * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
int f5_upload_t(DevArena& a, int precision, const float* host, size_t count, void** out) {
    if (precision == F5_PREC_FP32) return f5_upload_f32(a, host, count, (float**)out);
    std::vector<uint16_t> tmp(count);
    for (size_t i = 0; i < count; ++i) tmp[i] = f5_f32_to_bf16_bits(host[i]);
    F5_TRY(a.alloc(out, count * 2, false));
    F5_HIP(hipMemcpy(*out, tmp.data(), count * 2, hipMemcpyHostToDevice));
    return 0;
}

int f5_slot_set(SlotMap& slots, const char* name, const float* host, const int64_t* shape, int ndim) {
    auto it = slots.find(name);
    if (it == slots.end()) return f5_fail(F5_EINVAL, "unknown tensor name '%s'", name);
    TensorSlot& s = it->second;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != s.numel()) {
        std::string want;
        for (auto d : s.shape) want += std::to_string(d) + ",";
        return f5_fail(F5_EINVAL, "tensor '%s': got %lld elements, expected shape [%s]", name, (long long)n, want.c_str());
    }
    s.host.assign(host, host + n);
    s.set = true;
    return 0;
}
int f5_slots_all_set(const SlotMap& slots) {
    for (auto& kv : slots)
        if (!kv.second.set) return f5_fail(F5_ESTATE, "tensor '%s' was never set", kv.first.c_str());
    return 0;
}
