// api.hip -- version, status strings and the per-device CU-count cache of libasd_hip.so.
#include "common.hpp"

#include <atomic>

namespace asd {

namespace {
constexpr int kMaxDevices = 64;
std::atomic<int> g_cus[kMaxDevices];  // 0 = not queried yet
}  // namespace

int current_device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    const int n = asd_device_cu_count(dev);
    return n > 0 ? n : 256;
}

}  // namespace asd

ASD_EXPORT int asd_version(void) {
    return ASD_VERSION_MAJOR * 10000 + ASD_VERSION_MINOR * 100 + ASD_VERSION_PATCH;
}

ASD_EXPORT const char* asd_status_string(int status) {
    switch (status) {
        case ASD_OK: return "ok";
        case ASD_ERR_INVALID_ARG: return "invalid argument";
        case ASD_ERR_UNSUPPORTED: return "unsupported size or dtype";
        case ASD_ERR_WORKSPACE: return "workspace too small or misaligned";
        case ASD_ERR_HIP: return "HIP runtime error";
        case ASD_ERR_ALIGNMENT: return "misaligned operand";
        default: return "unknown status";
    }
}

ASD_EXPORT int asd_device_cu_count(int device) {
    if (device < 0) return ASD_ERR_INVALID_ARG;
    if (device < asd::kMaxDevices) {
        const int c = asd::g_cus[device].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0)
        return ASD_ERR_HIP;
    if (device < asd::kMaxDevices) asd::g_cus[device].store(v, std::memory_order_relaxed);
    return v;
}
