// common.hpp -- host-side helpers shared by the launchers of libasd_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asd_hip.h"

#define ASD_EXPORT extern "C" __attribute__((visibility("default")))

namespace asd {

inline int launch_status() {
    return hipGetLastError() == hipSuccess ? ASD_OK : ASD_ERR_HIP;
}

inline bool aligned_to(const void* p, size_t a) {
    return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0;
}

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int dtype_size(int dtype) {
    switch (dtype) {
        case ASD_DTYPE_F32: return 4;
        case ASD_DTYPE_BF16:
        case ASD_DTYPE_F16: return 2;
        default: return 0;
    }
}

// number of CUs of the CURRENT device (cached per device id; the query is slow)
int current_device_cus();

}  // namespace asd
