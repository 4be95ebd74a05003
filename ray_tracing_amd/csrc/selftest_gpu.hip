// Device-side self-tests of hardware properties the kernels' shortcuts rely on.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rt_hip.h"
#include "fastdiv.hpp"
#include "fastexact.hpp"

namespace rt {

// Largest distance, in units in the last place, between v_rcp_f32(d) and the correctly rounded
// 1 / d over every float d with 2^-24 <= |d| <= 2^24 (the divisor range of fastdiv.hpp).
__global__ __launch_bounds__(256) void rcp_ulp_kernel(uint32_t first_bits, uint32_t last_bits, uint32_t* max_err) {
    uint32_t worst = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)first_bits + blockIdx.x * blockDim.x + threadIdx.x; b <= last_bits; b += stride) {
        for (uint32_t sign = 0; sign < 2; ++sign) {
            const float d = __uint_as_float((uint32_t)b | (sign << 31));
            const float exact = 1.0f / d, fast = __builtin_amdgcn_rcpf(d);
            const int32_t e = (int32_t)(__float_as_uint(exact) & 0x7FFFFFFFu), f = (int32_t)(__float_as_uint(fast) & 0x7FFFFFFFu);
            const bool same_sign = ((__float_as_uint(exact) ^ __float_as_uint(fast)) >> 31) == 0u;
            const uint32_t err = same_sign ? (uint32_t)(e > f ? e - f : f - e) : 0xFFFFFFFFu;
            worst = err > worst ? err : worst;
        }
    }
    atomicMax(max_err, worst);
}

// Every operand of fastexact.hpp's routines against the compiler's correctly rounded results:
// counts[0] square roots, counts[1] reciprocals (both signs), counts[2] n / 65535 that differ.
__global__ __launch_bounds__(256) void fastmath_kernel(uint32_t first_bits, uint32_t last_bits, unsigned long long* counts) {
    unsigned long long bad_sqrt = 0, bad_rcp = 0, bad_div = 0;
    const uint32_t stride = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t b = (uint64_t)first_bits + tid; b <= last_bits; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        bad_sqrt += __float_as_uint(fx_sqrt(x)) != __float_as_uint(__builtin_sqrtf(x));
        bad_rcp += __float_as_uint(fx_rcp(x)) != __float_as_uint(1.0f / x);
        bad_rcp += __float_as_uint(fx_rcp(-x)) != __float_as_uint(1.0f / -x);
    }
    for (uint32_t n = tid; n < 65536u; n += stride) {
        const float f = (float)n;
        bad_div += __float_as_uint(fx_div_65535(f)) != __float_as_uint(f / 65535.0f);
    }
    if (bad_sqrt) atomicAdd(&counts[0], bad_sqrt);
    if (bad_rcp) atomicAdd(&counts[1], bad_rcp);
    if (bad_div) atomicAdd(&counts[2], bad_div);
}

}  // namespace rt

extern "C" int rt_selftest_rcp_device(int device, uint32_t* max_ulp_error) {
    if (!max_ulp_error) return RT_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return RT_ERR_HIP;
    uint32_t* d_err = nullptr;
    if (hipMalloc(&d_err, sizeof(uint32_t)) != hipSuccess) return RT_ERR_HIP;
    int rc = RT_OK;
    const uint32_t first = 0x33800000u, last = 0x4B800000u;  // 2^-24 .. 2^24
    static_assert(rt::kDivLo == 5.9604644775390625e-08f && rt::kDivHi == 16777216.0f, "range of the divisor");
    if (hipMemset(d_err, 0, sizeof(uint32_t)) != hipSuccess) rc = RT_ERR_HIP;
    if (rc == RT_OK) {
        hipLaunchKernelGGL(rt::rcp_ulp_kernel, dim3(4096), dim3(256), 0, 0, first, last, d_err);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = RT_ERR_HIP;
    }
    if (rc == RT_OK && hipMemcpy(max_ulp_error, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = RT_ERR_HIP;
    (void)hipFree(d_err);
    return rc;
}

extern "C" int rt_selftest_fastmath_device(int device, uint64_t mismatches[3]) {
    if (!mismatches) return RT_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return RT_ERR_HIP;
    unsigned long long* d_counts = nullptr;
    if (hipMalloc(&d_counts, 3 * sizeof(unsigned long long)) != hipSuccess) return RT_ERR_HIP;
    int rc = RT_OK;
    if (hipMemset(d_counts, 0, 3 * sizeof(unsigned long long)) != hipSuccess) rc = RT_ERR_HIP;
    if (rc == RT_OK) {
        hipLaunchKernelGGL(rt::fastmath_kernel, dim3(4096), dim3(256), 0, 0, __builtin_bit_cast(uint32_t, rt::kFxLo),
                           __builtin_bit_cast(uint32_t, rt::kFxHi), d_counts);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = RT_ERR_HIP;
    }
    unsigned long long h[3] = {0, 0, 0};
    if (rc == RT_OK && hipMemcpy(h, d_counts, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) rc = RT_ERR_HIP;
    for (int k = 0; k < 3; ++k) mismatches[k] = h[k];
    (void)hipFree(d_counts);
    return rc;
}
