// Checks the wave-wide DPP scans of kernels.hip (wave_scan.hpp) against a serial scan.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/dpp_scan_test.hip -o scripts/_build/dpp_scan_test
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../ray_tracing_amd/csrc/wave_scan.hpp"

__global__ void k(const uint32_t* in, uint32_t* sum, uint32_t* mx) {
    uint32_t v = in[blockIdx.x * 64 + threadIdx.x];
    sum[blockIdx.x * 64 + threadIdx.x] = rt::wave_incl_sum(v);
    mx[blockIdx.x * 64 + threadIdx.x] = rt::wave_incl_max(v);
}

int main() {
    const int waves = 4096, n = waves * 64;
    std::vector<uint32_t> h(n), s(n), m(n);
    uint32_t seed = 12345;
    for (int i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        uint32_t r = seed >> 8;
        h[i] = (i / 64) % 3 == 0 ? (r % 5 == 0 ? r % 17 : 0) : r % 1000;
    }
    uint32_t *d_in, *d_s, *d_m;
    if (hipMalloc(&d_in, n * 4) != hipSuccess || hipMalloc(&d_s, n * 4) != hipSuccess || hipMalloc(&d_m, n * 4) != hipSuccess) return 2;
    hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d_in, d_s, d_m);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(s.data(), d_s, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), d_m, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int w = 0; w < waves; ++w) {
        uint32_t acc = 0, mx = 0;
        for (int l = 0; l < 64; ++l) {
            acc += h[w * 64 + l];
            mx = h[w * 64 + l] > mx ? h[w * 64 + l] : mx;
            if (s[w * 64 + l] != acc || m[w * 64 + l] != mx) ++bad;
        }
    }
    printf("dpp scan mismatches: %ld of %d\n", bad, n);
    return bad ? 1 : 0;
}
