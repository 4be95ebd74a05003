// Device micro-probe: how many cycles of one SIMD a wave64 vector instruction takes, per class of
// instruction, at a chosen number of resident waves per SIMD.  The trace kernel is bound by
// vector-instruction issue, not by bytes; this probe is what prices that bound: it gives the
// instructions per second the chip can issue for the instruction mix of the hot steps, against
// which the kernel's own instruction rate is set.
//
// Method: every wave runs, for a fixed span of time, an unrolled body of 64 independent
// instructions of the class (inline assembly, so the opcodes are exactly the named ones) and
// counts the bodies.  Workgroups of 256 threads put one wave on each of a CU's four SIMDs; each
// workgroup asks for 1/W of the CU's LDS, so exactly W of them fit on a CU and a grid of W x CUs
// workgroups is resident all at once with W waves on every SIMD.  Cycles per instruction per
// SIMD = span x clock x SIMDs / instructions issued by all waves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/rt_hip.h"

namespace rt {

typedef float pf2 __attribute__((ext_vector_type(2)));

#define RT_P1(OP, k) asm volatile(OP " %0, %0, %1" : "+v"(r[k]) : "v"(a))
#define RT_P2(OP, k) asm volatile(OP " %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b))
#define RT_PK1(OP, k) asm volatile(OP " %0, %0, %1" : "+v"(p[k]) : "v"(pa))
#define RT_PK2(OP, k) asm volatile(OP " %0, %0, %1, %2" : "+v"(p[k]) : "v"(pa), "v"(pb))

#define RT_S1(k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sg[k]) : : "scc")

template <int MIX>
__device__ __forceinline__ void probe_body(float (&r)[16], pf2 (&p)[8], float a, float b, pf2 pa, pf2 pb,
                                           unsigned long long& m0, unsigned long long& m1, uint32_t (&sg)[8]) {
    if constexpr (MIX == RT_VALU_MIX_SALU) {
        // nothing but scalar instructions, eight independent chains
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) RT_S1(k);
    } else if constexpr (MIX == RT_VALU_MIX_VALU_SALU) {
        // full-rate vector and scalar instructions in turn, all independent: do the two pipes of a SIMD
        // take instructions side by side (from different waves), or one after the other?
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                RT_P1("v_mul_f32_e32", k);
                RT_S1(k & 7);
            }
    } else if constexpr (MIX == RT_VALU_MIX_SLOW_SALU) {
        // the same with half-rate vector instructions (selects on a scalar mask)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "s"(m0));
                RT_S1(k & 7);
            }
    } else if constexpr (MIX == RT_VALU_MIX_FMA) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P2("v_fma_f32", k);
    } else if constexpr (MIX == RT_VALU_MIX_MUL_ADD) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P1("v_mul_f32_e32", k);
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P1("v_add_f32_e32", k);
        }
    } else if constexpr (MIX == RT_VALU_MIX_PK_MUL_ADD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 8; ++k) RT_PK1("v_pk_mul_f32", k);
#pragma unroll
            for (int k = 0; k < 8; ++k) RT_PK1("v_pk_add_f32", k);
        }
    } else if constexpr (MIX == RT_VALU_MIX_PK_FMA) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) RT_PK2("v_pk_fma_f32", k);
    } else if constexpr (MIX == RT_VALU_MIX_MIN_MAX) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P1("v_min_f32_e32", k);
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P1("v_max_f32_e32", k);
        }
    } else if constexpr (MIX == RT_VALU_MIX_CNDMASK) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "s"(m0));
    } else if constexpr (MIX == RT_VALU_MIX_CMP) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & 1) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(r[k]), "v"(a));
                else asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(r[k]), "v"(b));
            }
    } else if constexpr (MIX == RT_VALU_MIX_INT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P1("v_add_u32_e32", k);
#pragma unroll
            for (int k = 0; k < 8; ++k) RT_P1("v_xor_b32_e32", k);
#pragma unroll
            for (int k = 8; k < 16; ++k) RT_P1("v_and_b32_e32", k);
        }
    } else if constexpr (MIX == RT_VALU_MIX_TRANS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(r[k]));
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(r[k]));
        }
    } else if constexpr (MIX == RT_VALU_MIX_MIN3_MAX3) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P2("v_min3_f32", k);
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P2("v_max3_f32", k);
        }
    } else if constexpr (MIX == RT_VALU_MIX_CNDMASK_VCC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // vcc comes from a vector compare, as in compiled code (one compare per 15 selects)
            asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(r[0]), "v"(a) : "vcc");
#pragma unroll
            for (int k = 1; k < 16; ++k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(a));
        }
    } else if constexpr (MIX == RT_VALU_MIX_CMP_VCC) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(r[k]), "v"(a) : "vcc");
    } else if constexpr (MIX == RT_VALU_MIX_MOV) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(r[k]) : "v"(r[(k + 1) & 15]));
    } else if constexpr (MIX == RT_VALU_MIX_SHIFT_CVT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(r[k]));
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(r[k]));
        }
    } else if constexpr (MIX == RT_VALU_MIX_MUL_LO) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P1("v_mul_lo_u32", k);
    } else if constexpr (MIX == RT_VALU_MIX_DPP) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[k]) : "v"(r[(k + 1) & 15]));
    } else if constexpr (MIX == RT_VALU_MIX_BFI) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[k]) : "v"(a), "v"(b));
    } else if constexpr (MIX == RT_VALU_MIX_AND_OR) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) RT_P2("v_and_or_b32", k);
    } else {
        // The BOX step's proportions: of 64 instructions 24 packed multiply / add, 12 min / max,
        // 8 compares into scalar masks, 12 selects, 8 integer operations.
        static_assert(MIX == RT_VALU_MIX_BOX_STEP, "unknown mix");
#pragma unroll
        for (int k = 0; k < 8; ++k) RT_PK1("v_pk_add_f32", k);
#pragma unroll
        for (int k = 0; k < 8; ++k) RT_PK1("v_pk_mul_f32", k);
#pragma unroll
        for (int k = 0; k < 8; ++k) RT_PK1("v_pk_add_f32", k);
#pragma unroll
        for (int k = 0; k < 6; ++k) RT_P1("v_min_f32_e32", k);
#pragma unroll
        for (int k = 6; k < 12; ++k) RT_P1("v_max_f32_e32", k);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k & 1) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(r[k]), "v"(a));
            else asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(r[k]), "v"(b));
        }
#pragma unroll
        for (int k = 0; k < 12; ++k)
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "s"(k & 1 ? m1 : m0));
#pragma unroll
        for (int k = 8; k < 16; ++k) RT_P1("v_add_u32_e32", k);
    }
}

#undef RT_S1
#undef RT_P1
#undef RT_P2
#undef RT_PK1
#undef RT_PK2

// Every wave runs the body until `ticks` ticks of the 100 MHz clock have passed since its own
// start (all waves of the grid start within a microsecond of each other, so they share the
// window) and reports how many bodies it got through: the SIMD's arbiter favours older waves,
// so equal work per wave would end with the SIMD half empty, while equal time keeps all W waves
// resident for the whole measurement.  stamps[wave] = {bodies, shader-clock cycles, ticks}, in a
// buffer of their own; `sink` only keeps the arithmetic alive.
template <int MIX>
__global__ __launch_bounds__(256) void valu_probe_kernel(uint32_t ticks, unsigned long long* stamps, float* sink) {
    extern __shared__ uint32_t occupancy_pad[];  // never touched: its size limits the workgroups per CU
    float r[16];
    pf2 p[8];
    const float seedv = 1.0f + (float)threadIdx.x * 1.0e-3f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = seedv + (float)k;
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = pf2{seedv + (float)k, seedv - (float)k};
    const float a = 1.0000001f, b = 1.0e-7f;
    const pf2 pa = {1.0000001f, 0.9999999f}, pb = {1.0e-7f, -1.0e-7f};
    unsigned long long m0 = __ballot(threadIdx.x & 1), m1 = __ballot(threadIdx.x & 2);
    uint32_t sg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sg[k] = blockIdx.x + (uint32_t)k;

    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long w1 = w0, bodies = 0;
    do {
#pragma unroll 1
        for (int i = 0; i < 8; ++i) probe_body<MIX>(r, p, a, b, pa, pb, m0, m1, sg);
        bodies += 8;
        w1 = __builtin_amdgcn_s_memrealtime();
    } while (w1 - w0 < ticks);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);

    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += r[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += p[k].x + p[k].y;
    s += (float)__popcll(m0 ^ m1);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (float)sg[k];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t wave = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u;
        stamps[4 * wave] = bodies;
        stamps[4 * wave + 1] = t1 - t0;
        stamps[4 * wave + 2] = w1 - w0;
        stamps[4 * wave + 3] = w0;
    }
}

template <int MIX>
hipError_t launch_probe(uint32_t n_blocks, size_t lds, uint32_t ticks, unsigned long long* stamps, float* sink) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(valu_probe_kernel<MIX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(valu_probe_kernel<MIX>, dim3(n_blocks), dim3(256), lds, 0, ticks, stamps, sink);
    return hipGetLastError();
}

}  // namespace rt

extern "C" int rt_probe_valu_issue(int device, int mix, uint32_t waves_per_simd, uint32_t duration_us, rt_valu_probe* out) {
    using namespace rt;
    if (!out || waves_per_simd < 1 || waves_per_simd > 8 || duration_us == 0 || duration_us > 1000000u || mix < 0 ||
        mix >= RT_VALU_MIX_COUNT)
        return RT_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return RT_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RT_ERR_HIP;
    const uint32_t n_cus = (uint32_t)prop.multiProcessorCount;
    const uint32_t n_blocks = n_cus * waves_per_simd, n_waves = n_blocks * 4u;
    const uint32_t ticks = duration_us * 100u;
    // exactly `waves_per_simd` workgroups fit the CU's LDS (the allocation granule is well below 1 KiB)
    const size_t lds_cu = prop.maxSharedMemoryPerMultiProcessor >= 65536 ? prop.maxSharedMemoryPerMultiProcessor : 163840;
    const size_t lds = waves_per_simd == 1 ? lds_cu / 2 + 1024 : (lds_cu / waves_per_simd) / 1024 * 1024 - 1024;
    unsigned long long* d_stamps = nullptr;
    float* d_sink = nullptr;
    int rc = RT_OK;
    if (hipMalloc(&d_stamps, (size_t)n_waves * 32) != hipSuccess) return RT_ERR_HIP;
    if (hipMalloc(&d_sink, (size_t)n_blocks * 256 * 4) != hipSuccess) rc = RT_ERR_HIP;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (rc == RT_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = RT_ERR_HIP;
    float ms = 0.0f;
    for (int pass = 0; pass < 2 && rc == RT_OK; ++pass) {  // the first pass warms the clocks and the code cache
        if (hipEventRecord(e0, 0) != hipSuccess) rc = RT_ERR_HIP;
        hipError_t e = hipErrorInvalidValue;
        switch (mix) {
#define RT_CASE(M) case M: e = launch_probe<M>(n_blocks, lds, ticks, d_stamps, d_sink); break
            RT_CASE(RT_VALU_MIX_FMA);
            RT_CASE(RT_VALU_MIX_MUL_ADD);
            RT_CASE(RT_VALU_MIX_PK_MUL_ADD);
            RT_CASE(RT_VALU_MIX_PK_FMA);
            RT_CASE(RT_VALU_MIX_MIN_MAX);
            RT_CASE(RT_VALU_MIX_CNDMASK);
            RT_CASE(RT_VALU_MIX_CMP);
            RT_CASE(RT_VALU_MIX_INT);
            RT_CASE(RT_VALU_MIX_TRANS);
            RT_CASE(RT_VALU_MIX_BOX_STEP);
            RT_CASE(RT_VALU_MIX_MIN3_MAX3);
            RT_CASE(RT_VALU_MIX_CNDMASK_VCC);
            RT_CASE(RT_VALU_MIX_CMP_VCC);
            RT_CASE(RT_VALU_MIX_MOV);
            RT_CASE(RT_VALU_MIX_SHIFT_CVT);
            RT_CASE(RT_VALU_MIX_MUL_LO);
            RT_CASE(RT_VALU_MIX_DPP);
            RT_CASE(RT_VALU_MIX_BFI);
            RT_CASE(RT_VALU_MIX_AND_OR);
            RT_CASE(RT_VALU_MIX_SALU);
            RT_CASE(RT_VALU_MIX_VALU_SALU);
            RT_CASE(RT_VALU_MIX_SLOW_SALU);
#undef RT_CASE
        }
        if (e != hipSuccess || hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) rc = RT_ERR_HIP;
        if (rc == RT_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = RT_ERR_HIP;
    }
    std::vector<unsigned long long> h((size_t)n_waves * 4);
    if (rc == RT_OK && hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = RT_ERR_HIP;
    if (rc == RT_OK) {
        std::vector<double> clk(n_waves), span(n_waves), bodies(n_waves);
        unsigned long long first = ~0ull, last = 0, total = 0;
        for (uint32_t w = 0; w < n_waves; ++w) {
            bodies[w] = (double)h[4 * w];
            total += h[4 * w];
            span[w] = (double)h[4 * w + 2];
            clk[w] = h[4 * w + 2] ? (double)h[4 * w + 1] / (double)h[4 * w + 2] * 100.0 : 0.0;
            first = std::min(first, h[4 * w + 3]);
            last = std::max(last, h[4 * w + 3]);
        }
        std::sort(clk.begin(), clk.end());
        std::sort(span.begin(), span.end());
        std::sort(bodies.begin(), bodies.end());
        out->valu_insts = total * 64ull;
        out->waves = n_waves;
        out->simds = n_cus * 4u;
        out->window_us = span[n_waves / 2] / 100.0;
        out->start_spread_us = (double)(last - first) / 100.0;
        out->clock_mhz = clk[n_waves / 2];
        out->kernel_ms = ms;
        out->wave_share_min = bodies[0] * n_waves / (double)total;
        out->wave_share_max = bodies[n_waves - 1] * n_waves / (double)total;
        // SIMD cycles spent = window x clock on each of the SIMDs; instructions issued = all of them
        out->simd_cycles_per_inst = out->window_us * out->clock_mhz * (double)out->simds / (double)out->valu_insts;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_stamps);
    (void)hipFree(d_sink);
    return rc;
}
