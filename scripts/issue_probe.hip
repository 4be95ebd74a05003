// Scratch micro-benchmark: SIMD cycles per instruction for hand-picked instruction snippets at
// 4 waves per SIMD (time-window method of ray_tracing_amd/csrc/probe_gpu.hip).  Development aid.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int ID> __device__ __forceinline__ void body(float (&r)[16], float a, float b, unsigned long long m0);

#define SNIPPET(ID, NAME, STMT)                                                                           \
    template <> __device__ __forceinline__ void body<ID>(float (&r)[16], float a, float b, unsigned long long m0) { \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { _Pragma("unroll") for (int k = 0; k < 16; ++k) { STMT; } }    \
    }

SNIPPET(0, "", asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(r[0]) : "v"(a)))
SNIPPET(1, "", asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(r[k & 1]) : "v"(a)))
SNIPPET(2, "", asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(r[k & 3]) : "v"(a)))
SNIPPET(3, "", asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(r[0]) : "v"(a)))
SNIPPET(4, "", asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(r[k & 1]) : "v"(a)))
SNIPPET(5, "", asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[0]) : "v"(a), "s"(m0)))
SNIPPET(6, "", asm volatile("v_cmp_lt_f32_e64 %2, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[0]) : "v"(a), "s"(m0)))
SNIPPET(7, "", asm volatile("v_cmp_lt_f32_e64 %2, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[k&3]) : "v"(a), "s"(m0)))
SNIPPET(8, "", asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&r[0]) : "v"(*(double*)&r[2])))
SNIPPET(9, "", asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(a), "v"(b)))
SNIPPET(10, "", asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(r[0]) : "v"(a)))
SNIPPET(11, "", asm volatile("v_mul_f32_e32 %0, %0, %1\n\tv_min_f32_e32 %0, %0, %2" : "+v"(r[0]) : "v"(a), "v"(b)))
SNIPPET(12, "", asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[0]) : "v"(a) : "vcc"))
SNIPPET(13, "", asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(r[k&7]), "+v"(r[8+(k&7)]) : "v"(a) : "vcc"))
SNIPPET(14, "", asm volatile("v_cmp_lt_f32_e32 vcc, %0, %3\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %3, vcc\n\tv_cndmask_b32_e32 %1, %1, %3, vcc\n\tv_cndmask_b32_e32 %2, %2, %3, vcc" : "+v"(r[k&3]), "+v"(r[4+(k&3)]), "+v"(r[8+(k&3)]) : "v"(a) : "vcc"))
SNIPPET(15, "", asm volatile("s_and_b64 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(a), "s"(m0), "s"(m0^5) : "vcc"))
SNIPPET(16, "", asm volatile("s_and_b64 vcc, %2, %3\n\tv_mul_f32_e32 %4, %4, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[k&7]) : "v"(a), "s"(m0), "s"(m0^5), "v"(r[8+(k&7)]) : "vcc"))
SNIPPET(17, "", asm volatile("s_and_b64 %3, %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(r[k]) : "v"(a), "s"(m0), "s"(m0^5)))
SNIPPET(18, "", asm volatile("s_and_saveexec_b64 %2, %3\n\tv_mov_b32_e32 %0, %1\n\ts_or_b64 exec, exec, %2" : "+v"(r[k]) : "v"(a), "s"(m0^9), "s"(m0)))
SNIPPET(19, "", asm volatile("s_and_saveexec_b64 %3, %4\n\tv_mov_b32_e32 %0, %2\n\tv_mov_b32_e32 %1, %2\n\ts_or_b64 exec, exec, %3" : "+v"(r[k&7]), "+v"(r[8+(k&7)]) : "v"(a), "s"(m0^9), "s"(m0)))
SNIPPET(20, "", asm volatile("s_mov_b64 exec, %2\n\tv_mov_b32_e32 %0, %1\n\ts_mov_b64 exec, -1" : "+v"(r[k]) : "v"(a), "s"(m0)))
SNIPPET(21, "", asm volatile("v_ashrrev_i32_e32 %0, 31, %0" : "+v"(r[k])))
SNIPPET(22, "", asm volatile("v_lshlrev_b32_e32 %0, 6, %0" : "+v"(r[k])))
SNIPPET(23, "", asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n\tv_cmp_gt_f32_e64 %3, %0, %2\n\ts_and_b64 %1, %1, %3" : : "v"(r[k]), "s"(m0), "v"(a), "s"(m0^7)))
constexpr int kN = 24;
const char* kNames[kN] = {"mul dep chain 1", "mul dep chain 2", "mul dep chain 4", "min dep chain 1", "min dep chain 2", "cnd64 dep chain",
  "cmp64,nop,cnd64 dep (2)", "cmp64,nop,cnd64 4 chains (2)", "pk_mul dep", "fma dep", "add_u32 dep", "mul,min dep (2)", "cmp32,nop,cnd32 dep (2)",
  "cmp32,nop,2cnd32 (3)", "cmp32,nop,3cnd32 (4)", "s_and vcc,nop,cnd32 (1v)", "s_and vcc,mul,cnd32 (2v)", "s_and,nop,cnd64 (1v)",
  "saveexec,mov,or (1v)", "saveexec,2mov,or (2v)", "s_mov exec,mov,restore (1v)", "ashrrev 31", "lshlrev 6", "2cmp64+s_and (2v)"};
const int kPerStmt[kN] = {1,1,1,1,1,1,2,2,1,1,1,2,2,3,4,1,2,1,1,2,1,1,1,2};

template <int ID>
__global__ __launch_bounds__(256) void probe(uint32_t ticks, unsigned long long* stamps, float* sink) {
    extern __shared__ uint32_t pad[];
    float r[16];
    const float seedv = 1.0f + (float)threadIdx.x * 1.0e-3f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = seedv + (float)k;
    const float a = 1.0000001f + seedv * 1e-9f, b = 1.0e-7f * seedv;
    unsigned long long m0 = __ballot(threadIdx.x & 1);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long w1 = w0, bodies = 0;
    do {
#pragma unroll 1
        for (int i = 0; i < 8; ++i) body<ID>(r, a, b, m0);
        bodies += 8;
        w1 = __builtin_amdgcn_s_memrealtime();
    } while (w1 - w0 < ticks);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)__popcll(m0);
#pragma unroll
    for (int k = 0; k < 16; ++k) s += r[k];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = blockIdx.x * 4 + threadIdx.x / 64;
        stamps[4 * w] = bodies; stamps[4 * w + 1] = t1 - t0; stamps[4 * w + 2] = w1 - w0; stamps[4 * w + 3] = w0;
    }
}

template <int ID> void run(int W, unsigned long long* d_st, float* d_sink, int n_cus) {
    const size_t lds = W == 1 ? 163840 / 2 + 1024 : (163840 / W) / 1024 * 1024 - 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<ID>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nb = n_cus * W, nw = nb * 4;
    for (int pass = 0; pass < 2; ++pass) hipLaunchKernelGGL(probe<ID>, dim3(nb), dim3(256), lds, 0, 150000u, d_st, d_sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw * 4);
    hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long total = 0, first = ~0ull, last = 0;
    std::vector<double> clk(nw), span(nw);
    for (int w = 0; w < nw; ++w) {
        total += h[4 * w]; span[w] = (double)h[4 * w + 2]; clk[w] = (double)h[4 * w + 1] / h[4 * w + 2] * 100.0;
        first = std::min(first, h[4 * w + 3]); last = std::max(last, h[4 * w + 3]);
    }
    std::sort(clk.begin(), clk.end()); std::sort(span.begin(), span.end());
    const double insts = (double)total * 64 * kPerStmt[ID];
    const double cpi = span[nw / 2] / 100.0 * clk[nw / 2] * (n_cus * 4) / insts;
    printf("%-22s W=%d: %6.3f cycles/inst  (%.0f MHz, starts within %.1f us)\n", kNames[ID], W, cpi, clk[nw / 2], (last - first) / 100.0);
    fflush(stdout);
}

template <int ID> void run_all(unsigned long long* d_st, float* d_sink, int n_cus) {
    run<ID>(4, d_st, d_sink, n_cus);
    run<ID>(1, d_st, d_sink, n_cus);
    if constexpr (ID + 1 < kN) run_all<ID + 1>(d_st, d_sink, n_cus);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cus = prop.multiProcessorCount;
    unsigned long long* d_st; float* d_sink;
    hipMalloc(&d_st, (size_t)n_cus * 8 * 4 * 32); hipMalloc(&d_sink, (size_t)n_cus * 8 * 256 * 4);
    run_all<0>(d_st, d_sink, n_cus);
    return 0;
}
