// Issue cost of the cross-lane moves the KS stepper's halo exchange uses, measured with one wave per SIMD (the C2
// regime): v_mov_b32 (plain), v_mov_b32_dpp row_ror:1 (inside a 16-lane row), v_mov_b32_dpp wave_ror:1 (across the whole
// wave), fp64 FMA for scale.  8 independent registers are cycled so that no instruction waits for its predecessor.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/dpp_rate tools/micro/dpp_rate.hip && /tmp/dpp_rate
#include <hip/hip_runtime.h>

#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(64) rate_kernel(int* out, int iters) {
    int r[8];
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r[i] = threadIdx.x * 8 + i;
        d[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MODE == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[(i + 1) & 7]));
                if constexpr (MODE == 1) r[i] = __builtin_amdgcn_mov_dpp(r[i], 0x121, 0xf, 0xf, false);   // row_ror:1
                if constexpr (MODE == 2) r[i] = __builtin_amdgcn_mov_dpp(r[i], 0x13C, 0xf, 0xf, false);   // wave_ror:1
                if constexpr (MODE == 3) d[i] = __builtin_fma(d[i], 1.0000001, 1e-12);
            }
        }
    }
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= r[i] ^ (int)d[i];
    if (acc == 0x7fffffff) out[0] = acc;
}

template <int MODE>
double run(const char* name, int* d_out) {
    const int iters = 20000, waves = 1024;           // 1 wave per SIMD on 256 CUs
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(waves), dim3(64), 0, 0, d_out, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(waves), dim3(64), 0, 0, d_out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr = (double)iters * 64;
    const double ns_per = ms * 1e6 / instr;
    printf("%-28s %8.3f ms  %6.3f ns per wave-instruction  (= %.2f cycles at 2.4 GHz)\n", name, ms, ns_per, ns_per * 2.4);
    return ns_per;
}

int main() {
    int* d_out;
    hipMalloc(&d_out, 4);
    run<0>("v_mov_b32", d_out);
    run<1>("v_mov_b32_dpp row_ror:1", d_out);
    run<2>("v_mov_b32_dpp wave_ror:1", d_out);
    run<3>("v_fma_f64", d_out);
    return 0;
}
