// What is the VALU issue roof of one MI355X, by the wall clock?  N resident waves per SIMD, each a long run of independent
// v_fma_f32 (8 accumulators); timed with HIP events around the launch, no in-kernel clock involved.
//   wave-instructions / s / SIMD  ->  cycles per wave64 instruction at the clock the run held (rocm-smi sclk ~2.4 GHz under load).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fma_peak.hip -o /tmp/fp && /tmp/fp
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PACKED>
__global__ __launch_bounds__(256) void fma_k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
        if (PACKED) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pa), "v"(pb));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
            }
        }
    }
    float s = PACKED ? p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y : x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (s == 12345.678f) out[0] = s;
}

template <int PACKED>
void run(float* out, int waves_per_simd) {
    const int iters = 20000, per_iter = 32;  // VALU instructions per wave per loop trip
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks: 4 waves = one per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_k<PACKED>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 1e-6f);
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fma_k<PACKED>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double winstr = (double)blocks * 4 * iters * per_iter;  // wave-instructions
    const double per_simd_per_s = winstr / (best * 1e-3) / 1024.0;
    const double tflops = winstr * 64 * 2 * (PACKED ? 2 : 1) / (best * 1e-3) / 1e12;
    printf("%-14s %d waves/SIMD: %7.3f ms  %6.1f TFLOP/s  %.3f G wave-instr/s/SIMD = %.2f cycles per wave64 instruction at 2.4 GHz\n",
           PACKED ? "v_pk_fma_f32" : "v_fma_f32", waves_per_simd, best, tflops, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    float* out;
    if (hipMalloc(&out, 64) != hipSuccess) return 1;
    for (int w : {1, 2, 4, 8}) run<0>(out, w);
    for (int w : {1, 2, 4, 8}) run<1>(out, w);
    return 0;
}
