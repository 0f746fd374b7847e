// VALU issue-rate microbenchmark for gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
// Prints, per instruction kind and per resident-waves-per-SIMD, the SIMD cycles one wave64 instruction costs.
// Round-2 method (round 1's table was contaminated by launch overhead and an assumed 2.4 GHz clock): every wave stamps
// s_memtime (shader-clock cycles) and s_memrealtime (constant 100 MHz) around its own loop, the kernel is one 256-thread
// workgroup per CU slot (4 waves = one per SIMD) x w workgroups per CU, and the figure is
//     median over waves of  (delta s_memtime) / (instructions of the wave x w)
// = cycles the SIMD spends per wave-instruction when w waves share it; the clock actually held is printed beside it
// (delta s_memtime / delta s_memrealtime x 100 MHz).  The `half` column repeats w = 2 with EXEC = the lower 32 lanes only: does
// a wave64 instruction whose upper half is idle cost a SIMD-32 one pass instead of two?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b, unsigned long long* stamps, int half_exec) {
    float acc[16];
    f2 acc2[16];
    const f2 a2 = {a, a}, b2 = {b, b};
    double accd[8];
    const double ad = (double)a;
    int sdump[4] = {0, 0, 0, 0};
    const unsigned long long smask = 0x5555555555555555ull;
    const int idx = ((threadIdx.x + 1) & 63) * 4;
    for (int i = 0; i < 8; i++) accd[i] = (double)i;
    if (MODE == 23) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
    for (int i = 0; i < 16; i++) acc[i] = (float)threadIdx.x * 1e-3f + i, acc2[i] = f2{acc[i], acc[i] + 1.f};
    unsigned long long saved_exec = 0;
    if (half_exec) asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 0xffffffff" : "=s"(saved_exec));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {  // independent v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (MODE == 1) {  // independent v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
            REP16(X)
#undef X
        } else if (MODE == 2) {  // dependent v_fma_f32 chain
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[0]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (MODE == 3) {  // independent v_exp_f32
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i]));
            REP16(X)
#undef X
        } else if (MODE == 4) {  // independent v_add_f32 with DPP row_ror:8
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]));
            REP16(X)
#undef X
        } else if (MODE == 5) {  // v_permlane32_swap
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]));
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
        } else if (MODE == 6) {  // v_cndmask_b32 (VCC operand)
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == 7) {  // v_pk_add_f32
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc2[i]) : "v"(a2));
            REP16(X)
#undef X
        } else if (MODE == 8) {  // v_rcp_f32
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i]));
            REP16(X)
#undef X
        } else if (MODE == 9) {  // dependent chain of mul -> add pairs across two accumulators (ILP 2)
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i & 1]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (MODE == 10) {  // ILP 4
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (MODE == 11) {  // v_cndmask_b32 e64, SGPR-pair mask
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "s"(smask));
            REP16(X)
#undef X
        } else if (MODE == 12) {  // v_cmp_gt_f32 -> vcc
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(acc[i]), "v"(a) : "vcc");
            REP16(X)
#undef X
        } else if (MODE == 13) {  // v_cmp + v_cndmask pairs (8 pairs = 16 instructions)
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : "vcc");
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
        } else if (MODE == 14) {  // v_and_b32
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == 15) {  // v_max_f32
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == 16) {  // v_med3_f32
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (MODE == 17) {  // v_readfirstlane_b32
#define X(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sdump[i & 3]) : "v"(acc[i]));
            REP16(X)
#undef X
        } else if (MODE == 18) {  // v_mov_b32 dpp quad_perm
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(acc[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if (MODE == 19) {  // v_permlane16_swap
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]));
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
        } else if (MODE == 20) {  // v_mul_f32
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == 21) {  // v_add_f64
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(accd[i & 7]) : "v"(ad));
            REP16(X)
#undef X
        } else if (MODE == 22) {  // v_cvt_f64_f32
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(accd[i & 7]) : "v"(acc[i]));
            REP16(X)
#undef X
        } else if (MODE == 23) {  // v_cndmask_b32 e32 vcc, vcc written once before the loop
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == 24) {  // v_add_f32 with a literal-free SGPR operand
#define X(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i]) : "s"(a));
            REP16(X)
#undef X
        } else if (MODE == 25) {  // v_ashrrev_i32
#define X(i) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(acc[i]));
            REP16(X)
#undef X
        } else if (MODE == 26) {  // ds_bpermute_b32 (LDS crossbar, no memory)
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(acc[i]) : "v"(idx));
            REP16(X)
#undef X
        } else if (MODE == 27) {  // v_readlane_b32 with an SGPR lane select
#define X(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sdump[i & 3]) : "v"(acc[i]));
            REP16(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (half_exec) asm volatile("s_mov_b64 exec, %0" : : "s"(saved_exec));
    if (stamps != nullptr && (threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0, stamps[2 * w + 1] = r1 - r0;
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc[i] + acc2[i].x + acc2[i].y;
    for (int i = 0; i < 8; i++) s += (float)accd[i];
    s += (float)(sdump[0] + sdump[1] + sdump[2] + sdump[3]);
    if (s == 123.456f) out[0] = s;
}

#include <algorithm>
static unsigned long long* g_stamps = nullptr;
static std::vector<unsigned long long> g_host;

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 40000;  // x 16 instructions: ~1-3 ms per wave, launch overhead invisible; stamped inside the kernel anyway
    printf("%-28s", name);
    double clk = 0.0;
    for (int pass = 0; pass < 5; pass++) {
        const int w = pass < 4 ? (1 << pass) : 2;
        const int half = pass == 4;
        const int grid = 256 * w;  // 256-thread workgroups: 4 waves = one per SIMD; w workgroups per CU
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 200, 1.0001f, 1e-6f, (unsigned long long*)nullptr, half);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f, g_stamps, half);
        (void)hipDeviceSynchronize();
        const size_t nw = (size_t)grid * 4;
        (void)hipMemcpy(g_host.data(), g_stamps, nw * 16, hipMemcpyDeviceToHost);
        std::vector<double> cyc(nw), mhz(nw);
        for (size_t i = 0; i < nw; i++) {
            cyc[i] = (double)g_host[2 * i] / ((double)iters * 16 * w);
            mhz[i] = (double)g_host[2 * i] / (double)g_host[2 * i + 1] * 100.0;
        }
        std::nth_element(cyc.begin(), cyc.begin() + nw / 2, cyc.end());
        std::nth_element(mhz.begin(), mhz.begin() + nw / 2, mhz.end());
        if (!half) printf("  w=%d: %5.2f", w, cyc[nw / 2]);
        else printf("  half(w=2): %5.2f", cyc[nw / 2]);
        clk = mhz[nw / 2];
    }
    printf("  cyc/instr   clock %4.0f MHz\n", clk);
}

int main() {
    float* out;
    if (hipMalloc(&out, 64) != hipSuccess) return 1;
    if (hipMalloc(&g_stamps, 2048 * 4 * 16) != hipSuccess) return 1;
    g_host.resize(2048 * 4 * 2);
    run<0>("v_fma_f32 indep", out);
    run<2>("v_fma_f32 dependent", out);
    run<9>("v_fma_f32 ILP2", out);
    run<10>("v_fma_f32 ILP4", out);
    run<1>("v_pk_fma_f32 indep", out);
    run<7>("v_pk_add_f32 indep", out);
    run<3>("v_exp_f32 indep", out);
    run<8>("v_rcp_f32 indep", out);
    run<4>("v_add_f32 dpp row_ror", out);
    run<5>("v_permlane32_swap", out);
    run<6>("v_cndmask_b32 vcc(unset)", out);
    run<23>("v_cndmask_b32 vcc(set once)", out);
    run<11>("v_cndmask_b32_e64 sgpr", out);
    run<12>("v_cmp_gt_f32 vcc", out);
    run<13>("v_cmp+v_cndmask pairs", out);
    run<14>("v_and_b32", out);
    run<25>("v_ashrrev_i32", out);
    run<15>("v_max_f32", out);
    run<16>("v_med3_f32", out);
    run<20>("v_mul_f32", out);
    run<24>("v_add_f32 sgpr operand", out);
    run<17>("v_readfirstlane_b32", out);
    run<27>("v_readlane_b32", out);
    run<18>("v_mov_b32 dpp quad_perm", out);
    run<19>("v_permlane16_swap", out);
    run<26>("ds_bpermute_b32 + wait", out);
    run<21>("v_add_f64", out);
    run<22>("v_cvt_f64_f32", out);
    (void)hipDeviceSynchronize();
    return 0;
}
