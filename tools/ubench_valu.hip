// VALU issue-rate microbenchmark for gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
// Prints, per instruction kind and per resident-waves-per-SIMD target, the SIMD cycles one wave64 instruction costs
// (assuming 2.4 GHz).  Used to decide what "VALU-issue bound" means for the blend kernels (profiles/README.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
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
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc[i] + acc2[i].x + acc2[i].y;
    for (int i = 0; i < 8; i++) s += (float)accd[i];
    s += (float)(sdump[0] + sdump[1] + sdump[2] + sdump[3]);
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 20000;
    printf("%-28s", name);
    for (int w : {1, 2, 4, 8}) {
        const int grid = 1024 * w;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 100, 1.0001f, 1e-6f);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0001f, 1e-6f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double cycles = ms * 1e-3 * 2.4e9;
        const double instr_per_simd = (double)w * iters * 16;  // wave instructions issued on one SIMD
        printf("  w=%d: %5.2f cyc/instr", w, cycles / instr_per_simd);
    }
    printf("\n");
}

int main() {
    float* out;
    if (hipMalloc(&out, 64) != hipSuccess) return 1;
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
