// Returning device-scope atomic adds on gfx950: what do N of them cost, by address pattern?  (Round-2 question: bin_count_kernel spends
// 40 us in its tile-histogram atomics whether it issues 0.7 M or 23 k of them.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_atomic.hip -o /tmp/ubench_atomic && /tmp/ubench_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void zero_k(unsigned* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
// every thread issues `per` returning atomics (all in flight together), to counter (hash(thread, k) % n_counters) * stride
template <int PER>
__global__ void atom_k(unsigned* cnt, int n_counters, int stride, unsigned* sink, int active_per_block) {
    if ((int)threadIdx.x >= active_per_block) return;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const unsigned h = (gid * 2654435761u + k * 40503u) % (unsigned)n_counters;
        r[k] = atomicAdd(&cnt[(size_t)h * stride], 1u);
    }
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) s += r[k];
    if (s == 0xffffffffu) sink[0] = s;
}

// the same with 64-bit counters (a pair of adjacent 32-bit tile counters taken by one atomic)
template <int PER>
__global__ void atom64_k(unsigned long long* cnt, int n_counters, int stride, unsigned* sink, int active_per_block) {
    if ((int)threadIdx.x >= active_per_block) return;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long r[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const unsigned h = (gid * 2654435761u + k * 40503u) % (unsigned)n_counters;
        r[k] = atomicAdd(&cnt[(size_t)h * stride], 0x100000001ull);
    }
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) s += r[k];
    if (s == ~0ull) sink[0] = (unsigned)s;
}

template <int PER>
float run64(unsigned long long* cnt, size_t words, int blocks, int threads, int active, int n_counters, int stride, unsigned* sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(zero_k, dim3((words + 255) / 256), dim3(256), 0, 0, (unsigned*)cnt, words);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(atom64_k<PER>, dim3(blocks), dim3(threads), 0, 0, cnt, n_counters, stride, sink, active);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e3f;
}

template <int PER>
float run(unsigned* cnt, size_t words, int blocks, int threads, int active, int n_counters, int stride, unsigned* sink, bool zero_first) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        if (zero_first) hipLaunchKernelGGL(zero_k, dim3((words + 255) / 256), dim3(256), 0, 0, cnt, words);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(atom_k<PER>, dim3(blocks), dim3(threads), 0, 0, cnt, n_counters, stride, sink, active);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e3f;
}

int main() {
    const int T = 3225;
    const size_t words = (size_t)T * 64;
    unsigned *cnt, *sink;
    if (hipMalloc(&cnt, words * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    printf("%-72s %10s\n", "case (returning atomicAdd, device scope)", "us");
    struct C { const char* name; int blocks, threads, active, per, ncnt, stride; bool zero; };
    std::vector<C> cases = {
        {"empty-ish: 1954 blocks x 256 thr, 1 active thread, 1 atomic", 1954, 256, 1, 1, T, 64, true},
        {"0.5 M atomics: 1954 x 256 x 1, 3225 counters 256 B apart", 1954, 256, 256, 1, T, 64, true},
        {"2 M atomics: 1954 x 256 x 4, 3225 counters 256 B apart", 1954, 256, 256, 4, T, 64, true},
        {"2 M atomics, same but counters packed (4 B apart)", 1954, 256, 256, 4, T, 1, true},
        {"2 M atomics, 206 k counters 4 B apart (no address shared much)", 1954, 256, 256, 4, (int)words, 1, true},
        {"23 k atomics: 770 blocks x 30 active threads x 1", 770, 256, 30, 1, T, 64, true},
        {"23 k atomics, no zeroing kernel in front", 770, 256, 30, 1, T, 64, false},
        {"215 k atomics: 1954 blocks x 110 active x 1", 1954, 256, 110, 1, T, 64, true},
        {"0.5 M atomics on ONE counter", 1954, 256, 256, 1, 1, 64, true},
    };
    for (auto& c : cases) {
        float us = c.per == 1 ? run<1>(cnt, words, c.blocks, c.threads, c.active, c.ncnt, c.stride, sink, c.zero)
                              : run<4>(cnt, words, c.blocks, c.threads, c.active, c.ncnt, c.stride, sink, c.zero);
        printf("%-72s %10.1f\n", c.name, us);
    }
    // 64-bit: 1612 pair counters 256 B apart
    printf("%-72s %10.1f\n", "2 M 64-bit atomics: 1954 x 256 x 4, 1612 pair counters 256 B apart",
           run64<4>((unsigned long long*)cnt, words, 1954, 256, 256, 1612, 32, sink));
    printf("%-72s %10.1f\n", "1 M 64-bit atomics: 977 x 256 x 4, 1612 pair counters 256 B apart",
           run64<4>((unsigned long long*)cnt, words, 977, 256, 256, 1612, 32, sink));
    printf("%-72s %10.1f\n", "1 M 32-bit atomics: 977 x 256 x 4, 3225 counters 256 B apart", run<4>(cnt, words, 977, 256, 256, T, 64, sink, true));
    printf("%-72s %10.1f\n", "0.5 M 64-bit atomics: 1954 x 256 x 1, 1612 pair counters", run64<1>((unsigned long long*)cnt, words, 1954, 256, 256, 1612, 32, sink));
    return 0;
}
