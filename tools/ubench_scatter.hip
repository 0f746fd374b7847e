// What does the memory side of the binning pass cost on gfx950?  N list entries, each: a returning atomicAdd on its tile's padded
// counter (the rank) and one 16-byte store to tile * bucket + rank (a random partial-line write).  Tiles drawn so that every tile gets
// about N / T entries, neighbouring threads hit different tiles (as in bin_count_kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_scatter.hip -o /tmp/ubench_scatter && /tmp/ubench_scatter
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void zero_k(unsigned* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
// MODE 0: atomics only (result consumed)   1: atomic -> dependent store   2: store only, rank = hash   3: store only, coalesced
// 4: non-returning atomic + store at a hashed rank
template <int MODE, int PER>
__global__ void scatter_k(unsigned* cnt, uint4* lists, int T, int bucket, int n_total, unsigned* sink) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    unsigned rank[PER], tile[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const unsigned e = gid * PER + k;
        tile[k] = (e * 2654435761u) % (unsigned)T;
        if (MODE == 0 || MODE == 1) rank[k] = e < (unsigned)n_total ? atomicAdd(&cnt[(size_t)tile[k] * 64], 1u) : 0u;
        else if (MODE == 4) { if (e < (unsigned)n_total) atomicAdd(&cnt[(size_t)tile[k] * 64], 1u); rank[k] = (e / (unsigned)T) % (unsigned)bucket; }
        else rank[k] = (e / (unsigned)T) % (unsigned)bucket;
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const unsigned e = gid * PER + k;
        if (e >= (unsigned)n_total) continue;
        if (MODE == 0) acc += rank[k];
        else if (MODE == 3) lists[e] = make_uint4(e, rank[k], tile[k], 0u);
        else if (rank[k] < (unsigned)bucket) lists[(size_t)tile[k] * bucket + rank[k]] = make_uint4(e, rank[k], tile[k], 0u);
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

template <int MODE, int PER>
float run(unsigned* cnt, size_t words, uint4* lists, int T, int bucket, int n, unsigned* sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float best = 1e9f;
    const int threads = 256, blocks = (n + threads * PER - 1) / (threads * PER);
    for (int rep = 0; rep < 6; rep++) {
        hipLaunchKernelGGL(zero_k, dim3((words + 255) / 256), dim3(256), 0, 0, cnt, words);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((scatter_k<MODE, PER>), dim3(blocks), dim3(threads), 0, 0, cnt, lists, T, bucket, n, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e3f;
}

int main() {
    const int T = 3225, bucket = 2048;
    const size_t words = (size_t)T * 64;
    unsigned *cnt, *sink;
    uint4* lists;
    if (hipMalloc(&cnt, words * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess ||
        hipMalloc(&lists, (size_t)T * bucket * sizeof(uint4)) != hipSuccess) return 1;
    for (int n : {710000, 3100000}) {
        printf("N = %d entries over %d tiles (bucket %d x 16 B), event time incl. ~6 us launch floor\n", n, T, bucket);
        printf("  returning atomics only, 1 per thread                 %8.1f us\n", run<0, 1>(cnt, words, lists, T, bucket, n, sink));
        printf("  returning atomics only, 4 in flight per thread       %8.1f us\n", run<0, 4>(cnt, words, lists, T, bucket, n, sink));
        printf("  atomic -> dependent scattered 16-B store, 1/thread   %8.1f us\n", run<1, 1>(cnt, words, lists, T, bucket, n, sink));
        printf("  atomic -> dependent scattered 16-B store, 4/thread   %8.1f us\n", run<1, 4>(cnt, words, lists, T, bucket, n, sink));
        printf("  scattered 16-B store only (rank from a hash), 4/thr  %8.1f us\n", run<2, 4>(cnt, words, lists, T, bucket, n, sink));
        printf("  non-returning atomic + scattered store, 4/thread     %8.1f us\n", run<4, 4>(cnt, words, lists, T, bucket, n, sink));
        printf("  coalesced 16-B store only, 4/thread                  %8.1f us\n", run<3, 4>(cnt, words, lists, T, bucket, n, sink));
    }
    return 0;
}
