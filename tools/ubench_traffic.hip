// Calibration of the memory-side request counters (TCC_EA0_RDREQ / WRREQ, MI355X_MICROARCH.md "HBM") on access patterns of KNOWN size,
// the patterns of gaussian_tail_kernel's Adam phase among them:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_traffic.hip -o /tmp/ubt
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d /tmp/t1 -o p --output-format csv -- /tmp/ubt
//   rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d /tmp/t2 -o p --output-format csv -- /tmp/ubt
// Every kernel prints the bytes it reads and writes by construction; tools/pmc_summary.py gives the requests per launch.
//   stream_copy_f4      : float4 per lane, coalesced, 256 MB in + 256 MB out
//   rows192_rw          : ROWS random rows of 192 B (48 floats) of a [P, 48] array, read as float4s (12 lanes per row) from three
//                         arrays and written back to the same rows (the SH pass of the Adam phase: p, m, v in and out)
//   rows12_rw           : ROWS random rows of 12 B of a [P, 3] array, element per lane, three arrays in and out (xyz / scaling pass)
//   rows16_rw, rows4_rw : the same for [P, 4] float4 rows (rotation) and [P] single floats (opacity)
//   gather64_r          : N random 64-byte records read as four float4s by one lane each (the partial-record gather of phase B)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stream_copy_f4(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
// float4 f of the row list: row f / 12, piece f % 12
__global__ void rows192_rw(const int* __restrict__ rows, int n_rows, float* a, float* b, float* c) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_rows * 12) return;
    const size_t e = (size_t)rows[f / 12] * 48 + 4 * (f % 12);
    float4 x = *reinterpret_cast<float4*>(a + e), y = *reinterpret_cast<float4*>(b + e), z = *reinterpret_cast<float4*>(c + e);
    x.x += 1.f, y.y += 1.f, z.z += 1.f;
    *reinterpret_cast<float4*>(a + e) = x, *reinterpret_cast<float4*>(b + e) = y, *reinterpret_cast<float4*>(c + e) = z;
}
template <int LEN>
__global__ void rows_small_rw(const int* __restrict__ rows, int n_rows, float* a, float* b, float* c) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_rows * LEN) return;
    const size_t e = (size_t)rows[f / LEN] * LEN + (f % LEN);
    a[e] += 1.f, b[e] += 1.f, c[e] += 1.f;
}
__global__ void rows16_rw(const int* __restrict__ rows, int n_rows, float4* a, float4* b, float4* c) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_rows) return;
    const size_t e = (size_t)rows[f];
    float4 x = a[e], y = b[e], z = c[e];
    x.x += 1.f, y.y += 1.f, z.z += 1.f;
    a[e] = x, b[e] = y, c[e] = z;
}
__global__ void gather64_r(const int* __restrict__ idx, int n, const float4* __restrict__ recs, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* r = recs + (size_t)idx[i] * 4;
    const float4 a = r[0], b = r[1], c = r[2], d = r[3];
    if (a.x + b.y + c.z + d.w == 12345.678f) out[0] = 1.f;
}

int main() {
    const int P = 500000, ROWS = 197000, NREC = 1000000, NSLOT = 2400000;
    std::mt19937 rng(7);
    std::vector<int> perm(P);
    std::iota(perm.begin(), perm.end(), 0);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int> rows(perm.begin(), perm.begin() + ROWS);
    std::sort(rows.begin(), rows.end());  // visible Gaussians in index order (a block's rows are near each other, not adjacent)
    std::vector<int> ridx(NREC);
    for (auto& v : ridx) v = (int)(rng() % NSLOT);
    int *d_rows, *d_ridx;
    CK(hipMalloc(&d_rows, ROWS * 4)); CK(hipMalloc(&d_ridx, NREC * 4));
    CK(hipMemcpy(d_rows, rows.data(), ROWS * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ridx, ridx.data(), NREC * 4, hipMemcpyHostToDevice));
    const size_t NS = (size_t)256 << 20;
    float *s_in, *s_out, *a48, *b48, *c48, *a4, *b4, *c4, *recs, *flag;
    CK(hipMalloc(&s_in, NS)); CK(hipMalloc(&s_out, NS));
    CK(hipMalloc(&a48, (size_t)P * 192)); CK(hipMalloc(&b48, (size_t)P * 192)); CK(hipMalloc(&c48, (size_t)P * 192));
    CK(hipMalloc(&a4, (size_t)P * 16)); CK(hipMalloc(&b4, (size_t)P * 16)); CK(hipMalloc(&c4, (size_t)P * 16));
    CK(hipMalloc(&recs, (size_t)NSLOT * 64)); CK(hipMalloc(&flag, 256));
    CK(hipMemset(s_in, 0, NS)); CK(hipMemset(a48, 0, (size_t)P * 192)); CK(hipMemset(b48, 0, (size_t)P * 192)); CK(hipMemset(c48, 0, (size_t)P * 192));
    CK(hipMemset(a4, 0, (size_t)P * 16)); CK(hipMemset(b4, 0, (size_t)P * 16)); CK(hipMemset(c4, 0, (size_t)P * 16)); CK(hipMemset(recs, 0, (size_t)NSLOT * 64));
    for (int rep = 0; rep < 3; rep++) {
        // (a 512 MB stream between the patterns pushes earlier lines out of the Infinity Cache, as a whole iteration does between two tails)
        stream_copy_f4<<<(unsigned)((NS / 16 + 255) / 256), 256>>>((const float4*)s_in, (float4*)s_out, NS / 16);
        rows192_rw<<<(ROWS * 12 + 255) / 256, 256>>>(d_rows, ROWS, a48, b48, c48);
        stream_copy_f4<<<(unsigned)((NS / 16 + 255) / 256), 256>>>((const float4*)s_in, (float4*)s_out, NS / 16);
        rows_small_rw<3><<<(ROWS * 3 + 255) / 256, 256>>>(d_rows, ROWS, a4, b4, c4);
        rows16_rw<<<(ROWS + 255) / 256, 256>>>(d_rows, ROWS, (float4*)a4, (float4*)b4, (float4*)c4);
        rows_small_rw<1><<<(ROWS + 255) / 256, 256>>>(d_rows, ROWS, a4, b4, c4);
        gather64_r<<<(NREC + 255) / 256, 256>>>(d_ridx, NREC, (const float4*)recs, flag);
    }
    CK(hipDeviceSynchronize());
    printf("bytes by construction (read / write) per launch:\n");
    printf("stream_copy_f4   %zu / %zu\n", NS, NS);
    printf("rows192_rw       %zu / %zu   (%d rows x 192 B x 3 arrays)\n", (size_t)ROWS * 192 * 3, (size_t)ROWS * 192 * 3, ROWS);
    printf("rows_small_rw<3> %zu / %zu   (%d rows x 12 B x 3 arrays)\n", (size_t)ROWS * 12 * 3, (size_t)ROWS * 12 * 3, ROWS);
    printf("rows16_rw        %zu / %zu   (%d rows x 16 B x 3 arrays)\n", (size_t)ROWS * 16 * 3, (size_t)ROWS * 16 * 3, ROWS);
    printf("rows_small_rw<1> %zu / %zu   (%d rows x 4 B x 3 arrays)\n", (size_t)ROWS * 4 * 3, (size_t)ROWS * 4 * 3, ROWS);
    printf("gather64_r       %zu / 0   (%d records x 64 B, random over %d slots)\n", (size_t)NREC * 64, NREC, NSLOT);
    return 0;
}
