// Does the lane -> address mapping inside a wave-contiguous 1 KB matter?  8 wavefronts per CU stream tiles of 2 KB with
// 16-byte loads, (a) lane l reads double2 l (linear), (b) lane l reads double2 (l & 15) * 4 + (l >> 4): the operand
// fetch pattern of the tile-packed factor (consecutive lanes 64 B apart, every quarter-wave touches all eight 128-byte
// lines of the 1 KB run).  Working sets: 128 KB per CU (L2 resident) and 16 MB per CU (beyond the caches).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PERM, int UNR>
__global__ __launch_bounds__(512) void stream(const double2* __restrict__ src, double* out, size_t n2, int reps) {
    const double2* p = src + (size_t)blockIdx.x * n2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int idx = PERM ? ((lane & 15) * 4 + (lane >> 4)) : lane;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)wv * 64 * UNR; i + 64 * UNR <= n2; i += 8 * 64 * UNR) {
            double2 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = p[i + u * 64 + idx];
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y;
        }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
int main() {
    double2* src; double* out;
    const size_t big = 16u << 20;
    hipMalloc(&src, 256 * big); hipMemset(src, 0, 256 * big); hipMalloc(&out, 256 * 512 * 8);
    for (size_t bytes : {(size_t)128 << 10, big}) {
        const size_t n2 = bytes / 16;
        const int reps = bytes < big ? 256 : 2;
        auto run = [&](auto kern, const char* name) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, src, out, n2, reps);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, src, out, n2, reps); hipEventRecord(e1);
            hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tot = (double)bytes * reps;
            printf("%-26s set %6zu KB/CU: %.1f GB/s per CU (%.1f B/clk @2.3GHz), %.2f TB/s chip\n", name, bytes >> 10,
                   tot / ms / 1e6, tot / (ms * 2.3e6), 256 * tot / ms / 1e9);
        };
        run(stream<0, 4>, "linear, 4 in flight");
        run(stream<1, 4>, "tile-permuted, 4 in flight");
        run(stream<0, 8>, "linear, 8 in flight");
        run(stream<1, 8>, "tile-permuted, 8 in flight");
    }
    return 0;
}
