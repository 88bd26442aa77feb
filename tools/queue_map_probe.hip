// Which hardware queue does the HIP runtime give the i-th stream of a process?  Creates N streams one after the other and launches
// one marker kernel per stream whose grid size is (i + 1) * 64 threads, so that rocprofv3 --kernel-trace shows stream -> Queue_Id:
//   hipcc --offload-arch=gfx950 -O2 tools/queue_map_probe.hip -o tools/queue_map_probe.bin
//   rocprofv3 --kernel-trace --output-format csv -d /tmp/qm -o t -- ./tools/queue_map_probe.bin 12 [destroy-first-k]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void marker(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1); }
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 12, drop = argc > 2 ? atoi(argv[2]) : 0;
    int* d = nullptr;
    (void)hipMalloc(&d, 4); (void)hipMemset(d, 0, 4);
    std::vector<hipStream_t> st(n);
    for (int i = 0; i < n; ++i) (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(marker, dim3(i + 1), dim3(64), 0, st[i], d);
    (void)hipDeviceSynchronize();
    // destroy the first `drop` streams, create as many again: where do the new ones land?  (grid sizes 100 + j)
    for (int i = 0; i < drop; ++i) (void)hipStreamDestroy(st[i]);
    std::vector<hipStream_t> again(drop);
    for (int i = 0; i < drop; ++i) (void)hipStreamCreateWithFlags(&again[i], hipStreamNonBlocking);
    for (int i = 0; i < drop; ++i) hipLaunchKernelGGL(marker, dim3(100 + i), dim3(64), 0, again[i], d);
    hipLaunchKernelGGL(marker, dim3(200), dim3(64), 0, 0, d);          // the null stream
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
