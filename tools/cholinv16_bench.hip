// Diagnostic: the 16 x 16 "factor and invert" step of the chain wavefront (qp_resident.hpp) alone on one wavefront -- the
// row-wise Gauss-Jordan against the MFMA-blocked form: cycles per call (s_memtime) and the error of both against a host
// computation.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ihybrid-drt_amd/csrc tools/cholinv16_bench.hip -o /tmp/cholinv16_bench
#include "qp_resident.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using namespace hipdrt;

__global__ void bench(const double* Din, double* Wout, unsigned long long* out, int reps, int mode) {
    __shared__ double dsc[16 * DLD];
    __shared__ double U[32 * PLD];
    OpsResidentT<false, 512> ops;
    ops.sm.dsc = dsc; ops.sm.U = U;
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) dsc[(i / 16) * DLD + i % 16] = Din[i];
    for (int i = lane; i < 32 * PLD; i += 64) U[i] = -77.0;
    __syncthreads();
    bool ok = true;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        ok = (mode ? ops.cholinv16_blocked(0, 0) : ops.cholinv16_rows(0, 0)) && ok;
        __builtin_amdgcn_wave_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    for (int i = lane; i < 256; i += 64) Wout[i] = U[(i / 16) * PLD + i % 16];
    if (lane == 0) { out[0] = (t1 - t0) / reps; out[1] = ok ? 1 : 0; }
}

int main() {
    std::mt19937_64 rng(3);
    std::normal_distribution<double> nd;
    std::vector<double> A(16 * 20), D(256), L(256, 0.0), W(256, 0.0);
    for (auto& v : A) v = nd(rng);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = (i == j) ? 0.5 : 0.0;
            for (int k = 0; k < 20; ++k) s += A[i * 20 + k] * A[j * 20 + k];
            D[i * 16 + j] = s;
        }
    for (int j = 0; j < 16; ++j) {                     // host reference: L, then W = L^-1
        double s = D[j * 16 + j];
        for (int k = 0; k < j; ++k) s -= L[j * 16 + k] * L[j * 16 + k];
        L[j * 16 + j] = std::sqrt(s);
        for (int i = j + 1; i < 16; ++i) {
            double t = D[i * 16 + j];
            for (int k = 0; k < j; ++k) t -= L[i * 16 + k] * L[j * 16 + k];
            L[i * 16 + j] = t / L[j * 16 + j];
        }
    }
    for (int c = 0; c < 16; ++c)
        for (int i = c; i < 16; ++i) {
            double t = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) t -= L[i * 16 + k] * W[k * 16 + c];
            W[i * 16 + c] = t / L[i * 16 + i];
        }
    double *dD, *dW; unsigned long long* dO;
    hipMalloc(&dD, 256 * 8); hipMalloc(&dW, 256 * 8); hipMalloc(&dO, 16);
    hipMemcpy(dD, D.data(), 256 * 8, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<double> Wd(256);
        unsigned long long o[2];
        hipLaunchKernelGGL(bench, dim3(1), dim3(64), 0, 0, dD, dW, dO, 1000, mode);
        hipDeviceSynchronize();
        hipMemcpy(Wd.data(), dW, 256 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(o, dO, 16, hipMemcpyDeviceToHost);
        double err = 0, mx = 0, up = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                mx = std::fmax(mx, std::fabs(W[i * 16 + j]));
                if (j <= i) err = std::fmax(err, std::fabs(Wd[i * 16 + j] - W[i * 16 + j]));
                else up = std::fmax(up, std::fabs(Wd[i * 16 + j]));
            }
        printf("%s: %llu cycles per call, ok %llu, max |W - W_host| / max |W| = %.2e, upper triangle max %.1e\n",
               mode ? "blocked (MFMA)" : "by rows       ", o[0], o[1], err / mx, up);
    }
    return 0;
}
