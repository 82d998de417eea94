// Times conv_gemm_kernel<128, 0> alone on the widest layer of the 8x8 training batch (512 boards:
// M = 32768 rows, 64 -> 128 channels, nine taps), optionally with parts of it compiled out
// (-DAZG_TRAIN_EXP=bits, see azg_train.hpp).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I alphazero_gomoku_amd/csrc -o conv_gemm_time tests/micro/conv_gemm_time.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "azg_train.hpp"
using namespace azg;

int main()
{
    const int n = 512, Nb = 8, S = 64, Cs = 64, N = 128, M = n * S, K = 9 * Cs;
    float *src, *w, *bias, *out;
    hipMalloc(&src, (size_t)M * Cs * 4);
    hipMalloc(&w, (size_t)N * K * 4);
    hipMalloc(&bias, N * 4);
    hipMalloc(&out, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * Cs);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, N * 4);
    ConvGemmArgs a{src, w, bias, nullptr, out, nullptr, M, N, Cs, K, 9, S, Nb, N};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid((M + TBM - 1) / TBM, 1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((conv_gemm_kernel<128, 0>), grid, dim3(256), 0, 0, a);
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((conv_gemm_kernel<128, 0>), grid, dim3(256), 0, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / reps;
    printf("exp %3d  %7.2f us per launch  %6.1f TFLOP/s (as if complete)\n", AZG_TRAIN_EXP, us, 2.0 * M * N * K / us * 1e-6);
    return hipGetLastError() != hipSuccess;
}
