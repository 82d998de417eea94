// How fast one short launch of v_mfma_f32_32x32x2f32 runs on this chip: the ceiling a 30-60 us
// float32 GEMM kernel of the training step is measured against.  Each wave issues `iters` groups
// of four independent MFMAs (the 2x2 tile arrangement of the GEMM kernels); no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));

// VALU > 0: that many independent integer additions (two interleaved dependency chains) issued by the
// same wave behind every matrix instruction: whether vector arithmetic hides in the matrix pipeline's shadow.
template <int LDSOPS, int VALU = 0>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float a0, float b0)
{
    int c0 = threadIdx.x, c1 = blockIdx.x;
    __shared__ float sh[4096];
    f16v acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = (float)i;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        float x0 = a, x1 = a, y0 = b, y1 = b;
        if (LDSOPS) {
            x0 = sh[(i * 64 + threadIdx.x) & 4095];
            x1 = sh[(i * 64 + threadIdx.x + 32) & 4095];
            y0 = sh[(i * 64 + threadIdx.x + 2048) & 4095];
            y1 = sh[(i * 64 + threadIdx.x + 2080) & 4095];
        }
#define VALU_FILL() \
        _Pragma("unroll") for (int v = 0; v < VALU; v += 2) asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(c0), "+v"(c1) : "v"(i))
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0], 0, 0, 0);
        VALU_FILL();
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[1], 0, 0, 0);
        VALU_FILL();
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[2], 0, 0, 0);
        VALU_FILL();
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[3], 0, 0, 0);
        VALU_FILL();
    }
    float s = (float)(c0 + c1);
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LDSOPS, int VALU = 0>
static void run(const char *name, int blocks, int iters, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rate_kernel<LDSOPS, VALU>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((rate_kernel<LDSOPS, VALU>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / reps;
    const double flops = (double)blocks * 4 * iters * 4 * 4096.0;
    printf("%-10s blocks %5d iters %5d  %8.2f us per launch  %7.1f TFLOP/s\n", name, blocks, iters, us, flops / us * 1e-6);
}

int main()
{
    float *out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    // 288 groups of four = the k extent of the widest 3x3 convolution (9 taps x 64 channels / 2)
    for (int blocks : {256, 512, 1024})
        for (int iters : {288, 576, 2880}) {
            run<0>("mfma only", blocks, iters, out);
            run<1>("with lds", blocks, iters, out);
        }
    run<0, 4>("valu 4", 256, 288, out);
    run<0, 8>("valu 8", 256, 288, out);
    run<0, 12>("valu 12", 256, 288, out);
    run<0, 16>("valu 16", 256, 288, out);
    run<0, 24>("valu 24", 256, 288, out);
    run<1, 8>("lds+valu 8", 256, 288, out);
    // two and four waves per SIMD: another wave's matrix instructions run while this one issues arithmetic
    run<0, 8>("valu 8", 512, 288, out);
    run<0, 16>("valu 16", 512, 288, out);
    run<1, 8>("lds+valu 8", 512, 288, out);
    run<1, 16>("lds+valu16", 512, 288, out);
    run<0, 8>("valu 8", 1024, 288, out);
    run<1, 16>("lds+valu16", 1024, 288, out);
    hipFree(out);
    return 0;
}
