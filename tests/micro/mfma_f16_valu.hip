// Does ordinary vector arithmetic hide behind v_mfma_f32_32x32x16_f16 (the search CNN's matrix instruction)
// the way it does NOT hide behind v_mfma_f32_32x32x2f32 (mfma_f32_rate.hip)?  Two kinds of waves on every SIMD:
//   "dense" waves issue groups of four independent f16 matrix instructions back to back;
//   "other" waves (OTHER > 0 of them per workgroup of 4 + OTHER waves) issue only integer additions, a fixed number.
// Reported: the launch time with the dense waves alone, with the arithmetic waves alone, and with both.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_valu mfma_f16_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// mode bit 0: dense waves work; bit 1: arithmetic waves work; SAME: the arithmetic is issued by the dense waves themselves
template <int VALU_PER_MFMA>
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode, int valu_iters)
{
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f16v acc[4];
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        h8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x + j); b[j] = (_Float16)(j + 1); }
        int c0 = threadIdx.x, c1 = blockIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VALU_PER_MFMA; v += 2)
                    asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(c0), "+v"(c1) : "v"(i));
            }
        }
        float s = (float)(c0 + c1);
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        int c0 = threadIdx.x, c1 = blockIdx.x, c2 = 3, c3 = 4;
        for (int i = 0; i < valu_iters; ++i)
            asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
                         "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(i));
        out[blockIdx.x * 512 + threadIdx.x] = (float)(c0 + c1 + c2 + c3);
    }
}

template <int V>
static double run(float *out, int threads, int iters, int mode, int valu_iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, out, iters, mode, valu_iters);
    const int reps = 20;
    (void)hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, out, iters, mode, valu_iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / reps;
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 1000;                 // 4000 matrix instructions per dense wave
    const double flops = 256.0 * 4 * iters * 4 * 32768.0;
    double t = run<0>(out, 256, iters, 1, 0);
    printf("dense waves alone (one per SIMD)              %8.2f us  %7.1f TFLOP/s\n", t, flops / t * 1e-6);
    t = run<4>(out, 256, iters, 1, 0);
    printf("  + 4 additions per matrix instruction, same wave %8.2f us\n", t);
    t = run<8>(out, 256, iters, 1, 0);
    printf("  + 8 additions per matrix instruction, same wave %8.2f us\n", t);
    t = run<16>(out, 256, iters, 1, 0);
    printf("  + 16 additions per matrix instruction, same wave %8.2f us\n", t);
    for (int valu_iters : {2000, 4000, 8000}) {  // x 8 additions: 16 k, 32 k, 64 k per arithmetic wave
        const double ta = run<0>(out, 512, iters, 2, valu_iters);
        const double tb = run<0>(out, 512, iters, 3, valu_iters);
        printf("arithmetic waves alone (%d additions each, one per SIMD) %8.2f us;  beside the dense waves %8.2f us\n",
               valu_iters * 8, ta, tb);
    }
    (void)hipFree(out);
    return 0;
}
