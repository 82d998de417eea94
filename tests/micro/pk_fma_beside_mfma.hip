// Micro-benchmark for one question: does a chain of v_pk_fma_f32 give the same sums as the same
// chain of v_fma_f32 when the wave that runs it shares its SIMD with a wave issuing MFMAs back to back?
//
// One 512-thread workgroup per CU.  Waves 4-7 stream rows of a float matrix from global memory (16
// rows of 1 KB in flight, four consecutive floats per lane) and accumulate row * x[row] three times
// from the SAME registers: packed (v_pk_fma_f32), single-lane (v_fma_f32), packed again.  Waves 0-3
// either idle (mode 0) or issue v_mfma_f32_32x32x16_f16 continuously (mode 1) until the others are
// done.  Any difference between the three sums of a lane is counted per lane.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize pk_fma_beside_mfma.hip -o pk_fma_beside_mfma
//   ./pk_fma_beside_mfma          prints one JSON line per mode
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int U = 16;

__global__ __launch_bounds__(512, 1) void kernel(const f32x4 *w, const float *x, int rows, int iters, int mfma_on, int hi_x,
                                                 unsigned *bad_by_lane, float *sink)
{
    __shared__ unsigned done;
    __shared__ float xs[4096];
    __shared__ float pairs[4096];                        // (row index as an integer, x) pairs
    for (int i = threadIdx.x; i < 2048; i += 512) { pairs[2 * i] = __int_as_float(i % 900); pairs[2 * i + 1] = x[i % rows]; }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) done = 0;
    for (int i = tid; i < rows && i < 4096; i += 512) xs[i] = x[i];
    __syncthreads();
    if (wave < 4) {
        if (!mfma_on) return;
        half8 a, b;
        for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
        f32x16 acc[4];
        for (int q = 0; q < 4; q++) for (int j = 0; j < 16; j++) acc[q][j] = 0.f;
        __builtin_amdgcn_s_setprio(2);
        for (int guard = 0; guard < (1 << 22); guard++) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
            if (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 4) break;
        }
        float s = 0.f;
        for (int q = 0; q < 4; q++) for (int j = 0; j < 16; j++) s += acc[q][j];
        if (s == 12345.678f) sink[blockIdx.x] = s;
        return;
    }
    const int w4 = wave - 4;
    unsigned bad = 0, bad_hi = 0;
    const f32x4 *wrow = w + lane;
    for (int it = 0; it < iters; it++) {
        f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const int r0 = ((it * 4 + w4) * U + blockIdx.x * 7) % (rows - U);
        f32x4 wv[U];
        float xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { wv[u] = wrow[(size_t)(r0 + u) * 64]; xv[u] = xs[(r0 + u) & 4095]; }
        if (hi_x) {
            // the form the sparse policy layer's first row takes: x is the HIGH dword of a (row index, x)
            // pair read from LDS with one 8-byte load, selected for both results with op_sel:[0,1,0]
            const f32x2 kx = *reinterpret_cast<const f32x2 *>(&pairs[2 * ((r0 & 2047))]);
            f32x2 h01 = {0.125f, -0.25f}, h23 = {0.5f, 0.75f};
            float g[4] = {0.125f, -0.25f, 0.5f, 0.75f};
            const f32x2 w01 = {wv[0][0], wv[0][1]}, w23 = {wv[0][2], wv[0][3]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(h01) : "v"(w01), "v"(kx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(h23) : "v"(w23), "v"(kx));
#pragma unroll
            for (int j = 0; j < 4; j++) { g[j] = __builtin_fmaf(wv[0][j], kx[1], g[j]); asm volatile("" : "+v"(g[j])); }
            if (!(h01[0] == g[0] && h01[1] == g[1] && h23[0] == g[2] && h23[1] == g[3])) bad_hi++;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const f32x2 x2 = {xv[u], xv[u]}, w01 = {wv[u][0], wv[u][1]}, w23 = {wv[u][2], wv[u][3]};
            p01 = __builtin_elementwise_fma(w01, x2, p01);
            p23 = __builtin_elementwise_fma(w23, x2, p23);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) { s[j] = __builtin_fmaf(wv[u][j], xv[u], s[j]); asm volatile("" : "+v"(s[j])); }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const f32x2 x2 = {xv[u], xv[u]}, w01 = {wv[u][0], wv[u][1]}, w23 = {wv[u][2], wv[u][3]};
            q01 = __builtin_elementwise_fma(w01, x2, q01);
            q23 = __builtin_elementwise_fma(w23, x2, q23);
            asm volatile("" : "+v"(q01), "+v"(q23));
        }
        const bool same = p01[0] == s[0] && p01[1] == s[1] && p23[0] == s[2] && p23[1] == s[3] &&
                          q01[0] == s[0] && q01[1] == s[1] && q23[0] == s[2] && q23[1] == s[3];
        bad += same ? 0u : 1u;
    }
    if (bad) atomicAdd(&bad_by_lane[lane], bad);
    if (bad_hi) atomicAdd(&bad_by_lane[64 + lane], bad_hi);
    if (lane == 0) atomicAdd(&done, 1u);
}

int main()
{
    const int rows = 900, iters = 4000, grid = 256;
    std::vector<float> hw((size_t)rows * 256), hx(rows);
    srand(1);
    for (auto &v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto &v : hx) v = (rand() / (float)RAND_MAX) * 0.1f;
    f32x4 *w; float *x, *sink; unsigned *bad;
    hipMalloc(&w, hw.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&bad, 128 * 4); hipMalloc(&sink, grid * 4);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(bad, 0, 128 * 4);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), 0, 0, w, x, rows, iters, mode, 1, bad, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("{\"error\": \"launch failed\"}\n"); return 1; }
        unsigned hb[128];
        hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
        unsigned long long tot = 0, q[4] = {0, 0, 0, 0}, toth = 0, qh[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; l++) { tot += hb[l]; q[l / 16] += hb[l]; toth += hb[64 + l]; qh[l / 16] += hb[64 + l]; }
        printf("{\"mfma_waves_beside\": %d, \"chains_checked\": %lld, \"lanes_with_different_sums\": %llu, \"by_lane_quarter\": [%llu, %llu, %llu, %llu], "
               "\"high_dword_x_form_different\": %llu, \"high_dword_x_by_lane_quarter\": [%llu, %llu, %llu, %llu]}\n",
               mode, (long long)grid * 4 * iters * 64, tot, q[0], q[1], q[2], q[3], toth, qh[0], qh[1], qh[2], qh[3]);
    }
    return 0;
}
