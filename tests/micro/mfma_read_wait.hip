// How many wait states does gfx950 need between v_mfma_f32_32x32x16_f16 and a VALU read of its result?
// The hardware does not interlock this pair; the compiler pads it with s_nop from its own table.
// For N = 0 .. 19 one wave runs:  acc = 0;  acc = mfma(a, b, acc);  s_nop N-1 (none for N = 0);  read
// the LAST register of the accumulator tuple (VGPR destination: v_mov_b32, AGPR destination:
// v_accvgpr_read_b32) and compares it with the value read after a full wait.  Prints, per destination
// kind, the smallest N from which on every lane reads the finished value.
//   hipcc --offload-arch=gfx950 -O2 mfma_read_wait.hip -o mfma_read_wait && ./mfma_read_wait
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int N, bool AGPR>
__device__ inline void probe(const half8 &a, const half8 &b, float &early, float &late)
{
    if constexpr (AGPR) {
        asm volatile(
            "v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\t"
            "v_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\t"
            "v_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\t"
            "v_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            "v_mfma_f32_32x32x16_f16 a[0:15], %2, %3, a[0:15]\n\t"
            ".if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\t"
            "v_accvgpr_read_b32 %0, a15\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_accvgpr_read_b32 %1, a15\n\t"
            : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "n"(N)
            : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    } else {
        asm volatile(
            "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\t"
            "v_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\t"
            "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"
            "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            "v_mfma_f32_32x32x16_f16 v[32:47], %2, %3, v[32:47]\n\t"
            ".if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\t"
            "v_mov_b32 %0, v47\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %1, v47\n\t"
            : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "n"(N)
            : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    }
}

template <int N, bool AGPR>
__device__ inline void run(const half8 &a, const half8 &b, unsigned *stale)
{
    float e, l;
    probe<N, AGPR>(a, b, e, l);
    if (e != l) atomicAdd(&stale[(AGPR ? 32 : 0) + N], 1u);
}

__global__ __launch_bounds__(64) void kernel(unsigned *stale)
{
    const int lane = threadIdx.x;
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.25f + 0.01f * (lane % 7 + j)); b[j] = (_Float16)(0.5f - 0.02f * (lane % 5 + j)); }
#define PROBE(N) run<N, false>(a, b, stale); run<N, true>(a, b, stale);
    PROBE(0) PROBE(1) PROBE(2) PROBE(3) PROBE(4) PROBE(5) PROBE(6) PROBE(7) PROBE(8) PROBE(9)
    PROBE(10) PROBE(11) PROBE(12) PROBE(13) PROBE(14) PROBE(15) PROBE(16) PROBE(17) PROBE(18) PROBE(19)
}

int main()
{
    unsigned *d, h[64] = {0};
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(kernel, dim3(64), dim3(64), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("{\"error\": \"launch failed\"}\n"); return 1; }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int k = 0; k < 2; k++) {
        int need = 0;
        for (int n = 0; n < 20; n++) if (h[32 * k + n]) need = n + 1;
        printf("{\"destination\": \"%s\", \"lanes_that_read_an_unfinished_value_by_wait_states\": [", k ? "AGPR (v_accvgpr_read_b32)" : "VGPR (v_mov_b32)");
        for (int n = 0; n < 20; n++) printf("%u%s", h[32 * k + n], n < 19 ? ", " : "");
        printf("], \"wait_states_needed\": %d}\n", need);
    }
    return 0;
}
