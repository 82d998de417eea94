// The training step of the policy/value net on gfx950, forward and backward, in float32.
//
// Reference semantics restated here (files relative to the reference tree):
//   Net.forward                      policy_value_net_pytorch.py:43-57
//   PolicyValueNet.train_step        policy_value_net_pytorch.py:117-148
//     loss = mse_loss(value, z) - mean(sum(pi * log p, 1)), Adam with weight decay 1e-4,
//     entropy = -mean(sum(p * log p, 1)) of the step's own forward pass
//   PolicyValueNet.policy_value      policy_value_net_pytorch.py:80-94 (the KL monitor of train.py:161-214)
//
// A mini-batch makes every layer a dense contraction over M = batch * cells rows, so all three
// convolution passes are GEMMs on v_mfma_f32_32x32x2f32 (float32 operands, float32 accumulation):
//   forward        Y[m][co]  = sum_{t,ci} X[m + off_t][ci] W[co][ci][t]          conv_gemm_kernel
//   backward data  dX[m][ci] = sum_{t,co} dY[m + off_t][co] W[co][ci][8 - t]     conv_gemm_kernel
//   backward weights dW[(t,ci)][co] = sum_m X[m + off_t][ci] dY[m][co]           wgrad_gemm_kernel
// Activations are channels-last ([row m = sample * S + cell][channel]), so a row of 4 consecutive k
// (one tap, four channels) is one 16-byte load and a 3x3 tap is a row offset; a tap that leaves the
// board reads zero.  Nothing is unfolded in memory.  The small layers (1x1 heads as a GEMM with one
// tap, fully connected layers, losses) and Adam are plain kernels; every sum over the batch has a
// fixed order, so a step gives the same bits every time.
#pragma once
#include <type_traits>
#include "azg_cnn.hpp"

namespace azg {

// Timing experiments on conv_gemm_kernel (results are WRONG with any bit set): 1 no global loads in the
// loop, 2 no staging stores, 4 no barrier, 8 no operand reads from LDS, 16 no output stores, 32 no
// matrix instructions.
#ifndef AZG_TRAIN_EXP
#define AZG_TRAIN_EXP 0
#endif
constexpr int TBM = 128;       // rows of a block tile (4 waves x 32)
constexpr int TKC = 16;        // k per LDS stage

struct ConvGemmArgs {
    const float *src;    // [M][Cs] channels-last source (activations forward, gradients backward)
    const float *w;      // [N][K], K = taps * Cs, k = tap * Cs + channel
    const float *bias;   // [N] (forward) or null
    const float *act;    // [M][ldo]: forward activation of the layer whose gradient is produced (ReLU mask), or null
    float *out;          // [M][ldo]; heads: policy part [B][4S]
    float *out2;         // heads: value part [B][2S]
    int M, N, Cs, K, taps, S, Nb, ldo;
};

// On this chip a float32 matrix instruction and the ordinary vector instructions of ANY wave of the same
// SIMD do not overlap: a wave that issues v_mfma_f32_32x32x2f32 back to back reaches 137 TFLOP/s in a
// 35 us launch, and every four vector additions per matrix instruction add 23 cycles to its 64, with one,
// two or four waves per SIMD alike (tests/micro/mfma_f32_rate.hip).  So the loops below are written for the
// fewest vector instructions per stage, not for interleaving:
//   * global loads are buffer loads: an element that is off the board, past K or past M gets an offset
//     beyond the buffer's size and the hardware returns zeros (no select, no branch, 32-bit offsets);
//   * which of the nine taps of a row's cell are on the board is a 9-bit mask computed once; a stage's
//     tap is uniform over the workgroup when the channel count is a multiple of the stage's 16 k, so its
//     row shift and bit are scalar arithmetic;
//   * tiles lie in LDS row-major with k innermost ([row][16 k + 4 pad]): a 16-byte global load is one
//     16-byte LDS store, and a lane's eight k of a stage are two 16-byte LDS loads.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBufferOob = 0x7fffffff;                       // an offset no buffer here reaches
__device__ __forceinline__ __amdgpu_buffer_rsrc_t train_buffer(const float *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 train_buffer_load(__amdgpu_buffer_rsrc_t r, int byte_offset)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 0));
}
constexpr int TLK = TKC + 4;   // words per tile row: 16-lane groups of the 16-byte operand loads hit every bank once

// MODE 0: out = relu(acc + bias);  1: out = act > 0 ? acc : 0;  2: the two 1x1 head convolutions:
// relu(acc + bias) of column c < 4 goes to out[b][c*S + p], of column 4, 5 to out2[b][(c-4)*S + p] (row stride ldo)
// (the order view(-1, 4*S) / view(-1, 2*S) flattens in, policy_value_net_pytorch.py:50,54);
// 3: out = acc + bias (no activation).
// CS4: the source has 4 channels (the input planes): a stage's 16 k are four taps, one per thread of a row's
// four; otherwise the channel count must be a multiple of 16.  The source holds M * Cs floats, w holds N * K.
template <int BN, int MODE, bool CS4 = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmArgs a)
{
    __shared__ __attribute__((aligned(16))) float As[2][TBM][TLK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][TLK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, l32 = lane & 31;
    const int m0 = blockIdx.x * TBM, n0 = blockIdx.y * BN;
    const int k4 = tid & 3;                                  // this thread's float4 inside a stage's 16 k
    const __amdgpu_buffer_rsrc_t rsrc = train_buffer(a.src, (unsigned)a.M * a.Cs * 4u);
    const __amdgpu_buffer_rsrc_t rw = train_buffer(a.w, (unsigned)a.N * a.K * 4u);
    // A rows of this thread: r = (tid >> 2) + 64 i; byte offset of the row's channel 4 k4 and the taps on the board
    int aoff[2];
    unsigned amask[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int m = m0 + (tid >> 2) + 64 * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / a.S, p = mm - b * a.S;
        const int pr = p / a.Nb, pc = p - pr * a.Nb;
        unsigned msk = 1u;
        if (a.taps == 9) {
            msk = 0u;
#pragma unroll
            for (int t = 0; t < 9; t++)
                msk |= (((unsigned)(pr + t / 3 - 1) < (unsigned)a.Nb) & ((unsigned)(pc + t % 3 - 1) < (unsigned)a.Nb)) ? 1u << t : 0u;
        }
        amask[i] = ok ? msk : 0u;
        aoff[i] = (mm * a.Cs + (CS4 ? 0 : 4 * k4)) * 4;
    }
    constexpr int NBL = (BN * 4 + 255) / 256;                // float4 of B per thread and stage
    int boff[NBL];
#pragma unroll
    for (int i = 0; i < NBL; i++) {
        const int n = (tid + 256 * i) >> 2;
        boff[i] = ((n < BN) & (n0 + n < a.N)) ? ((n0 + n) * a.K + 4 * k4) * 4 : kBufferOob;
    }
    const int nchunks = (a.K + TKC - 1) / TKC;

    // The stages are loaded in order: the stage's first k as (tap, channel) advances by 16 per call.  A k
    // past K is a tap past the last one: no bit of the mask, zeros.  (B rows then run into the next row
    // of w, or past its end into zeros: multiplied by A's zeros either way.)
    int s_tap = 0, s_c = 0, s_kb = 0;
    auto load = [&](f32x4 (&va)[2], f32x4 (&vb)[NBL]) {
        if (CS4) {
            const int tap = s_tap + k4;
            const int t3 = (tap * 11) >> 5;                  // tap / 3 for tap < 16
            const int shift = ((t3 - 1) * a.Nb + tap - 3 * t3 - 1) * 16;
#pragma unroll
            for (int i = 0; i < 2; i++) va[i] = train_buffer_load(rsrc, ((amask[i] >> tap) & 1u) ? aoff[i] + shift : kBufferOob);
            s_tap += 4;
        } else {
            const int t3 = (s_tap * 11) >> 5;
            const int cell = a.taps == 9 ? (t3 - 1) * a.Nb + s_tap - 3 * t3 - 1 : 0;
            const int shift = (cell * a.Cs + s_c) * 4;
            const unsigned bit = 1u << s_tap;
#pragma unroll
            for (int i = 0; i < 2; i++) va[i] = train_buffer_load(rsrc, (amask[i] & bit) ? aoff[i] + shift : kBufferOob);
            s_c += TKC;
            if (s_c >= a.Cs) { s_c = 0; s_tap++; }
        }
#pragma unroll
        for (int i = 0; i < NBL; i++) vb[i] = train_buffer_load(rw, boff[i] + s_kb);
        s_kb += TKC * 4;
    };
    auto stage = [&](int st, const f32x4 (&va)[2], const f32x4 (&vb)[NBL]) {
#pragma unroll
        for (int i = 0; i < 2; i++) *reinterpret_cast<f32x4 *>(&As[st][(tid >> 2) + 64 * i][4 * k4]) = va[i];
#pragma unroll
        for (int i = 0; i < NBL; i++) {
            const int n = (tid + 256 * i) >> 2;
            if (BN * 4 % 256 == 0 || n < BN) *reinterpret_cast<f32x4 *>(&Bs[st][n][4 * k4]) = vb[i];
        }
    };

    f32x16 acc[BN / 32];
#pragma unroll
    for (int t = 0; t < BN / 32; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // Global loads run TWO stages ahead of the multiplication (two register sets), the LDS is double
    // buffered: a stage's loads have two multiplication phases (about 1.7 us) to arrive.
    f32x4 va[2][2], vb[2][NBL];
    load(va[0], vb[0]);
    load(va[1], vb[1]);                                      // (past K: zeros)
    stage(0, va[0], vb[0]);
    __syncthreads();
    // (the stage index is a compile-time constant: register sets must not be indexed at run time)
    auto body = [&](auto stc) {
        constexpr int st = decltype(stc)::value;
        // lane half h multiplies k = 8 h .. 8 h + 7 of the stage, one per matrix instruction; every operand
        // is requested before the first product: one LDS round trip per stage
        f32x4 av[2], bv[BN / 32][2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (AZG_TRAIN_EXP & 8) {
                av[q] = f32x4{(float)lane, 1.f, 2.f, 3.f};
#pragma unroll
                for (int t = 0; t < BN / 32; t++) bv[t][q] = f32x4{(float)(lane + t), 1.f, 2.f, 3.f};
                continue;
            }
            av[q] = *reinterpret_cast<const f32x4 *>(&As[st][32 * wave + l32][8 * h + 4 * q]);
#pragma unroll
            for (int t = 0; t < BN / 32; t++) bv[t][q] = *reinterpret_cast<const f32x4 *>(&Bs[st][32 * t + l32][8 * h + 4 * q]);
        }
        if (!(AZG_TRAIN_EXP & 1)) load(va[st], vb[st]);      // set `st` was staged in the previous iteration (or above)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int t = 0; t < BN / 32; t++) {
                if (AZG_TRAIN_EXP & 32) acc[t][j] += av[j >> 2][j & 3] * bv[t][j >> 2][j & 3];
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j >> 2][j & 3], bv[t][j >> 2][j & 3], acc[t], 0, 0, 0);
            }
        if (!(AZG_TRAIN_EXP & 2)) stage(st ^ 1, va[st ^ 1], vb[st ^ 1]);  // (after the last stage: zeros nobody reads)
        if (!(AZG_TRAIN_EXP & 4)) __syncthreads();
    };
    // (always in pairs; an odd count ends with a stage of zeros: a conditional second half would be a
    //  path on which the first half's loads are still in flight at the loop's head, and the wait
    //  counts the compiler puts there would then drain every iteration's loads)
    for (int kc = 0; kc < nchunks; kc += 2) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
    }

#pragma unroll
    for (int t = 0; t < BN / 32; t++) {
        const int n = n0 + 32 * t + l32;
        if (n >= a.N || (MODE == 2 && n >= 6)) continue;
        const float bv = (MODE == 1 || !a.bias) ? 0.f : a.bias[n];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + 32 * wave + acc_row(r, h);
            if (m >= a.M) continue;
            float v = acc[t][r] + bv;
            if ((AZG_TRAIN_EXP & 16) && v != 12345.678f) continue;
            if (MODE == 0) {
                a.out[(size_t)m * a.ldo + n] = v > 0.f ? v : 0.f;
            } else if (MODE == 1) {
                a.out[(size_t)m * a.ldo + n] = a.act[(size_t)m * a.ldo + n] > 0.f ? v : 0.f;
            } else if (MODE == 3) {
                a.out[(size_t)m * a.ldo + n] = v;
            } else {
                v = v > 0.f ? v : 0.f;
                const int b = m / a.S, p = m - b * a.S;
                if (n < 4) a.out[(size_t)b * 4 * a.S + n * a.S + p] = v;
                else if (n < 6) a.out2[(size_t)b * a.ldo + (n - 4) * a.S + p] = v;
            }
        }
    }
}

struct WgradArgs {
    const float *src;    // [M][Cs] channels-last: the layer's input
    const float *dz;     // [M][ldz]: gradient at the layer's pre-activation
    float *part;         // [splits][I][J] partial sums, I = taps * Cs
    float *bpart;        // [splits][J] column sums of dz over the split's rows (bias gradients), or null
    int M, Cs, taps, I, J, ldz, S, Nb, rows_per_split;
};

// part[z][i = (t, ci)][j] = sum over the rows m of split z of  src[m + off_t][ci] * dz[m][j];
// bpart[z][j] = sum over the same rows of dz[m][j] (by the workgroups of the first row of tiles)
template <int BN>
__global__ __launch_bounds__(256, 2) void wgrad_gemm_kernel(WgradArgs a)
{
    // (k = the rows m of the split; both tiles arrive k-major from memory: [k][i] and [k][j])
    __shared__ __attribute__((aligned(16))) float As[2][TKC][TBM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][TKC][BN + 4];
    __shared__ unsigned short taps_on_board[256];            // per cell: bit t = tap t of that cell is on the board
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, l32 = lane & 31;
    const int i0 = blockIdx.x * TBM, j0 = blockIdx.y * BN, z = blockIdx.z;
    const int mbeg = z * a.rows_per_split, mend = min(a.M, mbeg + a.rows_per_split);
    const int nchunks = mend > mbeg ? (mend - mbeg + TKC - 1) / TKC : 0;
    if (tid < a.S) {
        unsigned msk = 1u;
        if (a.taps == 9) {
            const int pr = tid / a.Nb, pc = tid - pr * a.Nb;
            msk = 0u;
#pragma unroll
            for (int t = 0; t < 9; t++)
                msk |= (((unsigned)(pr + t / 3 - 1) < (unsigned)a.Nb) & ((unsigned)(pc + t % 3 - 1) < (unsigned)a.Nb)) ? 1u << t : 0u;
        }
        taps_on_board[tid] = (unsigned short)msk;
    }
    // A: 16 rows (m) x 32 float4 (i): 2 per thread;  B: 16 rows x BN/4 float4.  Rows from the split's end on
    // lie past the end of B's buffer (zeros from the hardware), so A needs no row test of its own.
    const __amdgpu_buffer_rsrc_t rsrc = train_buffer(a.src, (unsigned)a.M * a.Cs * 4u);
    const __amdgpu_buffer_rsrc_t rdz = train_buffer(a.dz, (unsigned)mend * a.ldz * 4u);
    const int i4 = tid & 31;
    const int ia = i0 + 4 * i4;
    const bool iok = ia < a.I;
    const int tap = iok ? ia / a.Cs : 0, ci = ia - tap * a.Cs;
    const int cell = a.taps == 9 ? (tap / 3 - 1) * a.Nb + tap % 3 - 1 : 0;
    const unsigned tapbit = iok ? 1u << tap : 0u;
    constexpr int BQ = BN / 4;                               // float4 per B row
    constexpr int NBL = (TKC * BQ + 255) / 256;
    int aoff[2], lp[2], boff[NBL];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int m = mbeg + (tid >> 5) + 8 * i;
        aoff[i] = ((m + cell) * a.Cs + ci) * 4;
        lp[i] = m % a.S;
    }
#pragma unroll
    for (int i = 0; i < NBL; i++) {
        const int q = tid + 256 * i, row = q / BQ, j4 = q - row * BQ, j = j0 + 4 * j4;
        boff[i] = ((row < TKC) & (j < a.ldz)) ? ((mbeg + row) * a.ldz + j) * 4 : kBufferOob;
    }
    __syncthreads();
    // The stages are loaded in order: this thread's rows advance by 16 per call, their cells p = m mod S with them.
    const int sstep = TKC % a.S, astep = TKC * a.Cs * 4, bstep = TKC * a.ldz * 4;
    int s_a = 0, s_b = 0;
    auto load = [&](f32x4 (&va)[2], f32x4 (&vb)[NBL]) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            va[i] = train_buffer_load(rsrc, (taps_on_board[lp[i]] & tapbit) ? aoff[i] + s_a : kBufferOob);
            lp[i] += sstep;
            lp[i] -= lp[i] >= a.S ? a.S : 0;
        }
#pragma unroll
        for (int i = 0; i < NBL; i++) vb[i] = train_buffer_load(rdz, boff[i] + s_b);
        s_a += astep; s_b += bstep;
    };
    auto stage = [&](int st, const f32x4 (&va)[2], const f32x4 (&vb)[NBL]) {
#pragma unroll
        for (int i = 0; i < 2; i++) *reinterpret_cast<f32x4 *>(&As[st][(tid >> 5) + 8 * i][4 * i4]) = va[i];
#pragma unroll
        for (int i = 0; i < NBL; i++) {
            const int q = tid + 256 * i, row = q / BQ, j4 = q - row * BQ;
            if (TKC * BQ % 256 == 0 || row < TKC) *reinterpret_cast<f32x4 *>(&Bs[st][row][4 * j4]) = vb[i];
        }
    };

    f32x16 acc[BN / 32];
#pragma unroll
    for (int t = 0; t < BN / 32; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    f32x4 va[2][2], vb[2][NBL];
    load(va[0], vb[0]);
    load(va[1], vb[1]);
    stage(0, va[0], vb[0]);
    __syncthreads();
    const bool sums = a.bpart != nullptr && blockIdx.x == 0 && tid < BN;
    float bsum = 0.f;
    auto body = [&](auto stc) {
        constexpr int st = decltype(stc)::value;
        // every operand of the stage is requested before the first product: one LDS round trip per stage
        float av[TKC / 2], bv[TKC / 2][BN / 32];
#pragma unroll
        for (int kk = 0; kk < TKC; kk += 2) {              // lane half h takes k = kk/2 + 8 h of the pair
            av[kk / 2] = As[st][kk / 2 + 8 * h][32 * wave + l32];
#pragma unroll
            for (int t = 0; t < BN / 32; t++) bv[kk / 2][t] = Bs[st][kk / 2 + 8 * h][32 * t + l32];
        }
        load(va[st], vb[st]);
#pragma unroll
        for (int kk = 0; kk < TKC; kk += 2)
#pragma unroll
            for (int t = 0; t < BN / 32; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk / 2], bv[kk / 2][t], acc[t], 0, 0, 0);
        if (sums) {
#pragma unroll
            for (int k = 0; k < TKC; k++) bsum += Bs[st][k][tid];        // rows past the split's end were staged as zeros
        }
        stage(st ^ 1, va[st ^ 1], vb[st ^ 1]);
        __syncthreads();
    };
    // (always in pairs, as in conv_gemm_kernel; an odd count ends with a stage of zeros)
    for (int kc = 0; kc < nchunks; kc += 2) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
    }
    if (sums && j0 + tid < a.J) a.bpart[(size_t)z * a.J + j0 + tid] = bsum;
#pragma unroll
    for (int t = 0; t < BN / 32; t++) {
        const int j = j0 + 32 * t + l32;
        if (j >= a.J) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = i0 + 32 * wave + acc_row(r, h);
            if (i < a.I) a.part[((size_t)z * a.I + i) * a.J + j] = acc[t][r];
        }
    }
}

// part[z][c] = sum of x[m][c] over the rows of split z; C <= 256 columns, row stride ld
__global__ __launch_bounds__(256) void colsum_kernel(const float *x, int M, int C, int ld, int rows_per_split, float *part)
{
    __shared__ float red[256];
    const int tid = threadIdx.x, z = blockIdx.x;
    int Cp = 1;
    while (Cp < C) Cp <<= 1;
    const int lanes = 256 / Cp;                              // row lanes per column
    const int c = tid % Cp, rl = tid / Cp;
    float s = 0.f;
    if (c < C) {
        const int mend = min(M, (z + 1) * rows_per_split);
        for (int m = z * rows_per_split + rl; m < mend; m += lanes) s += x[(size_t)m * ld + c];
    }
    red[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int l = 0; l < lanes; l++) t += red[l * Cp + tid];
        part[(size_t)z * C + tid] = t;
    }
}

// Board.current_state planes [n][4][S] (what PolicyValueNet.policy_value / train_step receive) ->
// channels-last [n*S][4]
__global__ void planes_to_rows_kernel(const float *planes, int n, int S, float *rows)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n * S) return;
    const int b = m / S, p = m - b * S;
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; c++) v[c] = planes[((size_t)b * 4 + c) * S + p];
    *reinterpret_cast<f32x4 *>(rows + (size_t)m * 4) = v;
}

// The 16 parameter tensors in the state_dict order of policy_value_net_pytorch.py:31-41.
struct TrainParams {
    float *p[16];
};

// flat offsets of the 16 tensors for S cells
struct ParamLayout {
    int off[17];
};
__host__ __device__ inline ParamLayout param_layout(int S)
{
    const int sz[16] = {32 * 4 * 9, 32, 64 * 32 * 9, 64, 128 * 64 * 9, 128, 4 * 128, 4, S * 4 * S, S, 2 * 128, 2, 64 * 2 * S, 64, 64, 1};
    ParamLayout l;
    l.off[0] = 0;
    for (int i = 0; i < 16; i++) l.off[i + 1] = l.off[i] + sz[i];
    return l;
}

// GEMM images of the convolution kernels, rebuilt from the parameters before every pass:
//   wf_l[co][t*Cin + ci]  = W_l[co][ci][t]          forward
//   wd_l[ci][t*Cout + co] = W_l[co][ci][8 - t]      backward data (conv2, conv3)
//   wh[c][ch] (32 rows, 6 used) = act_conv1 / val_conv1 kernels; bh[8] their biases
struct TrainImages {
    float *wf[3], *wd[3], *wh, *bh;
};
__global__ void train_pack_kernel(TrainParams P, TrainImages im)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int cin[3] = {4, 32, 64}, cout[3] = {32, 64, 128};
    int base = 0;
#pragma unroll
    for (int l = 0; l < 3; l++) {
        const int n = cout[l] * cin[l] * 9;
        if (idx >= base && idx < base + n) {
            const int e = idx - base;                        // PyTorch order: (co, ci, t)
            const int co = e / (cin[l] * 9), ci = (e / 9) % cin[l], t = e % 9;
            const float v = P.p[2 * l][e];
            im.wf[l][(size_t)co * cin[l] * 9 + t * cin[l] + ci] = v;
            if (l > 0) im.wd[l][(size_t)ci * cout[l] * 9 + (8 - t) * cout[l] + co] = v;
        }
        base += n;
    }
    if (idx >= base && idx < base + 32 * 128) {
        const int e = idx - base, c = e / 128, ch = e % 128;
        im.wh[e] = c < 4 ? P.p[6][c * 128 + ch] : (c < 6 ? P.p[10][(c - 4) * 128 + ch] : 0.f);
    }
    base += 32 * 128;
    if (idx >= base && idx < base + 8) {
        const int c = idx - base;
        im.bh[c] = c < 4 ? P.p[7][c] : (c < 6 ? P.p[11][c - 4] : 0.f);
    }
}
constexpr int kTrainPackThreads = 1152 + 18432 + 73728 + 32 * 128 + 8;

// ---------------------------------------------------------------- fully connected layers, losses
struct FcArgs {
    const float *hp, *hv;            // head outputs [n][4S], [n][2S] (after ReLU)
    const float *wp, *bp;            // act_fc1 [S][4S], [S]
    const float *wv1, *bv1;          // val_fc1 [64][2S], [64]
    const float *wv2, *bv2;          // val_fc2 [64], [1]
    const float *pi, *z;             // targets [n][S], [n]  (null: forward only)
    float *probs, *values;           // forward-only outputs [n][S], [n] (may be null)
    const float *oldp;               // [n][S] or null: earlier probabilities for the KL test of train.py:189-193
    double *klparts;                 // [n]: sum_o oldp * (log(oldp + 1e-10) - log(probs + 1e-10)) of the sample
    float *dlogit, *dh1;             // [n][Sp] (Sp = S rounded up to 4, padding zero), [n][64]
    float *g2;                       // [n][128]: columns 0-63 dpre2 * h1[j] (val_fc2's weight gradient terms), 64 dpre2, rest zero
    float *dh8;                      // [n*S][8]: gradient at the six head pre-activations, channels-last
    float *lossparts;                // [n][3]: value loss, policy loss, entropy of the sample
    int n, S, ldhv;                  // ldhv: row stride of hv (2S rounded up to 4)
};

__device__ inline float block_reduce(float v, float *red, int tid, bool is_max)
{
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = is_max ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// One sample per block: act_fc1 + log_softmax, val_fc1 + val_fc2 + tanh, the sample's losses and the
// gradients down to the head convolutions' pre-activations (policy_value_net_pytorch.py:49-56,128-141).
__global__ __launch_bounds__(256) void fc_sample_kernel(FcArgs a)
{
    __shared__ __attribute__((aligned(16))) float hp[4 * MAXS], hv[2 * MAXS];
    __shared__ float lg[MAXS], dl[MAXS], h1s[64], dh1s[64], red[256];
    const int tid = threadIdx.x, b = blockIdx.x, S = a.S;
    for (int k = tid; k < 4 * S; k += 256) hp[k] = a.hp[(size_t)b * 4 * S + k];
    for (int k = tid; k < 2 * S; k += 256) hv[k] = a.hv[(size_t)b * a.ldhv + k];
    __syncthreads();
    for (int item = tid; item < S + 64; item += 256) {
        if (item < S) {
            const float *w = a.wp + (size_t)item * 4 * S;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int k = 0; k < 4 * S; k += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + k);
                s0 = __builtin_fmaf(wv[0], hp[k], s0); s1 = __builtin_fmaf(wv[1], hp[k + 1], s1);
                s2 = __builtin_fmaf(wv[2], hp[k + 2], s2); s3 = __builtin_fmaf(wv[3], hp[k + 3], s3);
            }
            lg[item] = ((s0 + s1) + (s2 + s3)) + a.bp[item];
        } else {
            const int j = item - S;
            const float *w = a.wv1 + (size_t)j * 2 * S;
            float s0 = 0.f, s1 = 0.f;
            for (int k = 0; k < 2 * S; k += 2) { s0 = __builtin_fmaf(w[k], hv[k], s0); s1 = __builtin_fmaf(w[k + 1], hv[k + 1], s1); }
            const float v = (s0 + s1) + a.bv1[j];
            h1s[j] = v > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    // log_softmax over the S logits
    float mx = -3.0e38f;
    for (int o = tid; o < S; o += 256) mx = fmaxf(mx, lg[o]);
    mx = block_reduce(mx, red, tid, true);
    float se = 0.f;
    for (int o = tid; o < S; o += 256) se += expf(lg[o] - mx);
    se = block_reduce(se, red, tid, false);
    const float lse = logf(se);
    // value
    float pv = tid < 64 ? h1s[tid] * a.wv2[tid] : 0.f;
    pv = block_reduce(pv, red, tid, false);
    const float value = tanhf(pv + a.bv2[0]);
    if (a.probs)
        for (int o = tid; o < S; o += 256) a.probs[(size_t)b * S + o] = expf((lg[o] - mx) - lse);
    if (a.values && tid == 0) a.values[b] = value;
    if (a.oldp) {
        __shared__ double dred[256];
        double s = 0.0;
        for (int o = tid; o < S; o += 256) {
            const double old = a.oldp[(size_t)b * S + o], now = expf((lg[o] - mx) - lse);
            s += old * (log(old + 1e-10) - log(now + 1e-10));
        }
        dred[tid] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) dred[tid] += dred[tid + w];
            __syncthreads();
        }
        if (tid == 0) a.klparts[b] = dred[0];
    }
    if (!a.pi) return;

    // losses of the sample and the gradient at the logits: d(-mean_b sum_o pi log p)/dlogit = (p * sum(pi) - pi) / n
    float spi = 0.f, pl = 0.f, en = 0.f;
    for (int o = tid; o < S; o += 256) {
        const float logp = (lg[o] - mx) - lse, pr = expf(logp), t = a.pi[(size_t)b * S + o];
        spi += t; pl -= t * logp; en -= pr * logp;
    }
    spi = block_reduce(spi, red, tid, false);
    pl = block_reduce(pl, red, tid, false);
    en = block_reduce(en, red, tid, false);
    const float inv_n = 1.0f / (float)a.n;
    const int Sp = (S + 3) & ~3;
    for (int o = tid; o < Sp; o += 256) {
        float g = 0.f;
        if (o < S) {
            const float pr = expf((lg[o] - mx) - lse);
            g = (pr * spi - a.pi[(size_t)b * S + o]) * inv_n;
            dl[o] = g;
        }
        a.dlogit[(size_t)b * Sp + o] = g;
    }
    const float err = value - a.z[b];
    const float dpre2 = 2.0f * err * inv_n * (1.0f - value * value);      // through mse_loss's mean and tanh
    if (tid < 64) {
        const float g = h1s[tid] > 0.f ? dpre2 * a.wv2[tid] : 0.f;
        dh1s[tid] = g;
        a.dh1[(size_t)b * 64 + tid] = g;
    }
    if (tid < 128) a.g2[(size_t)b * 128 + tid] = tid < 64 ? dpre2 * h1s[tid] : (tid == 64 ? dpre2 : 0.f);
    if (tid == 0) {
        a.lossparts[3 * b] = err * err;
        a.lossparts[3 * b + 1] = pl;
        a.lossparts[3 * b + 2] = en;
    }
    __syncthreads();
    // down to the head convolutions: dh[k] = sum_o W[o][k] dlogit[o], masked by the head's ReLU
    for (int k = tid; k < 6 * S; k += 256) {
        float s = 0.f;
        int c, p;
        float hval;
        if (k < 4 * S) {
            for (int o = 0; o < S; o++) s = __builtin_fmaf(a.wp[(size_t)o * 4 * S + k], dl[o], s);
            c = k / S; p = k - c * S; hval = hp[k];
        } else {
            const int kv = k - 4 * S;
            for (int j = 0; j < 64; j++) s = __builtin_fmaf(a.wv1[(size_t)j * 2 * S + kv], dh1s[j], s);
            c = 4 + kv / S; p = kv - (c - 4) * S; hval = hv[kv];
        }
        a.dh8[((size_t)b * S + p) * 8 + c] = hval > 0.f ? s : 0.f;
    }
    for (int p = tid; p < S; p += 256) { a.dh8[((size_t)b * S + p) * 8 + 6] = 0.f; a.dh8[((size_t)b * S + p) * 8 + 7] = 0.f; }
}

// gradient at conv3's pre-activation from the six head channels: dz3[m][ch] = a3 > 0 ? sum_c wh[c][ch] dh8[m][c] : 0
__global__ void head_backward_kernel(const float *dh8, const float *wh /*[32][128]*/, const float *a3, int M, float *dz3)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (m, ch4)
    if (idx >= M * 32) return;
    const int m = idx >> 5, ch = (idx & 31) * 4;
    const f32x4 d0 = *reinterpret_cast<const f32x4 *>(dh8 + (size_t)m * 8);
    const f32x4 d1 = *reinterpret_cast<const f32x4 *>(dh8 + (size_t)m * 8 + 4);
    const float d[6] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1]};
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(wh + c * 128 + ch);
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] = __builtin_fmaf(w[j], d[c], s[j]);
    }
    const f32x4 av = *reinterpret_cast<const f32x4 *>(a3 + (size_t)m * 128 + ch);
#pragma unroll
    for (int j = 0; j < 4; j++) s[j] = av[j] > 0.f ? s[j] : 0.f;
    *reinterpret_cast<f32x4 *>(dz3 + (size_t)m * 128 + ch) = s;
}

// ---------------------------------------------------------------- gradient assembly, loss, Adam
// The flat gradient in the state_dict order.  Every partial-sum array is a segment of the kernel's
// index space, walked in ITS OWN order (consecutive threads read consecutive words of every split);
// an entry is the sum of its splits in ascending order and lands at its place in the flat vector.
struct GradSeg {
    const float *part;   // [splits][size]
    int start, size, splits, kind, a, b;     // kind: how a source index maps to the flat vector (see the kernel)
};
struct GradSegs {
    GradSeg s[16];
    int n, total;
};
enum { GS_CONV_W = 0, GS_PLAIN = 1, GS_HEAD_W = 2, GS_HEAD_B = 3, GS_FC_W = 4, GS_G2 = 5 };

// (64 entries per workgroup, four threads per entry: each sums a quarter of the splits, the quarters are
//  added in ascending order -- four times the loads in flight of one thread per entry, same bits every time)
constexpr int kAssembleEntries = 64;
__global__ __launch_bounds__(256) void grad_assemble_kernel(GradSegs g, int S, float *flat)
{
    __shared__ float quarter[4][kAssembleEntries];
    const int e = threadIdx.x & (kAssembleEntries - 1), zg = threadIdx.x / kAssembleEntries;
    const int idx = blockIdx.x * kAssembleEntries + e;
    int dst = -1;
    float s = 0.f;
    if (idx < g.total) {
        int si = 0;
#pragma unroll
        for (int i = 1; i < 16; i++) si += (i < g.n && idx >= g.s[i].start);
        const GradSeg sg = g.s[si];
        const int r = idx - sg.start;
        const ParamLayout L = param_layout(S);
        if (sg.kind == GS_CONV_W) {                    // [(tap, ci)][co] -> [co][ci][tap];  a = tensor, b = cin
            const int cout = sg.size / (9 * sg.b), i = r / cout, co = r - i * cout, tap = i / sg.b, ci = i - tap * sg.b;
            dst = L.off[sg.a] + (co * sg.b + ci) * 9 + tap;
        } else if (sg.kind == GS_PLAIN) {              // a = tensor
            dst = L.off[sg.a] + r;
        } else if (sg.kind == GS_HEAD_W) {             // [ch][8] -> act_conv1 [c][128] (c < 4), val_conv1 [c - 4][128]
            const int ch = r >> 3, c = r & 7;
            if (c < 4) dst = L.off[6] + c * 128 + ch;
            else if (c < 6) dst = L.off[10] + (c - 4) * 128 + ch;
        } else if (sg.kind == GS_HEAD_B) {
            if (r < 4) dst = L.off[7] + r;
            else if (r < 6) dst = L.off[11] + r - 4;
        } else if (sg.kind == GS_FC_W) {               // [k][out] -> [out][k];  a = tensor, b = outputs
            const int k = r / sg.b, o = r - k * sg.b, K = sg.size / sg.b;
            dst = L.off[sg.a] + o * K + k;
        } else {                                       // val_fc2: columns 0-63 weight, 64 bias
            if (r < 64) dst = L.off[14] + r;
            else if (r == 64) dst = L.off[15];
        }
        if (dst >= 0) {
            const int per = (sg.splits + 3) / 4, z0 = zg * per, z1 = min(sg.splits, z0 + per);
#pragma unroll 8
            for (int z = z0; z < z1; z++) s += sg.part[(size_t)z * sg.size + r];
        }
    }
    quarter[zg][e] = s;
    __syncthreads();
    if (zg == 0 && dst >= 0) flat[dst] = ((quarter[0][e] + quarter[1][e]) + quarter[2][e]) + quarter[3][e];
}

// loss = mean value loss + mean policy loss, entropy = mean entropy (policy_value_net_pytorch.py:128-147)
__global__ __launch_bounds__(256) void loss_kernel(const float *parts, int n, float *out)
{
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = tid; b < n; b += 256)
        for (int q = 0; q < 3; q++) s[q] += (double)parts[3 * b + q];
    for (int q = 0; q < 3; q++) red[q][tid] = s[q];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st)
            for (int q = 0; q < 3; q++) red[q][tid] += red[q][tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = (float)(red[0][0] / n + red[1][0] / n);
        out[1] = (float)(red[2][0] / n);
    }
}

// torch.optim.Adam (weight_decay added to the gradient, no amsgrad) on the flat state:
//   g += wd * p;  m = m + (1 - b1) * (g - m);  v = b2 * v + ((1 - b2) * g) * g;
//   p += (-(lr / (1 - b1^t)) * m) / (sqrt(v) / sqrt(1 - b2^t) + eps)        (the operation order of its float32 kernels)
// The KL test of train.py:187-193: mean over the batch of the samples' sums (fc_sample_kernel's klparts),
// in double, summed in a fixed order (one workgroup).
__global__ __launch_bounds__(256) void kl_kernel(const double *parts, int n, double *out)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += parts[i];
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) out[0] = red[0] / (double)n;
}

struct AdamArgs {
    const float *grad;
    float *m, *v;
    float one_minus_b1, b2, one_minus_b2, neg_step_size, bc2_sqrt, eps, wd;
    int total;
};
__global__ void adam_kernel(TrainParams P, ParamLayout L, AdamArgs a)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.total) return;
    int t = 0;
#pragma unroll
    for (int i = 1; i < 16; i++) t += idx >= L.off[i];
    float *pp = P.p[t] + (idx - L.off[t]);
    const float p = *pp;
    const float g = a.grad[idx] + a.wd * p;
    const float m = a.m[idx] + a.one_minus_b1 * (g - a.m[idx]);
    const float v = a.b2 * a.v[idx] + (a.one_minus_b2 * g) * g;
    a.m[idx] = m;
    a.v[idx] = v;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    *pp = p + (a.neg_step_size * m) / denom;
}

}  // namespace azg
