// Policy/value CNN forward for T positions per 256-thread workgroup, float32 end to end
// on the f32-input matrix cores (v_mfma_f32_32x32x2_f32), activations resident in LDS.
//
// Network (policy_value_net_pytorch.py:23-57): conv3x3 4->32->64->128 (pad 1, ReLU),
// policy head conv1x1 128->4 + ReLU + FC 4S->S + log-softmax (exp'ed by the caller at
// :88/:112), value head conv1x1 128->2 + ReLU + FC 2S->64 + ReLU + FC 64->1 + tanh.
//
// GEMM view of a 3x3 conv: D[co][pos] = sum_k W[co][k] * X[k][pos], k = (ci, tap).
// The MFMA A operand is the packed weight stream (global memory, L2 resident, read
// once per workgroup), the B operand is the activation image in LDS stored per
// channel with a one-cell zero halo, so a tap is a constant address offset and no
// bounds test is needed.  D comes out with the position on the lane and the output
// channel on the register index, which is also the layout the next layer reads.
// conv3's output never goes to LDS: the two 1x1 heads are reduced straight from the
// accumulators.
#pragma once
#include "azg_device.hpp"

namespace azg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N_, int T_>
struct CnnGeom {
    static constexpr int N = N_, T = T_, S = N_ * N_;
    static constexpr int PW = N_ + 2, PS = PW * PW;          // haloed image
    static constexpr int NP = T_ * S;                         // positions per workgroup
    static constexpr int NT = (NP + 31) / 32;                 // 32-position tiles
    static constexpr int NTP = NT * 32;
    static constexpr int CS = T_ * PS;                        // channel stride (floats)
    // LDS map (floats)
    static constexpr int IN_OFF = 0;                          // [4][CS]
    static constexpr int A1_OFF = IN_OFF + 4 * CS;            // [32][CS]
    static constexpr int A2_OFF = A1_OFF + 32 * CS;           // [64][CS]
    static constexpr int WH_OFF = A2_OFF + 64 * CS;           // head weights [6][128]
    static constexpr int END = WH_OFF + 6 * 128;
    // scratch that aliases IN + A1 once conv2 has consumed them
    static constexpr int HP_OFF = 0;                          // head partials [4 waves][6][NTP]
    static constexpr int VL = (2 * S + 3) & ~3;               // value-head features padded to 4
    static constexpr int HFS = 4 * S + VL;                    // head outputs per position
    static constexpr int HF_OFF = HP_OFF + 24 * NTP;          // head outputs [T][HFS]
    static constexpr int LG_OFF = HF_OFF + T_ * HFS;          // logits -> probs [T][S]
    static constexpr int HD_OFF = LG_OFF + T_ * S;            // value hidden [T][64]
    static constexpr int VAL_OFF = HD_OFF + T_ * 64;          // values [T]
    static_assert(VAL_OFF + T_ <= A2_OFF, "scratch does not fit under IN+A1");
    static_assert(T_ <= 4, "one wave per position in the softmax stage");
};

// register r of a 32x32 accumulator tile holds output row (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ inline int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// One 3x3 convolution layer.  wave w works on output-channel tile (w % CT) and on the
// position tiles sub*PT .. sub*PT+PT-1 with sub = w / CT.  `epi(i, tile, acc)` consumes
// the finished accumulator of the wave's i-th tile.
template <class G, int CIN, int COUT, class Epi>
__device__ inline void conv3x3_tiles(const float *__restrict__ wp, const float *__restrict__ bias,
                                     const float *in_lds, int wave, int lane, Epi epi)
{
    constexpr int CT = COUT / 32;
    constexpr int WPC = 4 / CT;
    constexpr int PT = (G::NT + WPC - 1) / WPC;
    constexpr int KS = (CIN / 2) * 9;
    const int cot = wave % CT;
    const int sub = wave / CT;
    const int half = lane >> 5;
    const int j = lane & 31;

    int posoff[PT];
#pragma unroll
    for (int i = 0; i < PT; i++) {
        const int tile = sub * PT + i;
        const int p = tile * 32 + j;
        int off = 0;
        if (tile < G::NT && p < G::NP) {
            const int leaf = p / G::S, cell = p - leaf * G::S;
            const int row = cell / G::N, col = cell - row * G::N;
            off = leaf * G::PS + row * G::PW + col;
        }
        posoff[i] = off;
    }

    f32x16 acc[PT];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float b = bias[cot * 32 + acc_row(r, half)];
#pragma unroll
        for (int i = 0; i < PT; i++) acc[i][r] = b;
    }

    const gptr<float> wrow = as_global(wp + (size_t)cot * KS * 64 + lane);
    float a_cur[9], a_nxt[9];
#pragma unroll
    for (int t = 0; t < 9; t++) a_cur[t] = wrow[t * 64];
    for (int cp = 0; cp < CIN / 2; cp++) {
        if (cp + 1 < CIN / 2) {
#pragma unroll
            for (int t = 0; t < 9; t++) a_nxt[t] = wrow[((cp + 1) * 9 + t) * 64];
        }
        const float *bin = in_lds + (2 * cp + half) * G::CS;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int tap = (t / 3) * G::PW + (t % 3);
#pragma unroll
            for (int i = 0; i < PT; i++) {
                const float b = bin[posoff[i] + tap];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], b, acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 9; t++) a_cur[t] = a_nxt[t];
    }
#pragma unroll
    for (int i = 0; i < PT; i++) {
        const int tile = sub * PT + i;
        if (tile < G::NT) epi(i, tile, acc[i], posoff[i]);
    }
}

// Everything after conv3: sum the head partials of the four channel tiles, the fully
// connected layers, softmax and tanh.  Shared by the f32 and the split-f16 convolution
// front ends (G supplies the scratch offsets, in floats).
template <class G>
__device__ inline void cnn_tail(const NetWeights &nw, float *lds, Prof &prof)
{
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // ---- head outputs: bias + the four channel-tile partials, ReLU, flattened [c][row][col]
    for (int idx = tid; idx < 6 * G::NP; idx += 256) {
        const int c = idx / G::NP, p = idx - c * G::NP;
        float s = nw.bhead[c];
#pragma unroll
        for (int w = 0; w < 4; w++) s += lds[G::HP_OFF + (w * 6 + c) * G::NTP + p];
        s = s > 0.0f ? s : 0.0f;
        const int leaf = p / G::S, cell = p - leaf * G::S;
        lds[G::HF_OFF + leaf * G::HFS + c * G::S + cell] = s;
    }
    if constexpr (G::VL > 2 * G::S) if (tid < G::T * (G::VL - 2 * G::S)) {   // zero the k padding of the value part
        const int leaf = tid / (G::VL - 2 * G::S), e = tid - leaf * (G::VL - 2 * G::S);
        lds[G::HF_OFF + leaf * G::HFS + 6 * G::S + e] = 0.0f;
    }
    __syncthreads();
    prof.mark(7);

    // ---- fully connected layers: policy logits (4S -> S) and value hidden (2S -> 64).
    //      One output per thread; the weights are packed [k/4][output][4] so a thread
    //      reads 16 bytes per load and keeps 16 loads (64 weights) in flight.
    {
        auto dot = [&](gptr<f32x4> w4, int stride, const f32x4 *x4, int nkb, float bias) {
            float s0 = bias, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            for (int kb0 = 0; kb0 < nkb; kb0 += 16) {
                f32x4 wv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int kb = kb0 + u < nkb ? kb0 + u : nkb - 1;      // clamp: keeps the load unconditional
                    wv[u] = w4[(size_t)kb * stride];
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    if (kb0 + u < nkb) {
                        const f32x4 xv = x4[kb0 + u];
                        s0 = __builtin_fmaf(wv[u][0], xv[0], s0);
                        s1 = __builtin_fmaf(wv[u][1], xv[1], s1);
                        s2 = __builtin_fmaf(wv[u][2], xv[2], s2);
                        s3 = __builtin_fmaf(wv[u][3], xv[3], s3);
                    }
                }
            }
            return (s0 + s1) + (s2 + s3);
        };
        for (int item = tid; item < G::T * G::S + G::T * 64; item += 256) {
            if (item < G::T * G::S) {
                const int leaf = item / G::S, o = item - leaf * G::S;
                const f32x4 *x4 = reinterpret_cast<const f32x4 *>(lds + G::HF_OFF + leaf * G::HFS);
                lds[G::LG_OFF + item] = dot(as_global(reinterpret_cast<const f32x4 *>(nw.wpfc) + o), G::S, x4, G::S, nw.bpfc[o]);
            } else {
                const int q = item - G::T * G::S;
                const int leaf = q >> 6, o = q & 63;
                const f32x4 *x4 = reinterpret_cast<const f32x4 *>(lds + G::HF_OFF + leaf * G::HFS + 4 * G::S);
                const float s = dot(as_global(reinterpret_cast<const f32x4 *>(nw.wvfc1) + o), 64, x4, G::VL / 4, nw.bvfc1[o]);
                lds[G::HD_OFF + q] = s > 0.0f ? s : 0.0f;
            }
        }
    }
    __syncthreads();
    prof.mark(8);

    // ---- softmax over the S logits and the scalar value, one wave per position
    if (wave < G::T) {
        const int leaf = wave;
        constexpr int PER = (G::S + 63) / 64;
        float x[PER];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int a = lane + 64 * i;
            x[i] = a < G::S ? lds[G::LG_OFF + leaf * G::S + a] : -3.0e38f;
            mx = fmaxf(mx, x[i]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int a = lane + 64 * i;
            if (a < G::S) sum += expf(x[i] - mx);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        const float lse = logf(sum);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int a = lane + 64 * i;
            if (a < G::S) lds[G::LG_OFF + leaf * G::S + a] = expf((x[i] - mx) - lse);
        }
        float v = lds[G::HD_OFF + leaf * 64 + lane] * nw.wvfc2[lane];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) lds[G::VAL_OFF + leaf] = tanhf(v + nw.bvfc2[0]);
    }
    __syncthreads();
    prof.mark(9);
}

// Full forward.  `input(leaf, plane, row, col)` supplies the 0/1 feature planes
// (game.py:92-116 layout).  Results: lds[LG_OFF + leaf*S + a] = probability of move a,
// lds[VAL_OFF + leaf] = value.  Every thread of the 256-thread workgroup must call
// this; it ends with a __syncthreads().  lds[A2_OFF..] halos must be zero on entry
// (cnn_prepare) and lds[WH_OFF..] must hold the head weights.
template <class G, class InputFn>
__device__ inline void cnn_forward(const NetWeights &nw_in, float *lds, InputFn input, Prof &prof)
{
    const NetWeights nw = opaque_weights(nw_in);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, j = lane & 31;

    // ---- input planes with halo; A1 cleared (its halo must read as zero)
    for (int idx = tid; idx < 4 * G::CS; idx += 256) {
        const int ch = idx / G::CS, rem = idx - ch * G::CS;
        const int leaf = rem / G::PS, q = rem - leaf * G::PS;
        const int hr = q / G::PW, hc = q - hr * G::PW;
        float v = 0.0f;
        if (hr >= 1 && hr <= G::N && hc >= 1 && hc <= G::N) v = input(leaf, ch, hr - 1, hc - 1);
        lds[G::IN_OFF + idx] = v;
    }
    for (int idx = tid; idx < 32 * G::CS; idx += 256) lds[G::A1_OFF + idx] = 0.0f;
    __syncthreads();
    prof.mark(2);

    // ---- conv1 4 -> 32, conv2 32 -> 64: ReLU, store with the halo offset
    auto store_relu = [&](float *out) {
        return [=](int, int tile, const f32x16 &acc, int posoff) {
            const int cotile = 0;
            (void)cotile;
            if (tile * 32 + j < G::NP) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = acc[r] > 0.0f ? acc[r] : 0.0f;
                    out[acc_row(r, half) * G::CS + posoff + G::PW + 1] = v;
                }
            }
        };
    };
    conv3x3_tiles<G, 4, 32>(nw.wconv[0], nw.bconv[0], lds + G::IN_OFF, wave, lane,
                            store_relu(lds + G::A1_OFF));
    __syncthreads();
    prof.mark(3);
    conv3x3_tiles<G, 32, 64>(nw.wconv[1], nw.bconv[1], lds + G::A1_OFF, wave, lane,
                             store_relu(lds + G::A2_OFF + (wave % 2) * 32 * G::CS));
    __syncthreads();
    prof.mark(4);

    // ---- conv3 64 -> 128 with the 1x1 heads folded into the epilogue
    {
        const float *wh = lds + G::WH_OFF;
        float *hp = lds + G::HP_OFF + wave * 6 * G::NTP;
        conv3x3_tiles<G, 64, 128>(nw.wconv[2], nw.bconv[2], lds + G::A2_OFF, wave, lane,
            [=](int, int tile, const f32x16 &acc, int) {
                float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = acc[r] > 0.0f ? acc[r] : 0.0f;
                    const int co = wave * 32 + acc_row(r, half);
#pragma unroll
                    for (int c = 0; c < 6; c++) s[c] = __builtin_fmaf(wh[c * 128 + co], v, s[c]);
                }
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const float o = __shfl_xor(s[c], 32);
                    const float lo = half ? o : s[c], hi = half ? s[c] : o;
                    if (!half) hp[c * G::NTP + tile * 32 + j] = lo + hi;
                }
            });
    }
    prof.mark(5);
    __syncthreads();
    prof.mark(6);

    cnn_tail<G>(nw, lds, prof);
}

// One-time LDS set-up for a workgroup that will call cnn_forward: zero the A2 image
// (halo included) and stage the head weights.
template <class G>
__device__ inline void cnn_prepare(const NetWeights &nw, float *lds)
{
    for (int idx = threadIdx.x; idx < 64 * G::CS; idx += 256) lds[G::A2_OFF + idx] = 0.0f;
    for (int idx = threadIdx.x; idx < 6 * 128; idx += 256) lds[G::WH_OFF + idx] = nw.whead[idx];
    __syncthreads();
}

}  // namespace azg
