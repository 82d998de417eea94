// Policy/value CNN forward on the f16 matrix cores with float32-grade accuracy.
//
// Same network and same GEMM view as azg_cnn.hpp (D[co][pos] = sum_k W[co][k] X[k][pos]),
// but every float32 operand is carried as a pair of halves, x = hi + lo with
// hi = half(x), lo = half(x - hi), and a product is evaluated as
//     w*x  ~=  w_hi*x_hi + w_lo*x_hi + w_hi*x_lo          (f32 accumulation)
// on v_mfma_f32_32x32x16_f16.  The dropped term w_lo*x_lo is below 2^-22 of the
// product, so the result matches a float32 convolution to ~1e-7 while running three
// MFMAs at 16x the f32-input MFMA rate.  Weights are pre-scaled by 2^8 before the
// split so that the low halves stay out of the f16 subnormal range; the accumulator is
// scaled back (exactly) in the epilogue.  The feature planes are exactly 0/1, so
// conv1 needs no x_lo term.
//
// LDS images are [position][channel] halves (a hi image and a lo image per layer), one
// 16-byte read per lane per MFMA operand; rows are padded by 16 bytes so that the 16
// lanes of a ds_read_b128 group spread over the banks.  There is no halo: every image
// ends with one all-zero row, and a tap that falls off the board reads that row
// instead (one address select per position tile and tap).  This keeps the 8x8
// footprint at ~70 KB so that two workgroups share a CU and one's tree / FC / softmax
// phases overlap the other's MFMA phases.
#pragma once
#include "azg_cnn.hpp"

namespace azg {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr float kWScale = 256.0f;            // weights are stored as half(w * 2^8)
constexpr float kWUnscale = 1.0f / 256.0f;

template <int N_, int T_>
struct Cnn16Geom {
    static constexpr int N = N_, T = T_, S = N_ * N_;
    static constexpr int NP = T_ * S;                         // positions per workgroup
    static constexpr int NT = (NP + 31) / 32;
    static constexpr int NTP = NT * 32;
    static constexpr int ROWS = NP + 1;                       // + the all-zero row
    static constexpr bool BPF = false;                        // explicit LDS prefetch (costs 8 registers per tile)
    // row strides in bytes (channels * 2 + 16 pad)
    static constexpr int RS_IN = 16 * 2 + 16, RS_A1 = 32 * 2 + 16, RS_A2 = 64 * 2 + 16;
    // LDS map in bytes
    static constexpr int IN_OFF = 0;                          // hi only (inputs are exact)
    static constexpr int A1H_OFF = IN_OFF + ROWS * RS_IN;
    static constexpr int A1L_OFF = A1H_OFF + ROWS * RS_A1;
    static constexpr int A2H_OFF = A1L_OFF + ROWS * RS_A1;
    static constexpr int A2L_OFF = A2H_OFF + ROWS * RS_A2;
    static constexpr int WH_OFF = A2L_OFF + ROWS * RS_A2;     // head weights f32 [6][128]
    static constexpr int END_BYTES = WH_OFF + 6 * 128 * 4;
    static constexpr int END = (END_BYTES + 3) / 4;           // in floats, like CnnGeom::END
    // f32 scratch that aliases IN + A1 once conv2 has consumed them (float offsets)
    static constexpr int VL = (2 * S + 3) & ~3;
    static constexpr int HFS = 4 * S + VL;
    static constexpr int HP_OFF = 0;
    static constexpr int HF_OFF = HP_OFF + 24 * NTP;
    static constexpr int LG_OFF = HF_OFF + T_ * HFS;
    static constexpr int HD_OFF = LG_OFF + T_ * S;
    static constexpr int VAL_OFF = HD_OFF + T_ * 64;
    static_assert((VAL_OFF + T_) * 4 <= A2H_OFF, "scratch does not fit under IN+A1");
    static_assert(T_ <= 4, "one wave per position in the softmax stage");
    static_assert(END_BYTES % 8 == 0, "GameLds behind the CNN region needs 8-byte alignment");
};

// One 3x3 convolution layer on split-f16 operands.  CINP = input channels padded to a
// multiple of 16.  wp: packed weights, per (cotile, kstep = tap*CG + cg) and lane 16
// halves: 8 hi then 8 lo of W[co = cotile*32 + (lane&31)][ci = cg*16 + 8*(lane>>5) + j].
// XLO = false skips the w_hi*x_lo term (exact inputs).  xh / xl: hi / lo image
// ([NP rows + zero row][RS bytes]).
template <class G, int CINP, int COUT, int RS, bool XLO, bool BPF, class Epi>
__device__ inline void conv3x3_f16(const half8 *wp, const char *xh, const char *xl, int wave, int lane,
                                   Epi epi)
{
    constexpr int CT = COUT / 32;
    constexpr int WPC = 4 / CT;
    constexpr int PT = (G::NT + WPC - 1) / WPC;
    constexpr int CG = CINP / 16;
    constexpr int KS = 9 * CG;
    const int cot = wave % CT;
    const int sub = wave / CT;
    const int half = lane >> 5;
    const int j = lane & 31;
    const int zero = G::NP * RS + half * 16;

    int base[PT];                        // byte offset of this lane's position row (+ lane-half offset)
    unsigned vmask[PT];                  // bit t set: tap t of this position is on the board
    int prow[PT];                        // position row index, -1 for padding lanes
#pragma unroll
    for (int i = 0; i < PT; i++) {
        const int tile = sub * PT + i;
        const int p = tile * 32 + j;
        base[i] = zero; vmask[i] = 0; prow[i] = -1;
        if (tile < G::NT && p < G::NP) {
            const int leaf = p / G::S, cell = p - leaf * G::S;
            const int row = cell / G::N, col = cell - row * G::N;
            base[i] = p * RS + half * 16;
            prow[i] = p;
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int r2 = row + t / 3 - 1, c2 = col + t % 3 - 1;
                if (r2 >= 0 && r2 < G::N && c2 >= 0 && c2 < G::N) m |= 1u << t;
            }
            vmask[i] = m;
        }
    }
    auto addr = [&](int i, int ks) {
        const int tap = ks / CG, cg = ks - tap * CG;
        const int off = ((tap / 3 - 1) * G::N + (tap % 3 - 1)) * RS;
        return (((vmask[i] >> tap) & 1u) ? base[i] + off : zero) + cg * 32;
    };

    f32x16 acc[PT];
#pragma unroll
    for (int i = 0; i < PT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // Software pipeline, fully unrolled (KS <= 36) so that every ring index and every
    // address offset is a compile-time constant: weights (global, L2 resident) run three
    // k-steps ahead in a ring of three register pairs, activations (LDS) one k-step ahead
    // in two register sets, so the MFMAs of a step never wait for a load of that step.
    const gptr<half8> wrow = as_global(wp + ((size_t)cot * KS * 64 + lane) * 2);
    half8 ah[3], al[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        if (d < KS) { ah[d] = wrow[(size_t)d * 128]; al[d] = wrow[(size_t)d * 128 + 1]; }
    }
    // BPF (activation prefetch) costs 8*PT registers; the two-workgroups-per-CU builds drop it
    // and let the other workgroup's waves cover the LDS latency instead.
    constexpr int NB = BPF ? 2 : 1;
    half8 bh[NB][PT], bl[NB][PT];
    if (BPF) {
#pragma unroll
        for (int i = 0; i < PT; i++) {
            bh[0][i] = *reinterpret_cast<const half8 *>(xh + addr(i, 0));
            if (XLO) bl[0][i] = *reinterpret_cast<const half8 *>(xl + addr(i, 0));
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const int cb = BPF ? (ks & 1) : 0, nb = BPF ? (cb ^ 1) : 0, ca = ks % 3;
        const int lk = BPF ? ks + 1 : ks;                      // k-step whose activations are loaded now
        if (lk < KS) {
#pragma unroll
            for (int i = 0; i < PT; i++) {
                const int a = addr(i, lk);
                bh[nb][i] = *reinterpret_cast<const half8 *>(xh + a);
                if (XLO) bl[nb][i] = *reinterpret_cast<const half8 *>(xl + a);
            }
        }
        const half8 wa = ah[ca], wl = al[ca];
        if (ks + 3 < KS) { ah[ca] = wrow[(size_t)(ks + 3) * 128]; al[ca] = wrow[(size_t)(ks + 3) * 128 + 1]; }
        // keep the loads above and the MFMAs below in this order: left alone, the
        // scheduler sinks each load to just before its first use and waits for it there
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PT; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, bh[cb][i], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[cb][i], acc[i], 0, 0, 0);
            if (XLO) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, bl[cb][i], acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < PT; i++)
        if (prow[i] >= 0) epi(i, sub * PT + i, acc[i], prow[i]);
}

// Full forward; interface as cnn_forward (azg_cnn.hpp): results at lds[LG_OFF..] / lds[VAL_OFF..]
// (float offsets).  Needs cnn16_prepare once per workgroup.
template <class G, class InputFn>
__device__ inline void cnn16_forward(const NetWeights &nw_in, float *lds, InputFn input, Prof &prof)
{
    const NetWeights nw = opaque_weights(nw_in);
    char *ldb = reinterpret_cast<char *>(lds);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, j = lane & 31;
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- input image: the 4 planes as halves in channels 0..3 of a 16-channel row; the
    //      zero rows of the IN and A1 images are rewritten (the f32 scratch overlaps them)
    for (int p = tid; p < G::NP; p += 256) {
        const int leaf = p / G::S, cell = p - leaf * G::S;
        const int row = cell / G::N, col = cell - row * G::N;
        half8 v0 = z8;
#pragma unroll
        for (int ch = 0; ch < 4; ch++) v0[ch] = (_Float16)input(leaf, ch, row, col);
        *reinterpret_cast<half8 *>(ldb + G::IN_OFF + p * G::RS_IN) = v0;
        *reinterpret_cast<half8 *>(ldb + G::IN_OFF + p * G::RS_IN + 16) = z8;
    }
    if (tid < G::RS_IN / 16) *reinterpret_cast<half8 *>(ldb + G::IN_OFF + G::NP * G::RS_IN + tid * 16) = z8;
    if (tid >= 64 && tid < 64 + G::RS_A1 / 16)
        *reinterpret_cast<half8 *>(ldb + G::A1H_OFF + G::NP * G::RS_A1 + (tid - 64) * 16) = z8;
    if (tid >= 128 && tid < 128 + G::RS_A1 / 16)
        *reinterpret_cast<half8 *>(ldb + G::A1L_OFF + G::NP * G::RS_A1 + (tid - 128) * 16) = z8;
    __syncthreads();
    prof.mark(2);

    // epilogue of conv1/conv2: scale back, bias, ReLU, split into halves, store [pos][channel]
    auto store_split = [&](const float *bias, int cobase, char *oh, char *ol, int rs) {
        return [=](int, int, const f32x16 &acc, int p) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                half4 h, l;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int co = cobase + 8 * g + 4 * half + e;
                    float v = __builtin_fmaf(acc[4 * g + e], kWUnscale, bias[co]);
                    v = v > 0.0f ? v : 0.0f;
                    const _Float16 hh = (_Float16)v;
                    h[e] = hh;
                    l[e] = (_Float16)(v - (float)hh);
                }
                const int off = p * rs + (cobase + 8 * g + 4 * half) * 2;
                *reinterpret_cast<half4 *>(oh + off) = h;
                *reinterpret_cast<half4 *>(ol + off) = l;
            }
        };
    };
    conv3x3_f16<G, 16, 32, G::RS_IN, false, G::BPF>(reinterpret_cast<const half8 *>(nw.w16[0]), ldb + G::IN_OFF,
                                            ldb + G::IN_OFF, wave, lane,
                                            store_split(nw.bconv[0], 0, ldb + G::A1H_OFF, ldb + G::A1L_OFF, G::RS_A1));
    __syncthreads();
    prof.mark(3);
    conv3x3_f16<G, 32, 64, G::RS_A1, true, G::BPF>(reinterpret_cast<const half8 *>(nw.w16[1]), ldb + G::A1H_OFF,
                                           ldb + G::A1L_OFF, wave, lane,
                                           store_split(nw.bconv[1], (wave % 2) * 32, ldb + G::A2H_OFF,
                                                       ldb + G::A2L_OFF, G::RS_A2));
    __syncthreads();
    prof.mark(4);

    // ---- conv3 with the two 1x1 heads reduced from the accumulators (float32 VALU)
    {
        const float *wh = lds + G::WH_OFF / 4;
        const float *b3 = nw.bconv[2];
        float *hp = lds + G::HP_OFF + wave * 6 * G::NTP;
        conv3x3_f16<G, 64, 128, G::RS_A2, true, G::BPF>(reinterpret_cast<const half8 *>(nw.w16[2]), ldb + G::A2H_OFF,
                                                ldb + G::A2L_OFF, wave, lane,
            [=](int, int tile, const f32x16 &acc, int) {
                float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int co = wave * 32 + acc_row(r, half);
                    float v = __builtin_fmaf(acc[r], kWUnscale, b3[co]);
                    v = v > 0.0f ? v : 0.0f;
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

// One-time LDS set-up: the zero rows of the A2 images and the head weights.
template <class G>
__device__ inline void cnn16_prepare(const NetWeights &nw, float *lds)
{
    char *ldb = reinterpret_cast<char *>(lds);
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int tid = threadIdx.x;
    if (tid < G::RS_A2 / 16) *reinterpret_cast<half8 *>(ldb + G::A2H_OFF + G::NP * G::RS_A2 + tid * 16) = z8;
    if (tid >= 64 && tid < 64 + G::RS_A2 / 16)
        *reinterpret_cast<half8 *>(ldb + G::A2L_OFF + G::NP * G::RS_A2 + (tid - 64) * 16) = z8;
    for (int idx = tid; idx < 6 * 128; idx += 256) lds[G::WH_OFF / 4 + idx] = nw.whead[idx];
    __syncthreads();
}

}  // namespace azg
