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
// footprint at ~72 KB so that two workgroups share a CU and one's tree / FC / softmax
// phases overlap the other's MFMA phases.
#pragma once
#include <type_traits>
#include "azg_cnn.hpp"

namespace azg {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// ReLU that lets a NaN through (max(x, 0) would replace it by 0).  An activation beyond the
// half-precision range gets an infinite high half and a low half of the opposite infinity; the next
// layer then forms inf - inf = NaN, and because every ReLU on the way passes it on, the NaN reaches
// all outputs of that position, where the kernels test for it (error AZG_ERANGE) instead of
// returning numbers computed from clipped activations.
__device__ inline float relu_nan(float v) { return v < 0.0f ? 0.0f : v; }

// Two float32 values as split-f16 pairs: h = half(v), l = half(v - h), packed (v0 in the low
// halves).  One packed conversion gives h; v_fma_mix{lo,hi}_f16 then takes a HALF of h as a source,
// widens it exactly, forms h * -1.0 + v in float32 (exact: the difference has at most 13 significant
// bits) and rounds once to f16 -- bit for bit what (half)(v - (float)h) computes in three instructions
// (checked on the GPU over 2^23 pairs including denormal, infinite and NaN inputs).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void split2_f16(float v0, float v1, unsigned &h, unsigned &l)
{
    const f32x2 v = {v0, v1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2v));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(v1));
}

// Output-channel tiles per wave in conv2 / conv3 (see ConvSplit): one for the two-leaf kernels, two
// for the one-leaf kernels (boards above 8x8: eight position tiles, whose accumulators would not fit
// beside the operands in the 256 registers a wave has when eight waves share a CU).  Every kernel of
// a board size -- search, evaluation -- uses the same value: the grouping of the head convolutions'
// partial sums follows it, and a position must evaluate to the same bits wherever it is evaluated.
#ifndef AZG_MFMA_ROTATE
#define AZG_MFMA_ROTATE 0     // 1: the k-step's products are issued accumulator after accumulator in rotation
#endif
#ifndef AZG_CW1
#define AZG_CW1 2
#endif
#ifndef AZG_CW2
#define AZG_CW2 1
#endif
#ifndef AZG_CW1_3
#define AZG_CW1_3 AZG_CW1     // conv3 of the one-leaf kernels (conv2 has two channel tiles in all, conv3 four)
#endif
template <class G>
struct ConvWaves {
    static constexpr int CW = G::T == 1 ? AZG_CW1 : AZG_CW2;            // conv2
    static constexpr int CW3 = G::T == 1 ? AZG_CW1_3 : AZG_CW2;         // conv3
    static constexpr int HPARTS = 4 / CW3;                               // channel groups of conv3 = partial sums per head output
};
// sum of a position's PARTS head partials (blocks of 8 rows, ntp floats per row, apart)
template <int PARTS>
__device__ inline float head_sum(const float *pp, int ntp)
{
    if (PARTS == 4) return (pp[0] + pp[8 * ntp]) + (pp[16 * ntp] + pp[24 * ntp]);
    if (PARTS == 2) return pp[0] + pp[8 * ntp];
    return pp[0];
}

constexpr float kWScale = 256.0f;            // weights are stored as half(w * 2^8)
constexpr float kWUnscale = 1.0f / 256.0f;

template <int N_, int T_>
struct Cnn16Geom {
    static constexpr int N = N_, T = T_, S = N_ * N_;
    static constexpr int NP = T_ * S;                         // positions per workgroup
    static constexpr int NT = (NP + 31) / 32;
    static constexpr int NTP = NT * 32;
    static constexpr int ROWS = NP + 1;                       // + the all-zero row
    // activation (LDS) prefetch of the 3x3 layers: 0 none, 1 one k-step ahead (two register sets of
    // all the wave's position tiles), 2 one position tile ahead (two sets of ONE tile)
#ifndef AZG_BPF
#define AZG_BPF 1
#endif
    static constexpr int BPF = AZG_BPF;
    // row strides in bytes (channels * 2 + 16 pad)
    static constexpr int RS_IN = 16 * 2 + 16, RS_A1 = 32 * 2 + 16, RS_A2 = 64 * 2 + 16;
    // LDS map in bytes
    static constexpr int IN_OFF = 0;                          // hi only (inputs are exact)
    static constexpr int A1H_OFF = IN_OFF + ROWS * RS_IN;
    static constexpr int A1L_OFF = A1H_OFF + ROWS * RS_A1;
    static constexpr int A2H_OFF = A1L_OFF + ROWS * RS_A1;
    static constexpr int A2L_OFF = A2H_OFF + ROWS * RS_A2;
    static constexpr int WH_OFF = A2L_OFF + ROWS * RS_A2;     // head weights as halves (4 KB, layout of pack_head16)
    // f32 constants: conv1 bias [32], conv2 [64], conv3 [128], heads [6 + 2 pad], val_fc1 bias [64],
    // val_fc2 weights [64], val_fc2 bias [1 + 7 pad], act_fc1 bias [S padded to 8]
    static constexpr int BIAS_OFF = WH_OFF + 4096;
    static constexpr int B_HEAD = 224, B_VFC1 = 232, B_WVFC2 = 296, B_VFC2 = 360, B_PFC = 368;
    static constexpr int END_BYTES = BIAS_OFF + (B_PFC + ((S + 7) & ~7)) * 4;
    static constexpr int END = (END_BYTES + 3) / 4;           // in floats, like CnnGeom::END
    // fully connected layers as MFMA GEMMs on v_mfma_f32_16x16x32_f16: D[leaf][output] =
    // sum_k x[leaf][k] * W[output][k], 16 leaf rows (T in use), 16 outputs and 32 inputs per
    // instruction; every wave takes a quarter of the k-steps of every output tile
    static constexpr int KWP = ((4 * S + 31) / 32 + 3) / 4;   // k-steps per wave, act_fc1
    static constexpr int KWV = ((2 * S + 31) / 32 + 3) / 4;   // k-steps per wave, val_fc1
    static constexpr int KPP = 4 * KWP * 32, KPV = 4 * KWV * 32;   // padded input lengths
    static constexpr int OTP = (S + 15) / 16;                 // output tiles of act_fc1 (val_fc1 has 4)
    static constexpr int OC = OTP % 4 == 0 ? 4 : OTP % 3 == 0 ? 3 : OTP % 2 == 0 ? 2 : 1;   // output tiles per accumulator group
    static constexpr int NOUT = OTP * 16 + 64;
    static_assert(OTP % OC == 0, "act_fc1 output tiles must split evenly into accumulator groups");
    // f32 scratch that aliases IN + A1 once conv2 has consumed them (float offsets)
    static constexpr int HP_OFF = 0;                          // head partials [4 waves][8][NTP] (rows 6, 7 unused)
    // One-leaf kernels (T == 1): the fully connected layers run on the VALU over the NON-ZERO head
    // outputs only.  Wave w owns the inputs k = 4*i + w; its (k, x) pairs sit in lists behind the
    // head partials, the per-wave partial sums in the (by then dead) head-partial region.
    static constexpr bool SPARSE_FC = T_ == 1 && N_ > 8;      // (one-leaf builds of the small boards keep the MFMA layers)
    static constexpr int PART_OFF = 0;                        // FC partials [4 waves][NOUT][T] (HP is dead by then)
    static constexpr int SPP_OFF = 0;                         // sparse FC: policy partials [4 waves][256]
    static constexpr int SPV_OFF = 1024;                      // sparse FC: value partials [4 waves][4 lane groups][64]
    static constexpr int LG_OFF = SPARSE_FC ? SPV_OFF + 1024 : PART_OFF + 4 * NOUT * T_;   // probabilities [T][S]
    static constexpr int VAL_OFF = LG_OFF + T_ * S;           // values [T]
    // head outputs as halves: hi [T][KPP], lo [T][KPP], hi [T][KPV], lo [T][KPV]; they start on the last
    // wave's rows 6 and 7 of the head partials, which are written (zeros) before but never read
    static constexpr int X_OFF = HP_OFF + 30 * NTP;
    static constexpr int X_END = SPARSE_FC ? X_OFF : X_OFF + T_ * (KPP + KPV);
    static constexpr int QP = S, QV = (2 * S + 3) / 4;        // sparse FC: list capacities per wave (policy, value)
    static constexpr int LP_OFF = X_OFF, LV_OFF = LP_OFF + 2 * 4 * QP, CNT_OFF = LV_OFF + 2 * 4 * QV, SP_END = CNT_OFF + 8;
    static_assert(!SPARSE_FC || (S <= 256 && SP_END * 4 <= A2H_OFF), "sparse FC scratch does not fit");
    static_assert(VAL_OFF + T_ <= X_OFF, "FC partials and results must fit in the head-partial region");
    static_assert(X_END * 4 <= A2H_OFF, "scratch does not fit under IN+A1");
    static_assert(T_ <= 4, "one wave per position in the softmax stage");
    static_assert(END_BYTES % 8 == 0, "GameLds behind the CNN region needs 8-byte alignment");
};

// Work split of one 3x3 layer over the four waves: a wave owns CW output-channel tiles (of 32) and
// PT position tiles (of 32) and keeps their CW*PT accumulators; per k-step it loads CW weight
// operands (global memory / L2) and PT activation operands (LDS) for CW*PT products.  With CT =
// COUT/32 channel tiles the waves form WC = CT/CW channel groups x WP = 4/WC position groups.
// CW = 2 everywhere: against one channel tile per wave it halves the LDS reads of a k-step (every
// activation operand then feeds two channel tiles) for twice the weight loads -- 8 KB instead of
// 10 KB of operands per k-step of conv3, split evenly between the LDS and the vector-memory path.
template <class G, int COUT, int CW>
struct ConvSplit {
    static constexpr int CT = COUT / 32, WC = CT / CW, WP = 4 / WC;
    static constexpr int PT = (G::NT + WP - 1) / WP;
    static_assert(CT % CW == 0 && 4 % WC == 0, "channel tiles must split evenly over the waves");
};

// The weight (A operand) stream of one wave for one layer runs RING k-steps ahead of the
// MFMAs in a ring of register pairs per channel tile.
template <int RING, int CW>
struct WRing {
    half8 ah[RING][CW], al[RING][CW];
};

// this lane's 16 halves of k-step 0 of channel tile `cot`; k-steps are 128 half8 apart
template <int CINP>
__device__ inline gptr<half8> conv_wrow(const half8 *wp, int cot, int lane)
{
    constexpr int KS = 9 * (CINP / 16);
    return as_global(wp + (size_t)cot * KS * 128 + lane * kWLane);
}

// conv_preload requests the first RING k-steps; it is called well before the layer starts (during
// the previous layer or the input fill) so that the L2 latency of the first k-steps is not exposed
// after the barrier that opens the layer.
template <class G, int CINP, int COUT, int RING, int CW>
__device__ inline void conv_preload(const half8 *wp, int wave, int lane, WRing<RING, CW> &r)
{
    constexpr int KS = 9 * (CINP / 16);
    static_assert(RING <= KS, "ring deeper than the layer");
    const int cg = wave % ConvSplit<G, COUT, CW>::WC;
#pragma unroll
    for (int c = 0; c < CW; c++) {
        const gptr<half8> wrow = conv_wrow<CINP>(wp, cg * CW + c, lane);
#pragma unroll
        for (int d = 0; d < RING; d++) { r.ah[d][c] = wrow[(size_t)d * 128]; r.al[d][c] = wrow[(size_t)d * 128 + kWLo]; }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// One 3x3 convolution layer on split-f16 operands.  CINP = input channels padded to a
// multiple of 16.  wp: packed weights, per (cotile, kstep = tap*CG + cg) and lane 16
// halves: 8 hi then 8 lo of W[co = cotile*32 + (lane&31)][ci = cg*16 + 8*(lane>>5) + j].
// XLO = false skips the w_hi*x_lo term (exact inputs).  xh / xl: hi / lo image
// ([NP rows + zero row][RS bytes]).  ring: filled by conv_preload for the same layer.
// KEEP_PRIO: the caller's waves hold the raised issue priority for their whole life.
// epi(c, cot, tile, acc, prow, ctx) is called for the wave's channel tiles c = 0..CW-1 (global
// index cot) of every position tile, channel tiles innermost.
template <class G, int CINP, int COUT, int RS, bool XLO, int BPF, int RING, int CW, bool KEEP_PRIO = false, class Prep, class Epi>
__device__ inline void conv3x3_f16(const half8 *wp, WRing<RING, CW> &ring, const char *xh, const char *xl, int wave,
                                   int lane, Prep prep, Epi epi, Prof *prof = nullptr, int slot = 11)
{
    using SP = ConvSplit<G, COUT, CW>;
    constexpr int PT = SP::PT;
    constexpr int CG = CINP / 16;
    constexpr int KS = 9 * CG;
    const int cg = wave % SP::WC;
    const int sub = wave / SP::WC;
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
            // taps t = 3*(dr+1) + (dc+1) that stay on the board
            unsigned m = 0x1FFu;
            if (row == 0) m &= ~0x007u;
            if (row == G::N - 1) m &= ~0x1C0u;
            if (col == 0) m &= ~0x049u;
            if (col == G::N - 1) m &= ~0x124u;
            vmask[i] = m;
        }
    }
    auto addr = [&](int i, int ks) {
        const int tap = ks / CG, cgi = ks - tap * CG;
        const int off = ((tap / 3 - 1) * G::N + (tap % 3 - 1)) * RS;
        return (((vmask[i] >> tap) & 1u) ? base[i] + off : zero) + cgi * 32;
    };

    f32x16 acc[CW][PT];
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (prof && slot != 11) prof->mark(slot - 1);

    // Software pipeline, fully unrolled (KS <= 36) so that every ring index and every
    // address offset is a compile-time constant: weights (global, L2 resident) run RING
    // k-steps ahead, activations (LDS) one k-step ahead in two register sets, so the MFMAs
    // of a step never wait for a load of that step.
    gptr<half8> wrow[CW];
#pragma unroll
#ifdef AZG_EXP_SAMECOT           // timing experiment (wrong results): every wave streams channel tile 0 (hits in the vector L1)
    for (int c = 0; c < CW; c++) wrow[c] = conv_wrow<CINP>(wp, 0, lane);
#else
    for (int c = 0; c < CW; c++) wrow[c] = conv_wrow<CINP>(wp, cg * CW + c, lane);
#endif
    // dense phase: take issue priority over the co-resident workgroup's scalar / f64 tree code
#if AZG_QUAD8 && AZG_QUAD_FINE
    if (COUT >= 64) azg_simd_gate_acquire();
#endif
    if (COUT >= 64) __builtin_amdgcn_s_setprio(2);
    if constexpr (BPF == 2) {
        // Activations one POSITION TILE ahead: the steps n = ks * PT + i form one sequence; the LDS
        // reads of step n + 1 are issued behind the first MFMAs of step n (CW * 3 MFMAs, 32 cycles
        // each, cover the LDS latency), into the register set step n - 1 used.
        half8 bh[2], bl[2];
        auto load = [&](int n, int buf) {
            const int a = addr(n % PT, n / PT);
            bh[buf] = *reinterpret_cast<const half8 *>(xh + a);
            if (XLO) bl[buf] = *reinterpret_cast<const half8 *>(xl + a);
        };
        load(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int ca = ks % RING;
            half8 wa[CW], wl[CW];
#pragma unroll
            for (int c = 0; c < CW; c++) {
                wa[c] = ring.ah[ca][c]; wl[c] = ring.al[ca][c];
                if (ks + RING < KS) { ring.ah[ca][c] = wrow[c][(size_t)(ks + RING) * 128]; ring.al[ca][c] = wrow[c][(size_t)(ks + RING) * 128 + kWLo]; }
            }
#pragma unroll
            for (int i = 0; i < PT; i++) {
                const int n = ks * PT + i, cb = n & 1;
                if (n + 1 < KS * PT) load(n + 1, cb ^ 1);
#pragma unroll
                for (int c = 0; c < CW; c++) {
                    acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[c], bh[cb], ks == 0 ? zacc : acc[c][i], 0, 0, 0);
                    acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[c], bh[cb], acc[c][i], 0, 0, 0);
                    if (XLO) acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[c], bl[cb], acc[c][i], 0, 0, 0);
                }
                constexpr int NM = CW * (XLO ? 3 : 2), NDS = XLO ? 2 : 1, NGL = 2 * CW;
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                 // two VALU (addresses)
                    if (m < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one LDS read
                    if (i == 0 && m >= NM - (NGL < NM ? NGL : NM)) __builtin_amdgcn_sched_group_barrier(0x020, NGL <= NM ? 1 : 2, 0);   // the k-step's weight loads
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
    // BPF (activation prefetch one k-step ahead) costs 8*PT registers for the second buffer set
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
        const int cb = BPF ? (ks & 1) : 0, nb = BPF ? (cb ^ 1) : 0, ca = ks % RING;
        const int lk = BPF ? ks + 1 : ks;                      // k-step whose activations are loaded now
        if (lk < KS) {
#pragma unroll
            for (int i = 0; i < PT; i++) {
                const int a = addr(i, lk);
                bh[nb][i] = *reinterpret_cast<const half8 *>(xh + a);
#ifdef AZG_EXP_NOLDSLO          // timing experiment (wrong results): no LDS read of the low halves
                if (XLO) bl[nb][i] = bh[nb][i];
#else
                if (XLO) bl[nb][i] = *reinterpret_cast<const half8 *>(xl + a);
#endif
            }
        }
        half8 wa[CW], wl[CW];
#pragma unroll
        for (int c = 0; c < CW; c++) {
            wa[c] = ring.ah[ca][c]; wl[c] = ring.al[ca][c];
#ifndef AZG_EXP_NOWLOAD          // timing experiment (wrong results): the weight ring is never refilled
            if (ks + RING < KS) { ring.ah[ca][c] = wrow[c][(size_t)(ks + RING) * 128]; ring.al[ca][c] = wrow[c][(size_t)(ks + RING) * 128 + kWLo]; }
#endif
        }
#if AZG_MFMA_ROTATE
        // the three products of a k-step go round the accumulators instead of queueing behind each
        // other on one (same sums in the same order per accumulator)
#pragma unroll
        for (int pr = 0; pr < (XLO ? 3 : 2); pr++)
#pragma unroll
            for (int i = 0; i < PT; i++)
#pragma unroll
                for (int c = 0; c < CW; c++) {
                    if (pr == 0) acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[c], bh[cb][i], ks == 0 ? zacc : acc[c][i], 0, 0, 0);
                    if (pr == 1) acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[c], bh[cb][i], acc[c][i], 0, 0, 0);
                    if (pr == 2) acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[c], bl[cb][i], acc[c][i], 0, 0, 0);
                }
#else
#pragma unroll
        for (int i = 0; i < PT; i++)
#pragma unroll
            for (int c = 0; c < CW; c++) {
                acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[c], bh[cb][i], ks == 0 ? zacc : acc[c][i], 0, 0, 0);
                acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[c], bh[cb][i], acc[c][i], 0, 0, 0);
                if (XLO) acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[c], bl[cb][i], acc[c][i], 0, 0, 0);
            }
#endif
        // Issue order inside the k-step: the loads of later k-steps and their address
        // arithmetic are spread between this step's MFMAs (an MFMA occupies the matrix pipe for
        // 32 cycles but the issue port for 8).  Left alone the scheduler sinks each load to
        // just before its first use and waits for it there; fenced off as one block in front
        // of the MFMAs they leave the matrix pipe idle while they issue.
        if (BPF) {
            constexpr int NM = CW * PT * (XLO ? 3 : 2), NDS = PT * (XLO ? 2 : 1), NGL = 2 * CW;
#pragma unroll
            for (int m = 0; m < NM; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                 // two VALU (addresses)
                if (m < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one LDS read
                if (m >= NM - NGL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // one global read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    }
    if (COUT >= 64 && !KEEP_PRIO) __builtin_amdgcn_s_setprio(0);
#if AZG_QUAD8 && AZG_QUAD_FINE
    if (COUT >= 64) azg_simd_gate_release();
#endif
    if (prof) prof->mark(slot);
#ifdef AZG_EXP_LANE0
    // Hazard experiment: an LDS store directly behind the unrolled MFMA loop, into a padding word of the
    // constants block.  1: by lane 0 alone (volatile: the address stays generic, flat_store_dword);
    // 5: by lane 0 alone as a genuine ds_write_b32; 3: by every lane.  Measured at 8x8: with 1 and 5 the
    // EVALUATION kernel's probabilities are off by 1.2e-3 (the same value every time, also with 32 wait
    // states or a full counter drain in front of the branch), the search kernel of the same build is
    // exact; with 3 both are exact.  Deterministic and indifferent to timing: generated code, not a
    // hardware wait-state window (the pair v_mfma_f32_32x32x16_f16 -> VALU read needs 12 wait states on
    // this part, which is what the compiler pads: tests/micro/mfma_read_wait.hip).
    {
        extern __shared__ __attribute__((aligned(16))) float azg_exp_lds[];
        float *pad = azg_exp_lds + kLdsHead + G::BIAS_OFF / 4 + G::B_VFC2 + 4;
        if (AZG_EXP_LANE0 == 5) {
            if (lane == 0) {
                __attribute__((address_space(3))) float *p3 = (__attribute__((address_space(3))) float *)pad;
                *p3 = 1.0f;
                asm volatile("" ::: "memory");
            }
        } else if (AZG_EXP_LANE0 == 3 || lane == 0) *reinterpret_cast<volatile float *>(pad) = 1.0f;
    }
#endif
    // prep() reads what every tile's epilogue needs from LDS once, before the first epilogue
    // store (the compiler must assume a later read could alias such a store and would
    // otherwise serialise read -> wait -> compute -> store per group of values)
    const auto ctx = prep();
#pragma unroll
    for (int i = 0; i < PT; i++)
        if (sub * PT + i < G::NT)                              // wave-uniform: epilogues may use MFMA
#pragma unroll
            for (int c = 0; c < CW; c++) epi(c, cg * CW + c, sub * PT + i, acc[c][i], prow[i], ctx);
}

// ---------------------------------------------------------------- conv3 in Winograd form
// F(2x2, 3x3): a 2x2 block of outputs is Y = A^T [ (G g G^T) .* (B^T d B) ] A summed over the input
// channels, d the 4x4 input patch around the block.  Per element xi = (a, b) of the 4x4 transform
// domain this is ONE GEMM over the input channels, M_xi[co][col] = sum_ci U_xi[co][ci] V_xi[ci][col],
// col = (leaf, block): 16 GEMMs of K = 64 instead of one of K = 576, 2.25 times fewer multiplies.
//   U = G g G^T comes packed from the host (split halves, MFMA A-operand order, k-step = xi*4 + cg).
//   V_xi = (B^T d B)[a][b] is +-d[r0][c0] +-d[r0][c1] +-d[r1][c0] +-d[r1][c1] (every row of B^T has
//   two entries +-1): the 256 threads build it from the A2 images into a 32-column image of the
//   same row format (hi / lo), double buffered, while the MFMAs of the previous xi run.
//   Y[i][j] += +-M_xi (A^T has entries 0, +-1) on the VALU, into four accumulator tiles per wave.
// For the geometries with two leaves per workgroup (boards up to 8x8): 2 * ceil(N/2)^2 <= 32 columns.
// OFF by default (-DAZG_WINO3=1 builds it in).  Measured at 8x8 / 400 playouts / 1024 games: the same
// outputs to 1e-8; 192 instead of 432 MFMAs per wave and layer and the XCDs clock 13 % higher (2.23
// instead of 1.97 GHz at 1 260 W), but the step takes 16 % more cycles -- 11.75 against 11.43 ms per
// launch: the two transforms cost 1 800 VALU instructions per wave, the LDS moves the same 1.15 MB
// per layer in half the time it had (V written and read back on top of the patch reads), and sixteen
// barriers tie the four waves together.  Without the input transform's arithmetic (timing only) it
// runs in 10.4 ms, which bounds what a cheaper transform could reach.
#ifndef AZG_WINO3
#define AZG_WINO3 0
#endif
#ifndef AZG_WINO_RING
#define AZG_WINO_RING 3       // k-steps of U in flight per wave
#endif
#ifndef AZG_WINO_PRIO
#define AZG_WINO_PRIO 2
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1 (every index a constant expression)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// rows of B^T (input transform): the two patch indices with a non-zero entry and their signs
__device__ __forceinline__ constexpr int wino_bt_idx(int a, int k) { return a == 0 ? (k ? 2 : 0) : a == 3 ? (k ? 3 : 1) : (k ? 2 : 1); }
__device__ __forceinline__ constexpr bool wino_bt_neg(int a, int k) { return a == 0 ? k == 1 : a == 1 ? false : a == 2 ? k == 0 : k == 1; }
// column a of A^T (output transform): sign of its entry in row i, 0 if none
__device__ __forceinline__ constexpr int wino_at(int i, int a) { return i == 0 ? (a < 3 ? 1 : 0) : (a == 0 ? 0 : a == 1 ? 1 : -1); }

// wp: U image of this layer; xh / xl: the A2 images; vbuf: 2 * 2 * 32 * RS_A2 bytes of LDS that are
// free until the epilogue (the epilogue may overwrite them: the loop ends with a barrier).
// epi(0, cot, -1, acc, p, ctx) is called four times (the 2x2 output positions of every block), p the
// position of this lane's column or -1.
template <class G, int RING, class Prep, class Epi>
__device__ inline void conv3_wino_f16(const half8 *wp, const char *xh, const char *xl, char *vbuf, int wave, int lane,
                                      int tid, Prep prep, Epi epi, Prof *prof)
{
    constexpr int N = G::N, TW = (N + 1) / 2, TPL = TW * TW, NCOL = G::T * TPL;
    static_assert(NCOL <= 32, "one 32-column tile holds every 2x2 block of the workgroup's leaves");
    constexpr int RS = G::RS_A2, VB = NCOL * RS, KS = 64;     // columns past NCOL are never written; what lanes read there is discarded
    static_assert(4 * VB <= G::A2H_OFF, "the V buffers must fit in front of the A2 images");
    const int half = lane >> 5, j = lane & 31;
    // ---- builder role of this thread: column bc, channels 8*cblk .. 8*cblk+7
    const int bc = tid & 31, cblk = tid >> 5;
    int poff[16];                                   // byte offset of patch element (r, c) of the column's block, or the zero row
    {
        const bool live = bc < NCOL;
        const int leaf = bc / TPL, t = bc - leaf * TPL, ty = t / TW, tx = t - ty * TW;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int y = 2 * ty - 1 + r, x = 2 * tx - 1 + c;
                const bool ok = live && y >= 0 && y < N && x >= 0 && x < N;
                poff[4 * r + c] = (ok ? leaf * G::S + y * N + x : G::NP) * RS + cblk * 16;
            }
    }
    const int voff = bc * RS + cblk * 16;           // where this thread's 8 channels of V go
    // +-1 as a register the compiler cannot see through: fma(half, one, acc) then selects
    // v_fma_mix_f32 (the half read straight from its packed pair, widened exactly), where a literal
    // +-1 would be folded into a conversion plus an addition -- two instructions per term
    float one = 1.0f;
    asm volatile("" : "+v"(one));

    // ---- GEMM role: this wave owns channel tile `wave`, all 32 columns
    const gptr<half8> wrow = as_global(wp + (size_t)wave * KS * 128 + lane * kWLane);
    half8 ah[RING], al[RING];
#pragma unroll
    for (int d = 0; d < RING; d++) { ah[d] = wrow[(size_t)d * 128]; al[d] = wrow[(size_t)d * 128 + kWLo]; }
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 y[2][2] = {{zacc, zacc}, {zacc, zacc}};
    const int boff = j * RS + half * 16;

    // One element xi of the transform domain is twelve MFMAs (four k-steps, three products) in a
    // dependent chain; everything else of the step is cut into twelve slices, one behind each MFMA
    // (scheduling fences between the slices: left to itself the compiler issues the whole input
    // transform first and the MFMAs last, and the matrix pipe idles through the transform):
    //   * the input transform of element xi + 1: per patch element q its high halves (8 mixed
    //     FMAs), then its low halves; the split and the two stores; LDS reads two slices ahead;
    //   * the output transform of element xi - 1 (its M is complete; 16 additions per affected
    //     Y tile): the first slices and the last;
    //   * the activation reads of the next k-step and the weight loads RING k-steps ahead.
    f32x16 mprev = zacc;
    half8 dh[4], dl[4], bh[4], bl[4];
    float acc[8];
    auto d_off = [&](auto xi_c, int q) {
        constexpr int xi = decltype(xi_c)::value;
        return poff[4 * wino_bt_idx(xi >> 2, q >> 1) + wino_bt_idx(xi & 3, q & 1)];
    };
    auto yadd = [&](auto xi_c, int part) {           // output transform of element xi from mprev; part 0: tiles (0,*), 1: (1,*)
        constexpr int xi = decltype(xi_c)::value;
        constexpr int a = xi >> 2, b = xi & 3;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const int sg = wino_at(part, a) * wino_at(jj, b);
            if (sg > 0) {
#pragma unroll
                for (int e = 0; e < 16; e++) y[part][jj][e] += mprev[e];
            } else if (sg < 0) {
#pragma unroll
                for (int e = 0; e < 16; e++) y[part][jj][e] -= mprev[e];
            }
        }
    };
    auto mix = [&](auto xi_c, int q, bool low) {      // one slice of the input transform of element xi
        constexpr int xi = decltype(xi_c)::value;
        const bool neg = wino_bt_neg(xi >> 2, q >> 1) != wino_bt_neg(xi & 3, q & 1);
        const float sg = neg ? -one : one;
        if (!low) {
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] = __builtin_fmaf((float)dh[q][e], sg, q == 0 ? 0.0f : acc[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] = __builtin_fmaf((float)dl[q][e], sg, acc[e]);
        }
    };
    auto vstore = [&](char *vb) {
        u32x4 vh, vl;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
            unsigned hh, ll;
            split2_f16(acc[2 * e2], acc[2 * e2 + 1], hh, ll);
            vh[e2] = hh;
            vl[e2] = ll;
        }
        if (NCOL == 32 || bc < NCOL) {
            *reinterpret_cast<u32x4 *>(vb + voff) = vh;
            *reinterpret_cast<u32x4 *>(vb + VB + voff) = vl;
        }
    };
    // element 0's transform, nothing to overlap it with
    {
        const std::integral_constant<int, 0> x0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            dh[q] = *reinterpret_cast<const half8 *>(xh + d_off(x0, q));
            dl[q] = *reinterpret_cast<const half8 *>(xl + d_off(x0, q));
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { mix(x0, q, false); mix(x0, q, true); }
        vstore(vbuf);
    }
    azg_block_barrier();
    __builtin_amdgcn_s_setprio(AZG_WINO_PRIO);
    static_for<0, 16>([&](auto xi_c) {
        constexpr int xi = decltype(xi_c)::value;
        constexpr bool more = xi + 1 < 16;
        const std::integral_constant<int, more ? xi + 1 : xi> xn;       // the element whose transform is built here
        const std::integral_constant<int, xi ? xi - 1 : 0> xp;          // the element whose M is folded into Y here
        const char *vc = vbuf + (xi & 1) * 2 * VB;
        char *vn = vbuf + ((xi + 1) & 1) * 2 * VB;
        f32x16 m = zacc;
        auto bload = [&](int cg) {
            bh[cg] = *reinterpret_cast<const half8 *>(vc + boff + cg * 32);
            bl[cg] = *reinterpret_cast<const half8 *>(vc + VB + boff + cg * 32);
        };
        auto dload = [&](int q) {
            if (more) {
                dh[q] = *reinterpret_cast<const half8 *>(xh + d_off(xn, q));
                dl[q] = *reinterpret_cast<const half8 *>(xl + d_off(xn, q));
            }
        };
        bload(0);
        dload(0);
        bload(1);
#pragma unroll
        for (int s = 0; s < 12; s++) {
            const int cg = s / 3, pr = s % 3, ks = xi * 4 + cg, ca = ks % RING;
            if (pr == 0) m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bh[cg], m, 0, 0, 0);
            if (pr == 1) m = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ca], bh[cg], m, 0, 0, 0);
            if (pr == 2) {
                m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bl[cg], m, 0, 0, 0);
                if (ks + RING < KS) { ah[ca] = wrow[(size_t)(ks + RING) * 128]; al[ca] = wrow[(size_t)(ks + RING) * 128 + kWLo]; }
            }
            // the slice behind MFMA s
            if (s == 0) { if (xi) yadd(xp, 0); dload(1); }
            if (s == 1) { if (xi) yadd(xp, 1); bload(2); }
            if (s >= 2 && s < 10 && more) mix(xn, (s - 2) >> 1, ((s - 2) & 1) != 0);
            if (s == 3) dload(2);
            if (s == 4) bload(3);
            if (s == 5) dload(3);
            if (s == 10 && more) vstore(vn);
            __builtin_amdgcn_sched_barrier(0);
        }
        mprev = m;
        azg_block_barrier();                 // V[xi + 1] is complete, V[xi] may be overwritten (after the last xi: by the epilogue)
    });
    yadd(std::integral_constant<int, 15>{}, 0);
    yadd(std::integral_constant<int, 15>{}, 1);
    __builtin_amdgcn_s_setprio(0);
    if (prof) prof->mark(11);
    const auto ctx = prep();
    {
        const bool live = j < NCOL;
        const int leaf = j / TPL, t = j - leaf * TPL, ty = t / TW, tx = t - ty * TW;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int yy = 2 * ty + i, xx = 2 * tx + jj;
                const int p = (live && yy < N && xx < N) ? leaf * G::S + yy * N + xx : -1;
                epi(0, wave, -1, y[i][jj], p, ctx);
            }
    }
}

// Wave-wide float reductions on the DPP network (lane^1, lane^2, mirror within 8, mirror within
// 16, then one lane of each row of 16): register-to-register, where a __shfl_xor butterfly costs
// six LDS-crossbar round trips.  All 64 lanes must be active; the result is wave-uniform.
template <int CTRL>
__device__ inline float dpp_f32(float v)
{
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(b, b, CTRL, 0xF, 0xF, false));
}
__device__ inline float lane_f32(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ inline float wave_max_f32(float v)
{
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}
__device__ inline float wave_sum_f32(float v)
{
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}

// conv1 (4 input planes, exact 0/1 values): the 36 products per output are packed as
// k = tap*4 + plane into three k-steps of 16 (a lane's eight k-values are two taps of four planes,
// two 8-byte LDS reads) instead of nine k-steps that carry 4 real channels in 16.  No w_hi*x_lo
// term.  wp: per k-step and lane 8 hi then 8 lo halves of W[co = lane&31][plane = j%4][tap =
// 4*ks + 2*(lane>>5) + j/4] * 2^8 (zero for tap >= 9).  ring: all three k-steps, preloaded.
struct W1Ring {
    half8 ah[3], al[3];
};
__device__ inline void conv1_preload(const half8 *wp, int lane, W1Ring &r)
{
    const gptr<half8> wrow = as_global(wp + (size_t)lane * kWLane);
#pragma unroll
    for (int d = 0; d < 3; d++) { r.ah[d] = wrow[(size_t)d * 128]; r.al[d] = wrow[(size_t)d * 128 + kWLo]; }
    __builtin_amdgcn_sched_barrier(0);
}

template <class G, class Prep, class Epi>
__device__ inline void conv1_packed_f16(W1Ring &ring, const char *xin, int wave, int lane, Prep prep, Epi epi, Prof *prof, int slot)
{
    constexpr int RS = G::RS_IN;
    constexpr int PT = (G::NT + 3) / 4;
    const int half = lane >> 5, j = lane & 31;
    const int zero = G::NP * RS;
    int base[PT], prow[PT];
    unsigned vmask[PT];
#pragma unroll
    for (int i = 0; i < PT; i++) {
        const int tile = wave * PT + i;
        const int p = tile * 32 + j;
        base[i] = zero; vmask[i] = 0; prow[i] = -1;
        if (tile < G::NT && p < G::NP) {
            const int leaf = p / G::S, cell = p - leaf * G::S;
            const int row = cell / G::N, col = cell - row * G::N;
            base[i] = p * RS;
            prow[i] = p;
            unsigned m = 0x1FFu;
            if (row == 0) m &= ~0x007u;
            if (row == G::N - 1) m &= ~0x1C0u;
            if (col == 0) m &= ~0x049u;
            if (col == G::N - 1) m &= ~0x124u;
            vmask[i] = m;
        }
    }
    if (prof) prof->mark(slot - 1);
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc[PT];
    half8 b[3][PT];
#pragma unroll
    for (int ks = 0; ks < 3; ks++)
#pragma unroll
        for (int i = 0; i < PT; i++) {
            half4 t2[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int tap = 4 * ks + 2 * half + e;                  // per-lane (two values: half = 0 / 1)
                const int dr = tap / 3 - 1, dc = tap - (tap / 3) * 3 - 1;
                const bool on = tap < 9 && ((vmask[i] >> tap) & 1u);
                const int a = on ? base[i] + (dr * G::N + dc) * RS : zero;
                t2[e] = *reinterpret_cast<const half4 *>(xin + a);
            }
            b[ks][i] = __builtin_shufflevector(t2[0], t2[1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
    for (int ks = 0; ks < 3; ks++)
#pragma unroll
        for (int i = 0; i < PT; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring.ah[ks], b[ks][i], ks == 0 ? zacc : acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring.al[ks], b[ks][i], acc[i], 0, 0, 0);
        }
    if (prof) prof->mark(slot);
    const auto ctx = prep();
#pragma unroll
    for (int i = 0; i < PT; i++)
        if (wave * PT + i < G::NT) epi(i, wave * PT + i, acc[i], prow[i], ctx);
}

#ifndef AZG_FC_RING
#define AZG_FC_RING 28        // weight units (k-step, tile) of a fully connected layer in flight per wave
#endif
// One group of 16-output tiles of a fully connected layer for one wave: NTG tiles, the wave's
// KW k-steps of each.  Units (k-step, tile) stream through a ring of RU register pairs.
template <int NTG, int KW>
struct FcRing {
    static constexpr int NU = NTG * KW;
    static constexpr int RU = NU < AZG_FC_RING ? NU : AZG_FC_RING;
    half8 ah[RU], al[RU];
};

// w: this lane's entry of tile 0 / k-step 0 of the group; tiles are ks_total k-steps apart.
template <int NTG, int KW>
__device__ inline gptr<half8> fc16_src(gptr<half8> w, int ks_total, int u)
{
    const int kk = u / NTG, i = u - kk * NTG;
    return w + ((size_t)i * ks_total + kk) * 128;
}

template <int NTG, int KW>
__device__ inline void fc16_preload(gptr<half8> w, int ks_total, FcRing<NTG, KW> &r)
{
#pragma unroll
    for (int u = 0; u < FcRing<NTG, KW>::RU; u++) {
        const gptr<half8> s = fc16_src<NTG, KW>(w, ks_total, u);
        r.ah[u] = s[0]; r.al[u] = s[kWLo];
    }
    __builtin_amdgcn_sched_barrier(0);
}

// x is the A operand (row = leaf) and W the B operand (column = output): lane l supplies the
// eight inputs 8*(l/16) .. +7 of leaf row / output column l%16, and lanes 0..15 end up with one
// output's value for leaf t in accumulator register t.
// xi: this lane's 16 bytes of the hi image at the wave's first k-step; lo image xlo_off further.
template <int T, int NTG, int KW>
__device__ inline void fc16_run(gptr<half8> w, int ks_total, FcRing<NTG, KW> &r, const char *xi, int xlo_off,
                                bool leaf_lane, float *out /* + (first output of the group) * T */, int kgrp, int n)
{
    using R = FcRing<NTG, KW>;
    static_assert(T <= 4, "leaf t is read from accumulator register t of the first lane group");
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const f32x4 zacc = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[NTG];
    half8 xh = z8, xl = z8;
#pragma unroll
    for (int u = 0; u < R::NU; u++) {
        const int kk = u / NTG, i = u - kk * NTG;
        if (i == 0) {
            xh = *reinterpret_cast<const half8 *>(xi + kk * 64);
            xl = *reinterpret_cast<const half8 *>(xi + kk * 64 + xlo_off);
            if (!leaf_lane) { xh = z8; xl = z8; }
        }
        const half8 wa = r.ah[u % R::RU], wl = r.al[u % R::RU];
        if (u + R::RU < R::NU) {
            const gptr<half8> s = fc16_src<NTG, KW>(w, ks_total, u + R::RU);
            r.ah[u % R::RU] = s[0]; r.al[u % R::RU] = s[kWLo];
        }
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wa, kk == 0 ? zacc : acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wa, acc[i], 0, 0, 0);
    }
    if (kgrp == 0) {
#pragma unroll
        for (int i = 0; i < NTG; i++)
#pragma unroll
            for (int t = 0; t < T; t++) out[(i * 16 + n) * T + t] = acc[i][t];
    }
}

#ifndef AZG_SPARSE_FC_ROWS
#define AZG_SPARSE_FC_ROWS 32
#endif
// The tail of a one-leaf forward pass (T == 1, the 15x15 kernels): head outputs, then both wide
// fully connected layers as sparse matrix-vector products on the VALU.  With one leaf the FC
// layers are bound by streaming their weights from L2 (0.8 MB for 15x15), and about half of the
// ReLU'd head outputs are exactly zero: only the weight rows of non-zero inputs are read.
// Wave w compacts and multiplies the inputs k = 4*i + w; a lane owns four consecutive outputs
// (16-byte loads, 32 rows of 1 KB in flight per wave).  Sums are float32 in a fixed order.
template <class G>
__device__ inline void cnn16_tail_sparse(const NetWeights &nw, float *lds, int tid, Prof &prof)
{
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float *lb = lds + G::BIAS_OFF / 4;
    struct Pair { int k; float x; };
    Pair *lp = reinterpret_cast<Pair *>(lds + G::LP_OFF) + wave * G::QP;
    Pair *lv = reinterpret_cast<Pair *>(lds + G::LV_OFF) + wave * G::QV;
    int *cnt = reinterpret_cast<int *>(lds + G::CNT_OFF);

    // ---- head outputs (bias + the four channel-tile partials, ReLU) and their non-zero lists
    auto compact = [&](int kmax, int c0, Pair *list) {
        const float *bh = lb + G::B_HEAD;
        int n = 0;
        for (int i0 = 0; 4 * i0 + wave < kmax; i0 += WAVE) {
            const int k = 4 * (i0 + lane) + wave;
            float v = 0.0f;
            if (k < kmax) {
                const int cc = k / G::S, cell = k - cc * G::S, c = c0 + cc;
                const float *pp = lds + G::HP_OFF + c * G::NTP + cell;
                v = __builtin_fmaf(head_sum<ConvWaves<G>::HPARTS>(pp, G::NTP), kWUnscale, bh[c]);
            }
            const bool nz = !(v <= 0.0f);                    // a NaN counts as non-zero (see relu_nan)
            const unsigned long long m = __ballot(nz);
            if (nz) {
                const int at = n + __popcll(m & ((1ULL << lane) - 1ULL));
                list[at].k = k;
                list[at].x = v;
            }
            n += __popcll(m);
        }
        return n;
    };
    const int np = compact(4 * G::S, 0, lp);
    const int nv = compact(2 * G::S, 4, lv);
    if (lane == 0) { cnt[wave] = np; cnt[4 + wave] = nv; }
    azg_block_barrier();
    prof.mark(7);

    // ---- act_fc1: this wave's inputs against all outputs, outputs 4*lane .. 4*lane+3 per lane
    {
        const gptr<f32x4> wrow = as_global(reinterpret_cast<const f32x4 *>(nw.wpfc_k)) + lane;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        constexpr int U = AZG_SPARSE_FC_ROWS;                                // weight rows (1 KB each) in flight per wave
        for (int i0 = 0; i0 < np; i0 += U) {
            f32x4 wv[U];
            float xv[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const Pair e = lp[i0 + u < np ? i0 + u : np - 1];            // clamp: keeps the load unconditional
                xv[u] = i0 + u < np ? e.x : 0.0f;
                wv[u] = wrow[(size_t)e.k * 64];
            }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc[j] = __builtin_fmaf(wv[u][j], xv[u], acc[j]);
                    asm volatile("" : "+v"(acc[j]));          // never two-wide (see the note at AZG_EXP_PKFC, azg_dual.hpp)
                }
        }
        *reinterpret_cast<f32x4 *>(lds + G::SPP_OFF + wave * 256 + 4 * lane) = acc;
    }
    // ---- val_fc1: lane group g = lane / 16 takes every fourth entry of the wave's list, a lane four outputs
    {
        const int grp = lane >> 4, l16 = lane & 15;
        const gptr<f32x4> wrow = as_global(reinterpret_cast<const f32x4 *>(nw.wvfc1_k)) + l16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int i0 = 0; i0 < nv; i0 += 4 * 8) {
            f32x4 wv[8];
            float xv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 4 * u + grp;
                const Pair e = lv[i < nv ? i : nv - 1];
                xv[u] = i < nv ? e.x : 0.0f;
                wv[u] = wrow[(size_t)e.k * 16];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc[j] = __builtin_fmaf(wv[u][j], xv[u], acc[j]);
                    asm volatile("" : "+v"(acc[j]));
                }
        }
        *reinterpret_cast<f32x4 *>(lds + G::SPV_OFF + (wave * 4 + grp) * 64 + 4 * l16) = acc;
    }
    azg_block_barrier();
    prof.mark(8);

    // ---- softmax over the S logits and the scalar value (wave 0; every partial is read before
    //      the first result is written, the result slots overlap no partial)
    if (wave == 0) {
        float x[4];
        float mx = -3.0e38f;
        {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const f32x4 p = *reinterpret_cast<const f32x4 *>(lds + G::SPP_OFF + w * 256 + 4 * lane);
#pragma unroll
                for (int j = 0; j < 4; j++) s[j] += p[j];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int a = 4 * lane + j;
                x[j] = a < G::S ? s[j] + lb[G::B_PFC + a] : -3.0e38f;
                mx = fmaxf(mx, x[j]);
            }
        }
        float hdn = lb[G::B_VFC1 + lane];
#pragma unroll
        for (int w = 0; w < 16; w++) hdn += lds[G::SPV_OFF + w * 64 + lane];
        hdn = relu_nan(hdn);
        mx = wave_max_f32(mx);
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (4 * lane + j < G::S) sum += expf(x[j] - mx);
        sum = wave_sum_f32(sum);
        const float lse = logf(sum);
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (4 * lane + j < G::S) lds[G::LG_OFF + 4 * lane + j] = expf((x[j] - mx) - lse);
        const float v = wave_sum_f32(hdn * lb[G::B_WVFC2 + lane]);
        if (lane == 0) lds[G::VAL_OFF] = tanhf(v + lb[G::B_VFC2]);
    }
    azg_block_barrier();
    prof.mark(9);
}

// Everything after conv3 for the split-f16 path: head outputs (bias + the four channel-tile
// partials, ReLU) written as half pairs, both fully connected layers on the matrix cores,
// softmax and tanh.  A wave takes k-steps [wave*KW, (wave+1)*KW) of every 16-output tile; the
// four partial sums per output meet in LDS and are added in a fixed order.  The first weights
// of both layers are requested before the head outputs are formed, so the L2 latency of the
// weight stream overlaps that stage.
template <class G>
__device__ inline void cnn16_tail(const NetWeights &nw, float *lds, int tid, Prof &prof)
{
    char *ldb = reinterpret_cast<char *>(lds);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int kgrp = lane >> 4, n = lane & 15;             // FC operand geometry: k group and row / column of the lane
    constexpr int NGP = G::OTP / G::OC;                    // accumulator groups of act_fc1
    const gptr<half8> wpol = as_global(reinterpret_cast<const half8 *>(nw.wfc16[0])) + (size_t)wave * G::KWP * 128 + lane * kWLane;
    const gptr<half8> wval = as_global(reinterpret_cast<const half8 *>(nw.wfc16[1])) + (size_t)wave * G::KWV * 128 + lane * kWLane;
    FcRing<G::OC, G::KWP> rpol;
    FcRing<4, G::KWV> rval;
    fc16_preload<G::OC, G::KWP>(wpol, 4 * G::KWP, rpol);
    fc16_preload<4, G::KWV>(wval, 4 * G::KWV, rval);

    // ---- head outputs, flattened [c][row][col] per head like the reference's view(-1, ...)
    {
        const float *bh = lds + G::BIAS_OFF / 4 + G::B_HEAD;
        _Float16 *xb = reinterpret_cast<_Float16 *>(lds + G::X_OFF);
        if constexpr (G::S % 4 == 0) {
            // four consecutive features (same head channel, consecutive cells) per thread
            for (int e4 = tid; e4 < G::T * (G::KPP + G::KPV) / 4; e4 += 256) {
                const int e = 4 * e4;
                const bool pol = e < G::T * G::KPP;
                const int e2 = pol ? e : e - G::T * G::KPP;
                const int kp = pol ? G::KPP : G::KPV;
                const int leaf = e2 / kp, k = e2 - leaf * kp;
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                if (k < (pol ? 4 : 2) * G::S) {
                    const int cc = k / G::S, cell = k - cc * G::S;
                    const int c = pol ? cc : 4 + cc, p = leaf * G::S + cell;
                    const float *pp = lds + G::HP_OFF + c * G::NTP + p;
                    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(pp), p1 = *reinterpret_cast<const f32x4 *>(pp + 8 * G::NTP);
                    f32x4 ps = {p0[0] + p1[0], p0[1] + p1[1], p0[2] + p1[2], p0[3] + p1[3]};
                    if (ConvWaves<G>::HPARTS == 4) {
                        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(pp + 16 * G::NTP), p3 = *reinterpret_cast<const f32x4 *>(pp + 24 * G::NTP);
#pragma unroll
                        for (int i = 0; i < 4; i++) ps[i] = ps[i] + (p2[i] + p3[i]);
                    }
                    const float b = bh[c];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float v = __builtin_fmaf(ps[i], kWUnscale, b);
                        s[i] = relu_nan(v);
                    }
                }
                half4 h, l;
#pragma unroll
                for (int i = 0; i < 4; i++) { h[i] = (_Float16)s[i]; l[i] = (_Float16)(s[i] - (float)h[i]); }
                _Float16 *dst = xb + (pol ? 0 : 2 * G::T * G::KPP) + leaf * kp + k;
                *reinterpret_cast<half4 *>(dst) = h;
                *reinterpret_cast<half4 *>(dst + G::T * kp) = l;
            }
        } else {
        for (int e = tid; e < G::T * (G::KPP + G::KPV); e += 256) {
            const bool pol = e < G::T * G::KPP;
            const int e2 = pol ? e : e - G::T * G::KPP;
            const int kp = pol ? G::KPP : G::KPV;
            const int leaf = e2 / kp, k = e2 - leaf * kp;
            float s = 0.0f;
            if (k < (pol ? 4 : 2) * G::S) {
                const int cc = k / G::S, cell = k - cc * G::S;
                const int c = pol ? cc : 4 + cc, p = leaf * G::S + cell;
                const float *pp = lds + G::HP_OFF + c * G::NTP + p;
                s = __builtin_fmaf(head_sum<ConvWaves<G>::HPARTS>(pp, G::NTP), kWUnscale, bh[c]);
                s = relu_nan(s);
            }
            const _Float16 hi = (_Float16)s;
            _Float16 *dst = xb + (pol ? 0 : 2 * G::T * G::KPP) + leaf * kp + k;
            dst[0] = hi;
            dst[G::T * kp] = (_Float16)(s - (float)hi);
        }
        }
    }
    azg_block_barrier();
    prof.mark(7);

    // ---- act_fc1 and val_fc1
    {
        const int leafc = n < G::T ? n : 0;
        const char *xp = ldb + G::X_OFF * 4 + (leafc * G::KPP + wave * G::KWP * 32 + 8 * kgrp) * 2;
        const char *xv = ldb + G::X_OFF * 4 + 2 * G::T * G::KPP * 2 + (leafc * G::KPV + wave * G::KWV * 32 + 8 * kgrp) * 2;
        float *part = lds + G::PART_OFF + wave * G::NOUT * G::T;
        fc16_run<G::T, G::OC, G::KWP>(wpol, 4 * G::KWP, rpol, xp, G::T * G::KPP * 2, n < G::T, part, kgrp, n);
        for (int grp = 1; grp < NGP; grp++) {
            const gptr<half8> wg = wpol + (size_t)grp * G::OC * (4 * G::KWP) * 128;
            fc16_preload<G::OC, G::KWP>(wg, 4 * G::KWP, rpol);
            fc16_run<G::T, G::OC, G::KWP>(wg, 4 * G::KWP, rpol, xp, G::T * G::KPP * 2, n < G::T,
                                          part + grp * G::OC * 16 * G::T, kgrp, n);
        }
        fc16_run<G::T, 4, G::KWV>(wval, 4 * G::KWV, rval, xv, G::T * G::KPV * 2, n < G::T,
                                  part + G::OTP * 16 * G::T, kgrp, n);
    }
    azg_block_barrier();
    prof.mark(8);

    // ---- softmax over the S logits and the scalar value, one wave per position
    if (wave < G::T) {
        const int leaf = wave;
        const float *lb = lds + G::BIAS_OFF / 4;
        auto fc_out = [&](int o, float bias) {
            const float *pp = lds + G::PART_OFF + o * G::T + leaf;
            const float s = (pp[0] + pp[G::NOUT * G::T]) + (pp[2 * G::NOUT * G::T] + pp[3 * G::NOUT * G::T]);
            return __builtin_fmaf(s, kWUnscale, bias);
        };
        constexpr int PER = (G::S + 63) / 64;
        float x[PER];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int a = lane + 64 * i;
            x[i] = a < G::S ? fc_out(a, lb[G::B_PFC + a]) : -3.0e38f;
            mx = fmaxf(mx, x[i]);
        }
        mx = wave_max_f32(mx);
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int a = lane + 64 * i;
            if (a < G::S) sum += expf(x[i] - mx);
        }
        sum = wave_sum_f32(sum);
        const float lse = logf(sum);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int a = lane + 64 * i;
            if (a < G::S) lds[G::LG_OFF + leaf * G::S + a] = expf((x[i] - mx) - lse);
        }
        float hdn = fc_out(G::OTP * 16 + lane, lb[G::B_VFC1 + lane]);
        hdn = relu_nan(hdn);
        float v = hdn * lb[G::B_WVFC2 + lane];
        v = wave_sum_f32(v);
        if (lane == 0) lds[G::VAL_OFF + leaf] = tanhf(v + lb[G::B_VFC2]);
    }
    azg_block_barrier();
    prof.mark(9);
}

// Full forward; results at lds[LG_OFF..] / lds[VAL_OFF..] (float offsets) as with cnn_forward
// (azg_cnn.hpp), but `input(leaf, row, col, out[4])` supplies the four planes of a cell at once.  Needs cnn16_prepare once per workgroup.
template <class G, class InputFn>
__device__ inline void cnn16_forward(const NetWeights &nw_in, float *lds, InputFn input, Prof &prof,
                                     const MatrixGate &gate)
{
    const NetWeights nw = opaque_weights(nw_in);
    char *ldb = reinterpret_cast<char *>(lds);
    // The thread index is re-read opaquely on every call: per-lane addresses and masks derived
    // from it would otherwise be hoisted out of the caller's playout loop and held in registers.
    int tid = azg_tid();
    asm volatile("" : "+v"(tid));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    __builtin_assume(wave >= 0 && wave < 4);               // 256-thread workgroups: lets tile-validity tests fold
    const int half = lane >> 5, j = lane & 31;
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    const float *lbias = lds + G::BIAS_OFF / 4;            // conv1 | conv2 | conv3 | head biases
#ifndef AZG_R2
#define AZG_R2 6
#endif
#ifndef AZG_R3
#define AZG_R3 3
#endif
    constexpr int CW2 = ConvWaves<G>::CW, CW3 = ConvWaves<G>::CW3;
    constexpr int R2 = CW2 == 2 ? 4 : AZG_R2, R3 = CW3 == 4 ? 1 : AZG_R3;                  // k-steps of weights in flight per layer (conv1: all three)
    constexpr bool kWino3 = AZG_WINO3 && G::T == 2 && CW3 == 1;      // conv3 in Winograd form (two-leaf geometries)
    const half8 *w1 = reinterpret_cast<const half8 *>(nw.w16[0]);
    const half8 *w2 = reinterpret_cast<const half8 *>(nw.w16[1]);
    const half8 *w3 = reinterpret_cast<const half8 *>(nw.w16[2]);
    W1Ring ring1;
    WRing<R2, CW2> ring2;
    WRing<R3, CW3> ring3;
    conv1_preload(w1, lane, ring1);

    // ---- input image: the 4 planes as halves in channels 0..3 of a 16-channel row; the
    //      zero rows of the IN and A1 images are rewritten (the f32 scratch overlaps them)
    for (int p = tid; p < G::NP; p += 256) {
        const int leaf = p / G::S, cell = p - leaf * G::S;
        const int row = cell / G::N, col = cell - row * G::N;
        float pl[4];
        input(leaf, row, col, pl);
        half8 v0 = z8;
#pragma unroll
        for (int ch = 0; ch < 4; ch++) v0[ch] = (_Float16)pl[ch];
        *reinterpret_cast<half8 *>(ldb + G::IN_OFF + p * G::RS_IN) = v0;
        *reinterpret_cast<half8 *>(ldb + G::IN_OFF + p * G::RS_IN + 16) = z8;
    }
    if (tid < G::RS_IN / 16) *reinterpret_cast<half8 *>(ldb + G::IN_OFF + G::NP * G::RS_IN + tid * 16) = z8;
    if (tid >= 64 && tid < 64 + G::RS_A1 / 16)
        *reinterpret_cast<half8 *>(ldb + G::A1H_OFF + G::NP * G::RS_A1 + (tid - 64) * 16) = z8;
    if (tid >= 128 && tid < 128 + G::RS_A1 / 16)
        *reinterpret_cast<half8 *>(ldb + G::A1L_OFF + G::NP * G::RS_A1 + (tid - 128) * 16) = z8;
    azg_block_barrier();
    prof.mark(2);

    // epilogue of conv1/conv2: scale back, bias, ReLU, split into halves, store [pos][channel]
    struct Bias16 { f32x4 b[CW2][4]; };
    // cot0: the wave's first channel tile; nc: channel tiles the wave owns
    auto load_bias = [&](const float *bias, int cot0, int nc) {
        return [=]() {
            Bias16 c;
#pragma unroll
            for (int t = 0; t < CW2; t++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    c.b[t][g] = *reinterpret_cast<const f32x4 *>(bias + (cot0 + (t < nc ? t : 0)) * 32 + 8 * g + 4 * half);
            return c;
        };
    };
    auto store_split = [&](char *oh, char *ol, int rs) {
        return [=](int ci, int cot, int, const f32x16 &acc, int p, const Bias16 &c) {
            if (G::NP != G::NTP && p < 0) return;           // padding lane of the last tile
            const int cobase = cot * 32;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                u32x2 h, l;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float v0 = relu_nan(__builtin_fmaf(acc[4 * g + 2 * e], kWUnscale, c.b[ci][g][2 * e]));
                    const float v1 = relu_nan(__builtin_fmaf(acc[4 * g + 2 * e + 1], kWUnscale, c.b[ci][g][2 * e + 1]));
                    unsigned hh, ll;
                    split2_f16(v0, v1, hh, ll);
                    h[e] = hh;
                    l[e] = ll;
                }
                const int off = p * rs + (cobase + 8 * g + 4 * half) * 2;
                *reinterpret_cast<u32x2 *>(oh + off) = h;
                *reinterpret_cast<u32x2 *>(ol + off) = l;
            }
        };
    };
    conv_preload<G, 32, 64, R2, CW2>(w2, wave, lane, ring2);
    {
        auto st1 = store_split(ldb + G::A1H_OFF, ldb + G::A1L_OFF, G::RS_A1);
        conv1_packed_f16<G>(ring1, ldb + G::IN_OFF, wave, lane, load_bias(lbias, 0, 1),
                            [=](int, int tile, const f32x16 &acc, int p, const Bias16 &c) { st1(0, 0, tile, acc, p, c); }, &prof, 13);
    }
    prof.mark(16);
    if (tid == 0) gate.acquire();
    azg_block_barrier();
    prof.mark(3);
    if constexpr (!kWino3) conv_preload<G, 64, 128, R3, CW3>(w3, wave, lane, ring3);
    conv3x3_f16<G, 32, 64, G::RS_A1, true, G::BPF, R2, CW2>(w2, ring2, ldb + G::A1H_OFF, ldb + G::A1L_OFF, wave, lane,
                                               load_bias(lbias + 32, (wave % ConvSplit<G, 64, CW2>::WC) * CW2, CW2),
                                               store_split(ldb + G::A2H_OFF, ldb + G::A2L_OFF, G::RS_A2),
                                               &prof, 15);
    prof.mark(17);
    azg_block_barrier();
    prof.mark(4);

    // ---- conv3; its output never goes to LDS: bias + ReLU in registers, split into halves,
    //      and the accumulator registers are the B operand of the MFMA that applies the two
    //      1x1 head convolutions (k runs over this wave's 32 channels in accumulator order,
    //      the head weights are packed to match).  The four waves' partial sums meet in LDS.
    {
        using SP3 = ConvSplit<G, 128, CW3>;
        const int cg3 = wave % SP3::WC;                    // channel group of this wave: channel tiles cg3*CW3 ..
        // The head tile has 32 rows and 6 head channels: rows 0-7 carry the high halves of the
        // head weights and rows 8-15 the low halves, so ONE instruction per activation half
        // forms both w_hi*x and w_lo*x; adding the two row blocks afterwards gives the full
        // (w_hi + w_lo)(x_hi + x_lo) with two MFMAs per k-step instead of three.
        struct Head16 { f32x4 b[CW3][4]; half8 w[CW3][2]; };
        float *hp = lds + G::HP_OFF + cg3 * 8 * G::NTP;
        f32x16 hd;                                         // head sums of the position tile in work, over the wave's channel tiles
        auto head_prep = [=]() {
                Head16 c;
#pragma unroll
                for (int t = 0; t < CW3; t++) {
                    const int cot = cg3 * CW3 + t;
                    const float *b3 = lbias + 96 + cot * 32;
                    const char *whp = ldb + G::WH_OFF + ((cot * 4 + half) * 8 + (j < 16 ? (j & 7) : 7)) * 32 + ((j >= 8 && j < 16) ? 16 : 0);
#pragma unroll
                    for (int g = 0; g < 4; g++) c.b[t][g] = *reinterpret_cast<const f32x4 *>(b3 + 8 * g + 4 * half);
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) c.w[t][s2] = *reinterpret_cast<const half8 *>(whp + s2 * 512);
                }
                return c;
            };
        auto head_epi = [=, &hd](int ci, int, int tile, const f32x16 &acc, int p, const Head16 &c) {
                const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (ci == 0) hd = zacc;
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    u32x4 xh4, xl4;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 8 * s2 + 2 * e;
                        const float v0 = relu_nan(__builtin_fmaf(acc[r], kWUnscale, c.b[ci][r >> 2][r & 3]));
                        const float v1 = relu_nan(__builtin_fmaf(acc[r + 1], kWUnscale, c.b[ci][(r + 1) >> 2][(r + 1) & 3]));
                        unsigned hh, ll;
                        split2_f16(v0, v1, hh, ll);
                        xh4[e] = hh;
                        xl4[e] = ll;
                    }
                    const half8 xh8 = __builtin_bit_cast(half8, xh4), xl8 = __builtin_bit_cast(half8, xl4);
                    hd = __builtin_amdgcn_mfma_f32_32x32x16_f16(c.w[ci][s2], xh8, hd, 0, 0, 0);
                    hd = __builtin_amdgcn_mfma_f32_32x32x16_f16(c.w[ci][s2], xl8, hd, 0, 0, 0);
                }
                if (ci != CW3 - 1) return;
                // rows 0-3 of the head tile sit in registers 0-3 of lanes 0-31, rows 4-7 in
                // registers 0-3 of lanes 32-63 (rows 6, 7 are zero: their weights are); the
                // low-half rows 8-15 sit in registers 4-7 of the same lanes
                if ((G::NP != G::NTP || kWino3) && p < 0) return;       // padding lane of the last tile / no such position
                // (the direct path addresses by tile and lane, not by p, although the two are equal there:
                // with p the compiler pairs the stores of tiles t and t + 2 into ds_write2st64_b32 and the
                // first leaf's sums came out wrong on the GPU -- measured, not understood)
                float *dst = hp + (4 * half) * G::NTP + (kWino3 ? p : tile * 32 + j);
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) dst[c4 * G::NTP] = hd[c4] + hd[4 + c4];
            };
        if constexpr (kWino3)
            conv3_wino_f16<G, AZG_WINO_RING>(reinterpret_cast<const half8 *>(nw.w16w3), ldb + G::A2H_OFF, ldb + G::A2L_OFF, ldb, wave, lane, tid,
                                 head_prep, head_epi, &prof);
        else
            conv3x3_f16<G, 64, 128, G::RS_A2, true, G::BPF, R3, CW3>(w3, ring3, ldb + G::A2H_OFF, ldb + G::A2L_OFF, wave, lane,
                                                                   head_prep, head_epi, &prof);
    }
    prof.mark(5);
    azg_block_barrier();
    if (tid == 0) gate.release();
    prof.mark(6);
    if constexpr (G::SPARSE_FC) cnn16_tail_sparse<G>(nw, lds, tid, prof);
    else cnn16_tail<G>(nw, lds, tid, prof);
}

// One-time LDS set-up: the zero rows of the A2 images and the head weights.
template <class G>
__device__ inline void cnn16_prepare(const NetWeights &nw, float *lds)
{
    char *ldb = reinterpret_cast<char *>(lds);
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int tid = azg_tid();
    if (tid < G::RS_A2 / 16) *reinterpret_cast<half8 *>(ldb + G::A2H_OFF + G::NP * G::RS_A2 + tid * 16) = z8;
    if (tid >= 64 && tid < 64 + G::RS_A2 / 16)
        *reinterpret_cast<half8 *>(ldb + G::A2L_OFF + G::NP * G::RS_A2 + (tid - 64) * 16) = z8;
    reinterpret_cast<half8 *>(ldb + G::WH_OFF)[tid] = reinterpret_cast<const half8 *>(nw.whead16)[tid];
    float *lb = lds + G::BIAS_OFF / 4;
    if (tid < 32) lb[tid] = nw.bconv[0][tid];
    else if (tid < 96) lb[tid] = nw.bconv[1][tid - 32];
    else if (tid < 224) lb[tid] = nw.bconv[2][tid - 96];
    else if (tid < 232) lb[tid] = tid < 230 ? nw.bhead[tid - 224] : 0.0f;
    if (tid < 64) { lb[G::B_VFC1 + tid] = nw.bvfc1[tid]; lb[G::B_WVFC2 + tid] = nw.wvfc2[tid]; }
    if (tid == 64) lb[G::B_VFC2] = nw.bvfc2[0];
    for (int a = tid; a < G::S; a += 256) lb[G::B_PFC + a] = nw.bpfc[a];
    azg_block_barrier();
}

}  // namespace azg
