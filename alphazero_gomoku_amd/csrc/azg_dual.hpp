// Search kernel with two wave groups per workgroup ("dual"): 512 threads, one workgroup per CU.
//
// MCTS.get_move_probs' playout loop (mcts_alphaZero.py:161-163) for 2*T games.  The playouts of ONE
// game are strictly sequential (select needs the previous backup), so a workgroup that owns a single
// batch of leaves leaves the matrix pipes idle during select / expand / backup / softmax and the tree
// waves idle during the convolutions.  Here a workgroup owns two SETS of T games and two groups of
// four waves:
//   net waves  (0-3): conv2, conv3 + head convolutions -- for the set whose leaf is ready, then
//                     for the other set, and so on;
//   tree waves (4-7): fully connected layers + softmax / tanh of the evaluated set, expand, backup,
//                     the next descent, the input planes of the new leaf and conv1 on them -- while
//                     the net waves work on the other set.
// conv1's output image exists once: the net waves count the conv2 layers they have finished
// (a1gen), and the tree waves start the n-th conv1 only after n-1 of them.
// The groups never meet at the hardware barrier (it would join all eight waves): each has its own
// software barrier, and batches travel through two LDS counters per set (req: leaves handed over,
// ack: evaluations finished).  Both groups visit the sets strictly in turn, every turn of the tree
// waves ends with exactly one hand-over (a batch, or "no more batches"), so neither can wait for
// something the other is not going to do.  Every wait is bounded: when one runs out, the abort word
// is set, every wave leaves, the slots are flagged and the host reports an error.
#pragma once
#include "azg_cnn16.hpp"
#include "azg_tree.hpp"

namespace azg {

#ifdef AZG_DUAL_DUMP            // diagnostic builds: the hand-over region of every game's FIRST evaluation of a launch
__device__ float g_dual_dump[256][2048];
#endif
constexpr int kSpinLimit = 1 << 22;             // polls of one wait before the workgroup gives up (~0.3 s)

struct DualCtl {
    unsigned bar[2];         // arrivals at the software barrier of the net waves / the tree waves
    unsigned req[2];         // per set: batches of leaves handed to the net waves
    unsigned ack[2];         // per set: batches evaluated
    unsigned fin[2];         // per set: no further batch will come
    unsigned abort;          // a wait ran out
    unsigned a1gen;          // conv2 layers finished by the net waves (each frees conv1's output image)
    unsigned pad[6];
};

__device__ inline unsigned lds_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Barrier among the four waves of one group.  The LDS executes a wave's operations in order and
// serves all waves of the CU through one queue, so whatever a wave wrote to LDS before its arrival
// is there for a wave that has seen the arrival; the fences keep the compiler from moving LDS
// accesses across.  Returns false when the workgroup is being abandoned.
#ifndef AZG_DUAL_TREE_NAP
#define AZG_DUAL_TREE_NAP 0   // 1: the tree waves poll their barrier and the net's acknowledgement at longer intervals
#endif
struct GroupSync {
    unsigned *ctr, *abort_word;
    unsigned target;
    bool nap;
    __device__ void init(unsigned *c, unsigned *a, bool n = false) { ctr = c; abort_word = a; target = 0; nap = n; }
    __device__ bool sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        target += 4;
        if ((threadIdx.x & 63) == 0) atomicAdd(ctr, 1u);
        bool ok = false;
        for (int spin = 0; spin < kSpinLimit; spin++) {
            if ((int)(lds_load(ctr) - target) >= 0) { ok = true; break; }
            if (lds_load(abort_word)) break;
            if (nap) __builtin_amdgcn_s_sleep(6); else __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) lds_store(abort_word, 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return ok;
    }
};

// wait until *word differs from `seen` or *alt is set; returns the value of *word (== seen: alt or abort)
__device__ inline unsigned wait_change(const unsigned *word, unsigned seen, const unsigned *alt, unsigned *abort_word)
{
    unsigned v = seen;
    for (int spin = 0; spin < kSpinLimit; spin++) {
        v = lds_load(word);
        if (v != seen) break;
        if ((alt && lds_load(alt)) || lds_load(abort_word)) { v = lds_load(word); break; }
        __builtin_amdgcn_s_sleep(2);
        if (spin == kSpinLimit - 1) lds_store(abort_word, 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return v;
}

// wait until *word >= want (counters that only grow); false: the workgroup is being abandoned
__device__ inline bool wait_ge(const unsigned *word, unsigned want, unsigned *abort_word)
{
    bool ok = false;
    for (int spin = 0; spin < kSpinLimit; spin++) {
        if ((int)(lds_load(word) - want) >= 0) { ok = true; break; }
        if (lds_load(abort_word)) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) lds_store(abort_word, 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

template <int N_, int T_>
struct DualGeom : Cnn16Geom<N_, T_> {
    using G = Cnn16Geom<N_, T_>;
    using GL = GameLdsT<N_ * N_ + 1>;
    static_assert(T_ == 1 && G::SPARSE_FC, "the dual kernel is built for the one-leaf geometries (boards above 8x8)");
    static constexpr int NG = 2 * T_;                                      // games per workgroup
    // float offsets from here on.  The input image (G::IN_OFF, inside the CNN region) is written and
    // read by the tree waves only, within one turn (planes, then conv1): one image serves both sets.
    static constexpr int GL_OFF = (G::END + 3) & ~3;
    static constexpr int GL_WORDS = (int)(sizeof(GL) / 4);
    // per set: the head convolutions' sums [6][S] (scaled by 2^8, without bias; the net waves ADD
    // their partial sums here, the tree waves clear it), then probabilities [S] and the value
    // (with one channel tile per wave conv3 has four channel groups: two such blocks, each fed by two
    // of them, added in the order of head_sum<4>)
    static constexpr int TX_OFF = 0;
    static constexpr int HB = ConvWaves<G>::HPARTS == 4 ? 2 : 1;         // blocks of head sums
    static constexpr int TXB = (6 * G::S + 7) & ~7;
    static constexpr int TX_WORDS = HB * TXB;
    static constexpr int TLG_OFF = TX_OFF + TX_WORDS;
    static constexpr int TVAL_OFF = TLG_OFF + ((G::S + 7) & ~7);
    static constexpr int TL_WORDS = TVAL_OFF + 8;
    static constexpr int TL_OFF = (GL_OFF + NG * GL_WORDS + 3) & ~3;       // 16-byte aligned: read and written as 4-word vectors
    // scratch of the tree waves' sparse FC, one block per wave: the (k, x) lists of the wave's
    // non-zero inputs (policy: QP pairs, value: QV pairs); once a list has been consumed, the wave's
    // partial sums take its place (policy sums [256] at word 0, value sums [4][64] at word 256)
    static constexpr int YV_OFF = 2 * G::QP > 256 ? 2 * G::QP : 256;      // the value list: behind the policy list AND the policy sums
    static constexpr int YL_WORDS = (YV_OFF + 2 * G::QV + 3) & ~3;
    static constexpr int YW_WORDS = YL_WORDS > 512 ? YL_WORDS : 512;
    static constexpr int YW_OFF = TL_OFF + 2 * TL_WORDS;
    static constexpr int CTL_OFF = YW_OFF + 4 * YW_WORDS;
    static constexpr int TOTAL = CTL_OFF + (int)(sizeof(DualCtl) / 4);
    static_assert(TOTAL * 4 <= 160 * 1024, "the dual kernel's LDS does not fit on a CU");
    static_assert(GL_WORDS % 2 == 0 && GL_OFF % 2 == 0 && TL_WORDS % 4 == 0 && YW_OFF % 4 == 0, "alignment of the records and of the vector accesses");
    __device__ static char *in_image(float *lds) { return reinterpret_cast<char *>(lds) + G::IN_OFF; }
};

// ---------------------------------------------------------------- convolution layers
#ifndef AZG_DUAL_BPF3
#define AZG_DUAL_BPF3 0      // conv3 of the dual kernels: LDS prefetch mode of conv3x3_f16 (2 measured 1 % slower: the tree waves bound the pair)
#endif
#ifndef AZG_DUAL_PRIO
#define AZG_DUAL_PRIO 2      // issue priority of the net waves (the critical path; they share their SIMDs with the tree waves)
#endif
#ifndef AZG_DUAL_TREE_PRIO
#define AZG_DUAL_TREE_PRIO 0 // issue priority of the tree waves
#endif
#ifndef AZG_DUAL_R3
#define AZG_DUAL_R3 2
#endif
#ifndef AZG_DUAL_R3_CW4
#define AZG_DUAL_R3_CW4 1    // conv3 weight ring depth when a wave streams all four channel tiles (8 registers per tile and k-step)
#endif
#ifndef AZG_DUAL_TAIL_SYNC
#define AZG_DUAL_TAIL_SYNC 0 // 1: a barrier of the tree waves behind the softmax (measured: not needed, costs ~0.6 k cycles per evaluation)
#endif
// Two-wide float32 arithmetic (v_pk_fma_f32) is kept out of these kernels.  Formed by the SLP
// vectoriser from the sparse policy layer's multiply-adds, the FIRST row's operation takes its x as the
// high dword of the (row index, x) register pair the list read returned (op_sel:[0,1,0]); on the tree
// waves, beside the matrix-dense net waves, the low half of the second such operation then comes out
// WITHOUT its product in lanes 48-55 in about one evaluation in a hundred (the low dword of the pair,
// an integer, read in place of x gives exactly that: a product below rounding).  Measured: 50-70 of 256
// identical searches wrong with the vectoriser on, or with only this loop written two-wide
// (-DAZG_EXP_PKFC=1); none when that one x is first moved out of its pair (-DAZG_EXP_PKFC=6, same
// waits, same registers otherwise), none with single-lane sums under the vectoriser.  It is not a
// late load (the single-lane sum from the same registers is right, and so is the two-wide one when
// every load is waited for), not register spilling, not the issue priorities; the same instruction
// forms in a stand-alone kernel beside MFMA-issuing waves are exact (tests/micro/pk_fma_beside_mfma.hip).
// The sums below are written so that no vectoriser can pair them, the library is built with
// -fno-slp-vectorize, and build.sh fails on any v_pk_*_f32 in the device code.
#ifndef AZG_EXP_PKFC
#define AZG_EXP_PKFC 0       // 1 / 6: the experiment builds described above
#endif
#ifndef AZG_FC_NT
#define AZG_FC_NT 0          // 1: the sparse FC reads its weight rows with the non-temporal hint
#endif
#if defined(AZG_EXP_NOFCLOAD)       // timing experiment (wrong results): the sparse FC reads one row over and over (vector-L1 hits)
#define AZG_FC_LOAD(p) (*(wrow))
#elif AZG_FC_NT
#define AZG_FC_LOAD(p) __builtin_nontemporal_load(p)
#else
#define AZG_FC_LOAD(p) (*(p))
#endif

// bias values of a wave's channel tiles, as the accumulator layout of v_mfma_f32_32x32x16 wants them
template <int CW>
struct Bias16 { f32x4 b[CW][4]; };

template <int CW>
struct LoadBias {
    const float *bias; int cot0, nc, half;
    __device__ Bias16<CW> operator()() const
    {
        Bias16<CW> c;
#pragma unroll
        for (int t = 0; t < CW; t++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                c.b[t][g] = *reinterpret_cast<const f32x4 *>(bias + (cot0 + (t < nc ? t : 0)) * 32 + 8 * g + 4 * half);
        return c;
    }
};

// epilogue of conv1 / conv2: bias, ReLU, split into f16 pairs, store to the hi / lo images
template <class G, int CW>
struct StoreSplit {
    char *oh, *ol; int rs, half;
    __device__ void operator()(int ci, int cot, int, const f32x16 &acc, int p, const Bias16<CW> &c) const
    {
        if (G::NP != G::NTP && p < 0) return;
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
    }
};

// conv1 of one leaf, run by the TREE waves (wave4 = 0..3) from the set's input image into the one
// conv1 output image; the caller has waited until the net waves are done with that image.
template <class D>
__device__ inline void dual_conv1(W1Ring &ring1, float *lds, const char *xin, int wave4, int lane)
{
    using G = typename D::G;
    char *ldb = reinterpret_cast<char *>(lds);
    constexpr int CW = ConvWaves<G>::CW;
    const float *lbias = lds + G::BIAS_OFF / 4;
    const StoreSplit<G, CW> st1 = {ldb + G::A1H_OFF, ldb + G::A1L_OFF, G::RS_A1, lane >> 5};
    conv1_packed_f16<G>(ring1, xin, wave4, lane, LoadBias<CW>{lbias, 0, 1, lane >> 5},
                        [=](int, int tile, const f32x16 &acc, int p, const Bias16<CW> &c) { st1(0, 0, tile, acc, p, c); }, nullptr, 13);
}

template <class D>
struct DualRings {
    WRing<3, ConvWaves<typename D::G>::CW> ring2;
};

// the first k-steps of conv2's weights: requested before the net waves wait for a leaf
template <class D>
__device__ inline void dual_preload(const NetWeights &nw_in, DualRings<D> &r)
{
    using G = typename D::G;
    const NetWeights nw = opaque_weights(nw_in);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    conv_preload<G, 32, 64, 3, ConvWaves<G>::CW>(reinterpret_cast<const half8 *>(nw.w16[1]), wave, lane, r.ring2);
}

// ---------------------------------------------------------------- net waves
// conv2, conv3 and the head convolutions of one leaf whose conv1 output the tree waves have left in
// the A1 images; tl: the set's hand-over region.  The body follows cnn16_forward (azg_cnn16.hpp) with
// the group's barriers and without the matrix gate (there is no second dense workgroup on the CU).
// conv3's epilogue ADDS each wave's head sums into tl[TX_OFF..] with LDS float atomics: every element
// gets exactly two addends (one per channel group of conv3) on top of the zero the tree waves left
// there, so the result does not depend on the order of arrival and equals the partial-sum addition of
// the one-workgroup kernels.  rings.ring2 comes in loaded (dual_preload) and leaves loaded for the
// next leaf.  a1gen / a1count: see DualCtl.
template <class D>
__device__ inline bool dual_front(const NetWeights &nw_in, float *lds, float *tl, DualRings<D> &rings, unsigned *a1gen, unsigned &a1count,
                                  GroupSync &gs, Prof &prof)
{
    using G = typename D::G;
    const NetWeights nw = opaque_weights(nw_in);
    char *ldb = reinterpret_cast<char *>(lds);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    __builtin_assume(wave >= 0 && wave < 4);
    const int half = lane >> 5, j = lane & 31;
    const float *lbias = lds + G::BIAS_OFF / 4;
    constexpr int CW = ConvWaves<G>::CW, CW3 = ConvWaves<G>::CW3, HPARTS = ConvWaves<G>::HPARTS;
    static_assert(HPARTS <= 2 * D::HB, "at most two addends per head sum: the order of the atomic additions must not matter");
    constexpr int R2 = 3, R3 = CW3 == 4 ? AZG_DUAL_R3_CW4 : AZG_DUAL_R3;
    constexpr int BPF3 = AZG_DUAL_BPF3;
    const half8 *w2 = reinterpret_cast<const half8 *>(nw.w16[1]);
    const half8 *w3 = reinterpret_cast<const half8 *>(nw.w16[2]);
    WRing<R3, CW3> ring3;
    conv_preload<G, 64, 128, R3, CW3>(w3, wave, lane, ring3);
    conv3x3_f16<G, 32, 64, G::RS_A1, true, G::BPF, R2, CW, AZG_DUAL_PRIO != 0>(w2, rings.ring2, ldb + G::A1H_OFF, ldb + G::A1L_OFF, wave, lane,
        LoadBias<CW>{lbias + 32, (wave % ConvSplit<G, 64, CW>::WC) * CW, CW, half},
        StoreSplit<G, CW>{ldb + G::A2H_OFF, ldb + G::A2L_OFF, G::RS_A2, half}, &prof, 15);
    prof.mark(17);
    if (!gs.sync()) return false;
    a1count++;
    if (tid == 0) lds_store(a1gen, a1count);           // every net wave is done with conv1's output
    prof.mark(4);
    {
        using SP3 = ConvSplit<G, 128, CW3>;
        const int cg3 = wave % SP3::WC;
        struct Head16 { f32x4 b[CW3][4]; half8 w[CW3][2]; };
        float *hx = tl + D::TX_OFF + (cg3 >> 1) * D::TXB;       // channel groups 2g and 2g + 1 share a block
        f32x16 hd;
        conv3x3_f16<G, 64, 128, G::RS_A2, true, BPF3, R3, CW3, AZG_DUAL_PRIO != 0>(w3, ring3, ldb + G::A2H_OFF, ldb + G::A2L_OFF, wave, lane,
            [=]() {
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
            },
            [=, &hd](int ci, int, int, const f32x16 &acc, int p, const Head16 &c) {
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
                if (p < 0) return;                               // padding lane of the last tile
                // head rows 0-3 sit in registers 0-3 of lanes 0-31, rows 4-7 in those of lanes 32-63
                // (6 and 7 are not head channels); the low-half rows 8-15 in registers 4-7
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++)
                    if (4 * half + c4 < 6) {
                        if (HPARTS == 1) hx[(4 * half + c4) * G::S + p] = hd[c4] + hd[4 + c4];      // the wave holds all channels of its positions
                        else atomicAdd(hx + (4 * half + c4) * G::S + p, hd[c4] + hd[4 + c4]);
                    }
            }, &prof);
    }
    prof.mark(5);
    conv_preload<G, 32, 64, R2, CW>(w2, wave, lane, rings.ring2);      // conv2's first weights for the next leaf
    return gs.sync();                      // every net wave's additions are in
}

// ---------------------------------------------------------------- tree waves: fully connected layers, softmax, tanh
// tid4: thread index inside the group (0..255).  Reads and then CLEARS the head sums at
// tl[TX_OFF..]; results at tl[TLG_OFF..], tl[TVAL_OFF].
template <class D>
__device__ inline bool dual_tail(const NetWeights &nw_in, float *lds, float *tl, int tid4, GroupSync &gs, Prof &prof)
{
    using G = typename D::G;
    const NetWeights nw = opaque_weights(nw_in);
    const int wave = __builtin_amdgcn_readfirstlane(tid4 >> 6), lane = tid4 & 63;
    const float *lb = lds + G::BIAS_OFF / 4;
    struct Pair { int k; float x; };
    float *yw = lds + D::YW_OFF + wave * D::YW_WORDS;        // this wave's scratch block
    Pair *lp = reinterpret_cast<Pair *>(yw);
    Pair *lv = reinterpret_cast<Pair *>(yw + D::YV_OFF);
    float *hx = tl + D::TX_OFF;
    const float *bh = lb + G::B_HEAD;
    // this wave's inputs k = 4*i + wave: head output = ReLU(sum * 2^-8 + bias); the non-zero ones as
    // (k, x) lists (see cnn16_tail_sparse); every sum is cleared once read, for the next evaluation
    auto compact = [&](int kmax, int c0, Pair *list) {
        int n = 0;
        for (int i0 = 0; 4 * i0 + wave < kmax; i0 += WAVE) {
            const int k = 4 * (i0 + lane) + wave;
            float v = 0.0f;
            if (k < kmax) {
                const int e = c0 * G::S + k;
                float sum = hx[e];
                hx[e] = 0.0f;
                if (D::HB == 2) { sum += hx[D::TXB + e]; hx[D::TXB + e] = 0.0f; }
                v = relu_nan(__builtin_fmaf(sum, kWUnscale, bh[c0 + k / G::S]));
            }
            const bool nz = !(v <= 0.0f);
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
    wave_sync();                                              // the lists are read back by this wave only
    prof.mark(18);
    // Both layers stream one weight row per non-zero input from L2; the tree waves' loads share the
    // CU's vector-memory path with the conv weights of the net waves.  (Keeping the loads of the next
    // block of rows in flight while one block is multiplied was measured 4 % SLOWER: the bursts
    // delay the net waves' weight stream.)
    {
        const gptr<f32x4> wrow = as_global(reinterpret_cast<const f32x4 *>(nw.wpfc_k)) + lane;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        constexpr int U = AZG_SPARSE_FC_ROWS;
        for (int i0 = 0; i0 < np; i0 += U) {
            f32x4 wv[U];
            float xv[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const Pair e = lp[i0 + u < np ? i0 + u : np - 1];
                xv[u] = i0 + u < np ? e.x : 0.0f;
                wv[u] = AZG_FC_LOAD(wrow + (size_t)e.k * 64);
            }
#if AZG_EXP_PKFC == 1 || AZG_EXP_PKFC == 6
            // Hazard experiment (SLP vectoriser off everywhere else): the same multiply-adds as explicit
            // two-wide operations.  1: as the vectoriser forms them -- the first row's x stays the HIGH dword
            // of the (row index, x) pair the list read returned, selected with op_sel:[0,1,0]: sums come out
            // one term short in lanes 48-55.  6: the same, but that x is moved out of its pair first
            // (op_sel_hi:[1,0,1] like every other row): exact.
            {
                if (AZG_EXP_PKFC == 6) asm volatile("" : "+v"(xv[0]));
                f32x2 a01 = {acc[0], acc[1]}, a23 = {acc[2], acc[3]};
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const f32x2 x2 = {xv[u], xv[u]}, w01 = {wv[u][0], wv[u][1]}, w23 = {wv[u][2], wv[u][3]};
                    a01 = __builtin_elementwise_fma(w01, x2, a01);
                    a23 = __builtin_elementwise_fma(w23, x2, a23);
                }
                acc[0] = a01[0]; acc[1] = a01[1]; acc[2] = a23[0]; acc[3] = a23[1];
            }
#else
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    acc[jj] = __builtin_fmaf(wv[u][jj], xv[u], acc[jj]);
                    asm volatile("" : "+v"(acc[jj]));         // never two-wide: see the note at AZG_EXP_PKFC
                }
#endif
        }
        wave_sync();                                          // the policy list is dead: its place takes the sums
        *reinterpret_cast<f32x4 *>(yw + 4 * lane) = acc;
    }
    prof.mark(19);
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
                wv[u] = AZG_FC_LOAD(wrow + (size_t)e.k * 16);
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    acc[jj] = __builtin_fmaf(wv[u][jj], xv[u], acc[jj]);
                    asm volatile("" : "+v"(acc[jj]));         // never two-wide: see the note at AZG_EXP_PKFC
                }
        }
        wave_sync();                                          // both lists are dead now
        *reinterpret_cast<f32x4 *>(yw + 256 + grp * 64 + 4 * l16) = acc;
    }
    prof.mark(3);
    if (!gs.sync()) return false;
    prof.mark(8);
    if (wave == 0) {
        const float *y0 = lds + D::YW_OFF;
        float x[4];
        float mx = -3.0e38f;
        {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const f32x4 p = *reinterpret_cast<const f32x4 *>(y0 + w * D::YW_WORDS + 4 * lane);
#pragma unroll
                for (int jj = 0; jj < 4; jj++) s[jj] += p[jj];
            }
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int a = 4 * lane + jj;
                x[jj] = a < G::S ? s[jj] + lb[G::B_PFC + a] : -3.0e38f;
                mx = fmaxf(mx, x[jj]);
            }
        }
        float hdn = lb[G::B_VFC1 + lane];
#pragma unroll
        for (int w = 0; w < 16; w++) hdn += y0[(w >> 2) * D::YW_WORDS + 256 + (w & 3) * 64 + lane];
        hdn = relu_nan(hdn);
        mx = wave_max_f32(mx);
        float sum = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
            if (4 * lane + jj < G::S) sum += expf(x[jj] - mx);
        sum = wave_sum_f32(sum);
        const float lse = logf(sum);
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
            if (4 * lane + jj < G::S) tl[D::TLG_OFF + 4 * lane + jj] = expf((x[jj] - mx) - lse);
        const float v = wave_sum_f32(hdn * lb[G::B_WVFC2 + lane]);
        if (lane == 0) tl[D::TVAL_OFF] = tanhf(v + lb[G::B_VFC2]);
    }
    prof.mark(9);
#if defined(AZG_DUAL_DUMP) || AZG_DUAL_TAIL_SYNC
    return gs.sync();
#else
    // No barrier here: only wave 0 reads what it has just written (expand, backup), and the other
    // waves meet it again at the barrier behind the next descent before they touch anything.
    return true;
#endif
}

// Board.current_state of the set's leaf (game.py:92-116) as the net's input image; tree waves.
template <class D, class GL>
__device__ inline void dual_fill_input(char *xin, const GL *q, int tid4)
{
    using G = typename D::G;
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = tid4; p < G::NP; p += 256) {
        const int row = p / G::N, col = p - row * G::N;
        float pl[4];
        planes_at(q->leaf_bb, G::N, q->leaf_ply, q->leaf_last, q->first_player, q->live != 0, row, col, pl);
        half8 v0 = z8;
#pragma unroll
        for (int ch = 0; ch < 4; ch++) v0[ch] = (_Float16)pl[ch];
        *reinterpret_cast<half8 *>(xin + p * G::RS_IN) = v0;
        *reinterpret_cast<half8 *>(xin + p * G::RS_IN + 16) = z8;
    }
}

// ---------------------------------------------------------------- the kernel body
template <int N, int T>
__device__ __forceinline__ void search_dual_body(DevState st)
{
    using D = DualGeom<N, T>;
    using G = typename D::G;
    using GL = typename D::GL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *ldb = reinterpret_cast<char *>(smem);
    GL *gls = reinterpret_cast<GL *>(smem + D::GL_OFF);
    DualCtl *ctl = reinterpret_cast<DualCtl *>(smem + D::CTL_OFF);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool net_group = wave < 4;

    // ---- prologue, all eight waves: per-game records, constants of the net, what stays zero
    if (wave < D::NG) {
        const int g = blockIdx.x * D::NG + wave;
        GL *gl = &gls[wave];
        const bool live = (g < st.B) && st.status[g] == 0;
        if (live && lane < 2 * MAXW) ((u64 *)gl->root_bb)[lane] = st.bb[(size_t)g * 2 * MAXW + lane];
        if (lane == 0) {
            gl->live = live; gl->need_eval = 0; gl->term = 0; gl->gid = g;
            if (live) {
                gl->root_ply = st.ply[g]; gl->root_last = st.last_move[g]; gl->first_player = st.first_player[g];
                gl->arena = st.cur[g]; gl->used = st.used[g]; gl->big = st.big_of[g]; gl->cap = arena_cap(st, g);
                gl->root_act = meta_act(tree_view(st, g, st.cur[g]).meta[0]);
            }
        }
    }
    if (tid < (int)(sizeof(DualCtl) / 4)) reinterpret_cast<unsigned *>(ctl)[tid] = 0;
    {
        // zero rows of the input images and of the conv1 output images (nothing overwrites them here),
        // the head sums of both sets
        const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (tid >= 256 && tid < 256 + G::RS_IN / 16) {
            *reinterpret_cast<half8 *>(D::in_image(smem) + G::NP * G::RS_IN + (tid - 256) * 16) = z8;
        }
        if (tid >= 320 && tid < 320 + G::RS_A1 / 16) {
            *reinterpret_cast<half8 *>(ldb + G::A1H_OFF + G::NP * G::RS_A1 + (tid - 320) * 16) = z8;
            *reinterpret_cast<half8 *>(ldb + G::A1L_OFF + G::NP * G::RS_A1 + (tid - 320) * 16) = z8;
        }
        for (int i = tid; i < 2 * D::TL_WORDS; i += 512) smem[D::TL_OFF + i] = 0.0f;
    }
    if (tid < 256) cnn16_prepare<G>(st.net, smem);          // ends with a hardware barrier: see below
    else __syncthreads();                                     // the other half joins that barrier
    bool any_live = false;
#pragma unroll
    for (int t = 0; t < D::NG; t++) any_live |= gls[t].live != 0;
    if (!any_live) return;

    GroupSync gs;
    gs.init(&ctl->bar[net_group ? 0 : 1], &ctl->abort, AZG_DUAL_TREE_NAP && !net_group);
    Prof prof;
    prof.start(st.stamps, net_group ? 0 : 256);       // diagnostic builds: thread 0 times the net waves, thread 256 the tree waves

    if (net_group) {
        // ---- net waves: serve the sets in turn.  They are the critical path of the workgroup and
        // share their SIMDs with the tree waves: issue priority for everything they do.
        if (AZG_DUAL_PRIO) __builtin_amdgcn_s_setprio(AZG_DUAL_PRIO);
        unsigned served[2] = {0, 0}, a1count = 0;
        bool over[2] = {false, false};
        DualRings<D> rings;
        dual_preload<D>(st.net, rings);
        while (!(over[0] && over[1])) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (over[s]) continue;
                const unsigned r = wait_change(&ctl->req[s], served[s], &ctl->fin[s], &ctl->abort);
                if (lds_load(&ctl->abort)) return;
                if (r == served[s]) { over[s] = true; continue; }          // fin: no batch will follow
                prof.mark(1);
                if (!dual_front<D>(st.net, smem, smem + D::TL_OFF + s * D::TL_WORDS, rings, &ctl->a1gen, a1count, gs, prof)) return;
                served[s]++;
                if (tid == 0) lds_store(&ctl->ack[s], served[s]);
                prof.mark(10);
            }
        }
        prof.finish();
        return;
    }

    // ---- tree waves
    if (AZG_DUAL_TREE_PRIO) __builtin_amdgcn_s_setprio(AZG_DUAL_TREE_PRIO);
    const int tid4 = tid - 256, wave4 = wave - 4;
    const bool tree_wave = wave4 < T;
    int it[2] = {0, 0};
    bool pending[2] = {false, false}, over[2] = {false, false};
#pragma unroll
    for (int s = 0; s < 2; s++)
        if (!gls[s].live) it[s] = st.n_playout;              // nothing to search: the set only reports "no more batches"
    unsigned reqs[2] = {0, 0}, n_conv1 = 0;
    long long n_eval[2] = {0, 0};
    while (!(over[0] && over[1])) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (over[s]) continue;
            GL *gl = &gls[s];
            const int g = gl->gid;
            const bool active = tree_wave && gl->live;
            float *tl = smem + D::TL_OFF + s * D::TL_WORDS;
            if (pending[s]) {
                // the evaluation of this set's leaf: wait for the net waves, then FC + softmax here
                const unsigned a = wait_change(&ctl->ack[s], reqs[s] - 1, nullptr, &ctl->abort);
                if (lds_load(&ctl->abort) || a != reqs[s]) { lds_store(&ctl->abort, 1u); break; }
                prof.mark(0);
                if (!dual_tail<D>(st.net, smem, tl, tid4, gs, prof)) break;
#ifdef AZG_DUAL_DUMP
                if (it[s] == 0 && gl->gid < 256)
                    for (int i = tid4; i < D::TL_WORDS && i < 2048; i += 256) g_dual_dump[gl->gid][i] = tl[i];
#endif
                if (active) {
                    Descent d;
                    d.leaf = gl->leaf; d.depth = gl->depth; d.ply = gl->leaf_ply; d.last = gl->leaf_last;
                    const TreeView tv = tree_view(st, g, gl->arena, gl->big);
                    const float *probs = tl + D::TLG_OFF;
                    const float vf = tl[D::TVAL_OFF];
                    const bool bad = !(vf == vf) || !(probs[0] == probs[0]);      // see search_body
                    if (bad && lane == 0) st.err[g] |= 4;
                    wave_expand(st, tv, gl, d, g, lane, [=](int c) { return bad ? 0.0f : probs[c]; });
                    wave_backup(tv, gl, d.depth, bad ? 0.0 : (double)vf, lane);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
                it[s]++;
                pending[s] = false;
                prof.mark(2);
            }
            bool broke = false;
            while (it[s] < st.n_playout) {
                int term = 0;
                Descent d = {0, 0, 0, 0};
                if (active) {
                    const TreeView tv = tree_view(st, g, gl->arena, gl->big);
                    d = wave_select<GL, false, (N * N + WAVE - 1) / WAVE>(st, tv, gl, lane);
                    if (d.depth > 1 && st.n_in_row >= 3 && st.n_in_row <= 8)
                        term = wave_last_move_end(gl->leaf_bb, st.N, st.S, st.n_in_row, d.ply, d.last,
                                                  player_at(gl->first_player, d.ply - 1), lane);
                    else if (st.NW == 1) term = game_end_1w(gl->leaf_bb[0][0], gl->leaf_bb[1][0], st.N, st.S, st.n_in_row, d.ply);
                    else term = wave_game_end(gl->leaf_bb, st.N, st.S, st.n_in_row, d.ply, lane);
                    if (lane == 0) { gl->term = term; gl->need_eval = term == 0; }
                    n_eval[s] += term == 0;
                }
                if (!gs.sync()) { broke = true; break; }
                prof.mark(6);
                if (gl->live && gl->need_eval) {
                    // the leaf goes to the net waves: its input planes and conv1 first (all four tree
                    // waves), conv1 as soon as the net waves have read the previous conv1 output
                    // (conv1's weights are requested first: their L2 round trip passes behind the planes and the waits)
                    W1Ring ring1;
                    conv1_preload(reinterpret_cast<const half8 *>(opaque_weights(st.net).w16[0]), lane, ring1);
                    dual_fill_input<D>(D::in_image(smem), gl, tid4);
                    if (!gs.sync()) { broke = true; break; }
                    prof.mark(7);
                    if (!wait_ge(&ctl->a1gen, n_conv1, &ctl->abort)) { broke = true; break; }
                    prof.mark(12);
                    dual_conv1<D>(ring1, smem, D::in_image(smem), wave4, lane);
                    n_conv1++;
                    if (!gs.sync()) { broke = true; break; }
                    prof.mark(13);
                    reqs[s]++;
                    if (tid4 == 0) lds_store(&ctl->req[s], reqs[s]);
                    pending[s] = true;
                    break;
                }
                // the leaf ended the game: back the result up, no evaluation
                if (active) {
                    const TreeView tv = tree_view(st, g, gl->arena, gl->big);
                    const double v = term == 3 ? 0.0 : (term == player_at(gl->first_player, d.ply) ? 1.0 : -1.0);
                    wave_backup(tv, gl, d.depth, v, lane);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
                it[s]++;
                if (!gs.sync()) { broke = true; break; }        // need_eval / term are rewritten by the next descent
                prof.mark(16);
            }
            if (broke) break;
            if (!pending[s] && it[s] >= st.n_playout) {
                over[s] = true;
                if (tid4 == 0) lds_store(&ctl->fin[s], 1u);
            }
        }
        if (lds_load(&ctl->abort)) break;
    }
    prof.finish();
    if (lds_load(&ctl->abort)) {
        // a wait ran out: the net waves may be anywhere; flag every slot of the workgroup
        for (int t = 0; t < D::NG; t++)
            if (gls[t].live && tid4 == 0) st.err[gls[t].gid] |= 8;
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
        GL *gl = &gls[s];
        if (tree_wave && gl->live && lane == 0) { st.evals[gl->gid] += n_eval[s]; st.used[gl->gid] = gl->used; }
    }
}

template <int N, int T>
__global__ __launch_bounds__(512, 1) void search_dual_kernel(DevState st)
{
    search_dual_body<N, T>(st);
}

}  // namespace azg
