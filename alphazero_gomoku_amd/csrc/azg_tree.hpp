// Wave-level (64 lanes, one game per wave) board and PUCT tree primitives.
//
// Reference semantics restated here (files relative to the reference tree):
//   Board.do_move / has_a_winner / game_end / current_state   game.py:92-169
//   TreeNode.select / get_value / expand / update_recursive   mcts_alphaZero.py:38-83
//   MCTS._playout                                             mcts_alphaZero.py:113-148
// All tree arithmetic is IEEE double in the reference's operation order; this file
// must be compiled with -ffp-contract=off.
#pragma once
#include "azg_device.hpp"

namespace azg {

typedef unsigned long long u64;

// ---------------------------------------------------------------- small helpers
__device__ inline u64 mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// bit c of a MAXW-word set kept in LDS or global memory
__device__ inline bool bb_has(const u64 *set, int c) { return (set[c >> 6] >> (c & 63)) & 1ULL; }

// Working copy of one game during a search (lives in LDS).  PL = longest possible path (root ..
// leaf): cells + 1; the CNN kernels instantiate it for their board so that two workgroups fit
// on a CU.
template <int PL>
struct GameLdsT {
    u64 root_bb[2][MAXW];
    u64 leaf_bb[2][MAXW];
    double path_q[PL];   // Q of the path's edges as read during the descent
    int path[PL];        // edge indices root .. leaf
    int path_n[PL];      // visit counts of the path's edges as read during the descent
    int used;                  // edges in use in the live arena (DevState::used during a launch)
    int root_act;              // action stored in the root edge
    int depth;                 // entries in path
    int leaf;                  // edge index of the leaf (= path[depth-1])
    int arena;                 // live arena of the game's tree
    int root_ply, root_last, first_player;
    int leaf_ply, leaf_last;
    int live;
    int term;                  // leaf ended: 0 no, 1/2 winner, 3 tie
    int need_eval;             // leaf needs priors/value from the provider
    int gid;                   // slot index
    int big;                   // large arena the game lives in, or -1 (DevState::big_of during a launch)
    int cap;                   // edges the game's arena holds
    int spec_blk[4], spec_k[4];   // child blocks (first edge, children) the last descent went through below the root's, or -1
};
typedef GameLdsT<MAXS + 1> GameLds;

struct TreeView {              // live arena of one game
    float *P; int *N; double *Q; int *child; unsigned *meta;
};

// big: DevState::big_of[g] (the large arena the game lives in, or -1)
__device__ inline TreeView tree_view(const DevState &st, int g, int arena, int big)
{
    TreeView t;
    if (big >= 0) {
        const size_t off = ((size_t)big * 2 + (size_t)arena) * (size_t)st.cap_big;
        t.P = st.bP + off; t.N = st.bN + off; t.Q = st.bQ + off;
        t.child = st.bChild + off; t.meta = st.bMeta + off;
    } else {
        const size_t off = ((size_t)g * 2 + (size_t)arena) * (size_t)st.cap;
        t.P = st.eP + off; t.N = st.eN + off; t.Q = st.eQ + off;
        t.child = st.eChild + off; t.meta = st.eMeta + off;
    }
    return t;
}
__device__ inline TreeView tree_view(const DevState &st, int g, int arena) { return tree_view(st, g, arena, st.big_of[g]); }
__device__ inline int arena_cap(const DevState &st, int g) { return st.big_of[g] >= 0 ? st.cap_big : st.cap; }

// Lanes of one wave exchange data through LDS (one lane writes, others read).  The
// hardware executes a wave's LDS instructions in order, but the compiler sees 64
// independent threads, so every such hand-over needs a wavefront-scope fence to keep
// the reads behind the join of the writing branch.
__device__ inline void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// Wave-wide reductions on the DPP network (register-to-register lane permutations at VALU rate)
// instead of ds_bpermute round trips through the LDS crossbar: four steps make every lane of a
// row of 16 hold the row's result (lane^1, lane^2, mirror within 8, mirror within 16), the four
// rows are then combined from one lane each.  The result is wave-uniform.
template <int CTRL>
__device__ inline int dpp_int(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ inline double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = dpp_int<CTRL>((int)(b & 0xFFFFFFFFLL)), hi = dpp_int<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

__device__ inline double lane_value_f64(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ inline double wave_max_f64(double v)
{
    v = fmax(v, dpp_f64<DPP_XOR1>(v));
    v = fmax(v, dpp_f64<DPP_XOR2>(v));
    v = fmax(v, dpp_f64<DPP_HALF_MIRROR>(v));
    v = fmax(v, dpp_f64<DPP_MIRROR>(v));
    return fmax(fmax(lane_value_f64(v, 0), lane_value_f64(v, 16)), fmax(lane_value_f64(v, 32), lane_value_f64(v, 48)));
}

__device__ inline float tree_wave_max_f32(float v)
{
    v = fmaxf(v, __int_as_float(dpp_int<DPP_XOR1>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_int<DPP_XOR2>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_int<DPP_HALF_MIRROR>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_int<DPP_MIRROR>(__float_as_int(v))));
    const int b = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}

__device__ inline int wave_min_i32(int v)
{
    v = min(v, dpp_int<DPP_XOR1>(v));
    v = min(v, dpp_int<DPP_XOR2>(v));
    v = min(v, dpp_int<DPP_HALF_MIRROR>(v));
    v = min(v, dpp_int<DPP_MIRROR>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// Outcome of a descent, identical in every lane.
struct Descent {
    int leaf;        // edge index of the leaf
    int depth;       // entries in GameLds::path (root .. leaf)
    int ply;         // stones on the leaf board
    int last;        // last move on the leaf board
};

// side to move (1 or 2) at a given ply
__device__ inline int player_at(int first_player, int ply)
{
    return (ply & 1) ? (3 - first_player) : first_player;
}

// ---------------------------------------------------------------- game_end
// game.py:129-169 on a stone-set pair.  Every lane owns cells lane, lane+64, ...;
// a cell "wins" when it anchors an n-long ray of its owner's stones in one of the
// four reference directions.  The reference scans anchors in ascending order and
// reports the owner of the first hit.  Returns 0 (not ended), 1 / 2 (winner) or
// 3 (tie); wave-uniform.
__device__ inline int wave_game_end(const u64 (*bb)[MAXW], int N, int S, int nrow, int nstones,
                                    int lane)
{
    if (nstones >= nrow + 2) {                                   // game.py:136-137
        for (int c0 = 0; c0 < S; c0 += WAVE) {
            const int c = c0 + lane;
            int p = 0;
            bool hit = false;
            if (c < S) {
                p = bb_has(bb[0], c) ? 1 : (bb_has(bb[1], c) ? 2 : 0);
                if (p) {
                    const u64 *mine = bb[p - 1];
                    const int h = c / N, w = c - h * N;
                    const bool wr = (w <= N - nrow), hr = (h <= N - nrow), wl = (w >= nrow - 1);
                    bool a = wr, b = hr, d = wr && hr, e = wl && hr;
                    for (int j = 1; j < nrow; j++) {
                        if (a) a = bb_has(mine, c + j);
                        if (b) b = bb_has(mine, c + j * N);
                        if (d) d = bb_has(mine, c + j * (N + 1));
                        if (e) e = bb_has(mine, c + j * (N - 1));
                    }
                    hit = a || b || d || e;
                }
            }
            const u64 m = __ballot(hit);
            if (m) {
                const int src = __ffsll((long long)m) - 1;
                return __shfl(p, src);
            }
        }
    }
    return nstones == S ? 3 : 0;                                 // game.py:167-168
}

// The same result from the last move alone, for a position whose predecessors had not ended (a
// leaf below the root: a node is only expanded when its game goes on, so a winning line at the
// leaf contains the stone just played and belongs to its mover; the reference's scan over all
// stones then reports exactly that player).  Lane 16*d + o + 8 tests the cell o steps from the
// last move along direction d (o = -8 .. 7, used up to n_in_row - 1); a direction wins when the
// run of the mover's stones through the last move is at least n_in_row long.  n_in_row <= 8.
__device__ inline int wave_last_move_end(const u64 (*bb)[MAXW], int N, int S, int nrow, int nstones, int last,
                                         int mover, int lane)
{
    if (nstones >= nrow + 2) {                                   // game.py:136-137
        const u64 *mine = bb[mover - 1];
        const int d = lane >> 4, o = (lane & 15) - 8;
        const int dr = d == 0 ? 0 : 1, dc = d == 1 ? 0 : (d == 3 ? -1 : 1);
        const int r0 = last / N, c0 = last - r0 * N;
        const int r = r0 + o * dr, c = c0 + o * dc;
        const bool on = o > -nrow && o < nrow && r >= 0 && r < N && c >= 0 && c < N && bb_has(mine, r * N + c);
        const u64 m = __ballot(on);
        for (int k = 0; k < 4; k++) {
            const unsigned w = (unsigned)(m >> (16 * k)) & 0xFFFFu;
            const unsigned up = ~(w >> 8) & 0x1FFu;              // bits 8..15 -> 0..7, bit 8 forced zero stop
            const int n_up = __ffs((int)up) - 1;                 // stones at offsets 0, 1, ... (the last move included)
            const unsigned dn = ~w & 0xFFu;                      // zeros among offsets -8 .. -1 (bits 0..7)
            const int n_dn = dn ? 7 - (31 - __clz((int)dn)) : 8; // stones at offsets -1, -2, ...
            if (n_up + n_dn >= nrow) return mover;
        }
    }
    return nstones == S ? 3 : 0;                                 // game.py:167-168
}

// The same test for boards of at most 64 cells (one word per colour), as whole-board bit
// operations: bit c of the anchor set is on when cells c, c+d, ..., c+(n-1)d are all the
// player's and the ray stays on the board (the reference's range tests, game.py:143-157).
// The reference reports the owner of the lowest anchor over both colours.  All operands are
// wave-uniform, so every lane computes the same result.
__device__ inline u64 win_anchors_1w(u64 m, int N, int nrow, u64 right_ok, u64 left_ok, u64 rows_ok)
{
    u64 a = m, b = m, d = m, e = m;
    for (int j = 1; j < nrow; j++) {
        a &= m >> j;
        b &= m >> (j * N);
        d &= m >> (j * (N + 1));
        e &= m >> (j * (N - 1));
    }
    return (a & right_ok) | (b & rows_ok) | (d & right_ok & rows_ok) | (e & left_ok & rows_ok);
}

__device__ inline int game_end_1w(u64 s1, u64 s2, int N, int S, int nrow, int nstones)
{
    if (nstones >= nrow + 2) {                                   // game.py:136-137
        const u64 row_right = (1ULL << (N - nrow + 1)) - 1ULL;               // w <= N - n
        const u64 row_left = ((1ULL << N) - 1ULL) & ~((1ULL << (nrow - 1)) - 1ULL);   // w >= n - 1
        u64 right_ok = 0, left_ok = 0;
        for (int r = 0; r < N; r++) { right_ok |= row_right << (r * N); left_ok |= row_left << (r * N); }
        const int nr = (N - nrow + 1) * N;                                   // h <= N - n
        const u64 rows_ok = nr >= 64 ? ~0ULL : (1ULL << nr) - 1ULL;
        const u64 w1 = win_anchors_1w(s1, N, nrow, right_ok, left_ok, rows_ok);
        const u64 w2 = win_anchors_1w(s2, N, nrow, right_ok, left_ok, rows_ok);
        const u64 any = w1 | w2;
        if (any) return ((w1 & (any & (~any + 1ULL))) != 0) ? 1 : 2;          // owner of the lowest anchor
    }
    return nstones == S ? 3 : 0;                                 // game.py:167-168
}

// ---------------------------------------------------------------- select
// Descend from the root to a leaf (mcts_alphaZero.py:122-129).  Lanes hold one child
// each (four at most on 16x16); the argmax keeps the first maximum in ascending child
// order, which is the order `max()` meets them in the reference.  CPL: children per lane the
// caller's board can need.  UNIFORM_P: every child of a
// node carries the prior 1/K in double (the rollout opponent's policy_value_fn,
// mcts_pure.py:33-46), computed instead of loaded.
#ifndef AZG_SELECT_SCREEN
#define AZG_SELECT_SCREEN 1       // 0: every level takes the double-precision pass
#endif
#ifndef AZG_SELECT_REG_STONES
#define AZG_SELECT_REG_STONES 1   // 0: lane 0 updates the leaf's stones in LDS on every level
#endif
#ifndef AZG_SELECT_PRIO
// A descent that goes deep (a trained net: six path entries on average instead of three) holds up its
// workgroup's whole playout iteration while the partner workgroup's dense loops, at priority 2, take
// the issue slots of the shared SIMDs.  From the third level below the root on it runs at priority 3:
// same-box A/B, two interleaved rounds: trained net 11.99 -> 11.59 ms per launch, random-init
// benchmark 11.06 -> 11.02 (from the start of the descent: 11.63 and 11.15, the benchmark loses).
#define AZG_SELECT_PRIO 3         // > 0: the descent runs at this issue priority (the dense loops of the CNN run at 2) ...
#endif
#ifndef AZG_SELECT_PRIO_LEVEL
#define AZG_SELECT_PRIO_LEVEL 3   // ... from this level below the root on (0: from the start)
#endif
// SCREEN: the float32 screen in front of the double-precision pass (below).  Its margin is derived
// for |Q| <= 1 and priors in [0, 1], which the engine's own providers guarantee; the descent that
// serves a caller-evaluated playout (arbitrary values and priors) is instantiated without it.
template <class GL, bool UNIFORM_P = false, int CPL = MAXS / WAVE, bool SCREEN = true>
__device__ inline Descent wave_select(const DevState &st, const TreeView &t, GL *gl, int lane, Prof *pf = nullptr)
{
    // (pf: phase timer of a diagnostic build, slots 20-23 per level: path record of the level above and
    //  loop head / loads issued -> screen evaluated / double pass and argmax / hand-over of the chosen child)
    // (only the one-child-per-lane kernels, where the descent shares its SIMD with a partner workgroup's
    // dense loops; in the two-wave-group kernel of the larger boards it measured 2.5 % slower)
    constexpr bool RAISE = AZG_SELECT_PRIO > 0 && CPL == 1 && !UNIFORM_P;
    if (RAISE && AZG_SELECT_PRIO_LEVEL == 0) __builtin_amdgcn_s_setprio(AZG_SELECT_PRIO);
    // leaf board starts as a copy of the root board (copy.deepcopy, :162)
    if (lane < 2 * MAXW) ((u64 *)gl->leaf_bb)[lane] = ((u64 *)gl->root_bb)[lane];
#if AZG_SELECT_REG_STONES
    // The stones along the path are kept in registers (every lane the same words) and stored once at
    // the leaf: a read-modify-write of the LDS copy by lane 0 on every level is a dependent LDS round
    // trip per level, and so is re-reading who moves first.
    constexpr int LW = CPL < MAXW ? CPL : MAXW;                  // stone words a board of this kernel has
    u64 lb[2][LW];
#pragma unroll
    for (int w = 0; w < LW; w++) { lb[0][w] = gl->root_bb[0][w]; lb[1][w] = gl->root_bb[1][w]; }
    const int first_player = gl->first_player;
#endif
    wave_sync();
    int e = 0;                       // root edge
    int depth = 1;
    int ply = gl->root_ply, last = gl->root_last;
    int child = t.child[0];
    int K = meta_k(t.meta[0]);
    int Np = t.N[0];
    const double q0 = t.Q[0];
    if (lane == 0) { gl->path[0] = 0; gl->path_n[0] = Np; gl->path_q[0] = q0; }
    while (child >= 0) {
        if (RAISE && AZG_SELECT_PRIO_LEVEL > 0 && depth == 1 + AZG_SELECT_PRIO_LEVEL) __builtin_amdgcn_s_setprio(AZG_SELECT_PRIO);
        double best = -HUGE_VAL, bq = 0.0;
        int bidx = 0x7fffffff, bchild = -1, bn = 0;
        unsigned bmeta = 0;
        if (pf) pf->mark_after(20, depth);
        // a node has up to 64 * CPL children (CPL: the caller's bound, all boards: MAXS / WAVE), CPL
        // per lane: all of a lane's loads go out before the first is used (clamped index instead of
        // a guard), one memory round trip per level
        double q_[CPL];
        float p_[CPL];
        int n_[CPL], ch_[CPL];
        unsigned mt_[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            if (c * WAVE < K) {                                  // wave-uniform
                const int i = lane + c * WAVE;
                const int ei = child + (i < K ? i : 0);
                q_[c] = t.Q[ei];
                if (!UNIFORM_P) p_[c] = t.P[ei];
                n_[c] = t.N[ei];
                ch_[c] = t.child[ei];
                mt_[c] = t.meta[ei];
            }
        }
        // Screen in single precision: the same score evaluated in float32 differs from the double
        // one by less than 5.4e-7 * (1 + |largest score|) (two roundings in c_puct * P, one each in
        // sqrt, its product, the reciprocal, its product, the conversion of Q and the sum; |Q| <= 1,
        // u >= 0).  A child whose float32 score stands more than 4e-6 * (1 + |max|) clear of every
        // other child's -- over three times twice that bound -- is therefore the maximum of the
        // double scores too, and the double-precision pass (square root, division, 64-bit wave
        // maximum: the bulk of a level's instructions) is skipped.  Ties and near-ties take that pass.
        int ri_screen = -1;
        if (AZG_SELECT_SCREEN && SCREEN && !UNIFORM_P) {
            const float sq32 = __builtin_sqrtf((float)Np), c32 = (float)st.c_puct;
            float s32_[CPL];
            float b32 = -HUGE_VALF;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int i = lane + c * WAVE;
                s32_[c] = -HUGE_VALF;
                if (c * WAVE < K && i < K) {
                    s32_[c] = (float)q_[c] + ((c32 * p_[c]) * sq32) * __builtin_amdgcn_rcpf((float)(1 + n_[c]));
                    b32 = fmaxf(b32, s32_[c]);
                }
            }
            const float m32 = tree_wave_max_f32(b32);
            const float bar = m32 - 4e-6f * (1.0f + fabsf(m32));
            int near = 0, at = -1;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                if (c * WAVE < K) {                              // wave-uniform
                    const u64 mk = __ballot(lane + c * WAVE < K && s32_[c] >= bar);
                    near += __popcll(mk);
                    if (mk && at < 0) at = c * WAVE + __ffsll((long long)mk) - 1;
                }
            }
            if (near == 1 && m32 < HUGE_VALF && m32 > -HUGE_VALF) ri_screen = at;
        }
        if (pf) pf->mark_after(21, ri_screen);
        if (ri_screen < 0) {
        const double sq = sqrt((double)Np);                      // np.sqrt(parent._n_visits)
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int i = lane + c * WAVE;
            // x / 1.0 is x: where none of these 64 children has been visited (most nodes below the
            // first levels) the double-precision division, the longest instruction sequence of the
            // descent, is skipped for the whole wave
            const bool any_visited = __ballot(i < K && n_[c] != 0) != 0ULL;
            if (i < K) {
                const double p = UNIFORM_P ? 1.0 / (double)K : (double)p_[c];
                // get_value (:82-83): c_puct * P * sqrt(Np) / (1 + n), left to right, then Q + u
                const double num = (st.c_puct * p) * sq;
                const double u = any_visited ? num / (double)(1 + n_[c]) : num;
                const double s = q_[c] + u;
                if (s > best) { best = s; bq = q_[c]; bidx = i; bchild = ch_[c]; bn = n_[c]; bmeta = mt_[c]; }
            }
        }
        // first maximum in ascending child order: the wave maximum, then the lowest child index
        // among the lanes that hold it (a lane already kept its own first maximum)
        }
        int ri;
        if (ri_screen >= 0) {
            ri = ri_screen;
            // the lane that holds child ri hands over that child's fields
#pragma unroll
            for (int c = 0; c < CPL; c++)
                if (c == (ri >> 6)) { bq = q_[c]; bchild = ch_[c]; bn = n_[c]; bmeta = mt_[c]; }
        } else {
            const double rs = wave_max_f64(best);
            // (a score that compares with nothing -- a NaN -- would leave no lane holding the maximum;
            // the index is kept inside the node's block whatever the scores are)
            ri = min(wave_min_i32(best == rs ? bidx : 0x7fffffff), K - 1);
        }
        if (pf) pf->mark_after(22, ri);
        const int src = ri & 63;
        e = child + ri;
        child = __builtin_amdgcn_readlane(bchild, src);
        Np = __builtin_amdgcn_readlane(bn, src);
        const double qsel = lane_value_f64(bq, src);
        const unsigned mt = (unsigned)__builtin_amdgcn_readlane((int)bmeta, src);
        K = meta_k(mt);
        const int act = meta_act(mt);
        if (pf) pf->mark_after(23, (int)mt);
        // Board.do_move (game.py:119-127) on the leaf copy
#if AZG_SELECT_REG_STONES
        {
            const int mover = player_at(first_player, ply);
            const u64 bit = 1ULL << (act & 63);
#pragma unroll
            for (int w = 0; w < LW; w++)
                if (w == (act >> 6)) { if (mover == 1) lb[0][w] |= bit; else lb[1][w] |= bit; }
        }
        if (lane == 0) {
            gl->path[depth] = e;
            gl->path_n[depth] = Np;
            gl->path_q[depth] = qsel;
        }
#else
        if (lane == 0) {
            const int mover = player_at(gl->first_player, ply);
            gl->leaf_bb[mover - 1][act >> 6] |= 1ULL << (act & 63);
            gl->path[depth] = e;
            gl->path_n[depth] = Np;
            gl->path_q[depth] = qsel;
        }
#endif
        depth++;
        ply++;
        last = act;
    }
    if (lane == 0) {
        gl->depth = depth; gl->leaf = e; gl->leaf_ply = ply; gl->leaf_last = last;
#if AZG_SELECT_REG_STONES
#pragma unroll
        for (int w = 0; w < LW; w++) { gl->leaf_bb[0][w] = lb[0][w]; gl->leaf_bb[1][w] = lb[1][w]; }
#endif
    }
    wave_sync();
    if (RAISE) __builtin_amdgcn_s_setprio(0);
    Descent d;
    d.leaf = e; d.depth = depth; d.ply = ply; d.last = last;
    return d;
}

// The same descent for boards of at most 64 cells (one child per lane), with the loads of the first
// levels issued TOGETHER: consecutive playouts of a game mostly share the top of their path (always
// with a peaked prior), so next to the root's child block the blocks the previous descent went
// through at the next AZG_SPEC_LEVELS levels are requested at once.  A level whose block is the
// guessed one finds its children in registers; any other level loads them as wave_select does.
// What a block holds is read after the previous playout's backup either way and blocks never move
// during a search, so the guess decides only WHEN data arrives, never what is chosen.
// Exact (every parity test passes on a build with it) and measured SLOWER on one machine, two
// interleaved rounds: with a trained net 12.87 ms per launch against 12.69 ms (three levels; one level
// 12.78), with the random-init net 11.60 against 11.50 -- so it is off (0) and wave_select is used.
#ifndef AZG_SPEC_LEVELS
#define AZG_SPEC_LEVELS 0
#endif
template <class GL>
__device__ inline Descent wave_select_spec(const DevState &st, const TreeView &t, GL *gl, int lane)
{
    constexpr int SP = AZG_SPEC_LEVELS + 1;                      // entry 0: the root's child block
    static_assert(AZG_SPEC_LEVELS >= 0 && AZG_SPEC_LEVELS <= 4, "spec_blk holds four levels");
    if (lane < 2 * MAXW) ((u64 *)gl->leaf_bb)[lane] = ((u64 *)gl->root_bb)[lane];
    wave_sync();
    int e = 0;
    int depth = 1;
    int ply = gl->root_ply, last = gl->root_last;
    int child = t.child[0];
    int K = meta_k(t.meta[0]);
    int Np = t.N[0];
    const double q0 = t.Q[0];
    int sb[SP], sk[SP];
    double sq_[SP];
    float sp_[SP];
    int sn_[SP], sc_[SP];
    unsigned sm_[SP];
    sb[0] = child; sk[0] = K;
#pragma unroll
    for (int l = 1; l < SP; l++) { sb[l] = gl->spec_blk[l - 1]; sk[l] = gl->spec_k[l - 1]; }
#pragma unroll
    for (int l = 0; l < SP; l++) {
        sq_[l] = 0.0; sp_[l] = 0.0f; sn_[l] = 0; sc_[l] = -1; sm_[l] = 0;
        if (sb[l] > 0) {                                         // wave-uniform
            const int ei = sb[l] + (lane < sk[l] ? lane : 0);
            sq_[l] = t.Q[ei]; sp_[l] = t.P[ei]; sn_[l] = t.N[ei]; sc_[l] = t.child[ei]; sm_[l] = t.meta[ei];
        }
    }
    if (lane == 0) { gl->path[0] = 0; gl->path_n[0] = Np; gl->path_q[0] = q0; }
    int level = 0;
    while (child >= 0) {
        const double sq = sqrt((double)Np);                      // np.sqrt(parent._n_visits)
        double q = 0.0;
        float p = 0.0f;
        int n = 0, ch = -1;
        unsigned mt = 0;
        bool hit = false;
#pragma unroll
        for (int l = 0; l < SP; l++)
            if (level == l && child == sb[l]) { q = sq_[l]; p = sp_[l]; n = sn_[l]; ch = sc_[l]; mt = sm_[l]; hit = true; }
        if (!hit) {
            const int ei = child + (lane < K ? lane : 0);
            q = t.Q[ei]; p = t.P[ei]; n = t.N[ei]; ch = t.child[ei]; mt = t.meta[ei];
        }
        const bool any_visited = __ballot(lane < K && n != 0) != 0ULL;
        double best = -HUGE_VAL;
        int bidx = 0x7fffffff;
        if (lane < K) {
            // get_value (:82-83): c_puct * P * sqrt(Np) / (1 + n), left to right, then Q + u
            const double num = (st.c_puct * (double)p) * sq;
            const double u = any_visited ? num / (double)(1 + n) : num;
            best = q + u;
            bidx = lane;
        }
        const double rs = wave_max_f64(best);
        const int ri = min(wave_min_i32(best == rs ? bidx : 0x7fffffff), K - 1);
        e = child + ri;
        child = __builtin_amdgcn_readlane(ch, ri);
        Np = __builtin_amdgcn_readlane(n, ri);
        const double qsel = lane_value_f64(q, ri);
        const unsigned mtsel = (unsigned)__builtin_amdgcn_readlane((int)mt, ri);
        K = meta_k(mtsel);
        const int act = meta_act(mtsel);
        level++;
        if (lane == 0) {
            const int mover = player_at(gl->first_player, ply);
            gl->leaf_bb[mover - 1][act >> 6] |= 1ULL << (act & 63);
            gl->path[depth] = e;
            gl->path_n[depth] = Np;
            gl->path_q[depth] = qsel;
            if (level < SP && child >= 0) { gl->spec_blk[level - 1] = child; gl->spec_k[level - 1] = K; }
        }
        depth++;
        ply++;
        last = act;
    }
    if (lane == 0) { gl->depth = depth; gl->leaf = e; gl->leaf_ply = ply; gl->leaf_last = last; }
    wave_sync();
    Descent d;
    d.leaf = e; d.depth = depth; d.ply = ply; d.last = last;
    return d;
}

// ---------------------------------------------------------------- expand
// TreeNode.expand (:38-45): one child per empty cell in ascending order, prior taken
// from the provider's per-cell output without renormalisation.  `prior_of(cell)`
// returns the float32 prior.  Returns false when the arena is full.
template <class GL, typename PriorFn>
__device__ inline bool wave_expand(const DevState &st, const TreeView &t, GL *gl,
                                   const Descent &d, int g, int lane, PriorFn prior_of)
{
    const int leaf = d.leaf;
    const int K = st.S - d.ply;
    const int base = gl->used;
    if (base + K > gl->cap) {
        if (lane == 0) st.err[g] |= 1;
        return false;
    }
    int before = 0;
    for (int c0 = 0; c0 < st.S; c0 += WAVE) {
        const int c = c0 + lane;
        const bool empty = (c < st.S) && !bb_has(gl->leaf_bb[0], c) && !bb_has(gl->leaf_bb[1], c);
        const u64 m = __ballot(empty);
        if (empty) {
            const int rank = before + __popcll(m & ((1ULL << lane) - 1ULL));
            const int ei = base + rank;
            t.P[ei] = prior_of(c);
            t.N[ei] = 0;
            t.Q[ei] = 0.0;
            t.child[ei] = -1;
            t.meta[ei] = meta_pack(c, 0);
        }
        before += __popcll(m);
    }
    wave_sync();                                   // every lane has read gl->used
    if (lane == 0) {
        // the leaf's own action is the last move of the descent (the root keeps what it had)
        t.child[leaf] = base;
        t.meta[leaf] = meta_pack(d.depth == 1 ? gl->root_act : d.last, K);
        gl->used = base + K;
    }
    return true;
}

// ---------------------------------------------------------------- backup
// update_recursive(-leaf_value) (:57-73): N += 1; Q += 1.0*(x - Q)/N from the leaf up
// to and including the root, x changing sign at every step.
template <class GL>
__device__ inline void wave_backup(const TreeView &t, const GL *gl, int depth, double leaf_value,
                                   int lane)
{
    const int d = depth;
    for (int j = lane; j < d; j += WAVE) {
        // N and Q of the path's edges were read by the descent (nothing else writes this tree
        // during a launch), so the update needs no further loads
        const int e = gl->path[j];
        const double x = ((d - 1 - j) & 1) ? leaf_value : -leaf_value;
        const int n = gl->path_n[j] + 1;
        const double q = gl->path_q[j];
        t.N[e] = n;
        t.Q[e] = q + (1.0 * (x - q)) / (double)n;
    }
}

// ---------------------------------------------------------------- synthetic providers
__device__ inline u64 state_key(const u64 (*bb)[MAXW], int NW, int current_player)
{
    u64 k = 0x9E3779B97F4A7C15ULL ^ (u64)current_player;
    for (int c = 0; c < 2; c++)
        for (int w = 0; w < NW; w++) k = mix64(k ^ bb[c][w]);
    return k;
}

__device__ inline float hash_prior(u64 key, int a)
{
    const u64 r = (mix64(key + 0x9E3779B97F4A7C15ULL * (u64)(a + 1)) >> 32) & 0xFFFULL;
    return (float)(r + 1) / 131072.0f;
}

__device__ inline double hash_value(u64 key)
{
    const u64 rv = (mix64(key ^ 0xD1B54A32D192ED03ULL) >> 32) & 0xFFFULL;
    return ((double)rv - 2048.0) / 2048.0;
}

// ---------------------------------------------------------------- feature planes
// Board.current_state (game.py:92-116) of one position: value of plane `pl` at plane
// coordinates (row, col); the cell shown at plane row r is board row N-1-r.
__device__ inline float plane_value(const u64 (*bb)[MAXW], int N, int ply, int last, int cur_player,
                                    int pl, int row, int col)
{
    const int cell = (N - 1 - row) * N + col;
    if (pl == 0) return bb_has(bb[cur_player - 1], cell) ? 1.0f : 0.0f;
    if (pl == 1) return bb_has(bb[2 - cur_player], cell) ? 1.0f : 0.0f;
    if (pl == 2) return (ply > 0 && cell == last) ? 1.0f : 0.0f;
    return (ply & 1) ? 0.0f : 1.0f;
}

// All four planes of one cell (same layout); `live` false gives zeros.  Every LDS word is
// read once and unconditionally, so the reads go out together.
__device__ inline void planes_at(const u64 (*bb)[MAXW], int N, int ply, int last, int first_player, bool live,
                                 int row, int col, float *out)
{
    const int cell = (N - 1 - row) * N + col;
    const u64 s1 = bb[0][cell >> 6], s2 = bb[1][cell >> 6];
    const bool first_to_move = ((ply & 1) == 0) == (first_player == 1);      // player 1 is to move
    const u64 mine = first_to_move ? s1 : s2, theirs = first_to_move ? s2 : s1;
    const float on = live ? 1.0f : 0.0f;
    out[0] = ((mine >> (cell & 63)) & 1ULL) ? on : 0.0f;
    out[1] = ((theirs >> (cell & 63)) & 1ULL) ? on : 0.0f;
    out[2] = (ply > 0 && cell == last) ? on : 0.0f;
    out[3] = (ply & 1) ? 0.0f : on;
}

}  // namespace azg
