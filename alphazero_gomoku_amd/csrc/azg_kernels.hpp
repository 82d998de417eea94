// Kernels of the self-play engine (gfx950).  One game per wave for all tree and board
// work; the search kernel keeps T games per 256-thread workgroup and runs a whole
// move's n_playout playouts in one launch with no inter-workgroup communication.
#pragma once
#include <type_traits>

#include "azg_cnn16.hpp"
#include "azg_tree.hpp"
#include "azg_dual.hpp"

namespace azg {

// The fence that orders one lane's global stores before other lanes' later loads of the
// same wave/workgroup (same CU, write-through L1).
__device__ inline void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// ---------------------------------------------------------------- search
// MCTS.get_move_probs' playout loop (mcts_alphaZero.py:161-163) for T games.
// MODE: AZG_PRIOR_NET (0), UNIFORM32 (1), HASH (2), PEAKED (3).  F16: CNN on split-f16 operands
// (azg_cnn16.hpp) instead of float32 MFMA (azg_cnn.hpp).
template <int N, int T, bool F16>
struct GeomPick { using type = typename std::conditional<F16, Cnn16Geom<N, T>, CnnGeom<N, T>>::type; };

template <class G, bool F16, class InputFn>
__device__ inline void net_forward(const NetWeights &nw, float *smem, InputFn input, Prof &prof,
                                   const MatrixGate &gate)
{
    if constexpr (F16) {
        cnn16_forward<G>(nw, smem, input, prof, gate);
    } else {
        cnn_forward<G>(nw, smem, [&](int leaf, int pl, int row, int col) -> float {
            float o[4];
            input(leaf, row, col, o);
            return o[pl];
        }, prof);
    }
}

template <class G, bool F16>
__device__ inline void net_prepare(const NetWeights &nw, float *smem)
{
    if constexpr (F16) cnn16_prepare<G>(nw, smem);
    else cnn_prepare<G>(nw, smem);
}

// The split-f16 kernels for boards up to 8x8 need < 80 KB of LDS; two waves per SIMD (second
// launch-bound argument) lets two workgroups share a CU.
#ifndef AZG_OCC8
#define AZG_OCC8 2      // workgroups per CU the split-f16 kernels for boards up to 8x8 are built for
#endif
#ifndef AZG_GATE8
#define AZG_GATE8 1     // per-CU matrix gate for those kernels
#endif
template <int N, int T, int MODE, bool F16>
__device__ __forceinline__ void search_body(DevState st)
{
    using G = typename GeomPick<(MODE == 0 ? N : 4), (MODE == 0 ? T : 1), F16>::type;
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    using GL = GameLdsT<(MODE == 0 ? N * N + 1 : MAXS + 1)>;
    // (experiment builds with two groups per workgroup: group `grp` owns its own region behind the
    // reserved head of the dynamic LDS and the game pair 2 * blockIdx.x + grp)
    constexpr bool kQuad = AZG_QUAD8 && F16 && MODE == 0;
    constexpr int kRegion = (((MODE == 0 ? G::END : 0) * 4 + T * (int)sizeof(GL) + 255) / 256) * 64;      // floats
    const int grp = kQuad ? (int)(threadIdx.x >> 8) : 0;
    float *smem = smem_all + (kQuad ? kLdsHead + grp * kRegion : 0);
    GL *gls = reinterpret_cast<GL *>(smem + (MODE == 0 ? G::END : 0));
    if (kQuad) {
        // the barrier and gate words start at zero; the only hardware barrier of the kernel
        if ((int)threadIdx.x < kLdsHead) reinterpret_cast<unsigned *>(smem_all)[threadIdx.x] = 0u;
        __syncthreads();
    }
    const int tid = kQuad ? azg_tid() : (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool tree_wave = wave < T;
    const int g = (kQuad ? 2 * blockIdx.x + grp : blockIdx.x) * T + wave;
    GL *gl = &gls[tree_wave ? wave : 0];

    if (tree_wave) {
        const bool live = (g < st.B) && st.status[g] == 0;
        if (live && lane < 2 * MAXW) ((u64 *)gl->root_bb)[lane] = st.bb[(size_t)g * 2 * MAXW + lane];
        if (lane == 0) {
            gl->live = live;
            gl->need_eval = 0;
            gl->term = 0;
            gl->gid = g;
            if (live) {
                gl->root_ply = st.ply[g];
                gl->root_last = st.last_move[g];
                gl->first_player = st.first_player[g];
            }
        }
    }
    if (kQuad) azg_block_barrier(); else __syncthreads();
    bool any_live = false;
#pragma unroll
    for (int t = 0; t < T; t++) any_live |= gls[t].live != 0;
    if (!any_live) return;
    if (MODE == 0) net_prepare<G, F16>(st.net, smem);

    const bool active = tree_wave && gl->live;
    if (active && lane == 0) {
        gl->arena = st.cur[g];
        gl->used = st.used[g];
        gl->big = st.big_of[g];
        gl->cap = arena_cap(st, g);
        gl->root_act = meta_act(tree_view(st, g, st.cur[g]).meta[0]);
        for (int l = 0; l < 4; l++) { gl->spec_blk[l] = -1; gl->spec_k[l] = 0; }
    }
    wave_sync();
    long long n_eval = 0;
    MatrixGate gate;
    gate.init(F16 && MODE == 0 && N <= 8 && AZG_GATE8);
    Prof prof;
    prof.start(st.stamps);

    for (int it = 0; it < st.n_playout; it++) {
        Descent d = {0, 0, 0, 0};
        int term = 0;
        if (active) {
            const TreeView tv = tree_view(st, g, gl->arena, gl->big);
            if constexpr (MODE == 0 && N * N <= WAVE && AZG_SPEC_LEVELS > 0) d = wave_select_spec<GL>(st, tv, gl, lane);
            else d = wave_select<GL, false, (MODE == 0 ? (N * N + WAVE - 1) / WAVE : MAXS / WAVE)>(st, tv, gl, lane, &prof);
            // (with n_in_row < 3 a line can exist before the reference's n_in_row + 2 stone threshold
            // lets it count, so it need not contain the last stone: full scan there)
            if (d.depth > 1 && st.n_in_row >= 3 && st.n_in_row <= 8)
                term = wave_last_move_end(gl->leaf_bb, st.N, st.S, st.n_in_row, d.ply, d.last,
                                          player_at(gl->first_player, d.ply - 1), lane);
            else if (st.NW == 1) term = game_end_1w(gl->leaf_bb[0][0], gl->leaf_bb[1][0], st.N, st.S, st.n_in_row, d.ply);
            else term = wave_game_end(gl->leaf_bb, st.N, st.S, st.n_in_row, d.ply, lane);
            if (lane == 0) { gl->term = term; gl->need_eval = term == 0; }
            n_eval += term == 0;
            prof.count(18, d.depth);
        }
        prof.mark(0);
        if (MODE == 0) {
            if (kQuad) azg_block_barrier(); else __syncthreads();
            prof.mark(1);
            bool any = false;
#pragma unroll
            for (int t = 0; t < T; t++) any |= gls[t].live && gls[t].need_eval;
            if (any) {
                net_forward<G, F16>(st.net, smem, [&](int leaf, int row, int col, float *out) {
                    const GL *q = &gls[leaf];
                    planes_at(q->leaf_bb, N, q->leaf_ply, q->leaf_last, q->first_player, q->live != 0, row, col, out);
                }, prof, gate);
            } else {
                if (kQuad) azg_block_barrier(); else __syncthreads();
            }
        }
        if (active) {
            if (MODE == 0) {
                // nothing of the descent is kept in registers across the CNN: it is re-read
                // from the game's LDS record (written before the barrier that opened the CNN)
                d.leaf = gl->leaf; d.depth = gl->depth; d.ply = gl->leaf_ply; d.last = gl->leaf_last;
                term = gl->term;
            }
            const TreeView tv = tree_view(st, g, gl->arena, gl->big);
            const int cur_player = player_at(gl->first_player, d.ply);
            double v;
            if (term) {
                // mcts_alphaZero.py:139-145: tie 0, else +1 if the winner is to move, else -1
                v = term == 3 ? 0.0 : (term == cur_player ? 1.0 : -1.0);
            } else if (MODE == 0) {
                const float *probs = smem + G::LG_OFF + wave * G::S;
                const float vf = smem[G::VAL_OFF + wave];
                // an activation beyond the half-precision range makes its high half infinite, the
                // next layer then forms inf - inf: a NaN that reaches every output of the position.
                // The slot is flagged (the host reports AZG_ERANGE) and NOTHING of such an evaluation
                // enters the tree: a NaN score would leave the PUCT argmax without a winner.
                const bool bad = F16 && (!(vf == vf) || !(probs[0] == probs[0]));
                if (bad && lane == 0) st.err[g] |= 4;
                wave_expand(st, tv, gl, d, g, lane, [=](int c) { return bad ? 0.0f : probs[c]; });
                v = bad ? 0.0 : (double)vf;
            } else if (MODE == 1) {
                const float p = 1.0f / (float)(st.S - d.ply);
                wave_expand(st, tv, gl, d, g, lane, [=](int) { return p; });
                v = 0.0;
            } else if (MODE == 3) {
                // one child carries almost all of the prior: the lowest empty cell
                int low = st.S;
                for (int w = st.NW - 1; w >= 0; w--) {
                    const u64 freew = ~(gl->leaf_bb[0][w] | gl->leaf_bb[1][w]);
                    if (freew) low = w * 64 + __ffsll((long long)freew) - 1;
                }
                wave_expand(st, tv, gl, d, g, lane, [=](int c) { return c == low ? 0.9375f : 0.0009765625f; });
                v = 0.0;
            } else {
                const u64 key = state_key(gl->leaf_bb, st.NW, cur_player);
                wave_expand(st, tv, gl, d, g, lane, [=](int c) { return hash_prior(key, c); });
                v = hash_value(key);
            }
            wave_backup(tv, gl, d.depth, v, lane);
            wg_fence();
        }
        prof.mark(10);
    }
    prof.finish();
    if (active && lane == 0) { st.evals[g] += n_eval; st.used[g] = gl->used; }
}

template <int N, int T, int MODE, bool F16>
__global__ __launch_bounds__(256, (F16 && MODE == 0 && N <= 8) ? AZG_OCC8 : 1) void search_kernel(DevState st)
{
    search_body<N, T, MODE, F16>(st);
}

// Entry points of the split-f16 kernels that run two workgroups per CU, capped at 240 registers:
// two waves per SIMD then leave 32 of the 512 registers free, which a wave of the runtime's copy
// kernel needs to run BESIDE the search (the noise upload of the pipelined move step); at 248 the
// copies wait for the whole search launch and a move step grows by 0.3 ms.  On gfx950 (unified
// register file) amdgpu_num_vgpr counts half of what it limits: 120 caps the kernel at 240.  The
// attribute takes no template-dependent argument, hence plain kernels around the common body.
#if AZG_OCC8 == 2
#ifndef AZG_VGPR_ATTR
#define AZG_VGPR_ATTR 120
#endif
#define AZG_CAPPED_SEARCH_KERNEL(NAME, N, T)                                                                      \
    __global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(AZG_VGPR_ATTR))) void NAME(DevState st) \
    {                                                                                                             \
        search_body<N, T, 0, true>(st);                                                                           \
    }
#if !defined(AZG_ONLY8) && !defined(AZG_ONLY15)
AZG_CAPPED_SEARCH_KERNEL(search_kernel_f16_6x6, 6, 2)
AZG_CAPPED_SEARCH_KERNEL(search_kernel_f16_7x7, 7, 2)
#endif
#ifndef AZG_ONLY15
AZG_CAPPED_SEARCH_KERNEL(search_kernel_f16_8x8, 8, 2)
#endif
#endif
#if AZG_QUAD8
// experiment: both game pairs of a CU in one workgroup (see azg_device.hpp)
__global__ __launch_bounds__(512, 1) void search_quad_kernel_8x8(DevState st) { search_body<8, 2, 0, true>(st); }
#endif

// ---------------------------------------------------------------- one playout, evaluated by the host
// MCTS._playout (mcts_alphaZero.py:113-148) split at the policy call (:132), for a caller whose
// policy_value_fn is arbitrary host code: playout_select_kernel descends to a leaf and parks the
// playout (path, leaf position) in `park`; the host evaluates the leaf; playout_finish_kernel
// expands with the host's priors (float32 by cell) and backs the value up.  One wave, one slot.
struct LeafInfo {                      // what the host needs to build the leaf position
    unsigned long long stones[2][MAXW];
    int ply, last_move, current_player, term;
};

__global__ __launch_bounds__(64) void playout_select_kernel(DevState st, int g, GameLds *park, LeafInfo *info)
{
    __shared__ GameLds s_gl;
    GameLds *gl = &s_gl;
    const int lane = threadIdx.x;
    if (lane < 2 * MAXW) ((u64 *)gl->root_bb)[lane] = st.bb[(size_t)g * 2 * MAXW + lane];
    if (lane == 0) {
        gl->live = 1; gl->gid = g; gl->need_eval = 0; gl->term = 0;
        gl->root_ply = st.ply[g]; gl->root_last = st.last_move[g]; gl->first_player = st.first_player[g];
        gl->arena = st.cur[g]; gl->used = st.used[g]; gl->big = st.big_of[g]; gl->cap = arena_cap(st, g);
        gl->root_act = meta_act(tree_view(st, g, st.cur[g]).meta[0]);
    }
    wave_sync();
    const TreeView tv = tree_view(st, g, gl->arena, gl->big);
    // (no float32 screen: the caller's values and priors may lie outside the range its margin assumes)
    const Descent d = wave_select<GameLds, false, MAXS / WAVE, false>(st, tv, gl, lane);
    const int term = wave_game_end(gl->leaf_bb, st.N, st.S, st.n_in_row, d.ply, lane);
    if (lane == 0) { gl->term = term; gl->need_eval = term == 0; }
    wave_sync();
    for (int i = lane; i < (int)(sizeof(GameLds) / 4); i += WAVE) ((int *)park)[i] = ((const int *)gl)[i];
    if (lane < 2 * MAXW) ((u64 *)info->stones)[lane] = ((const u64 *)gl->leaf_bb)[lane];
    if (lane == 0) {
        info->ply = d.ply; info->last_move = d.last;
        info->current_player = player_at(gl->first_player, d.ply);
        info->term = term;
    }
}

__global__ __launch_bounds__(64) void playout_finish_kernel(DevState st, int g, const GameLds *park, const float *priors,
                                                            double value)
{
    __shared__ GameLds s_gl;
    GameLds *gl = &s_gl;
    const int lane = threadIdx.x;
    for (int i = lane; i < (int)(sizeof(GameLds) / 4); i += WAVE) ((int *)gl)[i] = ((const int *)park)[i];
    wave_sync();
    Descent d;
    d.leaf = gl->leaf; d.depth = gl->depth; d.ply = gl->leaf_ply; d.last = gl->leaf_last;
    const TreeView tv = tree_view(st, g, gl->arena, gl->big);
    const int term = gl->term;
    double v = value;
    if (term) v = term == 3 ? 0.0 : (term == player_at(gl->first_player, d.ply) ? 1.0 : -1.0);   // mcts_alphaZero.py:139-145
    else wave_expand(st, tv, gl, d, g, lane, [=](int c) { return priors[c]; });
    wave_backup(tv, gl, d.depth, v, lane);
    wave_sync();
    if (lane == 0) st.used[g] = gl->used;
}

// ---------------------------------------------------------------- eval_batch
// PolicyValueNet.policy_value (policy_value_net_pytorch.py:80-94) on n states.
template <int N, int T, bool F16>
__global__ __launch_bounds__(256) void eval_kernel(NetWeights nw, const float *states, int n,
                                                   float *probs, float *values, int *bad)
{
    using G = typename GeomPick<N, T, F16>::type;
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float *smem = smem_all + ((AZG_QUAD8 && F16) ? kLdsHead : 0);
    if (AZG_QUAD8 && F16) {
        if ((int)threadIdx.x < kLdsHead) reinterpret_cast<unsigned *>(smem_all)[threadIdx.x] = 0u;
        __syncthreads();
    }
    const int n0 = blockIdx.x * T;
    net_prepare<G, F16>(nw, smem);
    Prof prof;
    prof.start(nullptr);
    net_forward<G, F16>(nw, smem, [&](int leaf, int row, int col, float *out) {
        const int i = n0 + leaf;
#pragma unroll
        for (int pl = 0; pl < 4; pl++)
            out[pl] = i < n ? states[((size_t)i * 4 + pl) * G::S + row * N + col] : 0.0f;
    }, prof, MatrixGate{nullptr});
    for (int idx = threadIdx.x; idx < T * G::S; idx += 256) {
        const int leaf = idx / G::S;
        if (n0 + leaf < n) probs[(size_t)n0 * G::S + idx] = smem[G::LG_OFF + idx];
    }
    if (threadIdx.x < T && n0 + threadIdx.x < n) {
        const float v = smem[G::VAL_OFF + threadIdx.x];
        values[n0 + threadIdx.x] = v;
        if (F16 && !(v == v)) atomicOr(bad, 1);             // see search_body: activations out of half range
    }
}

// ---------------------------------------------------------------- numpy-order sums
// numpy's add.reduce over a contiguous float64 vector (pairwise, 8 partial sums per
// block of at most 128), as used by np.sum in mcts_alphaZero.py:18.
__device__ inline double np_block_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
__device__ inline double np_sum_l1(const double *a, int n)
{
    if (n <= 128) return np_block_sum(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_block_sum(a, n2) + np_block_sum(a + n2, n - n2);
}
__device__ inline double np_sum(const double *a, int n)   // n <= 256
{
    if (n <= 128) return np_block_sum(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum_l1(a, n2) + np_sum_l1(a + n2, n - n2);
}

// ---------------------------------------------------------------- advance
enum { ADV_SAMPLE = 1, ADV_RECORD = 2, ADV_BOARD = 4, ADV_REROOT = 8, ADV_RESTART = 16 };

// index of sample t of the game recording into bank `bank` of slot g
__device__ inline size_t sample_index(const DevState &st, int g, int bank, int t)
{
    return ((size_t)g * 2 + (size_t)bank) * (size_t)st.S + (size_t)t;
}

// Start a fresh root (TreeNode(None, 1.0), mcts_alphaZero.py:108,192) in arena `arena`.
// A game that lived in a large arena gives it back.  One lane.
__device__ inline void fresh_root(const DevState &st, int g, int arena)
{
    const int big = st.big_of[g];
    if (big >= 0) {
        st.big_of[g] = -1;
        __hip_atomic_store(&st.big_owner[big], -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    TreeView t = tree_view(st, g, arena, -1);
    t.P[0] = 1.0f; t.N[0] = 0; t.Q[0] = 0.0; t.child[0] = -1; t.meta[0] = meta_pack(0xFFFF, 0);
    st.used[g] = 1;
    st.cur[g] = (unsigned char)arena;
}

// MCTS.update_with_move (mcts_alphaZero.py:183-192): the subtree below child edge `c` of the old
// arena becomes the whole tree of the new arena.
//
// Work list: the expanded nodes whose child block is still to be copied, as (new edge index, old
// block, number of children), a stack in LDS (rr: three rows of rr_cap ints, this wave's own).  A
// round pops up to RB nodes, gives their blocks the next free edges of the new arena, has the loads
// of all RB blocks in flight together and stores them; children that are expanded themselves go
// onto the stack, the others are final.  One memory round trip per RB nodes; nothing written here
// is read back.  An edge's child field is written exactly once: -1 with its block when it is a
// leaf, the new block address when its node is popped.
// When the stack is full its older half moves to the unused TAIL of the new arena (entry i at index
// cap-1-i of the N / child / meta arrays) and comes back when the LDS part has run empty.  The tail
// cannot meet the copied edges: every pending node still adds at least one edge, so
// edges written + pending nodes <= edges of the finished copy <= cap.
// The order in which nodes are copied (depth first here) decides only where blocks lie, not what
// the tree holds: children of a node stay contiguous and in ascending action order.
#ifndef AZG_REROOT_BLOCKS
#define AZG_REROOT_BLOCKS 8     // nodes per round (measured at 8x8: 4 -> +0.05 ms per move step, 16 -> +0.15 ms)
#endif
constexpr int kRerootStackMax = 512;            // LDS entries per wave (3 ints each)

__device__ inline void wave_reroot(const DevState &st, int g, int c, int lane, int *rr, int rr_cap)
{
    constexpr int RB = AZG_REROOT_BLOCKS;
    const int oa = st.cur[g], na = oa ^ 1;
    const TreeView o = tree_view(st, g, oa), n = tree_view(st, g, na);
    const int cap = arena_cap(st, g);
    int *se = rr, *so = rr + rr_cap, *sk = rr + 2 * rr_cap;
    const int rc = o.child[c];
    const unsigned rmeta = o.meta[c];
    if (lane == 0) {
        n.P[0] = o.P[c]; n.N[0] = o.N[c]; n.Q[0] = o.Q[c]; n.meta[0] = rmeta;
        if (rc < 0) n.child[0] = -1;
        else { se[0] = 0; so[0] = rc; sk[0] = meta_k(rmeta); }
    }
    wave_sync();
    int sp = rc >= 0 ? 1 : 0;          // entries on the LDS stack
    int spilled = 0;                   // entries in the arena tail
    int freep = 1;

    // children of a copied piece that are expanded themselves: onto the stack (at most 64)
    auto push = [&](bool on, int e_new, int oc, int k) {
        if (sp + WAVE > rr_cap) {
            // the older half of the stack moves to the tail, the newer half moves down
            const int h = rr_cap / 2;
            for (int i = lane; i < h; i += WAVE) {
                const int at = cap - 1 - (spilled + i);
                n.N[at] = se[i]; n.child[at] = so[i]; n.meta[at] = (unsigned)sk[i];
            }
            wave_sync();
            for (int i0 = 0; i0 < sp - h; i0 += WAVE) {
                const int i = i0 + lane;
                int a = 0, b = 0, d = 0;
                if (i < sp - h) { a = se[h + i]; b = so[h + i]; d = sk[h + i]; }
                wave_sync();
                if (i < sp - h) { se[i] = a; so[i] = b; sk[i] = d; }
                wave_sync();
            }
            spilled += h;
            sp -= h;
        }
        const u64 m = __ballot(on);
        if (on) {
            const int at = sp + __popcll(m & ((1ULL << lane) - 1ULL));
            se[at] = e_new; so[at] = oc; sk[at] = k;
        }
        sp += __popcll(m);
        wave_sync();
    };

    while (sp > 0 || spilled > 0) {
        if (sp == 0) {
            // the LDS part ran empty: the newest entries of the tail come back
            const int m = min(spilled, rr_cap / 2);
            wg_fence();
            for (int i = lane; i < m; i += WAVE) {
                const int at = cap - 1 - (spilled - m + i);
                se[i] = n.N[at]; so[i] = n.child[at]; sk[i] = (int)n.meta[at];
            }
            wave_sync();
            spilled -= m;
            sp = m;
        }
        const int m = min(RB, sp);
        int e4[RB], ob4[RB], kk4[RB], d4[RB];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const int at = sp - 1 - (u < m ? u : 0);
            e4[u] = se[at]; ob4[u] = u < m ? so[at] : -1; kk4[u] = u < m ? sk[at] : 0;
            d4[u] = freep;
            freep += kk4[u];
        }
        sp -= m;
        wave_sync();                                   // the popped entries are read before anything is pushed over them
        if (lane < m) {
            int e = e4[0], d = d4[0];
#pragma unroll
            for (int u = 1; u < RB; u++) if (lane == u) { e = e4[u]; d = d4[u]; }
            n.child[e] = d;
        }
        float p4[RB]; int n4[RB], c4[RB]; double q4[RB]; unsigned m4[RB]; bool on4[RB];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            on4[u] = lane < kk4[u];                    // the first 64 children of the block
            if (on4[u]) {
                p4[u] = o.P[ob4[u] + lane]; n4[u] = o.N[ob4[u] + lane]; q4[u] = o.Q[ob4[u] + lane];
                c4[u] = o.child[ob4[u] + lane]; m4[u] = o.meta[ob4[u] + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < RB; u++) {
            if (u >= m) break;                         // wave-uniform
            if (on4[u]) {
                const int at = d4[u] + lane;
                n.P[at] = p4[u]; n.N[at] = n4[u]; n.Q[at] = q4[u]; n.meta[at] = m4[u];
                if (c4[u] < 0) n.child[at] = -1;
            }
            push(on4[u] && c4[u] >= 0, d4[u] + lane, c4[u], (int)meta_k(m4[u]));
            // blocks wider than a wave (boards above 8x8): the remaining children
            for (int i0 = WAVE; i0 < kk4[u]; i0 += WAVE) {
                const int i = i0 + lane;
                const bool on = i < kk4[u];
                int cc = -1;
                unsigned mm = 0;
                if (on) {
                    const int from = ob4[u] + i, at = d4[u] + i;
                    cc = o.child[from]; mm = o.meta[from];
                    n.P[at] = o.P[from]; n.N[at] = o.N[from]; n.Q[at] = o.Q[from]; n.meta[at] = mm;
                    if (cc < 0) n.child[at] = -1;
                }
                push(on && cc >= 0, d4[u] + i, cc, (int)meta_k(mm));
            }
        }
    }
    if (lane == 0) { st.used[g] = freep; st.cur[g] = (unsigned char)na; }
}

// After a re-root: the next search adds at most n_playout nodes of at most S - ply children.  A
// tree that could outgrow its standard arena that way moves to a large arena (sized for any tree)
// BEFORE the search, so the search itself never runs out of room; edge indices are arena-relative,
// so the move is a plain copy.  When no large arena is free the tree is dropped instead (a fresh
// root: the game goes on without the re-used statistics) and counted in arena_stats[1].
__device__ inline void wave_fit_arena(const DevState &st, int g, int ply, int lane)
{
    wg_fence();
    const int used = st.used[g];
    if (st.big_of[g] >= 0 || (long long)used + (long long)st.n_playout * (st.S - ply) <= (long long)st.cap) return;
    int slot = -1;
    if (lane == 0) {
        for (int i = 0, p = g % max(st.n_big, 1); i < st.n_big; i++, p = p + 1 == st.n_big ? 0 : p + 1)
            if (atomicCAS(&st.big_owner[p], -1, g) == -1) { slot = p; break; }
    }
    slot = __shfl(slot, 0);
    const int a = st.cur[g];
    if (slot < 0) {
        if (lane == 0) { fresh_root(st, g, a); atomicAdd(&st.arena_stats[1], 1ULL); }
        return;
    }
    const TreeView o = tree_view(st, g, a, -1), n = tree_view(st, g, a, slot);
    for (int i = lane; i < used; i += WAVE) {
        n.P[i] = o.P[i]; n.N[i] = o.N[i]; n.Q[i] = o.Q[i]; n.child[i] = o.child[i]; n.meta[i] = o.meta[i];
    }
    wg_fence();
    if (lane == 0) { st.big_of[g] = slot; atomicAdd(&st.arena_stats[0], 1ULL); }
}

// MCTS.update_with_move (mcts_alphaZero.py:183-192) as the callers use it: child edge `c` becomes the
// root.  While the arena has room for another search behind everything it holds, that is five stores:
// the child's record goes to edge 0 of the SAME arena and the blocks below it stay where they are
// (the siblings' subtrees and the old root's block become garbage; edge indices are unchanged, so
// the tree the next search sees is the tree a copy would have made).  Only when
// used + n_playout * (S - ply) no longer fits is the kept subtree compacted into the other arena
// (wave_reroot), and if even the compacted tree may outgrow the arena it moves to a large one
// (wave_fit_arena).  ply: plies on the board the next search starts from.
__device__ inline void wave_update_with_move(const DevState &st, int g, int c, int ply, int lane, int *rr)
{
    const long long need = (long long)st.n_playout * (st.S - ply);
    if (st.rr_always || (long long)st.used[g] + need > (long long)arena_cap(st, g)) {
        wave_reroot(st, g, c, lane, rr, st.rr_cap);
        if (lane == 0) atomicAdd(&st.arena_stats[3], 1ULL);
    } else if (lane == 0) {
        atomicAdd(&st.arena_stats[2], 1ULL);
        const TreeView t = tree_view(st, g, st.cur[g]);
        const float p = t.P[c]; const int nv = t.N[c]; const double q = t.Q[c];
        const int ch = t.child[c]; const unsigned m = t.meta[c];
        t.P[0] = p; t.N[0] = nv; t.Q[0] = q; t.child[0] = ch; t.meta[0] = m;
    }
    wave_fit_arena(st, g, ply, lane);
}

// MCTSPlayer.get_action's tail (mcts_alphaZero.py:216-242) + Game.start_self_play's
// bookkeeping (game.py:381-391) for one game per wave.  noise: [B][S] Dirichlet draws
// in child order, u01: [B] uniforms (both from the game's numpy-legacy stream).
// ADV_RESTART (pipelined self-play): a slot whose game ends is recorded in fin_* and restarted
// at once as game first_id + k*stride, k drawn from the device counter (no restart once
// k >= total, total < 0 = no limit), so the next search needs no host round trip.
__global__ __launch_bounds__(256) void advance_kernel(DevState st, double temp, const double *noise,
                                                      const double *u01, const int *given, int flags,
                                                      int *moves_out, double *pi_out, long long first_id,
                                                      long long stride, long long total)
{
    __shared__ double s_e[4][MAXS];
    __shared__ double s_c[4][MAXS];
    __shared__ u64 s_bbw[4][2][MAXW];
    __shared__ int s_rr[4][3 * kRerootStackMax];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave;
    if (g >= st.B) return;
    int move = -1;
    const bool live = st.status[g] == 0;
    const bool want = live && ((flags & ADV_SAMPLE) || (given && given[g] >= 0));
    if ((flags & ADV_RESTART) && lane == 0) st.fin_flag[g] = 0;
    if (!want) {
        if (moves_out && lane == 0) moves_out[g] = -1;
        return;
    }
    double *e = s_e[wave], *cdf = s_c[wave];
    const int S = st.S;
    const TreeView t = tree_view(st, g, st.cur[g]);
    const int base = t.child[0];
    const int K = base >= 0 ? meta_k(t.meta[0]) : 0;
    const int ply = st.ply[g];
    int child_idx = -1;

    if ((flags & (ADV_SAMPLE | ADV_RECORD)) && K > 0) {
        // pi = softmax(1.0/temp * log(visits + 1e-10))   (mcts_alphaZero.py:16-19,179)
        double mx = -HUGE_VAL;
        for (int i = lane; i < K; i += WAVE) {
            const double x = (1.0 / temp) * log((double)t.N[base + i] + 1e-10);
            e[i] = x;
            mx = fmax(mx, x);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        for (int i = lane; i < K; i += WAVE) e[i] = exp(e[i] - mx);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        double tot = 0.0;
        if (lane == 0) tot = np_sum(e, K);
        tot = __shfl(tot, 0);
        for (int i = lane; i < K; i += WAVE) e[i] = e[i] / tot;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (flags & ADV_SAMPLE) {
            // np.random.choice(acts, p = 0.75*probs + 0.25*dirichlet): cdf = cumsum(p);
            // cdf /= cdf[-1]; index = searchsorted(cdf, u, side='right')
            // p = 0.75*probs + 0.25*noise element-wise (all lanes, coalesced noise read), then the
            // sequential cumsum of np.cumsum on one lane
            for (int i = lane; i < K; i += WAVE) cdf[i] = 0.75 * e[i] + 0.25 * noise[(size_t)g * S + i];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (lane == 0) {
                double run = 0.0;
                for (int i = 0; i < K; i++) {
                    run += cdf[i];
                    cdf[i] = run;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const double last = cdf[K - 1], u = u01[g];
            int cnt = 0;
            for (int i = lane; i < K; i += WAVE) cnt += (cdf[i] / last <= u) ? 1 : 0;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
            child_idx = min(cnt, K - 1);
            move = meta_act(t.meta[base + child_idx]);
        }
    }
    if (!(flags & ADV_SAMPLE)) {
        move = given[g];
        for (int i0 = 0; i0 < K; i0 += WAVE) {
            const int i = i0 + lane;
            const bool hit = i < K && meta_act(t.meta[base + i]) == move;
            const u64 m = __ballot(hit);
            if (m) { child_idx = i0 + __ffsll((long long)m) - 1; break; }
        }
    }
    if (moves_out && lane == 0) moves_out[g] = move;

    // training sample: state before the move, pi by cell (game.py:381-383)
    if (flags & ADV_RECORD) {
        const int sc = st.s_count[g];
        if (sc < S) {
            const size_t si = sample_index(st, g, st.s_bank[g], sc);
            double *pi = st.s_pi + si * S;
            for (int i = lane; i < S; i += WAVE) pi[i] = 0.0;
            wg_fence();
            for (int i = lane; i < K; i += WAVE) pi[meta_act(t.meta[base + i])] = e[i];
            if (lane < 2 * MAXW) st.s_bb[si * 2 * MAXW + lane] = st.bb[(size_t)g * 2 * MAXW + lane];
            if (lane == 0) { st.s_last[si] = st.last_move[g]; st.s_ply[si] = ply; st.s_count[g] = sc + 1; }
        }
    }
    if (pi_out) {
        for (int i = lane; i < S; i += WAVE) pi_out[(size_t)g * S + i] = 0.0;
        wg_fence();
        if (flags & (ADV_SAMPLE | ADV_RECORD))
            for (int i = lane; i < K; i += WAVE) pi_out[(size_t)g * S + meta_act(t.meta[base + i])] = e[i];
    }

    bool ended = false;
    if (flags & ADV_BOARD) {
        // Board.do_move (game.py:119-127) then game_end (game.py:162-169)
        const int mover = player_at(st.first_player[g], ply);
        if (lane < 2 * MAXW) ((u64 *)s_bbw[wave])[lane] = st.bb[(size_t)g * 2 * MAXW + lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (move < 0 || move >= S || bb_has(s_bbw[wave][0], move) || bb_has(s_bbw[wave][1], move)) {
            if (lane == 0) st.err[g] |= 2;
            return;
        }
        if (lane == 0) s_bbw[wave][mover - 1][move >> 6] |= 1ULL << (move & 63);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int term = wave_game_end(s_bbw[wave], st.N, S, st.n_in_row, ply + 1, lane);
        if (lane < 2 * MAXW) st.bb[(size_t)g * 2 * MAXW + lane] = ((u64 *)s_bbw[wave])[lane];
        if (lane == 0) {
            st.ply[g] = ply + 1;
            st.last_move[g] = move;
            if (term) { st.status[g] = 1; st.winner[g] = (signed char)(term == 3 ? -1 : term); }
        }
        ended = term != 0;
    }
    if (flags & ADV_REROOT) {
        if (ended || child_idx < 0) {
            if (lane == 0) fresh_root(st, g, st.cur[g]);      // reset_player / unknown move
        } else {
            wave_update_with_move(st, g, base + child_idx, (flags & ADV_BOARD) ? ply + 1 : ply, lane, s_rr[wave]);
        }
    }
    if ((flags & ADV_RESTART) && ended) {
        // Game.start_self_play returned (game.py:393-416); the next call's init_board /
        // reset_player (game.py:363, mcts_alphaZero.py:208-209) happen here for the slot
        long long k = 0;
        if (lane == 0) {
            st.fin_flag[g] = 1;
            st.fin_id[g] = st.game_id[g];
            st.fin_count[g] = st.s_count[g];
            st.fin_winner[g] = st.winner[g];
            st.fin_bank[g] = st.s_bank[g];
            k = (long long)atomicAdd(st.next_k, 1ULL);
        }
        k = __shfl(k, 0);
        if (total < 0 || k < total) {
            if (lane < 2 * MAXW) st.bb[(size_t)g * 2 * MAXW + lane] = 0;
            if (lane == 0) {
                st.ply[g] = 0; st.last_move[g] = -1; st.first_player[g] = 1;
                st.status[g] = 0; st.winner[g] = -1;
                st.s_count[g] = 0; st.s_bank[g] ^= 1;
                st.game_id[g] = first_id + k * stride;
            }
        }
    }
}

// MCTS.update_with_move alone (board untouched): moves[g] >= 0 re-root, -1 fresh, -2 skip.
__global__ __launch_bounds__(256) void reroot_kernel(DevState st, const int *moves)
{
    __shared__ int s_rr[4][3 * kRerootStackMax];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave;
    if (g >= st.B) return;
    const int move = moves[g];
    if (move == -2) return;
    const TreeView t = tree_view(st, g, st.cur[g]);
    const int base = t.child[0];
    const int K = base >= 0 ? meta_k(t.meta[0]) : 0;
    int child_idx = -1;
    if (move >= 0)
        for (int i0 = 0; i0 < K; i0 += WAVE) {
            const int i = i0 + lane;
            const bool hit = i < K && meta_act(t.meta[base + i]) == move;
            const u64 m = __ballot(hit);
            if (m) { child_idx = i0 + __ffsll((long long)m) - 1; break; }
        }
    if (child_idx < 0) { if (lane == 0) fresh_root(st, g, st.cur[g]); }
    else {
        wave_update_with_move(st, g, base + child_idx, st.ply[g], lane, s_rr[wave]);
    }
}

// ---------------------------------------------------------------- reset / set position
__global__ void reset_kernel(DevState st, const unsigned char *mask, const long long *ids)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= st.B || (mask && !mask[g])) return;
    for (int i = 0; i < 2 * MAXW; i++) st.bb[(size_t)g * 2 * MAXW + i] = 0;
    st.ply[g] = 0; st.last_move[g] = -1; st.first_player[g] = 1;
    st.status[g] = 0; st.winner[g] = -1; st.err[g] = 0;
    st.s_count[g] = 0; st.drained[g] = 0; st.fin_flag[g] = 0;
    st.game_id[g] = ids ? ids[g] : (long long)g;
    fresh_root(st, g, 0);
}

// ---------------------------------------------------------------- read-outs
// Root children statistics by cell (mcts_alphaZero.py:166-168).
__global__ void root_stats_kernel(DevState st, int *visits, double *q, double *prior, int *root_n)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave;
    if (g >= st.B) return;
    const TreeView t = tree_view(st, g, st.cur[g]);
    for (int i = lane; i < st.S; i += WAVE) {
        if (visits) visits[(size_t)g * st.S + i] = 0;
        if (q) q[(size_t)g * st.S + i] = 0.0;
        if (prior) prior[(size_t)g * st.S + i] = 0.0;
    }
    wg_fence();
    const int base = t.child[0];
    const int K = base >= 0 ? meta_k(t.meta[0]) : 0;
    for (int i = lane; i < K; i += WAVE) {
        const int a = meta_act(t.meta[base + i]);
        if (visits) visits[(size_t)g * st.S + a] = t.N[base + i];
        if (q) q[(size_t)g * st.S + a] = t.Q[base + i];
        if (prior) prior[(size_t)g * st.S + a] = (double)t.P[base + i];
    }
    if (root_n && lane == 0) root_n[g] = t.N[0];
}

// Board.current_state (game.py:92-116) of every slot, float32 [B][4][N][N].
__global__ void states_kernel(DevState st, float *out)
{
    const int g = blockIdx.x;
    const int S = st.S, N = st.N;
    const u64(*bb)[MAXW] = reinterpret_cast<const u64(*)[MAXW]>(st.bb + (size_t)g * 2 * MAXW);
    const int ply = st.ply[g], last = st.last_move[g];
    const int cur = player_at(st.first_player[g], ply);
    for (int idx = threadIdx.x; idx < 4 * S; idx += blockDim.x) {
        const int pl = idx / S, rem = idx - pl * S;
        out[(size_t)g * 4 * S + idx] = plane_value(bb, N, ply, last, cur, pl, rem / N, rem % N);
    }
}

// (state, pi, z) tuples of finished games (game.py:393-416).  map_g/map_t give slot and
// sample index of output row i.  Without map_aux the game is the one still sitting in the slot;
// with it (pipelined self-play: the slot may already hold the next game) map_aux[i] =
// bank | (winner + 1) << 8 and map_id[i] identify the finished game.
__global__ void drain_kernel(DevState st, int n, const int *map_g, const int *map_t, const int *map_aux,
                             const long long *map_id, float *states, double *pi, float *z, long long *gid,
                             int *ply_out, u64 *stones_out, int *last_out)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    const int g = map_g[i], t = map_t[i];
    const int S = st.S, N = st.N;
    const int bank = map_aux ? (map_aux[i] & 1) : st.s_bank[g];
    const size_t si = sample_index(st, g, bank, t);
    const u64(*bb)[MAXW] = reinterpret_cast<const u64(*)[MAXW]>(st.s_bb + si * 2 * MAXW);
    const int ply = st.s_ply[si], last = st.s_last[si];
    const int cur = player_at(st.first_player[g], ply);
    if (states)
        for (int idx = threadIdx.x; idx < 4 * S; idx += blockDim.x) {
            const int pl = idx / S, rem = idx - pl * S;
            states[(size_t)i * 4 * S + idx] = plane_value(bb, N, ply, last, cur, pl, rem / N, rem % N);
        }
    if (pi)
        for (int idx = threadIdx.x; idx < S; idx += blockDim.x)
            pi[(size_t)i * S + idx] = st.s_pi[si * S + idx];
    // the packed form of the state (what current_state is computed from): 2 x MAXW words + last move
    if (stones_out && threadIdx.x < 2 * MAXW) stones_out[(size_t)i * 2 * MAXW + threadIdx.x] = st.s_bb[si * 2 * MAXW + threadIdx.x];
    if (threadIdx.x == 0) {
        if (last_out) last_out[i] = last;
        const int w = map_aux ? (map_aux[i] >> 8) - 1 : (int)st.winner[g];
        if (z) z[i] = (w == -1) ? 0.0f : (w == cur ? 1.0f : -1.0f);
        if (gid) gid[i] = map_id ? map_id[i] : st.game_id[g];
        if (ply_out) ply_out[i] = ply;
    }
}

// Board.current_state (game.py:92-116) of n packed positions (stones [n][2][MAXW], last move, ply;
// the player who moves at even ply is `first_player`): float32 [n][4][N][N].  One block per position.
__global__ void unpack_states_kernel(int N, int n, int first_player, const u64 *stones, const int *last, const int *ply,
                                     float *out)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    const int S = N * N;
    const u64(*bb)[MAXW] = reinterpret_cast<const u64(*)[MAXW]>(stones + (size_t)i * 2 * MAXW);
    const int p = ply[i], l = last[i];
    const int cur = player_at(first_player, p);
    for (int idx = threadIdx.x; idx < 4 * S; idx += blockDim.x) {
        const int pl = idx / S, rem = idx - pl * S;
        out[(size_t)i * 4 * S + idx] = plane_value(bb, N, p, l, cur, pl, rem / N, rem % N);
    }
}

// TrainPipeline.get_equi_data (train.py:116-135): the 8 dihedral images of every (state, pi, z)
// tuple, in the reference's order: for i = 1..4 the counter-clockwise rotation by i*90 degrees,
// then that rotation mirrored left-right.  States are stored row-flipped (game.py:115), which is
// why pi is flipped up-down before the rotation and again after it.  One block per output tuple.
__device__ inline void rot90_src(int i, int N, int r, int c, int &sr, int &sc)
{
    // np.rot90(m, i)[r][c] == m[sr][sc]
    switch (i & 3) {
    case 1: sr = c; sc = N - 1 - r; break;
    case 2: sr = N - 1 - r; sc = N - 1 - c; break;
    case 3: sr = N - 1 - c; sc = r; break;
    default: sr = r; sc = c; break;
    }
}

__global__ void equi_kernel(int N, int n, const float *states, const double *pi, const float *z, float *states_out,
                            double *pi_out, float *z_out)
{
    const int o = blockIdx.x;
    if (o >= 8 * n) return;
    const int src = o >> 3, v = o & 7, i = (v >> 1) + 1, flip = v & 1;
    const int S = N * N;
    for (int idx = threadIdx.x; idx < 4 * S; idx += blockDim.x) {
        const int ch = idx / S, rem = idx - ch * S, r = rem / N, c = rem - r * N;
        int sr, sc;
        rot90_src(i, N, r, flip ? N - 1 - c : c, sr, sc);
        states_out[(size_t)o * 4 * S + idx] = states[((size_t)src * 4 + ch) * S + sr * N + sc];
    }
    for (int idx = threadIdx.x; idx < S; idx += blockDim.x) {
        const int r = idx / N, c = idx - r * N;
        // out = flipud(X); X = fliplr(R) or R; R = rot90(flipud(P), i)
        int sr, sc;
        rot90_src(i, N, N - 1 - r, flip ? N - 1 - c : c, sr, sc);
        pi_out[(size_t)o * S + idx] = pi[(size_t)src * S + (N - 1 - sr) * N + sc];
    }
    if (threadIdx.x == 0) z_out[o] = z[src];
}

// Stateless Board.do_move + game_end + current_state for n boards, one per wave.
__global__ __launch_bounds__(256) void board_step_kernel(int N, int nrow, int n, u64 *stones, int *ply,
                                                         int *last, const int *moves, const int *player,
                                                         unsigned char *ended, signed char *winner,
                                                         float *planes, int *bad)
{
    __shared__ u64 s_bbw[4][2][MAXW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= n) return;
    const int S = N * N;
    if (lane < 2 * MAXW) ((u64 *)s_bbw[wave])[lane] = stones[(size_t)b * 2 * MAXW + lane];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int mv = moves[b], pl = player[b];
    if (mv < 0 || mv >= S || bb_has(s_bbw[wave][0], mv) || bb_has(s_bbw[wave][1], mv) || pl < 1 || pl > 2) {
        if (lane == 0) atomicOr(bad, 1);
        return;
    }
    if (lane == 0) s_bbw[wave][pl - 1][mv >> 6] |= 1ULL << (mv & 63);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int np = ply[b] + 1;
    const int term = wave_game_end(s_bbw[wave], N, S, nrow, np, lane);
    if (lane < 2 * MAXW) stones[(size_t)b * 2 * MAXW + lane] = ((u64 *)s_bbw[wave])[lane];
    if (lane == 0) {
        ply[b] = np; last[b] = mv;
        ended[b] = term != 0;
        winner[b] = (signed char)((term == 0 || term == 3) ? -1 : term);
    }
    if (planes) {
        const int cur = 3 - pl;
        for (int idx = lane; idx < 4 * S; idx += WAVE) {
            const int p = idx / S, rem = idx - p * S;
            planes[(size_t)b * 4 * S + idx] = plane_value(s_bbw[wave], N, np, mv, cur, p, rem / N, rem % N);
        }
    }
}

}  // namespace azg
