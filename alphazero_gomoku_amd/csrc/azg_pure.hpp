// The rollout ("pure") MCTS opponent of the evaluation arena, one game per wave (gfx950).
//
// Reference semantics restated here (files relative to the reference tree):
//   MCTS._playout / _evaluate_rollout / get_move        mcts_pure.py:156-231
//   policy_value_fn / rollout_policy_fn                 mcts_pure.py:15-46
//   Board.get_move_target / move_in_rect                game.py:25-58
//   numpy legacy RandomState (MT19937, random_sample)   np.random.rand, third party
// The tree lives in the same edge arrays as the self-play search; priors are implicit (every
// child of a node has P = 1/K in double).  The rollouts consume the game's numpy-legacy stream
// exactly like np.random.rand(len(move_target)) per step, so the stream handed in by the host
// (np.random.get_state) can be handed back (np.random.set_state) afterwards.
#pragma once
#include "azg_tree.hpp"

namespace azg {

constexpr int MT_N = 624, MT_M = 397;

// MT19937 state of one game in LDS; the position lives in a wave-uniform register.
struct MtLds { unsigned mt[MT_N]; };

__device__ inline unsigned mt_temper(unsigned y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// The generator's block refresh by one wave.  Word k needs the OLD words k, k+1 and (k < 227)
// k+397, or (k >= 227) the NEW word k-227; word 623 needs the new words 0 and 396.  Chunks of 64
// words in ascending order, each chunk's reads finished before its writes, give exactly that.
__device__ inline void mt_twist(unsigned *mt, int lane)
{
    for (int c0 = 0; c0 < MT_N; c0 += WAVE) {
        const int k = c0 + lane;
        const bool on = k < MT_N;
        unsigned nv = 0;
        if (on) {
            const unsigned y = (mt[k] & 0x80000000u) | (mt[k + 1 == MT_N ? 0 : k + 1] & 0x7fffffffu);
            nv = mt[k < MT_N - MT_M ? k + MT_M : k - (MT_N - MT_M)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        wave_sync();
        if (on) mt[k] = nv;
        wave_sync();
    }
}

// Rows and columns spanned by the stones (game.py:43-46); wave-uniform.
struct BBox { int top, bottom, left, right; };

__device__ inline int wave_max_i32(int v) { return -wave_min_i32(-v); }

template <int CPL>
__device__ inline BBox stones_bbox(const u64 (*bb)[MAXW], int N, int S, int lane)
{
    int cmin = 0x7fff, cmax = -1, rmin = 0x7fff, rmax = -1;
#pragma unroll
    for (int i = 0; i < CPL; i++) {
        const int c = lane + WAVE * i;
        if (c < S && (bb_has(bb[0], c) || bb_has(bb[1], c))) {
            const int r = c / N, col = c - r * N;
            cmin = min(cmin, col); cmax = max(cmax, col);
            rmin = min(rmin, r); rmax = max(rmax, r);
        }
    }
    BBox b;
    b.top = wave_max_i32(rmax); b.bottom = wave_min_i32(rmin);
    b.left = wave_min_i32(cmin); b.right = wave_max_i32(cmax);
    return b;
}

// The cells of Board.get_move_target for a position with `nstones` stones: lane's cell
// lane + 64*i is a target when tgt[i]; mask[i] is the ballot of tgt[i]; returns their number.
template <int CPL>
__device__ inline int move_targets(const u64 (*bb)[MAXW], const BBox &box, int nstones, int N, int S, int lane,
                                   bool (&tgt)[CPL], u64 (&mask)[CPL])
{
    int top = 9, bottom = 5, left = 5, right = 9;                // game.py:40-41 (empty board)
    if (nstones > 0) {                                           // game.py:48-52, step = 2
        top = box.top <= N - 2 ? box.top + 2 : N;
        bottom = box.bottom >= 2 ? box.bottom - 2 : 0;
        left = box.left >= 2 ? box.left - 2 : 0;
        right = box.right <= N - 2 ? box.right + 2 : N;
    }
    int K = 0;
#pragma unroll
    for (int i = 0; i < CPL; i++) {
        const int c = lane + WAVE * i;
        const int r = c / N, col = c - r * N;
        tgt[i] = c < S && !bb_has(bb[0], c) && !bb_has(bb[1], c) && r >= bottom && r <= top && col >= left && col <= right;
        mask[i] = __ballot(tgt[i]);
        K += __popcll(mask[i]);
    }
    return K;
}

// rollout_policy_fn + max(..., key=itemgetter(1)) (mcts_pure.py:15-30,203-204): np.random.rand(K)
// gives the target cells, in ascending order, one double each (two generator words:
// (a >> 5) * 2^26 + (b >> 6), over 2^53); the first maximum is played.  pos: stream position.
template <int CPL>
__device__ inline int rollout_pick(MtLds *rs, int &pos, const bool (&tgt)[CPL], const u64 (&mask)[CPL], int K, int lane)
{
    unsigned a[CPL], b[CPL];
    int i0[CPL];
    int before = 0;
#pragma unroll
    for (int i = 0; i < CPL; i++) {
        const int rank = before + __popcll(mask[i] & ((1ULL << lane) - 1ULL));
        before += __popcll(mask[i]);
        i0[i] = pos + 2 * rank;
        a[i] = b[i] = 0;
        if (tgt[i]) {
            if (i0[i] < MT_N) a[i] = rs->mt[i0[i]];
            if (i0[i] + 1 < MT_N) b[i] = rs->mt[i0[i] + 1];
        }
    }
    const int end = pos + 2 * K;
    if (end > MT_N) {                                            // the draw runs past the block: refresh once
        wave_sync();
        mt_twist(rs->mt, lane);
#pragma unroll
        for (int i = 0; i < CPL; i++)
            if (tgt[i]) {
                if (i0[i] >= MT_N) a[i] = rs->mt[i0[i] - MT_N];
                if (i0[i] + 1 >= MT_N) b[i] = rs->mt[i0[i] + 1 - MT_N];
            }
        wave_sync();
        pos = end - MT_N;
    } else {
        pos = end;
    }
    double best = -1.0;
    int bcell = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < CPL; i++)
        if (tgt[i]) {
            const double u = ((double)(mt_temper(a[i]) >> 5) * 67108864.0 + (double)(mt_temper(b[i]) >> 6)) / 9007199254740992.0;
            if (u > best) { best = u; bcell = lane + WAVE * i; }
        }
    const double mx = wave_max_f64(best);
    return wave_min_i32(best == mx ? bcell : 0x7fffffff);
}

template <class GL>
__device__ inline int leaf_game_end(const DevState &st, const GL *gl, int nstones, int lane)
{
    if (st.NW == 1) return game_end_1w(gl->leaf_bb[0][0], gl->leaf_bb[1][0], st.N, st.S, st.n_in_row, nstones);
    return wave_game_end(gl->leaf_bb, st.N, st.S, st.n_in_row, nstones, lane);
}

// TreeNode.expand (mcts_pure.py:62-77) over zip(move_target, ones(K)/K).
template <int CPL, class GL>
__device__ inline bool wave_expand_targets(const DevState &st, const TreeView &t, GL *gl, const Descent &d, int g,
                                           int lane, const bool (&tgt)[CPL], const u64 (&mask)[CPL], int K)
{
    const int base = gl->used;
    if (base + K > gl->cap) {
        if (lane == 0) st.err[g] |= 1;
        return false;
    }
    const float p = (float)(1.0 / (double)K);                    // read-out only; the search recomputes 1/K in double
    int before = 0;
#pragma unroll
    for (int i = 0; i < CPL; i++) {
        if (tgt[i]) {
            const int ei = base + before + __popcll(mask[i] & ((1ULL << lane) - 1ULL));
            t.P[ei] = p; t.N[ei] = 0; t.Q[ei] = 0.0; t.child[ei] = -1; t.meta[ei] = meta_pack(lane + WAVE * i, 0);
        }
        before += __popcll(mask[i]);
    }
    wave_sync();                                                 // every lane has read gl->used
    if (lane == 0) {
        t.child[d.leaf] = base;
        t.meta[d.leaf] = meta_pack(d.depth == 1 ? gl->root_act : d.last, K);
        gl->used = base + K;
    }
    return true;
}

// MCTSPlayer.get_action of mcts_pure.py (:261-268) for every live slot: a fresh root, n_playout
// playouts (select / expand over move_target / random rollout to the end / backup), then the first
// most-visited root child (get_move, :214-231).  mt_io [B][625]: the slot's generator words and
// position, read at the start and written back at the end.  The tree stays in the arena (root
// statistics can be read out afterwards); the board is not touched.
template <int CPL>
__global__ __launch_bounds__(256) void pure_search_kernel(DevState st, int n_playout, unsigned *mt_io, int *moves_out)
{
    __shared__ MtLds s_mt[4];
    __shared__ GameLds s_gl[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave;
    if (g >= st.B) return;                                       // waves are independent: no workgroup barrier below
    if (st.status[g] != 0) {
        if (lane == 0) moves_out[g] = -1;
        return;
    }
    GameLds *gl = &s_gl[wave];
    MtLds *rs = &s_mt[wave];
    const int arena = st.cur[g];
    const TreeView tv = tree_view(st, g, arena);
    if (lane < 2 * MAXW) ((u64 *)gl->root_bb)[lane] = st.bb[(size_t)g * 2 * MAXW + lane];
    if (lane == 0) {
        gl->live = 1; gl->gid = g; gl->arena = arena; gl->big = st.big_of[g]; gl->cap = arena_cap(st, g);
        gl->root_ply = st.ply[g]; gl->root_last = st.last_move[g]; gl->first_player = st.first_player[g];
        // TreeNode(None, 1.0) (mcts_pure.py:139, :243)
        tv.P[0] = 1.0f; tv.N[0] = 0; tv.Q[0] = 0.0; tv.child[0] = -1; tv.meta[0] = meta_pack(0xFFFF, 0);
        gl->used = 1; gl->root_act = 0xFFFF;
    }
    for (int i = lane; i < MT_N; i += WAVE) rs->mt[i] = mt_io[(size_t)g * (MT_N + 1) + i];
    int pos = (int)mt_io[(size_t)g * (MT_N + 1) + MT_N];
    pos = min(max(pos, 0), MT_N);
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

    for (int it = 0; it < n_playout; it++) {
        const Descent d = wave_select<GameLds, true, CPL>(st, tv, gl, lane);
        int ply = d.ply;
        // below the root only the stone just played can have ended the game (see wave_last_move_end)
        int term = (d.depth > 1 && st.n_in_row >= 3 && st.n_in_row <= 8)
                       ? wave_last_move_end(gl->leaf_bb, st.N, st.S, st.n_in_row, ply, d.last, player_at(gl->first_player, ply - 1), lane)
                       : leaf_game_end(st, gl, ply, lane);
        BBox box = stones_bbox<CPL>(gl->leaf_bb, st.N, st.S, lane);
        bool tgt[CPL];
        u64 mask[CPL];
        int K = term ? 0 : move_targets<CPL>(gl->leaf_bb, box, ply, st.N, st.S, lane, tgt, mask);
        if (!term && K > 0) wave_expand_targets<CPL>(st, tv, gl, d, g, lane, tgt, mask, K);
        // _evaluate_rollout (mcts_pure.py:193-212) on the leaf board, in place
        const int player = player_at(gl->first_player, ply);
        while (!term) {
            if (K == 0) { term = 3; break; }                     // no target on a board that is not full: boards under 6x6 only
            const int move = rollout_pick<CPL>(rs, pos, tgt, mask, K, lane);
            if (lane == 0) {
                const int mover = player_at(gl->first_player, ply);
                gl->leaf_bb[mover - 1][move >> 6] |= 1ULL << (move & 63);
            }
            wave_sync();
            const int r = move / st.N, c = move - r * st.N;
            if (ply == 0) { box.top = box.bottom = r; box.left = box.right = c; }
            else { box.top = max(box.top, r); box.bottom = min(box.bottom, r); box.left = min(box.left, c); box.right = max(box.right, c); }
            ply++;
            term = (st.n_in_row >= 3 && st.n_in_row <= 8)
                       ? wave_last_move_end(gl->leaf_bb, st.N, st.S, st.n_in_row, ply, move, player_at(gl->first_player, ply - 1), lane)
                       : leaf_game_end(st, gl, ply, lane);
            if (!term) K = move_targets<CPL>(gl->leaf_bb, box, ply, st.N, st.S, lane, tgt, mask);
        }
        const double leaf_value = term == 3 ? 0.0 : (term == player ? 1.0 : -1.0);
        wave_backup(tv, gl, d.depth, leaf_value, lane);          // update_recursive(-leaf_value)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }

    // get_move: max over the root's children by visit count, first maximum in ascending order
    const int base = tv.child[0];
    const int K = base >= 0 ? meta_k(tv.meta[0]) : 0;
    int bn = -1, bi = 0x7fffffff;
    for (int i = lane; i < K; i += WAVE) {
        const int n = tv.N[base + i];
        if (n > bn) { bn = n; bi = i; }
    }
    const int mx = wave_max_i32(bn);
    const int ri = wave_min_i32(bn == mx ? bi : 0x7fffffff);
    if (lane == 0) {
        moves_out[g] = K > 0 ? meta_act(tv.meta[base + ri]) : -1;
        st.used[g] = gl->used;
        mt_io[(size_t)g * (MT_N + 1) + MT_N] = (unsigned)pos;
    }
    wave_sync();
    for (int i = lane; i < MT_N; i += WAVE) mt_io[(size_t)g * (MT_N + 1) + i] = rs->mt[i];
}

}  // namespace azg
