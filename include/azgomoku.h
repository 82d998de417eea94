/*
 * azgomoku.h -- C ABI of libazgomoku.so, the MI355X (gfx950) self-play engine.
 *
 * The reference (AbcEric/AlphaZero_Gomoku) is pure Python and has no FFI layer;
 * its boundary for this path is duck typing between train.py / game.py and
 * MCTSPlayer / PolicyValueNet.  Each entry point below names the reference
 * interface it stands in for (file:line relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success or a negative AZG_E* code; nothing
 *     throws across the ABI; azg_last_error() gives the text of the last failure
 *   - the caller owns every buffer it passes; the library owns all device memory
 *   - pointers documented "host or device" are copied with hipMemcpyDefault
 *   - one handle per GPU, one HIP stream per handle, not thread-safe
 *   - calls are synchronous on return unless documented "asynchronous"
 *   - cells are numbered move = row * N + col with row 0 at the bottom, as
 *     game.py:70-80 does; only square boards up to 16x16 are supported
 */
#ifndef AZGOMOKU_H
#define AZGOMOKU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZG_OK            0
#define AZG_EINVAL       -1   /* bad argument / unsupported configuration          */
#define AZG_EHIP         -2   /* a HIP runtime call failed                         */
#define AZG_ENOWEIGHTS   -3   /* net-driven search requested before azg_set_weights */
#define AZG_EARENA       -4   /* a game's tree outgrew its arena: arena_nodes < n_playout, or
                                 repeated searches without a move in between (raise arena_nodes) */
#define AZG_ESTATE       -5   /* call not valid in the current state               */
#define AZG_EMOVE        -6   /* illegal move (occupied cell / out of range)       */
#define AZG_ERANGE       -7   /* AZG_CNN_F16X3: an activation left the half-precision range */

/* where leaf priors and values come from */
#define AZG_PRIOR_NET        0   /* policy/value CNN (policy_value_net_pytorch.py:23-57) */
#define AZG_PRIOR_UNIFORM32  1   /* P = float(1)/float(K), v = 0 (test callback)         */
#define AZG_PRIOR_HASH       2   /* exact dyadic priors from a position hash (test)      */
#define AZG_PRIOR_PEAKED     3   /* P = 15/16 on the lowest empty cell, 1/1024 elsewhere,
                                    v = 0 (test: trees that keep growing under tree reuse) */

/* arithmetic of the CNN's 3x3 convolutions (both agree with a float32 reference to <1e-6) */
#define AZG_CNN_F16X3      0   /* split-f16 operands, three f16 MFMAs per product, f32 accumulate; boards 6x6 .. 16x16 */
#define AZG_CNN_F32        1   /* float32-input MFMA; boards 6x6, 8x8, 15x15                                           */

/* weight tensor layouts accepted by azg_set_weights */
#define AZG_LAYOUT_TORCH   0   /* state_dict order of policy_value_net_pytorch.py:31-41 */
#define AZG_LAYOUT_THEANO  1   /* 16-tensor list of policy_value_net_numpy.py:103-113   */

typedef struct azg_handle azg_handle;

typedef struct azg_config {
    int32_t board_n;       /* board width == height (game.py:15-16)                   */
    int32_t n_in_row;      /* game.py:21                                              */
    int32_t n_playout;     /* MCTS(n_playout) mcts_alphaZero.py:98                    */
    int32_t n_games;       /* concurrent game slots B                                 */
    double  c_puct;        /* mcts_alphaZero.py:98                                    */
    int32_t device;        /* HIP device ordinal                                      */
    int32_t prior_mode;    /* AZG_PRIOR_*                                             */
    int32_t arena_nodes;   /* expanded nodes an arena holds per game; 0 = 16*n_playout+64
                              (8x, 4x when all arenas would exceed 64 GB or a quarter of the free memory) */
    int32_t cnn_mode;      /* AZG_CNN_*                                               */
    int32_t big_arenas;    /* large tree arenas shared by the games whose re-used tree outgrows
                              arena_nodes; 0 = max(16, n_games / 4), at most n_games (fewer if they
                              would exceed 16 GB), < 0 = none */
} azg_config;

/* MCTSPlayer.__init__ / MCTS.__init__ (mcts_alphaZero.py:98-111,201-203) plus
 * Board.__init__ / init_board (game.py:14-23,60-68) for n_games boards. */
int  azg_create(const azg_config *cfg, azg_handle **out);
void azg_destroy(azg_handle *h);
const char *azg_last_error(const azg_handle *h);      /* h may be NULL: create errors */

/* PolicyValueNet.__init__(model_file) / load_state_dict
 * (policy_value_net_pytorch.py:76-78; numpy twin policy_value_net_numpy.py:87-90).
 * ptrs: 16 float32 tensors, host or device; repacked into the engine's own copy. */
int azg_set_weights(azg_handle *h, const float *const ptrs[16], int layout);

/* Board.init_board + MCTSPlayer.reset_player (game.py:60-68, mcts_alphaZero.py:208-209)
 * for every slot with mask[g] != 0 (mask NULL = all).  seeds[g] seeds that game's
 * numpy-legacy RandomState (np.random.seed, used by mcts_alphaZero.py:236-240);
 * game_ids[g] (may be NULL) is an opaque tag returned by azg_drain_samples. */
int azg_reset_games(azg_handle *h, const uint8_t *mask, const uint32_t *seeds,
                    const int64_t *game_ids);

/* Put slot `slot` into an arbitrary position (the board object a caller hands to
 * MCTSPlayer.get_action, mcts_alphaZero.py:212-216).  stones1/stones2: bit m of word
 * m/64 set when player 1/2 owns cell m (4 words each).  The search tree of the slot
 * is left untouched. */
int azg_set_position(azg_handle *h, int slot, const uint64_t stones1[4],
                     const uint64_t stones2[4], int last_move, int current_player);

/* MCTS.get_move_probs' playout loop (mcts_alphaZero.py:161-163) for every live
 * slot: n_playout sequential playouts per game, all games advancing together.
 * Asynchronous: returns after enqueueing; later calls on the handle order after it. */
int azg_search(azg_handle *h);

/* One playout of MCTS._playout (mcts_alphaZero.py:113-148) for slot `slot`, split at the call of
 * the policy function (:132) so that ANY policy_value_fn of the caller can supply priors and value
 * (mcts_alphaZero.py:98-103; human_play.py passes the numpy net's method).  azg_playout_begin
 * descends to a leaf and returns its position: stones of player 1 / 2 (bit m of word m/64), the
 * number of stones, the last move, the side to move and terminal = 0 (game goes on), 1 / 2
 * (winner) or -1 (tie).  azg_playout_end expands the leaf with priors[S] (float32 by cell, host or
 * device; entries of occupied cells are ignored; not renormalised, like policy_value_net_pytorch.py:113)
 * and backs `value` (for the side to move) up; for a terminal leaf both are ignored (priors may be
 * NULL) and the reference's +-1 / 0 is backed up.  n_playout pairs of calls equal one azg_search. */
int azg_playout_begin(azg_handle *h, int slot, uint64_t stones1[4], uint64_t stones2[4], int *ply,
                      int *last_move, int *current_player, int *terminal);
int azg_playout_end(azg_handle *h, int slot, const float *priors, double value);

/* Root statistics after azg_search (mcts_alphaZero.py:166-168): per slot and per
 * cell the child's visit count (0 for occupied cells), optionally Q and P, and the
 * root's own visit count.  Outputs host or device; any may be NULL. */
int azg_root_stats(azg_handle *h, int32_t *visits /*[B,S]*/, double *q /*[B,S]*/,
                   double *prior /*[B,S]*/, int32_t *root_visits /*[B]*/);

/* MCTSPlayer.get_action with is_selfplay=1 (mcts_alphaZero.py:216-242) followed by
 * the bookkeeping of Game.start_self_play (game.py:381-391) for every live slot:
 * pi = softmax(log(visits+1e-10)/temp); move ~ 0.75*pi + 0.25*Dirichlet(0.1) drawn
 * from the slot's RandomState; the (state, pi, player) sample is recorded; the move
 * is played; the tree is re-rooted at the move (update_with_move, :183-192); the
 * game-end test runs.  moves_out [B] (-1 for slots not live), pi_out [B,S]: host
 * pointers, may be NULL. */
int azg_sample_and_advance(azg_handle *h, double temp, int32_t *moves_out, double *pi_out);

/* Same bookkeeping with caller-chosen moves (host pointer, [B], -1 = leave slot
 * alone): Board.do_move + MCTS.update_with_move(move).  record != 0 also stores the
 * training sample with pi from the current root visits at `temp`. */
int azg_advance(azg_handle *h, const int32_t *moves, int record, double temp);

/* MCTS.update_with_move (mcts_alphaZero.py:183-192) alone, board untouched:
 * moves[g] >= 0 re-roots at that child (fresh root if it is no child), -1 discards
 * the tree (reset_player), -2 leaves the slot alone.  Host pointer [B]. */
int azg_update_with_move(azg_handle *h, const int32_t *moves);

/* Board.game_end / get_current_player for every slot (game.py:162-172):
 * ended[g] (0/1), winner[g] (1, 2 or -1), plies[g] = len(states).  Host pointers,
 * any may be NULL. */
int azg_status(azg_handle *h, uint8_t *ended, int8_t *winner, int32_t *plies);

/* Board.current_state for every slot (game.py:92-116): float32 [B,4,N,N], host or
 * device. */
int azg_states(azg_handle *h, float *out);

/* The (state, mcts_probs, winners_z) tuples Game.start_self_play returns
 * (game.py:393-416) for every finished, not yet drained game, slot order then ply
 * order.  states f32 [n,4,N,N], pi f64 [n,S], z f32 [n], game_id i64 [n], ply i32 [n]
 * (host or device, any may be NULL).  At most max_samples are written; *n_out gets
 * the count.  A game whose samples do not fit is left for the next call. */
int azg_drain_samples(azg_handle *h, int max_samples, float *states, double *pi, float *z,
                      int64_t *game_id, int32_t *ply, int *n_out);

/* Pipelined self-play: the loop of train.py:146 (collect_selfplay_data calling
 * Game.start_self_play, game.py:360-416) for all slots, with no host round trip between
 * a move step and the next search.  Game number k of the run (k = 0, 1, ...) has id
 * first_id + k*stride and plays exactly the game
 *   np.random.seed(seed0 + id); game.start_self_play(player, temp)
 * would play with the same weights; games are handed to slots in the order slots free
 * up.  total < 0: no limit, else no game with k >= total is started.
 *
 * azg_selfplay_begin resets every slot, starts games k = 0..n_games-1 and enqueues
 * the first search.  azg_selfplay_step finishes one move for every live slot
 * (azg_sample_and_advance semantics), restarts finished slots on the device and
 * enqueues the next search before it waits for anything; it reports how many games
 * ended in this step and how many samples of EARLIER finished games are now ready.
 * azg_selfplay_fetch copies those samples (layout as azg_drain_samples; host or device
 * pointers, any may be NULL) and must be called before the next step: a step taken while
 * ready samples are unfetched fails with AZG_ESTATE (a fetch with null pointers discards
 * them).  azg_selfplay_end waits for the work in flight, makes the samples of the last finished
 * games ready for a final fetch and returns the handle to the synchronous calls (which fail
 * with AZG_ESTATE in between). */
int azg_selfplay_begin(azg_handle *h, double temp, uint32_t seed0, int64_t first_id,
                       int64_t stride, int64_t total);
int azg_selfplay_step(azg_handle *h, int *games_finished, int *samples_ready);
int azg_selfplay_fetch(azg_handle *h, float *states, double *pi, float *z,
                       int64_t *game_id, int32_t *ply);
int azg_selfplay_end(azg_handle *h, int *samples_ready);

/* The ready samples in packed form, for the exchange between GPUs: instead of the four feature
 * planes (4*S floats) the position they are computed from, stones u64 [n,2,4] (bit m of word m/64
 * of colour c) and last_move i32 [n]; pi, z, game_id, ply as azg_selfplay_fetch.  Either fetch may
 * be called (or both) before the next step.  azg_unpack_states turns packed positions of self-play
 * games (stone 1 opens) back into Board.current_state planes (game.py:92-116), f32 [n,4,N,N]; all
 * pointers host or device. */
int azg_selfplay_fetch_packed(azg_handle *h, uint64_t *stones, int32_t *last_move, double *pi,
                              float *z, int64_t *game_id, int32_t *ply);
int azg_unpack_states(azg_handle *h, int n, const uint64_t *stones, const int32_t *last_move,
                      const int32_t *ply, float *states);

/* The evaluation opponent of train.py:249,264: mcts_pure.MCTSPlayer.get_action
 * (mcts_pure.py:261-268) for every live slot, i.e. MCTS.get_move (mcts_pure.py:214-231) from a
 * fresh root: n_playout playouts, each expanding the leaf over Board.get_move_target's cells
 * (game.py:35-58) with prior 1/K and finishing the game by random moves (_evaluate_rollout,
 * mcts_pure.py:193-212), then the first most-visited root child.  The rollouts draw
 * np.random.rand(len(move_target)) per step from the slot's numpy-legacy MT19937 stream:
 * mt_state [B][625] (host) holds, per slot, the 624 key words and the position as
 * np.random.get_state() returns them, and is updated in place, so that
 * np.random.set_state() continues the stream where the reference would be.  moves_out [B]
 * (host): the chosen cell, -1 for slots that are not live.  The board is not changed; the
 * tree stays in the slot (azg_root_stats shows the visit counts) until the next search or
 * azg_update_with_move.  Needs arena_nodes >= n_playout (AZG_EARENA otherwise). */
int azg_pure_search(azg_handle *h, int n_playout, uint32_t *mt_state, int32_t *moves_out);

/* PolicyValueNet.policy_value (policy_value_net_pytorch.py:80-94): states f32
 * [n,4,N,N] -> probs f32 [n,S] (exp of the log-softmax), values f32 [n].  Host or
 * device pointers. */
int azg_eval_batch(azg_handle *h, const float *states, int n, float *probs, float *values);

/* Stateless batched Board.do_move + game_end + current_state (game.py:92-127,
 * 162-169) for n independent boards: stones [n,2,4] words and ply/last_move [n] are
 * updated in place, moves [n] are played by `player[i]` (1 or 2; the side to move
 * afterwards is the other one), ended/winner [n] and planes f32 [n,4,N,N] are
 * produced.  All host pointers; planes may be NULL.  Illegal moves give AZG_EMOVE. */
int azg_board_step_batch(azg_handle *h, int n, uint64_t *stones, int32_t *ply,
                         int32_t *last_move, const int32_t *moves, const int32_t *player,
                         uint8_t *ended, int8_t *winner, float *planes);

/* azg_board_step_batch for callers that may not allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments,
 * results and errors as azg_board_step_batch, but staging, the illegal-move flag and the outputs
 * live in a fixed 4.3 MB array that every GPU running this library holds until the process ends,
 * and the call makes no hipMalloc / hipFree of any kind.  Like azg_board_step_batch it takes HOST
 * pointers, runs on the handle's own stream and is synchronous (it ends with a stream
 * synchronisation), so it cannot be recorded into a caller's HIP graph.  At most 1024 boards per
 * call (AZG_EINVAL beyond that).  All calls on one GPU share that array: calls that overlap in
 * time on one GPU race, whatever handle they use. */
int azg_board_step_batch_static(azg_handle *h, int n, uint64_t *stones, int32_t *ply,
                                int32_t *last_move, const int32_t *moves, const int32_t *player,
                                uint8_t *ended, int8_t *winner, float *planes);

/* TrainPipeline.get_equi_data (train.py:116-135): the 8 rotated / mirrored images of n
 * (state, pi, z) tuples (layout of azg_drain_samples), tuple i giving outputs 8i..8i+7
 * in the reference's order (rotation by 90, its left-right mirror, rotation by 180, ...).
 * Outputs hold 8n tuples; all pointers host or device.  Not valid while a pipelined
 * self-play search is in flight on the handle (it would wait for it). */
int azg_equi_data(azg_handle *h, int n, const float *states, const double *pi, const float *z,
                  float *states_out, double *pi_out, float *z_out);

/* PolicyValueNet.train_step (policy_value_net_pytorch.py:117-148) on a mini-batch of n tuples, in
 * float32, as three calls so that a data-parallel caller can average the gradients in between:
 *   azg_train_backward  forward pass (Net.forward, :43-57), loss = mse_loss(value, winners) -
 *                       mean(sum(mcts_probs * log p, 1)), its gradient with respect to the 16 parameter
 *                       tensors as ONE flat vector in state_dict order (azg_train_param_count floats),
 *                       and the two numbers train_step returns: loss and entropy = -mean(sum(p log p, 1))
 *                       (host pointers, may be NULL);
 *   azg_train_adam      torch.optim.Adam's update (:65-67: weight decay added to the gradient) of the
 *                       parameters in place from that vector; adam_m / adam_v are the caller's moment
 *                       vectors (zero before step 1), step counts from 1;
 *   azg_train_forward   the forward pass alone: probs = exp(log_softmax) [n,S] and values [n], what
 *                       PolicyValueNet.policy_value (:80-94) gives for the KL test of train.py:187-201;
 *                       with old_probs (device f32 [n,S], else NULL) also that test's number in *kl:
 *                       mean_n sum_a old * (log(old + 1e-10) - log(new + 1e-10)), summed in double
 *                       (train.py:189-193);
 * states f32 [n,4,N,N], mcts_probs f32 [n,S], winners f32 [n], params: the 16 tensors of
 * policy_value_net_pytorch.py:31-41 in PyTorch's layout; ALL of these, grads, adam_m, adam_v, probs and
 * values are DEVICE pointers.  The calls run on a stream of their own (not behind a search in flight)
 * and are synchronous on return. */
int azg_train_param_count(azg_handle *h);
int azg_train_forward(azg_handle *h, int n, const float *states, const float *const params[16],
                      float *probs, float *values, const float *old_probs, double *kl);
int azg_train_backward(azg_handle *h, int n, const float *states, const float *mcts_probs,
                       const float *winners, const float *const params[16], float *grads,
                       float *loss, float *entropy);
int azg_train_adam(azg_handle *h, float *const params[16], const float *grads, float *adam_m,
                   float *adam_v, int64_t step, double lr, double beta1, double beta2, double eps,
                   double weight_decay);
int azg_train_step(azg_handle *h, int n, const float *states, const float *mcts_probs,
                   const float *winners, float *const params[16], float *grads, float *adam_m,
                   float *adam_v, int64_t step, double lr, double beta1, double beta2, double eps,
                   double weight_decay, float *loss, float *entropy);
/* One epoch of TrainPipeline.policy_update's loop (train.py:176-201) in one call, one wait: azg_train_step on
 * the n training positions, then azg_train_forward of the n_all positions of the KL test with the updated
 * parameters (probs [n_all,S], values [n_all], *kl against old_probs). */
int azg_train_epoch(azg_handle *h, int n, const float *states, const float *mcts_probs,
                    const float *winners, float *const params[16], float *grads, float *adam_m,
                    float *adam_v, int64_t step, double lr, double beta1, double beta2, double eps,
                    double weight_decay, float *loss, float *entropy, int n_all,
                    const float *states_all, float *probs, float *values, const float *old_probs,
                    double *kl);

/* Timing of the search of a move (MCTS.get_move_probs, mcts_alphaZero.py:154-181), measured with HIP
 * events on the handle's stream: number of searches since the last call, summed milliseconds, and
 * summed leaf evaluations (playouts that reached the CNN).  Resets the counters.
 * A search is azg_search_parts() kernel launches back to back, each with its share of the move's
 * playouts (the trees do not depend on the number; AZG_SEARCH_PARTS in the environment of azg_create
 * sets it, azg_set_search_parts overrides it (0 = the default again): default 8 for net-driven boards up
 * to 8x8, 2 for larger ones, 1 for the synthetic priors.  Kernels of other streams -- a training
 * step -- get onto the GPU between two parts, not beside a resident one.) */
int azg_profile(azg_handle *h, int64_t *launches, double *ms, int64_t *playouts);
int azg_search_parts(azg_handle *h);
int azg_set_search_parts(azg_handle *h, int parts);

/* The reference's tree is unbounded under tree reuse (mcts_alphaZero.py:183-192 keeps the whole
 * subtree of the move played).  A game whose tree may outgrow its arena during the next search is
 * moved, before that search, into one of big_arenas large arenas that hold any tree of the
 * configuration; if none is free its tree is dropped (the game goes on from a fresh root, without
 * the carried-over statistics).  Counts since azg_create: games moved, trees dropped. */
int azg_arena_stats(azg_handle *h, int64_t *moved_to_large, int64_t *trees_dropped);

/* MCTS.update_with_move (mcts_alphaZero.py:183-192) keeps the subtree of the move played.  Counts
 * since azg_create: re-roots that left the kept subtree where it was (the child's record moved to
 * the root's place), and re-roots that compacted it into the game's other arena because the arena
 * had no room for another search behind what it held.  Both give the same tree. */
int azg_reroot_stats(azg_handle *h, int64_t *in_place, int64_t *compacted);

/* Host-side cost of the exploration noise (the numpy-legacy Dirichlet and uniform draws of
 * mcts_alphaZero.py:236-240, generated on host threads while a search runs): summed wall
 * milliseconds and number of move steps since the last call.  Resets the counters. */
int azg_host_timing(azg_handle *h, double *noise_ms, int64_t *noise_steps);

/* Diagnostics of the per-CU matrix gate (no counterpart in the reference).  Two search workgroups
 * share a compute unit; a word per CU in a device-wide table makes their dense convolution phases
 * alternate.  The gate is advisory (bounded wait, results never depend on it); azg_create and
 * azg_selfplay_begin clear the table.  azg_gate_stats: waits that ran out since the process started
 * on this GPU (a number > 0 means launches ran part of their time without the alternation: slower,
 * not wrong), and the table words that are set right now.  azg_gate_poke stores `value` in every
 * word (tests: a table left dirty by a launch that died). */
int azg_gate_stats(azg_handle *h, int64_t *giveups, int64_t *words_set);
int azg_gate_poke(azg_handle *h, int value);

/* The handle's HIP stream (hipStream_t), for callers that record their own events. */
void *azg_stream(azg_handle *h);

#ifdef __cplusplus
}
#endif
#endif
