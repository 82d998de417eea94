/*
 * az_oracle.h -- CPU restatement of the AlphaZero-Gomoku self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * engine and the "port" CPU baseline timed by bench.py.  Nothing in the
 * product package may include, link, load or call it.
 *
 * Every function restates a piece of the reference (AbcEric/AlphaZero_Gomoku)
 * and cites the file:line it follows.  Parity of this restatement is pinned by
 * the golden vectors under tests/golden/, which were produced by importing the
 * reference's own Python modules (tools/gen_golden.py).
 */
#ifndef AZ_ORACLE_H
#define AZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZO_MAX_CELLS 256

/* ---- board (game.py:11-172) ------------------------------------------- */
typedef struct azo_board {
    int width, height, n_in_row;
    int cell[AZO_MAX_CELLS];   /* 0 = empty, 1 / 2 = player (states dict)   */
    int avail[AZO_MAX_CELLS];  /* ascending list of empty cells (availables) */
    int n_avail;
    int current_player;        /* 1 or 2                                     */
    int last_move;             /* -1 on an empty board                       */
    int n_stones;              /* len(states)                                */
} azo_board;

int  azo_board_init(azo_board *b, int width, int height, int n_in_row, int start_player);
int  azo_board_do_move(azo_board *b, int move);
void azo_board_has_a_winner(const azo_board *b, int *win, int *winner);
void azo_board_game_end(const azo_board *b, int *end, int *winner);
/* out: 4*width*height doubles, layout [plane][row][col] after the row flip. */
void azo_board_current_state(const azo_board *b, double *out);

/* ---- policy/value providers ------------------------------------------- */
/* probs: one entry per CELL (only entries of empty cells are consumed, as the
 * reference zips availables with probs[availables]); value: for side to move. */
typedef void (*azo_policy_fn)(const azo_board *b, double *probs, double *value, void *ctx);

enum { AZO_POLICY_UNIFORM32 = 1, AZO_POLICY_HASH = 2, AZO_POLICY_NET = 3, AZO_POLICY_CALLBACK = 4 };

/* 16 weight tensors in the PyTorch layout of policy_value_net_pytorch.py:31-41
 * (conv weights [out,in,kh,kw], linear weights [out,in]). */
typedef struct azo_net {
    int width, height;
    const float *p[16];
    int precise;   /* 1: accumulate in double (numpy-net style); 0: float    */
} azo_net;

/* probs[S] (softmax over all cells), value; computed from a (4,W,H) state. */
void azo_net_forward(const azo_net *net, const double *state, double *probs, double *value);

void azo_policy_uniform32(const azo_board *b, double *probs, double *value, void *ctx);
void azo_policy_hash(const azo_board *b, double *probs, double *value, void *ctx);
void azo_policy_net(const azo_board *b, double *probs, double *value, void *ctx); /* ctx = azo_net* */

uint64_t azo_state_key(const azo_board *b);

/* ---- numpy legacy RandomState (MT19937) -------------------------------- */
typedef struct azo_rng { uint32_t mt[624]; int pos; } azo_rng;
void   azo_rng_seed(azo_rng *r, uint32_t seed);
double azo_rng_double(azo_rng *r);
double azo_rng_standard_gamma(azo_rng *r, double shape);
void   azo_rng_dirichlet(azo_rng *r, double alpha, int k, double *out);

/* ---- MCTS (mcts_alphaZero.py) ------------------------------------------ */
typedef struct azo_mcts azo_mcts;
azo_mcts *azo_mcts_new(azo_policy_fn fn, void *ctx, double c_puct, int n_playout);
void      azo_mcts_free(azo_mcts *m);
/* runs n_playout playouts from `state` (copied per playout); fills acts/visits/
 * q/prior of the root's children (ascending action order) and probs (softmax of
 * 1/temp*log(visits+1e-10)).  Returns the number of root children. */
int  azo_mcts_get_move_probs(azo_mcts *m, const azo_board *state, double temp,
                             int *acts, int *visits, double *q, double *prior, double *probs);
void azo_mcts_update_with_move(azo_mcts *m, int last_move);
int  azo_mcts_root_children(const azo_mcts *m, int *acts, int *visits, double *q, double *prior);
int  azo_mcts_root_visits(const azo_mcts *m);
long azo_mcts_node_count(const azo_mcts *m);

/* ---- pure (rollout) MCTS opponent (mcts_pure.py, game.py:25-58) ---------- */
int azo_board_move_target(const azo_board *b, int *out);
/* One MCTSPlayer.get_action of mcts_pure.py from a fresh root; the rollouts draw from rng
 * (the global numpy stream in the reference).  Returns the move. */
int azo_pure_get_move(const azo_board *state, double c_puct, int n_playout, azo_rng *rng,
                      int *n_children, int *acts, int *visits, double *q, int *root_visits);

/* ---- self-play (MCTSPlayer.get_action + Game.start_self_play) ----------- */
typedef struct azo_game_record {
    int n_plies;
    int winner;                              /* 1, 2 or -1 (tie)             */
    int moves[AZO_MAX_CELLS];
    int players[AZO_MAX_CELLS];              /* side to move at each ply     */
    double z[AZO_MAX_CELLS];
    double *pi;                              /* [n_plies][S]  (caller buffer) */
    int *visits;                             /* [n_plies][S]  (caller buffer; by cell, 0 if occupied) */
    double *states;                          /* [n_plies][4*S] or NULL       */
} azo_game_record;

/* One self-play game: np.random.seed(seed) semantics, Dirichlet(0.1) mix
 * 0.75/0.25, tree reuse.  rec->pi / rec->visits must hold S*S entries. */
int azo_self_play(int width, int height, int n_in_row, azo_policy_fn fn, void *ctx,
                  double c_puct, int n_playout, double temp, uint32_t seed,
                  azo_game_record *rec);

/* Many games, one per OpenMP thread (CPU baseline).  Returns total plies. */
long azo_self_play_many(int width, int height, int n_in_row, const azo_net *net,
                        double c_puct, int n_playout, double temp, uint32_t seed0,
                        int n_games, int n_threads, int *plies_out, int *winners_out);

#ifdef __cplusplus
}
#endif
#endif
