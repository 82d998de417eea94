/*
 * az_oracle.c -- CPU restatement of the AlphaZero-Gomoku self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see az_oracle.h).  Plain C, no dependency on the
 * product.  All floating-point tree arithmetic is IEEE double in the
 * reference's operation order; build with -ffp-contract=off.
 */
#include "az_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================= */
/* Board -- game.py:11-172                                                  */
/* ======================================================================= */

/* game.py:60-68 init_board: players [1,2], availables = range(W*H), no stones. */
int azo_board_init(azo_board *b, int width, int height, int n_in_row, int start_player)
{
    if (width < n_in_row || height < n_in_row) return -1;       /* game.py:61-62 raises */
    if (width * height > AZO_MAX_CELLS) return -2;
    memset(b, 0, sizeof(*b));
    b->width = width; b->height = height; b->n_in_row = n_in_row;
    b->current_player = start_player == 0 ? 1 : 2;              /* players[start_player] */
    b->n_avail = width * height;
    for (int i = 0; i < b->n_avail; i++) b->avail[i] = i;
    b->last_move = -1;
    b->n_stones = 0;
    return 0;
}

/* game.py:119-127 do_move: states[move]=cur; availables.remove(move); swap; last_move. */
int azo_board_do_move(azo_board *b, int move)
{
    int idx = -1;
    for (int i = 0; i < b->n_avail; i++) if (b->avail[i] == move) { idx = i; break; }
    if (idx < 0) return -1;                                     /* list.remove -> ValueError */
    b->cell[move] = b->current_player;
    for (int i = idx; i + 1 < b->n_avail; i++) b->avail[i] = b->avail[i + 1];
    b->n_avail--;
    b->n_stones++;
    b->current_player = (b->current_player == 2) ? 1 : 2;
    b->last_move = move;
    return 0;
}

/* A ray of n cells starting at m with the given stride is won when every cell
 * holds the same player (game.py:145 `len(set(states.get(i,-1) ...)) == 1`).
 * The anchor cell m is occupied, so "all equal" means all equal to its owner. */
static int ray_same(const azo_board *b, int m, int stride, int n)
{
    int p = b->cell[m];
    for (int j = 1; j < n; j++) {
        int i = m + j * stride;
        if (i < 0 || i >= b->width * b->height) return 0;      /* states.get(i,-1) -> -1 */
        if (b->cell[i] != p) return 0;
    }
    return 1;
}

/* game.py:129-160 has_a_winner */
void azo_board_has_a_winner(const azo_board *b, int *win, int *winner)
{
    const int width = b->width, height = b->height, n = b->n_in_row;
    *win = 0; *winner = -1;
    if (b->n_stones < n + 2) return;                            /* game.py:136-137 */
    for (int m = 0; m < width * height; m++) {                  /* moved, ascending */
        if (!b->cell[m]) continue;
        int h = m / width, w = m % width;
        int player = b->cell[m];
        if (w <= width - n && ray_same(b, m, 1, n)) { *win = 1; *winner = player; return; }
        if (h <= height - n && ray_same(b, m, width, n)) { *win = 1; *winner = player; return; }
        if (w <= width - n && h <= height - n && ray_same(b, m, width + 1, n)) { *win = 1; *winner = player; return; }
        if (w >= n - 1 && h <= height - n && ray_same(b, m, width - 1, n)) { *win = 1; *winner = player; return; }
    }
}

/* game.py:162-169 game_end */
void azo_board_game_end(const azo_board *b, int *end, int *winner)
{
    int win, who;
    azo_board_has_a_winner(b, &win, &who);
    if (win) { *end = 1; *winner = who; return; }
    if (b->n_avail == 0) { *end = 1; *winner = -1; return; }
    *end = 0; *winner = -1;
}

/* game.py:92-116 current_state: planes (own, opponent, last move, colour-to-play),
 * indexed [move // width, move % height], rows reversed on return. */
void azo_board_current_state(const azo_board *b, double *out)
{
    const int W = b->width, H = b->height;
    const int plane = W * H;
    memset(out, 0, sizeof(double) * 4 * plane);
    if (b->n_stones > 0) {
        for (int m = 0; m < plane; m++) {
            if (!b->cell[m]) continue;
            int r = m / W, c = m % H;
            int rr = W - 1 - r;                                 /* [:, ::-1, :] */
            if (b->cell[m] == b->current_player) out[0 * plane + rr * H + c] = 1.0;
            else                                  out[1 * plane + rr * H + c] = 1.0;
        }
        int r = b->last_move / W, c = b->last_move % H;
        out[2 * plane + (W - 1 - r) * H + c] = 1.0;
    }
    if (b->n_stones % 2 == 0)
        for (int i = 0; i < plane; i++) out[3 * plane + i] = 1.0;
}

/* ======================================================================= */
/* Synthetic exact policy providers (test callbacks)                        */
/* ======================================================================= */

static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* Position key over the two stone sets (bit m of word m/64) and side to move. */
uint64_t azo_state_key(const azo_board *b)
{
    const int S = b->width * b->height, W = (S + 63) / 64;
    uint64_t k = 0x9E3779B97F4A7C15ULL ^ (uint64_t)b->current_player;
    for (int c = 1; c <= 2; c++)
        for (int w = 0; w < W; w++) {
            uint64_t word = 0;
            for (int bit = 0; bit < 64 && w * 64 + bit < S; bit++)
                if (b->cell[w * 64 + bit] == c) word |= 1ULL << bit;
            k = mix64(k ^ word);
        }
    return k;
}

/* Uniform prior 1/K computed in float (IEEE division) then widened; value 0. */
void azo_policy_uniform32(const azo_board *b, double *probs, double *value, void *ctx)
{
    (void)ctx;
    const int S = b->width * b->height;
    float p = b->n_avail > 0 ? 1.0f / (float)b->n_avail : 0.0f;
    for (int i = 0; i < S; i++) probs[i] = (double)p;
    *value = 0.0;
}

/* Hash priors: exact dyadic rationals in (0, 2^-5]; value in [-1, 1). */
void azo_policy_hash(const azo_board *b, double *probs, double *value, void *ctx)
{
    (void)ctx;
    const int S = b->width * b->height;
    uint64_t k = azo_state_key(b);
    for (int a = 0; a < S; a++) {
        uint64_t r = (mix64(k + 0x9E3779B97F4A7C15ULL * (uint64_t)(a + 1)) >> 32) & 0xFFF;
        probs[a] = (double)((float)(r + 1) / 131072.0f);
    }
    uint64_t rv = (mix64(k ^ 0xD1B54A32D192ED03ULL) >> 32) & 0xFFF;
    *value = ((double)rv - 2048.0) / 2048.0;
}

/* ======================================================================= */
/* Policy/value CNN -- policy_value_net_pytorch.py:23-57 (Net.forward),     */
/* numpy twin policy_value_net_numpy.py:92-115                              */
/* ======================================================================= */

#define DEF_CONV3(NAME, T)                                                         \
static void NAME(const T *in, int cin, int cout, int W, int H,                     \
                 const float *w, const float *bias, T *out)                        \
{                                                                                  \
    /* in: [cin][W+2][H+2] zero padded; out: [cout][W+2][H+2] zero padded, ReLU */ \
    const int PW = W + 2, PH = H + 2;                                              \
    T acc[AZO_MAX_CELLS];                                                          \
    for (int co = 0; co < cout; co++) {                                            \
        for (int i = 0; i < W * H; i++) acc[i] = (T)bias[co];                      \
        for (int ci = 0; ci < cin; ci++)                                           \
            for (int ky = 0; ky < 3; ky++)                                         \
                for (int kx = 0; kx < 3; kx++) {                                   \
                    const T wv = (T)w[((co * cin + ci) * 3 + ky) * 3 + kx];        \
                    const T *src = in + (ci * PW + ky) * PH + kx;                  \
                    for (int y = 0; y < W; y++)                                    \
                        for (int x = 0; x < H; x++)                                \
                            acc[y * H + x] += wv * src[y * PH + x];                \
                }                                                                  \
        T *dst = out + co * PW * PH;                                               \
        memset(dst, 0, sizeof(T) * PW * PH);                                       \
        for (int y = 0; y < W; y++)                                                \
            for (int x = 0; x < H; x++) {                                          \
                T v = acc[y * H + x];                                              \
                dst[(y + 1) * PH + x + 1] = v > 0 ? v : 0;                         \
            }                                                                      \
    }                                                                              \
}

DEF_CONV3(conv3_d, double)
DEF_CONV3(conv3_f, float)

#define DEF_HEADS(NAME, T, CONV, EXPF, LOGF, TANHF)                                \
static void NAME(const azo_net *net, const double *state, double *probs, double *value) \
{                                                                                  \
    const int W = net->width, H = net->height, S = W * H, PW = W + 2, PH = H + 2;  \
    const int PS = PW * PH;                                                        \
    T *x0 = (T *)calloc((size_t)(4 + 32 + 64 + 128) * PS, sizeof(T));              \
    T *x1 = x0 + 4 * PS, *x2 = x1 + 32 * PS, *x3 = x2 + 64 * PS;                   \
    for (int c = 0; c < 4; c++)                                                    \
        for (int y = 0; y < W; y++)                                                \
            for (int x = 0; x < H; x++)                                            \
                x0[(c * PW + y + 1) * PH + x + 1] = (T)state[(c * W + y) * H + x]; \
    CONV(x0, 4, 32, W, H, net->p[0], net->p[1], x1);                               \
    CONV(x1, 32, 64, W, H, net->p[2], net->p[3], x2);                              \
    CONV(x2, 64, 128, W, H, net->p[4], net->p[5], x3);                             \
    /* 1x1 heads, ReLU, flattened [c][y][x] (view(-1, C*W*H)) */                   \
    T pa[4 * AZO_MAX_CELLS], va[2 * AZO_MAX_CELLS];                                \
    for (int c = 0; c < 4; c++)                                                    \
        for (int y = 0; y < W; y++)                                                \
            for (int x = 0; x < H; x++) {                                          \
                T s = (T)net->p[7][c];                                             \
                for (int ci = 0; ci < 128; ci++)                                   \
                    s += (T)net->p[6][c * 128 + ci] * x3[(ci * PW + y + 1) * PH + x + 1]; \
                pa[c * S + y * H + x] = s > 0 ? s : 0;                             \
            }                                                                      \
    for (int c = 0; c < 2; c++)                                                    \
        for (int y = 0; y < W; y++)                                                \
            for (int x = 0; x < H; x++) {                                          \
                T s = (T)net->p[11][c];                                            \
                for (int ci = 0; ci < 128; ci++)                                   \
                    s += (T)net->p[10][c * 128 + ci] * x3[(ci * PW + y + 1) * PH + x + 1]; \
                va[c * S + y * H + x] = s > 0 ? s : 0;                             \
            }                                                                      \
    /* act_fc1 + log_softmax + exp (pytorch.py:51, :112) */                        \
    T logit[AZO_MAX_CELLS];                                                        \
    T mx = 0;                                                                      \
    for (int o = 0; o < S; o++) {                                                  \
        T s = (T)net->p[9][o];                                                     \
        const float *wr = net->p[8] + (size_t)o * 4 * S;                           \
        for (int i = 0; i < 4 * S; i++) s += (T)wr[i] * pa[i];                     \
        logit[o] = s;                                                              \
        if (o == 0 || s > mx) mx = s;                                              \
    }                                                                              \
    T sum = 0;                                                                     \
    for (int o = 0; o < S; o++) sum += EXPF(logit[o] - mx);                        \
    T lse = LOGF(sum);                                                             \
    for (int o = 0; o < S; o++) probs[o] = (double)EXPF((logit[o] - mx) - lse);    \
    /* val_fc1 + ReLU, val_fc2 + tanh (pytorch.py:53-56) */                        \
    T hid[64];                                                                     \
    for (int o = 0; o < 64; o++) {                                                 \
        T s = (T)net->p[13][o];                                                    \
        const float *wr = net->p[12] + (size_t)o * 2 * S;                          \
        for (int i = 0; i < 2 * S; i++) s += (T)wr[i] * va[i];                     \
        hid[o] = s > 0 ? s : 0;                                                    \
    }                                                                              \
    T v = (T)net->p[15][0];                                                        \
    for (int i = 0; i < 64; i++) v += (T)net->p[14][i] * hid[i];                   \
    *value = (double)TANHF(v);                                                     \
    free(x0);                                                                      \
}

DEF_HEADS(net_forward_d, double, conv3_d, exp, log, tanh)
DEF_HEADS(net_forward_f, float, conv3_f, expf, logf, tanhf)

void azo_net_forward(const azo_net *net, const double *state, double *probs, double *value)
{
    if (net->precise) net_forward_d(net, state, probs, value);
    else              net_forward_f(net, state, probs, value);
}

/* policy_value_fn (pytorch.py:96-115): the net sees current_state(); the
 * probabilities come back in float32 and are widened once to double. */
void azo_policy_net(const azo_board *b, double *probs, double *value, void *ctx)
{
    const azo_net *net = (const azo_net *)ctx;
    double state[4 * AZO_MAX_CELLS];
    azo_board_current_state(b, state);
    azo_net_forward(net, state, probs, value);
    const int S = b->width * b->height;
    for (int i = 0; i < S; i++) probs[i] = (double)(float)probs[i];
    *value = (double)(float)*value;
}

/* ======================================================================= */
/* numpy legacy RandomState: MT19937 + legacy gamma / dirichlet             */
/* (third-party: numpy/random/_mt19937.pyx, legacy-distributions.c; used by */
/*  mcts_alphaZero.py:236-240 through np.random.dirichlet / np.random.choice) */
/* ======================================================================= */

void azo_rng_seed(azo_rng *r, uint32_t seed)
{
    r->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        r->mt[i] = 1812433253U * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->pos = 624;
}

static uint32_t rng_u32(azo_rng *r)
{
    if (r->pos == 624) {
        uint32_t *mt = r->mt;
        int i;
        for (i = 0; i < 624 - 397; i++) {
            uint32_t y = (mt[i] & 0x80000000U) | (mt[i + 1] & 0x7fffffffU);
            mt[i] = mt[i + 397] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        for (; i < 623; i++) {
            uint32_t y = (mt[i] & 0x80000000U) | (mt[i + 1] & 0x7fffffffU);
            mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        r->pos = 0;
    }
    uint32_t y = r->mt[r->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

double azo_rng_double(azo_rng *r)
{
    uint32_t a = rng_u32(r) >> 5, b = rng_u32(r) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

static double rng_std_exponential(azo_rng *r) { return -log(1.0 - azo_rng_double(r)); }

/* legacy_gauss: polar Box-Muller with one cached value (needed for shape > 1) */
typedef struct { int has; double g; } gauss_cache;

double azo_rng_standard_gamma(azo_rng *r, double shape)
{
    if (shape == 1.0) return rng_std_exponential(r);
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
        for (;;) {
            double U = azo_rng_double(r);
            double V = rng_std_exponential(r);
            if (U <= 1.0 - shape) {
                double X = pow(U, 1.0 / shape);
                if (X <= V) return X;
            } else {
                double Y = -log((1.0 - U) / shape);
                double X = pow(1.0 - shape + shape * Y, 1.0 / shape);
                if (X <= (V + Y)) return X;
            }
        }
    }
    /* shape > 1 (Marsaglia-Tsang) is never reached by the self-play path
     * (alpha = 0.1); not restated. */
    return NAN;
}

void azo_rng_dirichlet(azo_rng *r, double alpha, int k, double *out)
{
    double acc = 0.0;
    for (int i = 0; i < k; i++) { out[i] = azo_rng_standard_gamma(r, alpha); acc += out[i]; }
    double inv = 1.0 / acc;
    for (int i = 0; i < k; i++) out[i] = out[i] * inv;
}

/* ======================================================================= */
/* MCTS -- mcts_alphaZero.py:24-192                                         */
/* ======================================================================= */

typedef struct node {
    int parent;        /* -1 for the root (TreeNode._parent)                */
    int first_child;   /* -1 while a leaf (_children == {})                 */
    int n_children;
    int action;
    int n_visits;      /* _n_visits                                         */
    double Q;          /* _Q                                                */
    double P;          /* _P                                                */
} node;

struct azo_mcts {
    azo_policy_fn fn; void *ctx;
    double c_puct; int n_playout;
    node *nodes; long n_nodes, cap;
    int root;
};

static long pool_alloc(azo_mcts *m, long n)
{
    if (m->n_nodes + n > m->cap) {
        long ncap = m->cap * 2;
        while (ncap < m->n_nodes + n) ncap *= 2;
        m->nodes = (node *)realloc(m->nodes, sizeof(node) * (size_t)ncap);
        m->cap = ncap;
    }
    long at = m->n_nodes;
    m->n_nodes += n;
    return at;
}

static void node_init(node *nd, int parent, int action, double prior)
{   /* mcts_alphaZero.py:30-36 */
    nd->parent = parent; nd->first_child = -1; nd->n_children = 0;
    nd->action = action; nd->n_visits = 0; nd->Q = 0.0; nd->P = prior;
}

azo_mcts *azo_mcts_new(azo_policy_fn fn, void *ctx, double c_puct, int n_playout)
{   /* mcts_alphaZero.py:98-111 */
    azo_mcts *m = (azo_mcts *)calloc(1, sizeof(*m));
    m->fn = fn; m->ctx = ctx; m->c_puct = c_puct; m->n_playout = n_playout;
    m->cap = 1 << 16;
    m->nodes = (node *)malloc(sizeof(node) * (size_t)m->cap);
    m->root = (int)pool_alloc(m, 1);
    node_init(&m->nodes[m->root], -1, -1, 1.0);
    return m;
}

void azo_mcts_free(azo_mcts *m) { if (m) { free(m->nodes); free(m); } }
int  azo_mcts_root_visits(const azo_mcts *m) { return m->nodes[m->root].n_visits; }
long azo_mcts_node_count(const azo_mcts *m) { return m->n_nodes; }

/* mcts_alphaZero.py:113-148 _playout */
static void playout(azo_mcts *m, azo_board *state)
{
    int nd = m->root;
    while (m->nodes[nd].first_child >= 0) {                     /* not is_leaf() */
        /* select (:48-55): first maximum of get_value over children in
         * insertion (= ascending action) order.  get_value (:75-83):
         * u = c_puct * P * sqrt(parent.N) / (1 + N), evaluated left to right. */
        const node *par = &m->nodes[nd];
        const double sq = sqrt((double)par->n_visits);
        int best = -1; double best_v = 0.0;
        for (int i = 0; i < par->n_children; i++) {
            const node *ch = &m->nodes[par->first_child + i];
            double u = ((m->c_puct * ch->P) * sq) / (double)(1 + ch->n_visits);
            double v = ch->Q + u;
            if (best < 0 || v > best_v) { best = i; best_v = v; }
        }
        nd = m->nodes[nd].first_child + best;
        azo_board_do_move(state, m->nodes[nd].action);
    }
    double probs[AZO_MAX_CELLS], leaf_value;
    m->fn(state, probs, &leaf_value, m->ctx);                   /* :132, unconditional */
    int end, winner;
    azo_board_game_end(state, &end, &winner);                   /* :134 */
    if (!end) {
        /* expand (:38-45) over zip(availables, probs[availables]) */
        int k = state->n_avail;
        long base = pool_alloc(m, k);
        for (int i = 0; i < k; i++)
            node_init(&m->nodes[base + i], nd, state->avail[i], probs[state->avail[i]]);
        m->nodes[nd].first_child = (int)base;
        m->nodes[nd].n_children = k;
    } else {
        if (winner == -1) leaf_value = 0.0;
        else leaf_value = (winner == state->current_player) ? 1.0 : -1.0;
    }
    /* update_recursive(-leaf_value) (:57-73): every ancestor, sign flips per ply */
    double x = -leaf_value;
    for (int cur = nd; cur >= 0; cur = m->nodes[cur].parent) {
        node *p = &m->nodes[cur];
        p->n_visits += 1;
        p->Q += 1.0 * (x - p->Q) / (double)p->n_visits;
        x = -x;
    }
}

/* numpy add.reduce over a contiguous float64 vector: pairwise summation with
 * 8 running partial sums per block of <=128 (numpy/core/src/umath/loops_utils.h). */
static double np_pairwise_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* mcts_alphaZero.py:16-19 softmax; :179 softmax(1.0/temp * log(visits + 1e-10)) */
static void visit_softmax(const int *visits, int k, double temp, double *probs)
{
    double mx = 0.0;
    for (int i = 0; i < k; i++) {
        probs[i] = (1.0 / temp) * log((double)visits[i] + 1e-10);
        if (i == 0 || probs[i] > mx) mx = probs[i];
    }
    for (int i = 0; i < k; i++) probs[i] = exp(probs[i] - mx);
    double s = np_pairwise_sum(probs, k);
    for (int i = 0; i < k; i++) probs[i] /= s;
}

/* mcts_alphaZero.py:154-181 get_move_probs */
int azo_mcts_get_move_probs(azo_mcts *m, const azo_board *state, double temp,
                            int *acts, int *visits, double *q, double *prior, double *probs)
{
    for (int n = 0; n < m->n_playout; n++) {
        azo_board copy = *state;                                /* copy.deepcopy(state) */
        playout(m, &copy);
    }
    const node *root = &m->nodes[m->root];
    int k = root->n_children;
    for (int i = 0; i < k; i++) {
        const node *ch = &m->nodes[root->first_child + i];
        acts[i] = ch->action; visits[i] = ch->n_visits;
        if (q) q[i] = ch->Q;
        if (prior) prior[i] = ch->P;
    }
    if (probs && k > 0) visit_softmax(visits, k, temp, probs);
    return k;
}

/* Root children statistics without running playouts (inspection helper). */
int azo_mcts_root_children(const azo_mcts *m, int *acts, int *visits, double *q, double *prior)
{
    const node *root = &m->nodes[m->root];
    int k = root->first_child >= 0 ? root->n_children : 0;
    for (int i = 0; i < k; i++) {
        const node *ch = &m->nodes[root->first_child + i];
        acts[i] = ch->action; visits[i] = ch->n_visits; q[i] = ch->Q; prior[i] = ch->P;
    }
    return k;
}

/* Copy the subtree below `src` into a fresh pool (keeps memory bounded; the
 * reference simply drops the Python references to the rest of the tree). */
static void copy_subtree(const node *old, node **dst, long *n, long *cap, int src_first,
                         int src_k, int new_parent)
{
    if (*n + src_k > *cap) {
        while (*n + src_k > *cap) *cap *= 2;
        *dst = (node *)realloc(*dst, sizeof(node) * (size_t)*cap);
    }
    long base = *n;
    *n += src_k;
    (*dst)[new_parent].first_child = (int)base;
    (*dst)[new_parent].n_children = src_k;
    for (int i = 0; i < src_k; i++) {
        (*dst)[base + i] = old[src_first + i];
        (*dst)[base + i].parent = new_parent;
        (*dst)[base + i].first_child = -1;
        (*dst)[base + i].n_children = 0;
    }
    for (int i = 0; i < src_k; i++) {
        const node *oc = &old[src_first + i];
        if (oc->first_child >= 0)
            copy_subtree(old, dst, n, cap, oc->first_child, oc->n_children, (int)(base + i));
    }
}

/* mcts_alphaZero.py:183-192 update_with_move */
void azo_mcts_update_with_move(azo_mcts *m, int last_move)
{
    int found = -1;
    const node *root = &m->nodes[m->root];
    if (last_move >= 0)
        for (int i = 0; i < root->n_children; i++)
            if (m->nodes[root->first_child + i].action == last_move) { found = root->first_child + i; break; }
    long cap = 1 << 12, n = 1;
    node *fresh = (node *)malloc(sizeof(node) * (size_t)cap);
    if (found >= 0) {
        fresh[0] = m->nodes[found];
        fresh[0].parent = -1; fresh[0].first_child = -1; fresh[0].n_children = 0;
        if (m->nodes[found].first_child >= 0)
            copy_subtree(m->nodes, &fresh, &n, &cap, m->nodes[found].first_child,
                         m->nodes[found].n_children, 0);
    } else {
        node_init(&fresh[0], -1, -1, 1.0);
    }
    free(m->nodes);
    m->nodes = fresh; m->n_nodes = n; m->cap = cap; m->root = 0;
}

/* ======================================================================= */
/* Pure (rollout) MCTS opponent -- mcts_pure.py, with the move filter of     */
/* game.py:25-58 (Board.get_move_target)                                     */
/* ======================================================================= */

/* game.py:35-58 get_move_target: the empty cells inside the bounding box of all
 * stones grown by 2 (clipped exactly as the reference clips it; an empty board
 * uses rows/columns 5..9).  Ascending order, like availables.  Returns the count. */
int azo_board_move_target(const azo_board *b, int *out)
{
    const int W = b->width, H = b->height;
    int top, bottom, left, right;
    if (b->n_stones == 0) {
        top = 9; bottom = 5; left = 5; right = 9;               /* game.py:40-41 */
    } else {
        top = -1; bottom = 1 << 30; left = 1 << 30; right = -1;
        for (int m = 0; m < W * H; m++) {
            if (!b->cell[m]) continue;
            int r = m / W, c = m % W;
            if (r > top) top = r;
            if (r < bottom) bottom = r;
            if (c < left) left = c;
            if (c > right) right = c;
        }
        const int step = 2;                                     /* game.py:48-52 */
        top = top <= H - step ? top + step : H;
        bottom = bottom >= step ? bottom - step : 0;
        left = left >= step ? left - step : 0;
        right = right <= W - step ? right + step : W;
    }
    int k = 0;
    for (int i = 0; i < b->n_avail; i++) {                      /* game.py:25-33,56 */
        int m = b->avail[i], row = m / W, col = m % W;
        if (row >= bottom && row <= top && col >= left && col <= right) out[k++] = m;
    }
    return k;
}

/* mcts_pure.py:193-212 _evaluate_rollout with rollout_policy_fn (:15-30): every step draws
 * np.random.rand(len(move_target)) and plays the first maximum. */
static double pure_rollout(azo_board *b, azo_rng *rng)
{
    const int player = b->current_player;
    int end = 0, winner = -1;
    for (int i = 0; i < 1000; i++) {
        azo_board_game_end(b, &end, &winner);
        if (end) break;
        int tgt[AZO_MAX_CELLS];
        const int k = azo_board_move_target(b, tgt);
        int best = -1; double best_v = 0.0;
        for (int j = 0; j < k; j++) {
            double v = azo_rng_double(rng);
            if (best < 0 || v > best_v) { best = tgt[j]; best_v = v; }
        }
        azo_board_do_move(b, best);
    }
    if (winner == -1) return 0.0;
    return winner == player ? 1.0 : -1.0;
}

/* mcts_pure.py:156-191 _playout: descend, expand over move_target with P = 1/K (float64,
 * policy_value_fn :33-46), rollout, update_recursive(-leaf_value). */
static void pure_playout(azo_mcts *m, azo_board *state, azo_rng *rng)
{
    int nd = m->root;
    while (m->nodes[nd].first_child >= 0) {
        const node *par = &m->nodes[nd];
        const double sq = sqrt((double)par->n_visits);
        int best = -1; double best_v = 0.0;
        for (int i = 0; i < par->n_children; i++) {
            const node *ch = &m->nodes[par->first_child + i];
            double u = ((m->c_puct * ch->P) * sq) / (double)(1 + ch->n_visits);
            double v = ch->Q + u;
            if (best < 0 || v > best_v) { best = i; best_v = v; }
        }
        nd = m->nodes[nd].first_child + best;
        azo_board_do_move(state, m->nodes[nd].action);
    }
    int tgt[AZO_MAX_CELLS];
    const int k = azo_board_move_target(state, tgt);            /* :176 via policy_value_fn */
    int end, winner;
    azo_board_game_end(state, &end, &winner);
    if (!end && k > 0) {
        long base = pool_alloc(m, k);
        for (int i = 0; i < k; i++) node_init(&m->nodes[base + i], nd, tgt[i], 1.0 / (double)k);
        m->nodes[nd].first_child = (int)base;
        m->nodes[nd].n_children = k;
    }
    const double leaf_value = pure_rollout(state, rng);
    double x = -leaf_value;
    for (int cur = nd; cur >= 0; cur = m->nodes[cur].parent) {
        node *p = &m->nodes[cur];
        p->n_visits += 1;
        p->Q += 1.0 * (x - p->Q) / (double)p->n_visits;
        x = -x;
    }
}

/* mcts_pure.py:214-231 get_move from a fresh root (MCTSPlayer.get_action resets the tree after
 * every move, :261-268): n_playout playouts, then the first most-visited root child.  acts /
 * visits / q describe the root's children; returns the move (-1 when the root has none). */
int azo_pure_get_move(const azo_board *state, double c_puct, int n_playout, azo_rng *rng,
                      int *n_children, int *acts, int *visits, double *q, int *root_visits)
{
    azo_mcts *m = azo_mcts_new(NULL, NULL, c_puct, n_playout);
    for (int n = 0; n < n_playout; n++) {
        azo_board copy = *state;
        pure_playout(m, &copy, rng);
    }
    const node *root = &m->nodes[m->root];
    const int k = root->first_child >= 0 ? root->n_children : 0;
    int move = -1, best = -1;
    for (int i = 0; i < k; i++) {
        const node *ch = &m->nodes[root->first_child + i];
        if (acts) acts[i] = ch->action;
        if (visits) visits[i] = ch->n_visits;
        if (q) q[i] = ch->Q;
        if (ch->n_visits > best) { best = ch->n_visits; move = ch->action; }
    }
    if (n_children) *n_children = k;
    if (root_visits) *root_visits = root->n_visits;
    azo_mcts_free(m);
    return move;
}

/* ======================================================================= */
/* Self-play -- MCTSPlayer.get_action (mcts_alphaZero.py:212-260) inside     */
/* Game.start_self_play (game.py:360-416)                                   */
/* ======================================================================= */

int azo_self_play(int width, int height, int n_in_row, azo_policy_fn fn, void *ctx,
                  double c_puct, int n_playout, double temp, uint32_t seed,
                  azo_game_record *rec)
{
    const int S = width * height;
    azo_board board;
    if (azo_board_init(&board, width, height, n_in_row, 0) != 0) return -1;
    azo_mcts *m = azo_mcts_new(fn, ctx, c_puct, n_playout);
    azo_rng rng;
    azo_rng_seed(&rng, seed);                                   /* np.random.seed(seed) */
    int acts[AZO_MAX_CELLS], visits[AZO_MAX_CELLS];
    double probs[AZO_MAX_CELLS], noise[AZO_MAX_CELLS], cdf[AZO_MAX_CELLS];
    int t = 0;
    rec->winner = -1;
    for (;;) {
        int k = azo_mcts_get_move_probs(m, &board, temp, acts, visits, NULL, NULL, probs);
        /* move_probs = zeros(S); move_probs[acts] = probs (:216-222) */
        if (rec->pi) {
            for (int i = 0; i < S; i++) rec->pi[(size_t)t * S + i] = 0.0;
            for (int i = 0; i < k; i++) rec->pi[(size_t)t * S + acts[i]] = probs[i];
        }
        if (rec->visits) {
            for (int i = 0; i < S; i++) rec->visits[(size_t)t * S + i] = 0;
            for (int i = 0; i < k; i++) rec->visits[(size_t)t * S + acts[i]] = visits[i];
        }
        /* np.random.choice(acts, p=0.75*probs + 0.25*dirichlet(0.1*ones(k))) (:236-240):
         * dirichlet drawn first, then one uniform; cdf=cumsum(p); cdf/=cdf[-1];
         * searchsorted(side='right'). */
        azo_rng_dirichlet(&rng, 0.1 * 1.0, k, noise);
        double run = 0.0;
        for (int i = 0; i < k; i++) { run += 0.75 * probs[i] + 0.25 * noise[i]; cdf[i] = run; }
        double last = cdf[k - 1];
        for (int i = 0; i < k; i++) cdf[i] /= last;
        double u = azo_rng_double(&rng);
        int lo = 0, hi = k;                                     /* first index with cdf > u */
        while (lo < hi) { int mid = (lo + hi) / 2; if (u < cdf[mid]) hi = mid; else lo = mid + 1; }
        int idx = lo < k ? lo : k - 1;
        int move = acts[idx];
        azo_mcts_update_with_move(m, move);                     /* tree reuse (:242) */
        /* game.py:381-387 */
        if (rec->states) azo_board_current_state(&board, rec->states + (size_t)t * 4 * S);
        rec->players[t] = board.current_player;
        rec->moves[t] = move;
        azo_board_do_move(&board, move);
        t++;
        int end, winner;
        azo_board_game_end(&board, &end, &winner);
        if (end) {                                              /* game.py:393-416 */
            rec->n_plies = t; rec->winner = winner;
            for (int i = 0; i < t; i++)
                rec->z[i] = (winner == -1) ? 0.0 : (rec->players[i] == winner ? 1.0 : -1.0);
            break;
        }
    }
    azo_mcts_free(m);
    return 0;
}

long azo_self_play_many(int width, int height, int n_in_row, const azo_net *net,
                        double c_puct, int n_playout, double temp, uint32_t seed0,
                        int n_games, int n_threads, int *plies_out, int *winners_out)
{
    long total = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int g = 0; g < n_games; g++) {
        azo_game_record rec;
        memset(&rec, 0, sizeof(rec));
        azo_self_play(width, height, n_in_row, azo_policy_net, (void *)net, c_puct, n_playout,
                      temp, seed0 + (uint32_t)g, &rec);
        if (plies_out) plies_out[g] = rec.n_plies;
        if (winners_out) winners_out[g] = rec.winner;
        total += rec.n_plies;
    }
    return total;
}
