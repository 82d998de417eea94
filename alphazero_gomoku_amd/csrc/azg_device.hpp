// Shared host/device definitions of the self-play engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace azg {

constexpr int MAXW = 4;            // 64-bit words per colour (boards up to 16x16 = 256 cells)
constexpr int MAXS = 256;
constexpr int WAVE = 64;
constexpr int kStampWords = 32 + 2 * 2048;   // diagnostic counters (see Prof)
constexpr int kGateWords = 16 * 256;          // (XCC id, SE, SH, CU) -> gate word

// Split-half weight images (convolutions, fully connected layers): a lane's operand of one k-step is
// 8 high halves and 8 low halves.  Planar (1): per k-step the 64 lanes' high halves are contiguous
// (1 KB), then the 64 lanes' low halves, so each of the two 16-byte loads of a wave covers eight
// full 128-byte lines.  Interleaved (0): a lane's high and low halves are adjacent (32 bytes per
// lane), each load touching half of sixteen lines.
#ifndef AZG_WPLANAR
#define AZG_WPLANAR 0
#endif
constexpr int kWLo = AZG_WPLANAR ? 64 : 1;        // distance of a lane's low halves from its high halves, in 16-byte units
constexpr int kWLane = AZG_WPLANAR ? 1 : 2;       // distance between neighbouring lanes, in 16-byte units
// offset (in halves) of half j (0..7 high, 8..15 low) of `lane` inside one k-step's 1024-half block
__host__ __device__ inline int wsplit_off(int lane, int j)
{
    return AZG_WPLANAR ? ((j >> 3) * 64 + lane) * 8 + (j & 7) : lane * 16 + j;
}

// Edge meta word: action (cell id) in the low 16 bits, number of children in the high 16.
__host__ __device__ inline unsigned meta_pack(int act, int k) { return (unsigned)act | ((unsigned)k << 16); }
__host__ __device__ inline int meta_act(unsigned m) { return (int)(m & 0xFFFFu); }
__host__ __device__ inline int meta_k(unsigned m) { return (int)(m >> 16); }

// Packed CNN weights (engine-owned copy, float32).  Conv kernels are stored per
// 32-output-channel tile and k-step so that one wave reads 64 consecutive floats
// per MFMA A operand: wconv[l][(cotile*KS + kstep)*64 + lane] =
// W[co = cotile*32 + (lane&31)][ci = 2*(kstep/9) + (lane>>5)][tap = kstep%9].
struct NetWeights {
    const float *wconv[3];   // packed conv1..conv3 (float32 path)
    const void *w16[3];      // packed conv1..conv3 as split halves (azg_cnn16.hpp)
    const void *w16w3;       // conv3 in Winograd form, G g G^T as split halves (azg_cnn16.hpp, conv3_wino_f16)
    const void *wfc16[2];    // act_fc1 / val_fc1 as split halves in MFMA A-operand order (azg_cnn16.hpp)
    const void *whead16;     // the two 1x1 head convolutions as split halves (azg_cnn16.hpp), 4 KB
    const float *bconv[3];   // biases [32],[64],[128]
    const float *whead;      // [6][128]: rows 0-3 act_conv1, rows 4-5 val_conv1
    const float *bhead;      // [6]
    const float *wpfc;       // act_fc1 as [k/4][S outputs][4]   (k = input feature)
    const float *bpfc;       // [S]
    const float *wvfc1;      // val_fc1 as [k/4][64 outputs][4], k zero-padded to a multiple of 4
    const float *bvfc1;      // [64]
    const float *wvfc2;      // [64]
    const float *bvfc2;      // [1]
    const float *wpfc_k;     // act_fc1 as [k][256]: one row of outputs per input feature (sparse-input FC)
    const float *wvfc1_k;    // val_fc1 as [k][64]
};

// Returns p unchanged but opaque to the optimiser.  The CNN runs inside the playout loop and
// its weights never change during a launch; without this the compiler hoists weight loads
// out of that loop and spills hundreds of registers to keep them.
template <typename T>
__device__ inline const T *opaque(const T *p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// Global-memory view of a pointer.  After opaque() the compiler no longer knows the address
// space and would emit flat loads, which count on both memory counters and force full waits.
template <typename T>
using gptr = const T __attribute__((address_space(1))) *;
template <typename T>
__device__ inline gptr<T> as_global(const T *p) { return (gptr<T>)p; }

// A copy of the weight table whose pointers are opaque (call once per forward pass).
__device__ inline NetWeights opaque_weights(const NetWeights &w)
{
    NetWeights o;
    for (int i = 0; i < 3; i++) { o.wconv[i] = opaque(w.wconv[i]); o.w16[i] = opaque((const char *)w.w16[i]); o.bconv[i] = opaque(w.bconv[i]); }
    for (int i = 0; i < 2; i++) o.wfc16[i] = opaque((const char *)w.wfc16[i]);
    o.whead16 = opaque((const char *)w.whead16);
    o.w16w3 = opaque((const char *)w.w16w3);
    o.whead = opaque(w.whead); o.bhead = opaque(w.bhead); o.wpfc = opaque(w.wpfc); o.bpfc = opaque(w.bpfc);
    o.wvfc1 = opaque(w.wvfc1); o.bvfc1 = opaque(w.bvfc1); o.wvfc2 = opaque(w.wvfc2); o.bvfc2 = opaque(w.bvfc2);
    o.wpfc_k = opaque(w.wpfc_k); o.wvfc1_k = opaque(w.wvfc1_k);
    return o;
}

// Two workgroups share a compute unit and its matrix pipes.  Matrix instructions of two
// waves on one SIMD simply queue behind each other, so the pair gains only while one
// workgroup's convolutions overlap the other's tree / softmax / load phases.  The gate makes
// that the steady state: a workgroup takes its CU's gate before the dense convolution phase
// and gives it back after, so the partner's dense phase starts when this one ends and the two
// fall into opposite phase.  The gate is advisory: the wait is bounded and results never
// depend on it (a workgroup that gives up just runs its convolutions beside the partner's).
// The table is shared by every engine of the process on a device, so workgroups of two
// engines that run concurrently on separate streams pair up the same way.  azg_create and
// azg_selfplay_begin clear it (a launch that died inside a gated span would otherwise leave words
// set); a workgroup that gives up waiting counts that in g_gate_giveups, and its release() frees the
// word again, so a word left set costs one bounded wait per CU, not one per playout.
__device__ int g_matrix_gate[kGateWords];
__device__ unsigned int g_gate_giveups[4];    // [0]: acquires that ran out of tries since the process started

// Experiment builds (-DAZG_QUAD8=1, 8x8 only): ONE 512-thread workgroup per CU holds both game
// pairs, as two groups of four waves that run the per-pair code independently on their own LDS
// regions.  The hardware barrier would join all eight waves, so the groups use a barrier on two LDS
// words each (arrival count, generation) at the start of the dynamic LDS, and the matrix gate is an
// LDS word.  kLdsHead floats at the start of the dynamic LDS are reserved for these words.
#ifndef AZG_QUAD8
#define AZG_QUAD8 0
#endif
#if AZG_QUAD8
constexpr int kLdsHead = 16;
__device__ inline int azg_tid() { return (int)(threadIdx.x & 255u); }
__device__ inline void azg_block_barrier()
{
    extern __shared__ unsigned azg_dyn_lds[];
    unsigned *ctl = azg_dyn_lds + (threadIdx.x >> 8) * 4;          // [0] arrivals, [1] generation
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const unsigned g0 = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((threadIdx.x & 63u) == 0) {
        const unsigned old = atomicAdd(&ctl[0], 1u);
        if (old == 3u) {
            __hip_atomic_store(&ctl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&ctl[1], g0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    for (int spin = 0; spin < (1 << 22); spin++) {
        if (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != g0) break;
        __builtin_amdgcn_s_sleep(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#ifndef AZG_QUAD_FINE
#define AZG_QUAD_FINE 0       // 1: the matrix gate per SIMD, held by a wave for the length of a k-loop only (measured 15 % slower)
#endif
// One gate word per SIMD (words 8..11 of the LDS head): wave w of either group runs on SIMD
// HW_ID[5:4]; the two waves that share a SIMD take turns with their dense loops, nothing else waits.
__device__ inline void azg_simd_gate_acquire()
{
    extern __shared__ unsigned azg_dyn_lds[];
    unsigned *w = azg_dyn_lds + 8 + ((__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11))) & 3u);
    // Every lane issues the compare-and-swap (one instruction, the lanes are served one after the
    // other: the first finds 0, the rest 1).  A branch for lane 0 alone, next to the unrolled MFMA
    // loops with their scheduling directives, gave WRONG convolution results (measured: a lane-0
    // store placed behind the k-loop is enough, the same store by all lanes is harmless).
    for (int tries = 0; tries < 400000; tries++) {
        const unsigned old = atomicCAS(w, 0u, 1u);
        if (__ballot(old == 0u) != 0ULL) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ inline void azg_simd_gate_release()
{
    extern __shared__ unsigned azg_dyn_lds[];
    unsigned *w = azg_dyn_lds + 8 + ((__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11))) & 3u);
    __hip_atomic_store(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // all lanes, see acquire
}
#else
constexpr int kLdsHead = 0;
__device__ inline int azg_tid() { return (int)threadIdx.x; }
__device__ inline void azg_block_barrier() { __syncthreads(); }
#endif

#ifndef AZG_GATE_SLEEP
#define AZG_GATE_SLEEP 16     // s_sleep argument between two polls of a taken gate (x 64 cycles)
#endif
#ifndef AZG_GATE_RELEASE
#define AZG_GATE_RELEASE 0    // how the gate word is given back: 0 agent-scope store, 1 atomic exchange, 2 workgroup-scope store
#endif
#ifndef AZG_GATE_TIMED
#define AZG_GATE_TIMED 0      // 1: a waiting workgroup sleeps until the holder is due to release (see acquire)
#endif
#ifndef AZG_GATE_MARGIN
#define AZG_GATE_MARGIN 384   // cycles before the expected release at which close polling starts
#endif
struct MatrixGate {
    int *word;
    mutable int t_acq = 0, span = 0;       // thread 0: when it took the gate last (clock, low bits | 1), how long it held it
    __device__ void init(bool enabled)
    {
        // HW_REG_HW_ID bits: CU 11:8, SH 12, SE 15:13; HW_REG_XCC_ID bits 3:0
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        word = enabled ? g_matrix_gate + ((xcc & 15u) * 256u + (hw & 255u)) : nullptr;
#if AZG_QUAD8
        extern __shared__ unsigned azg_dyn_lds[];
        if (enabled) word = AZG_QUAD_FINE ? nullptr : reinterpret_cast<int *>(azg_dyn_lds + 12);      // both groups of the workgroup share it
#endif
    }
    // thread 0 of the workgroup, followed by a workgroup barrier
    __device__ void acquire() const
    {
        if (!word) return;
#if AZG_GATE_TIMED
        // The word holds the holder's acquire time.  Both workgroups of a CU do the same work inside
        // the gate, so the holder is due to release about `span` cycles after it took the gate: sleep
        // until shortly before, then poll closely.  (Polling closely all the time was measured slower.)
        int now = (int)__builtin_amdgcn_s_memtime() | 1;
        int old = atomicCAS(word, 0, now);
        if (old == 0) { t_acq = now; return; }
        const int due = old + span - AZG_GATE_MARGIN;
        for (int guard = 0; guard < 4096; guard++) {
            const int rem = due - (int)__builtin_amdgcn_s_memtime();
            if (rem <= 0 || rem > (1 << 20)) break;
            if (rem >= 1024) __builtin_amdgcn_s_sleep(15); else __builtin_amdgcn_s_sleep(2);
        }
        for (int tries = 0; tries < 3000; tries++) {
            now = (int)__builtin_amdgcn_s_memtime() | 1;
            if (atomicCAS(word, 0, now) == 0) { t_acq = now; return; }
            if (tries < 64) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(16);
        }
#else
        for (int tries = 0; tries < (AZG_QUAD8 ? 200000 : 2000 * 16 / AZG_GATE_SLEEP); tries++) {
            if (atomicCAS(word, 0, 1) == 0) return;
            if (AZG_QUAD8) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(AZG_GATE_SLEEP);
        }
#endif
        atomicAdd(&g_gate_giveups[0], 1u);         // the convolutions run beside the partner's this once
    }
    // thread 0 of the workgroup, after a workgroup barrier
    __device__ void release() const
    {
        if (!word) return;
#if AZG_GATE_TIMED
        const int held = (int)__builtin_amdgcn_s_memtime() - t_acq;
        if (held > 0 && held < (1 << 20)) span = held;
#endif
#if AZG_GATE_RELEASE == 1
        (void)__hip_atomic_exchange(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // at the L2, like the polls
#elif AZG_GATE_RELEASE == 2
        __hip_atomic_store(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        __hip_atomic_store(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
};

// Phase timer of the diagnostic build: thread 0 of every workgroup adds the cycles since its
// previous mark to slot i.  Compiles to nothing in the normal build.
struct Prof {
#ifdef AZG_STAMPS
    // s_getreg operands: id | offset << 6 | (width - 1) << 11
    static constexpr int GETREG_HW_ID = 4 | (0 << 6) | (31 << 11), GETREG_XCC_ID = 20 | (0 << 6) | (3 << 11);
    long long *out; long long last, t0, r0; bool on;
    unsigned acc[28];
    __device__ void start(long long *o, int tid_on = 0)
    {
        out = o; on = o != nullptr && (int)threadIdx.x == tid_on;
        for (int i = 0; i < 28; i++) acc[i] = 0;
        last = t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ void mark(int i) { if (on) { const long long now = __builtin_amdgcn_s_memtime(); acc[i] += (unsigned)(now - last); last = now; } }
    __device__ void count(int i, int v) { if (on) acc[i] += (unsigned)v; }      // a plain counter in slot i
    // a mark that waits until the vector value `dep` exists (the clock is read by a scalar instruction,
    // which the compiler and the hardware would otherwise let overtake the arithmetic it is meant to follow)
    __device__ void mark_after(int i, int dep)
    {
        if (on) {
            asm volatile("s_nop 0" :: "v"(dep) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            mark(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // slots 30 / 31: shader cycles and 100 MHz ticks of the whole loop (their ratio is the clock held)
    __device__ void finish()
    {
        if (!on) return;
        for (int i = 0; i < 28; i++) atomicAdd((unsigned long long *)&out[i], (unsigned long long)acc[i]);
        atomicAdd((unsigned long long *)&out[30], (unsigned long long)(__builtin_amdgcn_s_memtime() - t0));
        const unsigned long long rt = (unsigned long long)(__builtin_amdgcn_s_memrealtime() - r0);
        atomicAdd((unsigned long long *)&out[31], rt);
        if (blockIdx.x < 2048) {
            out[32 + 2 * blockIdx.x] = (long long)rt;
            out[33 + 2 * blockIdx.x] = ((long long)__builtin_amdgcn_s_getreg(GETREG_XCC_ID) << 32) | __builtin_amdgcn_s_getreg(GETREG_HW_ID);
        }
    }
#else
    __device__ void start(long long *, int = 0) {}
    __device__ void mark(int) {}
    __device__ void count(int, int) {}
    __device__ void mark_after(int, int) {}
    __device__ void finish() {}
#endif
};

// Everything the kernels need, passed by value.
struct DevState {
    int N, S, NW, n_in_row, B, cap, prior_mode, n_playout;
    double c_puct;
    // live boards: stones of player 1 / player 2 as bit sets
    unsigned long long *bb;        // [B][2][MAXW]
    int *ply;                      // [B] len(states)
    int *last_move;                // [B]
    int *first_player;             // [B] player (1/2) who moves at even ply
    signed char *status;           // [B] 0 live, 1 ended, 2 idle (never started)
    signed char *winner;           // [B] 1, 2, -1
    int *err;                      // [B] sticky error bits (1 = arena overflow)
    // search trees: two arenas per game, the live one selected by cur[g]
    float *eP;                     // [B][2][cap] prior
    int *eN;                       // [B][2][cap] visits
    double *eQ;                    // [B][2][cap] mean value
    int *eChild;                   // [B][2][cap] first child edge or -1
    unsigned *eMeta;               // [B][2][cap] action | children << 16
    int *used;                     // [B] edges in use in the live arena
    // large arenas for the few games whose re-used tree outgrows the standard one (a sharply peaked
    // net keeps most of the previous search below the move played: root visits approach
    // n_playout / (1 - share of the chosen child)).  cap_big holds ANY tree: after t moves a root has
    // at most (t+1)*n_playout visits and a node at most S-t children, (t+1)(S-t) <= ((S+1)/2)^2.
    int cap_big, n_big;
    int *big_of;                   // [B] large arena the game lives in, or -1
    int *big_owner;                // [n_big] slot that owns the large arena, or -1
    float *bP; int *bN; double *bQ; int *bChild; unsigned *bMeta;   // [n_big][2][cap_big]
    unsigned long long *arena_stats;   // [4] moves into a large arena | trees dropped because none was free | re-roots in place | re-roots that compacted
    unsigned char *cur;            // [B]
    int rr_cap;                    // LDS entries of a re-root's work stack (128 .. kRerootStackMax)
    int rr_always;                 // 1: every re-root compacts the kept subtree into the other arena (tests)
    long long *evals;              // [B] leaf evaluations that needed priors (profiling)
    // per-game training samples, two banks per slot: the game in progress records into bank
    // s_bank[g]; a slot restarted on the device (pipelined self-play) flips the bank, so the
    // finished game's samples stay readable until they are drained
    unsigned long long *s_bb;      // [B][2][S][2][MAXW] stones before the move
    int *s_last;                   // [B][2][S] last_move before the move
    int *s_ply;                    // [B][2][S] len(states) before the move
    double *s_pi;                  // [B][2][S][S]
    int *s_count;                  // [B] samples recorded (= plies played by the engine)
    unsigned char *s_bank;         // [B]
    long long *game_id;            // [B]
    unsigned char *drained;        // [B]
    // record of the game that ended in the slot during the last advance (pipelined self-play)
    long long *fin_id;             // [B] its game id
    int *fin_count;                // [B] its number of samples
    signed char *fin_flag;         // [B] 1: a game ended in this slot in the last advance
    signed char *fin_winner;       // [B]
    unsigned char *fin_bank;       // [B] bank holding its samples
    unsigned long long *next_k;    // [1] games handed out so far (device-side restart counter)
    long long *stamps;             // [kStampWords] counters of a diagnostic build (-DAZG_STAMPS)
    NetWeights net;
};

}  // namespace azg
