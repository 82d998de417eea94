// libazgomoku.so -- C ABI (include/azgomoku.h) over the HIP kernels in azg_kernels.hpp.
// Host side: memory, launches, the numpy-legacy noise stream, repacking of weights.
#include "../../include/azgomoku.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "azg_kernels.hpp"
#include "azg_pure.hpp"
#ifndef AZG_TRAIN_BN3
#define AZG_TRAIN_BN3 64      // columns per workgroup of conv3's forward GEMM (64: two workgroups per CU; 128 measured 3 % slower)
#endif
#ifndef AZG_TRAIN_BD3
#define AZG_TRAIN_BD3 32      // columns per workgroup of conv3's backward-data GEMM (64 columns in all; 64 measured 3 % slower)
#endif
#ifndef AZG_TRAIN_WCAP
#define AZG_TRAIN_WCAP 128    // most splits of the batch in the weight-gradient GEMMs of conv2 / conv3 (32: 25 % slower step)
#endif
#include "azg_train.hpp"
#include "np_legacy_rng.hpp"

using namespace azg;

static thread_local std::string g_create_error;

// Worker threads of one handle that draw the exploration noise of a move step (make_noise).  They
// are started on first use and sleep between steps; the calling thread takes chunks as well.
struct NoisePool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable wake, done;
    unsigned long long round = 0;
    int running = 0, chunks = 0;
    bool stop = false;
    std::atomic<int> next{0};
    const std::function<void(int)> *job = nullptr;

    void drain()
    {
        for (int c = next.fetch_add(1); c < chunks; c = next.fetch_add(1)) (*job)(c);
    }
    void loop()
    {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            wake.wait(lk, [&] { return stop || round != seen; });
            if (stop) return;
            seen = round;
            lk.unlock();
            drain();
            lk.lock();
            if (--running == 0) done.notify_one();
        }
    }
    // runs job(0) .. job(n_chunks - 1), each exactly once, and returns when all are finished
    void run(int n_chunks, int n_threads, const std::function<void(int)> &fn)
    {
        if (n_threads <= 1 || n_chunks <= 1) {
            for (int c = 0; c < n_chunks; c++) fn(c);
            return;
        }
        while ((int)workers.size() < n_threads - 1) workers.emplace_back([this] { loop(); });
        {
            std::lock_guard<std::mutex> lk(m);
            job = &fn; chunks = n_chunks; next.store(0);
            running = (int)workers.size();
            round++;
        }
        wake.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return running == 0; });
    }
    ~NoisePool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        wake.notify_all();
        for (auto &t : workers) t.join();
    }
};

// Scratch of azg_board_step_batch_static: one fixed array per GPU (every device that loads this
// code object holds its own copy at its own address), so that entry point never allocates.  In host
// code the array's name is only a handle for hipGetSymbolAddress; the device address of each GPU's
// copy is looked up once per device (static_scratch_ptr) and used everywhere else.
constexpr int kStaticMaxBoards = 1024;                       // boards per call, any supported board size
constexpr size_t kStaticStonesOff = 0;                                                  // u64 [n][2][MAXW]
constexpr size_t kStaticIntsOff = kStaticStonesOff + (size_t)kStaticMaxBoards * 2 * MAXW * 8;   // i32 ply|last|moves|player [4][max]
constexpr size_t kStaticBadOff = kStaticIntsOff + (size_t)4 * kStaticMaxBoards * 4;     // i32 [4] illegal-move flag
constexpr size_t kStaticEndOff = kStaticBadOff + 16;                                    // u8 [max] ended
constexpr size_t kStaticWinOff = kStaticEndOff + kStaticMaxBoards;                      // i8 [max] winner
constexpr size_t kStaticPlanesOff = kStaticWinOff + kStaticMaxBoards;                   // f32 [max][4][MAXS]
constexpr size_t kStaticBytes = kStaticPlanesOff + (size_t)kStaticMaxBoards * 4 * MAXS * 4;
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_static_mutex;
static unsigned char *g_static_ptr[kMaxDevices] = {nullptr};

// Device address of this GPU's copy of `scratch`; the device must be current.
static hipError_t static_scratch_ptr(int device, unsigned char **out)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_static_mutex);
    if (!g_static_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_static_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_static_ptr[device];
    return hipSuccess;
}

// Device addresses of this GPU's matrix-gate table and give-up counter (azg_device.hpp), looked up
// once per device like the scratch array's; the device must be current.
static int *g_gate_ptr[kMaxDevices] = {nullptr};
static unsigned *g_giveup_ptr[kMaxDevices] = {nullptr};
static hipError_t gate_ptrs(int device, int **gate, unsigned **giveups)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_static_mutex);
    if (!g_gate_ptr[device]) {
        void *p = nullptr, *q = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_matrix_gate));
        if (e == hipSuccess) e = hipGetSymbolAddress(&q, HIP_SYMBOL(g_gate_giveups));
        if (e != hipSuccess) return e;
        g_gate_ptr[device] = static_cast<int *>(p);
        g_giveup_ptr[device] = static_cast<unsigned *>(q);
    }
    *gate = g_gate_ptr[device];
    *giveups = g_giveup_ptr[device];
    return hipSuccess;
}

struct azg_handle {
    azg_config cfg{};
    DevState st{};
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void *> allocs;
    unsigned char *static_scratch = nullptr;     // this device's copy of the file-scope scratch array
    int *gate_table = nullptr;                   // this device's matrix-gate table (kGateWords words)
    unsigned *gate_giveups = nullptr;            // and its give-up counter
    int *h_static_bad = nullptr;                 // pinned: illegal-move flag read back by the static entry point
    int playout_parked = -1;                     // slot with a playout between azg_playout_begin and azg_playout_end
    bool have_weights = false;
    float *d_weights = nullptr;
    // host mirrors, refreshed after every state-changing call
    std::vector<int> h_ply, h_scount, h_err, h_fin_count;
    std::vector<signed char> h_status, h_winner, h_fin_flag, h_fin_winner;
    std::vector<unsigned char> h_drained, h_fin_bank;
    std::vector<long long> h_game_id, h_fin_id;
    std::vector<NpLegacyRng> rng;
    NoisePool noise_pool;
    double noise_ms = 0.0;            // host time spent drawing noise since the last azg_host_timing
    long long noise_steps = 0;
    // pinned staging
    double *h_noise = nullptr, *h_u = nullptr;
    int *h_moves = nullptr;
    // mirror block (one D2H copy per refresh): i64 game_id | fin_id, i32 ply | s_count | err |
    // fin_count, i8 status | winner | fin_flag | fin_winner | fin_bank  = 37 bytes per slot
    char *h_mirror = nullptr;        // pinned
    char *d_mirror = nullptr;
    bool mirrors_fresh = false;
    int search_parts = 0;                        // launches a move's playouts are spread over (AZG_SEARCH_PARTS; 0: the default of azg_search)
    size_t mirror_events = 0;        // profiling events recorded before the last mirror copy
    // pipelined self-play (azg_selfplay_*)
    struct Pipeline {
        bool active = false;
        double temp = 1.0;
        uint32_t seed0 = 0;
        long long first = 0, stride = 1, total = -1;
        hipEvent_t ev_mirror = nullptr, ev_noise = nullptr;
        hipStream_t copy_stream = nullptr;
        int par = 0;                             // parity of the sample set being produced
        int pend_n = 0, pend_games = 0;          // enqueued in the last step, complete by the next one
        int ready_n = 0, ready_games = 0, ready_par = 0;
        bool fetched = true;                     // the ready samples were fetched (or there are none)
        int *h_map = nullptr;                    // pinned: per parity [3][B*S] slot | sample | aux
        long long *h_ids = nullptr;              // pinned: per parity [B*S]
    } sp;
    double *d_noise = nullptr, *d_u = nullptr, *d_pi = nullptr;
    int *d_moves = nullptr, *d_given = nullptr;
    // grow-on-demand device scratch
    void *scratch[48] = {nullptr};
    size_t scratch_sz[48] = {0};
    // training step (azg_train_*): activations, gradients and partial sums of one mini-batch
    struct TrainWs {
        int cap_n = 0;                           // samples the buffers hold
        char *block = nullptr;                   // one allocation, carved up below
        hipStream_t stream = nullptr;            // the training kernels' own stream (never behind a search in flight)
        float *x4 = nullptr, *a[3] = {nullptr}, *dz[3] = {nullptr}, *hp = nullptr, *hv = nullptr, *dh8 = nullptr;
        float *dlogit = nullptr, *dh1 = nullptr, *g2 = nullptr, *lossparts = nullptr, *lossout = nullptr;
        TrainImages im{};
        float *wpart[3] = {nullptr}, *bpart[3] = {nullptr}, *hpart = nullptr, *hbpart = nullptr;
        float *ppart = nullptr, *pbpart = nullptr, *vpart = nullptr, *vbpart = nullptr, *g2part = nullptr;
        int wsplits[3] = {0}, wrows[3] = {0}, hsplits = 0, hrows = 0, fsplits = 0, frows = 0, gsplits = 0, grows = 0;
        const float *checked[16] = {nullptr};    // parameter pointers already seen to be device memory
        float *h_loss = nullptr;                 // pinned: loss, entropy, and a double (KL) at +8
        double *klparts = nullptr;               // [n] per-sample sums of the KL test
    } tw;
    // profiling
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_pending = 0;
    long long prof_launches = 0;
    double prof_ms = 0.0;
};

static int fail(azg_handle *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, AZG_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
static int dev_alloc(azg_handle *h, T **p, size_t count, bool zero = true)
{
    void *q = nullptr;
    HIPCHK(h, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    h->allocs.push_back(q);
    if (zero) HIPCHK(h, hipMemsetAsync(q, 0, std::max<size_t>(count, 1) * sizeof(T), h->stream));
    *p = reinterpret_cast<T *>(q);
    return AZG_OK;
}

// true when p points into device memory (then a kernel can use it without staging)
static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                              // plain host memory: not an error
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

static int scratch_get(azg_handle *h, int slot, size_t bytes, void **out)
{
    if (h->scratch_sz[slot] < bytes) {
        if (h->scratch[slot]) HIPCHK(h, hipFree(h->scratch[slot]));
        h->scratch[slot] = nullptr;
        h->scratch_sz[slot] = 0;
        size_t want = std::max<size_t>(bytes, 256);
        HIPCHK(h, hipMalloc(&h->scratch[slot], want));
        h->scratch_sz[slot] = want;
    }
    *out = h->scratch[slot];
    return AZG_OK;
}

// ---------------------------------------------------------------- kernel dispatch
struct SearchLaunch {
    void (*fn)(DevState);
    int T;                  // games per workgroup
    size_t lds;
    int threads = 256;
};

// Two wave groups per workgroup (azg_dual.hpp): TS games per set, two sets.
template <int N, int TS>
static SearchLaunch dual_launch()
{
    return {search_dual_kernel<N, TS>, 2 * TS, (size_t)DualGeom<N, TS>::TOTAL * 4, 512};
}

template <int N, int T, bool F16>
static SearchLaunch net_launch()
{
    using G = typename GeomPick<N, T, F16>::type;
    return {search_kernel<N, T, 0, F16>, T, (size_t)G::END * 4 + T * sizeof(GameLdsT<N * N + 1>)};
}

#ifndef AZG_DUAL
#define AZG_DUAL 1      // boards of 9x9 .. 15x15: the kernel with two wave groups per workgroup (azg_dual.hpp)
#endif
#ifndef AZG_T8
#define AZG_T8 2        // games per workgroup of the split-f16 8x8 search kernel
#endif
static bool pick_search(int N, int mode, int cnn_mode, SearchLaunch *out)
{
    if (mode == AZG_PRIOR_UNIFORM32) { *out = {search_kernel<0, 4, 1, false>, 4, 4 * sizeof(GameLds)}; return true; }
    if (mode == AZG_PRIOR_HASH) { *out = {search_kernel<0, 4, 2, false>, 4, 4 * sizeof(GameLds)}; return true; }
    if (mode == AZG_PRIOR_PEAKED) { *out = {search_kernel<0, 4, 3, false>, 4, 4 * sizeof(GameLds)}; return true; }
    const bool f16 = cnn_mode == AZG_CNN_F16X3;
#ifdef AZG_ONLY8                 // experiment builds: the 8x8 split-f16 kernels only (compiles in a third of the time)
    if (N != 8 || !f16) return false;
    *out = net_launch<8, AZG_T8, true>();
#if AZG_QUAD8
    {
        using G = typename GeomPick<8, 2, true>::type;
        const size_t region = (((size_t)G::END * 4 + 2 * sizeof(GameLdsT<65>) + 255) / 256) * 256;     // bytes, as in search_body
        *out = {search_quad_kernel_8x8, 4, (size_t)kLdsHead * 4 + 2 * region, 512};
    }
#elif AZG_OCC8 == 2 && AZG_T8 == 2
    out->fn = search_kernel_f16_8x8;
#endif
    return true;
#elif defined(AZG_ONLY15)        // experiment builds: 15x15 only
    if (N != 15 || !f16) return false;
    *out = AZG_DUAL ? dual_launch<15, 1>() : net_launch<15, 1, true>();
    return true;
#else
    switch (N) {
    case 6: *out = f16 ? net_launch<6, 2, true>() : net_launch<6, 2, false>(); break;
    case 8: *out = f16 ? net_launch<8, AZG_T8, true>() : net_launch<8, 2, false>(); break;
    case 15: *out = f16 ? (AZG_DUAL ? dual_launch<15, 1>() : net_launch<15, 1, true>()) : net_launch<15, 1, false>(); break;
    // (16x16: split-f16 arithmetic only -- the float32 images of one leaf, 128 channels x 18 x 18 with the halo, pass the LDS)
    case 7: *out = f16 ? net_launch<7, 2, true>() : net_launch<7, 2, false>(); break;
    case 9: *out = f16 ? (AZG_DUAL ? dual_launch<9, 1>() : net_launch<9, 1, true>()) : net_launch<9, 1, false>(); break;
    case 10: *out = f16 ? (AZG_DUAL ? dual_launch<10, 1>() : net_launch<10, 1, true>()) : net_launch<10, 1, false>(); break;
    case 11: *out = f16 ? (AZG_DUAL ? dual_launch<11, 1>() : net_launch<11, 1, true>()) : net_launch<11, 1, false>(); break;
    case 12: *out = f16 ? (AZG_DUAL ? dual_launch<12, 1>() : net_launch<12, 1, true>()) : net_launch<12, 1, false>(); break;
    case 13: *out = f16 ? (AZG_DUAL ? dual_launch<13, 1>() : net_launch<13, 1, true>()) : net_launch<13, 1, false>(); break;
    case 14: *out = f16 ? (AZG_DUAL ? dual_launch<14, 1>() : net_launch<14, 1, true>()) : net_launch<14, 1, false>(); break;
    case 16: if (!f16) return false; *out = net_launch<16, 1, true>(); break;
    default: return false;
    }
#if AZG_OCC8 == 2 && AZG_T8 == 2
    // same code, register-capped entry points (see AZG_CAPPED_SEARCH_KERNEL)
    if (f16 && N == 6) out->fn = search_kernel_f16_6x6;
    if (f16 && N == 7) out->fn = search_kernel_f16_7x7;
    if (f16 && N == 8) out->fn = search_kernel_f16_8x8;
#endif
    return true;
#endif
}

struct EvalLaunch {
    void (*fn)(NetWeights, const float *, int, float *, float *, int *);
    int T;
    size_t lds;
};

template <int N, int T, bool F16>
static EvalLaunch eval_launch()
{
    using G = typename GeomPick<N, T, F16>::type;
    return {eval_kernel<N, T, F16>, T, (size_t)G::END * 4 + (F16 ? (size_t)kLdsHead * 4 : 0)};
}

static bool pick_eval(int N, int cnn_mode, EvalLaunch *out)
{
    const bool f16 = cnn_mode == AZG_CNN_F16X3;
#ifdef AZG_ONLY8
    if (N != 8 || !f16) return false;
    *out = eval_launch<8, 2, true>();
    return true;
#elif defined(AZG_ONLY15)
    if (N != 15 || !f16) return false;
    *out = eval_launch<15, 1, true>();
    return true;
#else
    switch (N) {
    case 6: *out = f16 ? eval_launch<6, 2, true>() : eval_launch<6, 2, false>(); return true;
    case 8: *out = f16 ? eval_launch<8, 2, true>() : eval_launch<8, 2, false>(); return true;
    case 15: *out = f16 ? eval_launch<15, 1, true>() : eval_launch<15, 1, false>(); return true;
    case 7: *out = f16 ? eval_launch<7, 2, true>() : eval_launch<7, 2, false>(); return true;
    case 9: *out = f16 ? eval_launch<9, 1, true>() : eval_launch<9, 1, false>(); return true;
    case 10: *out = f16 ? eval_launch<10, 1, true>() : eval_launch<10, 1, false>(); return true;
    case 11: *out = f16 ? eval_launch<11, 1, true>() : eval_launch<11, 1, false>(); return true;
    case 12: *out = f16 ? eval_launch<12, 1, true>() : eval_launch<12, 1, false>(); return true;
    case 13: *out = f16 ? eval_launch<13, 1, true>() : eval_launch<13, 1, false>(); return true;
    case 14: *out = f16 ? eval_launch<14, 1, true>() : eval_launch<14, 1, false>(); return true;
    case 16: if (!f16) return false; *out = eval_launch<16, 1, true>(); return true;
    default: return false;
    }
#endif
}

// ---------------------------------------------------------------- mirrors / profiling
static int harvest_events(azg_handle *h, size_t count)
{
    count = std::min(count, h->ev_pending);
    for (size_t i = 0; i < count; i++) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second));
        h->prof_ms += ms;
        h->prof_launches++;
    }
    // events of launches still in flight (pipelined self-play) move to the front
    std::rotate(h->ev_pool.begin(), h->ev_pool.begin() + count, h->ev_pool.begin() + h->ev_pending);
    h->ev_pending -= count;
    return AZG_OK;
}

static size_t mirror_bytes(int B) { return (size_t)B * 37; }

static int mirrors_enqueue(azg_handle *h)
{
    HIPCHK(h, hipMemcpyAsync(h->h_mirror, h->d_mirror, mirror_bytes(h->cfg.n_games), hipMemcpyDeviceToHost, h->stream));
    h->mirror_events = h->ev_pending;
    return AZG_OK;
}

// after the copy of mirrors_enqueue has completed
static int mirrors_unpack(azg_handle *h)
{
    const int B = h->cfg.n_games;
    const long long *l = reinterpret_cast<const long long *>(h->h_mirror);
    const int *i32 = reinterpret_cast<const int *>(h->h_mirror + (size_t)B * 16);
    const signed char *i8 = reinterpret_cast<const signed char *>(h->h_mirror + (size_t)B * 32);
    for (int g = 0; g < B; g++) {
        h->h_game_id[g] = l[g];
        h->h_fin_id[g] = l[B + g];
        h->h_ply[g] = i32[g];
        h->h_scount[g] = i32[B + g];
        h->h_err[g] = i32[2 * B + g];
        h->h_fin_count[g] = i32[3 * B + g];
        h->h_status[g] = i8[g];
        h->h_winner[g] = i8[B + g];
        h->h_fin_flag[g] = i8[2 * B + g];
        h->h_fin_winner[g] = i8[3 * B + g];
        h->h_fin_bank[g] = (unsigned char)i8[4 * B + g];
    }
    h->mirrors_fresh = true;
    int rc = harvest_events(h, h->mirror_events);
    if (rc) return rc;
    for (int g = 0; g < B; g++) {
        if (h->h_err[g] & 1) return fail(h, AZG_EARENA, "slot %d: search tree outgrew its arena (arena_nodes=%d)", g, h->cfg.arena_nodes);
        if (h->h_err[g] & 2) return fail(h, AZG_EMOVE, "slot %d: illegal move", g);
        if (h->h_err[g] & 8) return fail(h, AZG_EHIP, "slot %d: the wave groups of the search kernel lost step (a wait ran out)", g);
        if (h->h_err[g] & 4) return fail(h, AZG_ERANGE, "slot %d: an activation of the net left the half-precision range "
                                         "(|x| > 65504) in the split-f16 convolutions; create the engine with AZG_CNN_F32", g);
    }
    return AZG_OK;
}

static int refresh_mirrors(azg_handle *h)
{
    int rc = mirrors_enqueue(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return mirrors_unpack(h);
}

// ---------------------------------------------------------------- create / destroy
extern "C" const char *azg_last_error(const azg_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" void azg_destroy(azg_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) (void)hipFree(p);
    for (void *p : h->scratch) if (p) (void)hipFree(p);
    if (h->d_weights) (void)hipFree(h->d_weights);
    if (h->h_noise) (void)hipHostFree(h->h_noise);
    if (h->h_u) (void)hipHostFree(h->h_u);
    if (h->h_moves) (void)hipHostFree(h->h_moves);
    if (h->h_mirror) (void)hipHostFree(h->h_mirror);
    if (h->h_static_bad) (void)hipHostFree(h->h_static_bad);
    if (h->tw.block) (void)hipFree(h->tw.block);
    if (h->tw.h_loss) (void)hipHostFree(h->tw.h_loss);
    if (h->tw.stream) (void)hipStreamDestroy(h->tw.stream);
    if (h->sp.h_map) (void)hipHostFree(h->sp.h_map);
    if (h->sp.h_ids) (void)hipHostFree(h->sp.h_ids);
    if (h->sp.ev_mirror) (void)hipEventDestroy(h->sp.ev_mirror);
    if (h->sp.ev_noise) (void)hipEventDestroy(h->sp.ev_noise);
    if (h->sp.copy_stream) (void)hipStreamDestroy(h->sp.copy_stream);
    for (auto &e : h->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static int create_impl(azg_handle *h)
{
    const azg_config &c = h->cfg;
    const int N = c.board_n, S = N * N, B = c.n_games;
    HIPCHK(h, hipSetDevice(c.device));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(h, static_scratch_ptr(c.device, &h->static_scratch));
    // the per-CU matrix gates start free: a launch of an earlier engine that died inside a gated
    // span must not cost this one a bounded wait per CU
    HIPCHK(h, gate_ptrs(c.device, &h->gate_table, &h->gate_giveups));
    HIPCHK(h, hipMemsetAsync(h->gate_table, 0, kGateWords * sizeof(int), h->stream));
    HIPCHK(h, hipHostMalloc((void **)&h->h_static_bad, 64));   // [0] board-step flag, bytes 16..47 arena and re-root counters, [12] CNN range flag
    DevState &st = h->st;
    st.N = N; st.S = S; st.NW = (S + 63) / 64; st.n_in_row = c.n_in_row; st.B = B;
    st.prior_mode = c.prior_mode; st.n_playout = c.n_playout; st.c_puct = c.c_puct;
    {
        // work stack of the re-root copy: AZG_REROOT_STACK in the environment makes it small, so that
        // tests drive the path that parks the stack's older half in the arena tail
        const char *e = getenv("AZG_REROOT_STACK");
        const int want = e ? atoi(e) : kRerootStackMax;
        st.rr_cap = want < 128 ? 128 : (want > kRerootStackMax ? kRerootStackMax : want);
        // AZG_REROOT_COPY=1: every re-root copies the kept subtree (otherwise only when the arena runs low)
        const char *a = getenv("AZG_REROOT_COPY");
        st.rr_always = a && atoi(a) != 0;
        // AZG_SEARCH_PARTS: launches a move's playouts are spread over (1 ... 8), see search_block
        const char *sp = getenv("AZG_SEARCH_PARTS");
        h->search_parts = sp ? std::max(1, std::min(64, atoi(sp))) : 0;    // (0: chosen below, once the kernel is known)
    }
    if (h->cfg.arena_nodes == 0) {
        // Default arena: room for the trees of 16 searches (8, 4 when memory is short).  A re-root
        // leaves the kept subtree where it is while the arena has room for the next search behind
        // it, so a roomy arena means a game compacts its tree once in many moves instead of copying
        // it every move; the five arrays of all games stay within 64 GB and a quarter of the free memory.
        size_t free_b = 0, total_b = 0;
        HIPCHK(h, hipMemGetInfo(&free_b, &total_b));
        const size_t budget = std::min<size_t>((size_t)64 << 30, free_b / 4);
        int mult = 16;
        while (mult > 4 && (size_t)B * 2 * (1 + ((size_t)mult * c.n_playout + 64) * S) * 24 > budget) mult /= 2;
        h->cfg.arena_nodes = mult * c.n_playout + 64;
        // within a tenth of the size that holds ANY tree of the configuration: take that size, and
        // the shared large arenas below are not needed at all
        const long long any = 2 + (long long)c.n_playout * ((S + 1) / 2) * ((S + 1) / 2);
        if ((1 + (long long)h->cfg.arena_nodes * S) * 10 >= any * 9)
            h->cfg.arena_nodes = (int)((any + S - 1) / S);
    }
    st.cap = 1 + h->cfg.arena_nodes * S;
    const size_t E = (size_t)B * 2 * st.cap;
    int rc;
    if ((rc = dev_alloc(h, &st.bb, (size_t)B * 2 * MAXW))) return rc;
    if ((rc = dev_alloc(h, &h->d_mirror, mirror_bytes(B) + 16))) return rc;
    st.game_id = reinterpret_cast<long long *>(h->d_mirror);
    st.fin_id = st.game_id + B;
    st.ply = reinterpret_cast<int *>(h->d_mirror + (size_t)B * 16);
    st.s_count = st.ply + B;
    st.err = st.ply + 2 * B;
    st.fin_count = st.ply + 3 * B;
    st.status = reinterpret_cast<signed char *>(h->d_mirror + (size_t)B * 32);
    st.winner = st.status + B;
    st.fin_flag = st.status + 2 * B;
    st.fin_winner = st.status + 3 * B;
    st.fin_bank = reinterpret_cast<unsigned char *>(st.status + 4 * B);
    if ((rc = dev_alloc(h, &st.s_bank, B))) return rc;
    if ((rc = dev_alloc(h, &st.next_k, 1))) return rc;
    if ((rc = dev_alloc(h, &st.last_move, B))) return rc;
    if ((rc = dev_alloc(h, &st.first_player, B))) return rc;
    if ((rc = dev_alloc(h, &st.eP, E, false))) return rc;
    if ((rc = dev_alloc(h, &st.eN, E, false))) return rc;
    if ((rc = dev_alloc(h, &st.eQ, E, false))) return rc;
    if ((rc = dev_alloc(h, &st.eChild, E, false))) return rc;
    if ((rc = dev_alloc(h, &st.eMeta, E, false))) return rc;
    if ((rc = dev_alloc(h, &st.used, B))) return rc;
    // large arenas (see DevState): as many as asked for; by default one per four game slots (a net
    // with peaked priors keeps most of its tree from move to move, and a game holds its large arena
    // until it ends), at least 16, within 16 GB
    st.cap_big = 2 + c.n_playout * ((S + 1) / 2) * ((S + 1) / 2);
    st.n_big = c.big_arenas < 0 ? 0 : std::min(B, c.big_arenas > 0 ? c.big_arenas : std::max(16, B / 4));
    if (c.big_arenas <= 0)
        while (st.n_big > 0 && (size_t)st.n_big * 2 * st.cap_big * 24 > ((size_t)16 << 30)) st.n_big--;
    if (st.cap_big <= st.cap) st.n_big = 0;                  // the standard arenas already hold any tree
    const size_t EB = (size_t)st.n_big * 2 * st.cap_big;
    if ((rc = dev_alloc(h, &st.big_of, B, false))) return rc;
    if ((rc = dev_alloc(h, &st.big_owner, st.n_big, false))) return rc;
    HIPCHK(h, hipMemsetAsync(st.big_of, 0xFF, (size_t)B * sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(st.big_owner, 0xFF, std::max<size_t>(st.n_big, 1) * sizeof(int), h->stream));
    if ((rc = dev_alloc(h, &st.bP, EB, false))) return rc;
    if ((rc = dev_alloc(h, &st.bN, EB, false))) return rc;
    if ((rc = dev_alloc(h, &st.bQ, EB, false))) return rc;
    if ((rc = dev_alloc(h, &st.bChild, EB, false))) return rc;
    if ((rc = dev_alloc(h, &st.bMeta, EB, false))) return rc;
    if ((rc = dev_alloc(h, &st.arena_stats, 4))) return rc;
    if ((rc = dev_alloc(h, &st.cur, B))) return rc;
    if ((rc = dev_alloc(h, &st.evals, B))) return rc;
    if ((rc = dev_alloc(h, &st.s_bb, (size_t)B * 2 * S * 2 * MAXW, false))) return rc;
    if ((rc = dev_alloc(h, &st.s_last, (size_t)B * 2 * S, false))) return rc;
    if ((rc = dev_alloc(h, &st.s_ply, (size_t)B * 2 * S, false))) return rc;
    if ((rc = dev_alloc(h, &st.s_pi, (size_t)B * 2 * S * S, false))) return rc;
    if ((rc = dev_alloc(h, &st.drained, B))) return rc;
    if ((rc = dev_alloc(h, &st.stamps, kStampWords))) return rc;
    if ((rc = dev_alloc(h, &h->d_noise, (size_t)B * S))) return rc;
    if ((rc = dev_alloc(h, &h->d_u, B))) return rc;
    if ((rc = dev_alloc(h, &h->d_pi, (size_t)B * S))) return rc;
    if ((rc = dev_alloc(h, &h->d_moves, B))) return rc;
    if ((rc = dev_alloc(h, &h->d_given, B))) return rc;
    HIPCHK(h, hipHostMalloc((void **)&h->h_noise, (size_t)B * S * sizeof(double)));
    HIPCHK(h, hipHostMalloc((void **)&h->h_u, (size_t)B * sizeof(double)));
    HIPCHK(h, hipHostMalloc((void **)&h->h_moves, (size_t)B * sizeof(int)));
    HIPCHK(h, hipHostMalloc((void **)&h->h_mirror, mirror_bytes(B) + 16));
    h->h_ply.assign(B, 0); h->h_scount.assign(B, 0); h->h_err.assign(B, 0); h->h_fin_count.assign(B, 0);
    h->h_status.assign(B, 2); h->h_winner.assign(B, -1); h->h_drained.assign(B, 1);
    h->h_fin_flag.assign(B, 0); h->h_fin_winner.assign(B, -1); h->h_fin_bank.assign(B, 0);
    h->h_game_id.assign(B, 0); h->h_fin_id.assign(B, 0);
    h->rng.resize(B);
    // every slot starts idle (status 2) until azg_reset_games
    HIPCHK(h, hipMemsetAsync(st.status, 2, B, h->stream));
    HIPCHK(h, hipMemsetAsync(st.drained, 1, B, h->stream));
    // opt in to the large dynamic LDS of the CNN kernels
    SearchLaunch sl;
    if (pick_search(N, c.prior_mode, c.cnn_mode, &sl))
        HIPCHK(h, hipFuncSetAttribute((const void *)sl.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl.lds));
    EvalLaunch el;
    if (pick_eval(N, c.cnn_mode, &el))
        HIPCHK(h, hipFuncSetAttribute((const void *)el.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)el.lds));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return AZG_OK;
}

extern "C" int azg_create(const azg_config *cfg, azg_handle **out)
{
    if (!cfg || !out) return fail(nullptr, AZG_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->board_n < 3 || cfg->board_n > 16) return fail(nullptr, AZG_EINVAL, "board_n must be in [3,16], got %d", cfg->board_n);
    if (cfg->n_in_row < 1 || cfg->n_in_row > cfg->board_n)
        return fail(nullptr, AZG_EINVAL, "board width and height can not be less than n_in_row (%d)", cfg->n_in_row);
    if (cfg->n_games < 1) return fail(nullptr, AZG_EINVAL, "n_games must be positive");
    if (cfg->n_playout < 1) return fail(nullptr, AZG_EINVAL, "n_playout must be positive");
    if (cfg->prior_mode < 0 || cfg->prior_mode > 3) return fail(nullptr, AZG_EINVAL, "bad prior_mode %d", cfg->prior_mode);
    SearchLaunch sl;
    if (cfg->cnn_mode != AZG_CNN_F16X3 && cfg->cnn_mode != AZG_CNN_F32) return fail(nullptr, AZG_EINVAL, "bad cnn_mode %d", cfg->cnn_mode);
    if (!pick_search(cfg->board_n, cfg->prior_mode, cfg->cnn_mode, &sl))
        return fail(nullptr, AZG_EINVAL, "no CNN search kernel is built for board_n=%d with this cnn_mode (split-f16: 6..16, float32: 6, 8, 15)", cfg->board_n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, AZG_EHIP, "no HIP device available");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, AZG_EINVAL, "device %d out of range (%d devices)", cfg->device, ndev);
    azg_handle *h = new azg_handle();
    h->cfg = *cfg;
    if (h->cfg.arena_nodes <= 0) h->cfg.arena_nodes = 0;      // chosen in create_impl, with the device current
    int rc = create_impl(h);
    if (rc != AZG_OK) {
        g_create_error = h->err;
        azg_destroy(h);
        return rc;
    }
    *out = h;
    return AZG_OK;
}

extern "C" void *azg_stream(azg_handle *h) { return h ? (void *)h->stream : nullptr; }

// ---------------------------------------------------------------- weights
static void pack_conv(const float *w, int cin, int cout, float *out)
{
    const int KS = (cin / 2) * 9;
    for (int cot = 0; cot < cout / 32; cot++)
        for (int ks = 0; ks < KS; ks++)
            for (int lane = 0; lane < 64; lane++) {
                const int co = cot * 32 + (lane & 31), ci = 2 * (ks / 9) + (lane >> 5), tap = ks % 9;
                out[((size_t)cot * KS + ks) * 64 + lane] = w[((size_t)co * cin + ci) * 9 + tap];
            }
}

// Split-f16 image of a conv kernel for azg_cnn16.hpp: per (cotile, kstep = tap*CG + cg) and
// lane, 8 high halves then 8 low halves of W[co][ci = cg*16 + 8*(lane>>5) + j][tap] * 2^8.
static void pack_conv16(const float *w, int cin, int cinp, int cout, _Float16 *out)
{
    const int CG = cinp / 16, KS = 9 * CG;
    for (int cot = 0; cot < cout / 32; cot++)
        for (int ks = 0; ks < KS; ks++)
            for (int lane = 0; lane < 64; lane++) {
                const int tap = ks / CG, cg = ks % CG;
                const int co = cot * 32 + (lane & 31);
                _Float16 *dst = out + ((size_t)cot * KS + ks) * 1024;
                for (int j = 0; j < 8; j++) {
                    const int ci = cg * 16 + 8 * (lane >> 5) + j;
                    const float v = ci < cin ? w[((size_t)co * cin + ci) * 9 + tap] * kWScale : 0.0f;
                    const _Float16 hi = (_Float16)v;
                    dst[wsplit_off(lane, j)] = hi;
                    dst[wsplit_off(lane, 8 + j)] = (_Float16)(v - (float)hi);
                }
            }
}

// conv3 in Winograd F(2x2, 3x3) form for azg_cnn16.hpp's conv3_wino_f16: U = G g G^T of every
// (co, ci) kernel, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], computed in double; per (cotile,
// kstep = xi*CG + cg) with xi = 4a + b the element of the 4x4 transform domain, and lane, 8 high halves
// then 8 low halves of U[a][b](co = cotile*32 + (lane&31), ci = cg*16 + 8*(lane>>5) + j) * 2^8.
// Returns the largest |U| (the caller checks the half-precision range).
static double pack_wino16(const float *w, int cin, int cout, _Float16 *out)
{
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int CG = cin / 16, KS = 16 * CG;
    double umax = 0.0;
    for (int cot = 0; cot < cout / 32; cot++)
        for (int ks = 0; ks < KS; ks++)
            for (int lane = 0; lane < 64; lane++) {
                const int xi = ks / CG, cg = ks % CG, a = xi >> 2, b = xi & 3;
                const int co = cot * 32 + (lane & 31);
                _Float16 *dst = out + ((size_t)cot * KS + ks) * 1024;
                for (int j = 0; j < 8; j++) {
                    const int ci = cg * 16 + 8 * (lane >> 5) + j;
                    const float *g = w + ((size_t)co * cin + ci) * 9;
                    double u = 0.0;
                    for (int r = 0; r < 3; r++)
                        for (int c = 0; c < 3; c++) u += Gm[a][r] * (double)g[3 * r + c] * Gm[b][c];
                    umax = std::max(umax, std::fabs(u));
                    const float v = (float)u * kWScale;
                    const _Float16 hi = (_Float16)v;
                    dst[wsplit_off(lane, j)] = hi;
                    dst[wsplit_off(lane, 8 + j)] = (_Float16)(v - (float)hi);
                }
            }
    return umax;
}

// Split-f16 image of conv1 for azg_cnn16.hpp's conv1_packed_f16: k = tap*4 + plane in three k-steps
// of 16; per k-step and lane 8 high then 8 low halves of W[co = lane&31][plane = j%4][tap = 4*ks +
// 2*(lane>>5) + j/4] * 2^8, zero for tap >= 9.
static void pack_conv1_16(const float *w /*[32][4][3][3]*/, _Float16 *out)
{
    for (int ks = 0; ks < 3; ks++)
        for (int lane = 0; lane < 64; lane++) {
            const int co = lane & 31;
            _Float16 *dst = out + (size_t)ks * 1024;
            for (int j = 0; j < 8; j++) {
                const int tap = 4 * ks + 2 * (lane >> 5) + j / 4, ci = j % 4;
                const float v = tap < 9 ? w[((size_t)co * 4 + ci) * 9 + tap] * kWScale : 0.0f;
                const _Float16 hi = (_Float16)v;
                dst[wsplit_off(lane, j)] = hi;
                dst[wsplit_off(lane, 8 + j)] = (_Float16)(v - (float)hi);
            }
        }
}

// Split-f16 image of a fully connected matrix W[O][K] (row-major) for azg_cnn16.hpp
// (v_mfma_f32_16x16x32_f16 B operand): per (output tile of 16, k-step of 32) and lane, 8 high
// halves then 8 low halves of W[o = tile*16 + (lane&15)][k = ks*32 + 8*(lane>>4) + j] * 2^8; rows >= O
// and columns >= K are zero.
static void pack_fc16(const float *w, int O, int K, int otiles, int ksteps, _Float16 *out)
{
    for (int ot = 0; ot < otiles; ot++)
        for (int ks = 0; ks < ksteps; ks++)
            for (int lane = 0; lane < 64; lane++) {
                const int o = ot * 16 + (lane & 15);
                _Float16 *dst = out + ((size_t)ot * ksteps + ks) * 1024;
                for (int j = 0; j < 8; j++) {
                    const int k = ks * 32 + 8 * (lane >> 4) + j;
                    const float v = (o < O && k < K) ? w[(size_t)o * K + k] * kWScale : 0.0f;
                    const _Float16 hi = (_Float16)v;
                    dst[wsplit_off(lane, j)] = hi;
                    dst[wsplit_off(lane, 8 + j)] = (_Float16)(v - (float)hi);
                }
            }
}

// Split-f16 image of the two 1x1 head convolutions (rows 0-3 act_conv1, 4-5 val_conv1) for
// azg_cnn16.hpp.  conv3's accumulators feed the head MFMA directly as its B operand, so the k
// index follows the accumulator layout: entry ((wave*2 + s)*2 + kb)*8 + c holds, for j = 0..7,
// W[c][co = wave*32 + acc_row(8*s + j, kb)] * 2^8 as 8 high halves then 8 low halves; rows 6, 7 are zero.
static void pack_head16(const float *wh /*[6][128]*/, _Float16 *out)
{
    for (int wave = 0; wave < 4; wave++)
        for (int s = 0; s < 2; s++)
            for (int kb = 0; kb < 2; kb++)
                for (int c = 0; c < 8; c++) {
                    _Float16 *dst = out + ((((size_t)wave * 2 + s) * 2 + kb) * 8 + c) * 16;
                    for (int j = 0; j < 8; j++) {
                        const int r = 8 * s + j;
                        const int co = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                        const float v = c < 6 ? wh[c * 128 + co] * kWScale : 0.0f;
                        const _Float16 hi = (_Float16)v;
                        dst[j] = hi;
                        dst[8 + j] = (_Float16)(v - (float)hi);
                    }
                }
}

extern "C" int azg_set_weights(azg_handle *h, const float *const ptrs[16], int layout)
{
    if (!h || !ptrs) return fail(h, AZG_EINVAL, "null argument");
    if (layout != AZG_LAYOUT_TORCH && layout != AZG_LAYOUT_THEANO) return fail(h, AZG_EINVAL, "bad layout %d", layout);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int S = h->st.S;
    const size_t sizes[16] = {32 * 4 * 9, 32, 64 * 32 * 9, 64, 128 * 64 * 9, 128, 4 * 128, 4,
                              (size_t)S * 4 * S, (size_t)S, 2 * 128, 2, (size_t)64 * 2 * S, 64, 64, 1};
    std::vector<std::vector<float>> t(16);
    for (int i = 0; i < 16; i++) {
        if (!ptrs[i]) return fail(h, AZG_EINVAL, "weight tensor %d is null", i);
        t[i].resize(sizes[i]);
        HIPCHK(h, hipMemcpy(t[i].data(), ptrs[i], sizes[i] * sizeof(float), hipMemcpyDefault));
    }
    if (layout == AZG_LAYOUT_THEANO) {
        // conv kernels are stored for true convolution (rotated 180 degrees) and FC
        // matrices as (in, out): policy_value_net_numpy.py:27-29,43-45
        for (int i : {0, 2, 4}) {
            const size_t n = sizes[i] / 9;
            for (size_t k = 0; k < n; k++) std::reverse(t[i].begin() + k * 9, t[i].begin() + k * 9 + 9);
        }
        auto transpose = [&](int i, int rows_in, int cols_out) {
            std::vector<float> o(sizes[i]);
            for (int r = 0; r < rows_in; r++)
                for (int c = 0; c < cols_out; c++) o[(size_t)c * rows_in + r] = t[i][(size_t)r * cols_out + c];
            t[i].swap(o);
        };
        transpose(8, 4 * S, S);
        transpose(12, 2 * S, 64);
        transpose(14, 64, 1);
    }
    // engine layout
    auto al = [](size_t n) { return (n + 63) / 64 * 64; };
    size_t off[16], total = 0;
    size_t off16[3];
    const size_t psz[14] = {(size_t)1 * 18 * 64, (size_t)2 * 144 * 64, (size_t)4 * 288 * 64, 32, 64, 128,
                            6 * 128, 6, (size_t)4 * S * S, (size_t)S, (size_t)((2 * S + 3) / 4) * 64 * 4, 64, 64, 1};
    for (int i = 0; i < 14; i++) { off[i] = total; total += al(psz[i]); }
    // split-f16 conv images: (cotiles * 9*CG k-steps * 64 lanes * 16 halves) / 2 floats
    const size_t psz16[3] = {(size_t)3 * 64 * 8, (size_t)2 * 18 * 64 * 8, (size_t)4 * 36 * 64 * 8};
    for (int i = 0; i < 3; i++) { off16[i] = total; total += al(psz16[i]); }
    // split-f16 FC images: k-steps padded to a multiple of 4 (one quarter per wave)
    const int otp = (S + 15) / 16, ksp = ((4 * S + 31) / 32 + 3) / 4 * 4, ksv = ((2 * S + 31) / 32 + 3) / 4 * 4;
    const size_t offfc[2] = {total, total + al((size_t)otp * ksp * 64 * 8)};
    total = offfc[1] + al((size_t)4 * ksv * 64 * 8);
    const size_t offh16 = total;
    total += 1024;                                           // 4 KB of head weights as halves
    // row-per-input images of the two wide FC layers (sparse-input FC of the one-leaf kernels)
    const size_t offpk = total, offvk = offpk + (size_t)4 * S * 256;
    total = offvk + (size_t)2 * S * 64;
    const size_t offw3 = total;                              // conv3 in Winograd form: 4 cotiles * 64 k-steps * 1024 halves
    if (AZG_WINO3) total += (size_t)4 * 64 * 512;
    if (h->cfg.cnn_mode == AZG_CNN_F16X3) {
        // halves hold w * 2^8: larger weights would overflow to infinity
        for (int i : {0, 2, 4, 6, 8, 10, 12})
            for (float v : t[i])
                if (!(std::fabs(v) < 240.0f))
                    return fail(h, AZG_EINVAL, "weight tensor %d holds |w| >= 240 (or NaN): out of range for the "
                                "split-f16 convolutions, create the engine with AZG_CNN_F32", i);
    }
    std::vector<float> host(total, 0.f);
    pack_conv(t[0].data(), 4, 32, &host[off[0]]);
    pack_conv(t[2].data(), 32, 64, &host[off[1]]);
    pack_conv(t[4].data(), 64, 128, &host[off[2]]);
    pack_conv1_16(t[0].data(), reinterpret_cast<_Float16 *>(&host[off16[0]]));
    pack_conv16(t[2].data(), 32, 32, 64, reinterpret_cast<_Float16 *>(&host[off16[1]]));
    pack_conv16(t[4].data(), 64, 64, 128, reinterpret_cast<_Float16 *>(&host[off16[2]]));
    if (AZG_WINO3) {
        const double umax = pack_wino16(t[4].data(), 64, 128, reinterpret_cast<_Float16 *>(&host[offw3]));
        if (h->cfg.cnn_mode == AZG_CNN_F16X3 && h->cfg.board_n <= 8 && !(umax < 240.0))
            return fail(h, AZG_EINVAL, "conv3 holds kernels whose Winograd transform reaches |u| >= 240: out of range for the "
                        "split-f16 convolutions, create the engine with AZG_CNN_F32");
    }
    pack_fc16(t[8].data(), S, 4 * S, otp, ksp, reinterpret_cast<_Float16 *>(&host[offfc[0]]));
    pack_fc16(t[12].data(), 64, 2 * S, 4, ksv, reinterpret_cast<_Float16 *>(&host[offfc[1]]));
    std::copy(t[1].begin(), t[1].end(), &host[off[3]]);
    std::copy(t[3].begin(), t[3].end(), &host[off[4]]);
    std::copy(t[5].begin(), t[5].end(), &host[off[5]]);
    std::copy(t[6].begin(), t[6].end(), &host[off[6]]);             // act_conv1 rows 0-3
    std::copy(t[10].begin(), t[10].end(), &host[off[6] + 4 * 128]); // val_conv1 rows 4-5
    std::copy(t[7].begin(), t[7].end(), &host[off[7]]);
    pack_head16(&host[off[6]], reinterpret_cast<_Float16 *>(&host[offh16]));
    std::copy(t[11].begin(), t[11].end(), &host[off[7] + 4]);
    for (int a = 0; a < S; a++)
        for (int k = 0; k < 4 * S; k++) host[off[8] + ((size_t)(k / 4) * S + a) * 4 + k % 4] = t[8][(size_t)a * 4 * S + k];
    std::copy(t[9].begin(), t[9].end(), &host[off[9]]);
    for (int o = 0; o < 64; o++)
        for (int k = 0; k < 2 * S; k++) host[off[10] + ((size_t)(k / 4) * 64 + o) * 4 + k % 4] = t[12][(size_t)o * 2 * S + k];
    for (int o = 0; o < S; o++)
        for (int k = 0; k < 4 * S; k++) host[offpk + (size_t)k * 256 + o] = t[8][(size_t)o * 4 * S + k];
    for (int o = 0; o < 64; o++)
        for (int k = 0; k < 2 * S; k++) host[offvk + (size_t)k * 64 + o] = t[12][(size_t)o * 2 * S + k];
    std::copy(t[13].begin(), t[13].end(), &host[off[11]]);
    std::copy(t[14].begin(), t[14].end(), &host[off[12]]);
    host[off[13]] = t[15][0];
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!h->d_weights) HIPCHK(h, hipMalloc((void **)&h->d_weights, total * sizeof(float)));
    HIPCHK(h, hipMemcpy(h->d_weights, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
    NetWeights &nw = h->st.net;
    const float *d = h->d_weights;
    nw.wconv[0] = d + off[0]; nw.wconv[1] = d + off[1]; nw.wconv[2] = d + off[2];
    nw.w16[0] = d + off16[0]; nw.w16[1] = d + off16[1]; nw.w16[2] = d + off16[2];
    nw.w16w3 = AZG_WINO3 ? d + offw3 : nullptr;
    nw.wfc16[0] = d + offfc[0]; nw.wfc16[1] = d + offfc[1];
    nw.whead16 = d + offh16;
    nw.bconv[0] = d + off[3]; nw.bconv[1] = d + off[4]; nw.bconv[2] = d + off[5];
    nw.whead = d + off[6]; nw.bhead = d + off[7];
    nw.wpfc = d + off[8]; nw.bpfc = d + off[9];
    nw.wvfc1 = d + off[10]; nw.bvfc1 = d + off[11];
    nw.wvfc2 = d + off[12]; nw.bvfc2 = d + off[13];
    nw.wpfc_k = d + offpk; nw.wvfc1_k = d + offvk;
    h->have_weights = true;
    return AZG_OK;
}

// ---------------------------------------------------------------- game control
#define NOT_WHILE_PIPELINED(h) \
    do { if ((h)->sp.active) return fail(h, AZG_ESTATE, "not valid between azg_selfplay_begin and azg_selfplay_end"); } while (0)

extern "C" int azg_reset_games(azg_handle *h, const uint8_t *mask, const uint32_t *seeds, const int64_t *game_ids)
{
    if (!h) return AZG_EINVAL;
    NOT_WHILE_PIPELINED(h);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_games;
    void *d_mask = nullptr, *d_ids = nullptr;
    int rc;
    if (mask) {
        if ((rc = scratch_get(h, 0, B, &d_mask))) return rc;
        HIPCHK(h, hipMemcpyAsync(d_mask, mask, B, hipMemcpyHostToDevice, h->stream));
    }
    if (game_ids) {
        if ((rc = scratch_get(h, 1, B * sizeof(long long), &d_ids))) return rc;
        HIPCHK(h, hipMemcpyAsync(d_ids, game_ids, B * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    }
    hipLaunchKernelGGL(reset_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->st,
                       (const unsigned char *)d_mask, (const long long *)d_ids);
    HIPCHK(h, hipGetLastError());
    for (int g = 0; g < B; g++)
        if (!mask || mask[g]) {
            h->rng[g].seed(seeds ? seeds[g] : (uint32_t)g);
            h->h_drained[g] = 0;
        }
    return refresh_mirrors(h);
}

extern "C" int azg_set_position(azg_handle *h, int slot, const uint64_t stones1[4], const uint64_t stones2[4],
                                int last_move, int current_player)
{
    if (!h || !stones1 || !stones2) return fail(h, AZG_EINVAL, "null argument");
    NOT_WHILE_PIPELINED(h);
    if (slot < 0 || slot >= h->cfg.n_games) return fail(h, AZG_EINVAL, "slot %d out of range", slot);
    if (current_player != 1 && current_player != 2) return fail(h, AZG_EINVAL, "current_player must be 1 or 2");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    unsigned long long w[2 * MAXW];
    int ply = 0;
    for (int i = 0; i < MAXW; i++) {
        if (stones1[i] & stones2[i]) return fail(h, AZG_EINVAL, "a cell is owned by both players");
        w[i] = stones1[i]; w[MAXW + i] = stones2[i];
        ply += __builtin_popcountll(stones1[i]) + __builtin_popcountll(stones2[i]);
    }
    const int first = (ply & 1) ? 3 - current_player : current_player;
    const signed char live = 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->st.bb + (size_t)slot * 2 * MAXW, w, sizeof(w), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->st.ply + slot, &ply, sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->st.last_move + slot, &last_move, sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->st.first_player + slot, &first, sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->st.status + slot, &live, 1, hipMemcpyHostToDevice));
    return refresh_mirrors(h);
}

// launches a move's playouts are spread over
static int search_parts_of(const azg_handle *h)
{
    int want = h->search_parts;
    if (want <= 0) want = h->cfg.prior_mode != AZG_PRIOR_NET ? 1 : (h->cfg.board_n <= 8 ? 8 : 2);
    return std::max(1, std::min(want, h->st.n_playout));
}

extern "C" int azg_search_parts(azg_handle *h) { return h ? search_parts_of(h) : AZG_EINVAL; }

extern "C" int azg_set_search_parts(azg_handle *h, int parts)
{
    if (!h) return AZG_EINVAL;
    if (parts < 0 || parts > 64) return fail(h, AZG_EINVAL, "search parts must be 0 (default) ... 64");
    h->search_parts = parts;
    return AZG_OK;
}

extern "C" int azg_search(azg_handle *h)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->cfg.prior_mode == AZG_PRIOR_NET && !h->have_weights)
        return fail(h, AZG_ENOWEIGHTS, "azg_set_weights must be called before a net-driven search");
    SearchLaunch sl;
    if (!pick_search(h->cfg.board_n, h->cfg.prior_mode, h->cfg.cnn_mode, &sl)) return fail(h, AZG_EINVAL, "no search kernel");
    const int grid = (h->cfg.n_games + sl.T - 1) / sl.T;
    if (h->ev_pending == h->ev_pool.size()) {
        hipEvent_t a, b;
        HIPCHK(h, hipEventCreate(&a));
        HIPCHK(h, hipEventCreate(&b));
        h->ev_pool.emplace_back(a, b);
    }
    auto &ev = h->ev_pool[h->ev_pending];
    HIPCHK(h, hipEventRecord(ev.first, h->stream));
    // The playouts of the move in search_parts launches back to back.  Between two playouts the whole
    // search state of a game is in global memory, so the trees do not depend on where the move is cut
    // (all parity tests run with 1 and with 4 parts).  Measured, same box, interleaved rounds, 8x8 /
    // 400 / 1024: 4 parts instead of 1: 11.40 -> 11.06 ms per move on one box, 10.98 -> 10.99 on another
    // (the single launch varies between boxes, the split one does not); with a trained net 11.50 -> 11.05
    // and 11.30 -> 10.72; 8 parts 11.07 / 10.89, 16 parts 11.13 / 11.0.  Kernels of other streams slip in
    // between two parts (beside a search launch they find no room on any CU): a learning run with one
    // policy update per move step plays 2 731 / 2 788 / 2 906 games/s including training with 4 / 8 / 16
    // parts.  So 8 by default, and the training loop asks for 16.  Dealing the games to other XCDs
    // in every part changed nothing on top (every launch still ends with its slowest XCD).  15x15 / 800 /
    // 512 (two-wave-group kernel): the search itself 43.6 -> 43.7 (2 parts) -> 44.0 ms (4), but the copy
    // kernels of the noise upload, which find no room beside that kernel, slip in between two parts:
    // move step - search 0.54 -> 0.14 ms, so 2 parts there.
    const int parts = search_parts_of(h);
    for (int p = 0; p < parts; p++) {
        DevState part = h->st;
        part.n_playout = h->st.n_playout / parts + (p < h->st.n_playout % parts ? 1 : 0);
        hipLaunchKernelGGL(sl.fn, dim3(grid), dim3(sl.threads), sl.lds, h->stream, part);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(ev.second, h->stream));
    h->ev_pending++;
    h->mirrors_fresh = false;
    return AZG_OK;
}

extern "C" int azg_root_stats(azg_handle *h, int32_t *visits, double *q, double *prior, int32_t *root_visits)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_games, S = h->st.S;
    void *dv, *dq, *dp, *dr;
    int rc;
    if ((rc = scratch_get(h, 2, (size_t)B * S * sizeof(int), &dv))) return rc;
    if ((rc = scratch_get(h, 3, (size_t)B * S * sizeof(double), &dq))) return rc;
    if ((rc = scratch_get(h, 4, (size_t)B * S * sizeof(double), &dp))) return rc;
    if ((rc = scratch_get(h, 5, (size_t)B * sizeof(int), &dr))) return rc;
    hipLaunchKernelGGL(root_stats_kernel, dim3((B + 3) / 4), dim3(256), 0, h->stream, h->st, (int *)dv,
                       (double *)dq, (double *)dp, (int *)dr);
    HIPCHK(h, hipGetLastError());
    if (visits) HIPCHK(h, hipMemcpyAsync(visits, dv, (size_t)B * S * sizeof(int), hipMemcpyDefault, h->stream));
    if (q) HIPCHK(h, hipMemcpyAsync(q, dq, (size_t)B * S * sizeof(double), hipMemcpyDefault, h->stream));
    if (prior) HIPCHK(h, hipMemcpyAsync(prior, dp, (size_t)B * S * sizeof(double), hipMemcpyDefault, h->stream));
    if (root_visits) HIPCHK(h, hipMemcpyAsync(root_visits, dr, (size_t)B * sizeof(int), hipMemcpyDefault, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return harvest_events(h, h->ev_pending);
}

static void make_noise(azg_handle *h)
{
    const int B = h->cfg.n_games, S = h->st.S;
    auto work = [&](int g0, int g1) {
        for (int g = g0; g < g1; g++) {
            if (h->h_status[g] != 0) continue;
            const int K = S - h->h_ply[g];
            if (K <= 0) continue;
            // np.random.dirichlet(0.1*np.ones(K)) then the uniform of np.random.choice
            h->rng[g].dirichlet(0.1, K, h->h_noise + (size_t)g * S);
            h->h_u[g] = h->rng[g].next_double();
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    int nthr = (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    if (B < 64) nthr = 1;
    // chunks of 16 games handed out through a counter: games near their end draw far fewer
    // gamma variates than games at their start, so equal ranges would not be equal work
    const int per = 16, chunks = (B + per - 1) / per;
    const std::function<void(int)> job = [&](int c) { work(c * per, std::min(B, (c + 1) * per)); };
    h->noise_pool.run(chunks, nthr, job);
    h->noise_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->noise_steps++;
}

static int advance_common(azg_handle *h, double temp, const int32_t *given, int flags, int32_t *moves_out, double *pi_out)
{
    NOT_WHILE_PIPELINED(h);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_games, S = h->st.S;
    if (!(temp > 0.0)) return fail(h, AZG_EINVAL, "temp must be positive");
    if (flags & ADV_SAMPLE) {
        make_noise(h);          // runs while the search kernel is still executing
        HIPCHK(h, hipMemcpyAsync(h->d_noise, h->h_noise, (size_t)B * S * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_u, h->h_u, (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {
        for (int g = 0; g < B; g++) h->h_moves[g] = given[g];
        HIPCHK(h, hipMemcpyAsync(h->d_given, h->h_moves, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    hipLaunchKernelGGL(advance_kernel, dim3((B + 3) / 4), dim3(256), 0, h->stream, h->st, temp, h->d_noise,
                       h->d_u, (flags & ADV_SAMPLE) ? (const int *)nullptr : h->d_given, flags, h->d_moves,
                       pi_out ? h->d_pi : (double *)nullptr, 0LL, 1LL, 0LL);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->h_moves, h->d_moves, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (pi_out) HIPCHK(h, hipMemcpyAsync(pi_out, h->d_pi, (size_t)B * S * sizeof(double), hipMemcpyDefault, h->stream));
    int rc = refresh_mirrors(h);
    if (moves_out) for (int g = 0; g < B; g++) moves_out[g] = h->h_moves[g];
    return rc;
}

extern "C" int azg_sample_and_advance(azg_handle *h, double temp, int32_t *moves_out, double *pi_out)
{
    if (!h) return AZG_EINVAL;
    return advance_common(h, temp, nullptr, ADV_SAMPLE | ADV_RECORD | ADV_BOARD | ADV_REROOT, moves_out, pi_out);
}

extern "C" int azg_advance(azg_handle *h, const int32_t *moves, int record, double temp)
{
    if (!h || !moves) return fail(h, AZG_EINVAL, "null argument");
    return advance_common(h, temp, moves, (record ? ADV_RECORD : 0) | ADV_BOARD | ADV_REROOT, nullptr, nullptr);
}

extern "C" int azg_update_with_move(azg_handle *h, const int32_t *moves)
{
    if (!h || !moves) return fail(h, AZG_EINVAL, "null argument");
    NOT_WHILE_PIPELINED(h);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_games;
    for (int g = 0; g < B; g++) h->h_moves[g] = moves[g];
    HIPCHK(h, hipMemcpyAsync(h->d_given, h->h_moves, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(reroot_kernel, dim3((B + 3) / 4), dim3(256), 0, h->stream, h->st, (const int *)h->d_given);
    HIPCHK(h, hipGetLastError());
    return refresh_mirrors(h);
}

extern "C" int azg_status(azg_handle *h, uint8_t *ended, int8_t *winner, int32_t *plies)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // every call that changes a game's state ends with a refresh, so the host copy is
    // normally current; only a search (which can raise the arena-overflow flag) is not
    int rc = h->mirrors_fresh ? AZG_OK : refresh_mirrors(h);
    const int B = h->cfg.n_games;
    for (int g = 0; g < B; g++) {
        if (ended) ended[g] = h->h_status[g] == 1;
        if (winner) winner[g] = h->h_winner[g];
        if (plies) plies[g] = h->h_ply[g];
    }
    return rc;
}

extern "C" int azg_states(azg_handle *h, float *out)
{
    if (!h || !out) return fail(h, AZG_EINVAL, "null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_games, S = h->st.S;
    void *d;
    int rc;
    if ((rc = scratch_get(h, 2, (size_t)B * 4 * S * sizeof(float), &d))) return rc;
    hipLaunchKernelGGL(states_kernel, dim3(B), dim3(256), 0, h->stream, h->st, (float *)d);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, d, (size_t)B * 4 * S * sizeof(float), hipMemcpyDefault, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return AZG_OK;
}

extern "C" int azg_drain_samples(azg_handle *h, int max_samples, float *states, double *pi, float *z,
                                 int64_t *game_id, int32_t *ply, int *n_out)
{
    if (!h || !n_out) return fail(h, AZG_EINVAL, "null argument");
    NOT_WHILE_PIPELINED(h);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_games, S = h->st.S;
    std::vector<int> map_g, map_t;
    std::vector<int> taken;
    for (int g = 0; g < B; g++) {
        if (h->h_status[g] != 1 || h->h_drained[g]) continue;
        const int cnt = h->h_scount[g];
        if ((int)map_g.size() + cnt > max_samples) break;
        for (int t = 0; t < cnt; t++) { map_g.push_back(g); map_t.push_back(t); }
        taken.push_back(g);
    }
    const int n = (int)map_g.size();
    *n_out = n;
    for (int g : taken) h->h_drained[g] = 1;
    if (n == 0) return AZG_OK;
    void *dg, *dt, *ds, *dp, *dz, *di, *dl;
    int rc;
    if ((rc = scratch_get(h, 0, (size_t)n * sizeof(int), &dg))) return rc;
    if ((rc = scratch_get(h, 1, (size_t)n * sizeof(int), &dt))) return rc;
    if ((rc = scratch_get(h, 2, (size_t)n * 4 * S * sizeof(float), &ds))) return rc;
    if ((rc = scratch_get(h, 3, (size_t)n * S * sizeof(double), &dp))) return rc;
    if ((rc = scratch_get(h, 4, (size_t)n * sizeof(float), &dz))) return rc;
    if ((rc = scratch_get(h, 5, (size_t)n * sizeof(long long), &di))) return rc;
    if ((rc = scratch_get(h, 6, (size_t)n * sizeof(int), &dl))) return rc;
    HIPCHK(h, hipMemcpyAsync(dg, map_g.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dt, map_t.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(drain_kernel, dim3(n), dim3(64), 0, h->stream, h->st, n, (const int *)dg, (const int *)dt,
                       (const int *)nullptr, (const long long *)nullptr, states ? (float *)ds : nullptr, pi ? (double *)dp : nullptr, (float *)dz, (long long *)di, (int *)dl,
                       (u64 *)nullptr, (int *)nullptr);
    HIPCHK(h, hipGetLastError());
    if (states) HIPCHK(h, hipMemcpyAsync(states, ds, (size_t)n * 4 * S * sizeof(float), hipMemcpyDefault, h->stream));
    if (pi) HIPCHK(h, hipMemcpyAsync(pi, dp, (size_t)n * S * sizeof(double), hipMemcpyDefault, h->stream));
    if (z) HIPCHK(h, hipMemcpyAsync(z, dz, (size_t)n * sizeof(float), hipMemcpyDefault, h->stream));
    if (game_id) HIPCHK(h, hipMemcpyAsync(game_id, di, (size_t)n * sizeof(long long), hipMemcpyDefault, h->stream));
    if (ply) HIPCHK(h, hipMemcpyAsync(ply, dl, (size_t)n * sizeof(int), hipMemcpyDefault, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return AZG_OK;
}

// ---------------------------------------------------------------- playouts evaluated by the caller
extern "C" int azg_playout_begin(azg_handle *h, int slot, uint64_t stones1[4], uint64_t stones2[4], int *ply,
                                 int *last_move, int *current_player, int *terminal)
{
    if (!h || !stones1 || !stones2 || !ply || !last_move || !current_player || !terminal) return fail(h, AZG_EINVAL, "null argument");
    NOT_WHILE_PIPELINED(h);
    if (slot < 0 || slot >= h->cfg.n_games) return fail(h, AZG_EINVAL, "slot %d out of range", slot);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->mirrors_fresh) { int rc = refresh_mirrors(h); if (rc) return rc; }
    if (h->h_status[slot] != 0) return fail(h, AZG_ESTATE, "slot %d holds no live game", slot);
    void *park, *info;
    int rc;
    if ((rc = scratch_get(h, 41, sizeof(GameLds), &park))) return rc;
    if ((rc = scratch_get(h, 42, sizeof(LeafInfo), &info))) return rc;
    hipLaunchKernelGGL(playout_select_kernel, dim3(1), dim3(64), 0, h->stream, h->st, slot, (GameLds *)park, (LeafInfo *)info);
    HIPCHK(h, hipGetLastError());
    LeafInfo li;
    HIPCHK(h, hipMemcpyAsync(&li, info, sizeof(li), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < MAXW; i++) { stones1[i] = li.stones[0][i]; stones2[i] = li.stones[1][i]; }
    *ply = li.ply; *last_move = li.last_move; *current_player = li.current_player;
    *terminal = li.term == 0 ? 0 : (li.term == 3 ? -1 : li.term);
    h->playout_parked = slot;
    return AZG_OK;
}

extern "C" int azg_playout_end(azg_handle *h, int slot, const float *priors, double value)
{
    if (!h) return AZG_EINVAL;
    NOT_WHILE_PIPELINED(h);
    if (h->playout_parked != slot || slot < 0) return fail(h, AZG_ESTATE, "azg_playout_begin has not been called for slot %d", slot);
    if (!std::isfinite(value)) return fail(h, AZG_EINVAL, "the leaf value must be finite");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int S = h->st.S;
    void *dp;
    int rc;
    if ((rc = scratch_get(h, 43, (size_t)MAXS * sizeof(float), &dp))) return rc;
    if (priors) HIPCHK(h, hipMemcpyAsync(dp, priors, (size_t)S * sizeof(float), hipMemcpyDefault, h->stream));
    else HIPCHK(h, hipMemsetAsync(dp, 0, (size_t)S * sizeof(float), h->stream));
    hipLaunchKernelGGL(playout_finish_kernel, dim3(1), dim3(64), 0, h->stream, h->st, slot, (const GameLds *)h->scratch[41],
                       (const float *)dp, value);
    HIPCHK(h, hipGetLastError());
    h->playout_parked = -1;
    h->mirrors_fresh = false;
    return AZG_OK;
}

// ---------------------------------------------------------------- rollout opponent
extern "C" int azg_pure_search(azg_handle *h, int n_playout, uint32_t *mt_state, int32_t *moves_out)
{
    if (!h || !mt_state || !moves_out) return fail(h, AZG_EINVAL, "null argument");
    NOT_WHILE_PIPELINED(h);
    if (n_playout < 1) return fail(h, AZG_EINVAL, "n_playout must be positive");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_games, S = h->st.S;
    // every playout expands at most one node of at most S children
    if ((long long)n_playout * S + 1 > (long long)h->st.cap)
        return fail(h, AZG_EARENA, "%d rollout playouts need arena_nodes >= %d (have %d)", n_playout, n_playout, h->cfg.arena_nodes);
    for (int g = 0; g < B; g++)
        if (mt_state[(size_t)g * 625 + 624] > 624u) return fail(h, AZG_EINVAL, "slot %d: generator position %u out of range", g, mt_state[(size_t)g * 625 + 624]);
    void *d_mt, *d_mv;
    int rc;
    if ((rc = scratch_get(h, 8, (size_t)B * 625 * sizeof(uint32_t), &d_mt))) return rc;
    if ((rc = scratch_get(h, 9, (size_t)B * sizeof(int), &d_mv))) return rc;
    HIPCHK(h, hipMemcpyAsync(d_mt, mt_state, (size_t)B * 625 * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    if (S <= 64)
        hipLaunchKernelGGL(pure_search_kernel<1>, dim3((B + 3) / 4), dim3(256), 0, h->stream, h->st, n_playout, (unsigned *)d_mt, (int *)d_mv);
    else
        hipLaunchKernelGGL(pure_search_kernel<4>, dim3((B + 3) / 4), dim3(256), 0, h->stream, h->st, n_playout, (unsigned *)d_mt, (int *)d_mv);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(mt_state, d_mt, (size_t)B * 625 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(moves_out, d_mv, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    rc = refresh_mirrors(h);                                  // synchronises the stream; picks up the arena-overflow flag
    if (rc != AZG_OK) return rc;
    for (int g = 0; g < B; g++)
        if (h->h_err[g] & 1) return fail(h, AZG_EARENA, "slot %d: the tree outgrew its arena", g);
    return AZG_OK;
}

extern "C" int azg_eval_batch(azg_handle *h, const float *states, int n, float *probs, float *values)
{
    if (!h || !states || n < 0) return fail(h, AZG_EINVAL, "bad argument");
    if (!h->have_weights) return fail(h, AZG_ENOWEIGHTS, "azg_set_weights must be called first");
    if (n == 0) return AZG_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    EvalLaunch el;
    if (!pick_eval(h->cfg.board_n, h->cfg.cnn_mode, &el)) return fail(h, AZG_EINVAL, "no CNN kernel is built for board_n=%d", h->cfg.board_n);
    const int S = h->st.S;
    void *ds = const_cast<float *>(states), *dp = probs, *dv = values;
    int rc;
    // device pointers are used in place, host pointers staged
    const bool st_dev = is_device_ptr(states), p_dev = probs && is_device_ptr(probs), v_dev = values && is_device_ptr(values);
    if (!st_dev) {
        if ((rc = scratch_get(h, 2, (size_t)n * 4 * S * sizeof(float), &ds))) return rc;
        HIPCHK(h, hipMemcpyAsync(ds, states, (size_t)n * 4 * S * sizeof(float), hipMemcpyDefault, h->stream));
    }
    if (!p_dev && (rc = scratch_get(h, 3, (size_t)n * S * sizeof(float), &dp))) return rc;
    if (!v_dev && (rc = scratch_get(h, 4, (size_t)n * sizeof(float), &dv))) return rc;
    void *dbad;
    if ((rc = scratch_get(h, 40, 16, &dbad))) return rc;
    HIPCHK(h, hipMemsetAsync(dbad, 0, 16, h->stream));
    hipLaunchKernelGGL(el.fn, dim3((n + el.T - 1) / el.T), dim3(256), el.lds, h->stream, h->st.net,
                       (const float *)ds, n, (float *)dp, (float *)dv, (int *)dbad);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->h_static_bad + 12, dbad, 4, hipMemcpyDeviceToHost, h->stream));
    if (probs && !p_dev) HIPCHK(h, hipMemcpyAsync(probs, dp, (size_t)n * S * sizeof(float), hipMemcpyDefault, h->stream));
    if (values && !v_dev) HIPCHK(h, hipMemcpyAsync(values, dv, (size_t)n * sizeof(float), hipMemcpyDefault, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->h_static_bad[12])
        return fail(h, AZG_ERANGE, "an activation of the net left the half-precision range (|x| > 65504) in the split-f16 "
                    "convolutions; create the engine with AZG_CNN_F32");
    return AZG_OK;
}

extern "C" int azg_board_step_batch(azg_handle *h, int n, uint64_t *stones, int32_t *ply, int32_t *last_move,
                                    const int32_t *moves, const int32_t *player, uint8_t *ended, int8_t *winner,
                                    float *planes)
{
    if (!h || n < 0 || !stones || !ply || !last_move || !moves || !player || !ended || !winner)
        return fail(h, AZG_EINVAL, "null argument");
    if (n == 0) return AZG_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int S = h->st.S;
    const size_t sb = (size_t)n * 2 * MAXW * 8, si = (size_t)n * 4;
    char *d = nullptr;
    const size_t total = sb + 4 * si + 2 * (size_t)n + 16 + (planes ? (size_t)n * 4 * S * 4 : 0);
    int rc;
    if ((rc = scratch_get(h, 7, total + 64, (void **)&d))) return rc;
    u64 *d_st = (u64 *)d;
    int *d_ply = (int *)(d + sb), *d_last = d_ply + n, *d_mv = d_last + n, *d_pl = d_mv + n, *d_bad = d_pl + n;
    float *d_planes = (float *)(d_bad + 4);
    unsigned char *d_end = (unsigned char *)(d_planes + (planes ? (size_t)n * 4 * S : 0));
    signed char *d_win = (signed char *)(d_end + n);
    HIPCHK(h, hipMemcpyAsync(d_st, stones, sb, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_ply, ply, si, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_last, last_move, si, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_mv, moves, si, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_pl, player, si, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(d_bad, 0, 16, h->stream));
    hipLaunchKernelGGL(board_step_kernel, dim3((n + 3) / 4), dim3(256), 0, h->stream, h->st.N, h->st.n_in_row, n,
                       d_st, d_ply, d_last, (const int *)d_mv, (const int *)d_pl, d_end, d_win,
                       planes ? d_planes : (float *)nullptr, d_bad);
    HIPCHK(h, hipGetLastError());
    int bad = 0;
    HIPCHK(h, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(stones, d_st, sb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ply, d_ply, si, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(last_move, d_last, si, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ended, d_end, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(winner, d_win, n, hipMemcpyDeviceToHost, h->stream));
    if (planes) HIPCHK(h, hipMemcpyAsync(planes, d_planes, (size_t)n * 4 * S * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (bad) return fail(h, AZG_EMOVE, "illegal move in batch (occupied cell, out of range, or bad player)");
    return AZG_OK;
}

// The same step with no allocation anywhere on the way: staging, the illegal-move flag and the
// outputs live in this GPU's copy of the file-scope `scratch` array (address looked up once per
// device in azg_create).  Calls on one GPU share that array, so calls that overlap on one GPU race.
extern "C" int azg_board_step_batch_static(azg_handle *h, int n, uint64_t *stones, int32_t *ply, int32_t *last_move,
                                           const int32_t *moves, const int32_t *player, uint8_t *ended, int8_t *winner,
                                           float *planes)
{
    if (!h || n < 0 || !stones || !ply || !last_move || !moves || !player || !ended || !winner)
        return fail(h, AZG_EINVAL, "null argument");
    if (n > kStaticMaxBoards)
        return fail(h, AZG_EINVAL, "azg_board_step_batch_static takes at most %d boards per call, got %d", kStaticMaxBoards, n);
    if (n == 0) return AZG_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int S = h->st.S;
    unsigned char *d = h->static_scratch;                     // device address, never the array's name
    const size_t sb = (size_t)n * 2 * MAXW * 8, si = (size_t)n * 4;
    u64 *d_st = (u64 *)(d + kStaticStonesOff);
    int *d_ply = (int *)(d + kStaticIntsOff), *d_last = d_ply + kStaticMaxBoards, *d_mv = d_last + kStaticMaxBoards,
        *d_pl = d_mv + kStaticMaxBoards, *d_bad = (int *)(d + kStaticBadOff);
    unsigned char *d_end = d + kStaticEndOff;
    signed char *d_win = (signed char *)(d + kStaticWinOff);
    float *d_planes = (float *)(d + kStaticPlanesOff);
    HIPCHK(h, hipMemcpyAsync(d_st, stones, sb, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_ply, ply, si, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_last, last_move, si, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_mv, moves, si, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_pl, player, si, hipMemcpyHostToDevice, h->stream));
    // the array keeps the previous call's contents: the flag must start at zero, and a board the
    // kernel rejects writes neither `ended` nor `winner`
    HIPCHK(h, hipMemsetAsync(d_bad, 0, 16, h->stream));
    HIPCHK(h, hipMemsetAsync(d_end, 0, (size_t)2 * kStaticMaxBoards, h->stream));
    hipLaunchKernelGGL(board_step_kernel, dim3((n + 3) / 4), dim3(256), 0, h->stream, h->st.N, h->st.n_in_row, n,
                       d_st, d_ply, d_last, (const int *)d_mv, (const int *)d_pl, d_end, d_win,
                       planes ? d_planes : (float *)nullptr, d_bad);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->h_static_bad, d_bad, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(stones, d_st, sb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ply, d_ply, si, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(last_move, d_last, si, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ended, d_end, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(winner, d_win, n, hipMemcpyDeviceToHost, h->stream));
    if (planes) HIPCHK(h, hipMemcpyAsync(planes, d_planes, (size_t)n * 4 * S * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->h_static_bad[0]) return fail(h, AZG_EMOVE, "illegal move in batch (occupied cell, out of range, or bad player)");
    return AZG_OK;
}


// ---------------------------------------------------------------- pipelined self-play
// The loop train.py:146 runs (collect_selfplay_data -> Game.start_self_play, game.py:360-416)
// with every slot kept busy and no host round trip between a move step and the next search:
//   search(k) | noise upload, advance(k) + on-device restart of finished slots, mirror copy |
//   search(k+1) | drain of the games that finished in step k | ...
// The host waits only for the mirror copy; it prepares the exploration noise of step k+1 and
// fetches the drained samples while search(k+1) runs.
static void sp_seed_slot(azg_handle *h, int g)
{
    h->rng[g].seed((uint32_t)((long long)h->sp.seed0 + h->h_game_id[g]));
}

extern "C" int azg_selfplay_begin(azg_handle *h, double temp, uint32_t seed0, int64_t first_id, int64_t stride,
                                  int64_t total)
{
    if (!h) return AZG_EINVAL;
    if (h->sp.active) return fail(h, AZG_ESTATE, "self-play pipeline already running");
    if (!(temp > 0.0)) return fail(h, AZG_EINVAL, "temp must be positive");
    if (stride < 1) return fail(h, AZG_EINVAL, "stride must be positive");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_games, S = h->st.S;
    auto &sp = h->sp;
    if (!sp.ev_mirror) HIPCHK(h, hipEventCreateWithFlags(&sp.ev_mirror, hipEventDisableTiming));
    if (!sp.ev_noise) HIPCHK(h, hipEventCreateWithFlags(&sp.ev_noise, hipEventDisableTiming));
    if (!sp.copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&sp.copy_stream, hipStreamNonBlocking));
    if (!sp.h_map) HIPCHK(h, hipHostMalloc((void **)&sp.h_map, (size_t)2 * 3 * B * S * sizeof(int)));
    if (!sp.h_ids) HIPCHK(h, hipHostMalloc((void **)&sp.h_ids, (size_t)2 * B * S * sizeof(long long)));
    sp.temp = temp; sp.seed0 = seed0; sp.first = first_id; sp.stride = stride; sp.total = total;
    sp.par = 0; sp.pend_n = sp.pend_games = sp.ready_n = sp.ready_games = 0;
    sp.fetched = true;
    // slots 0..B-1 start games k = 0..B-1 of this engine's sequence
    std::vector<uint8_t> mask(B, 0);
    std::vector<uint32_t> seeds(B, 0);
    std::vector<int64_t> ids(B, 0);
    int started = 0;
    for (int g = 0; g < B; g++) {
        if (total >= 0 && g >= total) continue;
        mask[g] = 1;
        ids[g] = first_id + (int64_t)g * stride;
        seeds[g] = (uint32_t)((long long)seed0 + ids[g]);
        started++;
    }
    HIPCHK(h, hipMemsetAsync(h->gate_table, 0, kGateWords * sizeof(int), h->stream));   // see create_impl
    int rc = azg_reset_games(h, mask.data(), seeds.data(), ids.data());
    if (rc) return rc;
    // slots that get no game must not keep an older one alive
    if (started < B) {
        HIPCHK(h, hipMemsetAsync(h->st.status + started, 2, B - started, h->stream));
        if ((rc = refresh_mirrors(h))) return rc;
    }
    const unsigned long long next = (unsigned long long)B;
    HIPCHK(h, hipMemcpyAsync(h->st.next_k, &next, sizeof(next), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = azg_search(h))) return rc;
    sp.active = true;
    return AZG_OK;
}

// The drain of the games that ended in the step just taken.  It goes to the copy stream: the advance
// that finished those games is complete (the host has its mirror), so the drain runs beside the
// search already in flight instead of between that search and the next advance.  The next step's
// noise upload follows it on the copy stream and the next advance waits for that upload, so by the
// time the host sees the next mirror the drained tuples are in place.
static int sp_enqueue_drain(azg_handle *h)
{
    auto &sp = h->sp;
    hipStream_t ds_ = sp.copy_stream;
    const int B = h->cfg.n_games, S = h->st.S;
    const size_t cap = (size_t)B * S;
    int *mg = sp.h_map + (size_t)sp.par * 3 * cap, *mt = mg + cap, *ma = mt + cap;
    long long *mi = sp.h_ids + (size_t)sp.par * cap;
    int n = 0, games = 0;
    for (int g = 0; g < B; g++) {
        if (!h->h_fin_flag[g]) continue;
        games++;
        const int aux = (h->h_fin_bank[g] & 1) | ((h->h_fin_winner[g] + 1) << 8);
        for (int t = 0; t < h->h_fin_count[g]; t++, n++) { mg[n] = g; mt[n] = t; ma[n] = aux; mi[n] = h->h_fin_id[g]; }
    }
    sp.pend_n = n; sp.pend_games = games;
    if (n == 0) return AZG_OK;
    const int base = 8 + sp.par * 9;
    void *dg, *dt, *da, *di, *ds, *dp, *dz, *dgid, *dply;
    int rc;
    if ((rc = scratch_get(h, base + 0, (size_t)n * sizeof(int), &dg))) return rc;
    if ((rc = scratch_get(h, base + 1, (size_t)n * sizeof(int), &dt))) return rc;
    if ((rc = scratch_get(h, base + 2, (size_t)n * sizeof(int), &da))) return rc;
    if ((rc = scratch_get(h, base + 3, (size_t)n * sizeof(long long), &di))) return rc;
    if ((rc = scratch_get(h, base + 4, (size_t)n * 4 * S * sizeof(float), &ds))) return rc;
    if ((rc = scratch_get(h, base + 5, (size_t)n * S * sizeof(double), &dp))) return rc;
    if ((rc = scratch_get(h, base + 6, (size_t)n * sizeof(float), &dz))) return rc;
    if ((rc = scratch_get(h, base + 7, (size_t)n * sizeof(long long), &dgid))) return rc;
    if ((rc = scratch_get(h, base + 8, (size_t)n * sizeof(int), &dply))) return rc;
    void *dst, *dla;
    if ((rc = scratch_get(h, 32 + sp.par * 2, (size_t)n * 2 * MAXW * sizeof(u64), &dst))) return rc;
    if ((rc = scratch_get(h, 33 + sp.par * 2, (size_t)n * sizeof(int), &dla))) return rc;
    HIPCHK(h, hipMemcpyAsync(dg, mg, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ds_));
    HIPCHK(h, hipMemcpyAsync(dt, mt, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ds_));
    HIPCHK(h, hipMemcpyAsync(da, ma, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ds_));
    HIPCHK(h, hipMemcpyAsync(di, mi, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, ds_));
    hipLaunchKernelGGL(drain_kernel, dim3(n), dim3(64), 0, ds_, h->st, n, (const int *)dg, (const int *)dt,
                       (const int *)da, (const long long *)di, (float *)ds, (double *)dp, (float *)dz,
                       (long long *)dgid, (int *)dply, (u64 *)dst, (int *)dla);
    HIPCHK(h, hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_selfplay_step(azg_handle *h, int *games_finished, int *samples_ready)
{
    if (!h) return AZG_EINVAL;
    if (!h->sp.active) return fail(h, AZG_ESTATE, "azg_selfplay_begin has not been called");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    auto &sp = h->sp;
    const int B = h->cfg.n_games, S = h->st.S;
    if (!sp.fetched)
        return fail(h, AZG_ESTATE, "%d samples made ready by the previous step were not fetched: call azg_selfplay_fetch "
                    "(null pointers discard them) before the next step", sp.ready_n);
    make_noise(h);                                  // while the search enqueued last time runs
    // the upload goes over the copy stream, i.e. also while that search runs (the previous
    // advance, the last reader of these buffers, completed before the last step returned)
    HIPCHK(h, hipMemcpyAsync(h->d_noise, h->h_noise, (size_t)B * S * sizeof(double), hipMemcpyHostToDevice, sp.copy_stream));
    HIPCHK(h, hipMemcpyAsync(h->d_u, h->h_u, (size_t)B * sizeof(double), hipMemcpyHostToDevice, sp.copy_stream));
    HIPCHK(h, hipEventRecord(sp.ev_noise, sp.copy_stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream, sp.ev_noise, 0));
    hipLaunchKernelGGL(advance_kernel, dim3((B + 3) / 4), dim3(256), 0, h->stream, h->st, sp.temp, h->d_noise, h->d_u,
                       (const int *)nullptr, ADV_SAMPLE | ADV_RECORD | ADV_BOARD | ADV_REROOT | ADV_RESTART,
                       (int *)nullptr, (double *)nullptr, (long long)sp.first, (long long)sp.stride, (long long)sp.total);
    HIPCHK(h, hipGetLastError());
    int rc = mirrors_enqueue(h);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(sp.ev_mirror, h->stream));
    if ((rc = azg_search(h))) return rc;            // the next search starts without waiting for the host
    HIPCHK(h, hipEventSynchronize(sp.ev_mirror));
    if ((rc = mirrors_unpack(h))) return rc;
    h->mirrors_fresh = false;                       // a search is in flight
    // what the previous step enqueued behind its search is complete now
    sp.ready_n = sp.pend_n; sp.ready_games = sp.pend_games; sp.ready_par = sp.par ^ 1;
    int fin = 0;
    for (int g = 0; g < B; g++) {
        if (!h->h_fin_flag[g]) continue;
        fin++;
        if (h->h_status[g] == 0) sp_seed_slot(h, g);   // restarted with a new game id
    }
    if ((rc = sp_enqueue_drain(h))) return rc;
    sp.par ^= 1;
    sp.fetched = sp.ready_n == 0;
    if (games_finished) *games_finished = fin;
    if (samples_ready) *samples_ready = sp.ready_n;
    return AZG_OK;
}

extern "C" int azg_selfplay_fetch(azg_handle *h, float *states, double *pi, float *z, int64_t *game_id, int32_t *ply)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    auto &sp = h->sp;
    const int n = sp.ready_n, S = h->st.S;
    sp.fetched = true;
    if (n == 0) return AZG_OK;
    const int base = 8 + sp.ready_par * 9;
    hipStream_t cs = sp.copy_stream;
    if (states) HIPCHK(h, hipMemcpyAsync(states, h->scratch[base + 4], (size_t)n * 4 * S * sizeof(float), hipMemcpyDefault, cs));
    if (pi) HIPCHK(h, hipMemcpyAsync(pi, h->scratch[base + 5], (size_t)n * S * sizeof(double), hipMemcpyDefault, cs));
    if (z) HIPCHK(h, hipMemcpyAsync(z, h->scratch[base + 6], (size_t)n * sizeof(float), hipMemcpyDefault, cs));
    if (game_id) HIPCHK(h, hipMemcpyAsync(game_id, h->scratch[base + 7], (size_t)n * sizeof(long long), hipMemcpyDefault, cs));
    if (ply) HIPCHK(h, hipMemcpyAsync(ply, h->scratch[base + 8], (size_t)n * sizeof(int), hipMemcpyDefault, cs));
    HIPCHK(h, hipStreamSynchronize(cs));
    return AZG_OK;
}

extern "C" int azg_selfplay_fetch_packed(azg_handle *h, uint64_t *stones, int32_t *last_move, double *pi, float *z,
                                         int64_t *game_id, int32_t *ply)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    auto &sp = h->sp;
    const int n = sp.ready_n, S = h->st.S;
    sp.fetched = true;
    if (n == 0) return AZG_OK;
    const int base = 8 + sp.ready_par * 9;
    hipStream_t cs = sp.copy_stream;
    if (stones) HIPCHK(h, hipMemcpyAsync(stones, h->scratch[32 + sp.ready_par * 2], (size_t)n * 2 * MAXW * sizeof(u64), hipMemcpyDefault, cs));
    if (last_move) HIPCHK(h, hipMemcpyAsync(last_move, h->scratch[33 + sp.ready_par * 2], (size_t)n * sizeof(int), hipMemcpyDefault, cs));
    if (pi) HIPCHK(h, hipMemcpyAsync(pi, h->scratch[base + 5], (size_t)n * S * sizeof(double), hipMemcpyDefault, cs));
    if (z) HIPCHK(h, hipMemcpyAsync(z, h->scratch[base + 6], (size_t)n * sizeof(float), hipMemcpyDefault, cs));
    if (game_id) HIPCHK(h, hipMemcpyAsync(game_id, h->scratch[base + 7], (size_t)n * sizeof(long long), hipMemcpyDefault, cs));
    if (ply) HIPCHK(h, hipMemcpyAsync(ply, h->scratch[base + 8], (size_t)n * sizeof(int), hipMemcpyDefault, cs));
    HIPCHK(h, hipStreamSynchronize(cs));
    return AZG_OK;
}

extern "C" int azg_unpack_states(azg_handle *h, int n, const uint64_t *stones, const int32_t *last_move, const int32_t *ply,
                                 float *states)
{
    if (!h || !stones || !last_move || !ply || !states || n < 0) return fail(h, AZG_EINVAL, "bad argument");
    if (n == 0) return AZG_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int S = h->st.S;
    hipStream_t cs = (h->sp.active && h->sp.copy_stream) ? h->sp.copy_stream : h->stream;   // never behind a search in flight
    void *ds, *dl, *dp, *os;
    int rc;
    if ((rc = scratch_get(h, 36, (size_t)n * 2 * MAXW * sizeof(u64), &ds))) return rc;
    if ((rc = scratch_get(h, 37, (size_t)n * sizeof(int), &dl))) return rc;
    if ((rc = scratch_get(h, 38, (size_t)n * sizeof(int), &dp))) return rc;
    if ((rc = scratch_get(h, 39, (size_t)n * 4 * S * sizeof(float), &os))) return rc;
    HIPCHK(h, hipMemcpyAsync(ds, stones, (size_t)n * 2 * MAXW * sizeof(u64), hipMemcpyDefault, cs));
    HIPCHK(h, hipMemcpyAsync(dl, last_move, (size_t)n * sizeof(int), hipMemcpyDefault, cs));
    HIPCHK(h, hipMemcpyAsync(dp, ply, (size_t)n * sizeof(int), hipMemcpyDefault, cs));
    hipLaunchKernelGGL(unpack_states_kernel, dim3(n), dim3(256), 0, cs, h->st.N, n, 1, (const u64 *)ds, (const int *)dl,
                       (const int *)dp, (float *)os);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(states, os, (size_t)n * 4 * S * sizeof(float), hipMemcpyDefault, cs));
    HIPCHK(h, hipStreamSynchronize(cs));
    return AZG_OK;
}

extern "C" int azg_selfplay_end(azg_handle *h, int *samples_ready)
{
    if (!h) return AZG_EINVAL;
    if (!h->sp.active) return fail(h, AZG_ESTATE, "azg_selfplay_begin has not been called");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    auto &sp = h->sp;
    sp.active = false;
    int rc = refresh_mirrors(h);                    // waits for the search in flight
    HIPCHK(h, hipStreamSynchronize(sp.copy_stream));   // and for the last drain
    sp.ready_n = sp.pend_n; sp.ready_games = sp.pend_games; sp.ready_par = sp.par ^ 1;
    sp.pend_n = sp.pend_games = 0;
    if (samples_ready) *samples_ready = sp.ready_n;
    return rc;
}

extern "C" int azg_equi_data(azg_handle *h, int n, const float *states, const double *pi, const float *z,
                             float *states_out, double *pi_out, float *z_out)
{
    if (!h || !states || !pi || !z || !states_out || !pi_out || !z_out || n < 0) return fail(h, AZG_EINVAL, "bad argument");
    if (n == 0) return AZG_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int S = h->st.S;
    // inputs and outputs may be host or device memory: device pointers are used in place, host
    // pointers staged.  While self-play is pipelined the work goes to the copy stream, so that it
    // never queues behind the search in flight.
    void *ds = const_cast<float *>(states), *dp = const_cast<double *>(pi), *dz = const_cast<float *>(z);
    void *os = states_out, *op = pi_out, *oz = z_out;
    int rc;
    hipStream_t cs = (h->sp.active && h->sp.copy_stream) ? h->sp.copy_stream : h->stream;
    const bool in_dev = is_device_ptr(states) && is_device_ptr(pi) && is_device_ptr(z);
    const bool out_dev = is_device_ptr(states_out) && is_device_ptr(pi_out) && is_device_ptr(z_out);
    if (!in_dev) {
        if ((rc = scratch_get(h, 26, (size_t)n * 4 * S * sizeof(float), &ds))) return rc;
        if ((rc = scratch_get(h, 27, (size_t)n * S * sizeof(double), &dp))) return rc;
        if ((rc = scratch_get(h, 28, (size_t)n * sizeof(float), &dz))) return rc;
        HIPCHK(h, hipMemcpyAsync(ds, states, (size_t)n * 4 * S * sizeof(float), hipMemcpyDefault, cs));
        HIPCHK(h, hipMemcpyAsync(dp, pi, (size_t)n * S * sizeof(double), hipMemcpyDefault, cs));
        HIPCHK(h, hipMemcpyAsync(dz, z, (size_t)n * sizeof(float), hipMemcpyDefault, cs));
    }
    if (!out_dev) {
        if ((rc = scratch_get(h, 29, (size_t)8 * n * 4 * S * sizeof(float), &os))) return rc;
        if ((rc = scratch_get(h, 30, (size_t)8 * n * S * sizeof(double), &op))) return rc;
        if ((rc = scratch_get(h, 31, (size_t)8 * n * sizeof(float), &oz))) return rc;
    }
    hipLaunchKernelGGL(equi_kernel, dim3(8 * n), dim3(256), 0, cs, h->st.N, n, (const float *)ds, (const double *)dp,
                       (const float *)dz, (float *)os, (double *)op, (float *)oz);
    HIPCHK(h, hipGetLastError());
    if (!out_dev) {
        HIPCHK(h, hipMemcpyAsync(states_out, os, (size_t)8 * n * 4 * S * sizeof(float), hipMemcpyDefault, cs));
        HIPCHK(h, hipMemcpyAsync(pi_out, op, (size_t)8 * n * S * sizeof(double), hipMemcpyDefault, cs));
        HIPCHK(h, hipMemcpyAsync(z_out, oz, (size_t)8 * n * sizeof(float), hipMemcpyDefault, cs));
    }
    HIPCHK(h, hipStreamSynchronize(cs));
    return AZG_OK;
}


// ---------------------------------------------------------------- training step
// PolicyValueNet.train_step (policy_value_net_pytorch.py:117-148) and the forward pass of
// PolicyValueNet.policy_value (:80-94) on a mini-batch, in float32 (kernels: azg_train.hpp).
static constexpr int kTrainCin[3] = {4, 32, 64}, kTrainCout[3] = {32, 64, 128};

// rows per split and number of splits of a reduction over M rows that should give about `want` workgroups
static void split_rows(int M, int want, int cap, int *rows, int *splits)
{
    want = std::max(1, std::min(want, cap));
    int r = (M + want - 1) / want;
    r = std::max(TKC, (r + TKC - 1) / TKC * TKC);
    *rows = r;
    *splits = (M + r - 1) / r;
}

static int train_prepare(azg_handle *h, int n)
{
    auto &w = h->tw;
    const int S = h->st.S;
    // (the GEMM kernels address their operands with 32-bit byte offsets: the widest activation array,
    //  batch * cells * 128 channels of float32, has to stay below 2 GiB)
    if ((long long)n * S * 128 * 4 >= (1LL << 31))
        return fail(h, AZG_EINVAL, "a training batch of %d positions is too large (at most %lld on this board)", n, (1LL << 31) / (512LL * S) - 1);
    if (!w.stream) HIPCHK(h, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    if (!w.h_loss) HIPCHK(h, hipHostMalloc((void **)&w.h_loss, 32));   // loss, entropy; a double at +8: KL
    const int M = n * S;
    for (int l = 0; l < 3; l++) {
        const int tiles = ((9 * kTrainCin[l] + TBM - 1) / TBM);
        split_rows(M, (AZG_TRAIN_WCAP * 4) / tiles, l == 0 ? 256 : AZG_TRAIN_WCAP, &w.wrows[l], &w.wsplits[l]);
    }
    split_rows(M, 256, 256, &w.hrows, &w.hsplits);
    split_rows(n, 8, 8, &w.frows, &w.fsplits);
    split_rows(n, 64, 64, &w.grows, &w.gsplits);
    if (n <= w.cap_n) return AZG_OK;
    if (w.block) HIPCHK(h, hipFree(w.block));
    w.block = nullptr;
    w.cap_n = 0;
    size_t off = 0;
    auto take = [&](size_t floats) { const size_t at = off; off += (floats * 4 + 255) / 256 * 256; return at; };
    const size_t Mz = (size_t)M;
    const int Sp = (S + 3) & ~3, ldhv = (2 * S + 3) & ~3;
    const size_t o_x4 = take(Mz * 4), o_a0 = take(Mz * 32), o_a1 = take(Mz * 64), o_a2 = take(Mz * 128);
    const size_t o_d0 = take(Mz * 32), o_d1 = take(Mz * 64), o_d2 = take(Mz * 128);
    const size_t o_hp = take((size_t)n * 4 * S), o_hv = take((size_t)n * ldhv), o_dh8 = take(Mz * 8);
    const size_t o_dl = take((size_t)n * Sp), o_dh1 = take((size_t)n * 64), o_g2 = take((size_t)n * 128);
    const size_t o_lp = take((size_t)n * 3), o_klp = take((size_t)n * 2), o_lo = take(8);   // lossout: loss, entropy, -, -, a double (KL)
    size_t o_wf[3], o_wd[3], o_wp[3], o_bp[3];
    for (int l = 0; l < 3; l++) {
        o_wf[l] = take((size_t)kTrainCout[l] * kTrainCin[l] * 9);
        o_wd[l] = take((size_t)kTrainCout[l] * kTrainCin[l] * 9);
        o_wp[l] = take((size_t)(l == 0 ? 256 : AZG_TRAIN_WCAP) * 9 * kTrainCin[l] * kTrainCout[l]);
        o_bp[l] = take((size_t)(l == 0 ? 256 : AZG_TRAIN_WCAP) * kTrainCout[l]);
    }
    const size_t o_wh = take(32 * 128), o_bh = take(8), o_hpt = take((size_t)256 * 128 * 8), o_hbp = take(256 * 8);
    const size_t o_pp = take((size_t)8 * 4 * S * S), o_pbp = take((size_t)8 * S), o_vp = take((size_t)8 * 2 * S * 64), o_vbp = take(8 * 64);
    const size_t o_g2p = take(64 * 128);
    HIPCHK(h, hipMalloc((void **)&w.block, off));
    HIPCHK(h, hipMemsetAsync(w.block, 0, off, w.stream));      // the padding columns of hv stay zero for good
    char *b = w.block;
    auto at = [&](size_t o) { return reinterpret_cast<float *>(b + o); };
    w.x4 = at(o_x4); w.a[0] = at(o_a0); w.a[1] = at(o_a1); w.a[2] = at(o_a2);
    w.dz[0] = at(o_d0); w.dz[1] = at(o_d1); w.dz[2] = at(o_d2);
    w.hp = at(o_hp); w.hv = at(o_hv); w.dh8 = at(o_dh8); w.dlogit = at(o_dl); w.dh1 = at(o_dh1); w.g2 = at(o_g2);
    w.lossparts = at(o_lp); w.lossout = at(o_lo); w.klparts = reinterpret_cast<double *>(at(o_klp));
    for (int l = 0; l < 3; l++) { w.im.wf[l] = at(o_wf[l]); w.im.wd[l] = at(o_wd[l]); w.wpart[l] = at(o_wp[l]); w.bpart[l] = at(o_bp[l]); }
    w.im.wh = at(o_wh); w.im.bh = at(o_bh); w.hpart = at(o_hpt); w.hbpart = at(o_hbp);
    w.ppart = at(o_pp); w.pbpart = at(o_pbp); w.vpart = at(o_vp); w.vbpart = at(o_vbp); w.g2part = at(o_g2p);
    w.cap_n = n;
    return AZG_OK;
}

// The 16 parameter pointers of a net do not change between calls: they are looked at once.
static int train_check_ptrs(azg_handle *h, const float *const params[16])
{
    if (!params) return fail(h, AZG_EINVAL, "null argument");
    bool same = true;
    for (int i = 0; i < 16; i++) same = same && params[i] == h->tw.checked[i];
    if (same && params[0]) return AZG_OK;
    for (int i = 0; i < 16; i++)
        if (!params[i] || !is_device_ptr(params[i])) return fail(h, AZG_EINVAL, "parameter tensor %d must be a device pointer", i);
    for (int i = 0; i < 16; i++) h->tw.checked[i] = params[i];
    return AZG_OK;
}

// (conv_gemm_kernel's stage arithmetic wants 4 source channels -- CS4 -- or a multiple of the 16 k of a
//  stage: the channel counts of this net, fixed at compile time)
static_assert(kTrainCin[0] == 4 && kTrainCin[1] % TKC == 0 && kTrainCin[2] % TKC == 0 && kTrainCout[2] % TKC == 0,
              "conv_gemm_kernel cannot take these channel counts");
template <int BN, int MODE, bool CS4 = false>
static void launch_conv_gemm(hipStream_t s, const ConvGemmArgs &a)
{
    hipLaunchKernelGGL((conv_gemm_kernel<BN, MODE, CS4>), dim3((a.M + TBM - 1) / TBM, (a.N + BN - 1) / BN), dim3(256), 0, s, a);
}

// forward pass of the batch already at w.x4 (channels-last) up to the head outputs hp / hv
static void train_forward_convs(azg_handle *h, int n, const TrainParams &P)
{
    auto &w = h->tw;
    const int S = h->st.S, N = h->st.N, M = n * S;
    hipStream_t s = w.stream;
    hipLaunchKernelGGL(train_pack_kernel, dim3((kTrainPackThreads + 255) / 256), dim3(256), 0, s, P, w.im);
    ConvGemmArgs a{};
    a.M = M; a.S = S; a.Nb = N; a.taps = 9;
    a.src = w.x4; a.Cs = 4; a.K = 36; a.w = w.im.wf[0]; a.bias = P.p[1]; a.out = w.a[0]; a.N = 32; a.ldo = 32;
    launch_conv_gemm<32, 0, true>(s, a);
    a.src = w.a[0]; a.Cs = 32; a.K = 288; a.w = w.im.wf[1]; a.bias = P.p[3]; a.out = w.a[1]; a.N = 64; a.ldo = 64;
    launch_conv_gemm<64, 0>(s, a);
    a.src = w.a[1]; a.Cs = 64; a.K = 576; a.w = w.im.wf[2]; a.bias = P.p[5]; a.out = w.a[2]; a.N = 128; a.ldo = 128;
    launch_conv_gemm<AZG_TRAIN_BN3, 0>(s, a);
    a.src = w.a[2]; a.Cs = 128; a.K = 128; a.taps = 1; a.w = w.im.wh; a.bias = w.im.bh; a.out = w.hp; a.out2 = w.hv; a.N = 32;
    a.ldo = (2 * S + 3) & ~3;
    launch_conv_gemm<32, 2>(s, a);
}

static FcArgs train_fc_args(azg_handle *h, int n, const TrainParams &P)
{
    auto &w = h->tw;
    FcArgs f{};
    f.hp = w.hp; f.hv = w.hv; f.wp = P.p[8]; f.bp = P.p[9]; f.wv1 = P.p[12]; f.bv1 = P.p[13]; f.wv2 = P.p[14]; f.bv2 = P.p[15];
    f.dlogit = w.dlogit; f.dh1 = w.dh1; f.g2 = w.g2; f.dh8 = w.dh8; f.lossparts = w.lossparts;
    f.n = n; f.S = h->st.S; f.ldhv = (2 * h->st.S + 3) & ~3;
    return f;
}

// the forward pass of n positions (and the KL test's number against old_probs, left in h_loss) on the
// training stream, no synchronisation
static int train_forward_enqueue(azg_handle *h, int n, const float *states, const TrainParams &P, float *probs, float *values,
                                 const float *old_probs)
{
    auto &w = h->tw;
    const int M = n * h->st.S;
    hipLaunchKernelGGL(planes_to_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, w.stream, states, n, h->st.S, w.x4);
    train_forward_convs(h, n, P);
    FcArgs f = train_fc_args(h, n, P);
    f.probs = probs; f.values = values; f.oldp = old_probs; f.klparts = w.klparts;
    hipLaunchKernelGGL(fc_sample_kernel, dim3(n), dim3(256), 0, w.stream, f);
    if (old_probs) {
        double *kl_dev = reinterpret_cast<double *>(w.lossout + 4), *kl_host = reinterpret_cast<double *>(w.h_loss + 2);
        hipLaunchKernelGGL(kl_kernel, dim3(1), dim3(256), 0, w.stream, (const double *)w.klparts, n, kl_dev);
        HIPCHK(h, hipMemcpyAsync(kl_host, kl_dev, 8, hipMemcpyDeviceToHost, w.stream));
    }
    HIPCHK(h, hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_train_forward(azg_handle *h, int n, const float *states, const float *const params[16], float *probs,
                                 float *values, const float *old_probs, double *kl)
{
    if (!h || n < 0 || !states || !probs || !values || (old_probs && !kl)) return fail(h, AZG_EINVAL, "null argument");
    if (old_probs && !is_device_ptr(old_probs)) return fail(h, AZG_EINVAL, "azg_train_forward takes device pointers");
    if (kl) *kl = 0.0;
    if (n == 0) return AZG_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = train_check_ptrs(h, params);
    if (rc) return rc;
    if (!is_device_ptr(states) || !is_device_ptr(probs)) return fail(h, AZG_EINVAL, "azg_train_forward takes device pointers");
    if ((rc = train_prepare(h, n))) return rc;
    TrainParams P;
    for (int i = 0; i < 16; i++) P.p[i] = const_cast<float *>(params[i]);
    if ((rc = train_forward_enqueue(h, n, states, P, probs, values, old_probs))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->tw.stream));
    if (old_probs) *kl = *reinterpret_cast<double *>(h->tw.h_loss + 2);
    return AZG_OK;
}

// forward + backward of one mini-batch, enqueued on the training stream (no synchronisation)
static int train_backward_enqueue(azg_handle *h, int n, const float *states, const float *mcts_probs, const float *winners,
                                  const TrainParams &P, float *grads)
{
    auto &w = h->tw;
    hipStream_t s = w.stream;
    const int S = h->st.S, N = h->st.N, M = n * S;
    hipLaunchKernelGGL(planes_to_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, s, states, n, S, w.x4);
    train_forward_convs(h, n, P);
    FcArgs f = train_fc_args(h, n, P);
    f.pi = mcts_probs; f.z = winners;
    hipLaunchKernelGGL(fc_sample_kernel, dim3(n), dim3(256), 0, s, f);
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, s, (const float *)w.lossparts, n, w.lossout);
    // fully connected layers: weight gradients as [input][output] partial sums, bias gradients with them
    {
        WgradArgs g{};
        g.taps = 1; g.M = n; g.S = 1; g.Nb = 1; g.rows_per_split = w.frows;
        g.src = w.hp; g.Cs = 4 * S; g.I = 4 * S; g.dz = w.dlogit; g.ldz = (S + 3) & ~3; g.J = S; g.part = w.ppart; g.bpart = w.pbpart;
        hipLaunchKernelGGL((wgrad_gemm_kernel<64>), dim3((g.I + TBM - 1) / TBM, (g.J + 63) / 64, w.fsplits), dim3(256), 0, s, g);
        g.src = w.hv; g.Cs = (2 * S + 3) & ~3; g.I = 2 * S; g.dz = w.dh1; g.ldz = 64; g.J = 64; g.part = w.vpart; g.bpart = w.vbpart;
        hipLaunchKernelGGL((wgrad_gemm_kernel<64>), dim3((g.I + TBM - 1) / TBM, 1, w.fsplits), dim3(256), 0, s, g);
        hipLaunchKernelGGL(colsum_kernel, dim3(w.gsplits), dim3(256), 0, s, (const float *)w.g2, n, 128, 128, w.grows, w.g2part);
    }
    // heads: weight and bias gradients (partials), gradient at conv3's pre-activation
    {
        WgradArgs g{};
        g.src = w.a[2]; g.Cs = 128; g.taps = 1; g.I = 128; g.dz = w.dh8; g.ldz = 8; g.J = 8; g.part = w.hpart; g.bpart = w.hbpart;
        g.M = M; g.S = S; g.Nb = N; g.rows_per_split = w.hrows;
        hipLaunchKernelGGL((wgrad_gemm_kernel<32>), dim3(1, 1, w.hsplits), dim3(256), 0, s, g);
        hipLaunchKernelGGL(head_backward_kernel, dim3((M * 32 + 255) / 256), dim3(256), 0, s, (const float *)w.dh8,
                           (const float *)w.im.wh, (const float *)w.a[2], M, w.dz[2]);
    }
    // the three 3x3 layers, last to first: weight / bias gradients of layer l, then the gradient at layer l-1
    for (int l = 2; l >= 0; l--) {
        WgradArgs g{};
        g.src = l == 0 ? w.x4 : w.a[l - 1]; g.Cs = kTrainCin[l]; g.taps = 9; g.I = 9 * kTrainCin[l];
        g.dz = w.dz[l]; g.ldz = kTrainCout[l]; g.J = kTrainCout[l]; g.part = w.wpart[l]; g.bpart = w.bpart[l];
        g.M = M; g.S = S; g.Nb = N; g.rows_per_split = w.wrows[l];
        const dim3 grid((g.I + TBM - 1) / TBM, 1, w.wsplits[l]);
        if (l == 2) hipLaunchKernelGGL((wgrad_gemm_kernel<128>), grid, dim3(256), 0, s, g);
        else if (l == 1) hipLaunchKernelGGL((wgrad_gemm_kernel<64>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((wgrad_gemm_kernel<32>), grid, dim3(256), 0, s, g);
        if (l > 0) {
            ConvGemmArgs a{};
            a.M = M; a.S = S; a.Nb = N; a.taps = 9;
            a.src = w.dz[l]; a.Cs = kTrainCout[l]; a.K = 9 * kTrainCout[l]; a.w = w.im.wd[l]; a.act = w.a[l - 1];
            a.out = w.dz[l - 1]; a.N = kTrainCin[l]; a.ldo = kTrainCin[l];
            if (l == 2) launch_conv_gemm<AZG_TRAIN_BD3, 1>(s, a);
            else launch_conv_gemm<32, 1>(s, a);
        }
    }
    GradSegs gs{};
    auto seg = [&](const float *part, int size, int splits, int kind, int a, int b) {
        GradSeg &q = gs.s[gs.n++];
        q.part = part; q.start = gs.total; q.size = size; q.splits = splits; q.kind = kind; q.a = a; q.b = b;
        gs.total += size;
    };
    for (int l = 0; l < 3; l++) {
        seg(w.wpart[l], 9 * kTrainCin[l] * kTrainCout[l], w.wsplits[l], GS_CONV_W, 2 * l, kTrainCin[l]);
        seg(w.bpart[l], kTrainCout[l], w.wsplits[l], GS_PLAIN, 2 * l + 1, 0);
    }
    seg(w.hpart, 128 * 8, w.hsplits, GS_HEAD_W, 0, 0);
    seg(w.hbpart, 8, w.hsplits, GS_HEAD_B, 0, 0);
    seg(w.ppart, 4 * S * S, w.fsplits, GS_FC_W, 8, S);
    seg(w.pbpart, S, w.fsplits, GS_PLAIN, 9, 0);
    seg(w.vpart, 2 * S * 64, w.fsplits, GS_FC_W, 12, 64);
    seg(w.vbpart, 64, w.fsplits, GS_PLAIN, 13, 0);
    seg(w.g2part, 128, w.gsplits, GS_G2, 0, 0);
    hipLaunchKernelGGL(grad_assemble_kernel, dim3((gs.total + kAssembleEntries - 1) / kAssembleEntries), dim3(256), 0, s, gs, S, grads);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(w.h_loss, w.lossout, 8, hipMemcpyDeviceToHost, s));
    return AZG_OK;
}

static int train_adam_enqueue(azg_handle *h, const TrainParams &P, const float *grads, float *adam_m, float *adam_v, int64_t step,
                              double lr, double beta1, double beta2, double eps, double weight_decay)
{
    const ParamLayout L = param_layout(h->st.S);
    AdamArgs a{};
    a.grad = grads; a.m = adam_m; a.v = adam_v; a.total = L.off[16];
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    a.one_minus_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_minus_b2 = (float)(1.0 - beta2);
    a.neg_step_size = (float)(-(lr / bc1)); a.bc2_sqrt = (float)std::sqrt(bc2); a.eps = (float)eps; a.wd = (float)weight_decay;
    hipLaunchKernelGGL(adam_kernel, dim3((a.total + 255) / 256), dim3(256), 0, h->tw.stream, P, L, a);
    HIPCHK(h, hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_train_backward(azg_handle *h, int n, const float *states, const float *mcts_probs, const float *winners,
                                  const float *const params[16], float *grads, float *loss, float *entropy)
{
    if (!h || n < 1 || !states || !mcts_probs || !winners || !grads) return fail(h, AZG_EINVAL, "bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = train_check_ptrs(h, params);
    if (rc) return rc;
    if (!is_device_ptr(states) || !is_device_ptr(grads))
        return fail(h, AZG_EINVAL, "azg_train_backward takes device pointers (loss and entropy are host pointers)");
    if ((rc = train_prepare(h, n))) return rc;
    TrainParams P;
    for (int i = 0; i < 16; i++) P.p[i] = const_cast<float *>(params[i]);
    if ((rc = train_backward_enqueue(h, n, states, mcts_probs, winners, P, grads))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->tw.stream));
    if (loss) *loss = h->tw.h_loss[0];
    if (entropy) *entropy = h->tw.h_loss[1];
    return AZG_OK;
}

extern "C" int azg_train_adam(azg_handle *h, float *const params[16], const float *grads, float *adam_m, float *adam_v,
                              int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay)
{
    if (!h || !grads || !adam_m || !adam_v || step < 1) return fail(h, AZG_EINVAL, "bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = train_check_ptrs(h, params);
    if (rc) return rc;
    if (!is_device_ptr(grads) || !is_device_ptr(adam_m) || !is_device_ptr(adam_v)) return fail(h, AZG_EINVAL, "azg_train_adam takes device pointers");
    if ((rc = train_prepare(h, std::max(1, h->tw.cap_n)))) return rc;
    TrainParams P;
    for (int i = 0; i < 16; i++) P.p[i] = params[i];
    if ((rc = train_adam_enqueue(h, P, grads, adam_m, adam_v, step, lr, beta1, beta2, eps, weight_decay))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->tw.stream));
    return AZG_OK;
}

extern "C" int azg_train_step(azg_handle *h, int n, const float *states, const float *mcts_probs, const float *winners,
                              float *const params[16], float *grads, float *adam_m, float *adam_v, int64_t step, double lr,
                              double beta1, double beta2, double eps, double weight_decay, float *loss, float *entropy)
{
    if (!h || n < 1 || !states || !mcts_probs || !winners || !grads || !adam_m || !adam_v || step < 1) return fail(h, AZG_EINVAL, "bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = train_check_ptrs(h, params);
    if (rc) return rc;
    if (!is_device_ptr(states) || !is_device_ptr(grads) || !is_device_ptr(adam_m))
        return fail(h, AZG_EINVAL, "azg_train_step takes device pointers (loss and entropy are host pointers)");
    if ((rc = train_prepare(h, n))) return rc;
    TrainParams P;
    for (int i = 0; i < 16; i++) P.p[i] = params[i];
    if ((rc = train_backward_enqueue(h, n, states, mcts_probs, winners, P, grads))) return rc;
    if ((rc = train_adam_enqueue(h, P, grads, adam_m, adam_v, step, lr, beta1, beta2, eps, weight_decay))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->tw.stream));
    if (loss) *loss = h->tw.h_loss[0];
    if (entropy) *entropy = h->tw.h_loss[1];
    return AZG_OK;
}

extern "C" int azg_train_epoch(azg_handle *h, int n, const float *states, const float *mcts_probs, const float *winners,
                               float *const params[16], float *grads, float *adam_m, float *adam_v, int64_t step, double lr,
                               double beta1, double beta2, double eps, double weight_decay, float *loss, float *entropy,
                               int n_all, const float *states_all, float *probs, float *values, const float *old_probs, double *kl)
{
    if (!h || n < 1 || n_all < 1 || !states || !mcts_probs || !winners || !grads || !adam_m || !adam_v || step < 1 || !states_all || !probs ||
        !values || !old_probs || !kl)
        return fail(h, AZG_EINVAL, "bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = train_check_ptrs(h, params);
    if (rc) return rc;
    if (!is_device_ptr(states) || !is_device_ptr(grads) || !is_device_ptr(adam_m) || !is_device_ptr(states_all) || !is_device_ptr(probs) ||
        !is_device_ptr(old_probs))
        return fail(h, AZG_EINVAL, "azg_train_epoch takes device pointers (loss, entropy and kl are host pointers)");
    // (room for the larger of the two batches; the partial-sum splits of the backward pass are those of ITS
    //  batch size, so the step is the same, bit for bit, as azg_train_step's)
    if ((rc = train_prepare(h, std::max(n, n_all)))) return rc;
    if ((rc = train_prepare(h, n))) return rc;
    TrainParams P;
    for (int i = 0; i < 16; i++) P.p[i] = params[i];
    if ((rc = train_backward_enqueue(h, n, states, mcts_probs, winners, P, grads))) return rc;
    if ((rc = train_adam_enqueue(h, P, grads, adam_m, adam_v, step, lr, beta1, beta2, eps, weight_decay))) return rc;
    if ((rc = train_forward_enqueue(h, n_all, states_all, P, probs, values, old_probs))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->tw.stream));
    if (loss) *loss = h->tw.h_loss[0];
    if (entropy) *entropy = h->tw.h_loss[1];
    *kl = *reinterpret_cast<double *>(h->tw.h_loss + 2);
    return AZG_OK;
}

extern "C" int azg_train_param_count(azg_handle *h)
{
    return h ? param_layout(h->st.S).off[16] : AZG_EINVAL;
}

// Diagnostic builds (-DAZG_STAMPS) only: 32 summed phase counters, then per workgroup of the
// last search launch (loop time in 100 MHz ticks, hardware id) for the first 2048 workgroups.
extern "C" int azg_debug_stamps(azg_handle *h, long long out[kStampWords])
{
    if (!h || !out) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->st.stamps, kStampWords * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemset(h->st.stamps, 0, kStampWords * sizeof(long long)));
    return AZG_OK;
}

extern "C" int azg_arena_stats(azg_handle *h, int64_t *moved_to_large, int64_t *trees_dropped)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    unsigned long long v[2] = {0, 0};
    // on the copy stream when self-play is pipelined: a search in flight must not be waited for
    hipStream_t s = (h->sp.active && h->sp.copy_stream) ? h->sp.copy_stream : h->stream;
    HIPCHK(h, hipMemcpyAsync(h->h_static_bad + 4, h->st.arena_stats, sizeof(v), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    memcpy(v, h->h_static_bad + 4, sizeof(v));
    if (moved_to_large) *moved_to_large = (int64_t)v[0];
    if (trees_dropped) *trees_dropped = (int64_t)v[1];
    return AZG_OK;
}

extern "C" int azg_reroot_stats(azg_handle *h, int64_t *in_place, int64_t *compacted)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    unsigned long long v[2] = {0, 0};
    hipStream_t s = (h->sp.active && h->sp.copy_stream) ? h->sp.copy_stream : h->stream;
    HIPCHK(h, hipMemcpyAsync(h->h_static_bad + 8, h->st.arena_stats + 2, sizeof(v), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    memcpy(v, h->h_static_bad + 8, sizeof(v));
    if (in_place) *in_place = (int64_t)v[0];
    if (compacted) *compacted = (int64_t)v[1];
    return AZG_OK;
}

extern "C" int azg_gate_stats(azg_handle *h, int64_t *giveups, int64_t *words_set)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (h->sp.active && h->sp.copy_stream) ? h->sp.copy_stream : h->stream;
    std::vector<int> words(kGateWords);
    unsigned n = 0;
    HIPCHK(h, hipMemcpyAsync(words.data(), h->gate_table, kGateWords * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(&n, h->gate_giveups, sizeof(n), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (giveups) *giveups = (int64_t)n;
    if (words_set) *words_set = (int64_t)std::count_if(words.begin(), words.end(), [](int w) { return w != 0; });
    return AZG_OK;
}

extern "C" int azg_gate_poke(azg_handle *h, int value)
{
    if (!h) return AZG_EINVAL;
    NOT_WHILE_PIPELINED(h);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    std::vector<int> words(kGateWords, value);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->gate_table, words.data(), kGateWords * sizeof(int), hipMemcpyHostToDevice));
    return AZG_OK;
}

#ifdef AZG_DUAL_DUMP
extern "C" int azg_debug_dual_dump(azg_handle *h, float *out)
{
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dual_dump), sizeof(float) * 256 * 2048));
    return AZG_OK;
}
#endif

extern "C" int azg_host_timing(azg_handle *h, double *noise_ms, int64_t *noise_steps)
{
    if (!h) return AZG_EINVAL;
    if (noise_ms) *noise_ms = h->noise_ms;
    if (noise_steps) *noise_steps = h->noise_steps;
    h->noise_ms = 0.0;
    h->noise_steps = 0;
    return AZG_OK;
}

extern "C" int azg_profile(azg_handle *h, int64_t *launches, double *ms, int64_t *playouts)
{
    if (!h) return AZG_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int rc = harvest_events(h, h->ev_pending);
    if (rc) return rc;
    const int B = h->cfg.n_games;
    std::vector<long long> ev(B);
    HIPCHK(h, hipMemcpy(ev.data(), h->st.evals, (size_t)B * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemset(h->st.evals, 0, (size_t)B * sizeof(long long)));
    long long tot = 0;
    for (long long v : ev) tot += v;
    if (launches) *launches = h->prof_launches;
    if (ms) *ms = h->prof_ms;
    if (playouts) *playouts = tot;
    h->prof_launches = 0;
    h->prof_ms = 0.0;
    return AZG_OK;
}
