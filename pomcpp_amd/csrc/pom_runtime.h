/*
 * pom_runtime.h — host runtime of the batched stepper: the batch handle, its streams, and how a step becomes launches of
 * pom_step_kernel (pom_kernels.h).  The C-ABI functions of include/pom_batch.h (pom_batch.hip) are thin wrappers over this.
 */
#ifndef POM_RUNTIME_H_
#define POM_RUNTIME_H_

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

#include "pom_kernels.h"
#include "pom_chain.h"

static thread_local char g_err[256] = "";
static void set_err(const char* what, hipError_t e)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
}
#define HIPCHK(call)                      \
    do {                                  \
        hipError_t e_ = (call);           \
        if (e_ != hipSuccess) {           \
            set_err(#call, e_);           \
            return POM_E_HIP;             \
        }                                 \
    } while (0)

struct PomBatch {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0, n_pad = 0, n_waves = 0, env_offset = 0; /* n_waves: counter slots, sized for the smallest EPW */
    int epw = 64;
    bool quad = false; /* EPW 16 with four lanes per env (pom_step_kernel<16, 4>) */
    int mode = POM_MODE_ENV, auto_reset = 0, max_steps = 0;
    uint32_t* state = nullptr;
    uint32_t* snap = nullptr;       /* restart snapshot, array of structs */
    uint32_t* terminal = nullptr;   /* POM_RESET_AT_END: final record of each env's last finished episode, array of structs */
    int32_t* moves_dev = nullptr;   /* n_pad x 4 */
    uint32_t* agent_mem = nullptr;  /* SimpleAgent memory, [2][4 * n_pad], allocated on first use */
    uint32_t* episode = nullptr;    /* games started per env (fresh boards) */
    uint64_t board_seed = 0;
    int fresh = 0;
    int32_t* staging = nullptr;     /* staging_envs x 251 dwords (AoS), also status scratch */
    int64_t staging_envs = 0;
    int64_t* wave_counters = nullptr;
    int64_t* totals_dev = nullptr;
    int* first_bad = nullptr;
    uint64_t tick = 0;
    /* A step is issued as `parts` kernels over contiguous tile ranges on internal streams: the launches are
     * independent (envs never interact), so one part's HBM load / store phases overlap the others' compute
     * instead of all wavefronts of the chip loading and storing in lock-step.  The caller's stream is forked
     * into the sub-streams lazily and joined again before anything else touches the batch. */
    enum { MAX_PARTS = 8, PROF_RING = 256 };
    int parts = 1;
    hipStream_t sub[MAX_PARTS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_PARTS] = {};
    bool forked = false;
    int forked_n = 0; /* how many streams (indices below it) the fork covers */
    int main_part = 1; /* part 0 of a split step runs on the caller's stream itself, parts 1.. on sub-streams: one stream
                          fewer for the same overlap (3 parts: 21.2 -> 20.6 us per step at 65,536 envs); POM_MAIN_PART=0: all
                          parts on sub-streams */
    /* how the launches of a several-tick call are issued (launch_many; PomBatchOptions.issue_mode) */
    int issue_mode = POM_ISSUE_THREADS;
    struct PomIssuer* issuers[MAX_PARTS] = {}; /* POM_ISSUE_THREADS: one helper thread per sub-stream part, created on first use */
    bool issuers_failed = false;
    int last_kind = 0;                         /* what has been launched on the streams since they were forked: POM_KIND_SPLIT / _CHAIN */               /* a helper thread could not be started: the calling thread issues everything */
    /* POM_ISSUE_GRAPH: multi-tick calls replay a captured chunk of launches (launch_many): POM_GRAPH_TICKS launches of every part as one HIP
     * graph per part.  tick_words[k] is the tick part k's replay starts at, read by the graph's kernels (StepParams.tick_base;
     * set by a one-lane kernel on the part's stream in front of each replay); tick_words[MAX_PARTS] stays 0 and is what every
     * launch outside a graph points at. */
    enum { MAX_GRAPHS = 4 };
    struct PomStepGraph* graphs[MAX_GRAPHS] = {};
    uint64_t graph_clock = 0;
    uint32_t* tick_words = nullptr;
    PomChain chain;          /* POM_ISSUE_CHAIN: the tiles' ticket words, set up on first use (pom_chain.h) */
    int chain_parts = 3;     /* ... and how many streams the chained launches may rotate over */
    bool chain_auto = true;  /* ... of which a call uses two if it is short and three if it is long (launch_many_chain) — unless the caller
                                asked for a number of streams */
    int chain_last_use = 2;  /* what the last chained call used (pom_batch_issue_info) */
    bool fuse_policy = true; /* pom_batch_step_simple: policy and tick in one kernel (quad shape); POM_FUSE=0 keeps them apart */
    /* optional per-launch timing (pom_batch_profile) */
    bool profiling = false;
    hipEvent_t prof_ev[2 * PROF_RING] = {};
    int prof_n = 0;
#if defined(POM_DIAG)
    long long* diag = nullptr;
    long long* diag_pol = nullptr;
#endif
};

static void drop_graphs(PomBatch* h);
static void stop_issuers(PomBatch* h);
enum { POM_KIND_SPLIT = 1, POM_KIND_CHAIN = 2 };
static int fork_parts(PomBatch* h, int kind = POM_KIND_SPLIT);
static int join_parts(PomBatch* h);
static int ensure_sub_streams(PomBatch* h, int parts);
static int chain_settle(PomBatch* h);
/* before anything but another chained call reads or changes the batch: the chained launches are joined and checked, tiles that
 * could not be played in the chain are caught up (chain_settle), then the sub-streams are joined into the caller's stream */
static int quiesce(PomBatch* h)
{
    if (int rc = chain_settle(h)) return rc;
    return join_parts(h);
}

static int check_range(const PomBatch* h, int64_t first, int64_t count)
{
    if (!h || first < 0 || count < 0 || first + count > h->n) {
        snprintf(g_err, sizeof g_err, "range [%lld, %lld) outside batch", (long long)first, (long long)(first + count));
        return POM_E_ARG;
    }
    return POM_OK;
}

static int ensure_sub_streams(PomBatch* h, int parts)
{
    if (parts <= 1) return POM_OK;
    if (!h->ev_fork) HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int k = 0; k < parts; k++) {
        if (!h->sub[k]) HIPCHK(hipStreamCreateWithFlags(&h->sub[k], hipStreamNonBlocking));
        if (!h->ev_join[k]) HIPCHK(hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
    }
    return POM_OK;
}

static bool runs_fresh(const PomBatch* h);
static bool runs_at_end(const PomBatch* h);
/* how many streams launches of a kind use: the sub-batch parts, or the streams chained launches rotate over (kind 0: what the
 * handle's several-tick calls will mostly be) */
static int streams_for(const PomBatch* h, int kind)
{
    const bool chains = h->issue_mode == POM_ISSUE_CHAIN && h->quad && h->chain_parts > 1;
    return kind == POM_KIND_CHAIN || (kind == 0 && chains) ? h->chain_parts : h->parts;
}

/* caller's stream -> sub-streams: everything already queued on the caller's stream happens before the parts.  Streams are
 * created when a kind of launch first needs them (a process has few hardware queues: a handle should not hold streams it never
 * uses); pom_batch_create and pom_batch_fork create what the handle's usual launches need. */
static int fork_parts(PomBatch* h, int kind)
{
    /* sub-batch launches (every stream its own tiles) and chained launches (every launch all tiles, pom_chain.h) must not be
     * in flight together: going from one kind to the other joins the streams first */
    const int need = streams_for(h, kind);
    if (h->forked && ((h->last_kind && kind && h->last_kind != kind) || need > h->forked_n))
        if (int jr = join_parts(h)) return jr;
    if (kind) h->last_kind = kind; /* kind 0: only the fork (pom_batch_fork), no launches yet */
    if (need == 1 || h->forked) return POM_OK;
    if (int er = ensure_sub_streams(h, need)) return er;
    HIPCHK(hipEventRecord(h->ev_fork, h->stream));
    for (int k = h->main_part; k < need; k++) HIPCHK(hipStreamWaitEvent(h->sub[k], h->ev_fork, 0));
    h->forked = true;
    h->forked_n = need;
    return POM_OK;
}
/* sub-streams -> caller's stream: whatever is queued on the caller's stream next sees all parts finished */
static int join_parts(PomBatch* h)
{
    if (!h->forked) return POM_OK;
    for (int k = h->main_part; k < h->forked_n; k++) {
        HIPCHK(hipEventRecord(h->ev_join[k], h->sub[k]));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join[k], 0));
    }
    h->forked = false;
    h->forked_n = 0;
    h->last_kind = 0;
    return POM_OK;
}

static int ensure_agent_mem(PomBatch* h)
{
    if (!h->agent_mem) {
        if (int jr = join_parts(h)) return jr; /* the next launch forks the sub-streams again, after this memset */
        HIPCHK(hipMalloc((void**)&h->agent_mem, (size_t)h->n_pad * 32));
        HIPCHK(hipMemsetAsync(h->agent_mem, 0, (size_t)h->n_pad * 32, h->stream));
    }
    return POM_OK;
}

static int fill_params(PomBatch* h, StepParams& p, const int32_t* moves_dev, uint64_t seed, int dist, int ticks)
{
    p.agent_mem = h->agent_mem;
    p.state = h->state;
    p.snap = h->snap;
    p.terminal = h->terminal;
    p.moves = moves_dev;
    p.wave_counters = h->wave_counters;
#if defined(POM_TRUNC)
    p.trunc = getenv("POM_TRUNC_AT") ? atoi(getenv("POM_TRUNC_AT")) : 990;
#endif
    p.n = h->n;
    p.n_pad = h->n_pad;
    p.env_offset = h->env_offset;
    p.seed = seed;
    p.tick0 = (uint32_t)h->tick;
    p.tick_base = h->tick_words + PomBatch::MAX_PARTS; /* the word that stays 0: outside a graph tick0 is the tick itself */
    p.dist = dist;
    p.ticks = ticks;
    p.mode = h->mode;
    p.auto_reset = h->auto_reset;
    p.max_steps = h->max_steps;
    p.episode = h->episode;
    p.board_seed = h->board_seed;
    p.fresh = h->fresh;
    p.block0 = p.block_end = 0;
    p.tile_seq = nullptr;
    p.chain_err = nullptr;
    p.chain_seq0 = 0;
    p.tape_len = 0;
    p.chain_wait_limit = 0;
    p.chain_rot = 0;
    p.obs_planes = nullptr;
    p.obs_agent_attrs = p.obs_env_attrs = nullptr;
    p.obs_dtype = p.obs_per_agent = 0;
#if defined(POM_DIAG)
    if (!h->diag) {
        HIPCHK(hipMalloc((void**)&h->diag, (size_t)h->n_waves * POM_PH_N * 8));
        HIPCHK(hipMemsetAsync(h->diag, 0, (size_t)h->n_waves * POM_PH_N * 8, h->stream));
    }
    p.diag = h->diag;
#endif
    return POM_OK;
}

/* Which instantiation of pom_step_kernel the handle runs.  Quad shape (the default): one instantiation per combination of
 * fresh boards / fused policy / reset at the end for launches of ONE tick, and the plain replay kernel for launches of
 * several ticks — the other several-tick combinations would need more than the 128 VGPRs that keep four wavefronts on a SIMD
 * (12-116 B of scratch each, round 2), so those modes always run one tick per launch (max_ticks_per_launch). */
typedef void (*PomStepKernel)(StepParams);
static bool runs_fresh(const PomBatch* h) { return h->fresh && h->mode == POM_MODE_ENV && h->auto_reset; }
static bool runs_at_end(const PomBatch* h) { return h->auto_reset == POM_RESET_AT_END && h->mode == POM_MODE_ENV; }
static int max_ticks_per_launch(const PomBatch* h, bool policy)
{
    return (!h->quad || (!policy && !runs_fresh(h) && !runs_at_end(h))) ? INT_MAX : 1;
}
static const void* step_kernel_for(const PomBatch* h, bool policy, bool one_tick)
{
    const bool fresh = runs_fresh(h), at_end = runs_at_end(h); /* at_end, policy: quad shape only, checked by the callers */
    PomStepKernel k;
    if (h->quad) {
        static const PomStepKernel single[8] = {
            pom_step_kernel<16, 4, false, false, false, true>, pom_step_kernel<16, 4, false, false, true, true>,
            pom_step_kernel<16, 4, false, true, false, true>,  pom_step_kernel<16, 4, false, true, true, true>,
            pom_step_kernel<16, 4, true, false, false, true>,  pom_step_kernel<16, 4, true, false, true, true>,
            pom_step_kernel<16, 4, true, true, false, true>,   pom_step_kernel<16, 4, true, true, true, true>};
        k = one_tick ? single[(fresh ? 4 : 0) | (policy ? 2 : 0) | (at_end ? 1 : 0)] : pom_step_kernel<16, 4, false, false, false, false>;
    } else if (h->epw == 64) {
        k = fresh ? pom_step_kernel<64, 1, true> : pom_step_kernel<64, 1, false>;
    } else if (h->epw == 32) {
        k = fresh ? pom_step_kernel<32, 1, true> : pom_step_kernel<32, 1, false>;
    } else {
        k = fresh ? pom_step_kernel<16, 1, true> : pom_step_kernel<16, 1, false>;
    }
    return reinterpret_cast<const void*>(k);
}

/* the one-tick kernel that also writes the observation of the state it leaves behind (explicit moves, quad shape) */
static const void* step_observe_kernel_for(const PomBatch* h)
{
    static const PomStepKernel k[4] = {
        pom_step_kernel<16, 4, false, false, false, true, false, true>, pom_step_kernel<16, 4, false, false, true, true, false, true>,
        pom_step_kernel<16, 4, true, false, false, true, false, true>,  pom_step_kernel<16, 4, true, false, true, true, false, true>};
    return reinterpret_cast<const void*>(k[(runs_fresh(h) ? 2 : 0) | (runs_at_end(h) ? 1 : 0)]);
}

/* one dispatch of the step kernel the handle is configured for, over tiles [p.block0, p.block_end) */
static hipError_t dispatch_step(const PomBatch* h, const StepParams& p, hipStream_t st, bool policy, hipEvent_t ev0, hipEvent_t ev1)
{
    const dim3 grid((unsigned)((p.block_end - p.block0 + POM_WPB - 1) / POM_WPB));
    StepParams q = p;
    void* args[1] = {&q};
    const void* kernel = p.obs_planes ? step_observe_kernel_for(h) : step_kernel_for(h, policy, p.ticks == 1);
    return hipExtLaunchKernel(kernel, grid, dim3(64 * POM_WPB), args, 0, st, ev0, ev1, 0);
}

/* `one_launch`: the whole batch in ONE launch on the caller's stream.  For steps that have to be joined with the caller's stream
 * every tick (explicit moves): forking into sub-streams and joining them again costs more than the overlap gains
 * (65,536 envs, MI355X: 23.3 us per step as one launch, 43.8 as two, 60.2 as three; profiles/r02a_explicit_streams.txt). */
struct PomObserveOut { /* pom_batch_step_device_observe: where the fused kernel writes the observation */
    void* planes;
    int32_t* agent_attrs;
    int32_t* env_attrs;
    int32_t dtype, per_agent;
};
static int launch_step(PomBatch* h, const int32_t* moves_dev, uint64_t seed, int dist, int ticks, bool policy = false, bool one_launch = false,
                       const PomObserveOut* obs = nullptr)
{
    StepParams p;
    if (int rc = chain_settle(h)) return rc; /* an ordinary launch after chained ones: every tile must stand on the tick the host thinks it does */
    if (int rc = fill_params(h, p, moves_dev, seed, dist, ticks)) return rc;
    if (obs) {
        p.obs_planes = obs->planes;
        p.obs_agent_attrs = obs->agent_attrs;
        p.obs_env_attrs = obs->env_attrs;
        p.obs_dtype = obs->dtype;
        p.obs_per_agent = obs->per_agent;
    }
    const int64_t tiles = h->n_pad / h->epw;
    const int parts = one_launch ? 1 : h->parts;
    int rc = one_launch ? join_parts(h) : fork_parts(h);
    if (rc) return rc;
    for (int k = 0; k < parts; k++) {
        const int64_t b0 = tiles * k / parts, b1 = tiles * (k + 1) / parts;
        if (b1 <= b0) continue;
        hipStream_t st = (parts == 1 || k < h->main_part) ? h->stream : h->sub[k];
        p.block0 = b0;
        p.block_end = b1;
        /* per-launch timing (pom_batch_profile): start / stop events attached to the dispatch itself, i.e. the kernel's own
         * duration as a profiler reports it, not the stream's period (events recorded around a launch also time the gap) */
        const bool prof = h->profiling && h->prof_n < PomBatch::PROF_RING;
        hipEvent_t ev0 = prof ? h->prof_ev[2 * h->prof_n] : nullptr, ev1 = prof ? h->prof_ev[2 * h->prof_n + 1] : nullptr;
        HIPCHK(dispatch_step(h, p, st, policy, ev0, ev1));
        if (prof) h->prof_n++;
    }
    return POM_OK;
}

/* ---- several ticks in one call ---------------------------------------------------------------------------------------------
 * The default up to 196,608 envs is POM_ISSUE_CHAIN (launch_many_chain below, pom_chain.h): one launch over all tiles per tick,
 * consecutive launches on different streams, a ticket word per tile ordering the tile's ticks — 9.2 us per step at 65,536 envs
 * where sub-batches take 14.5.  What follows is how SUB-BATCH launches are issued: what a handle does where launches are not
 * chained (several ticks per launch, batches from 196,608 envs up, the one-lane-per-env shapes, a handle asked for another mode).
 * A step is then `parts` launches (one per sub-batch, on parallel streams) every ~16 us at 65,536 envs: the host has ~5 us per
 * launch, and a call of K ticks is judged by how soon all parts' first launches are out and whether the queues stay fed.
 * Three ways to issue them (PomBatchOptions.issue_mode; results cannot depend on the choice: the same kernels with the same
 * arguments go to the same streams in the same per-stream order).  Measured on MI355X, 65,536 envs, 3 parts, per step
 * (profiles/r03_issue_modes.txt):
 *                                   20-tick call from an idle device (the bench driver's shape)      500-tick call
 *   POM_ISSUE_THREADS               18.0 us, run to run the same                                      15.5 us
 *   POM_ISSUE_DIRECT                17.2 .. 26.4 us: one thread issues all 60 launches (2.8 us each   15.6 - 16.0 us
 *                                   on a quiet host, then it is ahead of the device; on a busy one it is not)
 *   POM_ISSUE_GRAPH                 21.4 .. 23.5 us: queued in 48 us, but the replayed nodes run with   16.1 - 16.2 us
 *                                   wider gaps than plain launches
 * THREADS: part k's launches of ALL the ticks of the call are issued by a helper thread of its own (created on first use, one
 * per sub-stream part; the caller issues every part's first launch and the rest of its own part), the call returns when
 * everything is queued.  A helper that cannot be started is not an error: the calling thread issues its launches.
 * DIRECT: the calling thread issues everything, tick by tick; no library-owned threads.
 * GRAPH: part k's launches of POM_GRAPH_TICKS consecutive ticks are a HIP graph — a plain chain of kernel nodes, built once —
 * replayed on part k's stream with one hipGraphLaunch per part and chunk; no library-owned threads either.  The nodes'
 * arguments never change: a node carries its offset inside the chunk, and the tick the chunk starts at is a device word per
 * part (StepParams.tick_base) that a one-lane kernel sets on the part's stream in front of each replay.  (One graph holding
 * all parts as parallel branches was measured first: this runtime plays the branches one after the other — 18.5 us per step
 * in a 500-tick call.)  Ticks that do not fill a chunk are launched directly. */
struct PomIssuer {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, quit = false, busy = false;
    std::atomic<int> posted{0}; /* bumped with every job and by pom_batch_fork: a thread that has just worked (or was told that work
                                   is coming) polls this for up to a millisecond before it goes to sleep on the condition variable
                                   — a futex wake-up costs 10-30 us, a tenth of a 20-step burst */
    /* the job */
    StepParams p;
    hipStream_t st = nullptr;
    int launches = 0, ticks_per_launch = 1, last_ticks = 1;
    bool policy = false;
    hipError_t err = hipSuccess;
    /* a job of chained launches (launch_many_chain): these parameter blocks, one launch each, on `st` */
    std::vector<StepParams> chain_q;
    const void* chain_kernel = nullptr;
    unsigned chain_grid = 0;
    int chain_done = 0; /* how many of them were queued */
};

static void issuer_main(PomBatch* h, PomIssuer* w)
{
    (void)hipSetDevice(h->device);
    std::unique_lock<std::mutex> lk(w->mu);
    int seen = w->posted.load();
    for (;;) {
        if (!w->has_job && !w->quit) { /* poll briefly, then sleep */
            lk.unlock();
            const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(1);
            while (w->posted.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < until) {
            }
            lk.lock();
        }
        if (!w->has_job && !w->quit && w->posted.load() != seen) { /* woken by pom_batch_fork: a job is on its way */
            seen = w->posted.load();
            continue;
        }
        w->cv.wait(lk, [w, seen] { return w->has_job || w->quit || w->posted.load() != seen; });
        seen = w->posted.load();
        if (w->quit) return;
        if (!w->has_job) continue;
        w->has_job = false;
        if (w->chain_kernel) { /* chained launches: which launch plays which tick is the tiles' tickets' business, not the order here */
            hipError_t cerr = hipSuccess;
            int done = 0;
            for (StepParams& q : w->chain_q) {
                void* args[1] = {&q};
                cerr = hipExtLaunchKernel(w->chain_kernel, dim3(w->chain_grid), dim3(64 * POM_WPB), args, 0, w->st, nullptr, nullptr, 0);
                if (cerr != hipSuccess) break;
                done++;
            }
            w->chain_done = done;
            w->chain_kernel = nullptr;
            w->err = cerr;
            w->busy = false;
            w->cv.notify_all();
            continue;
        }
        StepParams p = w->p;
        hipError_t err = hipSuccess;
        for (int i = 0; i < w->launches && err == hipSuccess; i++) {
            p.ticks = i + 1 == w->launches ? w->last_ticks : w->ticks_per_launch;
            err = dispatch_step(h, p, w->st, w->policy, nullptr, nullptr);
            p.tick0 += (uint32_t)w->ticks_per_launch;
        }
        w->err = err;
        w->busy = false;
        w->cv.notify_all();
    }
}

static void stop_issuers(PomBatch* h)
{
    for (int k = 0; k < PomBatch::MAX_PARTS; k++) {
        PomIssuer* w = h->issuers[k];
        if (!w) continue;
        {
            std::lock_guard<std::mutex> g(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
        h->issuers[k] = nullptr;
    }
}


#ifndef POM_GRAPH_TICKS
#define POM_GRAPH_TICKS 20
#endif

__global__ void pom_set_word_kernel(uint32_t* dst, uint32_t v) { *dst = v; }

struct PomStepGraph {
    hipGraph_t graph[PomBatch::MAX_PARTS] = {};
    hipGraphExec_t exec[PomBatch::MAX_PARTS] = {};
    StepParams key;      /* everything the nodes were built with (tick0 / tick_base / blocks zeroed) */
    int launches = 0, parts = 0, ticks_per_launch = 0;
    bool policy = false;
    uint64_t used = 0;
};

static void free_graph(PomStepGraph* g)
{
    if (!g) return;
    for (int k = 0; k < PomBatch::MAX_PARTS; k++) {
        if (g->exec[k]) (void)hipGraphExecDestroy(g->exec[k]);
        if (g->graph[k]) (void)hipGraphDestroy(g->graph[k]);
    }
    delete g;
}

static void drop_graphs(PomBatch* h)
{
    for (int k = 0; k < PomBatch::MAX_GRAPHS; k++) {
        free_graph(h->graphs[k]);
        h->graphs[k] = nullptr;
    }
}

/* the graphs of `launches` launches per part for these parameters: from the cache, or built now (nullptr + *rc on failure) */
static PomStepGraph* step_graph(PomBatch* h, const StepParams& p, int launches, int ticks_per_launch, bool policy, int* rc)
{
    *rc = POM_OK;
    for (int k = 0; k < PomBatch::MAX_GRAPHS; k++) {
        PomStepGraph* g = h->graphs[k];
        if (g && g->launches == launches && g->parts == h->parts && g->ticks_per_launch == ticks_per_launch && g->policy == policy &&
            memcmp(&g->key, &p, sizeof p) == 0) {
            g->used = ++h->graph_clock;
            return g;
        }
    }
    PomStepGraph* g = new (std::nothrow) PomStepGraph();
    if (!g) {
        *rc = POM_E_NOMEM;
        return nullptr;
    }
    memcpy(&g->key, &p, sizeof p);
    g->launches = launches;
    g->parts = h->parts;
    g->ticks_per_launch = ticks_per_launch;
    g->policy = policy;
    hipError_t err = hipSuccess;
    const int64_t tiles = h->n_pad / h->epw;
    const void* fn = step_kernel_for(h, policy, ticks_per_launch == 1);
    for (int k = 0; k < h->parts && err == hipSuccess; k++) {
        const int64_t b0 = tiles * k / h->parts, b1 = tiles * (k + 1) / h->parts;
        if (b1 <= b0) continue;
        err = hipGraphCreate(&g->graph[k], 0);
        hipGraphNode_t prev = nullptr;
        for (int i = 0; i < launches && err == hipSuccess; i++) {
            StepParams q = p;
            q.block0 = b0;
            q.block_end = b1;
            q.ticks = ticks_per_launch;
            q.tick0 = (uint32_t)(i * ticks_per_launch); /* relative to *tick_base */
            q.tick_base = h->tick_words + k;
            void* args[1] = {&q};
            hipKernelNodeParams np;
            memset(&np, 0, sizeof np);
            np.func = const_cast<void*>(fn);
            np.gridDim = dim3((unsigned)((b1 - b0 + POM_WPB - 1) / POM_WPB));
            np.blockDim = dim3(64 * POM_WPB);
            np.kernelParams = args;
            hipGraphNode_t node = nullptr;
            err = hipGraphAddKernelNode(&node, g->graph[k], prev ? &prev : nullptr, prev ? 1 : 0, &np);
            prev = node;
        }
        if (err == hipSuccess) err = hipGraphInstantiate(&g->exec[k], g->graph[k], nullptr, nullptr, 0);
    }
    if (err != hipSuccess) {
        set_err("building the step graph", err);
        free_graph(g);
        *rc = POM_E_HIP;
        return nullptr;
    }
    int slot = 0; /* a free slot, or the least recently used one */
    for (int k = 0; k < PomBatch::MAX_GRAPHS; k++) {
        if (!h->graphs[k]) {
            slot = k;
            break;
        }
        if (h->graphs[k]->used < h->graphs[slot]->used) slot = k;
    }
    free_graph(h->graphs[slot]);
    g->used = ++h->graph_clock;
    h->graphs[slot] = g;
    return g;
}

/* POM_ISSUE_THREADS: the helper of part k, started on first use; nullptr if it cannot be had (the caller then issues that part) */
static PomIssuer* issuer_for(PomBatch* h, int k)
{
    if (h->issuers[k] || h->issuers_failed) return h->issuers[k];
    PomIssuer* w = new (std::nothrow) PomIssuer();
    if (w) {
        try {
            w->th = std::thread(issuer_main, h, w);
        } catch (...) { /* std::system_error: no thread to be had — nothing may cross the C boundary */
            delete w;
            w = nullptr;
        }
    }
    if (!w) h->issuers_failed = true;
    h->issuers[k] = w;
    return w;
}

/* THREADS / DIRECT: `launches` dispatches per part; the caller advances the tick */
static int launch_many_streams(PomBatch* h, const StepParams& p0, int launches, int ticks_per_launch, bool policy, bool threads)
{
    StepParams p = p0;
    const int64_t tiles = h->n_pad / h->epw;
    const int parts = h->parts;
    if (int rc = chain_settle(h)) return rc;
    if (int rc = fork_parts(h)) return rc;
    hipError_t err = hipSuccess;
    auto part_launch = [&](int k, const StepParams& base, int count) { /* `count` launches of part k from this thread */
        const int64_t b0 = tiles * k / parts, b1 = tiles * (k + 1) / parts;
        if (b1 <= b0) return;
        StepParams q = base;
        q.block0 = b0;
        q.block_end = b1;
        q.ticks = ticks_per_launch;
        const bool own = parts == 1 || k < h->main_part;
        for (int i = 0; i < count && err == hipSuccess; i++) {
            err = dispatch_step(h, q, own ? h->stream : h->sub[k], policy, nullptr, nullptr);
            q.tick0 += (uint32_t)ticks_per_launch;
        }
    };
    if (!threads) { /* tick by tick, the caller's own part first */
        for (int i = 0; i < launches && err == hipSuccess; i++) {
            for (int k = 0; k < parts; k++) part_launch(k, p, 1);
            p.tick0 += (uint32_t)ticks_per_launch;
        }
    } else {
        /* The first launch of EVERY part is issued right here, the caller's own part(s) first: a helper thread takes 10-25 us to
         * pick its job up (profiles/r02_region_trace.txt), and a part that starts a step late finishes a step late — alone on
         * the device.  The helpers get the remaining launches of their parts and have one step's time to wake up. */
        for (int pass = 0; pass < 2; pass++)
            for (int k = 0; k < parts; k++)
                if ((parts == 1 || k < h->main_part) == (pass == 0)) part_launch(k, p, 1);
        p.tick0 += (uint32_t)ticks_per_launch;
        const int rest = launches - 1;
        PomIssuer* started[PomBatch::MAX_PARTS] = {};
        for (int k = h->main_part; k < parts && rest > 0 && err == hipSuccess; k++) { /* the sub-stream parts: hand them to their threads */
            const int64_t b0 = tiles * k / parts, b1 = tiles * (k + 1) / parts;
            if (b1 <= b0) continue;
            PomIssuer* w = issuer_for(h, k);
            if (!w) { /* no helper: this thread does it */
                part_launch(k, p, rest);
                continue;
            }
            {
                std::lock_guard<std::mutex> g(w->mu);
                w->p = p;
                w->p.block0 = b0;
                w->p.block_end = b1;
                w->st = h->sub[k];
                w->launches = rest;
                w->ticks_per_launch = ticks_per_launch;
                w->last_ticks = ticks_per_launch;
                w->policy = policy;
                w->err = hipSuccess;
                w->busy = true;
                w->has_job = true;
                w->posted.fetch_add(1, std::memory_order_release);
            }
            w->cv.notify_all();
            started[k] = w;
        }
        for (int k = 0; k < (parts == 1 ? 1 : h->main_part) && rest > 0; k++) part_launch(k, p, rest); /* the rest of the caller's own part(s) */
        for (int k = 0; k < parts; k++) { /* everything is queued when the call returns */
            PomIssuer* w = started[k];
            if (!w) continue;
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return !w->busy; });
            if (w->err != hipSuccess && err == hipSuccess) err = w->err;
        }
    }
    if (err != hipSuccess) { /* some launches may be queued, others not: the handle's tick no longer describes its state */
        set_err("pom_step_kernel launch (the batch is in an undefined state: upload again or destroy it)", err);
        return POM_E_HIP;
    }
    return POM_OK;
}

#ifndef POM_CHAIN_LONG_CALL
#define POM_CHAIN_LONG_CALL 50
#endif
/* POM_ISSUE_CHAIN (pom_chain.h): `launches` one-tick launches, each over the WHOLE batch, dealt round-robin to the handle's
 * streams; the tiles' ticket words order the ticks.  *used = false: not available for this handle, nothing was launched, the
 * caller takes the ordinary path. */
static bool runs_chain(const PomBatch* h, bool policy, int ticks_per_launch)
{
    (void)policy; /* every one-tick instantiation of the quad shape has its chained twin */
    return h->issue_mode == POM_ISSUE_CHAIN && h->quad && h->chain_parts > 1 && ticks_per_launch == 1 && !(h->chain.tried && !h->chain.ok);
}
static const PomStepKernel* chain_kernels(bool chained)
{
    static const PomStepKernel chained_k[8] = {
        pom_step_kernel<16, 4, false, false, false, true, true>, pom_step_kernel<16, 4, false, false, true, true, true>,
        pom_step_kernel<16, 4, false, true, false, true, true>,  pom_step_kernel<16, 4, false, true, true, true, true>,
        pom_step_kernel<16, 4, true, false, false, true, true>,  pom_step_kernel<16, 4, true, false, true, true, true>,
        pom_step_kernel<16, 4, true, true, false, true, true>,   pom_step_kernel<16, 4, true, true, true, true, true>};
    static const PomStepKernel plain_k[8] = {
        pom_step_kernel<16, 4, false, false, false, true>, pom_step_kernel<16, 4, false, false, true, true>,
        pom_step_kernel<16, 4, false, true, false, true>,  pom_step_kernel<16, 4, false, true, true, true>,
        pom_step_kernel<16, 4, true, false, false, true>,  pom_step_kernel<16, 4, true, false, true, true>,
        pom_step_kernel<16, 4, true, true, false, true>,   pom_step_kernel<16, 4, true, true, true, true>};
    return chained ? chained_k : plain_k;
}

/* The check behind chained launches, and the recovery (pom_chain.h).  Joins the streams, lists the tiles some visitor could not
 * play (poisoned: their records stand on the last tick they really played), waits for the answer, and plays the missing ticks of
 * each such tile again — ordinary one-tile, one-tick launches of the twin kernel without the chain, with the parameters the
 * chained call was launched with (the log), on the caller's stream.  Afterwards every tile stands on the tick the host thinks it
 * does.  Costs one synchronisation of the caller's stream; free while there have been no chained launches since the last one. */
static int chain_settle(PomBatch* h)
{
    PomChain* c = &h->chain;
    if (!c->unverified || !c->ok) return POM_OK;
    if (int jr = join_parts(h)) return jr;
    const int64_t tiles = c->tiles;
    HIPCHK(hipMemsetAsync(c->aux + 1, 0, 4, h->stream));
    pom_chain_verify_kernel<<<dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, h->stream>>>(c->tile_seq, tiles, c->visits, c->aux);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->aux_host, c->aux, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    c->unverified = false;
    c->stat_settles++;
    const uint32_t flags = c->aux_host[0], bad = c->aux_host[1];
    if (!flags && !bad) {
        c->log.clear();
        return POM_OK;
    }
    if ((flags & POM_CHAIN_E_UNEVEN) || bad > (uint32_t)tiles) { /* not a tile left behind: launches that did not cover the tiles as assumed */
        c->ok = false;
        c->log.clear();
        snprintf(g_err, sizeof g_err, "chained launches did not visit every tile once (flags %u): the batch is in an undefined state — upload again "
                 "or destroy it; the handle launches sub-batches from here on", flags);
        return POM_E_HIP;
    }
    uint32_t* const list = new (std::nothrow) uint32_t[(size_t)2 * bad + 2]; /* (nothing may throw across the C boundary) */
    if (!list) {
        c->unverified = true; /* nothing has been replayed: the next call tries again */
        snprintf(g_err, sizeof g_err, "out of host memory while catching up tiles left behind by chained launches");
        return POM_E_NOMEM;
    }
    struct Free {
        uint32_t* p;
        ~Free() { delete[] p; }
    } free_list{list};
    /* from here on a failing runtime call leaves tiles half caught up: the handle says so instead of pretending otherwise (a retry
     * would replay from the old counts and play ticks twice) */
    struct Undefined {
        PomChain* c;
        bool armed;
        ~Undefined()
        {
            if (armed) {
                c->ok = false;
                c->log.clear();
            }
        }
    } undefined{c, true};
    if (bad) HIPCHK(hipMemcpy(list, c->aux + 2, (size_t)bad * 8, hipMemcpyDeviceToHost));
    if (getenv("POM_CHAIN_VERBOSE"))
        fprintf(stderr, "pom: chained launches left %u tile(s) behind (flags %u: %s%s%s); replaying their ticks\n", bad, flags,
                (flags & POM_CHAIN_E_TIMEOUT) ? "a wavefront waited out its limit " : "", (flags & POM_CHAIN_E_XCD) ? "a tile changed its XCD " : "",
                (flags & POM_CHAIN_E_TAPE) ? "a ticket outside the move tape" : "");
    const PomStepKernel* plain = chain_kernels(false);
    for (uint32_t k = 0; k < bad; k++) {
        const uint32_t tile = list[2 * k], stored = list[2 * k + 1];
        for (uint32_t v = stored; v != c->visits; v++) {
            const PomChainCall* call = nullptr;
            for (const PomChainCall& e : c->log)
                if (v - e.visit0 < e.launches) call = &e;
            if (!call) { /* cannot happen: every chained launch since the last settle is in the log */
                snprintf(g_err, sizeof g_err, "a tile left behind by chained launches cannot be replayed: the batch is in an undefined state");
                return POM_E_HIP;
            }
            StepParams q = call->p;
            const uint32_t d = v - q.chain_seq0; /* visits after the call's first */
            q.block0 = tile;
            q.block_end = (int64_t)tile + 1;
            q.ticks = 1;
            q.tick0 += d;
            q.tick_base = h->tick_words + PomBatch::MAX_PARTS;
            if (q.moves) q.moves += (int64_t)d * q.n * 4;
            q.tile_seq = nullptr;
            q.chain_err = nullptr;
            q.chain_seq0 = 0;
            q.tape_len = 0;
            const PomStepKernel kernel = plain[(runs_fresh(h) ? 4 : 0) | (call->policy ? 2 : 0) | (runs_at_end(h) ? 1 : 0)];
            void* args[1] = {&q};
            HIPCHK(hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), dim3(1), dim3(64 * POM_WPB), args, 0, h->stream, nullptr, nullptr, 0));
            c->stat_ticks_replayed++;
        }
        c->stat_tiles_recovered++;
    }
    /* the words start over: every tile is level again; only now are the flags that asked for this forgotten */
    HIPCHK(hipMemsetAsync(c->tile_seq, 0, (size_t)tiles * 8 * POM_CHAIN_WORD_STRIDE, h->stream));
    HIPCHK(hipMemsetAsync(c->aux, 0, 4, h->stream));
    undefined.armed = false;
    c->visits = 0;
    c->log.clear();
    if (flags & (POM_CHAIN_E_XCD | POM_CHAIN_E_TAPE)) c->ok = false; /* structural, not a matter of timing: no more chained launches on this handle */
    return POM_OK;
}

/* POM_ISSUE_CHAIN (pom_chain.h): `launches` one-tick launches, each over the WHOLE batch, dealt round-robin to the handle's
 * streams; the tiles' ticket words order the ticks.  p0.moves != nullptr: a move tape of `launches` ticks.  *used = false: not
 * available for this handle, nothing was launched, the caller takes the ordinary path. */
static int launch_many_chain(PomBatch* h, const StepParams& p0, int launches, bool policy, bool* used)
{
    *used = false;
    PomChain* c = &h->chain;
    const int64_t tiles = h->n_pad / h->epw;
    if (!c->tried) { /* a handle that was not created for chained launches (pom_batch_set_streams made them possible later) */
        if (int jr = join_parts(h)) return jr;
        if (!chain_setup(c, tiles, h->stream) && getenv("POM_CHAIN_VERBOSE"))
            fprintf(stderr, "pom: chained launches are not available on this device (allocation failed or the workgroup -> XCD probe did not find the "
                            "eight-XCD round-robin); launching sub-batches\n");
    }
    if (!c->ok) return POM_OK;
    /* the fields of the tile words must not run into each other: after 2^27 visits (20 minutes of stepping) the words start over.
     * (POM_CHAIN_RESET_AT: a smaller number, so that tests get to see it happen).  The log of calls is bounded the same way. */
    static const uint64_t reset_at = getenv("POM_CHAIN_RESET_AT") ? (uint64_t)atoll(getenv("POM_CHAIN_RESET_AT")) : (uint64_t)(1u << 27);
    if ((uint64_t)c->visits + (uint64_t)launches >= (reset_at < (1u << 27) && reset_at > 0 ? reset_at : (uint64_t)(1u << 27)) || c->log.size() >= 256) {
        if (int sr = chain_settle(h)) return sr;
        if (!c->ok) return POM_OK;
        if (int jr = join_parts(h)) return jr;
        HIPCHK(hipMemsetAsync(c->tile_seq, 0, (size_t)tiles * 8 * POM_CHAIN_WORD_STRIDE, h->stream));
        c->visits = 0;
    }
    StepParams p = p0; /* the same for every launch of the call: which tick a wavefront plays follows from its ticket */
    p.block0 = 0;
    p.block_end = tiles;
    p.ticks = 1;
    p.tile_seq = c->tile_seq;
    p.chain_err = c->aux;
    p.chain_seq0 = c->visits;
    p.tape_len = p.moves ? (uint32_t)launches : 0u;
    p.chain_wait_limit = c->wait_limit;
    const PomStepKernel kernel = chain_kernels(true)[(runs_fresh(h) ? 4 : 0) | (policy ? 2 : 0) | (runs_at_end(h) ? 1 : 0)];
    /* Launches in flight together must be interchangeable — "the j-th visitor of a tile plays the tile's j-th tick" holds only
     * if every launch would play that tick the same way: the same kernel, seed, move distribution, mode ... and the same offset
     * between ticks and visits.  A call that differs in any of that from the chained launches still in flight waits for them
     * (tests/test_gpu_chain.py: random sequences of calls with a seed of their own each).  A move tape belongs to its call: the
     * launches of a tape call never overlap another call's (and the tape was written on the caller's stream: the fork below
     * orders this call's launches behind it). */
    StepParams key = p;
    key.tick0 = p.tick0 - p.chain_seq0;
    key.chain_seq0 = 0;
    const bool continues = !p.moves && kernel == c->last_kernel && memcmp(&key, &c->last_key, sizeof key) == 0;
    if (h->forked && h->last_kind == POM_KIND_CHAIN && !continues)
        if (int jr = join_parts(h)) return jr;
    if (p.moves && h->forked)
        if (int jr = join_parts(h)) return jr;
    c->last_key = key;
    c->last_kernel = kernel;
    if (int rc = fork_parts(h, POM_KIND_CHAIN)) return rc;
    const dim3 grid((unsigned)(((tiles + POM_WPB - 1) / POM_WPB + 7) / 8 * 8)); /* a multiple of 8: every XCD gets as many workgroups as it has tiles */
    /* how many streams: a third launch in flight pays once the pipeline runs and costs while it fills and drains.  65,536 envs, us
     * per step on two / three streams at the round-4 kernels (with the rotation below on two): 30 ticks 10.47 / 11.13, 40 ticks
     * 10.20 / 10.76, 60 ticks 10.05 / 9.72, 100 ticks 9.79 / 9.64, 300 ticks 9.7 / 9.0.  Any mix is fine: the tickets order the
     * ticks, not the streams. */
    const int use = !h->chain_auto ? h->chain_parts : launches >= POM_CHAIN_LONG_CALL ? 3 : 2;
    /* On two streams, where one launch fills the chip's wavefront slots exactly (65,536 envs: 4,096 tiles, 16 slots on each of 256
     * CUs), every launch takes its XCD's tiles starting a sixteenth of them BEHIND where the launch before it started (a rotation
     * of the workgroup -> tile map: still every tile once per launch, still on its XCD).  With the same order in every launch the
     * two launches in flight meet on the same tiles more often than they must — the later visitor spins in a slot for the rest of
     * the earlier one's tick; short calls: 20 ticks 11.65 -> 11.31 us per step (ten runs each, medians; another box 11.75 -> 11.16),
     * 10 ticks 13.44 -> 13.08, 39 ticks unchanged.  Not on three streams (long calls: 9.0 -> 9.8 - 11.0 us, the launches there trail
     * each other by a whole cycle and the common order is what keeps them apart) and not at other sizes (32,768 and 16,384 envs: no
     * effect; 131,072: worse).  profiles/r04_chain_rotation.txt.  POM_CHAIN_ROT_DIV: the divisor (0: off). */
    static const int rot_div = getenv("POM_CHAIN_ROT_DIV") ? atoi(getenv("POM_CHAIN_ROT_DIV")) : 16;
    const uint32_t per_xcd = (uint32_t)(((tiles + POM_WPB - 1) / POM_WPB + 7) / 8);
    const bool fills_the_chip = c->wave_slots > 0 && tiles <= c->wave_slots && tiles * 4 >= (int64_t)c->wave_slots * 3;
    const uint32_t back = (use == 2 && fills_the_chip && rot_div > 0) ? per_xcd / (uint32_t)rot_div : 0u;
    h->chain_last_use = use;
    int issued = 0;
    hipError_t err = hipSuccess;
    auto params_of = [&](int i) { /* launch number i of the call */
        StepParams q = p;
        q.chain_rot = back ? (uint32_t)(((uint64_t)(c->visits + (uint32_t)i) * (uint64_t)(per_xcd - back)) % per_xcd) : 0u;
        return q;
    };
    const uint32_t turn0 = c->turn;
    auto part_of = [&](int i) { return (int)((turn0 + (uint32_t)i) % (uint32_t)use); };
    /* Who issues: the first launch of every stream the calling thread, at once; then the sub-streams' remaining launches their helper
     * threads (the ones sub-batch launches use: created on first use, spinning for a millisecond after pom_batch_fork / a job) while the
     * calling thread issues its own stream's.  A launch costs the host 2.5 - 3 us, sometimes 6 - 9: with one thread a 20-step call is
     * queued in 50 - 60 us on most runs and in 120 - 180 us on one in five, late enough for the device to wait for its launches
     * (profiles/r05_issue_helpers.txt).  Any interleaving is fine: the tiles' tickets order the ticks, not the launches' order. */
    static const bool helpers_on = !(getenv("POM_CHAIN_HELPERS") && atoi(getenv("POM_CHAIN_HELPERS")) == 0);
    const bool helpers = helpers_on && !h->profiling && use >= 2 && launches >= 3 * use && !h->issuers_failed;
    if (!helpers) {
        for (; issued < launches; issued++) {
            const int part = part_of(issued);
            hipStream_t st = part < h->main_part ? h->stream : h->sub[part];
            const bool prof = h->profiling && h->prof_n < PomBatch::PROF_RING;
            hipEvent_t ev0 = prof ? h->prof_ev[2 * h->prof_n] : nullptr, ev1 = prof ? h->prof_ev[2 * h->prof_n + 1] : nullptr;
            StepParams q = params_of(issued);
            void* args[1] = {&q};
            err = hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, dim3(64 * POM_WPB), args, 0, st, ev0, ev1, 0);
            if (err != hipSuccess) break;
            if (prof) h->prof_n++;
        }
    } else {
        auto launch_here = [&](int i) {
            const int part = part_of(i);
            StepParams q = params_of(i);
            void* args[1] = {&q};
            const hipError_t e = hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, dim3(64 * POM_WPB), args, 0,
                                                    part < h->main_part ? h->stream : h->sub[part], nullptr, nullptr, 0);
            if (e == hipSuccess) issued++;
            else if (err == hipSuccess) err = e;
        };
        for (int i = 0; i < use && err == hipSuccess; i++) launch_here(i); /* every stream's first launch: now */
        PomIssuer* started[PomBatch::MAX_PARTS] = {};
        for (int part = h->main_part; part < use && err == hipSuccess; part++) {
            PomIssuer* w = issuer_for(h, part);
            if (!w) continue; /* (no thread to be had: this thread issues that stream's launches below) */
            {
                std::lock_guard<std::mutex> g(w->mu);
                w->chain_q.clear();
                for (int i = use; i < launches; i++)
                    if (part_of(i) == part) w->chain_q.push_back(params_of(i));
                w->chain_kernel = reinterpret_cast<const void*>(kernel);
                w->chain_grid = grid.x;
                w->chain_done = 0;
                w->st = h->sub[part];
                w->err = hipSuccess;
                w->busy = true;
                w->has_job = true;
                w->posted.fetch_add(1, std::memory_order_release);
            }
            w->cv.notify_all();
            started[part] = w;
        }
        for (int i = use; i < launches && err == hipSuccess; i++) { /* the caller's own stream(s), and those without a helper */
            const int part = part_of(i);
            if (part < h->main_part || !started[part]) launch_here(i);
        }
        for (int part = 0; part < use; part++) { /* everything is queued when the call returns */
            PomIssuer* w = started[part];
            if (!w) continue;
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return !w->busy; });
            issued += w->chain_done;
            if (w->err != hipSuccess && err == hipSuccess) err = w->err;
        }
    }
    c->turn = turn0 + (uint32_t)launches;
    /* what was issued is accounted for even if a launch failed half-way: the tiles' words, the log and the host's tick agree */
    if (issued > 0) {
        if (continues && !c->log.empty() && c->log.back().visit0 + c->log.back().launches == c->visits && c->log.back().policy == policy) {
            c->log.back().launches += (uint32_t)issued; /* the same play goes on: one entry */
        } else {
            PomChainCall e;
            e.visit0 = c->visits;
            e.launches = (uint32_t)issued;
            e.p = p;
            e.policy = policy;
            c->log.push_back(e);
        }
        c->visits += (uint32_t)issued;
        c->stat_launches += issued;
        c->unverified = true;
        h->tick += (uint64_t)issued;
    }
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "pom_step_kernel launch: %s (%d of the call's %d ticks were queued and will be played; the rest were not)",
                 hipGetErrorString(err), issued, launches);
        *used = true;
        return POM_E_HIP;
    }
    *used = true;
    return POM_OK;
}

/* `launches` dispatches per part, ticks_per_launch ticks each; advances h->tick by what was queued */
static int launch_many(PomBatch* h, uint64_t seed, int dist, int launches, int ticks_per_launch, bool policy)
{
    static const int chunk = getenv("POM_GRAPH_TICKS") ? atoi(getenv("POM_GRAPH_TICKS")) : POM_GRAPH_TICKS;
    int done = 0;
    if (runs_chain(h, policy, ticks_per_launch) && launches >= 1) {
        StepParams p;
        memset(&p, 0, sizeof p); /* consecutive calls' parameters are compared byte for byte */
        if (int rc = fill_params(h, p, nullptr, seed, dist, 1)) return rc;
        bool used = false;
        if (int rc = launch_many_chain(h, p, launches, policy, &used)) return rc;
        if (used) return POM_OK; /* (the tick was advanced by what was queued) */
    }
    if (int rc = chain_settle(h)) return rc;
    if (h->issue_mode == POM_ISSUE_GRAPH && chunk >= 2 && launches >= chunk && !h->profiling) {
        StepParams p;
        memset(&p, 0, sizeof p); /* the cache compares the bytes */
        if (int rc = fill_params(h, p, nullptr, seed, dist, ticks_per_launch)) return rc;
        p.tick0 = 0;
        p.tick_base = nullptr;
        int rc = POM_OK;
        PomStepGraph* g = step_graph(h, p, chunk, ticks_per_launch, policy, &rc);
        if (!g) return rc;
        if (int fr = fork_parts(h)) return fr;
        for (; launches - done >= chunk; done += chunk) {
            for (int pass = 0; pass < 2; pass++) { /* the caller's own stream first: its part starts without a wait on the fork event */
                for (int k = 0; k < h->parts; k++) {
                    const bool own = h->parts == 1 || k < h->main_part;
                    if (own != (pass == 0) || !g->exec[k]) continue;
                    hipStream_t st = own ? h->stream : h->sub[k];
                    pom_set_word_kernel<<<dim3(1), dim3(1), 0, st>>>(h->tick_words + k, (uint32_t)h->tick);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipGraphLaunch(g->exec[k], st));
                }
            }
            h->tick += (uint64_t)chunk * (uint64_t)ticks_per_launch;
        }
    }
    if (done == launches) return POM_OK;
    if (h->profiling) { /* per-launch events: launch_step attaches them */
        for (; done < launches; done++) {
            if (int rc = launch_step(h, nullptr, seed, dist, ticks_per_launch, policy)) return rc;
            h->tick += (uint64_t)ticks_per_launch;
        }
        return POM_OK;
    }
    StepParams p;
    if (int rc = fill_params(h, p, nullptr, seed, dist, ticks_per_launch)) return rc;
    /* (a handle of chained launches issues what cannot be chained — policy, fresh boards, several ticks per launch — with the helper threads) */
    const bool threads = (h->issue_mode == POM_ISSUE_THREADS || h->issue_mode == POM_ISSUE_CHAIN) && h->parts > 1 && launches - done >= 2;
    if (int rc = launch_many_streams(h, p, launches - done, ticks_per_launch, policy, threads)) return rc;
    h->tick += (uint64_t)(launches - done) * (uint64_t)ticks_per_launch;
    return POM_OK;
}

static int launch_policy(PomBatch* h, uint64_t seed)
{
    if (int rc = chain_settle(h)) return rc;
    if (int rc = ensure_agent_mem(h)) return rc;
    PolicyParams p;
    p.state = h->state;
    p.snap = h->snap;
    p.agent_mem = h->agent_mem;
    p.moves = h->moves_dev;
    p.n = h->n;
    p.n_pad = h->n_pad;
    p.env_offset = h->env_offset;
    p.seed = seed;
    p.tick = (uint32_t)h->tick;
    p.mode = h->mode;
    p.auto_reset = h->auto_reset;
    p.episode = h->episode;
    p.board_seed = h->board_seed;
    p.fresh = h->fresh;
#if defined(POM_DIAG)
    if (!h->diag_pol) {
        HIPCHK(hipMalloc((void**)&h->diag_pol, (size_t)(h->n_pad / 16) * POM_PP_N * 8));
        HIPCHK(hipMemsetAsync(h->diag_pol, 0, (size_t)(h->n_pad / 16) * POM_PP_N * 8, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    p.diag = h->diag_pol;
#endif
    /* same split and the same streams as the tick, so that part k's policy -> tick -> policy chain pipelines */
    const int64_t tiles = h->n_pad / 16, step_tiles = h->n_pad / h->epw;
    int rc = fork_parts(h);
    if (rc) return rc;
    for (int k = 0; k < h->parts; k++) {
        /* the tick's part k covers envs [step_tiles*k/parts, ...) * epw: use the same env boundaries */
        const int64_t e0 = step_tiles * k / h->parts * h->epw, e1 = step_tiles * (k + 1) / h->parts * h->epw;
        const int64_t b0 = e0 / 16, b1 = e1 / 16;
        if (b1 <= b0) continue;
        (void)tiles;
        p.block0 = b0;
        hipStream_t st = (h->parts == 1 || k < h->main_part) ? h->stream : h->sub[k];
        pom_policy_kernel<<<dim3((unsigned)(b1 - b0)), dim3(64), 0, st>>>(p);
        HIPCHK(hipGetLastError());
    }
    return POM_OK;
}

#endif /* POM_RUNTIME_H_ */
