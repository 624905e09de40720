/*
 * pom_batch.hip — the C-ABI of include/pom_batch.h (libpom_batch.so): every entry point the reference-side binding links
 * (INTEGRATION.md).  Kernels: pom_kernels.h; streams, launches and the handle: pom_runtime.h.  gfx950 only; there is no CPU
 * path: without a HIP device pom_batch_create fails with POM_E_HIP.
 */
#include <atomic>
#include <mutex>

#include "pom_runtime.h"

extern "C" {

const char* pom_last_error(void) { return g_err; }

int pom_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pom_batch_destroy(PomBatch* h)
{
    if (!h) return POM_E_ARG;
    stop_issuers(h);
    (void)hipSetDevice(h->device);
    for (int k = 0; k < PomBatch::MAX_PARTS; k++)
        if (h->sub[k]) (void)hipStreamSynchronize(h->sub[k]);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    chain_destroy(&h->chain); /* after the streams have drained: launches in flight still use the tiles' words */
    drop_graphs(h);
    for (int k = 0; k < PomBatch::MAX_PARTS; k++) {
        if (h->sub[k]) (void)hipStreamDestroy(h->sub[k]);
        if (h->ev_join[k]) (void)hipEventDestroy(h->ev_join[k]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int k = 0; k < 2 * PomBatch::PROF_RING; k++)
        if (h->prof_ev[k]) (void)hipEventDestroy(h->prof_ev[k]);
    (void)hipFree(h->state);
    (void)hipFree(h->snap);
    (void)hipFree(h->terminal);
    (void)hipFree(h->moves_dev);
    (void)hipFree(h->agent_mem);
    (void)hipFree(h->episode);
    (void)hipFree(h->staging);
    (void)hipFree(h->wave_counters);
    (void)hipFree(h->totals_dev);
    (void)hipFree(h->first_bad);
    (void)hipFree(h->tick_words);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return POM_OK;
}

int pom_batch_create(PomBatch** out, int64_t n_envs, const PomBatchOptions* opts)
{
    if (!out || n_envs <= 0 || n_envs > (int64_t)1 << 30) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: bad arguments");
        return POM_E_ARG;
    }
    *out = nullptr;
    PomBatchOptions o;
    memset(&o, 0, sizeof o);
    o.mode = POM_MODE_ENV;
    if (opts) {
        if (opts->struct_size <= 0 || opts->struct_size > (int)sizeof o) {
            snprintf(g_err, sizeof g_err, "pom_batch_create: options struct_size %d not understood", opts->struct_size);
            return POM_E_ARG;
        }
        memcpy(&o, opts, (size_t)opts->struct_size);
    }
    if (o.mode != POM_MODE_RAW && o.mode != POM_MODE_ENV) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: bad mode %d", o.mode);
        return POM_E_ARG;
    }
    if (o.auto_reset < POM_RESET_OFF || o.auto_reset > POM_RESET_AT_END) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: auto_reset must be 0 (off), 1 (at the start of the next tick) or 2 (at the end of the tick)");
        return POM_E_ARG;
    }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (o.device < 0 || o.device >= ndev) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: device %d of %d", o.device, ndev);
        return POM_E_HIP;
    }
    HIPCHK(hipSetDevice(o.device));
    PomBatch* h = new (std::nothrow) PomBatch();
    if (!h) return POM_E_NOMEM;
    h->device = o.device;
    h->n = n_envs;
    h->n_pad = (n_envs + 63) / 64 * 64;
    h->n_waves = h->n_pad / 16;
    /* Kernel shape.  Default: the quad kernel (16 envs per wavefront, 4 adjacent lanes per env) — fastest at every batch
     * size measured (profiles/r01_quad.txt).  The one-lane-per-env variants (64 / 32 / 16 envs per wavefront) stay
     * selectable; all produce identical results. */
    h->epw = 16;
    h->quad = true;
    if (o.lanes_per_env != 0 && o.lanes_per_env != 1 && o.lanes_per_env != 4) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: lanes_per_env must be 0, 1 or 4");
        delete h;
        return POM_E_ARG;
    }
    if (o.envs_per_wave != 0 && o.envs_per_wave != 16 && o.envs_per_wave != 32 && o.envs_per_wave != 64) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: envs_per_wave must be 0, 16, 32 or 64");
        delete h;
        return POM_E_ARG;
    }
    if (o.envs_per_wave != 0) {
        h->epw = o.envs_per_wave;
        h->quad = h->epw == 16 && o.lanes_per_env != 1;
    } else if (o.lanes_per_env == 1) {
        h->quad = false;
        h->epw = h->n_pad <= 8192 ? 16 : 32;
    }
    if (o.lanes_per_env == 4 && !h->quad) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: lanes_per_env 4 needs envs_per_wave 16 (or 0)");
        delete h;
        return POM_E_ARG;
    }
    if (const char* ev = getenv("POM_EPW")) { /* tuning overrides for sweeps */
        const int v = atoi(ev);
        if (v == 16 || v == 32 || v == 64) {
            h->epw = v;
            h->quad = false;
        }
    }
    if (const char* ev = getenv("POM_QUAD")) h->quad = atoi(ev) != 0 && h->epw == 16;
    /* sub-batches per step: part 0 on the caller's stream, the others on internal streams.  Measured on MI355X at 65,536
     * envs: one launch 26.4 us, two parts 22.3, three 20.4; FOUR concurrent streams of one process serialize on this stack
     * (36 us; profiles/r01_streams.txt), and other streams of the process (RCCL) count against that budget, so the default
     * stays at three only for batches where it matters and a caller can measure (pom_batch_set_streams, as bench.py does). */
    h->parts = h->n_pad >= 49152 ? 3 : h->n_pad >= 8192 ? 2 : 1;
    if (o.streams >= 1 && o.streams <= PomBatch::MAX_PARTS) h->parts = o.streams;
    else if (o.streams != 0) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: streams must be 0..%d", (int)PomBatch::MAX_PARTS);
        delete h;
        return POM_E_ARG;
    }
    /* policy and tick in one kernel: faster at every batch size since the wavefront-cooperative searches (round 3: 33.7 / 39.0 /
     * 55.7 / 109.5 us per step fused against 42.2 / 47.9 / 62.8 / 112.8 as two kernels at 32,768 .. 262,144 envs).  Round 1
     * measured the two kernels 3 % ahead at 262,144 envs (profiles/r01_fuse.txt). */
    h->fuse_policy = true;
    if (const char* ev = getenv("POM_FUSE")) h->fuse_policy = atoi(ev) != 0;
    if (const char* ev = getenv("POM_MAIN_PART")) h->main_part = atoi(ev) != 0;
    if (o.issue_mode < POM_ISSUE_AUTO || o.issue_mode > POM_ISSUE_CHAIN) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: issue_mode must be one of POM_ISSUE_*");
        delete h;
        return POM_E_ARG;
    }
    /* AUTO: chained launches where they pay (measured up to 131,072 envs, even at 262,144: scripts/experiments/chain/sizes.sh)
     * and can be had (the one-tick replay shape: pom_runtime.h runs_chain); the helper threads otherwise */
    h->issue_mode = o.issue_mode != POM_ISSUE_AUTO ? o.issue_mode : (h->quad && h->n_pad <= 196608) ? POM_ISSUE_CHAIN : POM_ISSUE_THREADS;
    if (const char* ev = getenv("POM_ISSUE")) {
        if (!strcmp(ev, "direct")) h->issue_mode = POM_ISSUE_DIRECT;
        else if (!strcmp(ev, "threads")) h->issue_mode = POM_ISSUE_THREADS;
        else if (!strcmp(ev, "graph")) h->issue_mode = POM_ISSUE_GRAPH;
        else if (!strcmp(ev, "chain")) h->issue_mode = POM_ISSUE_CHAIN;
    }
    if (const char* ev = getenv("POM_STREAMS")) {
        const int v = atoi(ev);
        if (v >= 1 && v <= PomBatch::MAX_PARTS) h->parts = v;
    }
    if ((int64_t)h->parts > h->n_pad / h->epw) h->parts = (int)(h->n_pad / h->epw);
    /* chained launches rotate over two streams (short calls) or three (long ones) unless the caller asked for a number (options /
     * POM_STREAMS / pom_batch_set_streams: then that many; one stream = launches in a row, nothing to chain) */
    h->chain_auto = !(o.streams >= 1 || getenv("POM_STREAMS"));
    h->chain_parts = h->chain_auto ? 3 : h->parts;
    if (const char* ev = getenv("POM_CHAIN_STREAMS")) {
        const int v = atoi(ev);
        if (v >= 1 && v <= PomBatch::MAX_PARTS) {
            h->chain_parts = v;
            h->chain_auto = false;
        }
    }
    if (o.auto_reset == POM_RESET_AT_END && !h->quad) {
        snprintf(g_err, sizeof g_err, "pom_batch_create: auto_reset = POM_RESET_AT_END is built for the default kernel shape "
                 "(envs_per_wave 16, lanes_per_env 4) only");
        delete h;
        return POM_E_ARG;
    }
    h->mode = o.mode;
    h->auto_reset = o.auto_reset;
    h->max_steps = o.max_steps;
    h->env_offset = o.env_offset;
    h->fresh = o.fresh_boards != 0;
    h->board_seed = o.board_seed;
    h->staging_envs = h->n_pad < 16384 ? h->n_pad : 16384;
#define ALLOC(ptr, bytes)                                              \
    do {                                                               \
        hipError_t e_ = hipMalloc((void**)&(ptr), (size_t)(bytes));    \
        if (e_ != hipSuccess) {                                        \
            set_err("hipMalloc", e_);                                  \
            pom_batch_destroy(h);                                      \
            return e_ == hipErrorOutOfMemory ? POM_E_NOMEM : POM_E_HIP; \
        }                                                              \
    } while (0)
    if (o.stream) {
        h->stream = (hipStream_t)o.stream;
    } else {
        hipError_t e_ = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e_ != hipSuccess) {
            set_err("hipStreamCreate", e_);
            delete h;
            return POM_E_HIP;
        }
        h->own_stream = true;
    }
    if (ensure_sub_streams(h, streams_for(h, 0)) != POM_OK) { /* what its usual launches need; the rest when first needed */
        pom_batch_destroy(h);
        return POM_E_HIP;
    }
    const size_t rec_bytes = (size_t)POM_REC_DWORDS * 4 * (size_t)h->n_pad;
    ALLOC(h->state, rec_bytes);
    ALLOC(h->snap, rec_bytes);
    ALLOC(h->moves_dev, (size_t)h->n_pad * 16);
    ALLOC(h->staging, (size_t)h->staging_envs * POM_STATE_BYTES);
    ALLOC(h->wave_counters, (size_t)h->n_waves * POM_CNT_N * 8);
    ALLOC(h->totals_dev, POM_CNT_N * 8);
    ALLOC(h->first_bad, sizeof(int));
    ALLOC(h->episode, (size_t)h->n_pad * 4);
    ALLOC(h->tick_words, (PomBatch::MAX_PARTS + 1) * sizeof(uint32_t));
    if (h->auto_reset == POM_RESET_AT_END) ALLOC(h->terminal, rec_bytes);
#undef ALLOC
    /* all-zero records are inert blank boards; padded envs are marked finished */
    hipError_t e1 = hipMemsetAsync(h->state, 0, rec_bytes, h->stream);
    hipError_t e2 = hipMemsetAsync(h->snap, 0, rec_bytes, h->stream);
    hipError_t e3 = hipMemsetAsync(h->moves_dev, 0, (size_t)h->n_pad * 16, h->stream);
    hipError_t e4 = hipMemsetAsync(h->wave_counters, 0, (size_t)h->n_waves * POM_CNT_N * 8, h->stream);
    if (e4 == hipSuccess) e4 = hipMemsetAsync(h->episode, 0, (size_t)h->n_pad * 4, h->stream);
    if (e4 == hipSuccess) e4 = hipMemsetAsync(h->tick_words, 0, (PomBatch::MAX_PARTS + 1) * sizeof(uint32_t), h->stream);
    if (e4 == hipSuccess && h->terminal) e4 = hipMemsetAsync(h->terminal, 0, rec_bytes, h->stream);
    hipError_t e5 = hipStreamSynchronize(h->stream);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
        set_err("initial memset", e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3 != hipSuccess ? e3 : e4 != hipSuccess ? e4 : e5);
        pom_batch_destroy(h);
        return POM_E_HIP;
    }
    /* chained launches: the tiles' words and the probe of the workgroup -> XCD pattern, here rather than in the first step (the
     * probe synchronises the stream); where they cannot be had the handle launches sub-batches */
    if (h->issue_mode == POM_ISSUE_CHAIN && h->quad && h->chain_parts > 1) {
        if (!chain_setup(&h->chain, h->n_pad / h->epw, h->stream) && getenv("POM_CHAIN_VERBOSE"))
            fprintf(stderr, "pom: chained launches are not available on this device (allocation failed or the workgroup -> XCD probe did not find the "
                            "eight-XCD round-robin); launching sub-batches\n");
    }
    *out = h;
    return POM_OK;
}

int64_t pom_batch_size(const PomBatch* h) { return h ? h->n : -1; }


int pom_batch_upload(PomBatch* h, const void* states, int64_t first, int64_t count)
{
    int rc = check_range(h, first, count);
    if (rc || !states) return rc ? rc : POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    const int big = INT_MAX;
    HIPCHK(hipMemcpyAsync(h->first_bad, &big, sizeof big, hipMemcpyHostToDevice, h->stream));
    int64_t bad_env = -1;
    for (int64_t off = 0; off < count; off += h->staging_envs) {
        const int64_t c = count - off < h->staging_envs ? count - off : h->staging_envs;
        HIPCHK(hipMemcpyAsync(h->staging, (const char*)states + off * POM_STATE_BYTES, (size_t)c * POM_STATE_BYTES,
                              hipMemcpyHostToDevice, h->stream));
        pom_pack_kernel<<<dim3((unsigned)((c + 63) / 64)), dim3(64), 0, h->stream>>>(h->staging, first + off, c, h->state, h->snap,
                                                                                     h->n_pad, h->first_bad);
        HIPCHK(hipGetLastError());
        int fb = 0;
        HIPCHK(hipMemcpyAsync(&fb, h->first_bad, sizeof fb, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream)); /* staging is reused by the next chunk */
        if (fb != big && bad_env < 0) {
            bad_env = first + off + fb;
            HIPCHK(hipMemcpyAsync(h->first_bad, &big, sizeof big, hipMemcpyHostToDevice, h->stream));
        }
    }
    HIPCHK(hipMemsetAsync(h->episode + first, 0, (size_t)count * 4, h->stream)); /* an uploaded state is episode 0 of its env */
    if (h->terminal) HIPCHK(hipMemsetAsync(h->terminal + first * POM_REC_DWORDS, 0, (size_t)count * POM_REC_DWORDS * 4, h->stream));
    if (h->agent_mem) { /* uploaded envs start new games: fresh agents */
        HIPCHK(hipMemsetAsync(h->agent_mem + first * 4, 0, (size_t)count * 16, h->stream));
        HIPCHK(hipMemsetAsync(h->agent_mem + 4 * h->n_pad + first * 4, 0, (size_t)count * 16, h->stream));
    }
    if (bad_env >= 0) {
        snprintf(g_err, sizeof g_err, "pom_batch_upload: env %lld holds a value outside the representable game states "
                 "(it was replaced by a finished blank board)", (long long)bad_env);
        return POM_E_UNREPRESENTABLE;
    }
    return POM_OK;
}

int pom_batch_download(PomBatch* h, void* states, int64_t first, int64_t count)
{
    int rc = check_range(h, first, count);
    if (rc || !states) return rc ? rc : POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    for (int64_t off = 0; off < count; off += h->staging_envs) {
        const int64_t c = count - off < h->staging_envs ? count - off : h->staging_envs;
        HIPCHK(hipMemsetAsync(h->staging, 0, (size_t)c * POM_STATE_BYTES, h->stream));
        pom_unpack_kernel<<<dim3((unsigned)((c + 63) / 64)), dim3(64), 0, h->stream>>>(h->state, first + off, c, h->n_pad, h->staging);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync((char*)states + off * POM_STATE_BYTES, h->staging, (size_t)c * POM_STATE_BYTES, hipMemcpyDeviceToHost,
                              h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return POM_OK;
}

int pom_batch_snapshot(PomBatch* h)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    pom_snapshot_kernel<<<dim3((unsigned)((h->n_pad + 255) / 256)), dim3(256), 0, h->stream>>>(h->state, h->snap, h->n_pad);
    HIPCHK(hipGetLastError());
    return POM_OK;
}


int pom_batch_step_device(PomBatch* h, const int32_t* moves_dev)
{
    if (!h || !moves_dev) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    /* the moves were produced on the caller's stream and may be overwritten there right after this call */
    return launch_step(h, moves_dev, 0, 0, 1, false, true);
}

int pom_batch_step_device_many(PomBatch* h, const int32_t* moves_dev, int32_t ticks)
{
    if (!h || !moves_dev || ticks < 0) return POM_E_ARG;
    if (ticks == 0) return POM_OK;
    HIPCHK(hipSetDevice(h->device));
    /* chained where the handle chains: one launch over all tiles per tick, the launches on different streams, a tile's visitor at
     * distance d from the call's first visit reads tick d of the tape (pom_chain.h).  Elsewhere, and for a single tick: plain
     * launches in a row on the caller's stream, as pom_batch_step_device. */
    if (ticks >= 2 && runs_chain(h, false, 1)) {
        StepParams p;
        memset(&p, 0, sizeof p);
        if (int rc = fill_params(h, p, moves_dev, 0, 0, 1)) return rc;
        bool used = false;
        const uint64_t tick_before = h->tick;
        const int rc = launch_many_chain(h, p, ticks, false, &used);
        h->tick = tick_before; /* explicit moves do not advance the tick that keys the synthetic stream (as pom_batch_step_device) */
        if (rc || used) return rc;
    }
    for (int32_t t = 0; t < ticks; t++)
        if (int rc = launch_step(h, moves_dev + (int64_t)t * h->n * 4, 0, 0, 1, false, true)) return rc;
    return POM_OK;
}

int pom_batch_chain_stats(PomBatch* h, int64_t out[4])
{
    if (!h || !out) return POM_E_ARG;
    out[0] = h->chain.stat_launches;
    out[1] = h->chain.stat_settles;
    out[2] = h->chain.stat_tiles_recovered;
    out[3] = h->chain.stat_ticks_replayed;
    return POM_OK;
}

/* self-test of the chained launches' hand-off, without the game (pom_kernels.h pom_chain_litmus_kernel): `launches` launches over
 * `tiles` records round-robin over `streams` streams; out[0] records found stale or torn, out[1] dwords that differed, out[2]
 * visits played, out[3] visits expected, out[4] tiles whose final record or word is not what `launches` clean visits leave,
 * out[5] the kernels' failure flags (POM_CHAIN_E_*; a give-up leaves its tile behind: counted in out[4]) */
int pom_chain_litmus(int32_t device, int64_t tiles, int32_t launches, int32_t streams, int64_t out[6])
{
    if (!out || tiles < 8 || tiles > (1 << 22) || launches < 1 || launches > (1 << 20) || streams < 1 || streams > PomBatch::MAX_PARTS) return POM_E_ARG;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return POM_E_HIP;
    HIPCHK(hipSetDevice(device));
    PomChain c;
    hipStream_t st[PomBatch::MAX_PARTS] = {};
    uint32_t* data = nullptr;
    unsigned long long* res = nullptr;
    int rc = POM_OK;
    auto cleanup = [&] {
        for (int k = 0; k < streams; k++)
            if (st[k]) {
                (void)hipStreamSynchronize(st[k]);
                (void)hipStreamDestroy(st[k]);
            }
        chain_destroy(&c);
        (void)hipFree(data);
        (void)hipFree(res);
    };
    for (int k = 0; k < streams && rc == POM_OK; k++)
        if (hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) != hipSuccess) rc = POM_E_HIP;
    if (rc == POM_OK && !chain_setup(&c, tiles, st[0])) {
        snprintf(g_err, sizeof g_err, "pom_chain_litmus: chained launches are not available on this device");
        rc = POM_E_HIP;
    }
    const size_t bytes = (size_t)tiles * POM_TILE_DWORDS * 4;
    if (rc == POM_OK && (hipMalloc((void**)&data, bytes) != hipSuccess || hipMalloc((void**)&res, 64) != hipSuccess)) rc = POM_E_NOMEM;
    if (rc != POM_OK) {
        cleanup();
        return rc;
    }
    try { /* (host vectors below: nothing may throw across the C boundary) */
    /* visit 0 expects 0 ^ tag: fill the records accordingly (on the host: this is a test) */
    {
        std::vector<uint32_t> init((size_t)tiles * POM_TILE_DWORDS);
        for (int64_t t = 0; t < tiles; t++)
            for (int k = 0; k < POM_TILE_DWORDS; k++) init[(size_t)t * POM_TILE_DWORDS + k] = (uint32_t)t * 2654435761u + (uint32_t)k;
        if (hipMemcpy(data, init.data(), bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(res, 0, 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            cleanup();
            return POM_E_HIP;
        }
    }
    LitmusParams p;
    p.data = data;
    p.tile_seq = c.tile_seq;
    p.err = c.aux;
    p.out = res;
    p.tiles = tiles;
    p.chain_seq0 = 0;
    p.wait_limit = c.wait_limit;
    const dim3 grid((unsigned)((tiles + 7) / 8 * 8));
    for (int k = 0; k < launches && rc == POM_OK; k++) {
        pom_chain_litmus_kernel<<<grid, dim3(64), 0, st[k % streams]>>>(p);
        if (hipGetLastError() != hipSuccess) rc = POM_E_HIP;
    }
    for (int k = 0; k < streams; k++)
        if (hipStreamSynchronize(st[k]) != hipSuccess) rc = POM_E_HIP;
    if (rc == POM_OK) {
        unsigned long long r3[3] = {0, 0, 0};
        uint32_t flags = 0;
        std::vector<uint32_t> fin((size_t)tiles * POM_TILE_DWORDS);
        std::vector<unsigned long long> words((size_t)tiles * POM_CHAIN_WORD_STRIDE);
        if (hipMemcpy(r3, res, 24, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&flags, c.aux, 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(fin.data(), data, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(words.data(), c.tile_seq, words.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = POM_E_HIP;
        } else {
            int64_t wrong = 0;
            for (int64_t t = 0; t < tiles; t++) {
                const unsigned long long w = words[(size_t)t * POM_CHAIN_WORD_STRIDE];
                bool ok = ((uint32_t)w & POM_CHAIN_COUNT_MASK) == (uint32_t)launches && (uint32_t)(w >> POM_CHAIN_TICKET_SHIFT) == (uint32_t)launches &&
                          !((uint32_t)w & POM_CHAIN_POISON);
                for (int k = 0; k < POM_TILE_DWORDS && ok; k++)
                    ok = fin[(size_t)t * POM_TILE_DWORDS + k] == ((uint32_t)launches ^ ((uint32_t)t * 2654435761u + (uint32_t)k));
                wrong += !ok;
            }
            out[0] = (int64_t)r3[0];
            out[1] = (int64_t)r3[1];
            out[2] = (int64_t)r3[2];
            out[3] = tiles * (int64_t)launches;
            out[4] = wrong;
            out[5] = (int64_t)flags;
        }
    }
    } catch (...) {
        rc = POM_E_NOMEM;
    }
    cleanup();
    return rc;
}

int pom_batch_step(PomBatch* h, const int32_t* moves_host)
{
    if (!h || !moves_host) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    int rc = quiesce(h); /* the previous step's parts still read moves_dev */
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h->moves_dev, moves_host, (size_t)h->n * 16, hipMemcpyHostToDevice, h->stream));
    return launch_step(h, h->moves_dev, 0, 0, 1, false, true);
}

int pom_batch_step_random(PomBatch* h, uint64_t seed, int32_t dist, int32_t ticks, int32_t ticks_per_launch)
{
    if (!h || ticks < 0 || ticks_per_launch < 1 || dist < POM_DIST_HARMLESS || dist > POM_DIST_STRESS) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    /* ticks_per_launch is a request: results never depend on it, and the modes whose several-tick kernels would spill run one
     * tick per launch (max_ticks_per_launch) */
    const int32_t cap = max_ticks_per_launch(h, false);
    const int32_t tpl = ticks_per_launch < cap ? ticks_per_launch : cap;
    const int32_t whole = ticks / tpl;
    if (int rc = launch_many(h, seed, dist, whole, tpl, false)) return rc;
    const int32_t rest = ticks - whole * tpl;
    if (rest > 0) {
        if (int rc = launch_step(h, nullptr, seed, dist, rest)) return rc;
        h->tick += (uint64_t)rest;
    }
    return POM_OK;
}

int pom_batch_set_tick(PomBatch* h, int64_t tick)
{
    if (!h || tick < 0) return POM_E_ARG;
    h->tick = (uint64_t)tick;
    return POM_OK;
}

int pom_batch_status(PomBatch* h, int64_t first, int64_t count, int32_t* done, int32_t* winner, int32_t* draw, int32_t* alive,
                     int32_t* time_step, uint32_t* ubflags)
{
    int rc = check_range(h, first, count);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    void* outs[6] = {done, winner, draw, alive, time_step, ubflags};
    /* the AoS staging buffer doubles as scratch: 6 ints per env << 251 */
    for (int64_t off = 0; off < count; off += h->staging_envs) {
        const int64_t c = count - off < h->staging_envs ? count - off : h->staging_envs;
        pom_status_kernel<<<dim3((unsigned)((c + 255) / 256)), dim3(256), 0, h->stream>>>(h->state, first + off, c, h->n_pad, h->staging);
        HIPCHK(hipGetLastError());
        for (int k = 0; k < 6; k++)
            if (outs[k])
                HIPCHK(hipMemcpyAsync((int32_t*)outs[k] + off, h->staging + (int64_t)k * c, (size_t)c * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return POM_OK;
}

int pom_batch_last_results(PomBatch* h, int64_t first, int64_t count, int32_t* finished, int32_t* winner, int32_t* draw,
                           int32_t* length, int32_t* alive)
{
    int rc = check_range(h, first, count);
    if (rc) return rc;
    if (!h->terminal) {
        snprintf(g_err, sizeof g_err, "pom_batch_last_results: the batch was not created with auto_reset = POM_RESET_AT_END");
        return POM_E_ARG;
    }
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    void* outs[5] = {finished, winner, draw, length, alive};
    for (int64_t off = 0; off < count; off += h->staging_envs) {
        const int64_t c = count - off < h->staging_envs ? count - off : h->staging_envs;
        pom_results_kernel<<<dim3((unsigned)((c + 255) / 256)), dim3(256), 0, h->stream>>>(h->state, h->terminal, first + off, c, h->n_pad,
                                                                                            h->staging);
        HIPCHK(hipGetLastError());
        for (int k = 0; k < 5; k++)
            if (outs[k])
                HIPCHK(hipMemcpyAsync((int32_t*)outs[k] + off, h->staging + (int64_t)k * c, (size_t)c * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return POM_OK;
}

int pom_batch_download_terminal(PomBatch* h, void* states, int64_t first, int64_t count)
{
    int rc = check_range(h, first, count);
    if (rc || !states) return rc ? rc : POM_E_ARG;
    if (!h->terminal) {
        snprintf(g_err, sizeof g_err, "pom_batch_download_terminal: the batch was not created with auto_reset = POM_RESET_AT_END");
        return POM_E_ARG;
    }
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    for (int64_t off = 0; off < count; off += h->staging_envs) {
        const int64_t c = count - off < h->staging_envs ? count - off : h->staging_envs;
        HIPCHK(hipMemsetAsync(h->staging, 0, (size_t)c * POM_STATE_BYTES, h->stream));
        pom_unpack_aos_kernel<<<dim3((unsigned)((c + 63) / 64)), dim3(64), 0, h->stream>>>(h->terminal, first + off, c, h->staging);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync((char*)states + off * POM_STATE_BYTES, h->staging, (size_t)c * POM_STATE_BYTES, hipMemcpyDeviceToHost,
                              h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return POM_OK;
}

int pom_batch_counters_device(PomBatch* h, void* dev_int64x4)
{
    if (!h || !dev_int64x4) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = join_parts(h)) return jr;
    pom_reduce_counters_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(h->wave_counters, h->n_waves, (int64_t*)dev_int64x4);
    HIPCHK(hipGetLastError());
    return POM_OK;
}

int pom_batch_counters(PomBatch* h, int64_t out[POM_CNT_N])
{
    if (!h || !out) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int qr = quiesce(h)) return qr; /* (a tile left behind by chained launches is caught up first: its steps count) */
    int rc = pom_batch_counters_device(h, h->totals_dev);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, h->totals_dev, POM_CNT_N * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return POM_OK;
}

int pom_batch_reset_counters(PomBatch* h)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    HIPCHK(hipMemsetAsync(h->wave_counters, 0, (size_t)h->n_waves * POM_CNT_N * 8, h->stream));
    return POM_OK;
}

int pom_batch_sync(PomBatch* h)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    HIPCHK(hipStreamSynchronize(h->stream));
    return POM_OK;
}

/* what both observation entry points ask of their arguments */
static int observe_args(const char* who, const void* planes_dev, int32_t dtype, int32_t per_agent, const int32_t* agent_attrs_dev,
                        const int32_t* env_attrs_dev)
{
    if (!planes_dev || dtype < POM_OBS_U8 || dtype > POM_OBS_CODES) return POM_E_ARG;
    if (dtype == POM_OBS_CODES && per_agent) {
        snprintf(g_err, sizeof g_err, "%s: POM_OBS_CODES names the agents by id (10..13), there is no per-agent view of it", who);
        return POM_E_ARG;
    }
    const uintptr_t esz = dtype == POM_OBS_F16 ? 2 : dtype == POM_OBS_F32 ? 4 : 1;
    if (((uintptr_t)planes_dev & (4 * esz - 1)) || ((uintptr_t)agent_attrs_dev & 15) || ((uintptr_t)env_attrs_dev & 15) ||
        (dtype == POM_OBS_U8 && !per_agent && ((uintptr_t)planes_dev & 15)))
    {
        snprintf(g_err, sizeof g_err, "%s: output pointers must be 16-byte aligned", who);
        return POM_E_ARG;
    }
    return POM_OK;
}

int pom_batch_observe(PomBatch* h, void* planes_dev, int32_t dtype, int32_t per_agent, int32_t* agent_attrs_dev,
                      int32_t* env_attrs_dev)
{
    if (!h) return POM_E_ARG;
    if (int ar = observe_args("pom_batch_observe", planes_dev, dtype, per_agent, agent_attrs_dev, env_attrs_dev)) return ar;
    HIPCHK(hipSetDevice(h->device));
    /* quiesce, not only join: after chained launches a tile some visitor could not play is caught up first (chain_settle; free
     * while no chained launch has been issued since the last check) — the planes always describe the state a download returns */
    if (int jr = quiesce(h)) return jr;
    ObserveParams p;
    p.state = h->state;
    p.n = h->n;
    p.n_pad = h->n_pad;
    p.block0 = 0;
    p.planes = planes_dev;
    p.agent_attrs = agent_attrs_dev;
    p.env_attrs = env_attrs_dev;
    p.dtype = dtype;
    p.per_agent = per_agent ? 1 : 0;
    pom_observe_kernel<<<dim3((unsigned)((h->n + 15) / 16)), dim3(64), 0, h->stream>>>(p);
    HIPCHK(hipGetLastError());
    return POM_OK;
}

int pom_batch_step_device_observe(PomBatch* h, const int32_t* moves_dev, void* planes_dev, int32_t dtype, int32_t per_agent,
                                  int32_t* agent_attrs_dev, int32_t* env_attrs_dev)
{
    if (!h || !moves_dev) return POM_E_ARG;
    if (int ar = observe_args("pom_batch_step_device_observe", planes_dev, dtype, per_agent, agent_attrs_dev, env_attrs_dev)) return ar;
    HIPCHK(hipSetDevice(h->device));
    if (!h->quad) { /* the one-lane-per-env shapes have no fused twin: the two launches */
        if (int rc = pom_batch_step_device(h, moves_dev)) return rc;
        return pom_batch_observe(h, planes_dev, dtype, per_agent, agent_attrs_dev, env_attrs_dev);
    }
    const PomObserveOut obs = {planes_dev, agent_attrs_dev, env_attrs_dev, dtype, per_agent ? 1 : 0};
    return launch_step(h, moves_dev, 0, 0, 1, false, true, &obs);
}

/* closed-loop stepping: one tick for a range of whole tiles, one launch on the caller's stream, nothing forked or joined */
int pom_batch_step_device_range(PomBatch* h, int64_t first, int64_t count, const int32_t* moves_dev, void* stream, void* planes_dev,
                                int32_t dtype, int32_t per_agent, int32_t* agent_attrs_dev, int32_t* env_attrs_dev)
{
    int rc = check_range(h, first, count);
    if (rc || !moves_dev) return rc ? rc : POM_E_ARG;
    if (!h->quad || (first & 15) || ((count & 15) && first + count != h->n)) {
        snprintf(g_err, sizeof g_err, "pom_batch_step_device_range: whole tiles of 16 envs (first %lld, count %lld), quad launch shape", (long long)first,
                 (long long)count);
        return POM_E_ARG;
    }
    if (planes_dev)
        if (int ar = observe_args("pom_batch_step_device_range", planes_dev, dtype, per_agent, agent_attrs_dev, env_attrs_dev)) return ar;
    if (count == 0) return POM_OK;
    HIPCHK(hipSetDevice(h->device));
    if (int qr = quiesce(h)) return qr; /* (free, and no runtime call at all, once the handle is settled: what a graph capture needs) */
    StepParams p;
    if (int fr = fill_params(h, p, moves_dev, 0, 0, 1)) return fr;
    if (planes_dev) {
        p.obs_planes = planes_dev;
        p.obs_agent_attrs = agent_attrs_dev;
        p.obs_env_attrs = env_attrs_dev;
        p.obs_dtype = dtype;
        p.obs_per_agent = per_agent ? 1 : 0;
    }
    p.block0 = first / 16;
    p.block_end = (first + count + 15) / 16;
    const void* kernel = planes_dev ? step_observe_kernel_for(h) : step_kernel_for(h, false, true);
    void* args[1] = {&p};
    HIPCHK(hipLaunchKernel(kernel, dim3((unsigned)((p.block_end - p.block0 + POM_WPB - 1) / POM_WPB)), dim3(64 * POM_WPB), args, 0,
                           stream ? (hipStream_t)stream : h->stream));
    return POM_OK;
}

/* the stand-in policy of the closed-loop measurements (pom_batch.h): a workgroup takes 64 envs, reads their observations — 64 x 605
 * bytes, as dwords, coalesced; lane l sums w[k] * (2k + 1) over the group's dwords k = l, l + 256, ... — and every lane (env, agent) draws its move from its sum */
__global__ __launch_bounds__(256) void pom_bench_policy_kernel(const uint8_t* codes, int32_t* moves, int64_t first, int64_t count, uint32_t tick)
{
    const int64_t e0 = first + (int64_t)blockIdx.x * 64;
    const int64_t e = e0 + (threadIdx.x >> 2);
    uint32_t acc = 0;
    if (codes) {
        const int64_t envs = count - (int64_t)blockIdx.x * 64 < 64 ? count - (int64_t)blockIdx.x * 64 : 64;
        const uintptr_t lo = (uintptr_t)(codes + e0 * POM_OBS_CODE_PLANES * POM_CELLS), hi = lo + (uintptr_t)envs * POM_OBS_CODE_PLANES * POM_CELLS;
        const uint32_t* w = reinterpret_cast<const uint32_t*>((lo + 3) & ~(uintptr_t)3); /* whole dwords inside the range's bytes */
        const int64_t nw = (int64_t)((hi & ~(uintptr_t)3) - ((lo + 3) & ~(uintptr_t)3)) / 4;
        /* (a sum whose order does not matter: four loads in flight per lane) */
        uint32_t a1 = 0, a2 = 0, a3 = 0;
        int64_t k = threadIdx.x;
        for (; k + 768 < nw; k += 1024) {
            acc += w[k] * (uint32_t)(2 * k + 1);
            a1 += w[k + 256] * (uint32_t)(2 * (k + 256) + 1);
            a2 += w[k + 512] * (uint32_t)(2 * (k + 512) + 1);
            a3 += w[k + 768] * (uint32_t)(2 * (k + 768) + 1);
        }
        for (; k < nw; k += 256) acc += w[k] * (uint32_t)(2 * k + 1);
        acc += a1 + a2 + a3;
    }
    if (e < first + count) {
        uint32_t x = acc ^ ((uint32_t)e * 0x9E3779B1u) ^ ((threadIdx.x & 3u) * 0x85EBCA6Bu) ^ (tick * 0xC2B2AE35u);
        x ^= x >> 15;
        x *= 0x2C1B3C6Du;
        x ^= x >> 12;
        moves[e * 4 + (threadIdx.x & 3)] = (int32_t)((x >> 8) % 6u);
    }
}
int pom_bench_policy(const uint8_t* codes_dev, int32_t* moves_dev, int64_t first, int64_t count, uint32_t tick, void* stream)
{
    if (!moves_dev || first < 0 || count < 0) return POM_E_ARG;
    if (count == 0) return POM_OK;
    pom_bench_policy_kernel<<<dim3((unsigned)((count + 63) / 64)), dim3(256), 0, (hipStream_t)stream>>>(codes_dev, moves_dev, first, count, tick);
    HIPCHK(hipGetLastError());
    return POM_OK;
}

int pom_batch_generate(PomBatch* h, uint64_t board_seed)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    h->board_seed = board_seed;
    pom_generate_kernel<<<dim3((unsigned)((h->n + 15) / 16)), dim3(64), 0, h->stream>>>(h->state, h->snap, h->episode, h->n, h->n_pad,
                                                                                         h->env_offset, board_seed);
    HIPCHK(hipGetLastError());
    if (h->agent_mem) HIPCHK(hipMemsetAsync(h->agent_mem, 0, (size_t)h->n_pad * 32, h->stream)); /* new games: fresh agents */
    if (h->terminal) HIPCHK(hipMemsetAsync(h->terminal, 0, (size_t)h->n_pad * POM_REC_DWORDS * 4, h->stream));
    return POM_OK;
}

int pom_batch_episodes(PomBatch* h, int64_t first, int64_t count, uint32_t* out)
{
    int rc = check_range(h, first, count);
    if (rc || !out) return rc ? rc : POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    HIPCHK(hipMemcpyAsync(out, h->episode + first, (size_t)count * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return POM_OK;
}

int pom_batch_moves_device(PomBatch* h, int32_t** moves_dev)
{
    if (!h || !moves_dev) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = join_parts(h)) return jr; /* what the caller queues on the handle's stream next sees the policy's moves */
    *moves_dev = h->moves_dev;
    return POM_OK;
}

int pom_batch_stream(PomBatch* h, void** stream)
{
    if (!h || !stream) return POM_E_ARG;
    *stream = (void*)h->stream;
    return POM_OK;
}

int pom_batch_device_view(PomBatch* h, void** base, int64_t* n_pad, int32_t* rec_dwords)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int qr = quiesce(h)) return qr;
    if (base) *base = h->state;
    if (n_pad) *n_pad = h->n_pad;
    if (rec_dwords) *rec_dwords = POM_REC_DWORDS;
    return POM_OK;
}

#if defined(POM_CHAIN_DIAG)
/* diagnostic build only: per tile 4 sums over the chained launches so far (cycles to the ticket, cycles polling, cycles in all, polls); cleared */
int pom_chain_diag_read(PomBatch* h, unsigned long long* out, int64_t tiles)
{
    if (!h || !h->chain.tile_seq || tiles != h->n_pad / h->epw) return POM_E_ARG;
    if (int jr = join_parts(h)) return jr;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->chain.tile_seq + tiles * POM_CHAIN_WORD_STRIDE, (size_t)tiles * 544, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->chain.tile_seq + tiles * POM_CHAIN_WORD_STRIDE, 0, (size_t)tiles * 544));
    HIPCHK(hipDeviceSynchronize());
    return POM_OK;
}
#endif
#if defined(POM_DIAG)
/* diagnostic build only: the step kernel with zero ticks = HBM -> LDS -> HBM round trip of every record */
int pom_diag_copy_only(PomBatch* h)
{
    return launch_step(h, nullptr, 0, 0, 0);
}
/* diagnostic build only: read and clear the per-phase cycle sums (summed over wavefronts) */
int pom_diag_read(PomBatch* h, long long out[POM_PH_N])
{
    if (!h || !h->diag) return POM_E_ARG;
    long long* tmp = new long long[(size_t)h->n_waves * POM_PH_N];
    HIPCHK(hipMemcpyAsync(tmp, h->diag, (size_t)h->n_waves * POM_PH_N * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemsetAsync(h->diag, 0, (size_t)h->n_waves * POM_PH_N * 8, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int k = 0; k < POM_PH_N; k++) out[k] = 0;
    for (int64_t w = 0; w < h->n_waves; w++)
        for (int k = 0; k < POM_PH_N; k++) out[k] += tmp[w * POM_PH_N + k];
    delete[] tmp;
    return POM_OK;
}
#endif

#if defined(POM_DIAG)
/* diagnostic build only: the per-wavefront accumulators as they are (n_waves x POM_PH_N), then cleared */
extern "C" int pom_diag_read_raw(PomBatch* h, long long* out, long long max_waves)
{
    if (!h || !h->diag || max_waves < h->n_waves) return POM_E_ARG;
    if (int jr = join_parts(h)) return jr;
    HIPCHK(hipMemcpyAsync(out, h->diag, (size_t)h->n_waves * POM_PH_N * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemsetAsync(h->diag, 0, (size_t)h->n_waves * POM_PH_N * 8, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return POM_OK;
}
#endif

#if defined(POM_DIAG)
extern "C" int pom_diag_policy_read(PomBatch* h, long long out[POM_PP_N])
{
    if (!h || !h->diag_pol) return POM_E_ARG;
    const size_t nw = (size_t)(h->n_pad / 16);
    if (int jr = join_parts(h)) return jr;
    long long* tmp = new long long[nw * POM_PP_N];
    HIPCHK(hipMemcpyAsync(tmp, h->diag_pol, nw * POM_PP_N * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemsetAsync(h->diag_pol, 0, nw * POM_PP_N * 8, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int k = 0; k < POM_PP_N; k++) out[k] = 0;
    for (size_t w = 0; w < nw; w++)
        for (int k = 0; k < POM_PP_N; k++) out[k] += tmp[w * POM_PP_N + k];
    delete[] tmp;
    return POM_OK;
}
#endif

int pom_batch_set_streams(PomBatch* h, int32_t streams)
{
    if (!h || streams < 1 || streams > PomBatch::MAX_PARTS) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    const int64_t tiles = h->n_pad / h->epw;
    const int want = (int64_t)streams > tiles ? (int)tiles : streams;
    if (int er = ensure_sub_streams(h, streams)) return er;
    h->parts = want;
    h->chain_parts = streams; /* chained launches cover all tiles: their stream count is not bounded by the tiles */
    h->chain_auto = false;
    return POM_OK;
}


int pom_batch_policy_simple(PomBatch* h, uint64_t seed, int32_t* moves_out_host)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    int rc = launch_policy(h, seed);
    if (rc) return rc;
    if (moves_out_host) {
        if (int jr = quiesce(h)) return jr;
        HIPCHK(hipMemcpyAsync(moves_out_host, h->moves_dev, (size_t)h->n * 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return POM_OK;
}

int pom_batch_step_policy(PomBatch* h)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    int rc = launch_step(h, h->moves_dev, 0, 0, 1);
    if (rc) return rc;
    h->tick += 1;
    return POM_OK;
}

int pom_batch_step_simple(PomBatch* h, uint64_t seed, int32_t ticks)
{
    if (!h || ticks < 0) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (h->quad && h->fuse_policy) { /* the fused kernel: policy and tick on one load of the record */
        if (int rc = ensure_agent_mem(h)) return rc;
        return launch_many(h, seed, 0, ticks, 1, true);
    }
    for (int32_t t = 0; t < ticks; t++) {
        int rc = launch_policy(h, seed);
        if (!rc) rc = launch_step(h, h->moves_dev, 0, 0, 1);
        if (rc) return rc;
        h->tick += 1;
    }
    return POM_OK;
}

int pom_batch_policy_memory(PomBatch* h, int64_t first, int64_t count, int32_t* out16)
{
    int rc = check_range(h, first, count);
    if (rc || !out16) return rc ? rc : POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    if (!h->agent_mem) {
        memset(out16, 0, (size_t)count * 4 * 16 * sizeof(int32_t));
        return POM_OK;
    }
    uint32_t* tmp = new (std::nothrow) uint32_t[(size_t)count * 8];
    if (!tmp) return POM_E_NOMEM;
    hipError_t e1 = hipMemcpyAsync(tmp, h->agent_mem + first * 4, (size_t)count * 16, hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipMemcpyAsync(tmp + count * 4, h->agent_mem + 4 * h->n_pad + first * 4, (size_t)count * 16, hipMemcpyDeviceToHost, h->stream);
    hipError_t e3 = hipStreamSynchronize(h->stream);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        delete[] tmp;
        set_err("policy memory download", e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3);
        return POM_E_HIP;
    }
    for (int64_t k = 0; k < count * 4; k++) pom_policy_mem_unpack(tmp[k], tmp[count * 4 + k], out16 + 16 * k);
    delete[] tmp;
    return POM_OK;
}

int pom_batch_fork(PomBatch* h)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    for (int k = 0; k < PomBatch::MAX_PARTS; k++) /* the issuing threads of multi-tick calls: work is coming, stay awake for it */
        if (h->issuers[k]) {
            h->issuers[k]->posted.fetch_add(1, std::memory_order_release);
            h->issuers[k]->cv.notify_all();
        }
    return fork_parts(h, 0);
}

int pom_batch_flush(PomBatch* h)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    return join_parts(h);
}

int pom_batch_profile(PomBatch* h, int enable)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (enable && !h->prof_ev[0])
        for (int k = 0; k < 2 * PomBatch::PROF_RING; k++) HIPCHK(hipEventCreate(&h->prof_ev[k]));
    h->profiling = enable != 0;
    h->prof_n = 0;
    return POM_OK;
}

int pom_batch_profile_read(PomBatch* h, double* mean_ms, int64_t* launches)
{
    if (!h) return POM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (int jr = quiesce(h)) return jr;
    HIPCHK(hipStreamSynchronize(h->stream));
    double sum = 0;
    for (int k = 0; k < h->prof_n; k++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h->prof_ev[2 * k], h->prof_ev[2 * k + 1]));
        sum += ms;
    }
    if (mean_ms) *mean_ms = h->prof_n ? sum / h->prof_n : 0.0;
    if (launches) *launches = h->prof_n;
    h->prof_n = 0;
    return POM_OK;
}

int pom_batch_launch_shape(PomBatch* h, int32_t* envs_per_wave, int32_t* lanes_per_env, int32_t* launches_per_step)
{
    if (!h) return POM_E_ARG;
    if (envs_per_wave) *envs_per_wave = h->epw;
    if (lanes_per_env) *lanes_per_env = h->quad ? 4 : 1;
    if (launches_per_step) *launches_per_step = runs_chain(h, false, 1) ? 1 : h->parts; /* chained: one launch over all tiles per tick */
    return POM_OK;
}

int pom_batch_issue_info(PomBatch* h, int32_t* issue_mode, int32_t* streams)
{
    if (!h) return POM_E_ARG;
    const bool chains = runs_chain(h, false, 1);
    if (issue_mode) *issue_mode = h->issue_mode == POM_ISSUE_CHAIN && !chains ? POM_ISSUE_THREADS : h->issue_mode;
    if (streams) *streams = !chains ? h->parts : h->chain_auto ? h->chain_last_use : h->chain_parts;
    return POM_OK;
}

/* pom_step / pom_env_step: a pinned, device-mapped page the kernel reads the State from and writes it back to
 * (pom_step_one_kernel), the host polling the kernel's last store.  The reference's Step is re-entrant over distinct States and
 * its performance test steps one env per std::thread (unit_test/bboard/performance_test.cpp:40-50,71-94): every calling thread
 * gets a slot of its own — a page and a sequence number — and posts its request there.  Launches are COMBINED: whichever
 * thread gets hold of the launch lock launches one kernel for every request pending at that moment (a workgroup per request),
 * the others find their page answered without having launched anything — the HIP runtime issues launches of one process one
 * after the other (~5 us each), so a launch per call would cap eight threads at three times one thread's rate.  Device 0.
 * POM_ONE_SLOTS slots; threads beyond that share (slot = thread number mod slots), which is what the per-slot mutex is for. */
struct PomOneShared {
    std::mutex init_mu, launch_mu;
    std::mutex slot_mu[POM_ONE_SLOTS];
    uint32_t seq[POM_ONE_SLOTS] = {};
    int32_t* io = nullptr;     /* host address of the pages */
    int32_t* io_dev = nullptr; /* the same pages as the device sees them */
    enum { STREAMS = 4 };
    hipStream_t stream[STREAMS] = {};
    unsigned turn = 0;
    std::atomic<uint64_t> pending{0};
    std::atomic<uint64_t> failed{0}; /* requests whose launch failed (answered by the launching thread) */
    std::atomic<bool> ready{false};
};
static PomOneShared g_one;
static std::atomic<unsigned> g_one_threads{0};

static int one_init(PomOneShared& o)
{
    std::lock_guard<std::mutex> g(o.init_mu);
    if (o.ready.load()) return POM_OK;
    HIPCHK(hipSetDevice(0));
    const size_t bytes = (size_t)POM_ONE_SLOTS * POM_ONE_PAGE_DWORDS * 4;
    if (!o.io) HIPCHK(hipHostMalloc((void**)&o.io, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&o.io_dev, o.io, 0));
    for (int k = 0; k < PomOneShared::STREAMS; k++)
        if (!o.stream[k]) HIPCHK(hipStreamCreateWithFlags(&o.stream[k], hipStreamNonBlocking));
    memset(o.io, 0, bytes);
    o.ready.store(true);
    return POM_OK;
}

/* launch_mu held: one kernel for every request posted so far */
static void one_launch_pending(PomOneShared& o)
{
    const uint64_t mask = o.pending.exchange(0);
    if (!mask) return;
    StepOneParams p;
    p.io_base = o.io_dev;
    p.slots = mask;
    (void)hipSetDevice(0);
    pom_step_one_kernel<<<dim3((unsigned)__builtin_popcountll(mask)), dim3(64), 0, o.stream[o.turn++ % PomOneShared::STREAMS]>>>(p);
    if (hipGetLastError() != hipSuccess) o.failed.fetch_or(mask); /* their owners report it */
}

static int step_one(void* state_1004, const int32_t moves[4], int32_t mode, int32_t max_steps, int32_t status4[4])
{
    if (!state_1004 || !moves) return POM_E_ARG;
    PomOneShared& o = g_one;
    if (!o.ready.load())
        if (int rc = one_init(o)) return rc;
    static thread_local int my_slot = -1;
    if (my_slot < 0) my_slot = (int)(g_one_threads.fetch_add(1) % POM_ONE_SLOTS);
    const uint64_t bit = 1ull << my_slot;
    std::lock_guard<std::mutex> lock(o.slot_mu[my_slot]);
    int32_t* io = o.io + (size_t)my_slot * POM_ONE_PAGE_DWORDS;
    const uint32_t seq = ++o.seq[my_slot] ? o.seq[my_slot] : ++o.seq[my_slot]; /* never 0: the page starts zeroed */
    memcpy(io, state_1004, POM_STATE_BYTES);
    memcpy(io + POM_ONE_MOVES, moves, 16);
    io[POM_ONE_MODE] = mode;
    io[POM_ONE_MAX_STEPS] = max_steps;
    io[POM_ONE_REQ] = (int32_t)seq;
    o.pending.fetch_or(bit, std::memory_order_release); /* posted: the page is complete */
    /* until the page is answered: launch what is pending whenever the launch lock is free (my own request, unless somebody
     * else's launch has taken it along), and watch the sequence word — the kernel's last store */
    volatile uint32_t* seq_word = reinterpret_cast<volatile uint32_t*>(io) + POM_ONE_SEQ;
    const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(5);
    bool synced = false;
    for (;;) {
        if (*seq_word == seq) break;
        if (o.launch_mu.try_lock()) {
            one_launch_pending(o);
            o.launch_mu.unlock();
        }
        if (o.failed.load() & bit) {
            o.failed.fetch_and(~bit);
            snprintf(g_err, sizeof g_err, "pom_step: the kernel launch failed");
            return POM_E_HIP;
        }
        for (int spin = 0; spin < 64 && *seq_word != seq; spin++) __builtin_ia32_pause();
        if (*seq_word != seq && std::chrono::steady_clock::now() > until) {
            if (synced) {
                snprintf(g_err, sizeof g_err, "pom_step: the kernel finished without reporting");
                return POM_E_HIP;
            }
            /* not within milliseconds: wait the ordinary way — which is also where a device error surfaces */
            std::lock_guard<std::mutex> g(o.launch_mu);
            one_launch_pending(o);
            for (int k = 0; k < PomOneShared::STREAMS; k++) HIPCHK(hipStreamSynchronize(o.stream[k]));
            synced = true;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (io[POM_ONE_BAD]) {
        snprintf(g_err, sizeof g_err, "pom_step: the State holds a value outside the representable game states (it was left as it is)");
        return POM_E_UNREPRESENTABLE;
    }
    memcpy(state_1004, io + POM_ONE_OUT, POM_STATE_BYTES);
    if (status4) memcpy(status4, io + POM_ONE_STATUS, 16);
    return POM_OK;
}

int pom_step(void* state_1004, const int32_t moves[4]) { return step_one(state_1004, moves, POM_MODE_RAW, 0, nullptr); }

int pom_env_step(void* state_1004, const int32_t moves[4], int32_t max_steps, int32_t* done, int32_t* winner, int32_t* draw,
                 uint32_t* ubflags)
{
    int32_t st[4] = {0, -1, 0, 0};
    const int rc = step_one(state_1004, moves, POM_MODE_ENV, max_steps, st);
    if (rc) return rc;
    if (done) *done = st[0];
    if (winner) *winner = st[1];
    if (draw) *draw = st[2];
    if (ubflags) *ubflags = (uint32_t)st[3];
    return POM_OK;
}

} /* extern "C" */
