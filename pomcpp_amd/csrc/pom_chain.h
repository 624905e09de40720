/*
 * pom_chain.h — chained launches (POM_ISSUE_CHAIN): every launch covers ALL tiles and plays ONE tick, consecutive launches go
 * to DIFFERENT streams, and what orders a tile's ticks is a ticket word per tile (StepParams.tile_seq) instead of the order of
 * the launches (host side; the device side is the CHAIN instantiation of pom_step_kernel, pom_kernels.h).
 *
 * Why: at 65,536 envs a launch is ONE round of wavefronts, so it lasts as long as its slowest wavefront (mean 18.5 k cycles,
 * slowest 30 - 39 k: DESIGN.md §4), and the launches of a stream wait for each other — although a tile's next tick depends on
 * that tile's previous tick only.  With sub-batches on parallel streams (the other issue modes) each sub-batch still waits
 * for ITS slowest wavefront every tick.  Here the j-th wavefront that visits a tile plays the tile's j-th tick, whichever
 * launch it belongs to: a wavefront takes a ticket (one atomic), waits until the visit before it is stored — mostly it already
 * is — and publishes the tile when its own stores have arrived.  A tile's next tick starts as soon as the tile is stored and
 * any launch in flight offers a wavefront for it.
 *
 * What was tried first (scripts/experiments/chain/): packets WITHOUT the AQL barrier bit on one queue of the library's own
 * (HIP always sets the bit on gfx9; hipExtAnyOrderLaunch is ignored).  They do overlap — across XCDs only: each XCD still
 * plays its share of consecutive packets of one queue one after the other (the next packet's first wavefront starts 10 - 15 us
 * after the previous packet's, whatever the occupancy: chain_diag.py), so the XCDs drift apart but no tile ever sees an early
 * successor.  Different queues overlap freely, hence the streams.
 *
 * Launches that are in flight together must be interchangeable (the same instantiation, seed, distribution, mode, offset between
 * ticks and visits): launch_many_chain joins the streams before a call that differs from the launches still in flight.
 *
 * What is relied on, and how each point is checked at run time:
 *  - a tile is always handled by the same XCD (workgroup id -> XCD round-robin, the same grid every launch): its record then
 *    goes from tick to tick through ONE L2 with no cache maintenance; the loads bypass the CU's vector cache (sc1).  The kernel
 *    compares the XCD it runs on with the one recorded in the tile's word and refuses to step otherwise (POM_CHAIN_E_XCD);
 *  - a wavefront only ever waits for the holder of the previous ticket of its tile, which therefore is resident and running:
 *    no order of dispatch can deadlock.  Should a wait still not end within POM_CHAIN_WAIT_LIMIT_US of wall-clock time (the
 *    holder descheduled for seconds: processes time-slicing the GPU, a debugger) the wavefront gives up (POM_CHAIN_E_TIMEOUT):
 *    no launch can hang the device.
 * Either way the wavefront POISONS the tile (a bit in its word) instead of stepping it: every later visitor leaves a poisoned
 * tile alone, so its record stays exactly what its last stored tick made it and its word says how many ticks that was.  The
 * host keeps a log of the chained calls since the last check (what each was launched with); chain_settle (pom_runtime.h) —
 * run by every API call that reads or changes the batch other than another chained call — joins the streams, lets a small
 * kernel list the poisoned tiles, and REPLAYS their missing ticks with ordinary one-tile launches from that log: a give-up
 * costs time, never correctness (tests/test_gpu_chain.py forces it with a wait limit of zero).  Uneven ticket counts
 * (POM_CHAIN_E_UNEVEN: a launch did not visit every tile exactly once — the workgroup -> XCD assignment is not the round-robin
 * assumed) cannot be replayed and fail the call with POM_E_HIP; the handle then launches the ordinary way.  Before a handle's
 * first chained launch (in pom_batch_create) a 64-workgroup probe checks the workgroup -> XCD pattern itself (a partitioned
 * device, or another chip, does not have it): no chained launch is ever issued where it does not hold.
 */
#ifndef POM_CHAIN_H_
#define POM_CHAIN_H_

#include <vector>

/* one chained call (or several that continue the same play back to back): everything needed to play any of its ticks again */
struct PomChainCall {
    uint32_t visit0 = 0, launches = 0; /* the tiles' visits visit0 .. visit0 + launches - 1 */
    StepParams p;                      /* as launched: p.tick0 is the tick of visit p.chain_seq0 */
    bool policy = false;
};

struct PomChain {
    bool tried = false, ok = false;
    unsigned long long* tile_seq = nullptr; /* device, one word per tile (pom_kernels.h: StepParams.tile_seq) */
    uint32_t* aux = nullptr;                /* device: [0] the kernels' POM_CHAIN_E_* flags, [1] number of poisoned tiles, then (tile, stored) pairs */
    uint32_t* aux_host = nullptr;           /* pinned: where chain_settle reads aux[0..1] */
    int64_t tiles = 0;
    uint32_t visits = 0;                    /* visits every tile has had since its word was last zeroed */
    uint32_t turn = 0;                      /* which stream the next launch goes to */
    bool unverified = false;                /* chained launches since the tiles' words were last checked (chain_settle) */
    uint64_t wait_limit = 0;                /* StepParams.chain_wait_limit */
    int64_t wave_slots = 0;                 /* wavefronts of the step kernel the device holds at once: 16 per CU (launch_many_chain's rotation) */
    StepParams last_key;                    /* what the chained launches possibly still in flight were launched with (tick0 = the */
    void (*last_kernel)(StepParams) = nullptr; /* offset between ticks and visits), and which instantiation */
    std::vector<PomChainCall> log;          /* the chained calls since the last settle */
    int64_t stat_launches = 0, stat_settles = 0, stat_tiles_recovered = 0, stat_ticks_replayed = 0; /* pom_batch_chain_stats */
};

static void chain_destroy(PomChain* c)
{
    if (c->tile_seq) (void)hipFree(c->tile_seq);
    if (c->aux) (void)hipFree(c->aux);
    if (c->aux_host) (void)hipHostFree(c->aux_host);
    c->tile_seq = nullptr;
    c->aux = c->aux_host = nullptr;
    c->ok = false;
    c->log.clear();
}

/* the tile words and the flag / list buffer (zeroed on `stream`, which the chained launches' streams are forked from), and the
 * probe of the workgroup -> XCD pattern; synchronises `stream` (pom_batch_create calls it).  false: chained launches are not to
 * be had on this device (the caller launches the ordinary way); nothing stays allocated then */
static bool chain_setup(PomChain* c, int64_t tiles, hipStream_t stream)
{
    if (c->tried) return c->ok;
    c->tried = true;
    c->tiles = tiles;
    /* how long a wavefront waits for its tile: POM_CHAIN_WAIT_US (tests force give-ups with 0), default two seconds */
    const char* wl = getenv("POM_CHAIN_WAIT_US");
    c->wait_limit = (uint64_t)(wl ? atoll(wl) : (long long)POM_CHAIN_WAIT_LIMIT_US) * 100u; /* the wall clock ticks at 100 MHz */
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            c->wave_slots = (int64_t)cus * 20; /* the step kernel's occupancy: 5 wavefronts per SIMD (84 - 93 registers; the LDS tile would allow 22 per CU) */
    }
#if defined(POM_CHAIN_DIAG)
    const size_t words = (size_t)tiles * (POM_CHAIN_WORD_STRIDE + 68); /* + 68 diagnostic words per tile */
#else
    const size_t words = (size_t)tiles * POM_CHAIN_WORD_STRIDE;
#endif
    const size_t aux_bytes = (size_t)(2 + 2 * tiles) * 4;
    bool good = hipMalloc((void**)&c->tile_seq, words * 8) == hipSuccess &&
                hipMemsetAsync(c->tile_seq, 0, words * 8, stream) == hipSuccess && /* on the handle's stream: the fork orders the launches behind it */
                hipMalloc((void**)&c->aux, aux_bytes) == hipSuccess && hipMemsetAsync(c->aux, 0, aux_bytes, stream) == hipSuccess &&
                hipHostMalloc((void**)&c->aux_host, 64, hipHostMallocDefault) == hipSuccess;
    /* the assumption behind the tile choice, probed twice: workgroups b, b + 8, b + 16 ... of a launch land on one XCD, and the
     * eight residues on eight different XCDs numbered 0 .. 7 */
    enum { PROBE = 64 };
    uint32_t* probe_dev = reinterpret_cast<uint32_t*>(c->tile_seq); /* not yet in use; zeroed again below */
    for (int pass = 0; pass < 2 && good; pass++) {
        uint32_t got[PROBE];
        pom_chain_probe_kernel<<<dim3(PROBE), dim3(64), 0, stream>>>(probe_dev);
        good = hipGetLastError() == hipSuccess && hipMemcpyAsync(got, probe_dev, sizeof got, hipMemcpyDeviceToHost, stream) == hipSuccess &&
               hipStreamSynchronize(stream) == hipSuccess;
        uint32_t seen = 0;
        for (int b = 0; b < PROBE && good; b++) {
            if (got[b] >= 8u || got[b] != got[b % 8]) good = false;
            if (b < 8) seen |= 1u << got[b];
        }
        if (seen != 0xFFu) good = false;
    }
    if (good) good = hipMemsetAsync(c->tile_seq, 0, PROBE * 4, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    if (!good) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream);
        chain_destroy(c); /* nothing is kept for a handle that will never chain */
        return false;
    }
    try { /* the call log never grows past 256 entries (launch_many_chain settles first): reserved here, so that logging a call cannot throw */
        c->log.reserve(260);
    } catch (...) {
        chain_destroy(c);
        return false;
    }
    c->ok = true;
    return true;
}

#endif /* POM_CHAIN_H_ */
