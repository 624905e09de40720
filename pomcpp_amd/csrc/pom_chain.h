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
 *    compares the XCD it runs on with the one recorded in the tile's word and refuses to step otherwise (flag bit 1);
 *  - a wavefront only ever waits for the holder of the previous ticket of its tile, which therefore is resident and running:
 *    no order of dispatch can deadlock.  Should a wait still not end (400 k polls) the wavefront gives up and flags it (bit 0):
 *    no launch can hang the device.
 * A flag makes the next call that looks (every call that joins or launches) fail with POM_E_HIP; the handle then launches the
 * ordinary way.  Before the first chained launch of a handle a 64-workgroup probe checks the workgroup -> XCD pattern itself (a
 * partitioned device, or another chip, does not have it): no chained launch is ever issued where it does not hold.
 */
#ifndef POM_CHAIN_H_
#define POM_CHAIN_H_

struct PomChain {
    bool tried = false, ok = false;
    unsigned long long* tile_seq = nullptr; /* device, one word per tile (pom_kernels.h: StepParams.tile_seq) */
    uint32_t* err_host = nullptr;           /* pinned, device-visible: the kernel's failure flags */
    uint32_t* err_dev = nullptr;
    uint32_t visits = 0;                    /* visits every tile has had since its word was last zeroed */
    uint32_t turn = 0;                      /* which stream the next launch goes to */
    bool unverified = false;                /* chained launches since the visit counts were last checked */
    StepParams last_key;                    /* what the chained launches possibly still in flight were launched with (tick0 = the */
    void (*last_kernel)(StepParams) = nullptr; /* offset between ticks and visits), and which instantiation */
};

static void chain_destroy(PomChain* c)
{
    if (c->tile_seq) (void)hipFree(c->tile_seq);
    if (c->err_host) (void)hipHostFree(c->err_host);
    *c = PomChain();
}

/* the tile words (zeroed on `stream`, which the chained launches' streams are forked from) and the flag page, on first use;
 * false: not to be had (the caller launches the ordinary way) */
static bool chain_setup(PomChain* c, int64_t tiles, hipStream_t stream)
{
    if (c->tried) return c->ok;
    c->tried = true;
#if defined(POM_CHAIN_DIAG)
    const size_t words = (size_t)tiles * (POM_CHAIN_WORD_STRIDE + 68); /* + 68 diagnostic words per tile */
#else
    const size_t words = (size_t)tiles * POM_CHAIN_WORD_STRIDE;
#endif
    if (hipMalloc((void**)&c->tile_seq, words * 8) != hipSuccess ||
        hipMemsetAsync(c->tile_seq, 0, words * 8, stream) != hipSuccess || /* on the handle's stream: the fork orders the launches behind it */
        hipHostMalloc((void**)&c->err_host, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->err_dev, c->err_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    *c->err_host = 0;
    /* the assumption behind the tile choice, probed once (twice, on this stream): workgroups b, b + 8, b + 16 ... of a launch land
     * on one XCD, and the eight residues on eight different XCDs numbered 0 .. 7 */
    enum { PROBE = 64 };
    uint32_t* probe_dev = reinterpret_cast<uint32_t*>(c->tile_seq); /* not yet in use; zeroed again below */
    for (int pass = 0; pass < 2; pass++) {
        uint32_t got[PROBE];
        pom_chain_probe_kernel<<<dim3(PROBE), dim3(64), 0, stream>>>(probe_dev);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(got, probe_dev, sizeof got, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        uint32_t seen = 0;
        for (int b = 0; b < PROBE; b++) {
            if (got[b] >= 8u || got[b] != got[b % 8]) return false;
            if (b < 8) seen |= 1u << got[b];
        }
        if (seen != 0xFFu) return false;
    }
    if (hipMemsetAsync(c->tile_seq, 0, PROBE * 4, stream) != hipSuccess) return false;
    c->ok = true;
    return true;
}

#endif /* POM_CHAIN_H_ */
