/*
 * pom_hsa.h — POM_ISSUE_CHAIN_HSA: chained launches (pom_chain.h) written as AQL packets to HSA queues of the library's own instead
 * of being launched through HIP streams (host side only: the kernels are the CHAIN instantiations HIP loaded, found through the
 * loader extension, so both paths run the same code object).
 *
 * Why: HIP ends every kernel with a release and starts the next one of the stream with an acquire (cache write-back / invalidate):
 * a stream's next launch starts ~2.7 us after its previous one has ended (scripts/experiments/chain/chain_diag.py), and a step costs
 * (launch duration + that gap) / streams.  Between chained launches those fences do nothing — a tile goes from one visit to the
 * next through one L2, ordered by its ticket word — and an AQL packet can say so: fence scope NONE on every packet but a call's
 * first (acquire: what HIP wrote before) and last (release: what HIP reads after).
 *
 * A call BLOCKS until its launches are done: HIP (streams, events, hipDeviceSynchronize) knows nothing of these queues, so the
 * library itself is the only one who can order them with everything else; what HIP has queued for the batch is waited for before
 * the first packet is written.  Kernel arguments live in device memory written through the PCIe BAR (from host memory every
 * wavefront would fetch them over PCIe: 31 instead of 5 us per launch, scripts/experiments/chain/hsa_chain_test.hip); they are the
 * same for all launches of a call (the ticket decides the tick).  Anything that cannot be set up (no HSA agent for the HIP device,
 * no host-writable device memory, kernel not found) leaves the handle with HIP's streams.
 */
#ifndef POM_HSA_H_
#define POM_HSA_H_

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>

struct PomHsa {
    bool tried = false, ok = false, inited = false;
    hsa_agent_t gpu{}, cpu{};
    enum { MAX_Q = 4, KA_SLOTS = 32, KA_BYTES = 512, N_KERNELS = 8, PROF_MAX = 256 };
    hsa_queue_t* q[MAX_Q] = {};
    hsa_signal_t done[MAX_Q] = {};
    char* kernargs = nullptr; /* device memory, host-visible: KA_SLOTS blocks of KA_BYTES */
    int ka_next = 0;
    uint64_t kernel_object[N_KERNELS] = {};
    uint32_t lds[N_KERNELS] = {};
    bool kernel_tried[N_KERNELS] = {};
    hsa_signal_t prof_sig[PROF_MAX] = {}; /* pom_batch_profile: a completion signal per launch */
    char why[200] = {};
};

#define POM_HSA_TRY(x)                                                               \
    do {                                                                             \
        const hsa_status_t s_ = (x);                                                 \
        if (s_ != HSA_STATUS_SUCCESS) {                                              \
            const char* m_ = "";                                                     \
            hsa_status_string(s_, &m_);                                              \
            snprintf(c->why, sizeof c->why, "%s: %s", #x, m_);                       \
            return false;                                                            \
        }                                                                            \
    } while (0)

struct PomHsaFind {
    hsa_agent_t gpu, cpu;
    bool have_gpu, have_cpu;
    uint32_t want_bdf, want_domain;
    hsa_amd_memory_pool_t pool;
    bool have_pool;
    const char* name;
    uint64_t kobj;
    uint32_t lds, priv, kasize;
    bool found;
};

static hsa_status_t hsa_agent_cb(hsa_agent_t a, void* data)
{
    PomHsaFind* f = (PomHsaFind*)data;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) {
        f->cpu = a;
        f->have_cpu = true;
    }
    if (t == HSA_DEVICE_TYPE_GPU && !f->have_gpu) {
        uint32_t bdf = 0, domain = 0;
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
        if ((bdf >> 3) == (f->want_bdf >> 3) && domain == f->want_domain) { /* bus and device; the function is 0 */
            f->gpu = a;
            f->have_gpu = true;
        }
    }
    return HSA_STATUS_SUCCESS;
}
/* device memory the host may write: the coarse-grained pool of the GPU */
static hsa_status_t hsa_pool_cb(hsa_amd_memory_pool_t pool, void* data)
{
    PomHsaFind* f = (PomHsaFind*)data;
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL)
        return HSA_STATUS_SUCCESS;
    bool alloc = false;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    hsa_amd_agent_memory_pool_get_info(f->cpu, pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
    if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && acc != HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED && !f->have_pool) {
        f->pool = pool;
        f->have_pool = true;
    }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t hsa_exe_cb(hsa_executable_t exe, void* data)
{
    PomHsaFind* f = (PomHsaFind*)data;
    hsa_executable_symbol_t sym;
    if (hsa_executable_get_symbol_by_name(exe, f->name, &f->gpu, &sym) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &f->kobj);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &f->lds);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &f->priv);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &f->kasize);
    f->found = true;
    return HSA_STATUS_SUCCESS;
}

static void hsa_destroy(PomHsa* c)
{
    for (int k = 0; k < PomHsa::MAX_Q; k++) {
        if (c->q[k]) hsa_queue_destroy(c->q[k]);
        if (c->done[k].handle) hsa_signal_destroy(c->done[k]);
    }
    for (int k = 0; k < PomHsa::PROF_MAX; k++)
        if (c->prof_sig[k].handle) hsa_signal_destroy(c->prof_sig[k]);
    if (c->kernargs) hsa_amd_memory_pool_free(c->kernargs);
    const bool inited = c->inited;
    *c = PomHsa();
    if (inited) hsa_shut_down(); /* hsa_init is reference-counted; HIP keeps its own reference */
}

/* queues, kernel-argument ring, signals: on first use.  false: not to be had here (c->why says why) */
static bool hsa_setup(PomHsa* c, int device, int queues)
{
    if (c->tried) return c->ok;
    c->tried = true;
    POM_HSA_TRY(hsa_init());
    c->inited = true;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        snprintf(c->why, sizeof c->why, "no properties for HIP device %d", device);
        return false;
    }
    PomHsaFind f;
    memset(&f, 0, sizeof f);
    f.want_bdf = ((uint32_t)prop.pciBusID << 8) | ((uint32_t)prop.pciDeviceID << 3);
    f.want_domain = (uint32_t)prop.pciDomainID;
    POM_HSA_TRY(hsa_iterate_agents(hsa_agent_cb, &f));
    if (!f.have_gpu || !f.have_cpu) {
        snprintf(c->why, sizeof c->why, "no HSA agent for HIP device %d (pci %04x:%02x:%02x)", device, prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
        return false;
    }
    c->gpu = f.gpu;
    c->cpu = f.cpu;
    POM_HSA_TRY(hsa_amd_agent_iterate_memory_pools(f.gpu, hsa_pool_cb, &f));
    if (!f.have_pool) {
        snprintf(c->why, sizeof c->why, "no device memory pool the host can write (no large BAR?)");
        return false;
    }
    for (int k = 0; k < queues && k < PomHsa::MAX_Q; k++) {
        POM_HSA_TRY(hsa_queue_create(c->gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &c->q[k]));
        POM_HSA_TRY(hsa_signal_create(0, 0, nullptr, &c->done[k]));
    }
    POM_HSA_TRY(hsa_amd_memory_pool_allocate(f.pool, (size_t)PomHsa::KA_SLOTS * PomHsa::KA_BYTES, 0, (void**)&c->kernargs));
    hsa_agent_t both[2] = {c->gpu, c->cpu};
    POM_HSA_TRY(hsa_amd_agents_allow_access(2, both, nullptr, c->kernargs));
    c->ok = true;
    return true;
}

/* the kernel object of chained instantiation `which` (bit 2 fresh boards, bit 1 fused policy, bit 0 reset at the end) as HIP
 * loaded it; 0 if it cannot be found */
static uint64_t hsa_kernel(PomHsa* c, int which, const void* host_fn, uint32_t* lds)
{
    if (!c->kernel_tried[which]) {
        c->kernel_tried[which] = true;
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, host_fn) != hipSuccess) return 0; /* (makes HIP load the code object) */
        hsa_ven_amd_loader_1_03_pfn_t loader;
        if (hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof loader, &loader) != HSA_STATUS_SUCCESS) return 0;
        char name[128];
        snprintf(name, sizeof name, "_Z15pom_step_kernelILi16ELi4ELb%dELb%dELb%dELb1ELb1EEv10StepParams.kd", (which >> 2) & 1, (which >> 1) & 1, which & 1);
        PomHsaFind f;
        memset(&f, 0, sizeof f);
        f.gpu = c->gpu;
        f.name = name;
        if (loader.hsa_ven_amd_loader_iterate_executables(hsa_exe_cb, &f) != HSA_STATUS_SUCCESS || !f.found) return 0;
        /* what the packet below assumes: no scratch, StepParams + the hidden arguments in one block */
        if (f.priv != 0 || f.kasize > PomHsa::KA_BYTES || f.kasize < ((sizeof(StepParams) + 7) & ~size_t(7)) + 24) return 0;
        c->kernel_object[which] = f.kobj;
        c->lds[which] = f.lds;
    }
    *lds = c->lds[which];
    return c->kernel_object[which];
}

/* one AQL kernel dispatch packet: `blocks` workgroups of `block_threads` */
static void hsa_write_packet(hsa_queue_t* q, uint64_t kernel_object, uint32_t lds, const void* kernarg, uint32_t blocks, uint32_t block_threads,
                             int acquire, int release, hsa_signal_t completion, bool barrier)
{
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
    while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) { /* the ring is full: the device is 1024 launches behind */
    }
    hsa_kernel_dispatch_packet_t* pk = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
    pk->workgroup_size_x = (uint16_t)block_threads;
    pk->workgroup_size_y = 1;
    pk->workgroup_size_z = 1;
    pk->reserved0 = 0;
    pk->grid_size_x = blocks * block_threads;
    pk->grid_size_y = 1;
    pk->grid_size_z = 1;
    pk->private_segment_size = 0;
    pk->group_segment_size = lds;
    pk->kernel_object = kernel_object;
    pk->kernarg_address = const_cast<void*>(kernarg);
    pk->reserved2 = 0;
    pk->completion_signal = completion;
    /* the barrier bit makes a packet wait for the queue's previous one; without it each XCD still plays its share of the queue's
     * packets in order (scripts/experiments/chain/chain_diag.py) — either way the tickets are what orders a tile's ticks */
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                       (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    __atomic_store_n(reinterpret_cast<uint32_t*>(pk), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
}

/* a packet that does nothing (a barrier-AND without dependencies): rung before work is about to come (pom_batch_fork), so that a
 * queue that has been idle for a while is mapped and awake when the first launch arrives */
static void hsa_poke(PomHsa* c)
{
    for (int k = 0; k < PomHsa::MAX_Q; k++) {
        hsa_queue_t* q = c->q[k];
        if (!q) continue;
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {
        }
        hsa_barrier_and_packet_t* pk = (hsa_barrier_and_packet_t*)q->base_address + (idx & (q->size - 1));
        memset(reinterpret_cast<char*>(pk) + 4, 0, sizeof *pk - 4);
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (HSA_FENCE_SCOPE_NONE << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                           (HSA_FENCE_SCOPE_NONE << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        __atomic_store_n(reinterpret_cast<uint32_t*>(pk), (uint32_t)header, __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
    }
}

#endif /* POM_HSA_H_ */
