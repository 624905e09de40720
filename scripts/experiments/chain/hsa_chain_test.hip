// Experiment (round 3): can dependent launches of ONE queue overlap when the per-tile dependency is carried by a sequence word
// instead of the AQL barrier bit?  HIP always sets the barrier bit on gfx9 (hipExtAnyOrderLaunch is ignored: anyorder.hip), so the
// packets are written to an HSA queue of our own.  The kernel object is the one HIP loaded (found through the loader extension).
// Toy kernel: block b of launch k waits until seq[b] == k, "works" for a pseudo-random time, stores seq[b] = k + 1.
//   usage: hsa_chain_test [launches] [blocks] [work_ticks]
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HSACHK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = ""; hsa_status_string(s_, &m_); fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m_); exit(2); } } while (0)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct ToyParams {
    uint32_t* seq;
    uint32_t* err;  /* [0] spin limit hit, [1] XCC changed, [2] max spins seen */
    uint32_t* xcc;
    uint32_t expect, work, chain, pad;
    unsigned long long* stamps; /* per launch: first start, last end (100 MHz) */
    uint64_t id0; /* ~0: expect is given; else expect = dispatch id - id0 (one kernarg block for all launches of a run) */
};

extern "C" __device__ uint64_t pom_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id"); /* the AQL packet's index in its queue */

__global__ __launch_bounds__(64) void toy(ToyParams p)
{
    const uint32_t b = blockIdx.x;
    if (p.id0 != ~0ull) p.expect = (uint32_t)(pom_dispatch_id() - p.id0);
    uint32_t xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    xcc_id &= 0xF;
    if (p.chain) {
        uint32_t seen = 0, spins = 0;
        for (;;) {
            seen = __hip_atomic_load(p.seq + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen == p.expect || ++spins > 100000u) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (threadIdx.x == 0 && spins > 1000) atomicMax(p.err + 2, spins);
        if (seen != p.expect) {
            if (threadIdx.x == 0) atomicAdd(p.err, 1u);
            return;
        }
    }
    if (threadIdx.x == 0 && p.stamps && p.expect < 64) atomicMin(p.stamps + 2 * p.expect, (unsigned long long)wall_clock64());
    if (threadIdx.x == 0) {
        if (p.expect == 0) p.xcc[b] = xcc_id;
        else if (p.xcc[b] != xcc_id) atomicAdd(p.err + 1, 1u);
    }
    uint32_t h = (b * 2654435761u) ^ (p.expect * 40503u);
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    uint32_t iters = p.work / 2 + (h % (p.work + 1));           /* 0.5 .. 1.5 x work iterations of a dependent ALU chain */
    if ((h >> 20) % 50 == 0) iters += p.work;                    /* a slow one now and then */
    uint32_t v = h + threadIdx.x;
    for (uint32_t i = 0; i < iters; i++) {
        v = v * 1664525u + 1013904223u;
        v ^= v >> 13;
        v = v * 22695477u + 1u;
        v ^= v >> 11;
    }
    if (v == 0x12345678u) p.err[3] = v; /* keep the chain alive */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_store(p.seq + b, p.expect + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0 && p.stamps && p.expect < 64) atomicMax(p.stamps + 2 * p.expect + 1, (unsigned long long)wall_clock64());
}

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_amd_memory_pool_t g_kernarg_pool;
static bool g_have_pool = false;

static hsa_status_t agent_cb(hsa_agent_t a, void*)
{
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void*)
{
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_pool) { g_kernarg_pool = pool; g_have_pool = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_vram_pool;
static bool g_have_vram = false;
static hsa_status_t vram_cb(hsa_amd_memory_pool_t pool, void*)
{
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    bool alloc = false;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    hsa_amd_agent_memory_pool_get_info(g_cpu, pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
    printf("gpu pool: flags %#x alloc %d cpu access %d\n", flags, (int)alloc, (int)acc);
    const int want = getenv("VRAM_FINE") ? HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED : HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED;
    if (alloc && (flags & want) && acc != HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED && !g_have_vram) { g_vram_pool = pool; g_have_vram = true; }
    return HSA_STATUS_SUCCESS;
}
struct Find { const char* name; uint64_t kobj; uint32_t lds, priv, kasize; bool found; };
static hsa_status_t exe_cb(hsa_executable_t exe, void* data)
{
    Find* f = (Find*)data;
    hsa_executable_symbol_t sym;
    if (hsa_executable_get_symbol_by_name(exe, f->name, &g_gpu, &sym) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &f->kobj);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &f->lds);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &f->priv);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &f->kasize);
    f->found = true;
    return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int blocks = argc > 2 ? atoi(argv[2]) : 4096;
    const int work = argc > 3 ? atoi(argv[3]) : 800; /* 8 us */
    HIPCHK(hipSetDevice(0));
    uint32_t *seq, *err, *xcc;
    HIPCHK(hipMalloc(&seq, blocks * 4));
    HIPCHK(hipMalloc(&err, 16));
    HIPCHK(hipMalloc(&xcc, blocks * 4));
    unsigned long long* stamps;
    HIPCHK(hipMalloc(&stamps, 128 * 8));
    hipFuncAttributes fa;
    HIPCHK(hipFuncGetAttributes(&fa, (const void*)toy)); /* makes HIP load the code object */
    { /* and one ordinary launch, to be sure it is resident */
        ToyParams p{seq, err, xcc, 0, 1, 0, 0, nullptr, ~0ull};
        HIPCHK(hipMemset(seq, 0, blocks * 4));
        toy<<<1, 64>>>(p);
        HIPCHK(hipDeviceSynchronize());
    }
    HSACHK(hsa_init());
    HSACHK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu || !g_have_cpu) { fprintf(stderr, "no agents\n"); return 2; }
    HSACHK(hsa_amd_agent_iterate_memory_pools(g_cpu, pool_cb, nullptr));
    if (!g_have_pool) { fprintf(stderr, "no kernarg pool\n"); return 2; }
    hsa_ven_amd_loader_1_03_pfn_t loader;
    HSACHK(hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof loader, &loader));
    Find f{"_Z3toy9ToyParams.kd", 0, 0, 0, 0, false};
    HSACHK(loader.hsa_ven_amd_loader_iterate_executables(exe_cb, &f));
    if (!f.found) { fprintf(stderr, "kernel symbol not found among the loaded executables\n"); return 2; }
    printf("kernel object %#llx lds %u private %u kernarg %u\n", (unsigned long long)f.kobj, f.lds, f.priv, f.kasize);

    hsa_queue_t* q = nullptr;
    HSACHK(hsa_queue_create(g_gpu, 1024, getenv("QMULTI") ? HSA_QUEUE_TYPE_MULTI : HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    if (getenv("QPRIO")) HSACHK(hsa_amd_queue_set_priority(q, HSA_AMD_QUEUE_PRIORITY_HIGH));
    const int RING = 256;
    const int RING_ONCE = getenv("RING_ONCE") ? atoi(getenv("RING_ONCE")) : 0;
    const int SIG_EVERY = getenv("SIG_EVERY") ? atoi(getenv("SIG_EVERY")) : 1;
    const int ACQ = getenv("ACQ") ? atoi(getenv("ACQ")) : HSA_FENCE_SCOPE_AGENT, REL = getenv("REL") ? atoi(getenv("REL")) : HSA_FENCE_SCOPE_AGENT;
    char* kernargs = nullptr;
    const size_t KA = 512;
    if (getenv("KA_VRAM")) {
        HSACHK(hsa_amd_agent_iterate_memory_pools(g_gpu, vram_cb, nullptr));
        if (!g_have_vram) { fprintf(stderr, "no host-accessible device pool\n"); return 2; }
        HSACHK(hsa_amd_memory_pool_allocate(g_vram_pool, KA * RING, 0, (void**)&kernargs));
        hsa_agent_t both[2] = {g_gpu, g_cpu};
        HSACHK(hsa_amd_agents_allow_access(2, both, nullptr, kernargs));
    } else {
        HSACHK(hsa_amd_memory_pool_allocate(g_kernarg_pool, KA * RING, 0, (void**)&kernargs));
        HSACHK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kernargs));
    }
    std::vector<hsa_signal_t> sig(RING);
    for (auto& s : sig) HSACHK(hsa_signal_create(0, 0, nullptr, &s));

    auto run = [&](int chain, int barrier) {
        HIPCHK(hipMemset(seq, 0, blocks * 4));
        HIPCHK(hipMemset(err, 0, 16));
        {
            unsigned long long init[128];
            for (int i = 0; i < 64; i++) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
            HIPCHK(hipMemcpy(stamps, init, sizeof init, hipMemcpyHostToDevice));
        }
        HIPCHK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < launches; k++) {
            const int slot = k % RING;
            const bool signals = (k % SIG_EVERY) == SIG_EVERY - 1 || k == launches - 1;
            { /* the slot's previous user belongs to the group that ends at the next signalling slot of the previous lap */
                int g = slot;
                while (g < RING - 1 && (g % SIG_EVERY) != SIG_EVERY - 1) g++;
                while (hsa_signal_wait_scacquire(sig[g], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {
                }
            }
            if (signals) hsa_signal_store_relaxed(sig[slot], 1);
            static int run_no = 0;
            const bool one_ka = getenv("ONE_KA") != nullptr;
            if (k == 0) run_no++;
            char* ka = kernargs + KA * (one_ka ? run_no % RING : slot);
            if (!one_ka || k == 0) {
                char img[512];
                memset(img, 0, sizeof img);
                ToyParams p{seq, err, xcc, (uint32_t)k, (uint32_t)work, (uint32_t)chain, 0, stamps, one_ka ? hsa_queue_load_write_index_relaxed(q) : ~0ull};
                memcpy(img, &p, sizeof p);
                /* hidden arguments (code object v5): block counts, group sizes, remainders, ..., grid dims */
                const size_t hid = (sizeof p + 7) & ~size_t(7);
                uint32_t bc[3] = {(uint32_t)blocks, 1, 1};
                uint16_t gs[6] = {64, 1, 1, 0, 0, 0};
                memcpy(img + hid, bc, 12);
                memcpy(img + hid + 12, gs, 12);
                memcpy(ka, img, KA);
                if (getenv("KA_VRAM")) { /* the writes went through the BAR: push them out and make sure they have landed */
                    __builtin_ia32_sfence();
                    volatile uint32_t* rb = (volatile uint32_t*)(ka + hid);
                    if (*rb != (uint32_t)blocks) { fprintf(stderr, "kernarg readback mismatch\n"); exit(2); }
                }
            }
            const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
            while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {
            }
            hsa_kernel_dispatch_packet_t* pk = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
            pk->workgroup_size_x = 64; pk->workgroup_size_y = 1; pk->workgroup_size_z = 1;
            pk->reserved0 = 0;
            pk->grid_size_x = (uint32_t)blocks * 64; pk->grid_size_y = 1; pk->grid_size_z = 1;
            pk->private_segment_size = f.priv;
            pk->group_segment_size = f.lds;
            pk->kernel_object = f.kobj;
            pk->kernarg_address = ka;
            pk->reserved2 = 0;
            pk->completion_signal = signals ? sig[slot] : hsa_signal_t{0};
            const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                               (ACQ << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                               (((signals ? HSA_FENCE_SCOPE_AGENT : REL)) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
            const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            __atomic_store_n((uint32_t*)pk, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
            if (!RING_ONCE || k == launches - 1) hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
        }
        for (int s = 0; s < RING; s++)
            while (hsa_signal_wait_scacquire(sig[s], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {
            }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        std::vector<uint32_t> hs(blocks);
        uint32_t he[4];
        HIPCHK(hipMemcpy(hs.data(), seq, blocks * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(he, err, 16, hipMemcpyDeviceToHost));
        int wrong = 0;
        for (int b = 0; b < blocks; b++) wrong += hs[b] != (uint32_t)launches;
        printf("sig/%d acq %d rel %d chain %d barrier %d: %d launches of %d blocks, work %.1f us mean: %.2f us per launch; wrong seq %d, spin-limit %u, xcc changes %u, max spins %u\n",
               SIG_EVERY, ACQ, REL, chain, barrier, launches, blocks, work / 100.0, us / launches, wrong, he[0], he[1], he[2]);
        if (getenv("STAMPS")) {
            unsigned long long st[128];
            HIPCHK(hipMemcpy(st, stamps, sizeof st, hipMemcpyDeviceToHost));
            for (int i = 0; i < 8 && i < launches; i++)
                printf("    launch %d: first wave starts at %6.2f us, last wave ends at %6.2f us\n", i, (st[2 * i] - st[0]) / 100.0, (st[2 * i + 1] - st[0]) / 100.0);
        }
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; rep++) { /* the same launches through HIP, for comparison */
        HIPCHK(hipMemset(seq, 0, blocks * 4));
        HIPCHK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < launches; k++) {
            ToyParams p{seq, err, xcc, (uint32_t)k, (uint32_t)work, 0, 0, nullptr, ~0ull};
            toy<<<blocks, 64>>>(p);
        }
        HIPCHK(hipDeviceSynchronize());
        printf("HIP stream: %.2f us per launch\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches);
    }
    run(0, 1);
    run(1, 0);
    run(0, 1);
    run(1, 0);
    hsa_queue_destroy(q);
    return 0;
}
