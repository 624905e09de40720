/*
 * pom_kernels.h — the gfx950 kernels of the batched Pommerman stepper (device side only; the host runtime that launches
 * them is pom_runtime.h, the C-ABI of include/pom_batch.h is pom_batch.hip).  Written for MI355X only: 64-lane wavefronts,
 * one wavefront per workgroup, each env's board / bomb queue / flame queue in a column of the wavefront's LDS tile
 * ([row][lane]: every per-lane dynamic index is an LDS address, no shuffles, no scratch).
 *
 * The tick itself is pom_step_body.h; this file is the data movement around it:
 *   HBM (packed records in 16-env tiles, pom_packed.h) -> LDS tile + VGPRs -> tick(s) -> HBM,
 * the SimpleAgent policy kernel, the observation export, the board generator, the pack / unpack between the boundary's 1004-byte States and the tiles,
 * status extraction and counters.
 */
#ifndef POM_KERNELS_H_
#define POM_KERNELS_H_

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "pom_batch.h"
#include "pom_boardgen_body.h"
#include "pom_packed.h"
#include "pom_policy_body.h"
#include "pom_step_body.h"

/* ---------------------------------------------------------------------------------------------
 * LDS tile of one wavefront, [row][EPW] dwords.  Rows 0..79 mirror the HBM record row for row (pom_packed.h), so the
 * whole record moves in groups of 64/EPW rows; then 5 rows of bomb-destination bytes and 31 rows that hold loop B's per-cell
 * counters while it chooses its bombs and the explosion frames (21 rows) from then on — the counters are read for the last time
 * before the first blast of loop B can push a frame, and a wavefront's envs go through both in step —: 116 rows = 29.7 / 14.8 /
 * 7.4 KB for 64 / 32 / 16 envs per wavefront (137 rows until round 5: 18 wavefronts of 16 envs per CU; now 22).
 * ------------------------------------------------------------------------------------------- */
enum {
    ROW_BOARD = POM_REC_BOARD,    /* 31 rows: four 8-bit cells per dword            */
    ROW_BOMBS = POM_REC_BOMBS,    /* 20 rows: raw bomb words, physical queue slots  */
    ROW_FLAMES = POM_REC_FLAMES,  /* 20 rows                                        */
    ROW_BDEST = POM_REC_DWORDS,   /*  5 rows: 20 bytes, bomb destination snapshot   */
    ROW_STACK = POM_REC_DWORDS + 5, /* 21 rows: explosion frames                    */
    ROW_CLAIMS = POM_REC_DWORDS + 5,  /* 31 rows OVER the frames: a byte per cell and env, [env][124] (PomStepper::loop_b_todo) */
    LDS_ROWS = POM_REC_DWORDS + 36
};
static_assert(POM_REC_DWORDS % 2 == 0, "the one-lane shapes move the record in groups of 1 or 2 rows");

/*
 * EPW = envs per wavefront (64, 32 or 16), G = lanes that work on one env during the tick (pom_step_body.h):
 *   G = 1  one lane per env runs the tick; with EPW < 64 the other lanes only help moving the record (each DMA /
 *          store instruction then covers 64/EPW rows), trading idle lanes for more resident wavefronts per SIMD;
 *   G = 4  (EPW = 16) the four ADJACENT lanes 4e..4e+3 run env e's tick in lock-step and split the order-free
 *          parts; their cross-lane traffic is DPP quad permutes (one VALU op, no LDS).
 * The LDS tile is [row][EPW]; bank = (row*EPW + env) mod 32; the G lanes of an env read the same address (broadcast).
 */
/* Where cell c of the wavefront's env number el lives in the LDS tile, in bytes.  The tile is a copy of 64 / 16 ... EPW / 16 HBM tiles
 * side by side, row for row, and an HBM tile's board is laid out by cell (pom_packed.h): for the shipped 16 envs per wavefront the
 * offset is c * 16 + el — ONE v_lshl_add_u32 per access. */
template <int EPW>
__device__ __forceinline__ int tile_cell_byte(int el, int c)
{
    return EPW == 16 ? c * 16 + el : (c >> 2) * (4 * EPW) + (c & 3) * 16 + (el >> 4) * 64 + (el & 15);
}

template <int EPW, int GG>
struct LdsEnv {
    static constexpr int G = GG;
    uint32_t* t; /* &tile[env_in_wave] */
    int sub_;    /* lane's index within its env's group; 0 = owner */
    uint8_t* b;  /* the env's cell 0 (tile_cell_byte) */
    uint32_t* tile0; /* the wavefront's tile (wave-uniform) */
    __device__ LdsEnv(uint32_t* tile, int el, int sub)
        : t(tile + el), sub_(sub), b(reinterpret_cast<uint8_t*>(tile) + tile_cell_byte<EPW>(el, 0)), tile0(tile) {}
    /* the env's cell counters: 124 bytes of its own behind the frames, env after env — worked out where they are needed (loop B of a
     * tick with moving bombs) instead of living in a register through the whole tick */
    __device__ uint8_t* claim_map() const { return reinterpret_cast<uint8_t*>(tile0 + ROW_CLAIMS * EPW) + (int)(t - tile0) * 124; }
    __device__ int sub() const { return G == 1 ? 0 : sub_; }
    __device__ bool owner() const { return G == 1 || sub_ == 0; }
    /* quad reductions: lane ^ 1, then lane ^ 2 */
    __device__ static int dpp_x1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
    __device__ static int dpp_x2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }
    __device__ int gor(int v) const
    {
        if (G == 1) return v;
        v |= dpp_x1(v);
        return v | dpp_x2(v);
    }
    __device__ int gadd(int v) const
    {
        if (G == 1) return v;
        v += dpp_x1(v);
        return v + dpp_x2(v);
    }
    template <int J>
    __device__ int gbcast(int v) const /* the value held by member J of the quad */
    {
        if (G == 1) return v;
        return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xF, 0xF, true);
    }
    __device__ int gmin(int v) const
    {
        if (G == 1) return v;
        const int a = dpp_x1(v);
        v = a < v ? a : v;
        const int b = dpp_x2(v);
        return b < v ? b : v;
    }
    /* "the other lanes' LDS writes so far are visible from here on": true at every instruction of a wavefront in lock-step, so
     * nothing to do; the four-lane host model of tests/emul makes its lanes meet here */
    __device__ void sync() const {}
    __device__ int cell(int c) const { return b[EPW == 16 ? c * 16 : (c >> 2) * (4 * EPW) + (c & 3) * 16]; }
    __device__ void put_cell(int c, int v) { b[EPW == 16 ? c * 16 : (c >> 2) * (4 * EPW) + (c & 3) * 16] = (uint8_t)v; }
    __device__ int bomb(int s) const { return (int)t[(ROW_BOMBS + s) * EPW]; }
    __device__ void put_bomb(int s, int v) { t[(ROW_BOMBS + s) * EPW] = (uint32_t)v; }
    __device__ int flame(int s) const { return (int)t[(ROW_FLAMES + s) * EPW]; }
    __device__ void put_flame(int s, int v) { t[(ROW_FLAMES + s) * EPW] = (uint32_t)v; }
    __device__ int bdest(int i) const { return reinterpret_cast<const uint8_t*>(t + ROW_BDEST * EPW)[(i >> 2) * (4 * EPW) + (i & 3)]; }
    __device__ void put_bdest(int i, int v) { reinterpret_cast<uint8_t*>(t + ROW_BDEST * EPW)[(i >> 2) * (4 * EPW) + (i & 3)] = (uint8_t)v; }
    __device__ int frame(int d) const { return (int)t[(ROW_STACK + d) * EPW]; }
    __device__ int ag1(int i) const { return (int)t[(POM_REC_AGENTS + 2 * i + 1) * EPW]; }
    __device__ void put_ag1(int i, int v) { t[(POM_REC_AGENTS + 2 * i + 1) * EPW] = (uint32_t)v; }
    __device__ void set_ag1(int i, int v) { if (owner()) put_ag1(i, v); }
    /* (loop_b_todo) cleared by the env's lanes together, a dword each per round; counted up with LDS atomics (two lanes of an env
     * may count the same cell) */
    __device__ void claims_clear()
    {
        uint32_t* cm = reinterpret_cast<uint32_t*>(claim_map());
#pragma unroll
        for (int k = 0; k < (31 + G - 1) / G; k++)
            if (G * k + G - 1 < 31 || sub() + G * k < 31) cm[sub() + G * k] = 0u;
    }
    __device__ void claim(int c)
    {
        __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(claim_map() + (c & ~3)), 1u << (8 * (c & 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ int claims(int c) const { return claim_map()[c]; }
    /* replicated code: all G lanes get here with the same value, the owner writes */
    __device__ void set_cell(int c, int v) { if (owner()) put_cell(c, v); }
    __device__ void set_bomb(int s, int v) { if (owner()) put_bomb(s, v); }
    __device__ void set_flame(int s, int v) { if (owner()) put_flame(s, v); }
    __device__ void set_bdest(int i, int v) { if (owner()) put_bdest(i, v); }
    __device__ void set_frame(int d, int v) { if (owner()) t[(ROW_STACK + d) * EPW] = (uint32_t)v; }
};

struct StepParams {
    uint32_t* state;
    const uint32_t* snap;
    const int32_t* moves; /* device int32[n][4], or nullptr: synthetic stream */
    int64_t* wave_counters;
    int64_t n, n_pad, env_offset;
    uint64_t seed;
    uint32_t tick0;            /* the launch's first tick is tick0 + *tick_base */
    const uint32_t* tick_base; /* a device word: 0 for a launch of its own, the chunk's first tick for a node of a replayed graph (pom_runtime.h) */
    int32_t dist, ticks, mode, auto_reset, max_steps;
    int64_t block0, block_end; /* this launch covers tiles block0 .. block_end - 1 (sub-batch of a split step) */
    uint32_t* terminal;  /* auto_reset == POM_RESET_AT_END: the last finished episode's final record per env, array of structs */
    uint32_t* agent_mem; /* POLICY instantiation: SimpleAgent memory, [2][4 * n_pad] */
    uint32_t* episode;   /* games started so far per env (fresh boards: keys the next board) */
    uint64_t board_seed;
    int32_t fresh;       /* a restarting env gets the next board of pom_boardgen.h instead of its snapshot */
    /* CHAIN instantiation (pom_chain.h): launches of one queue that do NOT wait for each other — a tile's tick waits for the
     * same tile's previous tick instead.  Per tile one 64-bit word: bits 63..36 visits begun (tickets), 35..32 1 + the XCD
     * that stored the tile last (0: none yet), 27..0 visits stored.  The j-th visitor of a tile plays tick tick0 + (j - chain_seq0)
     * once the word says j visits are stored; which launch a visitor belongs to does not matter. */
    unsigned long long* tile_seq;
    uint32_t* chain_err; /* a device word of POM_CHAIN_E_* flags, read back by the host after the next join (chain_settle) */
    uint32_t chain_seq0;
    uint32_t tape_len;         /* CHAIN with explicit moves: `moves` is a tape int32[tape_len][n][4], the visit at distance d from
                                  chain_seq0 plays tick d of it (0: `moves` is one tick's Move[4] array or nullptr) */
    uint32_t chain_rot; /* chained: the launch's workgroups take their XCD's tiles starting this many positions in (< tiles per XCD; launch_many_chain) */
    uint64_t chain_wait_limit; /* how long a wavefront waits for the visit before its own, in ticks of the 100 MHz wall clock */
    /* OBS instantiation (pom_batch_step_device_observe): the planes / attributes of the state the tick leaves behind, written by
     * the same launch while the tile is still in LDS (pom_batch.h pom_batch_observe for the layout) */
    void* obs_planes;
    int32_t* obs_agent_attrs;
    int32_t* obs_env_attrs;
    int32_t obs_dtype, obs_per_agent;
#if defined(POM_DIAG)
    long long* diag; /* POM_PH_N accumulators per wavefront, diagnostic build only */
#endif
#if defined(POM_TRUNC)
    int32_t trunc; /* the tick stops after this phase (POM_CUT in pom_step_body.h); environment POM_TRUNC_AT, default: all of it */
#endif
};

/*
 * One env's start board (pom_boardgen.h), drawn by the whole wavefront into the column of env number `el` of an
 * [row][EPW] tile with ROWS rows.  `key` must be wave-uniform.  Lane l draws cells l and l+64; the two ballots of "wood" are
 * the wood set, so the flag pass runs on uniform values (scalar unit) and only its few cell writes touch a lane.
 */
template <int EPW, int ROWS>
__device__ __forceinline__ void pom_boardgen_wave(uint32_t* tile, int el, uint32_t key, int lane)
{
    uint32_t* col = tile + el;
    uint8_t* cells = reinterpret_cast<uint8_t*>(tile);
    const uint32_t k0 = pom_board_cell_kind(key, lane);
    cells[tile_cell_byte<EPW>(el, lane)] = (uint8_t)pom_board_cell_code(k0);
    const int c1 = lane + 64; /* cells 64..123: the three behind the board are 0, as pom_pack_state writes them */
    uint32_t k1 = 0u;
    if (c1 < 4 * POM_REC_BOARD_DWORDS) {
        k1 = c1 < POM_CELLS ? pom_board_cell_kind(key, c1) : 0u;
        cells[tile_cell_byte<EPW>(el, c1)] = (uint8_t)pom_board_cell_code(k1);
    }
    const uint64_t w0 = __ballot(k0 == 2u), w1 = __ballot(k1 == 2u);
    const int r = POM_REC_TIMESTEP + lane;
    if (r < ROWS) col[r * EPW] = pom_fresh_row(r);
    /* the flags (pom_boardgen_body.h, step 2).  Per lane, in parallel: how many woods are still to come at each of my cells
     * (prefix counts of the ballots) and the selection threshold that follows from it.  Then the countdown, wave-uniform:
     * one readlane and a compare per wood.  The chosen cells come back as two masks and are rewritten by their lanes. */
    const int woods = __popcll(w0) + __popcll(w1);
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane)); /* the lanes before mine */
    const int t0 = (int)pom_board_threshold(key, lane, woods - __popcll(w0 & below));
    const int t1 = (int)pom_board_threshold(key, c1 < POM_CELLS ? c1 : 0, woods - __popcll(w0) - __popcll(w1 & below));
    uint64_t ch0 = 0, ch1 = 0;
    {
        int need = (woods + 1) >> 1;
        uint64_t w = w0;
        POM_NOUNROLL
        while (w != 0 && need > 0) {
            const int l = __builtin_ctzll(w);
            w &= w - 1;
            if (__builtin_amdgcn_readlane(t0, l) < need) {
                ch0 |= 1ull << l;
                need--;
            }
        }
        w = w1;
        POM_NOUNROLL
        while (w != 0 && need > 0) {
            const int l = __builtin_ctzll(w);
            w &= w - 1;
            if (__builtin_amdgcn_readlane(t1, l) < need) {
                ch1 |= 1ull << l;
                need--;
            }
        }
    }
    if ((ch0 >> lane) & 1) cells[tile_cell_byte<EPW>(el, lane)] = (uint8_t)pom_board_flag_code(key, lane);
    if ((ch1 >> lane) & 1) cells[tile_cell_byte<EPW>(el, c1)] = (uint8_t)pom_board_flag_code(key, c1);
    if (lane < POM_AGENT_COUNT) cells[tile_cell_byte<EPW>(el, pom_corner_cell(lane))] = (uint8_t)(POM_C_AGENT + lane);
}

/*
 * HBM -> LDS without touching VGPRs: `global_load_lds_dword` takes a per-lane global address and writes LDS
 * at (wave-uniform base) + 4*lane.  With the [row][EPW] tile one instruction therefore lands 64/EPW
 * consecutive rows (lane l -> row r0 + l/EPW, env l%EPW); all rows are in flight at once, no ds_write.
 * `col` may differ per lane: state column or snapshot column of the lane's env.
 */
__device__ __forceinline__ void dma_rows(const uint32_t* g, uint32_t* lds_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}
template <int EPW>
__device__ __forceinline__ void load_tile(const uint32_t* col, int64_t np /* row stride of a column */, uint32_t* tile, int sub)
{
    constexpr int GR = 64 / EPW; /* rows per instruction */
    /* a rolled loop with a running pointer: fully unrolled, hipcc precomputes all per-lane 64-bit addresses first
     * (200+ VGPRs) */
    const uint32_t* g = col + (int64_t)sub * np;
    const int64_t stride = (int64_t)GR * np;
#pragma unroll 4
    for (int r0 = 0; r0 < POM_REC_DWORDS; r0 += GR) {
        dma_rows(g, tile + r0 * EPW);
        g += stride;
    }
}
template <int EPW>
__device__ __forceinline__ void store_tile(uint32_t* col, int64_t np, const uint32_t* tile, int sub, int el)
{
    constexpr int GR = 64 / EPW;
    uint32_t* g = col + (int64_t)sub * np;
    const int64_t stride = (int64_t)GR * np;
    const uint32_t* l = tile + sub * EPW + el;
#pragma unroll 8
    for (int r0 = 0; r0 < POM_REC_DWORDS; r0 += GR) {
        *g = l[r0 * EPW];
        g += stride;
    }
}
/* EPW = 16: the same movement in 16-byte pieces (gfx950's global_load_lds_dwordx4 / dwordx4 stores).  A tile row is 16 envs
 * = 64 contiguous bytes in HBM and in LDS, so lane l takes envs 4(l%4)..+3 of row r0 + l/4 and one instruction covers 16 rows:
 * for the 80 rows of a record five whole instructions per direction (a row count that is no multiple of 16 ends with one in which
 * only the lanes of the remaining rows take part).  `base` = the tile's first dword, np = the row stride: with the buffers laid out tile by tile
 * (pom_packed.h) np = 16 and a whole instruction moves 1,024 contiguous bytes. */
/* AUX: the instruction's cache policy bits (16 = sc1: served by the L2, never by this CU's vector cache) */
template <int ROWS = POM_REC_DWORDS, int AUX = 0>
__device__ __forceinline__ void load_tile16_x4(const uint32_t* base, int64_t np, uint32_t* tile, int lane)
{
    static_assert(ROWS <= POM_REC_DWORDS, "inside the record");
    const uint32_t* g = base + (int64_t)(lane >> 2) * np + 4 * (lane & 3);
    const int64_t stride = 16 * np;
#pragma unroll
    for (int r0 = 0; r0 + 16 <= ROWS; r0 += 16) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(tile + r0 * 16), 16, 0, AUX);
        g += stride;
    }
    if (ROWS % 16 != 0 && (lane >> 2) < ROWS % 16) /* the last rows: the lanes beyond them sit this one out */
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(tile + (ROWS / 16) * 16 * 16), 16, 0, AUX);
}
template <bool NT = false>
__device__ __forceinline__ void store_tile16_x4(uint32_t* base, int64_t np, const uint32_t* tile, int lane)
{
    constexpr int ROWS = POM_REC_DWORDS;
    uint4* g = reinterpret_cast<uint4*>(base + (int64_t)(lane >> 2) * np + 4 * (lane & 3));
    const int64_t stride = 4 * np; /* in uint4 */
    const uint4* l = reinterpret_cast<const uint4*>(tile) + lane;
    typedef uint32_t pom_u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int r0 = 0; r0 + 16 <= ROWS; r0 += 16) {
        if (NT) {
            const uint4 v = l[r0 * 4];
            __builtin_nontemporal_store(pom_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<pom_u32x4*>(g));
        } else *g = l[r0 * 4];
        g += stride;
    }
    if (ROWS % 16 != 0 && (lane >> 2) < ROWS % 16) {
        if (NT) {
            const uint4 v = l[(ROWS / 16) * 16 * 4];
            __builtin_nontemporal_store(pom_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<pom_u32x4*>(g));
        } else *g = l[(ROWS / 16) * 16 * 4];
    }
}
/*
 * The restart snapshot is kept array-of-structs: env e's record is the 328 contiguous bytes snap[e * 82 .. e * 82 + 81].
 * A restart needs ONE env's whole record, and in the column layout of the state buffer that is 82 dwords in 82
 * different 64-byte sectors (4 useful bytes each: 18 MB of HBM fetch per step at 65,536 envs for the 3.8 % of envs that
 * restart, a fifth of the kernel's whole traffic; profiles/r02a_head1_summary.txt).  Here the whole wavefront fetches the
 * record of one restarting env — lane l takes dwords l and l + 64, whole sectors — and writes it into that env's
 * tile column.  `src` = &snap[e * POM_REC_DWORDS]; ROWS = how many leading rows the caller needs.
 */
template <int EPW, int ROWS = POM_REC_DWORDS>
__device__ __forceinline__ void restart_column(uint32_t* tile, int el, const uint32_t* src, int lane)
{
    const uint32_t v0 = src[lane];
    const uint32_t v1 = lane + 64 < ROWS ? src[lane + 64] : 0u;
    if (lane < POM_REC_BOARD_DWORDS) { /* a dense record holds four cells per dword, the tile lays the board out by cell */
        uint8_t* cells = reinterpret_cast<uint8_t*>(tile);
#pragma unroll
        for (int j = 0; j < 4; j++) cells[tile_cell_byte<EPW>(el, 4 * lane + j)] = (uint8_t)(v0 >> (8 * j));
    } else tile[lane * EPW + el] = v0;
    if (lane + 64 < ROWS) tile[(lane + 64) * EPW + el] = v1;
}
/* the other way: env number el's column of the tile as a dense record (terminal buffer, snapshot) */
template <int EPW>
__device__ __forceinline__ void column_to_record(const uint32_t* tile, int el, uint32_t* dst, int lane)
{
    if (lane < POM_REC_BOARD_DWORDS) {
        const uint8_t* cells = reinterpret_cast<const uint8_t*>(tile);
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) v |= (uint32_t)cells[tile_cell_byte<EPW>(el, 4 * lane + j)] << (8 * j);
        dst[lane] = v;
    } else dst[lane] = tile[lane * EPW + el];
    if (lane + 64 < POM_REC_DWORDS) dst[lane + 64] = tile[(lane + 64) * EPW + el];
}

/* the register-resident rows (timeStep, meta, agents) of one env, out of / into its tile column */
__device__ __forceinline__ void lane_from_tile(PomLane& L, int& time_step, uint32_t& status, const uint32_t* t, int epw)
{
    uint32_t ag[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ag[k] = t[(POM_REC_AGENTS + k) * epw];
    pom_lane_load(L, ag, status);
    time_step = (int)t[POM_REC_TIMESTEP * epw];
}

/* the four lanes of an env as the SimpleAgent policy sees them (pom_policy_body.h): lane = agent, 16 envs per wavefront */
struct PolicyStore {
    const uint32_t* tile0; /* the wavefront's tile */
    const uint32_t* t; /* &tile[env_in_wave], row stride 16 dwords */
    uint8_t* dcol;     /* &danger[env_in_wave], a byte per cell, row stride 16 */
    uint32_t* scol;    /* &sets[env_in_wave], row stride 16 */
    int who;           /* lane % 4 */
#if defined(POM_TRUNC)
    int trunc = 990;
#endif
    __device__ int member() const { return who; }
    __device__ int danger(int c) const { return dcol[c * 16]; }
    __device__ void danger_init(int c) { dcol[c * 16] = (uint8_t)POM_DANGER_NONE; }
    __device__ void danger_put(int c, int tm) { dcol[c * 16] = (uint8_t)tm; }
    __device__ uint32_t setw(int k) const { return scol[k * 16]; }
    __device__ void set_put(int k, uint32_t bits) { scol[k * 16] = bits; }
    /* (t = tile + env: the env's cell c is the tile's byte c * 16 + env, pom_packed.h) */
    __device__ int cell(int c) const { return reinterpret_cast<const uint8_t*>(t)[c * 16 - 3 * (int)(t - tile0)]; }
    __device__ uint32_t cells4(int k) const /* the codes of cells 4k .. 4k+3, a byte each */
    {
        const uint8_t* b = reinterpret_cast<const uint8_t*>(tile0) + (t - tile0) + 64 * k;
        return (uint32_t)b[0] | ((uint32_t)b[16] << 8) | ((uint32_t)b[32] << 16) | ((uint32_t)b[48] << 24);
    }
    __device__ int bomb(int s) const { return (int)t[(POM_REC_BOMBS + s) * 16]; }
};

/* ---------------------------------------------------------------------------------------------
 * SimpleAgent's two searches, run for all agents of a wavefront TOGETHER (round 3).  With a lane per agent and the 121-bit sets
 * in four registers of that lane, a wavefront pays its LONGEST flood (23.5 + 15.9 levels where the average act() runs 2.9 + 2.0,
 * tests/emul/flood_levels.sh) at ~70 VALU per level, while 80 % (forward) / 60 % (backward) of its lanes have no flood to run.
 * Here the wavefront's flood jobs — whoever's they are — are dealt to its 16 quads, 16 at a time: job i of a round goes to quad i,
 * whose four lanes hold the job's sets one 32-cell word each (pom_quad_*_level, pom_policy_body.h: two DPP word exchanges and a
 * dozen logic ops per level).  A round lasts as long as its longest job, at a third of the price per level.  No LDS memory is
 * used for the hand-over (the fused kernel has none to spare): a job's inputs are pulled from its owner lane and its answer is
 * pulled back by the owner with ds_bpermute; which lane owns job i comes from one ds_permute of the lanes' ranks.
 * ------------------------------------------------------------------------------------------- */
struct PomQuadLanes {
    typedef uint32_t W;
    int k;
    uint32_t c0_, c10_, valid_, not_first_, not_last_;
    __device__ explicit PomQuadLanes(int k_) : k(k_)
    {
        const uint32_t a[4] = {0x00400801u, 0x00801002u, 0x01002004u, 0x00004008u};
        const uint32_t b[4] = {0x00200400u, 0x00400801u, 0x00801002u, 0x01002004u};
        c0_ = pick4(k, a);
        c10_ = pick4(k, b);
        valid_ = k == 3 ? 0x01FFFFFFu : ~0u;
        not_first_ = k == 0 ? 0u : ~0u;
        not_last_ = k == 3 ? 0u : ~0u;
    }
    __device__ static int qor(int v)
    {
        v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
        return v | __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    }
    /* quad_perm [0,0,1,2]: lane j reads lane j-1; [1,2,3,3]: lane j reads lane j+1 */
    __device__ W prev(W w) const { return (W)__builtin_amdgcn_update_dpp(0, (int)w, 0x90, 0xF, 0xF, true) & not_first_; }
    __device__ W next(W w) const { return (W)__builtin_amdgcn_update_dpp(0, (int)w, 0xF9, 0xF, 0xF, true) & not_last_; }
    __device__ bool any(W w) const { return qor((int)w) != 0; }
    __device__ W bit(int c) const { return (c >> 5) == k ? 1u << (c & 31) : 0u; }
    __device__ W col0() const { return c0_; }
    __device__ W col10() const { return c10_; }
    __device__ W valid() const { return valid_; }
    __device__ int lowest(W w) const
    {
        int r = w ? 32 * k + __builtin_ctz(w) : 999;
        const int a = __builtin_amdgcn_update_dpp(0, r, 0xB1, 0xF, 0xF, true);
        r = a < r ? a : r;
        const int b = __builtin_amdgcn_update_dpp(0, r, 0x4E, 0xF, 0xF, true);
        return b < r ? b : r;
    }
    __device__ int gates_hit(W hit, uint32_t g8) const
    {
        int m = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int g = (int)((g8 >> (8 * j)) & 0xFF); /* 0xFF (no gate): word 7, nobody's */
            m |= (int)(((g >> 5) == k) & ((hit >> (g & 31)) & 1u)) << j;
        }
        return qor(m);
    }
};

/* which lane owns job i (for i < the number of jobs), in lane i: every lane sends its id to a slot of its own — the job lanes to
 * their rank, the others behind them — one ds_permute */
__device__ __forceinline__ int pom_job_owners(uint64_t jobs, bool need, int lane, int& rank, int& njobs)
{
    njobs = __popcll(jobs);
    rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(jobs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)jobs, 0u));
    const int slot = need ? rank : njobs + (lane - rank);
    return __builtin_amdgcn_ds_permute(slot << 2, lane);
}

/* FillRMap's reach + MoveTowardsSafePlace's choice for every lane with `need`: the lowest cell of (window & reachable & safe) or
 * -1.  src: the agent's cell; radius: of the scan window (the danger value; the worker lanes build the window word by word,
 * pom_window_word); sets: the wavefront's prepared cell sets, rows [word][16 envs] (0..3 walkable, 4..7 agents, 8..11 safe).
 * Every lane of the wavefront calls this. */
__device__ __forceinline__ int pom_coop_forward(bool need, int src, int radius, const uint32_t* sets, int lane)
{
    const uint64_t jobs = __ballot(need);
    int result = -1;
    if (jobs == 0) return result;
    int rank, njobs;
    const int owner_tab = pom_job_owners(jobs, need, lane, rank, njobs);
    const int k = lane & 3, q = lane >> 2;
    const PomQuadLanes Q(k);
    POM_NOUNROLL
    for (int base = 0; base < njobs; base += 16) {
        const int j = base + q;
        const bool has = j < njobs;
        const int ow = __builtin_amdgcn_ds_bpermute((has ? j : lane) << 2, owner_tab);
        const int owner = has ? ow : lane;
        const int desc = __builtin_amdgcn_ds_bpermute(owner << 2, src | (radius << 8));
        const int jsrc = desc & 0xFF, jrad = desc >> 8;
        const int env = owner >> 2;
        const uint32_t me = ~Q.bit(jsrc); /* FillRMap never re-enters its source */
        const uint32_t walk = sets[k * 16 + env] & me, agents = sets[(4 + k) * 16 + env] & me, safe = sets[(8 + k) * 16 + env];
        uint32_t front = has ? Q.bit(jsrc) : 0u, all = 0u;
        bool live = has;
        POM_NOUNROLL
        while (__any(live)) {
            if (live) live = pom_quad_forward_level(Q, walk, agents, front, all);
        }
        const int jy = div11(jsrc);
        const int low = Q.lowest(pom_window_word(k, jsrc - jy * POM_N, jy, jrad) & all & safe);
        const int got = __builtin_amdgcn_ds_bpermute((rank & 15) << 4, low); /* lane 0 of the quad that worked on my job */
        if (need && (rank >> 4) == (base >> 4)) result = got == 999 ? -1 : got;
    }
    return result;
}

/* MoveTowardsPosition's backward flood for every lane with `need`: which of its gates (g8: a cell per byte, DOWN UP RIGHT LEFT of
 * the agent's cell, 0xFF = none) the first level that reaches any of them reaches, as a 4-bit mask (0: the target is out of
 * reach).  Every lane of the wavefront calls this. */
__device__ __forceinline__ int pom_coop_backward(bool need, int src, int target, uint32_t g8, const uint32_t* sets, int lane)
{
    const uint64_t jobs = __ballot(need);
    int result = 0;
    if (jobs == 0) return result;
    int rank, njobs;
    const int owner_tab = pom_job_owners(jobs, need, lane, rank, njobs);
    const int k = lane & 3, q = lane >> 2;
    const PomQuadLanes Q(k);
    POM_NOUNROLL
    for (int base = 0; base < njobs; base += 16) {
        const int j = base + q;
        const bool has = j < njobs;
        const int ow = __builtin_amdgcn_ds_bpermute((has ? j : lane) << 2, owner_tab);
        const int owner = has ? ow : lane;
        const int jsrc = __builtin_amdgcn_ds_bpermute(owner << 2, src);
        const int jt = __builtin_amdgcn_ds_bpermute(owner << 2, target);
        const uint32_t jg = (uint32_t)__builtin_amdgcn_ds_bpermute(owner << 2, (int)g8);
        const int env = owner >> 2;
        const uint32_t walk = sets[k * 16 + env] & ~Q.bit(jsrc);
        const uint32_t gates = Q.bit((int)(jg & 0xFF)) | Q.bit((int)((jg >> 8) & 0xFF)) | Q.bit((int)((jg >> 16) & 0xFF)) | Q.bit((int)(jg >> 24));
        uint32_t front = has ? Q.bit(jt) : 0u, seen = front;
        int r = has ? -1 : 0;
        POM_NOUNROLL
        while (__any(r < 0)) {
            if (r < 0) r = pom_quad_backward_level(Q, walk, gates, jg, front, seen);
        }
        const int got = __builtin_amdgcn_ds_bpermute((rank & 15) << 4, r);
        if (need && (rank >> 4) == (base >> 4)) result = got;
    }
    return result;
}

/* SimpleAgent::act for the 64 agents of a wavefront (lane = agent, quad = env): the per-agent pieces of _Decide
 * (pom_policy_body.h) with the two searches run cooperatively in between.  actor: this lane's agent is asked for a move (a live
 * agent of an env that plays this tick).  Every lane of the wavefront calls this; returns the lane's Move (IDLE if not an actor). */
template <class Store>
__device__ __forceinline__ int pom_policy_wave(Store& st, const PomPolicyEnv& E, int member, uint32_t& m0, uint32_t& m1, bool actor, int draw,
                                               const uint32_t* sets, int lane
#if defined(POM_DIAG)
                                               , long long& t_last, long long* t_acc
#endif
)
{
    PomSimplePolicy<Store> pol(st, E, member, m0, m1);
#if defined(POM_DIAG)
    pol.t_last = t_last;
    for (int k = 0; k < POM_PP_N; k++) pol.t_acc[k] = 0;
#endif
    int in_danger = 0;
    if (actor) in_danger = pol.begin();
    int mv = POM_MOVE_IDLE;
#if defined(POM_TRUNC)
    if (st.trunc <= 1) return actor ? (pol.danger_ | pol.adj1_ | pol.near_ | pol.looping_ | pol.can_bomb_) & 1 : mv; /* diagnostic build: keep the predicates alive */
#endif
    __builtin_amdgcn_wave_barrier(); /* the sets of all 16 envs were written by their own lanes: no read of them may be scheduled earlier */
    const int safe_cell = pom_coop_forward(actor && in_danger, pol.src_cell(), pol.danger_, sets, lane);
    int target = -1, found = POM_MOVE_IDLE, flood = 0;
    uint32_t g8 = 0xFFFFFFFFu;
    if (actor) {
        target = pol.pick_target(safe_cell);
#if defined(POM_DIAG)
        { const long long now_ = (long long)clock64(); pol.t_acc[POM_PP_TARGET] += now_ - pol.t_last; pol.t_last = now_; }
#endif
        flood = pol.path_begin(target, found, g8);
    }
#if defined(POM_TRUNC)
    if (st.trunc <= 2) return target & 1;
#endif
    const int hit = pom_coop_backward(actor && flood, pol.src_cell(), target, g8, sets, lane);
    if (actor) {
        const int step = pol.path_end(target, found, hit);
#if defined(POM_TRUNC)
        if (st.trunc <= 3) return step;
#endif
        mv = pol.finish(step, draw);
        m0 = pol.m0;
        m1 = pol.m1;
    }
#if defined(POM_DIAG)
    t_last = pol.t_last;
    for (int k = POM_PP_PREDICATES; k <= POM_PP_TAIL; k++) t_acc[k] = pol.t_acc[k];
#endif
    return mv;
}

/* ------------------------------------------------------------------------------------------------
 * Observation export (row f4, pom_batch.h pom_batch_observe).  A wavefront takes a tile of 16 envs into LDS as the tick does,
 * then env by env: zero a 1936-byte staging area, scatter one byte per cell / bomb (each cell sets exactly one
 * of the planes 0..11), and stream the area out — for uint8 global views a straight 16-byte copy, fully coalesced; other
 * element types and the per-agent plane order take runs of 4 bytes through a byte funnel, convert and store 4 elements.  HBM-write-bound: 1936 B x elements per env.
 * ------------------------------------------------------------------------------------------- */
/* Envs staged at a time for the 16 planes (template parameter PE below): the export kernel stages two (9 KB of LDS per wavefront, 18 per
 * CU; 992 + 544 + 279 instructions per wavefront against 1,248 + 738 + 385 with one, alone 23.7 -> 23.1 us: a pass's fixed work — the
 * queue slots' phases keep 20 lanes busy per env — is paid half as often), the step kernel's fused export one (two cost it 0.5 us:
 * it sits at its register and LDS limits); four needed 15 KB and ran 25 % slower. */
enum { OBS_ENV_BYTES = POM_OBS_PLANES * POM_CELLS, OBS_PASS_ENVS_ALONE = 2, OBS_PASS_ENVS_FUSED = 1 };
static_assert(OBS_ENV_BYTES % 16 == 0, "an env's planes are a whole number of 16-byte stores");

struct ObserveParams {
    const uint32_t* state;
    int64_t n, n_pad, block0;
    void* planes;
    int32_t* agent_attrs;
    int32_t* env_attrs;
    int32_t dtype, per_agent;
};

/* four staged bytes starting at byte offset `b` of the staging area (any alignment): two aligned dwords and a byte funnel */
__device__ __forceinline__ uint32_t obs_bytes4(const uint32_t* stage_w, int b)
{
    const uint32_t lo = stage_w[b >> 2], hi = stage_w[(b >> 2) + 1];
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(b & 3));
}
template <class T>
__device__ __forceinline__ void obs_store4(T* dst, uint32_t bytes);
template <>
__device__ __forceinline__ void obs_store4<uint8_t>(uint8_t* dst, uint32_t bytes) { *reinterpret_cast<uint32_t*>(dst) = bytes; }
template <>
__device__ __forceinline__ void obs_store4<_Float16>(_Float16* dst, uint32_t bytes)
{
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<half4*>(dst) = half4{(_Float16)(float)(bytes & 0xFF), (_Float16)(float)((bytes >> 8) & 0xFF),
                                           (_Float16)(float)((bytes >> 16) & 0xFF), (_Float16)(float)(bytes >> 24)};
}
template <>
__device__ __forceinline__ void obs_store4<float>(float* dst, uint32_t bytes)
{
    *reinterpret_cast<float4*>(dst) = make_float4((float)(bytes & 0xFF), (float)((bytes >> 8) & 0xFF), (float)((bytes >> 16) & 0xFF),
                                                  (float)(bytes >> 24));
}

/* the generic way out of the staging area: element type T, `views` plane orders per env.  A lane takes 4 consecutive output
 * elements per round (one aligned 4 / 8 / 16-byte store): their source bytes are consecutive in the staging area unless the
 * group crosses a plane boundary (planes are 121 bytes, and the four agent planes are permuted per view), so it fetches the
 * run starting at its first element and the run ending at its last one and splices them at the boundary.  Plane and offset
 * advance incrementally (64 lanes x 4 elements = 2 planes + 14 per round): no division in the loop. */
template <class T, int PE>
__device__ __forceinline__ void obs_gather_out(const ObserveParams& p, const uint32_t* stage_w, int64_t e0, int lane)
{
    const int views = p.per_agent ? 4 : 1;
    T* out = reinterpret_cast<T*>(p.planes);
    for (int ei = 0; ei < PE && e0 + ei < p.n; ei++) {
        for (int a = 0; a < views; a++) {
            T* dst = out + ((e0 + ei) * views + a) * (int64_t)OBS_ENV_BYTES;
            int pl = (4 * lane) / POM_CELLS, off = 4 * lane - pl * POM_CELLS; /* of the group's first element */
            POM_NOUNROLL
            for (int el = 4 * lane; el < OBS_ENV_BYTES; el += 256) {
                const int src0 = (pl >= 8 && pl < 12) ? 8 + ((pl - 8 + a) & 3) : pl;
                const int pn = pl + 1, src1 = (pn >= 8 && pn < 12) ? 8 + ((pn - 8 + a) & 3) : pn;
                const int room = POM_CELLS - off; /* elements left in this plane, >= 1 */
                const uint32_t head = obs_bytes4(stage_w, ei * OBS_ENV_BYTES + src0 * POM_CELLS + off);
                /* the next plane's first bytes, placed where they belong in the group (only read when the group crosses) */
                const uint32_t tail = obs_bytes4(stage_w, ei * OBS_ENV_BYTES + (room < 4 ? src1 * POM_CELLS - room : src0 * POM_CELLS + off));
                const uint32_t keep = room >= 4 ? 0xFFFFFFFFu : (1u << (8 * room)) - 1u;
                obs_store4<T>(dst + el, (head & keep) | (tail & ~keep));
                off += 256 - 2 * POM_CELLS; /* 256 = 2 x 121 + 14 */
                pl += 2;
                if (off >= POM_CELLS) {
                    off -= POM_CELLS;
                    pl++;
                }
            }
        }
    }
}

/* the planes and attributes of one tile's 16 envs, from the tile as it lies in LDS ([row][16] dwords); `stage`: obs_stage_vecs(PE)
 * uint4 of LDS.  Called by the whole wavefront (one wavefront per workgroup: the barriers only order its own LDS traffic). */
enum {
    OBS_CODE_PLANES = 5,                              /* POM_OBS_CODES: board codes, bomb strength / life / direction, flame life */
    OBS_CODE_ENV_BYTES = OBS_CODE_PLANES * POM_CELLS, /* 605 */
    OBS_CODE_PASS_ENVS = 4,                           /* 4 x 605 B = 605 dwords: a pass's output starts on a dword */
    OBS_CODE_SHIFT_MAX = 12,                          /* a pass is staged 0 / 4 / 8 / 12 bytes into the area: where its output lies relative to a 16-byte line */
    OBS_CODE_STAGE_VECS = (OBS_CODE_PASS_ENVS * OBS_CODE_ENV_BYTES + OBS_CODE_SHIFT_MAX + 8 + 15) / 16,
};
/* uint4s of LDS the export needs with PE envs of planes per pass (+ 16 B: the byte funnel reads one dword past a run) */
constexpr int obs_stage_vecs(int pe)
{
    return pe * OBS_ENV_BYTES / 16 + 1 > OBS_CODE_STAGE_VECS ? pe * OBS_ENV_BYTES / 16 + 1 : (int)OBS_CODE_STAGE_VECS;
}
static_assert(POM_OBS_CODE_PLANES == OBS_CODE_PLANES, "pom_batch.h");

/* One pass of the export: E envs of the tile (q-th group of E) into the staging area — the 16 planes of an env (CODES = false,
 * E = 1) or the five code planes of four envs (CODES = true).  "The first live bomb on a cell speaks for it" and "the first live
 * flame spawned at a flame cell's FLAME_ID" (State::GetBomb's order, bboard.cpp:277-287; bboard.hpp:98-101) are answered without
 * searching the queues per cell: a lane per queue slot writes its key (offset in the queue + 1) to its cell in a plane of the
 * staging area that is not needed yet — the bombs into the strength plane, the flames into the bomb-direction plane — and lowers
 * it until the smallest key of a cell has won; a flame cell then reads its flame with one look-up, a bomb slot that finds its own
 * key is the first on its cell.  The flames' keys are wiped before the bombs write their three planes.  (Round 4; until then every
 * flame cell walked the flame queue — a wavefront paid its longest queue on nearly every group of 64 cells, three quarters of the
 * kernel's instructions.) */
/* Between the phases of the export: "the other lanes' LDS writes so far are visible from here on".  One wavefront per workgroup and
 * the LDS serves a wavefront's instructions in order, so the hardware has nothing to wait for; the compiler must not move LDS
 * accesses across.  (Until round 4 this was __syncthreads(), whose s_waitcnt vmcnt(0) made every pass wait for its own global
 * stores to be acknowledged: 16 passes x the store latency — three quarters of the wavefront's life, profiles/r04_observe_pmc.txt.) */
__device__ __forceinline__ void obs_lds_order() { asm volatile("" ::: "memory"); }

template <bool CODES, int PE>
__device__ __forceinline__ void pom_observe_stage(const uint32_t* tile, uint4* stage, int q, int lane, int shift = 0)
{
    constexpr int E = CODES ? (int)OBS_CODE_PASS_ENVS : PE, EB = CODES ? (int)OBS_CODE_ENV_BYTES : (int)OBS_ENV_BYTES;
    constexpr int P_STRENGTH = CODES ? 1 : 12, P_DIR = P_STRENGTH + 2, P_FLAME = P_STRENGTH + 3;
    constexpr int VECS = (E * EB + (CODES ? (int)OBS_CODE_SHIFT_MAX + 8 : 0) + 15) / 16, SLOT_IT = (E * POM_Q + 63) / 64;
    static_assert((E & (E - 1)) == 0, "lanes per env");
    static_assert(obs_stage_vecs(PE) * 16 >= E * EB + 8 + (CODES ? (int)OBS_CODE_SHIFT_MAX : 0), "eight bytes behind the planes take the writes of lanes that have nothing to write");
    /* The phases below are written without branches where a lane-varying `if` would do (hipcc makes exec-mask forests of those, and
     * the kernel is bound by the instructions it issues, scalar ones included): a lane with nothing to write writes to `dump`. */
    const int dump = E * EB + (lane & 7);
    uint8_t* stage_b = reinterpret_cast<uint8_t*>(stage) + shift;
#pragma unroll
    for (int i = 0; i < (VECS + 63) / 64; i++)
        if (64 * i + 63 < VECS || lane + 64 * i < VECS) stage[lane + 64 * i] = make_uint4(0, 0, 0, 0);
    obs_lds_order();
    /* the queue slots' keys.  Code planes: lane -> (env lane / 16, slot lane % 16) of the pass's four envs; the slots 16 .. 19 get a second
     * round only if some queue of the pass is that long (round 5; until then two full rounds over 80 slots every pass) */
    constexpr int SLOT_ROUNDS = CODES ? 2 : 1;
    static_assert(!CODES || (E == 4 && POM_Q == 20), "the slot rounds of the code planes");
    static_assert(CODES || SLOT_IT == 1, "an env's slots are one round");
    auto slot = [&](int i, int& ei, int& k, int& ok) {
        if (CODES) {
            ei = i == 0 ? lane >> 4 : (lane >> 2) & 3;
            k = i == 0 ? lane & 15 : 16 + (lane & 3);
            ok = i == 0 ? 1 : (int)(lane < 16);
        } else {
            ei = E == 1 ? 0 : (lane * 3277) >> 16; /* lane / 20 */
            k = lane - ei * POM_Q;
            ok = (int)(lane < E * POM_Q);
            ei = ok ? ei : 0;
        }
    };
    int key_f[SLOT_ROUNDS], key_b[SLOT_ROUNDS], bomb[SLOT_ROUNDS]; /* where a live slot's key goes (byte offset in the staging area; `dump`: not live) */
    bool tail = false; /* the second round is needed */
#pragma unroll
    for (int i = 0; i < SLOT_ROUNDS; i++) {
        key_f[i] = key_b[i] = dump;
        bomb[i] = 0;
        if (i == 0 || tail) {
            int ei, k, ok;
            slot(i, ei, k, ok);
            const int ec = q * E + ei, kk = ok ? k : 0;
            /* the queues' indices and counts: the top bytes of four agent words (pom_packed.h) */
            const int bIdx = (int)(tile[(POM_REC_AGENTS + 2) * 16 + ec] >> 24), bCnt = (int)(tile[(POM_REC_AGENTS + 4) * 16 + ec] >> 24),
                      fIdx = (int)(tile[(POM_REC_AGENTS + 6) * 16 + ec] >> 24), fCnt = (int)(tile[(POM_REC_AGENTS + 1) * 16 + ec] >> 24);
            if (CODES && i == 0) tail = __any((int)(bCnt > 16) | (int)(fCnt > 16)) != 0;
            const uint32_t f = tile[(POM_REC_FLAMES + wrap20(fIdx + kk)) * 16 + ec];
            const int b = (int)tile[(POM_REC_BOMBS + wrap20(bIdx + kk)) * 16 + ec];
            bomb[i] = b;
            const int fc = (int)(f & 0xFF) + POM_N * (int)((f >> 8) & 0xFF);
            key_f[i] = (ok & (int)(k < fCnt) & (int)(fc < POM_CELLS)) ? ei * EB + P_DIR * POM_CELLS + fc : dump;
            key_b[i] = (ok & (int)(k < bCnt) & (int)(pb_x(b) < POM_N) & (int)(pb_y(b) < POM_N)) ? ei * EB + P_STRENGTH * POM_CELLS + pb_y(b) * POM_N + pb_x(b) : dump;
            stage_b[key_f[i]] = (uint8_t)(k + 1);
            stage_b[key_b[i]] = (uint8_t)(k + 1);
        }
    }
    POM_NOUNROLL
    for (;;) { /* several slots on one cell: one of their writes landed — the smaller keys write again until the smallest stands */
        obs_lds_order();
        int low_f[SLOT_ROUNDS], low_b[SLOT_ROUNDS], again = 0;
#pragma unroll
        for (int i = 0; i < SLOT_ROUNDS; i++) {
            low_f[i] = low_b[i] = 0;
            if (i == 0 || tail) {
                int ei, k, ok;
                slot(i, ei, k, ok);
                low_f[i] = (int)(key_f[i] != dump) & (int)(stage_b[key_f[i]] > k + 1);
                low_b[i] = (int)(key_b[i] != dump) & (int)(stage_b[key_b[i]] > k + 1);
                again |= low_f[i] | low_b[i];
            }
        }
        if (!__any(again)) break; /* (the usual case: no two live slots on one cell) */
        for (int i = 0; i < SLOT_ROUNDS; i++) { /* (one or two rounds: unrolled without being asked) */
            if (i == 0 || tail) {
                int ei, k, ok;
                slot(i, ei, k, ok);
                stage_b[low_f[i] ? key_f[i] : dump] = (uint8_t)(k + 1);
                stage_b[low_b[i] ? key_b[i] : dump] = (uint8_t)(k + 1);
            }
        }
    }
    obs_lds_order();
    if constexpr (CODES) {
        /* The code planes, a pass = four envs: with the board laid out by cell the four envs' codes of cell c ARE one dword of the tile
         * (bytes 4q .. 4q + 3 of the cell's 16), so a lane takes a cell and turns four codes into four Item numbers at once (round 5;
         * until then a lane took two cells of one env, ~22 instructions per cell): the 16-entry table (0 passage, 1 rigid, 3 Item::BOMB,
         * 6 / 7 / 8 the power-ups, 2 wood with any flag, 10 + i agent i — the small numbers of the reference's Item enum, bboard.hpp:54-71)
         * is two byte permutes on the codes' low three bits, chosen between by bit 3; every code from 15 up is a flame (4). */
        static_assert(E == 4, "a pass of the code planes is the four envs that share a dword of every cell");
        const int q4 = 4 * q;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int c = lane + 64 * i;
            if (i == 0 || c < POM_CELLS) {
                const uint32_t d = tile[c * 4 + q];
                const uint32_t flame = (((d & 0x7F7F7F7Fu) + 0x71717171u) | d) & 0x80808080u; /* the bytes >= 15 */
                const uint32_t lo3 = d & 0x07070707u;
                const uint32_t t_lo = __builtin_amdgcn_perm(0x02020807u, 0x06030100u, lo3); /* codes 0 .. 7 */
                const uint32_t t_hi = __builtin_amdgcn_perm(0x040D0C0Bu, 0x0A020202u, lo3); /* codes 8 .. 14 */
                const uint32_t b3 = (d >> 3) & 0x01010101u, m3 = (b3 << 8) - b3;            /* 0xFF in the bytes with bit 3 */
                const uint32_t f1 = flame >> 7, mf = (f1 << 8) - f1;
                uint32_t v = (t_hi & m3) | (t_lo & ~m3);
                v = (0x04040404u & mf) | (v & ~mf);
                stage_b[c] = (uint8_t)v;
                stage_b[EB + c] = (uint8_t)(v >> 8);
                stage_b[2 * EB + c] = (uint8_t)(v >> 16);
                stage_b[3 * EB + c] = (uint8_t)(v >> 24);
                /* the flame cells among the four: the life of the first live flame spawned at the cell's FLAME_ID (key table above) */
                uint32_t todo = flame;
                POM_NOUNROLL
                while (todo) {
                    const int ei = __builtin_ctz(todo) >> 3, ec = q4 + ei;
                    todo &= todo - 1u;
                    const int code = (int)((d >> (8 * ei)) & 0xFFu);
                    int origin = code - POM_C_FLAME;
                    if (code >= POM_C_FLAGGED) origin = pom_flame_origin(code, c); /* (a burnt wood with a power-up under it: rare) */
                    const int key = stage_b[ei * EB + P_DIR * POM_CELLS + origin];
                    const int fIdx = (int)(tile[(POM_REC_AGENTS + 6) * 16 + ec] >> 24); /* flames.index: the top byte of agent 3's first word */
                    const int tl = pom_sext8(tile[(POM_REC_FLAMES + wrap20(fIdx + (key ? key - 1 : 0))) * 16 + ec] >> 16);
                    stage_b[ei * EB + P_FLAME * POM_CELLS + c] = (uint8_t)((int)(key != 0) & (int)(tl > 0) ? tl : 0);
                }
            }
        }
    } else
    /* cells: a byte each into the plane its code names (or its Item number into the board plane); flame cells look their flame up.  A lane
     * takes half a dword of the board — two cells — of one env per round: lane -> (env of the pass, unit of two cells), so that every
     * address is a base plus a constant */
    {
        constexpr int UNITS = (POM_CELLS + 1) / 2; /* 61 */
        constexpr int LPE = 64 / E, ROW_IT = (UNITS + LPE - 1) / LPE; /* lanes per env; rounds over the 61 two-cell units */
        const int ei = E == 1 ? 0 : lane & (E - 1), j = E == 1 ? lane : lane / E, ec = q * E + ei;
        const int fIdx = (int)(tile[(POM_REC_AGENTS + 6) * 16 + ec] >> 24); /* flames.index: the top byte of agent 3's first word */
        auto put = [&](int code, int c, int ok) { /* the cell's byte; returns whether the cell is a flame */
            /* What the cell's code (pom_packed.h: 0 passage, 1 rigid, 2 bomb, 3..5 power-ups, 6..10 wood, 11..14 agents, 15.. flames) sets.
             * The 16 planes: passage 0, rigid 1, wood (any flag) 2, Item::BOMB 3, flames 4, the three power-ups 5, 6, 7, agent i 8 + i.
             * POM_OBS_CODES: the small numbers of the reference's Item enum (bboard.hpp:54-71; the Python Pommerman board uses the same
             * ones) — 0 passage, 1 rigid, 2 wood, 3 bomb, 4 flames, 6 extra-bomb, 7 incr-range, 8 kick, 10 + i agent i.  One table, a
             * nibble per code, every flame code reading entry 15. */
            const int k = code < 15 ? code : 15;
            const int v = (int)(((CODES ? 0x4DCBA22222876310ull : 0x4BA9822222765310ull) >> (4 * k)) & 15u);
            if (CODES) stage_b[ok ? ei * EB + c : dump] = (uint8_t)v;
            else stage_b[ok ? ei * EB + v * POM_CELLS + c : dump] = 1;
            return ok & pc_is_flame(code);
        };
        auto life = [&](int code, int c, int is_flame) { /* unconditional: a cell that is no flame looks at its own (zero) byte */
            int origin = code - POM_C_FLAME; /* FLAME_ID: the cell the flame was spawned at */
            if (code >= POM_C_FLAGGED) origin = pom_flame_origin(code, c); /* (a burnt wood with a power-up under it: rare) */
            const int key = stage_b[is_flame ? ei * EB + P_DIR * POM_CELLS + origin : ei * EB + P_FLAME * POM_CELLS + c];
            const int tl = pom_sext8(tile[(POM_REC_FLAMES + wrap20(fIdx + (key ? key - 1 : 0))) * 16 + ec] >> 16);
            stage_b[is_flame ? ei * EB + P_FLAME * POM_CELLS + c : dump] = (uint8_t)((int)(key != 0) & (int)(tl > 0) ? tl : 0);
        };
#pragma unroll
        for (int i = 0; i < ROW_IT; i++) {
            const int r = j + LPE * i;
            const int ok = (LPE * i + LPE - 1 < UNITS) | (int)(r < UNITS); /* (a lane past the board reads some other row of the tile) */
            const uint8_t* tile_b = reinterpret_cast<const uint8_t*>(tile); /* the board by cell: cell c of env ec at byte c * 16 + ec */
            const int lo = tile_b[(2 * r) * 16 + ec], hi = tile_b[(2 * r + 1) * 16 + ec];
            const int f0 = put(lo, 2 * r, ok);
            const int f1 = put(hi, 2 * r + 1, ok & (int)(r < UNITS - 1)); /* the board's last unit holds one cell */
            if (f0 | f1) {
                life(lo, 2 * r, f0);
                life(hi, 2 * r + 1, f1);
            }
        }
    }
    int first[SLOT_ROUNDS];
#pragma unroll
    for (int i = 0; i < SLOT_ROUNDS; i++) {
        first[i] = 0;
        if (i == 0 || tail) {
            int ei, k, ok;
            slot(i, ei, k, ok);
            first[i] = (int)(key_b[i] != dump) & (int)(stage_b[key_b[i]] == k + 1);
        }
    }
    obs_lds_order();
#pragma unroll
    for (int i = 0; i < SLOT_ROUNDS; i++)
        if (i == 0 || tail) stage_b[key_f[i]] = 0; /* the direction plane is the bombs' again */
    obs_lds_order();
#pragma unroll
    for (int i = 0; i < SLOT_ROUNDS; i++) {
        if (i == 0 || tail) {
            stage_b[first[i] ? key_b[i] : dump] = (uint8_t)pb_strength(bomb[i]);
            stage_b[first[i] ? key_b[i] + 1 * POM_CELLS : dump] = (uint8_t)pb_time(bomb[i]);
            stage_b[first[i] ? key_b[i] + 2 * POM_CELLS : dump] = (uint8_t)pb_dir(bomb[i]);
        }
    }
    obs_lds_order();
}

/* The compact layout (POM_OBS_CODES): uint8 [n][5][11][11] — 605 bytes per env instead of 1,936, for training loops that expand
 * the board codes themselves (an embedding look-up).  Four envs are staged at a time (2,420 B, a whole number of dwords) and leave
 * as dword stores; the batch's last bytes, where n is no multiple of 4, leave byte by byte. */
__device__ __forceinline__ void pom_observe_tile_codes(const ObserveParams& p, const uint32_t* tile, uint4* stage, int64_t tile_id, int lane)
{
    /* A tile's 16 x 605 bytes start on a 16-byte line if the array does, and pass q's 2,420 bytes 4 q bytes into one: staged that many
     * bytes into the area, the pass leaves as 16-byte stores of aligned LDS reads (round 5; until then 10 rounds of dword stores), the
     * up to three dwords before the first and after the last whole line one by one */
    const bool lines = ((reinterpret_cast<uintptr_t>(p.planes) + (uintptr_t)(tile_id * 16 * OBS_CODE_ENV_BYTES)) & 15u) == 0u;
    for (int q = 0; q < 16 / OBS_CODE_PASS_ENVS; q++) {
        const int64_t e0 = tile_id * 16 + q * OBS_CODE_PASS_ENVS;
        if (e0 >= p.n) break;
        const int64_t left = p.n - e0;
        const bool fast = lines && left >= OBS_CODE_PASS_ENVS;
        const int shift = fast ? 4 * q : 0;
        pom_observe_stage<true, 1>(tile, stage, q, lane, shift);
        uint8_t* out_b = reinterpret_cast<uint8_t*>(p.planes) + e0 * OBS_CODE_ENV_BYTES; /* e0 is a multiple of 4: on a dword */
        const uint32_t* stage_w = reinterpret_cast<const uint32_t*>(stage);
        if (fast) {
            constexpr int DW = OBS_CODE_PASS_ENVS * OBS_CODE_ENV_BYTES / 4; /* 605 */
            uint4* out_v = reinterpret_cast<uint4*>(out_b - shift);
            uint32_t* out_w = reinterpret_cast<uint32_t*>(out_b - shift);
            const int beg = q, end = q + DW, v0 = (beg + 3) >> 2, v1 = end >> 2; /* dwords of the area; whole lines [v0, v1) */
#pragma unroll
            for (int i = 0; i < (DW / 4 + 1 + 63) / 64; i++) {
                const int v = lane + 64 * i;
                if (v >= v0 && v < v1) out_v[v] = stage[v];
            }
            const int dw = lane < 4 ? lane : 4 * v1 + lane - 4;
            if (lane < 4 ? (dw >= beg && dw < 4 * v0) : (lane < 8 && dw < end)) out_w[dw] = stage_w[dw];
        } else {
            const uint8_t* stage_b = reinterpret_cast<const uint8_t*>(stage);
            const int bytes = (int)(left < OBS_CODE_PASS_ENVS ? left : OBS_CODE_PASS_ENVS) * OBS_CODE_ENV_BYTES;
            uint32_t* out_w = reinterpret_cast<uint32_t*>(out_b);
#pragma unroll
            for (int i = 0; i < (OBS_CODE_PASS_ENVS * OBS_CODE_ENV_BYTES / 4 + 63) / 64; i++) {
                const int idx = lane + 64 * i;
                if (idx < (bytes >> 2)) out_w[idx] = stage_w[idx];
            }
            if (lane < (bytes & 3)) out_b[(bytes & ~3) + lane] = stage_b[(bytes & ~3) + lane];
        }
        obs_lds_order();
    }
}

template <int PE>
__device__ __forceinline__ void pom_observe_tile(const ObserveParams& p, const uint32_t* tile, uint4* stage, int64_t tile_id, int lane)
{
    static_assert(PE * POM_Q <= 64 && 16 % PE == 0, "a pass's queue slots are one round of lanes");
    if (p.dtype == POM_OBS_CODES) pom_observe_tile_codes(p, tile, stage, tile_id, lane);
    else
    for (int q = 0; q < 16 / PE; q++) {
        const int64_t e0 = tile_id * 16 + q * PE;
        if (e0 >= p.n) break;
        const int64_t left = p.n - e0;
        const int VECS = (int)(left < PE ? left : PE) * (OBS_ENV_BYTES / 16); /* 121 per env */
        constexpr int VECS_MAX = PE * OBS_ENV_BYTES / 16;
        pom_observe_stage<false, PE>(tile, stage, q, lane);
        if (p.dtype == POM_OBS_U8 && !p.per_agent) {
            uint4* out = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(p.planes) + e0 * OBS_ENV_BYTES);
#pragma unroll
            for (int i = 0; i < (VECS_MAX + 63) / 64; i++) {
                const int idx = lane + 64 * i;
                if (idx < VECS) out[idx] = stage[idx]; /* (non-temporal stores: 45 us against 40 fused) */
            }
        } else if (p.dtype == POM_OBS_U8) {
            obs_gather_out<uint8_t, PE>(p, reinterpret_cast<const uint32_t*>(stage), e0, lane);
        } else if (p.dtype == POM_OBS_F16) {
            obs_gather_out<_Float16, PE>(p, reinterpret_cast<const uint32_t*>(stage), e0, lane);
        } else {
            obs_gather_out<float, PE>(p, reinterpret_cast<const uint32_t*>(stage), e0, lane);
        }
        obs_lds_order();
    }
    /* attributes: lane -> (env lane/4, agent lane%4), 32 contiguous bytes each */
    const int ec = lane >> 2, id = lane & 3;
    const int64_t e = tile_id * 16 + ec;
    if (e < p.n && p.agent_attrs) {
        const uint32_t a0 = tile[(POM_REC_AGENTS + 2 * id) * 16 + ec], a1 = tile[(POM_REC_AGENTS + 2 * id + 1) * 16 + ec];
        const int bc = ag_bombcount((int)a0), mx = ag_max_bombs((int)a1);
        int4* o = reinterpret_cast<int4*>(p.agent_attrs + (e * 4 + id) * POM_OBS_AGENT_ATTRS);
        o[0] = make_int4(ag_x((int)a0), ag_y((int)a0), !ag_dead((int)a0), mx - bc);
        o[1] = make_int4(bc, mx, ag_strength((int)a1), ag_kick((int)a0));
    }
    if (lane < 16 && tile_id * 16 + lane < p.n && p.env_attrs) {
        const uint32_t m = pom_rec_meta(tile + lane, 16), st = (pom_rec_meta2(tile + lane, 16) >> 8) & 0xFF;
        const int status = (int)((st & POM_ST_DONE) ? 1 : 0) | (int)((st & POM_ST_DRAW) ? 2 : 0) | (int)((st & POM_ST_TIMEOUT) ? 4 : 0) |
                           (int)((st & POM_ST_RESTARTED) ? 8 : 0);
        reinterpret_cast<int4*>(p.env_attrs)[tile_id * 16 + lane] =
            make_int4((int)tile[POM_REC_TIMESTEP * 16 + lane], pom_sext8(m), status, (int)((st >> POM_ST_WINNER_SHIFT) & 7) - 1);
    }
}

/* occupancy target of the quad kernel: 4 wavefronts per SIMD = 16 per CU = every one of 65,536 envs' wavefronts resident at
 * once.  The kernel needs 120-128 VGPRs; the target keeps the compiler from drifting past 128 (which would drop a whole
 * round's worth of wavefronts to a second round) — at the price of a spill or two if it ever has to.  History: an early
 * 157-VGPR version capped at 128 spilled 60 B/lane and ran 12 % slower than uncapped (profiles/r01_quad.txt); since the tile
 * mirrors the whole record the kernel fits. */
#ifndef POM_QUAD_WAVES
#define POM_QUAD_WAVES 4
#endif
/* wavefronts per workgroup.  Every wavefront works on a tile of its own in its own slice of the workgroup's LDS and never
 * synchronises with the others; more than one per workgroup only means fewer workgroups for the dispatcher to place. */
#ifndef POM_WPB
#define POM_WPB 1
#endif
/* POLICY: the moves are not read but decided here — lane m of an env's quad is agent m and runs SimpleAgent::act
 * (pom_policy_body.h) on the tile the tick is about to work on: Environment::Step with four SimpleAgents in ONE kernel, one
 * record load per tick instead of two and no Move[4] round trip (pom_batch_step_simple). */
/* SINGLE: the launch plays exactly one tick (p.ticks == 1: every launch of the bench, of an RL loop, of the explicit-move
 * steps) — an instantiation without the tick loop, so that nothing is kept alive "for the next tick". */
/* CHAIN: a launch of a queue without barriers between its packets (pom_chain.h): the wavefront first waits until its tile's
 * previous tick has been stored (StepParams.tile_seq), fetches the record past the vector cache, and publishes the tile when
 * its own stores have arrived.  A launch then no longer lasts as long as its slowest wavefront: the next launch's
 * wavefronts start on the tiles that are ready. */
extern "C" __device__ uint64_t pom_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id"); /* the AQL packet's index in its queue */
/* how long a wavefront waits for the visit before its own before it gives up (it must never hang the device): wall-clock time, so
 * that a holder descheduled for a while (several processes time-slicing one GPU, a debugger) is waited for; a give-up poisons the
 * tile and the host replays its ticks after the next join — slow, never wrong (pom_runtime.h chain_settle) */
enum { POM_CHAIN_WAIT_LIMIT_US = 2000000 };
#ifndef POM_CHAIN_WORD_STRIDE
#define POM_CHAIN_WORD_STRIDE 16 /* 64-bit words between the ticket words of neighbouring tiles: a 128-byte line each (10.28 - 10.30 us
                                    per step against 10.41 - 10.48 with the words packed: atomics of neighbouring tiles do not queue on one line) */
#endif

/* ---- a chained launch's visit of a tile (pom_chain.h): the hand-off both the CHAIN instantiations of pom_step_kernel and the
 * hand-off litmus (pom_chain_litmus_kernel) go through ------------------------------------------------------------------------- */
struct PomChainVisit {
    int32_t dist = 0;            /* how many visits after the launch's chain_seq0 this one is: picks the tick (and the tape's tick) */
    unsigned long long done = 0; /* what the wavefront adds to the tile's word when its stores have arrived */
#if defined(POM_CHAIN_DIAG)
    long long t0 = 0, t1 = 0, t2 = 0, rt0 = 0;
    int polls = 0;
    uint32_t visit = 0;
#endif
};
/* which XCD this workgroup runs on (0..7; >= 8: not the machine this was written for) */
__device__ __forceinline__ uint32_t pom_chain_xcd()
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xFu;
}
/* Take a ticket for the tile's word and wait for the visit before it.  true: the tile is this wavefront's to play — its record, as
 * the previous visit stored it, may be loaded NOW (past the vector cache: sc1).  false: nothing may be touched (the tile is, or
 * has just been, poisoned; the flag word says why). */
__device__ __forceinline__ bool pom_chain_enter(unsigned long long* word, uint32_t chain_xcd, uint32_t chain_seq0, uint32_t tape_len, uint64_t wait_limit,
                                                uint32_t* err, int lane, PomChainVisit& v)
{
    const uint32_t xcd = chain_xcd + 1u;
#if defined(POM_CHAIN_DIAG)
    v.t0 = (long long)__builtin_readcyclecounter();
    v.rt0 = (long long)wall_clock64();
#endif
    /* take a ticket: the old value says which visit of the tile this is — and, mostly, that the visit before it is stored */
    unsigned long long w = 0;
    if (lane == 0) w = __hip_atomic_fetch_add(word, 1ull << POM_CHAIN_TICKET_SHIFT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
    const uint32_t visit = (uint32_t)(w >> POM_CHAIN_TICKET_SHIFT);
    /* which tick: the call's first tick + how far this visit is from the call's first visit — a signed distance: launches of
     * two calls may be in flight together, and a wavefront of the later call can draw a ticket of the earlier one */
    v.dist = (int32_t)pom_chain_visit_distance(visit, chain_seq0);
#if defined(POM_CHAIN_DIAG)
    v.t1 = (long long)__builtin_readcyclecounter();
#endif
    /* stored visits == this visit's number: its turn.  Until then poll — for as long as the wall clock allows; a poisoned tile
     * (somebody before this visit could not play) is nobody's turn any more */
    int polls = 0;
    bool timed_out = false;
    if (!((uint32_t)w & POM_CHAIN_POISON) && (int32_t)pom_chain_visit_distance((uint32_t)w & POM_CHAIN_COUNT_MASK, visit) < 0) { /* wave-uniform */
        const uint64_t t_wait0 = wall_clock64();
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
            polls++;
            if (((uint32_t)w & POM_CHAIN_POISON) || (int32_t)pom_chain_visit_distance((uint32_t)w & POM_CHAIN_COUNT_MASK, visit) >= 0) break;
            if (wall_clock64() - t_wait0 >= wait_limit) {
                timed_out = true;
                break;
            }
        }
    }
    (void)polls;
    const uint32_t was_on = (uint32_t)(w >> 32) & 0xFu;
    const bool poisoned = ((uint32_t)w & POM_CHAIN_POISON) != 0;
    const bool wrong_xcd = was_on != 0u && was_on != xcd;
    const bool off_tape = tape_len != 0u && (uint32_t)v.dist >= tape_len;
    if (poisoned || timed_out || wrong_xcd || off_tape) {
        /* The visit before this one never arrived in time, or it was stored through another XCD's L2 (what this L2 holds of
         * the tile may be stale), or the ticket lies outside the move tape (launches of two tape calls in flight together:
         * the host joins between them, so never).  Nothing is stepped and NOTHING IS COUNTED: the tile is poisoned, every
         * later visitor leaves it alone, its stored count stays at the ticks it really played, and the host — which finds
         * the flag after the next join — replays the rest with ordinary launches (chain_settle, pom_runtime.h). */
        if (lane == 0 && !poisoned) {
            __hip_atomic_fetch_or(word, (unsigned long long)POM_CHAIN_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_or(err, (uint32_t)(timed_out ? POM_CHAIN_E_TIMEOUT : wrong_xcd ? POM_CHAIN_E_XCD : POM_CHAIN_E_TAPE), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
        }
        return false;
    }
#if defined(POM_CHAIN_DIAG)
    v.t2 = (long long)__builtin_readcyclecounter();
    v.polls = polls;
    v.visit = visit;
#endif
    v.done = 1ull + ((unsigned long long)(xcd - was_on) << 32);
    return true;
}
/* Hand the tile on: called after the record's stores have been ISSUED; waits until the L2 has acknowledged them, then counts the
 * visit as stored (which is what the next visitor polls for). */
__device__ __forceinline__ void pom_chain_leave(unsigned long long* word, int lane, const PomChainVisit& v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(word, v.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* a dword that a wavefront of an earlier, still unfinished launch may have written (chained launches): past this CU's vector cache */
template <bool CHAIN>
__device__ __forceinline__ uint32_t pom_load_shared(const uint32_t* ptr)
{
    return CHAIN ? __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *ptr;
}

/* OBS: the launch also writes the observation of the state it leaves behind (pom_observe_tile) — an RL tick is then one launch
 * and one read of the record instead of two of each. */
template <int EPW, int G, bool FRESH, bool POLICY = false, bool ATEND = false, bool SINGLE = false, bool CHAIN = false, bool OBS = false>
__global__ __launch_bounds__(64 * POM_WPB, (POLICY ? 4 : G == 4 ? POM_QUAD_WAVES : EPW == 16 ? 3 : EPW == 32 ? 2 : 1)) void pom_step_kernel(StepParams p)
{
    static_assert(!CHAIN || SINGLE, "chained launches play one tick each");
    static_assert(!OBS || (SINGLE && !CHAIN && !POLICY && POM_WPB == 1), "the fused observation exists for the one-tick explicit-move shape");
    static_assert(!SINGLE || G == 4, "the one-tick instantiation exists for the quad shape");
    static_assert(G == 1 || (G == 4 && EPW == 16), "a quad per env needs 16 envs per wavefront");
    static_assert(!POLICY || G == 4, "the policy runs one agent per lane of the quad");
    static_assert(!ATEND || G == 4, "the end-of-tick reset is built for the quad shape only");
    /* POLICY: the danger map (32 rows of bytes) and the cell sets (12 rows) live where the tick keeps its bomb destinations
     * and explosion frames — the policy of a tick is over before its tick begins.  139 rows = 8,896 B: 17 wavefronts per
     * CU would fit, 16 (all of 65,536 envs resident at once) are needed. */
    /* OBS: the staging area — one env's planes (1,936 + 16 B) or four envs' code planes (2,420 B: 38 rows) — lies over the same
     * scratch rows — the tick is over when the observation begins */
    constexpr int OBS_ROWS = (obs_stage_vecs(OBS_PASS_ENVS_FUSED) * 16 + EPW * 4 - 1) / (EPW * 4);
    constexpr int OVERLAY = POLICY ? 44 : OBS ? OBS_ROWS : 0; /* rows behind the record that the policy / the export use when the tick does not */
    constexpr int ROWS = POM_REC_DWORDS + OVERLAY > LDS_ROWS ? POM_REC_DWORDS + OVERLAY : LDS_ROWS;
    static_assert(ROWS >= LDS_ROWS, "the tick's scratch rows fit under the overlay");
    __shared__ __attribute__((aligned(16))) uint32_t tiles_[POM_WPB][ROWS * EPW];
    uint32_t* const tile = tiles_[POM_WPB == 1 ? 0 : threadIdx.x >> 6];
    uint8_t* const danger = reinterpret_cast<uint8_t*>(tile + POM_REC_DWORDS * EPW);
    uint32_t* const sets = tile + (POM_REC_DWORDS + 32) * EPW;
    const int lane = threadIdx.x & 63;
    /* XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one, each XCD has its own
     * L2).  Neighbouring tiles are given to workgroups of the SAME XCD (bijective for any grid size): with the buffers laid
     * out row-major over all envs (rounds 1-2) neighbouring tiles shared the 128-B lines of every record row and the
     * second touch of a line became an L2 hit instead of a second HBM fetch; with a tile contiguous in memory (round 3) it
     * keeps an XCD's traffic in one region of memory. */
    int64_t tile_local;
    uint32_t chain_xcd = 0;
    if (CHAIN) {
        /* the XCD the workgroup IS on decides its tile (launches of different queues start their round-robin at different
         * XCDs): XCD x plays tiles x * q .. x * q + q - 1, its k-th workgroup (workgroup ids x0, x0 + 8, ...) the k-th of them.
         * The grid is padded to a multiple of 8 workgroups.  chain_rot rotates that order within the XCD (a bijection: still every
         * tile once per launch). */
        chain_xcd = pom_chain_xcd();
        const int64_t q = gridDim.x / 8;
        int64_t k_in_xcd = (int64_t)(blockIdx.x / 8) + p.chain_rot;
        if (k_in_xcd >= q) k_in_xcd -= q;
        tile_local = (int64_t)chain_xcd * q + k_in_xcd;
        if (chain_xcd >= 8u) { /* not the machine this was written for: nothing is stepped, no ticket is drawn (the verify pass reports it) */
            if (lane == 0) __hip_atomic_fetch_or(p.chain_err, (uint32_t)POM_CHAIN_E_XCD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (POM_WPB == 1 && p.block0 + tile_local >= p.block_end) return; /* a workgroup of the padding */
    } else {
#if defined(POM_NO_XCD_REMAP)
    tile_local = blockIdx.x;
#else
    {
        const int64_t b = blockIdx.x, nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8;
        tile_local = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
    }
#endif
    }
    if (POM_WPB > 1) {
        tile_local = tile_local * POM_WPB + (threadIdx.x >> 6);
        if (p.block0 + tile_local >= p.block_end) return; /* the last workgroup of a launch may be short of tiles */
    }
    const int64_t tile_id = p.block0 + tile_local;
    const int64_t np = p.n_pad;
    const bool env_mode = p.mode == POM_MODE_ENV;
    /* data movement: lane -> (env el, row group sub) so that one DMA / store instruction covers 64/EPW rows */
    const int el = lane % EPW, sub = lane / EPW;
    const int64_t e_d = tile_id * EPW + el;
    uint32_t* col_d = p.state + pom_rec_col(e_d); /* buffers hold n_pad columns: in range for every lane */
    /* the tick: G = 1: the lanes with sub == 0 own env el; G = 4: lane -> (env lane/4, member lane%4), all lanes run */
    const int ec = G == 1 ? el : lane >> 2;
    const int member = G == 1 ? 0 : lane & 3;
    const int64_t e = tile_id * EPW + ec;
    const bool owner = G == 1 ? sub == 0 : member == 0; /* the lane that speaks for env e (register rows, counters) */
    const bool runs = G == 1 ? sub == 0 : true;         /* the lanes that execute env e's tick */
    const bool valid = runs && e < p.n;
    const uint32_t env_key = (uint32_t)(p.env_offset + e); /* the env's number in the whole job: what its move draws are keyed by */
    uint32_t* t = tile + ec;

#if defined(POM_DIAG)
    const long long t_begin = (long long)clock64();
#endif
    long long c_steps = 0, c_episodes = 0, c_resets = 0, c_ub = 0;

    /* the launch's first tick: asked for BEFORE the record, so that waiting for it (in-order vmcnt) does not wait for the record */
    uint32_t tick0 = CHAIN ? p.tick0 : p.tick0 + *p.tick_base; /* (graphs replay sub-batch launches, never chained ones) */
    PomChainVisit cv; /* CHAIN: this wavefront's visit of its tile */
    if (CHAIN) {
        unsigned long long* const word = p.tile_seq + tile_id * POM_CHAIN_WORD_STRIDE;
        if (!pom_chain_enter(word, chain_xcd, p.chain_seq0, p.tape_len, p.chain_wait_limit, p.chain_err, lane, cv)) return;
        tick0 += (uint32_t)cv.dist;
        /* the loads below are issued after the word has been seen: the record they fetch is the stored one */
        load_tile16_x4<POM_REC_DWORDS, 16>(p.state + tile_id * POM_TILE_DWORDS, POM_TILE_ENVS, tile, lane);
    } else if (EPW == 16) load_tile16_x4(p.state + tile_id * POM_TILE_DWORDS, POM_TILE_ENVS, tile, lane); /* 16-byte pieces, 7 instructions of 1 KB */
    else load_tile<EPW>(col_d, POM_TILE_ENVS, tile, sub);
    uint32_t m0 = 0, m1 = 0; /* POLICY: this lane's agent's memory */
    if (POLICY) {
        m0 = pom_load_shared<CHAIN>(p.agent_mem + tile_id * 64 + lane);
        m1 = pom_load_shared<CHAIN>(p.agent_mem + 4 * np + tile_id * 64 + lane);
    }
    /* the first tick's moves do not depend on the record: hash / fetch them while the record is on its way */
    uint64_t draw0 = 0;
    int4 moves0 = make_int4(0, 0, 0, 0);
    /* explicit moves come one tick per launch: the several-tick quad kernel never sees any (step_kernel_for) */
    constexpr bool TAKES_MOVES = SINGLE || G == 1;
    if (!POLICY) {
        if (TAKES_MOVES && p.moves) {
            /* chained: tick `chain_dist` of the caller's move tape (pom_batch_step_device_many) */
            if (valid) moves0 = reinterpret_cast<const int4*>(p.moves)[(CHAIN ? (int64_t)cv.dist * p.n : (int64_t)0) + e];
        } else {
            if (G == 4 && !SINGLE) { /* several ticks per launch: the draw is made where it is used, nothing to carry */
            } else if (G == 4) draw0 = pom_rng_draw_half(p.seed, env_key, tick0, member >> 1); /* lane m needs agent m's 16 bits only */
            else draw0 = pom_rng_draw(p.seed, env_key, tick0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the DMA rows have landed (one wavefront per workgroup: no barrier) */
    PomLane L;
    int time_step = 0;
    uint32_t status = 0;
    lane_from_tile(L, time_step, status, t, EPW);
    bool restarted = false;

    LdsEnv<EPW, G> acc(tile, ec, member);
    PomStepper<LdsEnv<EPW, G>> stepper(acc, L);
#if defined(POM_TRUNC)
    L.trunc = p.trunc;
#endif
#if defined(POM_DIAG)
    for (int k = 0; k < POM_PH_N; k++) L.t_acc[k] = 0;
    L.t_last = t_begin;
    POM_STAMP(L, POM_PH_LOAD);
#endif

    const int n_ticks = SINGLE ? 1 : p.ticks;
    for (int tk = 0; tk < n_ticks; tk++) {
        if (ATEND) {
            /* POM_RESET_AT_END (a separate instantiation): nobody is finished when a tick begins; see the end of the tick */
        } else if (FRESH) { /* a separate instantiation: the replay kernel carries none of this */
            /* fresh boards: a finished env starts its next game on the board (board_seed, env, games played) of
             * pom_boardgen.h, drawn into its tile column by the whole wavefront, one restarting env after the other —
             * the one place, first tick included */
            const bool reload = valid && env_mode && p.auto_reset == POM_RESET_AT_START && (status & POM_ST_DONE);
            uint64_t todo = __ballot(reload && owner);
            while (todo) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)todo) - 1); /* an owner lane */
                todo &= todo - 1;
                const int ec_u = G == 1 ? src : src >> 2;
                uint32_t ep = 0;
                if (lane == src) {
                    ep = pom_load_shared<CHAIN>(p.episode + e) + 1u;
                    p.episode[e] = ep;
                }
                ep = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)ep, src));
                const uint32_t key = pom_board_key(p.board_seed, (uint32_t)(p.env_offset + tile_id * EPW + ec_u), ep);
                pom_boardgen_wave<EPW, POM_REC_DWORDS>(tile, ec_u, (uint32_t)__builtin_amdgcn_readfirstlane((int)key), lane);
            }
            asm volatile("" ::: "memory"); /* other lanes wrote this lane's column: no read of it may be scheduled earlier */
            if (reload) lane_from_tile(L, time_step, status, t, EPW); /* the register-resident rows, from the new record */
            restarted = reload;
            c_resets += __popcll(__ballot(reload && owner));
        } else {
            /* a finished env restarts from its snapshot (tick 0: as the record says; later ticks of a launch: as the epilogue
             * found), one restarting env after the other, by the whole wavefront */
            const bool reload = valid && env_mode && p.auto_reset == POM_RESET_AT_START && (status & POM_ST_DONE);
            uint64_t todo = __ballot(reload && owner);
            if (todo) {
                c_resets += __popcll(todo);
                int lane_now = lane; /* (as at the end of the tick: addresses worked out where they are used — a launch of several ticks
                                        would carry them from tick to tick) */
                if (!SINGLE) asm volatile("" : "+v"(lane_now));
                do {
                    const int src = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)todo) - 1); /* an owner lane */
                    todo &= todo - 1;
                    const int ec_u = G == 1 ? src : src >> 2;
                    restart_column<EPW>(tile, ec_u, p.snap + (tile_id * EPW + ec_u) * POM_REC_DWORDS, lane_now);
                } while (todo);
                asm volatile("" ::: "memory"); /* other lanes wrote this lane's column: no read of it may be scheduled earlier */
                if (reload) lane_from_tile(L, time_step, status, t, EPW); /* the register-resident rows, from the new record */
            }
            restarted = reload;
        }
        const bool active = valid && !(env_mode && (status & POM_ST_DONE));
        bool newly_done = false, new_ub = false;
        int mv_own = POM_MOVE_IDLE;
        if (POLICY) {
            if (tk > 0) { /* the memory does not stay in registers through the tick */
                m0 = p.agent_mem[tile_id * 64 + lane];
                m1 = p.agent_mem[4 * np + tile_id * 64 + lane];
            }
            if (restarted) m0 = m1 = 0; /* a new game gets fresh agents */
            restarted = false;
            PolicyStore st{tile, t, danger + ec, sets + ec, member};
#if defined(POM_TRUNC)
            st.trunc = p.trunc; /* cuts -4 .. -1: none of the policy, + clear, + fill, + safe (act and the tick skipped); 0: the whole policy */
            if (active) {
                if (p.trunc > -4) pom_policy_prepare_clear(st);
                if (p.trunc > -3) pom_policy_prepare_fill(st, PomPolicyEnv{{L.a0[0], L.a0[1], L.a0[2], L.a0[3]}, {acc.ag1(0), acc.ag1(1), acc.ag1(2), acc.ag1(3)}, L.bIdx, L.bCnt});
                if (p.trunc > -2) pom_policy_prepare_safe(st);
            }
            if (p.trunc > -1)
#else
            if (active) { /* all four lanes of the env, dead agents' lanes included */
                pom_policy_prepare_clear(st);
                pom_policy_prepare_fill(st, PomPolicyEnv{{L.a0[0], L.a0[1], L.a0[2], L.a0[3]}, {acc.ag1(0), acc.ag1(1), acc.ag1(2), acc.ag1(3)}, L.bIdx, L.bCnt});
                pom_policy_prepare_safe(st);
            }
#endif
            { /* act() is only asked of live agents (environment.cpp:139-146); the wavefront's searches run together: every lane goes in */
                const PomPolicyEnv E{{L.a0[0], L.a0[1], L.a0[2], L.a0[3]}, {acc.ag1(0), acc.ag1(1), acc.ag1(2), acc.ag1(3)}, L.bIdx, L.bCnt};
                const uint32_t r = pom_rng_draw_half(p.seed, env_key, tick0 + (uint32_t)tk, member >> 1);
                const bool actor = active && !ag_dead(sel4(member, L.a0));
#if defined(POM_DIAG)
                long long pt_last = 0, pt_acc[POM_PP_N];
                mv_own = pom_policy_wave(st, E, member, m0, m1, actor, (int)((((r >> (16 * (member & 1))) & 0xFFFFu) * 5u) >> 16), sets, lane, pt_last, pt_acc);
#else
                mv_own = pom_policy_wave(st, E, member, m0, m1, actor, (int)((((r >> (16 * (member & 1))) & 0xFFFFu) * 5u) >> 16), sets, lane);
#endif
            }
            p.agent_mem[tile_id * 64 + lane] = m0;
            p.agent_mem[4 * np + tile_id * 64 + lane] = m1;
        }
        if (active) {
            /* the moves, a nibble per agent.  With a quad per env lane m works out agent m's and the quad exchanges them */
            uint32_t mvp;
            if (G == 4) {
                int mine;
                if (POLICY) {
                    mine = mv_own;
                } else if (TAKES_MOVES && p.moves) { /* explicit moves: one tick per launch */
                    const int lo = (member & 1) ? moves0.y : moves0.x, hi = (member & 1) ? moves0.w : moves0.z;
                    mine = (member & 2) ? hi : lo;
                } else {
                    const uint32_t r = SINGLE ? (uint32_t)draw0 : pom_rng_draw_half(p.seed, env_key, tick0 + (uint32_t)tk, member >> 1);
                    mine = pom_rng_pick((r >> (16 * (member & 1))) & 0xFFFFu, p.dist);
                }
                mvp = stepper.pack_moves_quad(mine);
            } else {
                int mv[4];
                if (p.moves) {
                    mv[0] = moves0.x; mv[1] = moves0.y; mv[2] = moves0.z; mv[3] = moves0.w;
                } else {
                    const uint64_t r = tk == 0 ? draw0 : pom_rng_draw(p.seed, env_key, tick0 + (uint32_t)tk);
#pragma unroll
                    for (int i = 0; i < 4; i++) mv[i] = pom_rng_pick((uint32_t)(r >> (16 * i)) & 0xFFFFu, p.dist);
                }
                mvp = stepper.pack_moves(mv);
            }
            L.ub = 0; /* this tick's flags; the record's (in the top bytes of two agent words, untouched by the tick) are added afterwards */
            status &= ~(uint32_t)POM_ST_RESTARTED;
            POM_STAMP(L, POM_PH_RESTART); /* diagnostic builds: the restarts and the move draw */
#if defined(POM_TRUNC)
            if (p.trunc <= 0) {
            } else
#endif
            if (POLICY) {
                /* a fresh view of the tile for the tick: keeps the compiler from computing the tick's addresses before the
                 * policy and carrying them through it (the fused kernel otherwise wants 170 VGPRs) */
                uint32_t* t2 = tile;
                asm volatile("" : "+v"(t2));
                LdsEnv<EPW, G> acc2(t2, ec, member);
                PomStepper<LdsEnv<EPW, G>> stepper2(acc2, L);
                stepper2.step_packed(mvp);
            } else {
                stepper.step_packed(mvp);
            }
            new_ub = L.ub != 0;
            L.ub |= (t[(POM_REC_AGENTS + 5) * EPW] >> 24) | ((t[(POM_REC_AGENTS + 7) * EPW] >> 24) << 8);
            if (env_mode) {
                /* timeStep is looked at here only: read back from the tile instead of living in a register through the tick */
                time_step = (int)t[POM_REC_TIMESTEP * EPW] + 1;
                if (owner) t[POM_REC_TIMESTEP * EPW] = (uint32_t)time_step;
                status = pom_env_epilogue(L, time_step, p.max_steps, status);
                newly_done = (status & POM_ST_DONE) != 0;
            }
        }
        c_steps += __popcll(__ballot(active && owner));
        c_episodes += __popcll(__ballot(newly_done && owner));
        c_ub += __popcll(__ballot(new_ub && owner));
        if (ATEND) {
            /* The tick that finishes an episode also starts the next one: the final record goes to the terminal buffer, the
             * env's column is replaced by its start state (snapshot record, or the next generated board), the agents' memory
             * is wiped.  What the caller reads next — state, observation, status — is the new episode's first state, marked
             * POM_ST_RESTARTED; the move it then supplies is a move for THAT state. */
            uint64_t todo = __ballot(newly_done && owner);
            if (todo) {
                c_resets += __popcll(todo);
                if (newly_done && owner) { /* the register-resident rows of the final record (timeStep is in the tile already) */
#pragma unroll
                    for (int k = 0; k < 8; k++) t[(POM_REC_AGENTS + k) * EPW] = pom_lane_agent_word(L, status, k, t[(POM_REC_AGENTS + k) * EPW]);
                }
                asm volatile("" ::: "memory");
                int lane_end = lane; /* (a lane id the compiler cannot see through: the global addresses below are worked out here, not at
                                        the top of the kernel and carried through the tick) */
                asm volatile("" : "+v"(lane_end));
                do {
                    const int src = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)todo) - 1); /* an owner lane */
                    todo &= todo - 1;
                    const int ec_u = G == 1 ? src : src >> 2;
                    const int64_t e_u = tile_id * EPW + ec_u;
                    uint32_t* tr = p.terminal + e_u * POM_REC_DWORDS;
                    column_to_record<EPW>(tile, ec_u, tr, lane_end);
                    if (FRESH) {
                        uint32_t ep = 0;
                        if (lane == src) {
                            ep = pom_load_shared<CHAIN>(p.episode + e_u) + 1u;
                            p.episode[e_u] = ep;
                        }
                        ep = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)ep, src));
                        const uint32_t key = pom_board_key(p.board_seed, (uint32_t)(p.env_offset + e_u), ep);
                        pom_boardgen_wave<EPW, POM_REC_DWORDS>(tile, ec_u, (uint32_t)__builtin_amdgcn_readfirstlane((int)key), lane_end);
                    } else {
                        restart_column<EPW>(tile, ec_u, p.snap + e_u * POM_REC_DWORDS, lane_end);
                    }
                } while (todo);
                asm volatile("" ::: "memory");
                if (newly_done && runs) {
                    lane_from_tile(L, time_step, status, t, EPW);
                    status |= POM_ST_RESTARTED;
                    if (POLICY) { /* a new game gets fresh agents */
                        int lane_here = lane; /* (a lane id the compiler cannot see through: the two addresses are worked out here, not carried through the tick) */
                        asm volatile("" : "+v"(lane_here));
                        p.agent_mem[tile_id * 64 + lane_here] = 0u;
                        p.agent_mem[4 * np + tile_id * 64 + lane_here] = 0u;
                    }
                }
            }
        }
        POM_STAMP(L, POM_PH_EPILOGUE);
    }

    /* write back: the owner puts the register-resident rows into the tile, then the whole record leaves in row groups */
    if (owner) { /* (timeStep went into the tile with the tick's epilogue) */
#pragma unroll
        for (int k = 0; k < 8; k++) t[(POM_REC_AGENTS + k) * EPW] = pom_lane_agent_word(L, status, k, (k & 1) ? t[(POM_REC_AGENTS + k) * EPW] : 0u);
    }
    if (EPW == 16) {
        /* the store addresses are functions of the lane id and the arguments only: left alone the compiler computes them at
         * the top of the kernel and keeps 14 registers alive (or spilled) through the whole tick — hand it a lane id it cannot
         * see through, so that they are computed here */
        int lane_late = lane;
        asm volatile("" : "+v"(lane_late));
        /* chained: with the non-temporal hint (-2.5 % per step: the stores are acknowledged sooner and the wavefront's slot is free
         * sooner; on the sub-batch kernels' stores, or on the loads, the hint gains nothing or loses) */
        store_tile16_x4<CHAIN>(p.state + tile_id * POM_TILE_DWORDS, POM_TILE_ENVS, tile, lane_late);
    } else {
        store_tile<EPW>(col_d, POM_TILE_ENVS, tile, sub, el);
    }
    if (OBS) {
        /* the observation of what the tick left in the tile, while the record's stores are on their way (they have been read out
         * of LDS into registers; the staging area lies behind the record rows) */
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ObserveParams op;
        op.state = nullptr;
        op.n = p.n;
        op.n_pad = p.n_pad;
        op.block0 = 0;
        op.planes = p.obs_planes;
        op.agent_attrs = p.obs_agent_attrs;
        op.env_attrs = p.obs_env_attrs;
        op.dtype = p.obs_dtype;
        op.per_agent = p.obs_per_agent;
        pom_observe_tile<OBS_PASS_ENVS_FUSED>(op, tile, reinterpret_cast<uint4*>(tile + POM_REC_DWORDS * EPW), tile_id, lane);
    }
    if (CHAIN) { /* the record's stores have been acknowledged by the L2 before the word that hands the tile on is written */
        pom_chain_leave(p.tile_seq + tile_id * POM_CHAIN_WORD_STRIDE, lane, cv);
#if defined(POM_CHAIN_DIAG)
        if (lane == 0) { /* per tile, summed over launches: cycles to the ticket, cycles polling, cycles in all, polls */
            unsigned long long* d = p.tile_seq + (p.block_end - p.block0) * POM_CHAIN_WORD_STRIDE + 68 * tile_id;
            d[4 + 2 * (cv.visit & 31)] = (unsigned long long)cv.rt0; /* the last 32 visits: start and end on the 100 MHz clock */
            d[5 + 2 * (cv.visit & 31)] = (unsigned long long)wall_clock64();
            d[0] += (unsigned long long)(cv.t1 - cv.t0);
            d[1] += (unsigned long long)(cv.t2 - cv.t1);
            d[2] += (unsigned long long)((long long)__builtin_readcyclecounter() - cv.t0);
            d[3] += (unsigned long long)cv.polls;
        }
#endif
    }

#if defined(POM_DIAG)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    POM_STAMP(L, POM_PH_STORE);
    if (p.diag) { /* one slot per wavefront: no contended atomics that would distort the timing.  Phases entered by some envs
                     only (the blast engines) are booked by the lanes inside; everybody else books the same time on the phase
                     around them — so the wavefront reports the lane that spent the most time inside */
        long long inside = 0;
        for (int k = POM_PH_X_SCAN; k <= POM_PH_X_SHORT; k++) inside += L.t_acc[k];
        long long best = inside;
        for (int o = 32; o > 0; o >>= 1) {
            const long long w = __shfl_xor(best, o);
            best = w > best ? w : best;
        }
        const uint64_t who = __ballot(inside == best);
        if (lane == __ffsll((unsigned long long)who) - 1)
            for (int k = 0; k < POM_PH_N; k++) p.diag[tile_id * POM_PH_N + k] += L.t_acc[k];
    }
#endif
    if (lane == 0) {
        /* each wavefront owns its slot, so nothing contends; the adds are returnless atomics only because those are
         * fire-and-forget — a load / add / store would keep the finished wavefront alive for a global round trip */
        unsigned long long* wc = reinterpret_cast<unsigned long long*>(p.wave_counters + tile_id * POM_CNT_N);
        __hip_atomic_fetch_add(&wc[POM_CNT_STEPS], (unsigned long long)c_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c_episodes) __hip_atomic_fetch_add(&wc[POM_CNT_EPISODES], (unsigned long long)c_episodes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c_resets) __hip_atomic_fetch_add(&wc[POM_CNT_RESETS], (unsigned long long)c_resets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c_ub) __hip_atomic_fetch_add(&wc[POM_CNT_UB_TICKS], (unsigned long long)c_ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* The hand-off by itself (pom_chain_litmus, tests): the same tile choice, ticket, wait, sc1 DMA load, non-temporal store and
 * publication as the CHAIN step kernels, with a "tick" whose result gives every stale or torn read away — visit v of a tile must
 * find ALL 1,312 dwords of the record equal to v (the j-th dword XOR-tagged with its index, so that a record of another tile or a
 * shifted piece cannot pass either) and leaves them at v + 1.  out[0]: records that were not what the visit before left (counted
 * per wavefront), out[1]: dwords that differed, out[2]: visits played. */
struct LitmusParams {
    uint32_t* data;              /* tiles x POM_TILE_DWORDS */
    unsigned long long* tile_seq;
    uint32_t* err;
    unsigned long long* out;
    int64_t tiles;
    uint32_t chain_seq0;
    uint64_t wait_limit;
};
__global__ __launch_bounds__(64) void pom_chain_litmus_kernel(LitmusParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[POM_REC_DWORDS * 16];
    const int lane = threadIdx.x;
    const uint32_t xcd = pom_chain_xcd();
    if (xcd >= 8u) return;
    const int64_t tile_id = (int64_t)xcd * (gridDim.x / 8) + blockIdx.x / 8;
    if (tile_id >= p.tiles) return;
    unsigned long long* const word = p.tile_seq + tile_id * POM_CHAIN_WORD_STRIDE;
    PomChainVisit cv;
    if (!pom_chain_enter(word, xcd, p.chain_seq0, 0u, p.wait_limit, p.err, lane, cv)) return;
    load_tile16_x4<POM_REC_DWORDS, 16>(p.data + tile_id * POM_TILE_DWORDS, POM_TILE_ENVS, tile, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t expect = p.chain_seq0 + (uint32_t)cv.dist; /* = this visit's number: as many visits are stored */
    const uint32_t tag = (uint32_t)tile_id * 2654435761u;
    int bad = 0;
#pragma unroll 4
    for (int k = lane; k < POM_TILE_DWORDS; k += 64) {
        bad += tile[k] != (expect ^ (tag + (uint32_t)k));
        tile[k] = (expect + 1u) ^ (tag + (uint32_t)k);
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    store_tile16_x4<true>(p.data + tile_id * POM_TILE_DWORDS, POM_TILE_ENVS, tile, lane);
    pom_chain_leave(word, lane, cv);
    if (lane == 0) {
        if (bad) {
            atomicAdd(p.out + 0, 1ull);
            atomicAdd(p.out + 1, (unsigned long long)bad);
        }
        atomicAdd(p.out + 2, 1ull);
    }
}

/* chained launches, before the first one: which XCD does workgroup b of a small grid land on?  (pom_chain.h checks that it is the
 * round-robin over eight XCDs the tile choice assumes — a partitioned device, or another chip, is not) */
__global__ void pom_chain_probe_kernel(uint32_t* xcd_of_block)
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) xcd_of_block[blockIdx.x] = x & 0xFu;
}

/* chained launches, after a join: every tile must have drawn exactly `visits` tickets, and have stored as many visits — or be
 * poisoned (a visitor could not play: POM_CHAIN_POISON), in which case (tile, visits really stored) goes onto the list the host
 * replays from (chain_settle).  aux[0]: the POM_CHAIN_E_* flags, aux[1]: length of the list, list: aux + 2, two dwords per entry. */
__global__ void pom_chain_verify_kernel(const unsigned long long* tile_seq, int64_t tiles, uint32_t visits, uint32_t* aux)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const unsigned long long w = tile_seq[t * POM_CHAIN_WORD_STRIDE];
    const uint32_t stored = (uint32_t)w & POM_CHAIN_COUNT_MASK;
    if ((uint32_t)(w >> POM_CHAIN_TICKET_SHIFT) != visits) atomicOr(aux, (uint32_t)POM_CHAIN_E_UNEVEN);
    if ((uint32_t)w & POM_CHAIN_POISON) {
        const uint32_t k = atomicAdd(aux + 1, 1u);
        aux[2 + 2 * k] = (uint32_t)t;
        aux[3 + 2 * k] = stored;
        if (stored >= visits) atomicOr(aux, (uint32_t)POM_CHAIN_E_UNEVEN); /* poisoned means: at least one visit not played */
    } else if (stored != visits) {
        atomicOr(aux, (uint32_t)POM_CHAIN_E_UNEVEN);
    }
}

/* ---------------------------------------------------------------------------------------------
 * SimpleAgent policy (pom_policy_body.h): one lane per AGENT, the quad 4e..4e+3 = the four agents of env e,
 * 16 envs per wavefront.  The wavefront DMAs record rows 0..63 (board, meta, agents, bombs: 0..61) of its 16 envs into a shared
 * tile; per env the four lanes together prepare a danger map ([128][16] bytes) and three cell sets.  The reachability
 * questions are flood fills on 121-bit cell sets in registers, so a wavefront needs only 9 KB of LDS (17 wavefronts per CU).  Output:
 * Move[4] per env into the handle's move buffer, agent memory (2 dwords per agent) updated in place.  A finished env that the
 * next step will restart is read from its snapshot column — or, with fresh boards, drawn here exactly as the tick will draw
 * it — and gets fresh (zero) agent memory, so policy and tick see the same game.
 * ------------------------------------------------------------------------------------------- */
enum { POL_ROWS = POM_REC_FLAMES, POL_LOAD_ROWS = 64 }; /* the policy reads rows 0..59 (board, timeStep, agents, bombs); they arrive 16 rows per instruction */
static_assert(POL_ROWS <= POL_LOAD_ROWS && POL_LOAD_ROWS <= POM_REC_DWORDS, "policy rows");


struct PolicyParams {
    const uint32_t* state;
    const uint32_t* snap;
    uint32_t* agent_mem; /* [2][4 * n_pad] */
    int32_t* moves;      /* [n_pad][4] */
    int64_t n, n_pad, env_offset, block0;
    uint64_t seed;
    uint32_t tick;
    int32_t mode, auto_reset;
    const uint32_t* episode; /* fresh boards: a restarting env is judged on the board the tick will generate for it */
    uint64_t board_seed;
    int32_t fresh;
#if defined(POM_DIAG)
    long long* diag; /* POM_PP_N accumulators per wavefront, diagnostic build only */
#endif
};

__global__ __launch_bounds__(64) void pom_policy_kernel(PolicyParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[POL_LOAD_ROWS * 16];
    __shared__ uint8_t danger[128 * 16]; /* 121 cells; the safe-set pass reads whole 32-cell words */
    __shared__ uint32_t sets[12 * 16];
    const int lane = threadIdx.x;
    const int64_t np = p.n_pad;
    int64_t tile_local;
    {
        const int64_t b = blockIdx.x, nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8;
        tile_local = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
    }
    const int64_t tile_id = p.block0 + tile_local;
    const bool env_mode = p.mode == POM_MODE_ENV;
#if defined(POM_DIAG)
    long long t_last = (long long)clock64(), t_acc[POM_PP_N] = {0, 0, 0, 0, 0, 0, 0};
#endif
    load_tile16_x4<POL_LOAD_ROWS>(p.state + tile_id * POM_TILE_DWORDS, POM_TILE_ENVS, tile, lane);
    /* the policy: lane -> (env lane/4, agent lane%4) */
    const int ec = lane >> 2, id = lane & 3;
    const int64_t e = tile_id * 16 + ec;
    const int64_t slot = e * 4 + id; /* = tile_id * 64 + lane */
    uint32_t m0 = p.agent_mem[slot], m1 = p.agent_mem[4 * np + slot];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    /* an env the tick is about to restart is judged on the board it will restart on: its snapshot record (array of structs,
     * fetched by the whole wavefront: restart_column) or, with fresh boards, the board the tick kernel is about to draw for it
     * (the tick counts the episode) */
    const bool restart = e < p.n && env_mode && p.auto_reset == POM_RESET_AT_START && ((pom_rec_meta2(tile + ec, 16) >> 8) & POM_ST_DONE);
    if (restart) m0 = m1 = 0; /* a new game gets fresh agents */
    {
        uint64_t todo = __ballot(restart && id == 0);
        while (todo) {
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)todo) - 1);
            todo &= todo - 1;
            const int ec_u = src >> 2;
            if (p.fresh) {
                const uint32_t ep = p.episode[tile_id * 16 + ec_u] + 1u;
                const uint32_t key = pom_board_key(p.board_seed, (uint32_t)(p.env_offset + tile_id * 16 + ec_u),
                                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)ep));
                pom_boardgen_wave<16, POL_ROWS>(tile, ec_u, (uint32_t)__builtin_amdgcn_readfirstlane((int)key), lane);
            } else {
                restart_column<16, POL_ROWS>(tile, ec_u, p.snap + (tile_id * 16 + ec_u) * POM_REC_DWORDS, lane);
            }
        }
        asm volatile("" ::: "memory");
    }
    const uint32_t* t = tile + ec;
    PomPolicyEnv E;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        E.a0[i] = (int)t[(POM_REC_AGENTS + 2 * i) * 16];
        E.a1[i] = (int)t[(POM_REC_AGENTS + 2 * i + 1) * 16];
    }
    const uint32_t meta = pom_rec_meta(t, 16), meta2 = pom_rec_meta2(t, 16);
    E.bIdx = (int)((meta >> 8) & 0xFF);
    E.bCnt = (int)((meta >> 16) & 0xFF);
    const bool frozen = env_mode && ((meta2 >> 8) & POM_ST_DONE); /* finished and not restarted: Environment::Step returns */
    int mv = POM_MOVE_IDLE;
    PolicyStore st{tile, t, danger + ec, sets + ec, id};
    POM_PSTAMP(POM_PP_LOAD);
    if (e < p.n && !frozen) { /* all four lanes of the env, dead agents' lanes included */
        pom_policy_prepare_clear(st);
        pom_policy_prepare_fill(st, E);
        pom_policy_prepare_safe(st);
    }
    POM_PSTAMP(POM_PP_PREPARE);
    { /* act() is only asked of live agents (environment.cpp:139-146); the wavefront's searches run together: every lane goes in */
        const bool actor = e < p.n && !frozen && !ag_dead(sel4(id, E.a0));
        const uint32_t r = pom_rng_draw_half(p.seed, (uint32_t)(p.env_offset + e), p.tick, id >> 1);
        const int draw = (int)((((r >> (16 * (id & 1))) & 0xFFFFu) * 5u) >> 16);
#if defined(POM_DIAG)
        mv = pom_policy_wave(st, E, id, m0, m1, actor, draw, sets, lane, t_last, t_acc);
#else
        mv = pom_policy_wave(st, E, id, m0, m1, actor, draw, sets, lane);
#endif
    }
    p.moves[slot] = mv;
    p.agent_mem[slot] = m0;
    p.agent_mem[4 * np + slot] = m1;
#if defined(POM_DIAG)
    POM_PSTAMP(POM_PP_STORE);
    /* a lane that skipped a phase books the wavefront's time on its next stamp: per phase, the lanes that ran it agree, so
     * take the largest (an idle wavefront's act phases read 0) */
    for (int k = 0; k < POM_PP_N; k++) {
        long long v = t_acc[k];
        for (int o = 32; o > 0; o >>= 1) {
            const long long w = __shfl_xor(v, o);
            v = w > v ? w : v;
        }
        t_acc[k] = v;
    }
    if (lane == 0 && p.diag)
        for (int k = 0; k < POM_PP_N; k++) p.diag[tile_id * POM_PP_N + k] += t_acc[k];
#endif
}

__global__ __launch_bounds__(64) void pom_observe_kernel(ObserveParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[POM_REC_DWORDS * 16];
    __shared__ uint4 stage[obs_stage_vecs(OBS_PASS_ENVS_ALONE)];
    const int lane = threadIdx.x;
    int64_t tile_local;
    {
        const int64_t b = blockIdx.x, nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8;
        tile_local = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
    }
    const int64_t tile_id = p.block0 + tile_local;
    load_tile16_x4(p.state + tile_id * POM_TILE_DWORDS, POM_TILE_ENVS, tile, lane);
    /* The builtin, not inline asm: the compiler's own wait-count bookkeeping must SEE that the LDS-DMA rows have landed before the
     * pass loop begins — otherwise it assumes at the loop's head that they may still be in flight and puts an s_waitcnt vmcnt(0) in
     * front of the first tile read of EVERY pass, which (stores count in vmcnt too) waits for the previous pass's global stores:
     * 31 of a wavefront's 58 k cycles, profiles/r04_observe_pmc.txt.  0x0F70: vmcnt(0), the other counters left alone. */
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    pom_observe_tile<OBS_PASS_ENVS_ALONE>(p, tile, stage, tile_id, lane);
}

/* every env's first board (episode 0) into its state and snapshot columns, through an LDS tile so that the records leave in
 * the tick's coalesced row groups */
__global__ __launch_bounds__(64) void pom_generate_kernel(uint32_t* state, uint32_t* snap, uint32_t* episode, int64_t n, int64_t np,
                                                          int64_t env_offset, uint64_t board_seed)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[POM_REC_DWORDS * 16];
    const int lane = threadIdx.x;
    const int64_t tile_id = blockIdx.x;
    if (tile_id * 16 + 16 > n) { /* the last, partial tile: its unused columns leave as blank records */
        for (int k = lane; k < POM_REC_DWORDS * 4; k += 64) reinterpret_cast<uint4*>(tile)[k] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    for (int ec = 0; ec < 16 && tile_id * 16 + ec < n; ec++)
        pom_boardgen_wave<16, POM_REC_DWORDS>(tile, ec, pom_board_key(board_seed, (uint32_t)(env_offset + tile_id * 16 + ec), 0u), lane);
    if (tile_id * 16 + lane < n && lane < 16) episode[tile_id * 16 + lane] = 0u;
    __syncthreads();
    /* whole tiles: the buffers hold n_pad columns; columns past n are blank records here, as after creation */
    store_tile16_x4(state + tile_id * POM_TILE_DWORDS, POM_TILE_ENVS, tile, lane);
    for (int ec = 0; ec < 16; ec++) { /* the snapshot: array of structs (restart_column); columns past n are blank like the state's */
        column_to_record<16>(tile, ec, snap + (tile_id * 16 + ec) * POM_REC_DWORDS, lane);
    }
}

/* ---- boundary kernels ----------------------------------------------------------------------- */
__global__ void pom_pack_kernel(const int32_t* __restrict__ aos, int64_t first, int64_t count, uint32_t* state, uint32_t* snap,
                                int64_t np, int* first_bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t* col = state + pom_rec_col(first + i);
    const int64_t rs = POM_TILE_ENVS; /* row stride of a column */
    int bad = pom_pack_state(aos + i * (POM_STATE_BYTES / 4), col, rs, (int)((first + i) & 15));
    /* live bombs must sit on the board and belong to a real agent: they index cells and agents */
    {
        const uint32_t m = pom_rec_meta(col, rs);
        const int bIdx = (m >> 8) & 0xFF, bCnt = (m >> 16) & 0xFF;
        if (!bad) {
            for (int k = 0; k < bCnt; k++) {
                const int b = (int)col[(POM_REC_BOMBS + (bIdx + k) % POM_Q) * rs];
                bad |= (pb_x(b) >= POM_N) | (pb_y(b) >= POM_N) | (pb_id(b) >= POM_AGENT_COUNT);
            }
        }
    }
    if (bad) {
        atomicMin(first_bad, (int)(i > INT_MAX - 1 ? INT_MAX - 1 : i));
        for (int c = 0; c < 4 * POM_REC_BOARD_DWORDS; c++) pom_rec_set_cell(col, rs, c, 0, (int)((first + i) & 15)); /* inert blank board ... */
        for (int d = POM_REC_TIMESTEP; d < POM_REC_DWORDS; d++) col[d * rs] = 0;
        pom_rec_set_meta(col, rs, 0u, (uint32_t)POM_ST_DONE << 8); /* ... that is never stepped in ENV mode */
    }
    uint32_t* s = snap + (first + i) * POM_REC_DWORDS; /* the snapshot is array-of-structs (restart_column): a dense record */
    for (int c = 0; c < 4 * POM_REC_BOARD_DWORDS; c++) pom_rec_set_cell(s, 1, c, pom_rec_cell(col, rs, c, (int)((first + i) & 15)));
    for (int d = POM_REC_TIMESTEP; d < POM_REC_DWORDS; d++) s[d] = col[d * rs];
}

__global__ void pom_unpack_kernel(const uint32_t* __restrict__ state, int64_t first, int64_t count, int64_t np, int32_t* aos)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    pom_unpack_state(state + pom_rec_col(first + i), POM_TILE_ENVS, aos + i * (POM_STATE_BYTES / 4), (int)((first + i) & 15));
}

/* ---------------------------------------------------------------------------------------------
 * ONE State, one tick, one launch: the literal `bboard::Step(State*, Move*)` (POM_MODE_RAW) and `Environment::Step`'s tick +
 * bookkeeping (POM_MODE_ENV; environment.cpp:123-169) for callers that hold a single host State (pom_step, pom_env_step).
 * `io` is pinned host memory the device reads and writes directly — no staging copies, no second and third launch:
 *   dwords   0..250  in:  the State (include/pom_state.h)        252..255  in:  Move[4]
 *   dwords 256..506  out: the State after the tick               508..511  out: done, winner, draw, ubflags
 *   dword  512       out: 1 if the State is outside the representable game states (nothing else is written then)
 *   dword  513       out: `seq`, written LAST (system-scope release): the host polls it
 *   dwords 514..516  in:  mode, max_steps, seq of this request
 * One wavefront: all 64 lanes fetch, pack (pom_pack_state's fields, a dword per lane), the quad of lanes 0..3 plays the tick with
 * the same PomStepper as pom_step_kernel, all lanes unpack and write back.
 * ------------------------------------------------------------------------------------------- */
struct StepOneParams {
    int32_t* io_base;  /* POM_ONE_SLOTS pages of POM_ONE_PAGE_DWORDS dwords each */
    uint64_t slots;    /* bit s: slot s holds a request; workgroup b serves the b-th set bit */
};
/* a slot's page: the layout above, then the request's own parameters — one launch serves whatever requests are pending, each
 * with its own mode (pom_step / pom_env_step), bound and sequence number */
enum { POM_ONE_MOVES = 252, POM_ONE_OUT = 256, POM_ONE_STATUS = 508, POM_ONE_BAD = 512, POM_ONE_SEQ = 513, POM_ONE_MODE = 514,
       POM_ONE_MAX_STEPS = 515, POM_ONE_REQ = 516, POM_ONE_PAGE_DWORDS = 1024, POM_ONE_SLOTS = 64 };

__global__ __launch_bounds__(64) void pom_step_one_kernel(StepOneParams q)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[LDS_ROWS * 16];
    __shared__ int32_t aos[256];
    const int lane = threadIdx.x;
    /* which slot: the blockIdx-th set bit of the request mask (wave-uniform) */
    uint64_t slot_bits = q.slots;
    for (unsigned b = 0; b < blockIdx.x; b++) slot_bits &= slot_bits - 1;
    struct { int32_t* io; int32_t mode, max_steps; uint32_t seq; } p;
    p.io = q.io_base + (int64_t)(__ffsll((unsigned long long)slot_bits) - 1) * POM_ONE_PAGE_DWORDS;
    p.mode = p.io[POM_ONE_MODE];
    p.max_steps = p.io[POM_ONE_MAX_STEPS];
    p.seq = (uint32_t)p.io[POM_ONE_REQ];
#pragma unroll
    for (int k = 0; k < 4; k++) aos[lane + 64 * k] = p.io[lane + 64 * k]; /* State + Move[4]: four 256-B reads of host memory */
    for (int k = lane; k < LDS_ROWS * 16; k += 64) tile[k] = 0u;           /* columns 1..15 stay blank and are never stepped */
    __syncthreads();
    /* pack into column 0: exactly pom_pack_state + the live-bomb test of pom_pack_kernel, a record dword per lane */
    const int32_t* st = aos;
    int bad = 0;
    {
        const int c = lane + 64, e0 = pom_cell_encode(st[lane], lane), e1 = c < POM_CELLS ? pom_cell_encode(st[c], c) : 0;
        bad |= (e0 < 0) | (e1 < 0);
        uint8_t* cells = reinterpret_cast<uint8_t*>(tile); /* env 0 of the tile: cell c at byte c * 16 */
        cells[lane * 16] = (uint8_t)e0; /* (the tile was zeroed: the three bytes past cell 120 stay 0) */
        if (c < POM_CELLS) cells[c * 16] = (uint8_t)e1;
    }
    const int32_t alive = st[122], bIdx = st[167], bCnt = st[168], fIdx = st[249], fCnt = st[250];
    if (lane == 61) {
        bad |= (alive < -128) | (alive > 127);
        bad |= (bIdx < 0) | (bIdx >= POM_MAX_BOMBS) | (bCnt < 0) | (bCnt > POM_MAX_BOMBS);
        bad |= (fIdx < 0) | (fIdx >= POM_MAX_BOMBS) | (fCnt < 0) | (fCnt > 255);
        tile[POM_REC_TIMESTEP * 16] = (uint32_t)st[121];
    }
    if (lane < POM_AGENT_COUNT) {
        const int32_t* a = st + 123 + 6 * lane;
        const uint32_t flags = (uint32_t)a[5];
        const int kick = (flags & 0xFF) != 0, dead = ((flags >> 8) & 0xFF) != 0;
        bad |= (a[0] < 0) | (a[0] >= POM_BOARD_SIZE) | (a[1] < 0) | (a[1] >= POM_BOARD_SIZE);
        bad |= (a[2] < -128) | (a[2] > 127);
        bad |= (a[3] < -32768) | (a[3] > 32767) | (a[4] < 0) | (a[4] > 255);
        /* the top bytes: aliveAgents, bombs.index, bombs.count, flames.index in the agents' first words; flames.count (and the clear
         * status and flags) in their second words (pom_packed.h) */
        const uint32_t m0 = (uint32_t)(lane == 0 ? alive : lane == 1 ? bIdx : lane == 2 ? bCnt : fIdx) & 0xFFu, m1 = lane == 0 ? (uint32_t)fCnt & 0xFFu : 0u;
        tile[(POM_REC_AGENTS + 2 * lane) * 16] = ((uint32_t)a[0] & 0xF) | (((uint32_t)a[1] & 0xF) << 4) | (((uint32_t)a[2] & 0xFF) << 8) |
                                                 (kick ? (uint32_t)POM_AG_KICK : 0u) | (dead ? (uint32_t)POM_AG_DEAD : 0u) | (m0 << 24);
        tile[(POM_REC_AGENTS + 2 * lane + 1) * 16] = ((uint32_t)a[3] & 0xFFFF) | (((uint32_t)a[4] & 0xFF) << 16) | (m1 << 24);
    }
    if (lane >= 20 && lane < 20 + POM_MAX_BOMBS) {
        const int k = lane - 20;
        const int b = st[147 + k];
        tile[(POM_REC_BOMBS + k) * 16] = (uint32_t)b;
        /* live bombs must sit on the board and belong to a real agent: they index cells and agents (pom_pack_kernel) */
        const int age = k - bIdx + (k < bIdx ? POM_Q : 0); /* slot k is the age-th bomb of the queue */
        if (bIdx >= 0 && bIdx < POM_MAX_BOMBS && age < bCnt) bad |= (pb_x(b) >= POM_N) | (pb_y(b) >= POM_N) | (pb_id(b) >= POM_AGENT_COUNT);
    }
    if (lane >= 40 && lane < 40 + POM_MAX_BOMBS) {
        const int32_t* f = st + 169 + 4 * (lane - 40);
        bad |= (f[0] < 0) | (f[0] >= POM_BOARD_SIZE) | (f[1] < 0) | (f[1] >= POM_BOARD_SIZE); /* also the stale slots */
        bad |= (f[2] < -128) | (f[2] > 127) | (f[3] < 0) | (f[3] > 255);
        tile[(POM_REC_FLAMES + lane - 40) * 16] =
            (uint32_t)f[0] | ((uint32_t)f[1] << 8) | (((uint32_t)f[2] & 0xFF) << 16) | ((uint32_t)f[3] << 24);
    }
    if (__ballot(bad != 0)) { /* the caller's State is left alone; pom_step reports POM_E_UNREPRESENTABLE */
        if (lane == 0) {
            p.io[POM_ONE_BAD] = 1;
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<uint32_t*>(p.io) + POM_ONE_SEQ, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    __syncthreads();

    /* the tick: lane -> (env lane / 4, member lane % 4) as in pom_step_kernel<16, 4>; env 0 is the only one there is */
    const int ec = lane >> 2, member = lane & 3;
    uint32_t* t = tile + ec;
    PomLane L;
    int time_step = 0;
    uint32_t status = 0;
    lane_from_tile(L, time_step, status, t, 16);
#if defined(POM_DIAG)
    for (int k = 0; k < POM_PH_N; k++) L.t_acc[k] = 0;
    L.t_last = 0;
#endif
#if defined(POM_TRUNC)
    L.trunc = 990;
#endif
    LdsEnv<16, 4> acc(tile, ec, member);
    PomStepper<LdsEnv<16, 4>> stepper(acc, L);
    const bool env_mode = p.mode == POM_MODE_ENV;
    if (ec == 0) {
        const uint32_t mvp = stepper.pack_moves_quad(aos[POM_ONE_MOVES + member]);
        L.ub = 0;
        stepper.step_packed(mvp);
        if (env_mode) {
            time_step++;
            status = pom_env_epilogue(L, time_step, p.max_steps, status);
        }
        if (member == 0) { /* the register-resident rows */
            t[POM_REC_TIMESTEP * 16] = (uint32_t)time_step;
#pragma unroll
            for (int k = 0; k < 8; k++) t[(POM_REC_AGENTS + k) * 16] = pom_lane_agent_word(L, status, k, (k & 1) ? t[(POM_REC_AGENTS + k) * 16] : 0u);
        }
    }
    __syncthreads();

    /* unpack column 0 (pom_unpack_state, a few State dwords per lane) straight into host memory */
    int32_t* out = p.io + POM_ONE_OUT;
    const uint32_t m = pom_rec_meta(tile, 16), m2 = pom_rec_meta2(tile, 16);
    out[lane] = pom_cell_decode(pom_rec_cell(tile, 16, lane, 0), lane);
    if (lane + 64 < POM_CELLS) out[lane + 64] = pom_cell_decode(pom_rec_cell(tile, 16, lane + 64, 0), lane + 64);
    if (lane == 61) {
        out[121] = (int32_t)tile[POM_REC_TIMESTEP * 16];
        out[122] = pom_sext8(m);
        out[167] = (int32_t)((m >> 8) & 0xFF);
        out[168] = (int32_t)((m >> 16) & 0xFF);
        out[249] = (int32_t)(m >> 24);
        out[250] = (int32_t)(m2 & 0xFF);
        const uint32_t s8 = (m2 >> 8) & 0xFF;
        p.io[POM_ONE_STATUS + 0] = (s8 & POM_ST_DONE) ? 1 : 0;
        p.io[POM_ONE_STATUS + 1] = (int)((s8 >> POM_ST_WINNER_SHIFT) & 7) - 1;
        p.io[POM_ONE_STATUS + 2] = (s8 & POM_ST_DRAW) ? 1 : 0;
        p.io[POM_ONE_STATUS + 3] = (int32_t)(m2 >> 16);
        p.io[POM_ONE_BAD] = 0;
    }
    if (lane < POM_AGENT_COUNT) {
        const uint32_t a0 = tile[(POM_REC_AGENTS + 2 * lane) * 16], a1 = tile[(POM_REC_AGENTS + 2 * lane + 1) * 16];
        int32_t* a = out + 123 + 6 * lane;
        a[0] = ag_x((int)a0);
        a[1] = ag_y((int)a0);
        a[2] = ag_bombcount((int)a0);
        a[3] = ag_max_bombs((int)a1);
        a[4] = ag_strength((int)a1);
        a[5] = ag_kick((int)a0) | (ag_dead((int)a0) << 8);
    }
    if (lane >= 20 && lane < 20 + POM_MAX_BOMBS) out[147 + lane - 20] = (int32_t)tile[(POM_REC_BOMBS + lane - 20) * 16];
    if (lane >= 40 && lane < 40 + POM_MAX_BOMBS) {
        const uint32_t f = tile[(POM_REC_FLAMES + lane - 40) * 16];
        int32_t* o = out + 169 + 4 * (lane - 40);
        o[0] = (int32_t)(f & 0xFF);
        o[1] = (int32_t)((f >> 8) & 0xFF);
        o[2] = pom_sext8(f >> 16);
        o[3] = (int32_t)(f >> 24);
    }
    __threadfence_system(); /* every lane's stores have left before ... */
    __syncthreads();
    if (lane == 0) /* ... the word the host is polling changes */
        __hip_atomic_store(reinterpret_cast<uint32_t*>(p.io) + POM_ONE_SEQ, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* out: 6 arrays of `count` int32: done, winner, draw, alive, timeStep, ubflags */
__global__ void pom_status_kernel(const uint32_t* __restrict__ state, int64_t first, int64_t count, int64_t np, int32_t* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t* col = state + pom_rec_col(first + i);
    const uint32_t m = pom_rec_meta(col, POM_TILE_ENVS), m2 = pom_rec_meta2(col, POM_TILE_ENVS);
    const uint32_t st = (m2 >> 8) & 0xFF;
    out[0 * count + i] = (st & POM_ST_DONE) ? 1 : 0;
    out[1 * count + i] = (int)((st >> POM_ST_WINNER_SHIFT) & 7) - 1;
    out[2 * count + i] = (st & POM_ST_DRAW) ? 1 : 0;
    out[3 * count + i] = pom_sext8(m);
    out[4 * count + i] = (int32_t)col[POM_REC_TIMESTEP * POM_TILE_ENVS];
    out[5 * count + i] = (int32_t)(m2 >> 16);
}

/* POM_RESET_AT_END: out = 5 arrays of `count` int32: finished (the state's "restarted" mark), then winner, draw, length, alive
 * of the terminal record (array of structs; all-zero = no episode finished yet) */
__global__ void pom_results_kernel(const uint32_t* __restrict__ state, const uint32_t* __restrict__ terminal, int64_t first, int64_t count,
                                   int64_t np, int32_t* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t now = (pom_rec_meta2(state + pom_rec_col(first + i), POM_TILE_ENVS) >> 8) & 0xFF;
    const uint32_t* rec = terminal + (first + i) * POM_REC_DWORDS;
    const uint32_t st = (pom_rec_meta2(rec, 1) >> 8) & 0xFF;
    out[0 * count + i] = (now & POM_ST_RESTARTED) ? 1 : 0;
    out[1 * count + i] = (int)((st >> POM_ST_WINNER_SHIFT) & 7) - 1;
    out[2 * count + i] = (st & POM_ST_DRAW) ? 1 : 0;
    out[3 * count + i] = (int32_t)rec[POM_REC_TIMESTEP];
    out[4 * count + i] = (st & POM_ST_DONE) ? pom_sext8(pom_rec_meta(rec, 1)) : 0;
}

__global__ void pom_unpack_aos_kernel(const uint32_t* __restrict__ recs, int64_t first, int64_t count, int32_t* aos)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    pom_unpack_state(recs + (first + i) * POM_REC_DWORDS, 1, aos + i * (POM_STATE_BYTES / 4));
}

__global__ void pom_snapshot_kernel(const uint32_t* __restrict__ state, uint32_t* snap, int64_t np)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= np) return;
    const uint32_t* col = state + pom_rec_col(e);
    uint32_t* rec = snap + e * POM_REC_DWORDS; /* array of structs (restart_column): a dense record */
    for (int c = 0; c < 4 * POM_REC_BOARD_DWORDS; c++) pom_rec_set_cell(rec, 1, c, pom_rec_cell(col, POM_TILE_ENVS, c, (int)(e & 15)));
    for (int d = POM_REC_TIMESTEP; d < POM_REC_DWORDS; d++) {
        rec[d] = col[d * POM_TILE_ENVS];
    }
    pom_rec_set_meta(rec, 1, pom_rec_meta(rec, 1), pom_rec_meta2(rec, 1) & 0xFFu); /* a snapshot starts an episode: status and flags clear */
}

__global__ __launch_bounds__(1024) void pom_reduce_counters_kernel(const int64_t* __restrict__ wc, int64_t n_waves, int64_t* out)
{
    /* one workgroup of 1,024 lanes: 4,096 wavefront slots are four independent 32-byte loads per lane (the kernel sits at
     * the end of a caller's timed region, after the sub-streams' join: a 256-lane loop of 16 dependent trips took 5-10 us) */
    __shared__ long long part[POM_CNT_N][16];
    long long acc[POM_CNT_N] = {0, 0, 0, 0};
    for (int64_t w = threadIdx.x; w < n_waves; w += blockDim.x) {
        const longlong2 lo = *reinterpret_cast<const longlong2*>(wc + w * POM_CNT_N);
        const longlong2 hi = *reinterpret_cast<const longlong2*>(wc + w * POM_CNT_N + 2);
        acc[0] += lo.x; acc[1] += lo.y; acc[2] += hi.x; acc[3] += hi.y;
    }
    static_assert(POM_CNT_N == 4, "two 16-byte loads per slot");
    for (int k = 0; k < POM_CNT_N; k++)
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < POM_CNT_N; k++) part[k][threadIdx.x >> 6] = acc[k];
    __syncthreads();
    if (threadIdx.x < POM_CNT_N) {
        long long t = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += part[threadIdx.x][i];
        out[threadIdx.x] = t;
    }
}

#endif /* POM_KERNELS_H_ */
