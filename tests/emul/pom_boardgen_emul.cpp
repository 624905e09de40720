/*
 * pom_boardgen_emul.cpp — TEST-ONLY host build of the device board generator's pieces (pomcpp_amd/csrc/pom_boardgen_body.h),
 * glued together lane by lane the way pom_boardgen_wave does on the device (lane l: cells l and l+64; ballots = the wood
 * set; lanes 0..50: the other rows), to check them against oracle/pom_boardgen_oracle.c without a GPU.  Never linked into
 * the product.
 */
#include <cstring>

#include "pom_boardgen_body.h"

static void put_cell(uint32_t* rec, int c, int code)
{
    uint32_t& w = rec[POM_REC_BOARD + (c >> 2)];
    w = (w & ~(0xFFu << (8 * (c & 3)))) | ((uint32_t)(code & 0xFF) << (8 * (c & 3)));
}

extern "C" void pom_emul_boardgen(uint64_t seed, uint32_t env, uint32_t episode, void* state_1004)
{
    uint32_t rec[POM_REC_DWORDS];
    std::memset(rec, 0xA5, sizeof rec); /* whatever a finished game left behind: the generator must overwrite all of it */
    const uint32_t key = pom_board_key(seed, env, episode);
    uint64_t ballot[2] = {0, 0};
    for (int lane = 0; lane < 64; lane++) {
        for (int half = 0; half < 2; half++) {
            const int c = lane + 64 * half;
            if (c >= POM_CELLS) continue;
            const uint32_t kind = pom_board_cell_kind(key, c);
            put_cell(rec, c, pom_board_cell_code(kind));
            if (kind == 2u) ballot[half] |= 1ull << lane;
        }
        if (POM_REC_TIMESTEP + lane < POM_REC_DWORDS) rec[POM_REC_TIMESTEP + lane] = pom_fresh_row(POM_REC_TIMESTEP + lane);
    }
    pom_board_flags(key, ballot[0], ballot[1], [&](int c, int code) { put_cell(rec, c, code); });
    for (int a = 0; a < 4; a++) put_cell(rec, pom_corner_cell(a), POM_C_AGENT + a);
    rec[POM_REC_BOARD + 30] &= 0x000000FFu; /* the three bytes past cell 120 are not part of the record's content */
    std::memset(state_1004, 0, POM_STATE_BYTES);
    pom_unpack_state(rec, 1, (int32_t*)state_1004);
}
