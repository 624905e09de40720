// general_test.cpp — the reference's "[general]" vectors for FixedQueue (unit_test/bboard/general_test.cpp:8-61: fill ten, pop three,
// RemoveAt in the middle / at the head / at the tail, with the ring starting at index 0, 5 and 2) run against the FixedQueue that
// drop-in agents compile against (include/pom_bboard.hpp) — the queue semantics SURVEY.md §8(c) lists as pinning the path (stale
// slots, RemoveAt's shifting and the modular index are what Q1 / Q2 rest on).  No GPU needed.  Prints "general ok".
#include <cstdio>
#include <memory>

#include "pom_bboard.hpp"

static int fails = 0;
#define EXPECT(c)                                                           \
    do {                                                                    \
        if (!(c)) {                                                         \
            std::printf("EXPECT failed line %d: %s\n", __LINE__, #c);      \
            fails++;                                                        \
        }                                                                   \
    } while (0)

static void exercise(int start_index)
{
    auto q = std::make_unique<bboard::FixedQueue<bboard::Bomb, 10>>();
    q->index = start_index;
    for (int v = 0; v < 10; v++) {  // NextPos + count++, as PlantBomb writes a slot
        q->NextPos() = v;
        q->count++;
    }
    EXPECT(q->count == 10);
    EXPECT(q->RemainingCapacity() == 0);
    for (int k = 0; k < 3; k++) q->PopElem();
    EXPECT(q->count == 7);
    EXPECT((*q)[0] == 3);
    q->RemoveAt(5);  // live range 3 4 5 6 7 8 9 -> 3 4 5 6 7 9
    EXPECT(q->count == 6);
    EXPECT((*q)[4] == 7);
    EXPECT((*q)[5] == 9);
    q->RemoveAt(0);  // -> 4 5 6 7 9
    EXPECT((*q)[0] == 4);
    q->RemoveAt(4);  // -> 4 5 6 7
    EXPECT(q->count == 4);
    EXPECT((*q)[3] == 7);
    // what the tick's restatement relies on beyond the reference's assertions: RemoveAt leaves the slot behind the live range
    // holding a stale copy, and the physical index never leaves 0 .. TSize-1
    EXPECT(q->index >= 0 && q->index < 10);
    EXPECT(q->queue[(q->index + q->count) % 10] == 9);
}

int main()
{
    for (int idx : {0, 5, 2}) exercise(idx);
    static_assert(sizeof(bboard::FixedQueue<bboard::Bomb, 20>) == 88, "bombs queue: 20 slots + index + count (bboard.hpp:115-188)");
    static_assert(sizeof(bboard::State) == 1004, "State layout");
    if (fails) return 1;
    std::printf("general ok\n");
    return 0;
}
