// acme_jl_amd -- placement of a launch's waves by their measured cost (acme_batch_set_balance).
//
// A launch of the 16-lane kernel ends with its slowest wave.  Every wave carries its four instances through all T
// samples (the state lives in its registers and its block's LDS), two blocks share a compute unit, and wave i of
// either block runs on the unit's SIMD i: the SIMD that happens to get the two most expensive waves finishes last while
// the others idle.  On the bench grid the most expensive wave needs 20 ... 60 % more Newton passes than the average one
// during the first second of a signal, 8 % after three seconds and less and less as the solution caches fill:
// dealing the waves in pairs of heavy and light is worth 3.4 % of a launch at t = 3 s (tools/balance_probe.py),
// 1.6 % over seconds 1-4 of the bench (bench.py --warmup 1 --steps 3: 324.1 -> 319.1 ms per step, twice) and nothing
// once the signal has run for 5 s (300.6 / 300.4 ms).
// What an instance costs changes slowly along a signal, so the previous launches say what the next one will cost: two
// tiny kernels on the launch's own stream (no host synchronisation; works in asynchronous pipelines)
//   bal_weight   wave k's weight = the largest number of Newton iterations one of its instances needed since the last
//                balancing (the wave iterates until its last instance has converged),
//   bal_place    its rank among all waves (heaviest first, ties by index: deterministic), and from the rank its slot,
// write the slot -> instance map the run kernel reads (KArgs::inst_map, the same indirection the isolation of slow
// instances uses).  Waves stay together: what an instance computes does not depend on where it runs (bit-identical
// results, tests/test_emu_parity.py::test_emulated_balance_is_invisible).
// Slots: blocks are dispatched in order, one per compute unit, then a second round on top of the first -- block b and
// block b + CUs share a unit (of the pairings probed, this one gave the most).  With no more than two rounds the
// heaviest waves fill the first round in order and the lightest wave goes on top of the heaviest; with more rounds the
// order is plain heaviest-first (the dispatcher then places each block where a unit has become free).
//
// Slots can be empty (inst_map = -1): those an incomplete last wave leaves, and -- a developer's knob, ACME_WAVE_DENSITY --
// those of waves deliberately filled with two or one instance (`per`); see wave_density in acme_api.inc for why the
// library never does that by itself.
#pragma once
#include "acme_common.h"

namespace acme {

struct BalArgs {
    const long long *report;    // [n][RW_WORDS]
    long long *prev;            // [n]: RW_ITERS_TOTAL at the last placement
    unsigned *weight;           // [nu]
    int *map;                   // [nu * BAL_SLOTS]: slot -> instance, -1: empty
    long long n;                // instances of the batch
    int per;                    // instances per wave: 1, 2 or 4
    int nu;                     // waves = ceil(n / per)
    int first_round;            // wave slots of one round of blocks (waves per block x compute units)
};

constexpr int BAL_SLOTS = 4;    // instance slots of a wave of the 16-lane kernel (64 lanes / GROUP)

// thread k < nu
ACME_HD inline void bal_weight(const BalArgs &A, int k) {
    long long w = 0;
    for (int j = 0; j < A.per; ++j) {
        const long long i = (long long)k * A.per + j;
        if (i >= A.n) break;
        const long long it = A.report[i * RW_WORDS + RW_ITERS_TOTAL];
        long long d = it - A.prev[i];
        if (d < 0) d = it;                    // (the report was reset in between)
        A.prev[i] = it;
        w = d > w ? d : w;
    }
    A.weight[k] = w > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)w;
}

ACME_HD inline int bal_slot(int rank, int nu, int first_round) {
    if (first_round > 0 && nu > first_round && nu <= 2 * first_round)
        return rank < first_round ? rank : first_round + (nu - 1 - rank);
    return rank;
}

ACME_HD inline void bal_place(const BalArgs &A, int k) {
    const unsigned wk = A.weight[k];
    int rank = 0;
    for (int j = 0; j < A.nu; ++j) {
        const unsigned wj = A.weight[j];
        rank += (wj > wk || (wj == wk && j < k)) ? 1 : 0;
    }
    const int q = bal_slot(rank, A.nu, A.first_round);
    for (int j = 0; j < BAL_SLOTS; ++j) {
        const long long i = (long long)k * A.per + j;
        A.map[(long long)q * BAL_SLOTS + j] = (j < A.per && i < A.n) ? (int)i : -1;
    }
}

}  // namespace acme
