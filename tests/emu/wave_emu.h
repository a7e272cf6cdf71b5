// wave_emu.h -- CPU wave emulator for the kernels of acme_jl_amd/csrc/acme_kernel.h.
//
// TEST INFRASTRUCTURE ONLY.  Lets the GPU-less unit tests execute the *unmodified* device
// source: every GPU thread of a 256-thread block becomes a fiber (hand-rolled x86-64
// context switch) on one OS thread, and the cross-lane primitives of namespace wv
// (row_newbcast / row_ror / ds_bpermute / ballot) are implemented as exchanges through a
// per-wave buffer with a wave-wide rendezvous.  The rendezvous also checks that every lane
// of the wave reaches the same cross-lane call site -- i.e. that cross-lane operations only
// occur in wave-uniform control flow, which the DPP-based kernel relies on.
// It is never linked into libacme_hip.so and never loaded by the acme_jl_amd package.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define ACME_DEV inline
#define ACME_LAMBDA
#define ACME_HD
#define ACME_DBG(fmt, ...) do { if (emu::g_debug) fprintf(stderr, fmt "\n", __VA_ARGS__); } while (0)

namespace emu {

constexpr int BLOCK = 256;
constexpr size_t STACK_BYTES = 512 * 1024;

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    int tid = 0;
    bool done = false;
    unsigned parity = 0;  // exchange-buffer parity, identical across a wave by construction
};

struct WaveSync {
    int arrived = 0;
    unsigned generation = 0;
    int alive = 64;
    // double-buffered exchange slots
    uint64_t slot[2][64];
    int site[2][64];
};

struct BlockCtx {
    Fiber fibers[BLOCK];
    WaveSync waves[BLOCK / 64];
    int block_arrived = 0;
    unsigned block_generation = 0;
    int block_alive = BLOCK;
    int bid = 0;
    int cur = 0;
    void *sched_sp = nullptr;
    void (*entry)(void *) = nullptr;
    void *entry_arg = nullptr;
};

extern BlockCtx *g_blk;
extern int g_debug;
extern long long g_count_allmax, g_count_shfl, g_count_recip;
extern "C" void acme_emu_switch(void **save_sp, void *load_sp);

[[noreturn]] inline void die(const char *msg) {
    fprintf(stderr, "wave emulator: %s\n", msg);
    abort();
}

inline void switch_to(int next) {
    BlockCtx *b = g_blk;
    int prev = b->cur;
    if (next == prev) return;
    b->cur = next;
    acme_emu_switch(&b->fibers[prev].sp, b->fibers[next].sp);
}

// rendezvous of all live lanes of the calling lane's wave
inline void wave_sync() {
    BlockCtx *b = g_blk;
    int me = b->cur, w = me >> 6;
    WaveSync &ws = b->waves[w];
    unsigned gen = ws.generation;
    if (++ws.arrived == ws.alive) {
        ws.arrived = 0;
        ++ws.generation;
        return;
    }
    int spins = 0;
    while (ws.generation == gen) {
        int nxt = me;
        for (int k = 1; k <= 64; ++k) {
            int c = (w << 6) + ((me + k) & 63);
            if (!b->fibers[c].done) { nxt = c; break; }
        }
        if (nxt == me || ++spins > 100000) die("deadlock: cross-lane operation in divergent control flow");
        switch_to(nxt);
    }
}

inline void block_sync() {
    BlockCtx *b = g_blk;
    int me = b->cur;
    unsigned gen = b->block_generation;
    if (++b->block_arrived == b->block_alive) {
        b->block_arrived = 0;
        ++b->block_generation;
        return;
    }
    int spins = 0;
    while (b->block_generation == gen) {
        int nxt = me;
        for (int k = 1; k <= BLOCK; ++k) {
            int c = (me + k) % BLOCK;
            if (!b->fibers[c].done) { nxt = c; break; }
        }
        if (nxt == me || ++spins > 1000000) die("deadlock in block_sync");
        switch_to(nxt);
    }
}

// exchange: every lane publishes `v`, then reads the value published by lane `src`
inline uint64_t exchange(uint64_t v, int src_lane, int site) {
    BlockCtx *b = g_blk;
    int me = b->cur, w = me >> 6, lane = me & 63;
    Fiber &f = b->fibers[me];
    unsigned par = f.parity & 1;
    f.parity++;
    WaveSync &ws = b->waves[w];
    ws.slot[par][lane] = v;
    ws.site[par][lane] = site;
    wave_sync();
    if (ws.site[par][src_lane] != site) die("lanes of one wave reached different cross-lane call sites");
    return ws.slot[par][src_lane];
}

inline uint64_t ballot_bits(bool p, int site) {
    BlockCtx *b = g_blk;
    int me = b->cur, w = me >> 6, lane = me & 63;
    Fiber &f = b->fibers[me];
    unsigned par = f.parity & 1;
    f.parity++;
    WaveSync &ws = b->waves[w];
    ws.slot[par][lane] = p ? 1 : 0;
    ws.site[par][lane] = site;
    wave_sync();
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) {
        if (b->fibers[(w << 6) + l].done) continue;
        if (ws.site[par][l] != site) die("ballot reached from different call sites");
        if (ws.slot[par][l]) m |= (1ull << l);
    }
    return m;
}

inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

// run one block of `BLOCK` fibers, each executing entry(arg)
void run_block(int bid, void (*entry)(void *), void *arg);

}  // namespace emu

namespace wv {
ACME_DEV constexpr bool lockstep() { return false; }
ACME_DEV int tid() { return emu::g_blk->fibers[emu::g_blk->cur].tid; }
ACME_DEV int bid() { return emu::g_blk->bid; }
ACME_DEV void block_sync() { emu::block_sync(); }
ACME_DEV void lds_order() { emu::wave_sync(); }
ACME_DEV void wave_fence() { emu::wave_sync(); }  // lanes run one after another here: LDS hand-offs need a rendezvous
template <int K> ACME_DEV double bcast16(double v) {
    int lane = tid() & 63;
    return emu::u2d(emu::exchange(emu::d2u(v), (lane & ~15) + K, 100 + K));
}
template <int K> ACME_DEV int bcast16(int v) {
    int lane = tid() & 63;
    return (int)(int64_t)emu::exchange((uint64_t)(int64_t)v, (lane & ~15) + K, 200 + K);
}
template <int K, bool SAFE> ACME_DEV void fmac_bcast_self(double &acc, double mul) { acc = fma(bcast16<K>(acc), mul, acc); }
template <int I, int N, int M> ACME_DEV void fmac_self_chain_(double &acc, const double (&mul)[M]) {
    if constexpr (I < N) { fmac_bcast_self<I, true>(acc, mul[I]); fmac_self_chain_<I + 1, N, M>(acc, mul); }
}
template <int N, int M> ACME_DEV void fmac_self_chain(double &acc, const double (&mul)[M]) { fmac_self_chain_<0, N, M>(acc, mul); }
template <int K, bool SAFE> ACME_DEV double bcast16_ordered(double v) { return bcast16<K>(v); }
template <int K> ACME_DEV void fmac_bcast(double &acc, double src, double mul) { acc = fma(bcast16<K>(src), mul, acc); }
template <int I, int N, int OFF, int M> ACME_DEV void fmac_bcast_chain_(double &acc, double src, const double (&mul)[M]) {
    if constexpr (I < N) { fmac_bcast<I>(acc, src, mul[OFF + I]); fmac_bcast_chain_<I + 1, N, OFF, M>(acc, src, mul); }
}
template <int N, bool WAIT, int OFF = 0, int M> ACME_DEV void fmac_bcast_chain(double &acc, double src, const double (&mul)[M]) { fmac_bcast_chain_<0, N, OFF, M>(acc, src, mul); }
template <int I, int L0, int N, int MB, int M> ACME_DEV void fmac_bcast_chain_from_(double &acc, double src, const double (&mul)[M]) {
    if constexpr (I < N) { fmac_bcast<L0 + I>(acc, src, mul[MB + I]); fmac_bcast_chain_from_<I + 1, L0, N, MB, M>(acc, src, mul); }
}
template <int L0, int N, bool WAIT, int MB = 0, int M> ACME_DEV void fmac_bcast_chain_from(double &acc, double src, const double (&mul)[M]) { fmac_bcast_chain_from_<0, L0, N, MB, M>(acc, src, mul); }
template <int I, int L0, int N, int M> ACME_DEV void fmac_self_chain_from_(double &acc, const double (&mul)[M]) {
    if constexpr (I < N) { fmac_bcast_self<L0 + I, true>(acc, mul[I]); fmac_self_chain_from_<I + 1, L0, N, M>(acc, mul); }
}
template <int L0, int N, int M> ACME_DEV void fmac_self_chain_from(double &acc, const double (&mul)[M]) { fmac_self_chain_from_<0, L0, N, M>(acc, mul); }
ACME_DEV void dpp_wait() {}
ACME_DEV bool lanes(unsigned long long mask);
ACME_DEV double recip(double d);
template <int K, bool SAFE>
ACME_DEV void gj_step_head(double ak, double &dinv, unsigned long long &pivlanes, double &nlm, double &vmx, double &frz) {
    const double piv = bcast16<K>(ak);
    const double inv = recip(piv);
    nlm = ak * -inv;
    if (lanes(pivlanes)) { dinv = inv; nlm = 0.0; frz = vmx; }
    pivlanes <<= 1;
    vmx = fmax(vmx, fabs(nlm));      // (v_max_f64: a NaN multiplier is ignored here; the result check catches it)
}
template <int K, int CNT, bool SAFE>
ACME_DEV void gj_step(double ak, double &dinv, unsigned long long &pivlanes, double &nlm, double &vmx, double &frz, double *const (&rp)[CNT]) {
    gj_step_head<K, SAFE>(ak, dinv, pivlanes, nlm, vmx, frz);
    for (int j = 0; j < CNT; ++j) *rp[j] = fma(bcast16<K>(*rp[j]), nlm, *rp[j]);
}
// two rows per lane (acme_coop.h): step head, row updates and replay of the Gauss-Jordan elimination -- the operations of
// the asm statements in acme_wave_hip.h, one by one (recip() here is the correctly rounded 1 / x)
template <int K> ACME_DEV void gj2_head(double &ak, double &ao, double &dinv, double &vmxk, double &vmxo, double &frz) {
    const double piv = bcast16<K>(ak);
    const double inv = recip(piv);
    ak = ak * -inv;
    ao = ao * -inv;
    if (lanes((1ull << K) * 0x0001000100010001ull)) { dinv = inv; ak = 0.0; frz = vmxk; }
    vmxk = fmax(vmxk, fabs(ak));
    vmxo = fmax(vmxo, fabs(ao));
}
template <int K, int M> ACME_DEV void gj2_update(double nlk, double nlo, double *const (&kp)[M], double *const (&op)[M]) {
    for (int j = 0; j < M; ++j) {
        const double b = bcast16<K>(*kp[j]);
        *op[j] = fma(b, nlo, *op[j]);
        *kp[j] = fma(b, nlk, *kp[j]);
    }
}
template <int K0, int CNT, int C0, int M> ACME_DEV void replay2_seg(double &xk, double &xo, const double (&mk)[M], const double (&mo)[M]) {
    if constexpr (CNT > 0) {
        const double b = bcast16<K0 % 16>(xk);
        xo = fma(b, mo[C0], xo);
        xk = fma(b, mk[C0], xk);
        replay2_seg<K0 + 1, CNT - 1, C0 + 1>(xk, xo, mk, mo);
    }
}
template <int R> ACME_DEV double ror16(double v) {
    int lane = tid() & 63;
    // row_ror:R -- lane i receives the value of lane (i - R) mod 16 of its row
    return emu::u2d(emu::exchange(emu::d2u(v), (lane & ~15) + ((lane - R) & 15), 300 + R));
}
ACME_DEV double allmax16(double v) {
    if ((tid() & 63) == 0) emu::g_count_allmax++;
    v = fmax(v, ror16<8>(v));
    v = fmax(v, ror16<4>(v));
    v = fmax(v, ror16<2>(v));
    v = fmax(v, ror16<1>(v));
    return v;
}
ACME_DEV double allmax16_nn(double v) { return allmax16(v); }
ACME_DEV double allmin16(double v) {
    v = fmin(v, ror16<8>(v));
    v = fmin(v, ror16<4>(v));
    v = fmin(v, ror16<2>(v));
    v = fmin(v, ror16<1>(v));
    return v;
}
ACME_DEV unsigned long long lanev64(unsigned long long v, int lane) { return emu::exchange(v, lane & 63, 701); }
ACME_DEV unsigned long long first64(unsigned long long v) { return v; }          // (every lane holds it alike)
template <int K> ACME_DEV double lane64(double v) { return emu::u2d(emu::exchange(emu::d2u(v), K, 600 + K)); }
ACME_DEV double allmax64(double v) {
    v = allmax16(v);
    return fmax(fmax(lane64<0>(v), lane64<16>(v)), fmax(lane64<32>(v), lane64<48>(v)));
}
ACME_DEV double allmin64(double v) {
    v = allmin16(v);
    return fmin(fmin(lane64<0>(v), lane64<16>(v)), fmin(lane64<32>(v), lane64<48>(v)));
}
ACME_DEV double allsum16(double v) {
    v += ror16<8>(v);
    v += ror16<4>(v);
    v += ror16<2>(v);
    v += ror16<1>(v);
    return v;
}
ACME_DEV int shfl16(int v, int src) {
    int lane = tid() & 63;
    return (int)(int64_t)emu::exchange((uint64_t)(int64_t)v, (lane & ~15) + (src & 15), 400);
}
ACME_DEV double shfl16(double v, int src) {
    int lane = tid() & 63;
    if (lane == 0) emu::g_count_shfl++;
    return emu::u2d(emu::exchange(emu::d2u(v), (lane & ~15) + (src & 15), 401));
}
ACME_DEV unsigned long long ballot(bool p) { return emu::ballot_bits(p, 500); }
ACME_DEV int ffs32(int v) { return __builtin_ffs(v); }
ACME_DEV double recip(double d) { if ((tid() & 63) == 0) emu::g_count_recip++; return 1.0 / d; }
ACME_DEV double keep(double v) { return v; }
ACME_DEV double settle(double v) { return v; }
ACME_DEV int keepi(int v) { return v; }
ACME_DEV void touch(double) {}
ACME_DEV int opaque(int v) { return v; }
// (streamed host runs: the emulator launches synchronously, the host has copied everything before)
ACME_DEV long long load_system(const long long *p) { return *p; }
ACME_DEV void acquire_system() {}
ACME_DEV void nap() {}
ACME_DEV unsigned long long pin(unsigned long long m) { return m; }
ACME_DEV double sconst(double v) { return v; }
ACME_DEV double clamp_s(double k, double lo, double hi) { return fmin(fmax(k, lo), hi); }
struct ExpTab {
    double v[16];
    double operator[](int i) const { return v[i]; }
};
ACME_DEV ExpTab load_exp_tab() {
    return ExpTab{{1.4426950408889634, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
                   1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07,
                   2.7557319223985893e-06, 2.48015873015873e-05, 1.984126984126984e-04, 1.388888888888889e-03,
                   8.333333333333333e-03, 4.1666666666666664e-02, 1.6666666666666666e-01, -2100.0, 2100.0}};
}
template <class T> ACME_DEV const T *uniform_ro(const T *p) { return p; }
struct pair_t { double lo, hi; };
ACME_DEV pair_t ld2(const double *p) { return pair_t{p[0], p[1]}; }
ACME_DEV void st2(double *p, double lo, double hi) { p[0] = lo; p[1] = hi; }
ACME_DEV void sched_fence() {}
ACME_DEV void lds_add(long long *p, long long v) { *p += v; }
ACME_DEV void lds_max(long long *p, long long v) { if (v > *p) *p = v; }
ACME_DEV bool lanes(unsigned long long mask) { return (mask >> (tid() & 63)) & 1ull; }
template <unsigned ROW16> ACME_DEV bool lanes_here() { return lanes((unsigned long long)(ROW16 & 0xFFFFu) * 0x0001000100010001ull); }
}  // namespace wv
