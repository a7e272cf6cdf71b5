// acme_emu.cpp -- the C ABI of include/acme_hip.h on top of the CPU wave emulator.
//
// TEST INFRASTRUCTURE ONLY (see wave_emu.h).  Shares acme_api.inc, acme_pack.h and
// acme_kernel.h with the HIP library; only the device backend differs.  "Device memory" is
// host memory, a "launch" runs the blocks one after another on fibers.
#include "wave_emu.h"

#include <chrono>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <sys/mman.h>

#include "../../include/acme_hip.h"
#include "../../acme_jl_amd/csrc/acme_kernel.h"
#include "../../acme_jl_amd/csrc/acme_lane_kernel.h"
#include "../../acme_jl_amd/csrc/acme_pack.h"
#include "../../acme_jl_amd/csrc/acme_coop.h"

using namespace acme;

namespace emu {
BlockCtx *g_blk = nullptr;
long long g_count_allmax = 0, g_count_shfl = 0, g_count_recip = 0;
struct CountPrinter { ~CountPrinter() { if (getenv("ACME_EMU_COUNTS")) fprintf(stderr, "emu counts: allmax16 %lld shfl16(double) %lld recip %lld\n", g_count_allmax, g_count_shfl, g_count_recip); } } g_count_printer;
int g_debug = getenv("ACME_EMU_DEBUG") ? atoi(getenv("ACME_EMU_DEBUG")) : 0;

asm(R"(
.text
.globl acme_emu_switch
.type acme_emu_switch,@function
acme_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size acme_emu_switch,.-acme_emu_switch
)");

static void fiber_main() {
    BlockCtx *b = g_blk;
    b->entry(b->entry_arg);
    // exit: this lane leaves its wave and the block
    int me = b->cur;
    Fiber &f = b->fibers[me];
    f.done = true;
    WaveSync &ws = b->waves[me >> 6];
    --ws.alive;
    --b->block_alive;
    if (ws.alive > 0 && ws.arrived == ws.alive) {  // the others were waiting for this lane only
        ws.arrived = 0;
        ++ws.generation;
    }
    if (b->block_alive > 0 && b->block_arrived == b->block_alive) {
        b->block_arrived = 0;
        ++b->block_generation;
    }
    for (;;) {  // hand control to any live fiber, or back to the scheduler
        int nxt = -1;
        for (int k = 1; k <= BLOCK; ++k) {
            int c = (me + k) % BLOCK;
            if (!b->fibers[c].done) { nxt = c; break; }
        }
        void *dummy;
        if (nxt < 0) acme_emu_switch(&dummy, b->sched_sp);
        b->cur = nxt;
        acme_emu_switch(&dummy, b->fibers[nxt].sp);
    }
}

static char *g_stacks = nullptr;

void run_block(int bid, void (*entry)(void *), void *arg) {
    if (!g_stacks) {
        g_stacks = (char *)mmap(nullptr, STACK_BYTES * BLOCK, PROT_READ | PROT_WRITE,
                                MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char *)MAP_FAILED) die("mmap of fiber stacks failed");
    }
    auto blk = std::make_unique<BlockCtx>();
    BlockCtx *b = blk.get();
    b->bid = bid;
    b->entry = entry;
    b->entry_arg = arg;
    for (int t = 0; t < BLOCK; ++t) {
        Fiber &f = b->fibers[t];
        f.tid = t;
        f.stack = g_stacks + (size_t)t * STACK_BYTES;
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
        void **sp = (void **)top;
        *--sp = nullptr;               // fake return address of fiber_main (never used)
        *--sp = (void *)&fiber_main;   // `ret` target of the first switch
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.sp = (void *)sp;
    }
    g_blk = b;
    b->cur = 0;
    acme_emu_switch(&b->sched_sp, b->fibers[0].sp);
    g_blk = nullptr;
}
}  // namespace emu

// ------------------------------------------------------------------------------------------
// kernel table
// ------------------------------------------------------------------------------------------
struct KernelFns {
    const void *fn = nullptr, *fn_jac = nullptr, *fn_solve = nullptr, *fn_stream = nullptr;
    int (*launch)(const KArgs &, unsigned grid, size_t lds_bytes, void *) = nullptr;
    int (*launch_stream)(const KArgs &, unsigned grid, size_t lds_bytes, void *) = nullptr;
    int (*launch_jac)(const KArgs &, unsigned grid, size_t lds_bytes, void *) = nullptr;
    int (*launch_solve)(const KArgs &, unsigned grid, size_t lds_bytes, void *) = nullptr;
};
struct KernelEntry {
    Dims d;
    KernelFns lds, low;
    const void *fn_lane;
    int lds_shared, lds_per_inst, lds_low, state, cache_lds;
    int lds_tab, lds_tab_low;      // doubles more per block when the element tables are per instance (KArgs::table_stride)
    int lds_lane_plain, lds_lane_caching;
    int (*launch_lane)(const KArgs &, unsigned grid, size_t lds_bytes, void *);
};

struct LaunchCtx {
    const KArgs *A;
    double *lds;
};

// KIND 0: run kernel, 1: Jacobian export, 2: lane-per-instance run kernel, 3: solve kernel, 4: run kernel of a streamed
// host-buffer run; LOW: the LOW-LDS variant
template <class S, int KIND, bool LOW> static void fiber_entry(void *p) {
    LaunchCtx *c = (LaunchCtx *)p;
    if constexpr (KIND == 0) wave_main<S, MODE_RUN, LOW>(*c->A, c->lds);
    else if constexpr (KIND == 1) { if constexpr (S::NN > 0) wave_main<S, MODE_JAC, LOW>(*c->A, c->lds); }
    else if constexpr (KIND == 3) { if constexpr (S::NN > 0) wave_main<S, MODE_SOLVE, LOW>(*c->A, c->lds); }
    else if constexpr (KIND == 4) { if constexpr (S::NU > 0) wave_main<S, MODE_RUN_STREAM, LOW>(*c->A, c->lds); }
    else { if constexpr (LaneShape<S>::supported) lane_main<S>(*c->A, c->lds); }
}

template <class S, int KIND, bool LOW = false> static int launch_any(const KArgs &A, unsigned grid, size_t lds_bytes, void *) {
    static_assert(emu::BLOCK == LANE_BLOCK && emu::BLOCK == WAVES_PER_BLOCK * 64, "one emulated block = one kernel block");
    std::vector<double> lds(lds_bytes / sizeof(double) + 64);
    LaunchCtx c{&A, lds.data()};
    for (unsigned b = 0; b < grid; ++b) {
        // poison the LDS with NaNs: a kernel reading uninitialised LDS into a result shows up
        for (auto &v : lds) v = std::nan("");
        emu::run_block((int)b, &fiber_entry<S, KIND, LOW>, &c);
    }
    return 0;
}
template <class S> static int lane_lds(bool caching) {
    if constexpr (LaneShape<S>::supported) return LaneShape<S>::lds_doubles(caching);
    else return 0;
}
static int emu_fn_tag;       // (the emulator has no kernel entry points: any non-null value marks "exists")
template <class S, bool LOW> static KernelFns make_fns() {
    KernelFns f;
    if constexpr (LOW && !S::HAS_LOW) return f;
    f.fn = &emu_fn_tag;
    f.launch = &launch_any<S, 0, LOW>;
    if constexpr (S::NU > 0) {
        f.fn_stream = &emu_fn_tag;
        f.launch_stream = &launch_any<S, 4, LOW>;
    }
    if constexpr (S::NN > 0) {
        f.fn_jac = f.fn_solve = &emu_fn_tag;
        f.launch_jac = &launch_any<S, 1, LOW>;
        f.launch_solve = &launch_any<S, 3, LOW>;
    }
    return f;
}
template <class S> static KernelEntry make_entry() {
    return KernelEntry{Dims{S::NN, S::NQ, S::NP, S::NX, S::NU, S::NY, S::RARE ? 1 : 0, S::NSUB, S::NL},
                       make_fns<S, false>(), make_fns<S, true>(), nullptr,
                       S::lds_doubles(false), S::lds_doubles(true), S::lds_doubles_low(), S::STATE, S::CACHEI,
                       S::lds_doubles(true, true) - S::lds_doubles(true, false), S::lds_doubles_low(true) - S::lds_doubles_low(false),
                       lane_lds<S>(false), lane_lds<S>(true), &launch_any<S, 2>};
}

static const std::vector<KernelEntry> &kernel_table() {
    static const std::vector<KernelEntry> t = {
#define ACME_X(nn, nq, np, nx, nu, ny, rare, nsub, nl) make_entry<Shape<nn, nq, np, nx, nu, ny, rare, nsub, nl>>(),
        ACME_EMU_SHAPES(ACME_X)
#undef ACME_X
    };
    return t;
}

static const KernelEntry *find_kernel(const Dims &d) {
    for (const auto &k : kernel_table())
        if (k.d.nn == d.nn && k.d.nq == d.nq && k.d.np == d.np && k.d.nx == d.nx && k.d.nu == d.nu && k.d.ny == d.ny && k.d.rare == d.rare && k.d.nsub == d.nsub && k.d.nl == d.nl)
            return &k;
    return nullptr;
}

// ------------------------------------------------------------------------------------------
// host-memory "device" backend
// ------------------------------------------------------------------------------------------
namespace be {
using stream_t = void *;
struct Ev { std::chrono::steady_clock::time_point t; };
using event_t = Ev *;
static inline const char *err_string(int) { return "emulator error"; }
static inline int device_count(int *n) { *n = 1; return 0; }
static inline int set_device(int) { return 0; }
static inline int get_device(int *d) { *d = 0; return 0; }
static inline int set_max_lds(const void *, int) { return 0; }
static inline int dmalloc(void **p, size_t n) { *p = malloc(n ? n : 8); return *p ? 0 : 1; }
static inline int dfree(void *p) { free(p); return 0; }
static inline int dmalloc_try(void **p, size_t n) { return dmalloc(p, n); }
static inline int stream_wait_event(void *, struct Ev *) { return 0; }
static inline int copy_h2d(void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
static inline int copy_d2h(void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
static inline int copy_h2d_async(void *d, const void *s, size_t n, stream_t) { memcpy(d, s, n); return 0; }
static inline int copy_d2h_async(void *d, const void *s, size_t n, stream_t) { memcpy(d, s, n); return 0; }
static inline int copy2d_h2d_async(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, stream_t) {
    for (size_t r = 0; r < height; ++r) memcpy((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return 0;
}
static inline int copy2d_d2h_async(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, stream_t st) {
    return copy2d_h2d_async(d, dpitch, s, spitch, width, height, st);
}
static int g_registered = 0;     // (ranges currently "page-locked": the emulator only counts them, tests read the count)
static inline int host_register(void *, size_t) { ++g_registered; return 0; }
static inline int host_unregister(void *) { --g_registered; return 0; }
static inline int host_device_pointer(void **d, void *h) { *d = h; return 0; }
static inline int stream_create_nonblocking(stream_t *st) { *st = nullptr; return 0; }
static inline bool stream_idle(stream_t) { return true; }
static inline int stream_destroy(stream_t) { return 0; }
static inline int device_sync() { return 0; }
static inline int stream_sync(stream_t) { return 0; }
static inline int event_create(event_t *e) { *e = new Ev(); return 0; }
static inline int event_destroy(event_t e) { delete e; return 0; }
static inline int event_record(event_t e, stream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
static inline int event_sync(event_t) { return 0; }
static inline int event_elapsed(float *ms, event_t a, event_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return 0;
}
// the generic kernel has no cross-lane operation: its "lanes" run one after the other
static inline int launch_generic(const GArgs &A, stream_t) {
    for (long long i = 0; i < A.n_inst; ++i) gen_main(A, i);
    return 0;
}
// the mid-size kernel (acme_coop.h): an emulated block (four waves) is one of its blocks of GArgs::coop_wpb waves -- the
// waves beyond that leave at once
struct CoopLaunch { const GArgs *A; double *lds; int bid; };
static void coop_fiber_entry(void *p) {
    CoopLaunch *c = (CoopLaunch *)p;
    const int tid = wv::tid(), wave = tid >> 6, wpb = c->A->coop_wpb;
    if (wave >= wpb) return;
    double *lds = c->lds;
    const int wg = c->bid * wpb + wave, lane = tid & 63;
    const bool img = c->A->coop_imgl != 0;
    switch (c->A->coop_nc) {
    case 20: img ? coop_main<true, 20>(*c->A, lds, wave, wg, lane) : coop_main<false, 20>(*c->A, lds, wave, wg, lane); break;
    case 24: img ? coop_main<true, 24>(*c->A, lds, wave, wg, lane) : coop_main<false, 24>(*c->A, lds, wave, wg, lane); break;
    case 28: img ? coop_main<true, 28>(*c->A, lds, wave, wg, lane) : coop_main<false, 28>(*c->A, lds, wave, wg, lane); break;
    case 32: img ? coop_main<true, 32>(*c->A, lds, wave, wg, lane) : coop_main<false, 32>(*c->A, lds, wave, wg, lane); break;
    case -1: img ? coop_main<true, -1>(*c->A, lds, wave, wg, lane) : coop_main<false, -1>(*c->A, lds, wave, wg, lane); break;
    case -2: img ? coop_main<true, -2>(*c->A, lds, wave, wg, lane) : coop_main<false, -2>(*c->A, lds, wave, wg, lane); break;
    case -3: img ? coop_main<true, -3>(*c->A, lds, wave, wg, lane) : coop_main<false, -3>(*c->A, lds, wave, wg, lane); break;
    case -4: img ? coop_main<true, -4>(*c->A, lds, wave, wg, lane) : coop_main<false, -4>(*c->A, lds, wave, wg, lane); break;
    case COOP_WAVE64: img ? coop_main<true, COOP_WAVE64>(*c->A, lds, wave, wg, lane) : coop_main<false, COOP_WAVE64>(*c->A, lds, wave, wg, lane); break;
    default: img ? coop_main<true, 0>(*c->A, lds, wave, wg, lane) : coop_main<false, 0>(*c->A, lds, wave, wg, lane); break;
    }
}
static inline int coop_prepare(const GArgs &, size_t) { return 0; }
static inline int launch_coop(const GArgs &A, size_t lds_bytes, stream_t) {
    std::vector<double> lds(lds_bytes / sizeof(double) + 64);
    const long long waves = (A.n_inst + A.coop_gpw - 1) / A.coop_gpw;
    for (long long b = 0; b * A.coop_wpb < waves; ++b) {
        for (auto &v : lds) v = std::nan("");
        CoopLaunch c{&A, lds.data(), (int)b};
        emu::run_block((int)b, &coop_fiber_entry, &c);
    }
    return 0;
}
// acme_batch_run_const: the full input rows of a time slice, put together
static inline int launch_expand(double *dst, const double *uv, const double *uc, unsigned long long mask, long long n, long long T,
                                long long pitch, int nu, int nuv, stream_t) {
    for (long long i = 0; i < n; ++i)
        for (long long t = 0; t < T; ++t) {
            int v = 0;
            for (int k = 0; k < nu; ++k)
                dst[(i * T + t) * nu + k] = (mask >> k & 1ull) ? uc[i * nu + k] : uv[(i * pitch + t) * nuv + v++];
        }
    return 0;
}
// placement of the waves by their measured cost (acme_balance.h): the two passes, one "thread" after the other
static inline int launch_balance(const BalArgs &A, stream_t) {
    for (int k = 0; k < A.nu; ++k) bal_weight(A, k);
    for (int k = 0; k < A.nu; ++k) bal_place(A, k);
    return 0;
}
// streamed host runs: a launch here is synchronous -- the host copies everything first (acme_api.inc)
static inline int flag_alloc(long long **h, const long long **d) { *h = (long long *)calloc(8, 8); *d = *h; return *h ? 0 : 1; }
static inline int flag_free(long long *h) { free(h); return 0; }
static inline bool kernels_run_async() { return false; }
// (ACME_EMU_CUS: a small "chip", so that batches of a few dozen instances have two rounds of blocks; without it the
// number of compute units is unknown and the launcher places nothing -- full waves cost the emulator least)
static inline int cu_count(int *n) {
    const char *e = getenv("ACME_EMU_CUS");
    *n = e ? atoi(e) : 0;
    return 0;
}
// the fiber scheduler below is not re-entrant: asynchronous runs (acme_batch_run_async) take turns
static inline std::mutex *run_mutex() { static std::mutex m; return &m; }
}  // namespace be

#include "../../acme_jl_amd/csrc/acme_api.inc"
