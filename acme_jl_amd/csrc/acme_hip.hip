// acme_hip.hip -- C ABI (include/acme_hip.h) over the gfx950 kernels of acme_kernel.h.
//
// Host side is deliberately thin: pack the model once, keep per-instance state resident in
// HBM, launch one kernel per run! call on the caller's stream.  No CPU fallback exists:
// without a usable HIP device every compute entry point fails with ACME_ERR_NO_DEVICE.
#include <chrono>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/acme_hip.h"
#include "acme_kernels.h"
#include "acme_pack.h"
#include "acme_coop_kernel.h"

using namespace acme;

// LDS budgets the performance of the BASELINE workloads rests on (a CU has 160 KB: two blocks -- two waves per
// SIMD -- need 80 KB each, solution caches included)
#ifndef ACME_DEV_SHAPES
static_assert(sizeof(double) * (Shape<13, 29, 11, 11, 4, 1, 0, 1>::lds_doubles(false) +
                                INST_PER_BLOCK * Shape<13, 29, 11, 11, 4, 1, 0, 1>::CACHEI) <= 80 * 1024,
              "headline shape: two blocks per CU");
static_assert(sizeof(double) * (Shape<7, 14, 5, 11, 1, 1, 0, 1>::lds_doubles(true) +
                                INST_PER_BLOCK * Shape<7, 14, 5, 11, 1, 1, 0, 1>::CACHEI) <= 80 * 1024,
              "fixed-pot superover with 16 private images per block (BASELINE config 4): two blocks per CU, one round");
#endif

struct KernelEntry {
    Dims d;
    KernelFns lds, low;            // images / caches in LDS; the LOW-LDS variants (null where Shape::HAS_LOW is false)
    const void *fn_lane;
    int lds_shared, lds_per_inst, lds_low;  // doubles
    int state;                     // doubles of state per instance
    int cache_lds;                 // LDS doubles per instance of the solution caches (Shape::CACHEI)
    int lds_tab, lds_tab_low;      // doubles more per block when the element tables are per instance (KArgs::table_stride)
    int lds_lane_plain, lds_lane_caching;   // doubles, lane kernel (0: shape not supported by it)
    int (*launch_lane)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t);
};

template <class S> static int lane_lds(bool caching) {
    if constexpr (LaneShape<S>::supported) return LaneShape<S>::lds_doubles(caching);
    else return 0;
}
// (this translation unit instantiates no kernel: the entry points come from the parts)
template <class S> static KernelEntry make_entry(int index) {
    ShapeFns f;
    // (every part fills in what it holds of the shape: a shape's LOW-LDS variants can live in another unit than the rest)
    bool any = false;
    any |= acme_shape_fns_part0(index, &f); any |= acme_shape_fns_part1(index, &f); any |= acme_shape_fns_part2(index, &f);
    any |= acme_shape_fns_part3(index, &f); any |= acme_shape_fns_part4(index, &f); any |= acme_shape_fns_part5(index, &f);
    any |= acme_shape_fns_part6(index, &f); any |= acme_shape_fns_part7(index, &f);
    if (!any || !f.lds.fn) abort();
    return KernelEntry{Dims{S::NN, S::NQ, S::NP, S::NX, S::NU, S::NY, S::RARE ? 1 : 0, S::NSUB, S::NL}, f.lds, f.low, f.fn_lane,
                       S::lds_doubles(false), S::lds_doubles(true), S::lds_doubles_low(), S::STATE, S::CACHEI,
                       S::lds_doubles(true, true) - S::lds_doubles(true, false), S::lds_doubles_low(true) - S::lds_doubles_low(false),
                       lane_lds<S>(false), lane_lds<S>(true), f.launch_lane};
}

static const std::vector<KernelEntry> &kernel_table() {
    static const std::vector<KernelEntry> t = [] {
        std::vector<KernelEntry> v;
        int index = 0;
#define ACME_X(nn, nq, np, nx, nu, ny, rare, nsub, nl) v.push_back(make_entry<Shape<nn, nq, np, nx, nu, ny, rare, nsub, nl>>(index++));
        ACME_SHAPES(ACME_X)
#undef ACME_X
        return v;
    }();
    return t;
}

// the generic kernel (acme_generic.h): one lane per instance
__global__ __launch_bounds__(64) void acme_generic_kernel(GArgs A) {
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
    if (i < A.n_inst) gen_main(A, i);
}

// acme_batch_run_const: the full input rows of a time slice put together in HBM -- row k of sample t of instance i is the
// row's constant (mask bit k) or the next of the varying rows the caller handed over ([N][pitch][nuv])
struct ExpandArgs { double *dst; const double *uv, *uc; unsigned long long mask; long long n, T, pitch; int nu, nuv; };
__global__ __launch_bounds__(256) void acme_expand_kernel(ExpandArgs A) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= A.n * A.T) return;
    const long long i = idx / A.T, t = idx - i * A.T;
    const double *src = A.uv + (i * A.pitch + t) * A.nuv, *c = A.uc + i * A.nu;
    double *dst = A.dst + idx * A.nu;
    int v = 0;
    for (int k = 0; k < A.nu; ++k) dst[k] = (A.mask >> k & 1ull) ? c[k] : src[v++];
}

// placement of the waves by their measured cost (acme_balance.h): one thread per wave
__global__ __launch_bounds__(256) void acme_balance_weight_kernel(BalArgs A) {
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k < A.nu) bal_weight(A, k);
}
__global__ __launch_bounds__(256) void acme_balance_place_kernel(BalArgs A) {
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k < A.nu) bal_place(A, k);
}

static const KernelEntry *find_kernel(const Dims &d) {
    for (const auto &k : kernel_table())
        if (k.d.nn == d.nn && k.d.nq == d.nq && k.d.np == d.np && k.d.nx == d.nx && k.d.nu == d.nu && k.d.ny == d.ny && k.d.rare == d.rare && k.d.nsub == d.nsub && k.d.nl == d.nl)
            return &k;
    return nullptr;
}


// ------------------------------------------------------------------------------------------
// HIP device backend for acme_api.inc
// ------------------------------------------------------------------------------------------
namespace be {
using stream_t = hipStream_t;
using event_t = hipEvent_t;
static inline const char *err_string(int e) { return hipGetErrorString((hipError_t)e); }
static inline int device_count(int *n) { return (int)hipGetDeviceCount(n); }
static inline int set_device(int d) { return (int)hipSetDevice(d); }
static inline int get_device(int *d) { return (int)hipGetDevice(d); }
static inline int set_max_lds(const void *fn, int bytes) {
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
static inline int dmalloc(void **p, size_t n) { return (int)hipMalloc(p, n ? n : 8); }
static inline int dfree(void *p) { return p ? (int)hipFree(p) : 0; }
static inline int copy_h2d(void *d, const void *s, size_t n) { return (int)hipMemcpy(d, s, n, hipMemcpyHostToDevice); }
static inline int copy_d2h(void *d, const void *s, size_t n) { return (int)hipMemcpy(d, s, n, hipMemcpyDeviceToHost); }
static inline int copy_h2d_async(void *d, const void *s, size_t n, stream_t st) {
    return (int)hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st);
}
static inline int copy_d2h_async(void *d, const void *s, size_t n, stream_t st) {
    return (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st);
}
// strided host <-> device copies on a stream (a time slice of every instance's u / y rows)
static inline int copy2d_h2d_async(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, stream_t st) {
    return (int)hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyHostToDevice, st);
}
static inline int copy2d_d2h_async(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, stream_t st) {
    return (int)hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyDeviceToHost, st);
}
// (the callers tolerate a failure of these three -- memory that cannot be locked, a range the caller has freed -- and
// go on with another path: the runtime's sticky error record must not outlive the decision)
static inline int tolerated(hipError_t e) { if (e != hipSuccess) (void)hipGetLastError(); return (int)e; }
static inline int host_register(void *p, size_t n) { return tolerated(hipHostRegister(p, n, hipHostRegisterDefault)); }
static inline int host_unregister(void *p) { return tolerated(hipHostUnregister(p)); }
static inline int host_device_pointer(void **d, void *h) { return tolerated(hipHostGetDevicePointer(d, h, 0)); }
static inline int dmalloc_try(void **p, size_t n) { return tolerated(hipMalloc(p, n ? n : 8)); }
static inline int stream_wait_event(stream_t st, event_t e) { return (int)hipStreamWaitEvent(st, e, 0); }
static inline int stream_create_nonblocking(stream_t *st) { return (int)hipStreamCreateWithFlags(st, hipStreamNonBlocking); }
static inline bool stream_idle(stream_t st) { return hipStreamQuery(st) == hipSuccess; }
static inline int stream_destroy(stream_t st) { return st ? (int)hipStreamDestroy(st) : 0; }
static inline int device_sync() { return (int)hipDeviceSynchronize(); }
static inline int stream_sync(stream_t st) { return (int)hipStreamSynchronize(st); }
static inline int event_create(event_t *e) { return (int)hipEventCreate(e); }
static inline int event_destroy(event_t e) { return (int)hipEventDestroy(e); }
static inline int event_record(event_t e, stream_t st) { return (int)hipEventRecord(e, st); }
static inline int event_sync(event_t e) { return (int)hipEventSynchronize(e); }
static inline int event_elapsed(float *ms, event_t a, event_t b) { return (int)hipEventElapsedTime(ms, a, b); }
static inline int launch_generic(const GArgs &A, stream_t st) {
    return ACME_LAUNCH(acme_generic_kernel, dim3((unsigned)((A.n_inst + 63) / 64)), dim3(64), 0, st, A);
}
// the entry point of a launch shape (GArgs::coop_imgl / coop_nc)
static inline const void *coop_fn(const GArgs &A) {
    switch (A.coop_nc) {
    case 20: return acme_coop_fn_nc20(A.coop_imgl);
    case 24: return acme_coop_fn_nc24(A.coop_imgl);
    case 28: return acme_coop_fn_nc28(A.coop_imgl);
    case 32: return acme_coop_fn_nc32(A.coop_imgl);
    case -1: return acme_coop_fn_lds1(A.coop_imgl);
    case -2: return acme_coop_fn_lds2(A.coop_imgl);
    case -3: return acme_coop_fn_lds3(A.coop_imgl);
    case -4: return acme_coop_fn_lds4(A.coop_imgl);
    case COOP_WAVE64: return acme_coop_fn_wave64(A.coop_imgl);
    default: return coop_fns_of<0>(A.coop_imgl);
    }
}
// dynamic LDS beyond 64 KB has to be asked for: per entry point and DEVICE (the caller is on the batch's device), at batch
// creation and whenever the batch's launch shape changes -- not from the launch path, which worker threads of several
// batches share
static inline int coop_prepare(const GArgs &A, size_t lds_bytes) { return set_max_lds(coop_fn(A), (int)lds_bytes); }
static inline int launch_coop(const GArgs &A, size_t lds_bytes, stream_t st) {
    const long long waves = (A.n_inst + A.coop_gpw - 1) / A.coop_gpw;
    const dim3 grid((unsigned)((waves + A.coop_wpb - 1) / A.coop_wpb));
    GArgs args = A;
    void *params[] = {&args};
    return ACME_LAUNCH_FN(coop_fn(A), grid, dim3(64 * A.coop_wpb), lds_bytes, st, params);
}
static inline int launch_expand(double *dst, const double *uv, const double *uc, unsigned long long mask, long long n, long long T,
                                long long pitch, int nu, int nuv, stream_t st) {
    const ExpandArgs A{dst, uv, uc, mask, n, T, pitch, nu, nuv};
    return ACME_LAUNCH(acme_expand_kernel, dim3((unsigned)((n * T + 255) / 256)), dim3(256), 0, st, A);
}
static inline int launch_balance(const BalArgs &A, stream_t st) {
    const unsigned g = (unsigned)((A.nu + 255) / 256);
    const int rc = ACME_LAUNCH(acme_balance_weight_kernel, dim3(g), dim3(256), 0, st, A);
    return rc ? rc : ACME_LAUNCH(acme_balance_place_kernel, dim3(g), dim3(256), 0, st, A);
}
// a word of host memory the device can read while a kernel runs (streamed host runs: KArgs::u_ready)
static inline int flag_alloc(long long **h, const long long **d) {
    hipError_t e = hipHostMalloc((void **)h, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return (int)e;
    return (int)hipHostGetDevicePointer((void **)d, *h, 0);
}
static inline int flag_free(long long *h) { return h ? (int)hipHostFree(h) : 0; }
static inline bool kernels_run_async() { return true; }
static inline int cu_count(int *n) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceGetAttribute(n, hipDeviceAttributeMultiprocessorCount, d);
}
static inline std::mutex *run_mutex() { return nullptr; }     // HIP: runs of distinct batches are concurrent
}  // namespace be

#include "acme_api.inc"
