// acme_hip.hip -- C ABI (include/acme_hip.h) over the gfx950 kernels of acme_kernel.h.
//
// Host side is deliberately thin: pack the model once, keep per-instance state resident in
// HBM, launch one kernel per run! call on the caller's stream.  No CPU fallback exists:
// without a usable HIP device every compute entry point fails with ACME_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/acme_hip.h"
#include "acme_wave_hip.h"
#include "acme_kernel.h"
#include "acme_lane_kernel.h"
#include "acme_pack.h"

using namespace acme;

// ------------------------------------------------------------------------------------------
// kernels: one instantiation per shape of acme_shapes.h
// ------------------------------------------------------------------------------------------
template <class S>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, 2) void acme_run_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    wave_main<S>(A, acme_lds);
}

// the small companion kernel: get_extrapolation_jacobian for every instance (wave_main MODE_JAC)
template <class S>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, 2) void acme_jac_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    wave_main<S, MODE_JAC>(A, acme_lds);
}

// ... and solve(solver, p), once per instance (wave_main MODE_SOLVE): the solver-plugin contract
template <class S>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, 2) void acme_solve_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    wave_main<S, MODE_SOLVE>(A, acme_lds);
}

// run! for small models, one lane per instance (acme_lane_kernel.h): one wave per SIMD is all these
// batches offer, so the kernel is built for the shortest dependent chain per sample, not for occupancy
template <class S>
__global__ __launch_bounds__(LANE_BLOCK, 1) void acme_lane_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    lane_main<S>(A, acme_lds);
}

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
    const void *fn, *fn_jac, *fn_solve, *fn_lane;
    int lds_shared, lds_per_inst;  // doubles
    int state;                     // doubles of state per instance
    int cache_lds;                 // LDS doubles per instance of the solution caches (Shape::CACHEI)
    int lds_lane_plain, lds_lane_caching;   // doubles, lane kernel (0: shape not supported by it)
    int (*launch)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t);
    int (*launch_jac)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t);
    int (*launch_solve)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t);
    int (*launch_lane)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t);
};

template <class S> static int launch_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    hipLaunchKernelGGL(acme_run_kernel<S>, dim3(grid), dim3(WAVES_PER_BLOCK * 64), lds_bytes, st, A);
    return (int)hipGetLastError();
}

template <class S> static int launch_jac_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    if constexpr (S::NN > 0) {
        hipLaunchKernelGGL(acme_jac_kernel<S>, dim3(grid), dim3(WAVES_PER_BLOCK * 64), lds_bytes, st, A);
        return (int)hipGetLastError();
    } else {
        return (int)hipErrorInvalidValue;      // linear models have no nonlinear solver
    }
}
template <class S> static int launch_solve_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    if constexpr (S::NN > 0 && !S::SOLVE_SPLIT) {
        return launch_shape<S>(A, grid, lds_bytes, st);      // (A.p_in != nullptr tells the run kernel)
    } else if constexpr (S::NN > 0) {
        hipLaunchKernelGGL(acme_solve_kernel<S>, dim3(grid), dim3(WAVES_PER_BLOCK * 64), lds_bytes, st, A);
        return (int)hipGetLastError();
    } else {
        return (int)hipErrorInvalidValue;
    }
}
template <class S> static const void *solve_fn() {
    if constexpr (S::NN > 0 && S::SOLVE_SPLIT) return (const void *)acme_solve_kernel<S>;
    else return nullptr;
}
template <class S> static const void *jac_fn() {
    if constexpr (S::NN > 0) return (const void *)acme_jac_kernel<S>;
    else return nullptr;
}

template <class S> static int launch_lane_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    if constexpr (LaneShape<S>::supported) {
        hipLaunchKernelGGL(acme_lane_kernel<S>, dim3(grid), dim3(LANE_BLOCK), lds_bytes, st, A);
        return (int)hipGetLastError();
    } else {
        return (int)hipErrorInvalidValue;
    }
}
template <class S> static const void *lane_fn() {
    if constexpr (LaneShape<S>::supported) return (const void *)acme_lane_kernel<S>;
    else return nullptr;
}
template <class S> static int lane_lds(bool caching) {
    if constexpr (LaneShape<S>::supported) return LaneShape<S>::lds_doubles(caching);
    else return 0;
}

static const std::vector<KernelEntry> &kernel_table() {
    static const std::vector<KernelEntry> t = {
#define ACME_X(nn, nq, np, nx, nu, ny, rare, nsub)                                                              \
    KernelEntry{Dims{nn, nq, np, nx, nu, ny, rare, (nn) > 0 ? (nsub) : 0}, (const void *)acme_run_kernel<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>, \
                jac_fn<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>(),                                             \
                solve_fn<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>(),                                           \
                lane_fn<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>(),                                            \
                Shape<nn, nq, np, nx, nu, ny, rare, nsub>::lds_doubles(false),                                   \
                Shape<nn, nq, np, nx, nu, ny, rare, nsub>::lds_doubles(true), Shape<nn, nq, np, nx, nu, ny, rare, nsub>::STATE,  \
                Shape<nn, nq, np, nx, nu, ny, rare, nsub>::CACHEI,                                                \
                lane_lds<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>(false),                                      \
                lane_lds<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>(true),                                       \
                &launch_shape<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>,                                        \
                &launch_jac_shape<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>,                                    \
                &launch_solve_shape<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>,                                  \
                &launch_lane_shape<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>},
        ACME_SHAPES(ACME_X)
#undef ACME_X
    };
    return t;
}

static const KernelEntry *find_kernel(const Dims &d) {
    for (const auto &k : kernel_table())
        if (k.d.nn == d.nn && k.d.nq == d.nq && k.d.np == d.np && k.d.nx == d.nx && k.d.nu == d.nu && k.d.ny == d.ny && k.d.rare == d.rare && k.d.nsub == d.nsub)
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
static inline int stream_create_nonblocking(stream_t *st) { return (int)hipStreamCreateWithFlags(st, hipStreamNonBlocking); }
static inline int stream_destroy(stream_t st) { return st ? (int)hipStreamDestroy(st) : 0; }
static inline int device_sync() { return (int)hipDeviceSynchronize(); }
static inline int stream_sync(stream_t st) { return (int)hipStreamSynchronize(st); }
static inline int event_create(event_t *e) { return (int)hipEventCreate(e); }
static inline int event_destroy(event_t e) { return (int)hipEventDestroy(e); }
static inline int event_record(event_t e, stream_t st) { return (int)hipEventRecord(e, st); }
static inline int event_sync(event_t e) { return (int)hipEventSynchronize(e); }
static inline int event_elapsed(float *ms, event_t a, event_t b) { return (int)hipEventElapsedTime(ms, a, b); }
static inline std::mutex *run_mutex() { return nullptr; }     // HIP: runs of distinct batches are concurrent
}  // namespace be

#include "acme_api.inc"
