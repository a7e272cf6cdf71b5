// acme_kernels.h -- the __global__ kernels over acme_kernel.h / acme_lane_kernel.h and their launchers,
// shared by the library's translation units: acme_hip.hip (the C ABI, no kernels) and acme_hip_part<k>.hip
// (the kernels of every ACME_NPARTS-th shape), which are compiled in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "acme_wave_hip.h"
#include "acme_kernel.h"
#include "acme_lane_kernel.h"
#include "acme_shapes.h"

namespace acme {

// one instantiation per shape of acme_shapes.h (LOW: model images read from HBM, Shape::lds_doubles_low)
template <class S, bool LOW>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, S::OCC) void acme_run_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    wave_main<S, MODE_RUN, LOW>(A, acme_lds);
}

// run! of a streamed host-buffer run (wave_main MODE_RUN_STREAM: u arrives in HBM while the kernel runs)
template <class S, bool LOW>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, S::OCC) void acme_run_stream_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    wave_main<S, MODE_RUN_STREAM, LOW>(A, acme_lds);
}

// the small companion kernel: get_extrapolation_jacobian for every instance (wave_main MODE_JAC)
template <class S, bool LOW>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, S::OCC) void acme_jac_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    wave_main<S, MODE_JAC, LOW>(A, acme_lds);
}

// ... and solve(solver, p), once per instance (wave_main MODE_SOLVE): the solver-plugin contract
template <class S, bool LOW>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, S::OCC) void acme_solve_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    wave_main<S, MODE_SOLVE, LOW>(A, acme_lds);
}

// run! for small models, one lane per instance (acme_lane_kernel.h): one wave per SIMD is all these
// batches offer, so the kernel is built for the shortest dependent chain per sample, not for occupancy
template <class S>
__global__ __launch_bounds__(LANE_BLOCK, 1) void acme_lane_kernel(KArgs A) {
    extern __shared__ double acme_lds[];
    lane_main<S>(A, acme_lds);
}

// A launch's status: hipGetLastError() right behind it.  The record is NOT cleared beforehand: the tolerated
// failures of the host-buffer path (a page-locking that was refused, an unregister of memory the caller had already freed)
// clear it where they are tolerated (be::tolerated), so whatever else an earlier call on this thread left there is a real
// fault and is reported by the launch that finds it.
#define ACME_LAUNCH(...) ({ hipLaunchKernelGGL(__VA_ARGS__); (int)hipGetLastError(); })
// ... of an entry point chosen at run time (the mid-size kernel's instantiations): params = {&args}
#define ACME_LAUNCH_FN(fn, grid, block, lds, st, params) ({ (void)hipLaunchKernel((fn), (grid), (block), (params), (lds), (st)); (int)hipGetLastError(); })

// the three 16-lane kernels of one shape in one placement (LDS / LOW): entry points for
// hipFuncSetAttribute and launchers
struct KernelFns {
    const void *fn = nullptr, *fn_jac = nullptr, *fn_solve = nullptr, *fn_stream = nullptr;
    int (*launch)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t) = nullptr;
    int (*launch_stream)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t) = nullptr;
    int (*launch_jac)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t) = nullptr;
    int (*launch_solve)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t) = nullptr;
};

template <class S, bool LOW> static int launch_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    return ACME_LAUNCH((acme_run_kernel<S, LOW>), dim3(grid), dim3(WAVES_PER_BLOCK * 64), lds_bytes, st, A);
}
template <class S, bool LOW> static int launch_stream_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    return ACME_LAUNCH((acme_run_stream_kernel<S, LOW>), dim3(grid), dim3(WAVES_PER_BLOCK * 64), lds_bytes, st, A);
}
template <class S, bool LOW> static int launch_jac_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    return ACME_LAUNCH((acme_jac_kernel<S, LOW>), dim3(grid), dim3(WAVES_PER_BLOCK * 64), lds_bytes, st, A);
}
template <class S, bool LOW> static int launch_solve_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    return ACME_LAUNCH((acme_solve_kernel<S, LOW>), dim3(grid), dim3(WAVES_PER_BLOCK * 64), lds_bytes, st, A);
}
template <class S, bool LOW> static KernelFns make_fns() {
    KernelFns f;
    f.fn = (const void *)acme_run_kernel<S, LOW>;
    f.launch = &launch_shape<S, LOW>;
    if constexpr (S::NU > 0) {       // (a model without inputs has nothing to stream)
        f.fn_stream = (const void *)acme_run_stream_kernel<S, LOW>;
        f.launch_stream = &launch_stream_shape<S, LOW>;
    }
    if constexpr (S::NN > 0) {       // (linear models have no nonlinear solver)
        f.fn_jac = (const void *)acme_jac_kernel<S, LOW>;
        f.fn_solve = (const void *)acme_solve_kernel<S, LOW>;
        f.launch_jac = &launch_jac_shape<S, LOW>;
        f.launch_solve = &launch_solve_shape<S, LOW>;
    }
    return f;
}

// everything the host needs of one shape's kernels: the 16-lane kernels with the images in LDS and (where the
// shape can need them, Shape::HAS_LOW) their LOW-LDS variants, and the lane-per-instance run kernel
struct ShapeFns {
    KernelFns lds, low;
    const void *fn_lane = nullptr;
    int (*launch_lane)(const KArgs &, unsigned grid, size_t lds_bytes, hipStream_t) = nullptr;
};
template <class S> static int launch_lane_shape(const KArgs &A, unsigned grid, size_t lds_bytes, hipStream_t st) {
    if constexpr (LaneShape<S>::supported) {
        return ACME_LAUNCH(acme_lane_kernel<S>, dim3(grid), dim3(LANE_BLOCK), lds_bytes, st, A);
    } else {
        return (int)hipErrorInvalidValue;
    }
}
// (a shape's kernels can be split over two translation units -- shape_part / shape_part_low below --: each fills in its half)
template <class S, bool MAIN, bool LOWHALF> static void fill_shape_fns(ShapeFns &f) {
    if constexpr (MAIN) {
        f.lds = make_fns<S, false>();
        if constexpr (LaneShape<S>::supported) f.fn_lane = (const void *)acme_lane_kernel<S>;
        f.launch_lane = &launch_lane_shape<S>;
    }
    if constexpr (LOWHALF && S::HAS_LOW) f.low = make_fns<S, true>();
}

// The kernels are instantiated in ACME_NPARTS translation units (acme_hip_part<k>.hip: the shapes with
// shape_part(number in ACME_SHAPES) == k), compiled in parallel: the run kernel of one shape alone is
// 10 ... 30 thousand instructions.  Each part answers for its own shapes.
constexpr int ACME_NPARTS = 8;
// Which part a shape lives in.  Not only for build time: the parts are compiled with different instruction
// schedulers (__graft_entry__.py: HIP_UNIT_FLAGS).  Part 0 -- the smallest models, whose lane-per-instance kernels
// prefer the compiler's default scheduler (the diode clipper sweep loses 1.2 % with max-ilp); parts 1-5 with -amdgpu-sched-strategy=max-ilp (birdie +3.9 %, config 4 +1.4 %,
// headline +0.2 %; max-memory-clause: -0.6 ... -2 %).  (Parts 4 and 5: the two decomposed shapes, on their own for the
// build's wall time -- 195 s for the slowest unit with four parts.)
constexpr int shape_part(int index) {
    constexpr int table[] = {0 /* diode clipper */, 1 /* superover, fixed pots */, 2 /* superover, pots as inputs */,
                             0 /* birdie, fixed vol */, 1 /* birdie, vol as input */, 2 /* linear */, 3 /* generic small */,
                             3 /* generic medium */, 3 /* generic large */, 5 /* decomposed small */, 4 /* decomposed medium */,
                             0 /* superover, pots as inputs, condensed -- the headline kernel: the default scheduler again (round 5, one solver copy: 275.1 ms against 278.6 with max-ilp; round 4's two-copy kernel gained 0.6 % with max-ilp) */,
                             6 /* decomposed, up to 8 small sub-problems: a unit of its own (five minutes of compile time) */};
    return index < (int)(sizeof(table) / sizeof(table[0])) ? table[index] : index % ACME_NPARTS;
}
// ... and the part that holds a shape's LOW-LDS variants: its own, except for the shape of up to 8 sub-problems -- eight
// kernels of 7.4 minutes of compile time together -- whose LOW variants are the last unit's
constexpr int shape_part_low(int index) { return index == 12 ? 7 : shape_part(index); }
bool acme_shape_fns_part0(int index, ShapeFns *out);
bool acme_shape_fns_part1(int index, ShapeFns *out);
bool acme_shape_fns_part2(int index, ShapeFns *out);
bool acme_shape_fns_part3(int index, ShapeFns *out);
bool acme_shape_fns_part4(int index, ShapeFns *out);
bool acme_shape_fns_part5(int index, ShapeFns *out);
bool acme_shape_fns_part6(int index, ShapeFns *out);
bool acme_shape_fns_part7(int index, ShapeFns *out);

}  // namespace acme
