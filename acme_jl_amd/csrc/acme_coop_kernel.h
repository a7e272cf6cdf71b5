// acme_coop_kernel.h -- the mid-size kernel's entry points (device code: acme_coop.h).  One translation unit per column
// count of the register instantiations (acme_hip_coop<NC>.hip, compiled in parallel: each is four kernels of ten to
// twenty thousand instructions), the any-size instantiation in acme_hip.hip; each unit answers for its own kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "acme_coop.h"

// GArgs::coop_wpb waves per block, GArgs::coop_gpw instances per wave, their working arrays in LDS.
//   NC:  the factor matrix's columns in registers -- 17 ... 32 unknowns -- or 0: factors in LDS, any size
//   THR: the elimination in a learnt row order with threshold pivoting (the default of the register instantiations), or
//        the reference's pivoting literally (ACME_COOP_LITERAL=1; the any-size instantiation always)
// The any-size instantiation is held to 256 registers (two waves per SIMD); the register instantiations take what one wave
// per SIMD may have.  DESIGN.md 3 "The zeros of the unbounded any-size build" has the story of that bound.
template <bool IMGL, int NC, bool THR> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NC == 0 ? 2 : 1)))
void acme_coop_kernel(acme::GArgs A) {
    extern __shared__ double acme_lds[];
    const int wave = (int)threadIdx.x >> 6;
    acme::coop_main<IMGL, NC, THR>(A, acme_lds, wave, (int)blockIdx.x * A.coop_wpb + wave, (int)threadIdx.x & 63);
}

namespace acme {
// the entry point of (image staged in LDS?, threshold path?) in the unit of NC columns
const void *acme_coop_fn_nc20(int imgl, int thr);
const void *acme_coop_fn_nc24(int imgl, int thr);
const void *acme_coop_fn_nc28(int imgl, int thr);
const void *acme_coop_fn_nc32(int imgl, int thr);
template <int NC> static inline const void *coop_fns_of(int imgl, int thr) {
    if (imgl) return thr ? (const void *)acme_coop_kernel<true, NC, true> : (const void *)acme_coop_kernel<true, NC, false>;
    return thr ? (const void *)acme_coop_kernel<false, NC, true> : (const void *)acme_coop_kernel<false, NC, false>;
}
}  // namespace acme
