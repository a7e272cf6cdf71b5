// acme_coop_kernel.h -- the mid-size kernel's entry points (device code: acme_coop.h).  One translation unit per column
// count of the register instantiations (acme_hip_coop<NC>.hip, compiled in parallel: each is two kernels of fifteen to
// twenty thousand instructions), the any-size instantiation in acme_hip.hip; each unit answers for its own kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "acme_coop.h"

// GArgs::coop_wpb waves per block, GArgs::coop_gpw instances per wave, their working arrays in LDS.
//   NC:  17 ... 32 unknowns -- the Jacobian's rows in registers (NC columns), eliminated in the instance's learnt row order
//        with threshold pivoting; -1 ... -4: the same elimination on one matrix in LDS, that many rows per lane (up to 64
//        unknowns); 0: the reference's LU literally, factors in LDS, any size (ACME_COOP_LITERAL=1)
// The any-size instantiation is held to 256 registers -- no accumulation registers (__graft_entry__.py checks the code
// object) --, two waves per SIMD; the register instantiations take what one wave per SIMD may have.  DESIGN.md 3 "The zeros
// of the unbounded any-size build" has the story of that bound.
template <bool IMGL, int NC> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NC <= 0 ? 2 : 1)))
void acme_coop_kernel(acme::GArgs A) {
    extern __shared__ double acme_lds[];
    const int wave = (int)threadIdx.x >> 6;
    acme::coop_main<IMGL, NC>(A, acme_lds, wave, (int)blockIdx.x * A.coop_wpb + wave, (int)threadIdx.x & 63);
}

namespace acme {
// the entry point of (image staged in LDS?) in the unit of NC columns
const void *acme_coop_fn_nc20(int imgl);
const void *acme_coop_fn_nc24(int imgl);
const void *acme_coop_fn_nc28(int imgl);
const void *acme_coop_fn_nc32(int imgl);
const void *acme_coop_fn_lds1(int imgl);          // (the threshold path on a matrix in LDS, 1 ... 4 rows per lane: NC = -1 ... -4)
const void *acme_coop_fn_lds2(int imgl);
const void *acme_coop_fn_lds3(int imgl);
const void *acme_coop_fn_lds4(int imgl);
const void *acme_coop_fn_wave64(int imgl);        // (... and with one instance per wave, one row per lane: NC = COOP_WAVE64)
template <int NC> static inline const void *coop_fns_of(int imgl) {
    return imgl ? (const void *)acme_coop_kernel<true, NC> : (const void *)acme_coop_kernel<false, NC>;
}
}  // namespace acme
