// acme_hip_low.hip -- second translation unit of libacme_hip.so: the LOW-LDS variants of the 16-lane
// kernels (wave_main<S, MODE, true>: model images read from HBM / L2 instead of LDS, Shape::lds_doubles_low)
// for the shapes that can need them (Shape::HAS_LOW).  A translation unit of its own so that it compiles in
// parallel with acme_hip.hip.
#include "acme_kernels.h"

namespace acme {

template <class S> static KernelFns low_fns() {
    if constexpr (S::HAS_LOW) return make_fns<S, true>();
    else return KernelFns{};
}

KernelFns acme_low_fns(int index) {
    int i = 0;
#define ACME_X(nn, nq, np, nx, nu, ny, rare, nsub) if (index == i++) return low_fns<Shape<nn, nq, np, nx, nu, ny, rare, nsub>>();
    ACME_SHAPES(ACME_X)
#undef ACME_X
    return KernelFns{};
}

}  // namespace acme
