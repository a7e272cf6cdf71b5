// acme_shapes.h -- the (NN,NQ,NP,NX,NU,NY,RARE,NSUB,NL) shapes the kernel is instantiated for.
//
// Loop bounds, DPP lane selects and register-array subscripts are compile-time constants
// (the reference gets the same effect from Julia specialising StaticArrays sizes per
// circuit).  A model whose dimensions are not listed is zero-padded on the host to the
// cheapest listed shape that contains it (see acme_pack.h); the first five are the exact
// shapes of the BASELINE circuits (dimensions verified by tests/test_frontend.py against
// the np(model,k) pins of test/runtests.jl:699,724,734,744,777).
#pragma once
// clang-format off
#if defined(ACME_DEV_SHAPES) && ACME_DEV_SHAPES + 0 == 2   /* developer builds (tools/isa.sh): one shape, compiles in seconds */
#define ACME_SHAPES(X)                                                                     \
    X( 2,  4,  1,  1, 1, 1, 0, 1, 0)
#elif defined(ACME_DEV_SHAPES) && ACME_DEV_SHAPES + 0 == 4
#define ACME_SHAPES(X)                                                                     \
    X( 7, 14,  5, 11, 1, 1, 0, 1, 0)
#elif defined(ACME_DEV_SHAPES) && ACME_DEV_SHAPES + 0 == 5
#define ACME_SHAPES(X)                                                                     \
    X( 4,  9,  3,  3, 2, 1, 0, 1, 0)
#elif defined(ACME_DEV_SHAPES) && ACME_DEV_SHAPES + 0 == 16
#define ACME_SHAPES(X)                                                                     \
    X(16, 32, 16, 32, 8, 8, 1, 1, 0)
#elif defined(ACME_DEV_SHAPES) && ACME_DEV_SHAPES + 0 == 13
#define ACME_SHAPES(X)                                                                     \
    X(13, 29, 11, 11, 4, 1, 0, 1, 0)
#elif defined(ACME_DEV_SHAPES)
#define ACME_SHAPES(X)                                                                     \
    X(13, 29, 11, 11, 4, 1, 0, 1, 6)
#else
#define ACME_SHAPES(X)                                                                     \
    X( 2,  4,  1,  1, 1, 1, 0, 1, 0)  /* examples/diodeclipper.jl                          */ \
    X( 7, 14,  5, 11, 1, 1, 0, 1, 0)  /* examples/superover.jl, fixed potentiometers       */ \
    X(13, 29, 11, 11, 4, 1, 0, 1, 0)  /* examples/superover.jl, drive/tone/level as inputs */ \
    X( 2,  4,  2,  3, 1, 1, 0, 1, 0)  /* examples/birdie.jl, fixed vol                     */ \
    X( 4,  9,  3,  3, 2, 1, 0, 1, 0)  /* examples/birdie.jl, vol as input                  */ \
    X( 0,  0,  0, 32, 2, 2, 0, 1, 0)  /* linear models (no nonlinear sub-problem)          */ \
    X( 4, 12,  4,  4, 2, 4, 1, 1, 0)  /* generic small  (all element kinds)                */ \
    X( 8, 24,  8, 16, 4, 4, 1, 1, 0)  /* generic medium (all element kinds)                */ \
    X(16, 32, 16, 32, 8, 8, 1, 1, 0)  /* generic large  (all element kinds)                */ \
    X( 4, 12,  4, 16, 4, 4, 1, 4, 0)  /* decomposed nonlinearity: up to 4 small sub-problems */ \
    X( 8, 24,  8, 16, 4, 4, 1, 4, 0)  /* decomposed nonlinearity: up to 4 medium sub-problems */ \
    X(13, 29, 11, 11, 4, 1, 0, 1, 6)  /* superover, pots as inputs: the 6 potentiometer rows condensed (the HEADLINE kernel) */ \
    X( 4, 12,  4, 16, 4, 4, 1, 8, 0)  /* decomposed nonlinearity: up to 8 small sub-problems */
// clang-format on
#endif
