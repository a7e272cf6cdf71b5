// acme_lane_kernel.h -- batched run!(::DiscreteModel, u) for SMALL models: one LANE per circuit instance.
//
// The row-per-lane kernel of acme_kernel.h gives every instance 16 lanes.  For the small BASELINE
// circuits (diode clipper nn = 2, birdie nn = 2 / 4) at their stated batch sizes (4 096 / 2 048
// instances per GPU) that fills at most one wave per SIMD, and a launch then lasts as long as ONE
// wave needs for its T samples: what counts is the length of a wave's dependent instruction chain
// per sample, not how many lanes it keeps busy.  Here a lane owns a whole instance: the nn x nn
// Jacobian, its LU factors and the solver state live in that lane's registers, there is no
// cross-lane traffic at all (only wave ballots for the data-dependent loop counts), the model's
// constants -- the same for every lane -- are loaded ONCE per launch into vector registers (LCR below;
// fetched per use, even by scalar loads, they cost a lone wave more than the arithmetic), and LDS only
// holds the stored p's of the solution caches where they do not fit the registers (np > 2).  A sample costs a wave ~1/4 of the instructions of the 16-lane kernel and none of its
// LDS round trips, and the wave advances 64 instances instead of 4.
//
// Being free of the row-per-lane layout, the linear algebra here is the reference's own:
// setlhs! (partial pivoting, first strict maximum, full-row interchange, reciprocal on the
// diagonal, exact-zero pivot = failure; src/solvers.jl:46-93) and solve! (:95-132), the
// extrapolated start as  last_z - last_lu \ (last_Jp (p - last_p))  (:209-215) -- the same operations on
// the same pivots; only the ORDER of the equations may differ from the reference's (rows are evaluated in
// the packed position order of acme_pack.h, which can change which of two equal-sized candidates setlhs!
// meets first).  State, solution caches and reports use the formats of acme_kernel.h, so that
// acme_batch_solve / get_extrapolation_jacobian (16-lane kernels) keep working on the same batch.
//
// Restrictions (the dispatcher in acme_api.inc checks them): one shared model image, one nonlinear
// sub-problem, unified element rows (diode, Ebers-Moll BJT, potentiometer), nn <= 4, nx + ny <= 16.
#pragma once
#include "acme_kernel.h"

namespace acme {

constexpr int LANE_BLOCK = WAVES_PER_BLOCK * 64;   // instances per block (its waves never talk to each other)

template <class S> struct LaneShape {
    // LDS doubles per block: the stored p's of its instances' solution caches, cp[j][entry][lane]
    // ... unless they fit the registers (np <= 2: 16 or 32 doubles per lane; slots never written hold
    // +inf, which no distance beats, so the scan does not look at the fill count either)
    static constexpr bool CREG = S::NP * CACHE <= 32;
    static constexpr int CACHE_LDS = CREG ? 0 : S::NP * CACHE * LANE_BLOCK;
    ACME_HD static constexpr int lds_doubles(bool caching) { return caching ? CACHE_LDS + 2 : 2; }
    static constexpr LaneLayout LL = make_lane_layout(S::NN, S::NP, S::NX, S::NU, S::NY);
    // the model's constants stay in REGISTERS for the whole launch (every lane holds the same
    // values): a lone wave per SIMD has nothing to hide a memory access behind -- fetched per use,
    // even as scalar loads, they cost more than the arithmetic (measured: 0.92e9 instead of ...)
    static constexpr bool supported = S::NN >= 1 && S::NN <= 4 && !S::RARE && S::NSUB == 1 && S::NX + S::NY <= GROUP &&
                                      S::NP <= 4 && S::NX <= 4 && S::NU <= 2 && S::NY <= 2 && LL.total <= 112;
};

// setlhs! (src/solvers.jl:46-93) on a register-resident n x n matrix f[i][j]; row interchanges
// through selects (the pivot row index differs from lane to lane).  Returns false for an exactly
// zero pivot (the elimination still runs to the end; its result is not used then).
template <int N> ACME_DEV bool lane_lu_factor(double (&f)[N][N], int (&ipiv)[N]) {
    bool ok = true;
    sfor<0, N>([&](auto kc) ACME_LAMBDA {
        constexpr int k = decltype(kc)::value;
        int kp = k;
        double amax = 0.0;
        sfor<k, N>([&](auto ic) ACME_LAMBDA {          // find index max: first strict maximum
            constexpr int i = decltype(ic)::value;
            const double absi = fabs(f[i][k]);
            const bool c = absi > amax;
            kp = c ? i : kp;
            amax = c ? absi : amax;
        });
        ipiv[k] = kp;
        sfor<0, N>([&](auto jc) ACME_LAMBDA {          // interchange rows k <-> kp, all columns
            constexpr int j = decltype(jc)::value;
            const double tk = f[k][j];
            double nk = tk;
            sfor<k + 1, N>([&](auto ic) ACME_LAMBDA {
                constexpr int i = decltype(ic)::value;
                const bool c = kp == i;
                nk = c ? f[i][j] : nk;
                f[i][j] = c ? tk : f[i][j];
            });
            f[k][j] = nk;
        });
        ok = ok && (f[k][k] != 0.0);
        const double fkkinv = wv::recip(f[k][k]);      // inv(): 1 ulp here, correctly rounded in the reference
        f[k][k] = fkkinv;
        sfor<k + 1, N>([&](auto ic) ACME_LAMBDA { f[decltype(ic)::value][k] *= fkkinv; });
        sfor<k + 1, N>([&](auto jc) ACME_LAMBDA {      // update the rest
            constexpr int j = decltype(jc)::value;
            sfor<k + 1, N>([&](auto ic) ACME_LAMBDA {
                constexpr int i = decltype(ic)::value;
                f[i][j] = fma(-f[i][k], f[k][j], f[i][j]);
            });
        });
    });
    return ok;
}

// solve! (src/solvers.jl:95-132): x <- A^-1 x with the factors of lane_lu_factor
template <int N> ACME_DEV void lane_lu_solve(const double (&f)[N][N], const int (&ipiv)[N], double (&x)[N]) {
    sfor<0, N>([&](auto ic) ACME_LAMBDA {              // x[i], x[ipiv[i]] = x[ipiv[i]], x[i]
        constexpr int i = decltype(ic)::value;
        const double xi = x[i];
        double ni = xi;
        sfor<i + 1, N>([&](auto mc) ACME_LAMBDA {
            constexpr int m = decltype(mc)::value;
            const bool c = ipiv[i] == m;
            ni = c ? x[m] : ni;
            x[m] = c ? xi : x[m];
        });
        x[i] = ni;
    });
    sfor<0, N>([&](auto jc) ACME_LAMBDA {              // unit lower triangular
        constexpr int j = decltype(jc)::value;
        sfor<j + 1, N>([&](auto ic) ACME_LAMBDA { x[decltype(ic)::value] = fma(-f[decltype(ic)::value][j], x[j], x[decltype(ic)::value]); });
    });
    sfor_down<N>([&](auto jc) ACME_LAMBDA {            // upper triangular, reciprocals on the diagonal
        constexpr int j = decltype(jc)::value;
        x[j] = f[j][j] * x[j];
        sfor<0, j>([&](auto ic) ACME_LAMBDA { x[decltype(ic)::value] = fma(-f[decltype(ic)::value][j], x[j], x[decltype(ic)::value]); });
    });
}

template <class S> ACME_DEV void lane_main(const KArgs &A, double *lds) {
    constexpr int NN = S::NN, NP = S::NP, NX = S::NX, NU = S::NU, NY = S::NY, NT = 3;
    constexpr int NPr = NP > 0 ? NP : 1, NXr = NX > 0 ? NX : 1, NUr = NU > 0 ? NU : 1, NYr = NY > 0 ? NY : 1;
    static_assert(LaneShape<S>::supported, "shape not supported by the lane-per-instance kernel");

    const int lane = wv::tid();                 // lane within the block
    const int dens = A.lane_density > 0 && A.lane_density < 64 ? A.lane_density : 64;
    const long long inst = ((long long)wv::bid() * WAVES_PER_BLOCK + (lane >> 6)) * dens + (lane & 63);
    const bool valid = (lane & 63) < dens && inst < A.n_inst;
    const long long ii = valid ? inst : 0;
    // The model's constants (LaneLayout block: everything a row needs is contiguous): read-only for the
    // whole launch and addressed uniformly by the wave, so they are FETCHED by scalar loads
    // (wv::uniform_ro tells the compiler so) -- once, here -- and then pinned in vector registers.
    constexpr LaneLayout LL = make_lane_layout(NN, NP, NX, NU, NY);
    double LCR[LL.total];                       // register-resident copy (same values in every lane)
    {
        const auto LC = wv::uniform_ro(A.lanec);
        // (wv::keep pins each value in a VECTOR register: left alone, the compiler notices that the
        // values are wave-uniform, moves them to scalar registers, runs out of those and brings them
        // back with one v_readlane per use)
        sfor<0, LL.total>([&](auto ec) ACME_LAMBDA { LCR[decltype(ec)::value] = wv::keep(LC[decltype(ec)::value]); });
    }
    auto row_ptr = [&](int off) ACME_LAMBDA -> const double * { return LCR + off; };
    const wv::ExpTab etab = wv::load_exp_tab();  // the exponential's 16 constants: scalar registers, once
    const bool caching = A.solver == SOLVER_CACHING_HOMOTOPY;
    const bool has_bjt = A.has_bjt != 0;
    double *cpl = lds + lane;                   // stored p's of this lane's cache: cpl[(j * CACHE + e) * LANE_BLOCK]
    double *cag = A.cache + ii * S::CACHEIH;    // HBM image of the cache (layout of acme_common.h)

    // ---- per-instance state in registers -------------------------------------------------------
    double x[NXr], lp[NPr], lz[NN], z[NN];
    const double *st = A.state + ii * S::STATE;
    sfor<0, NX>([&](auto c) ACME_LAMBDA { x[decltype(c)::value] = valid ? st[decltype(c)::value] : 0.0; });
    sfor<0, NP>([&](auto c) ACME_LAMBDA { lp[decltype(c)::value] = valid ? st[NX + decltype(c)::value] : 0.0; });
    sfor<0, NN>([&](auto c) ACME_LAMBDA { lz[decltype(c)::value] = valid ? st[NX + NP + decltype(c)::value] : 0.0; z[decltype(c)::value] = 0.0; });
    int ccount = 0, chead = 0;
    constexpr bool CREG = LaneShape<S>::CREG;
    double cpr[CREG ? NPr : 1][CREG ? CACHE : 1];     // the stored p's, register-resident (LaneShape::CREG)
    if constexpr (CREG)
        sfor<0, NP>([&](auto jc) ACME_LAMBDA { sfor<0, CACHE>([&](auto ec) ACME_LAMBDA { cpr[decltype(jc)::value][decltype(ec)::value] = (double)INFINITY; }); });
    if (caching) {
        if (valid) {
            const int *meta = reinterpret_cast<const int *>(cag + NP * CACHE);
            ccount = meta[0];
            chead = meta[1];
            sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                if constexpr (CREG) {
                    sfor<0, CACHE>([&](auto ec) ACME_LAMBDA {
                        constexpr int e = decltype(ec)::value;
                        const double v = cag[j * CACHE + e];
                        cpr[j][e] = e < ccount ? v : (double)INFINITY;
                    });
                } else {
                    for (int e = 0; e < CACHE; ++e) cpl[(j * CACHE + e) * LANE_BLOCK] = cag[j * CACHE + e];
                }
            });
        }
    }
    long long r_nwarn = 0, r_first_nonconv = -1, r_first_nonfinite = -1, r_iters_total = 0, r_iters_max = 0;
    if (valid) {
        const long long *rp = A.report + inst * RW_WORDS;
        r_nwarn = rp[RW_NWARN]; r_first_nonconv = rp[RW_FIRST_NONCONV]; r_first_nonfinite = rp[RW_FIRST_NONFINITE];
        r_iters_total = rp[RW_ITERS_TOTAL]; r_iters_max = rp[RW_ITERS_MAX];
    }
    bool dead = !valid || r_first_nonfinite >= 0;

    // extrapolation origin: LU factors and Jp at (lp, lz)  (last_linsolver, last_Jp)
    double olu[NN][NN], ojp[NN][NPr];
    int oipiv[NN];
    // work
    double pf[NN][NT], tv[NN][NT], res[NN], jm[NN][NN];
    int ipiv[NN];
    // The lanes of `doit` take a new origin (factors f, pivots pv, Jp, p, z).  (The origin stays in registers:
    // kept in LDS -- it is read once and written once per sample, but carried through every loop of the solver at
    // a register copy per trip -- the LDS latency, which a lone wave has nothing to hide behind, cost 6 %.)
    auto origin_store = [&](bool doit, const double (&f)[NN][NN], const int (&pv)[NN], const double (&jp)[NN][NPr],
                            const double (&pp)[NPr], const double (&zz)[NN]) ACME_LAMBDA {
        sfor<0, NN>([&](auto ic) ACME_LAMBDA {
            constexpr int i = decltype(ic)::value;
            oipiv[i] = doit ? pv[i] : oipiv[i];
            sfor<0, NN>([&](auto jc) ACME_LAMBDA { olu[i][decltype(jc)::value] = sel(doit, f[i][decltype(jc)::value], olu[i][decltype(jc)::value]); });
            sfor<0, NP>([&](auto jc) ACME_LAMBDA { ojp[i][decltype(jc)::value] = sel(doit, jp[i][decltype(jc)::value], ojp[i][decltype(jc)::value]); });
            lz[i] = sel(doit, zz[i], lz[i]);
        });
        sfor<0, NP>([&](auto jc) ACME_LAMBDA { lp[decltype(jc)::value] = sel(doit, pp[decltype(jc)::value], lp[decltype(jc)::value]); });
    };

    // pfull <- q0 + pexp p, only the (<= NT) entries every residual row needs (set_p, src/ACME.jl:237-243)
    auto set_p = [&](const double (&p)[NPr]) ACME_LAMBDA {
        sfor<0, NN>([&](auto rc_) ACME_LAMBDA {
            constexpr int r = decltype(rc_)::value;
            const auto R = row_ptr(r * LL.row);
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                double acc = R[LL.q0 + t];
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    acc = fma(R[LL.pexp + t * NP + j], p[j], acc);
                });
                pf[r][t] = acc;
            });
        });
    };
    // evaluate!(nleq, z): res, Jq non-zeros tv, J = Jq fq -> jm.  Returns finite(res, J).
    auto evaluate = [&](const double (&zz)[NN]) ACME_LAMBDA -> bool {
        double chk = 0.0;
        // q entries of every row first ...
        double e[NN][NT];
        sfor<0, NN>([&](auto rc_) ACME_LAMBDA {
            constexpr int r = decltype(rc_)::value;
            const auto R = row_ptr(r * LL.row);
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                double acc = pf[r][t];
                sfor<0, NN>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    acc = fma(R[LL.fq + t * NN + j], zz[j], acc);
                });
                e[r][t] = acc;
            });
        });
        // ... then the exponentials of the unified element rows (acme_common.h), two at a time in lockstep
        // (exp_junction2: the same arithmetic per argument as exp_junction, the two dependency chains interleaved
        // -- a lone wave waits out every dependent fp64 instruction): with a BJT in the MODEL both junctions of a
        // row (rows without a junction have sA = sB = 0: exp(0) - 1 = 0), without one the first junctions of two
        // ROWS.  ONE scalar branch for all rows -- a branch per row, or per row KIND (the kinds are wave-uniform
        // too), costs ~30 cycles each with nothing to hide them.
        double exA[NN], exB[NN];
        auto sa = [&](auto rc_) ACME_LAMBDA { return (row_ptr(decltype(rc_)::value * LL.row) + (LL.ur - UR_SA))[UR_SA]; };
        auto sb = [&](auto rc_) ACME_LAMBDA { return (row_ptr(decltype(rc_)::value * LL.row) + (LL.ur - UR_SA))[UR_SB]; };
        if (has_bjt) {
            sfor<0, NN>([&](auto rc_) ACME_LAMBDA {
                constexpr int r = decltype(rc_)::value;
                exp_junction2(e[r][0] * sa(rc_), e[r][1] * sb(rc_), exA[r], exB[r], etab);
            });
        } else {
            sfor<0, NN / 2>([&](auto pc) ACME_LAMBDA {
                constexpr int r = 2 * decltype(pc)::value;
                exp_junction2(e[r][0] * sa(std::integral_constant<int, r>{}), e[r + 1][0] * sa(std::integral_constant<int, r + 1>{}),
                              exA[r], exA[r + 1], etab);
            });
            if constexpr (NN % 2 == 1) exA[NN - 1] = exp_junction<false>(e[NN - 1][0] * sa(std::integral_constant<int, NN - 1>{}), etab);
            sfor<0, NN>([&](auto rc_) ACME_LAMBDA { exB[decltype(rc_)::value] = 1.0; });
        }
        sfor<0, NN>([&](auto rc_) ACME_LAMBDA {
            constexpr int r = decltype(rc_)::value;
            const auto R = row_ptr(r * LL.row);
            const auto U = R + (LL.ur - UR_SA);           // U[UR_x] = unified row constant x
            const double cA = U[UR_CA], cB = U[UR_CB], dA = U[UR_DA], dB = U[UR_DB], h = U[UR_H];
            const double g0 = U[UR_G0], g1 = U[UR_G1], g2 = U[UR_G2], w0 = U[UR_W0], w1 = U[UR_W1];
            const double hw = h * fma(w1, e[r][2], w0);
            double rr = cA * (exA[r] - 1.0);
            rr = fma(cB, exB[r] - 1.0, rr);
            rr = fma(g0, e[r][0], rr);
            rr = fma(g1, e[r][1], rr);
            rr = fma(g2, e[r][2], rr);
            res[r] = fma(hw, e[r][1], rr);
            tv[r][0] = fma(dA, exA[r], g0);
            tv[r][1] = fma(dB, exB[r], g1 + hw);
            tv[r][2] = fma(h, e[r][1], g2);
            sfor<0, NN>([&](auto jc) ACME_LAMBDA {      // J row = Jq row * fq
                constexpr int j = decltype(jc)::value;
                double acc = tv[r][0] * R[LL.fq + 0 * NN + j];
                acc = fma(tv[r][1], R[LL.fq + 1 * NN + j], acc);
                acc = fma(tv[r][2], R[LL.fq + 2 * NN + j], acc);
                jm[r][j] = acc;
                chk = fma(acc, 0.0, chk);
            });
            chk = fma(res[r], 0.0, chk);
        });
        return chk == 0.0;
    };
    // calc_Jp (src/ACME.jl:246-251): Jp = Jq pexp at the latest evaluate!
    auto calc_jp = [&](double (&jp)[NN][NPr]) ACME_LAMBDA {
        sfor<0, NN>([&](auto rc_) ACME_LAMBDA {
            constexpr int r = decltype(rc_)::value;
            const auto R = row_ptr(r * LL.row);
            sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                double acc = tv[r][0] * R[LL.pexp + 0 * NP + j];
                acc = fma(tv[r][1], R[LL.pexp + 1 * NP + j], acc);
                acc = fma(tv[r][2], R[LL.pexp + 2 * NP + j], acc);
                jp[r][j] = acc;
            });
        });
    };
    // set_extrapolation_origin(solver, p, z) (src/solvers.jl:183-196) for the lanes of `doit`
    auto set_origin = [&](bool doit) ACME_LAMBDA {
        set_p(lp);
        (void)evaluate(lz);
        double jp[NN][NPr];
        calc_jp(jp);
        (void)lane_lu_factor<NN>(jm, ipiv);
        origin_store(doit, jm, ipiv, jp, lp, lz);       // (lp, lz: the caller's, already those of the new origin)
    };
    // the factors and Jp at the origin are recomputed from the persistent (lp, lz) at launch start
    sfor<0, NN>([&](auto ic) ACME_LAMBDA {
        constexpr int i = decltype(ic)::value;
        oipiv[i] = i;
        sfor<0, NN>([&](auto jc) ACME_LAMBDA { olu[i][decltype(jc)::value] = 0.0; });
        sfor<0, NP>([&](auto jc) ACME_LAMBDA { ojp[i][decltype(jc)::value] = 0.0; });
    });
    set_origin(true);

    // solve(::SimpleSolver, p) (src/solvers.jl:207-236) for the lanes with `need`
    auto base_solve = [&](const double (&target)[NPr], bool need, int &its) ACME_LAMBDA -> bool {
        set_p(target);
        double t[NN];
        sfor<0, NN>([&](auto ic) ACME_LAMBDA {          // tmp_nn = last_Jp (p - last_p)
            constexpr int i = decltype(ic)::value;
            double acc = 0.0;
            sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                acc = fma(ojp[i][j], target[j] - lp[j], acc);
            });
            t[i] = acc;
        });
        lane_lu_solve<NN>(olu, oipiv, t);
        sfor<0, NN>([&](auto ic) ACME_LAMBDA { z[decltype(ic)::value] = sel(need, lz[decltype(ic)::value] - t[decltype(ic)::value], z[decltype(ic)::value]); });
        bool act = need, conv = false, accepted = false;
        its = 0;
        // (do-while: every update below is predicated on `act`, so a pass for a wave none of whose lanes needs
        // one -- it does not happen on the direct attempt -- changes nothing; but what the loop hands on, the
        // factors and Jq of the last pass, then needs no merge with the values from before a loop that might
        // not run: a register copy each per iteration, 10 ... 40 of them)
        do {
            its += act ? 1 : 0;
            const bool finite = evaluate(z);
            // resmaxabs < tol (src/solvers.jl:203,218): false for a NaN / inf residual, and independent of
            // whether J is finite (solve() returns early then, hasconverged still looks at the residual alone)
            int small_i = 1;
            sfor<0, NN>([&](auto ic) ACME_LAMBDA { small_i &= (int)(fabs(res[decltype(ic)::value]) < A.tol); });
            const bool small = small_i != 0;
            double dz[NN];
            sfor<0, NN>([&](auto ic) ACME_LAMBDA { dz[decltype(ic)::value] = res[decltype(ic)::value]; });
            const bool ok = lane_lu_factor<NN>(jm, ipiv);       // LU before the convergence test (:223-226)
            const bool want = act && finite && ok && small;
            const bool stop_bad = act && (!finite || !ok);
            conv = stop_bad ? small : conv;                     // hasconverged looks at resmaxabs alone
            accepted = accepted || want;
            const bool step = act && !stop_bad && !want;
            lane_lu_solve<NN>(jm, ipiv, dz);
            sfor<0, NN>([&](auto ic) ACME_LAMBDA { z[decltype(ic)::value] = sel(step, z[decltype(ic)::value] - dz[decltype(ic)::value], z[decltype(ic)::value]); });
            act = step && (its < A.maxiter);
        } while (wv::ballot(act));
        // The accepted iterate is the new extrapolation origin (src/solvers.jl:227-233).  Taken HERE, once:
        // a lane that has stopped keeps its z, so every later pass of the loop (run for the lanes still
        // iterating) recomputes the same evaluate! and the same factors for it -- at the exit jm / ipiv / tv
        // are those of every lane's last iterate.
        if (wv::ballot(accepted)) {
            double jp[NN][NPr];
            calc_jp(jp);
            origin_store(accepted, jm, ipiv, jp, target, z);
        }
        return conv || accepted;
    };

    // solve(::CachingSolver, p) (src/solvers.jl:347-396) with the bounded store of acme_common.h
    auto cached_solve = [&](const double (&target)[NPr], bool need, int &its) ACME_LAMBDA -> bool {
        if (caching) {
            double best = 0.0;
            sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                const double d = target[j] - lp[j];
                best = fma(d, d, best);
            });
            // nearest stored p: all the distances, their minimum as a tree (4 levels instead of a 16-deep chain of
            // compare + three selects), and the ENTRY only when there is a hit -- the first one at the minimum,
            // as the reference's scan in storing order finds it.  A NaN distance never wins (minNum).
            double dist[CACHE];
            sfor<0, CACHE>([&](auto ec) ACME_LAMBDA {
                constexpr int e = decltype(ec)::value;
                double d = 0.0;
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    double cv;
                    if constexpr (CREG) cv = cpr[j][e];       // (slots never written hold +inf)
                    else cv = cpl[(j * CACHE + e) * LANE_BLOCK];
                    const double tt = cv - target[j];
                    d = fma(tt, tt, d);
                });
                dist[e] = (CREG || e < ccount) ? d : (double)INFINITY;
            });
            double mt[CACHE];
            sfor<0, CACHE>([&](auto ec) ACME_LAMBDA { mt[decltype(ec)::value] = dist[decltype(ec)::value]; });
            sfor<0, 4>([&](auto lc) ACME_LAMBDA {
                constexpr int w = CACHE >> (decltype(lc)::value + 1);
                sfor<0, w>([&](auto ec) ACME_LAMBDA { mt[decltype(ec)::value] = __builtin_fmin(mt[decltype(ec)::value], mt[decltype(ec)::value + w]); });
            });
            static_assert(CACHE == 16, "four levels");
            const bool hit = need && mt[0] < best;
            if (ACME_RARE(wv::ballot(hit))) {
                int idx = 0;
                sfor_down<CACHE>([&](auto ec) ACME_LAMBDA { idx = dist[decltype(ec)::value] == mt[0] ? decltype(ec)::value : idx; });
                const int e = hit ? idx : 0;
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    double v = 0.0;
                    if constexpr (CREG) sfor<0, CACHE>([&](auto ec) ACME_LAMBDA { v = e == decltype(ec)::value ? cpr[j][decltype(ec)::value] : v; });
                    else v = cpl[(j * CACHE + e) * LANE_BLOCK];
                    lp[j] = hit ? v : lp[j];
                });
                sfor<0, NN>([&](auto ic) ACME_LAMBDA {
                    constexpr int i = decltype(ic)::value;
                    const double v = cag[S::CACHEPM + e * NN + i];
                    lz[i] = hit ? v : lz[i];
                });
                // set_extrapolation_origin(base, p_c, z_c): re-linearise there; the other lanes keep
                // theirs (their (lp, lz) did not change, so the recomputed values are not taken)
                set_origin(hit);
            }
        }
        const bool c = base_solve(target, need, its);
        if (caching) {
            const bool keep = need && c && its > 5;
            if (keep) {
                const int slot = ccount < CACHE ? ccount : chead;
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (CREG) sfor<0, CACHE>([&](auto ec) ACME_LAMBDA { cpr[j][decltype(ec)::value] = slot == decltype(ec)::value ? target[j] : cpr[j][decltype(ec)::value]; });
                    else cpl[(j * CACHE + slot) * LANE_BLOCK] = target[j];
                });
                sfor<0, NN>([&](auto ic) ACME_LAMBDA { cag[S::CACHEPM + slot * NN + decltype(ic)::value] = z[decltype(ic)::value]; });
                chead = ccount < CACHE ? chead : (chead + 1) & (CACHE - 1);
                ccount = ccount < CACHE ? ccount + 1 : ccount;
            }
        }
        return c;
    };

    // ---- time loop -------------------------------------------------------------------------------
    const long long T = A.T;
    const int nu_io = A.nu_io, ny_io = A.ny_io;
    const double *ug = A.u + ii * T * nu_io;
    double *yg = A.y + ii * T * ny_io;
    // u and y go straight to and from HBM, one sample at a time: the inputs of sample n + 1 are requested
    // before sample n is worked on (a sample takes a lone wave ~2.5 us, several memory latencies), y is
    // stored as it is formed.  A lane touches 8 bytes of its own row per access -- a 64-byte line serves 8
    // consecutive samples out of the L2, so HBM sees every byte once.  (Rounds 1-2 kept 8-sample tiles in
    // registers as shift registers: 8 ... 30 register moves per sample, half of them through AGPRs.)
    auto fetch = [&](long long n, double (&dst)[NUr]) ACME_LAMBDA {
        // (no per-lane predicate: an idle lane reads instance 0's row; a model without inputs has u == NULL --
        // run!(model, zeros(0, T)) -- and takes the uniform branch)
        if (nu_io > 0) {
            sfor<0, NU>([&](auto kc) ACME_LAMBDA {
                constexpr int k = decltype(kc)::value;
                dst[k] = ug[(n < T ? n : T - 1) * nu_io + (k < nu_io ? k : 0)];   // past the end: the last sample again
            });
        } else {
            sfor<0, NU>([&](auto kc) ACME_LAMBDA { dst[decltype(kc)::value] = 0.0; });
        }
    };
    double unext[NUr];
    fetch(0, unext);
    // (waited for before the loop: a value entering the loop in flight makes the compiler wait at its first
    // use in EVERY iteration -- right behind the load of the next one)
    sfor<0, NU>([&](auto kc) ACME_LAMBDA { unext[decltype(kc)::value] = wv::keep(unext[decltype(kc)::value]); });
    {
        for (long long n1 = 0; n1 < T; ++n1) {
            const long long n = A.sample_base + n1;
            const bool alive = !dead;
            double us[NUr];
            sfor<0, NU>([&](auto kc) ACME_LAMBDA { us[decltype(kc)::value] = unext[decltype(kc)::value]; });
            fetch(n1 + 1, unext);
            // p = dq x + eq u  (src/ACME.jl:678-686)
            double p[NPr];
            sfor<0, NP>([&](auto ic) ACME_LAMBDA {
                constexpr int i = decltype(ic)::value;
                const auto R = row_ptr(LL.p0 + i * LL.pstr);
                double acc = 0.0;
                sfor<0, NX>([&](auto jc) ACME_LAMBDA { acc = fma(R[decltype(jc)::value], x[decltype(jc)::value], acc); });
                sfor<0, NU>([&](auto kc) ACME_LAMBDA { acc = fma(R[NX + decltype(kc)::value], us[decltype(kc)::value], acc); });
                p[i] = acc;
            });
            // solve(::HomotopySolver, p) (src/solvers.jl:268-296), per-lane state machine
            bool need = alive, conv = false;
            int mode = 0, its_sample = 0;
            double ha = 0.5, hbest = 0.0, startp[NPr], target[NPr];
            sfor<0, NP>([&](auto jc) ACME_LAMBDA { target[decltype(jc)::value] = p[decltype(jc)::value]; startp[decltype(jc)::value] = 0.0; });
            // The direct attempt first, outside any loop: it almost always settles the sample, and as
            // straight-line code it costs none of the register copies a loop head needs for everything that
            // lives across it (the solution cache, the origin, the state: ~120 moves per sample as a
            // loop).  The bisection loop -- a second, cold copy of the solver -- only runs when an instance's
            // direct attempt failed.
            auto hstep = [&](bool c) ACME_LAMBDA {          // bookkeeping of solve(::HomotopySolver) after one base solve
                const bool direct = need && mode == 0, homot = need && mode == 1;
                const bool start = direct && !c;
                sfor<0, NP>([&](auto jc) ACME_LAMBDA { startp[decltype(jc)::value] = sel(start, lp[decltype(jc)::value], startp[decltype(jc)::value]); });
                const bool hgood = homot && c;
                hbest = sel(hgood, ha, hbest);
                const double new_a = (ha + hbest) / 2.0;
                const bool hbreak = homot && !c && !(hbest < new_a && new_a < ha);
                ha = sel(hgood, 1.0, sel(homot && !c, new_a, ha));
                ha = sel(start, 0.5, ha);
                hbest = sel(start, 0.0, hbest);
                mode = sel(start, 1, mode);
                need = need && !(direct && c) && !hbreak && !(homot && hbest >= 1.0);
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    double pa = startp[j] * (1.0 - ha);
                    pa = pa + ha * p[j];
                    target[j] = sel(need, pa, target[j]);
                });
            };
            {
                int its;
                const bool c = cached_solve(target, need, its);
                its_sample = need ? its : 0;
                conv = need ? c : conv;
                if (ACME_USUAL(A.solver == SOLVER_SIMPLE || !wv::ballot(need && !c))) {
                    need = false;
                } else {
                    hstep(c);
                    while (wv::ballot(need)) {
                        int its2;
                        const bool c2 = cached_solve(target, need, its2);
                        its_sample += need ? its2 : 0;
                        conv = need ? c2 : conv;
                        if (A.solver == SOLVER_SIMPLE || !wv::ballot(need && !(mode == 0 && c2))) need = false;
                        else hstep(c2);
                    }
                }
            }
            // convergence policy of step! (src/ACME.jl:688-694)
            const bool failed = alive && !conv;
            double zchk = 0.0;
            sfor<0, NN>([&](auto ic) ACME_LAMBDA { zchk = fma(z[decltype(ic)::value], 0.0, zchk); });
            const bool zfinite = zchk == 0.0;
            const bool warn = failed && zfinite, die = failed && !zfinite;
            r_nwarn += warn ? 1 : 0;
            r_first_nonconv = (warn && r_first_nonconv < 0) ? n : r_first_nonconv;
            r_first_nonfinite = (die && r_first_nonfinite < 0) ? n : r_first_nonfinite;
            dead = dead || die;
            r_iters_total += alive ? its_sample : 0;
            r_iters_max = (alive && its_sample > r_iters_max) ? its_sample : r_iters_max;
            const bool live = !dead;
            // the next sample's inputs have arrived by now: a use HERE, so that the wait sits at the end of the
            // sample and not -- for a value still in flight across the loop's back edge -- right behind the load;
            // and before y is stored, or it would wait for that store as well
            sfor<0, NU>([&](auto kc) ACME_LAMBDA { unext[decltype(kc)::value] = wv::keep(unext[decltype(kc)::value]); });
            // y = y0 + dy x + ey u + fy z with the OLD x (:699-706), then x = x0 + a x + b u + c z (:708-714)
            sfor<0, NY>([&](auto ic) ACME_LAMBDA {
                constexpr int i = decltype(ic)::value;
                const auto R = row_ptr(LL.y0 + i * LL.xstr);
                double acc = R[0];
                sfor<0, NX>([&](auto jc) ACME_LAMBDA { acc = fma(R[1 + decltype(jc)::value], x[decltype(jc)::value], acc); });
                sfor<0, NU>([&](auto kc) ACME_LAMBDA { acc = fma(R[1 + NX + decltype(kc)::value], us[decltype(kc)::value], acc); });
                sfor<0, NN>([&](auto jc) ACME_LAMBDA { acc = fma(R[1 + NX + NU + decltype(jc)::value], alive ? z[decltype(jc)::value] : 0.0, acc); });
                if (valid && i < ny_io) yg[n1 * ny_io + i] = live ? acc : (double)NAN;
            });
            double xn[NXr];
            sfor<0, NX>([&](auto ic) ACME_LAMBDA {
                constexpr int i = decltype(ic)::value;
                const auto R = row_ptr(LL.x0 + i * LL.xstr);
                double acc = R[0];
                sfor<0, NX>([&](auto jc) ACME_LAMBDA { acc = fma(R[1 + decltype(jc)::value], x[decltype(jc)::value], acc); });
                sfor<0, NU>([&](auto kc) ACME_LAMBDA { acc = fma(R[1 + NX + decltype(kc)::value], us[decltype(kc)::value], acc); });
                sfor<0, NN>([&](auto jc) ACME_LAMBDA { acc = fma(R[1 + NX + NU + decltype(jc)::value], alive ? z[decltype(jc)::value] : 0.0, acc); });
                xn[i] = acc;
            });
            sfor<0, NX>([&](auto ic) ACME_LAMBDA { x[decltype(ic)::value] = sel(live, xn[decltype(ic)::value], x[decltype(ic)::value]); });
        }
    }

    // ---- write back ------------------------------------------------------------------------------
    if (valid) {
        double *so = A.state + inst * S::STATE;
        sfor<0, NX>([&](auto c) ACME_LAMBDA { so[decltype(c)::value] = x[decltype(c)::value]; });
        sfor<0, NP>([&](auto c) ACME_LAMBDA { so[NX + decltype(c)::value] = lp[decltype(c)::value]; });
        sfor<0, NN>([&](auto c) ACME_LAMBDA { so[NX + NP + decltype(c)::value] = lz[decltype(c)::value]; });
        if (caching) {
            int *meta = reinterpret_cast<int *>(cag + NP * CACHE);
            meta[0] = ccount;
            meta[1] = chead;
            sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                if constexpr (CREG) {
                    sfor<0, CACHE>([&](auto ec) ACME_LAMBDA { cag[j * CACHE + decltype(ec)::value] = decltype(ec)::value < ccount ? cpr[j][decltype(ec)::value] : 0.0; });
                } else {
                    for (int e = 0; e < CACHE; ++e) cag[j * CACHE + e] = cpl[(j * CACHE + e) * LANE_BLOCK];
                }
            });
        }
        long long *rp = A.report + inst * RW_WORDS;
        rp[RW_NWARN] = r_nwarn; rp[RW_FIRST_NONCONV] = r_first_nonconv; rp[RW_FIRST_NONFINITE] = r_first_nonfinite;
        rp[RW_ITERS_TOTAL] = r_iters_total; rp[RW_ITERS_MAX] = r_iters_max;
    }
}

}  // namespace acme
