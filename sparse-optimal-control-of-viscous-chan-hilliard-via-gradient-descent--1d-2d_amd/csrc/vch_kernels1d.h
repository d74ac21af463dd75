// vch_kernels1d.h — the 1D hot path (reference: src/1D/Vch_control_1D/{Forward_solver,
// backward_solver,cost_and_function,GD_1D}.py = F1, B1, C1, G1).
//
// One PERSISTENT workgroup per trajectory runs a whole time march (or adjoint sweep) on the
// device: the N+1 <= 4097 nodes of a trajectory are a few values per thread, all per-trajectory
// decisions (Newton stop test, step ceiling, Armijo) are workgroup reductions, and there is no
// host round trip inside a march.  Batches of trajectories fill the 256 CUs.
//
// Linear systems.  The reference solves the dense 2(N+1) Newton system and the dense (N+1)
// adjoint system with LAPACK dgesv (F1:185, B1:94,116).  Both are 2x2-block tridiagonal in the
// unknowns (dphi_i, dmu_i) resp. (p_i, q_i = -(Lp)_i):
//   Newton (F1:111-137):  [ -kappa/2 L + D   -1/2 I ] [dphi]   [-R_phi]
//                         [  I/dt            -1/2 L ] [dmu ] = [-R_mu ]
//   adjoint (B1:99-101):  [ I - tau L   -dt/2 (L - D) ] [p]   [rhs]
//                         [ L            I            ] [q] = [ 0 ]      (<=> (I - tau L + dt/2 L^2 - dt/2 D L) p = rhs)
// They are solved by cyclic reduction: the first LVL levels are formed on the fly from the
// analytic level-0 rows (constant off-diagonal blocks, diagonal block from phi_i), the remaining
// <= 1025 block rows live in LDS (14 doubles per row, SoA), log2 levels forward and backward, then
// the implicit levels are back-substituted.  Any N+1 works (missing neighbours = zero blocks).
#pragma once
#include "vch_common.h"

#ifndef VCH_T1
#define VCH_T1 512
#endif
constexpr int T1 = VCH_T1;              // threads per trajectory workgroup
constexpr int NR_MAX = 1025;            // block rows kept in LDS
constexpr double DSEP1 = 1e-2;          // F1:42

struct Phys1 {
    double tau, gamma, c1, c2, kappa, Lx;
};

struct M2 {            // 2x2 block, row-major
    double a, b, c, d;
};
struct V2 {
    double x, y;
};
__device__ __forceinline__ M2 mm(const M2 &p, const M2 &q) {
    return M2{p.a * q.a + p.b * q.c, p.a * q.b + p.b * q.d, p.c * q.a + p.d * q.c, p.c * q.b + p.d * q.d};
}
__device__ __forceinline__ V2 mv(const M2 &p, const V2 &v) { return V2{p.a * v.x + p.b * v.y, p.c * v.x + p.d * v.y}; }
__device__ __forceinline__ M2 minv(const M2 &p) {
    const double id = 1.0 / (p.a * p.d - p.b * p.c);
    return M2{p.d * id, -p.b * id, -p.c * id, p.a * id};
}
__device__ __forceinline__ M2 madd(const M2 &p, const M2 &q) { return M2{p.a + q.a, p.b + q.b, p.c + q.c, p.d + q.d}; }
__device__ __forceinline__ M2 mneg(const M2 &p) { return M2{-p.a, -p.b, -p.c, -p.d}; }

struct Row {           // A x_{i-s} + B x_i + C x_{i+s} = d
    M2 A, B, C;
    V2 d;
};

// Level-0 rows.  SYS 0: Newton system at iterate phi (F1:111-137); SYS 1: adjoint system at phi_n.
struct SysArgs {
    const double *phi;       // D_i from phi_i
    const double *r0, *r1;   // right-hand side components (r1 may be NULL = 0)
    double dt, a;            // a = 1/h^2
    double tau, c1, c2, kappa;
    int n;
};

template <int SYS>
__device__ __forceinline__ Row row0(const SysArgs &S, int i) {
    const double fl = i == 0 ? 0.0 : (i == S.n - 1 ? 2.0 : 1.0);      // weight of neighbour i-1
    const double fu = i == S.n - 1 ? 0.0 : (i == 0 ? 2.0 : 1.0);      // weight of neighbour i+1
    Row R;
    const double ph = S.phi[i];
    if (SYS == 0) {
        const double D = S.tau / S.dt + 2.0 * S.c1 / (1.0 - ph * ph);           // F1:122-124 (not clipped)
        R.B = M2{S.kappa * S.a + D, -0.5, 1.0 / S.dt, S.a};
        R.A = M2{-0.5 * S.kappa * S.a * fl, 0.0, 0.0, -0.5 * S.a * fl};
        R.C = M2{-0.5 * S.kappa * S.a * fu, 0.0, 0.0, -0.5 * S.a * fu};
    } else {
        const double p = fmin(fmax(ph, -1.0 + 1e-8), 1.0 - 1e-8);               // B1:45-46
        const double D = 2.0 * S.c1 / (1.0 - p * p) - 2.0 * S.c2;
        R.B = M2{1.0 + 2.0 * S.tau * S.a, 0.5 * S.dt * (2.0 * S.a + D), -2.0 * S.a, 1.0};
        R.A = M2{-S.tau * S.a * fl, -0.5 * S.dt * S.a * fl, S.a * fl, 0.0};
        R.C = M2{-S.tau * S.a * fu, -0.5 * S.dt * S.a * fu, S.a * fu, 0.0};
    }
    R.d = V2{S.r0[i], S.r1 ? S.r1[i] : 0.0};
    return R;
}

// one cyclic-reduction step: eliminate the neighbours lo and hi (where present) from row mid.  The neighbours are
// passed by value with presence flags: a `present ? &row : nullptr` argument makes the compiler keep the rows in
// private (scratch) memory, 232 bytes per lane and a memory round trip per field (profiles/r02_1d_scratch.txt).
__device__ __forceinline__ Row cr_reduce(bool hl, const Row &lo, const Row &mid, bool hh, const Row &hi) {
    Row R = mid;
    R.A = M2{0, 0, 0, 0};
    R.C = M2{0, 0, 0, 0};
    if (hl) {
        const M2 al = mneg(mm(mid.A, minv(lo.B)));
        R.A = mm(al, lo.A);
        R.B = madd(R.B, mm(al, lo.C));
        const V2 t = mv(al, lo.d);
        R.d.x += t.x; R.d.y += t.y;
    }
    if (hh) {
        const M2 ga = mneg(mm(mid.C, minv(hi.B)));
        R.C = mm(ga, hi.C);
        R.B = madd(R.B, mm(ga, hi.A));
        const V2 t = mv(ga, hi.d);
        R.d.x += t.x; R.d.y += t.y;
    }
    return R;
}

__device__ __forceinline__ Row row_identity() {          // placeholder for an absent neighbour (never used in arithmetic)
    return Row{M2{0, 0, 0, 0}, M2{1, 0, 0, 1}, M2{0, 0, 0, 0}, V2{0, 0}};
}

// row of the system at implicit level LVL (stride 2^LVL) centred at level-0 index i
template <int SYS, int LVL>
struct RowAt {
    static __device__ __forceinline__ Row get(const SysArgs &S, int i) {
        constexpr int s = 1 << (LVL - 1);
        const Row mid = RowAt<SYS, LVL - 1>::get(S, i);
        const bool hl = i - s >= 0, hh = i + s < S.n;
        const Row lo = hl ? RowAt<SYS, LVL - 1>::get(S, i - s) : row_identity();
        const Row hi = hh ? RowAt<SYS, LVL - 1>::get(S, i + s) : row_identity();
        return cr_reduce(hl, lo, mid, hh, hi);
    }
};
template <int SYS>
struct RowAt<SYS, 0> {
    static __device__ __forceinline__ Row get(const SysArgs &S, int i) { return row0<SYS>(S, i); }
};
template <int SYS>
__device__ __forceinline__ Row row_at(const SysArgs &S, int i, int lvl) {
    if (lvl == 0) return RowAt<SYS, 0>::get(S, i);
    if (lvl == 1) return RowAt<SYS, 1>::get(S, i);
    return RowAt<SYS, 2>::get(S, i);
}

// LDS image of the explicit rows: 14 arrays of NR_MAX doubles (A,B,C: 4 each; d: 2; x overwrites d)
struct CrLds {
    double *v;      // base
    // rows are XOR-swizzled inside their block of 32: the cyclic-reduction levels address rows at strides
    // 2, 4, ..., 1024, which un-swizzled fall on one or two LDS banks; (m >> 5) & 31 is 0 for m = 1024
    __device__ __forceinline__ double &at(int comp, int m) { return v[comp * NR_MAX + (m ^ ((m >> 5) & 31))]; }
    __device__ __forceinline__ Row load(int m) {
        Row R;
        R.A = M2{at(0, m), at(1, m), at(2, m), at(3, m)};
        R.B = M2{at(4, m), at(5, m), at(6, m), at(7, m)};
        R.C = M2{at(8, m), at(9, m), at(10, m), at(11, m)};
        R.d = V2{at(12, m), at(13, m)};
        return R;
    }
    __device__ __forceinline__ void store(int m, const Row &R) {
        at(0, m) = R.A.a; at(1, m) = R.A.b; at(2, m) = R.A.c; at(3, m) = R.A.d;
        at(4, m) = R.B.a; at(5, m) = R.B.b; at(6, m) = R.B.c; at(7, m) = R.B.d;
        at(8, m) = R.C.a; at(9, m) = R.C.b; at(10, m) = R.C.c; at(11, m) = R.C.d;
        at(12, m) = R.d.x; at(13, m) = R.d.y;
    }
};

// Solve the block-tridiagonal system; results x0[i], x1[i] (global).  All T1 threads of the
// workgroup must call it.  lvl = number of implicit levels (0..2), nr = rows kept in LDS.
template <int SYS>
__device__ void cr_solve(const SysArgs &S, int lvl, double *lds, double *x0, double *x1) {
    CrLds Q{lds};
    const int tid = threadIdx.x, n = S.n, st = 1 << lvl;
    const int nr = (n - 1) / st + 1;
    // explicit level: rows m <-> level-0 index m * st
    for (int m = tid; m < nr; m += T1) {
        Q.store(m, row_at<SYS>(S, m * st, lvl));
    }
    __syncthreads();
    int s = 1;
    for (; s < nr; s <<= 1) {                // forward reduction: rows m = 0, 2s, 4s, ...
        // row m is written by its owner only and rows m +- s are not touched at this level, so
        // the update is done in place without a barrier between loads and stores
        for (int m = tid * 2 * s; m < nr; m += T1 * 2 * s) {
            const Row mid = Q.load(m);
            const bool hl = m - s >= 0, hh = m + s < nr;
            const Row lo = hl ? Q.load(m - s) : row_identity();
            const Row hi = hh ? Q.load(m + s) : row_identity();
            Q.store(m, cr_reduce(hl, lo, mid, hh, hi));
        }
        __syncthreads();
    }
    if (tid == 0) {
        Row R = Q.load(0);
        V2 x = mv(minv(R.B), R.d);
        Q.at(12, 0) = x.x; Q.at(13, 0) = x.y;
    }
    __syncthreads();
    for (s >>= 1; s >= 1; s >>= 1) {         // back substitution: rows m = s, 3s, 5s, ...
        for (int m = s + tid * 2 * s; m < nr; m += T1 * 2 * s) {
            Row R = Q.load(m);
            V2 r = R.d;
            V2 t = mv(R.A, V2{Q.at(12, m - s), Q.at(13, m - s)});
            r.x -= t.x; r.y -= t.y;
            if (m + s < nr) {
                t = mv(R.C, V2{Q.at(12, m + s), Q.at(13, m + s)});
                r.x -= t.x; r.y -= t.y;
            }
            V2 x = mv(minv(R.B), r);
            Q.at(12, m) = x.x; Q.at(13, m) = x.y;
        }
        __syncthreads();
    }
    for (int m = tid; m < nr; m += T1) {
        x0[m * st] = Q.at(12, m);
        x1[m * st] = Q.at(13, m);
    }
    __syncthreads();
    // implicit levels: nodes i = s, 3s, ... with s = st/2, ..., 1
    for (int l = lvl - 1; l >= 0; --l) {
        const int sp = 1 << l;
        for (int i = sp + tid * 2 * sp; i < n; i += T1 * 2 * sp) {
            Row R = row_at<SYS>(S, i, l);
            V2 r = R.d;
            V2 t = mv(R.A, V2{x0[i - sp], x1[i - sp]});
            r.x -= t.x; r.y -= t.y;
            if (i + sp < n) {
                t = mv(R.C, V2{x0[i + sp], x1[i + sp]});
                r.x -= t.x; r.y -= t.y;
            }
            V2 x = mv(minv(R.B), r);
            x0[i] = x.x; x1[i] = x.y;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// workgroup reductions (T1 threads); result broadcast to all threads
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double wsum1(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wmin1(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wmax1(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
template <int OP>      // 0 sum, 1 min, 2 max
__device__ __forceinline__ double block_red(double v, double *s4) {
    v = OP == 0 ? wsum1(v) : (OP == 1 ? wmin1(v) : wmax1(v));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = s4[0];
    for (int w = 1; w < T1 / 64; ++w) r = OP == 0 ? r + s4[w] : (OP == 1 ? fmin(r, s4[w]) : fmax(r, s4[w]));
    return r;
}

__device__ __forceinline__ double lap1(const double *v, int i, int n, double a) {      // F1:64-80
    if (i == 0) return 2.0 * a * (v[1] - v[0]);
    if (i == n - 1) return 2.0 * a * (v[n - 2] - v[n - 1]);
    return a * ((v[i - 1] + v[i + 1]) - 2.0 * v[i]);
}
__device__ __forceinline__ double reglog1(double phi) {                                  // F1:57-62
    const double eps = 0.5 * DSEP1;
    const double p = fmin(fmax(phi, -1.0 + eps), 1.0 - eps);
    return log((1.0 + p) / (1.0 - p));
}

// Scratch arrays of one trajectory (global memory, L2 resident): n doubles each.
struct Scr1 {
    double *phi, *mu, *w, *wnew, *phin, *mun, *phit, *mut, *dphi, *dmu, *cphi, *cmu, *rp, *rm;
};
constexpr int NSCR1 = 14;
__device__ __forceinline__ Scr1 scr_of(double *base, int b, int n) {
    double *q = base + (long)b * NSCR1 * n;
    return Scr1{q, q + n, q + 2 * n, q + 3 * n, q + 4 * n, q + 5 * n, q + 6 * n, q + 7 * n, q + 8 * n,
                q + 9 * n, q + 10 * n, q + 11 * n, q + 12 * n, q + 13 * n};
}

struct Newton1Stats {
    int iters;       // residual norms recorded
    int solves;
    int trials;
    int failed_ls;   // left through the line-search-failure return (F1:227-229)
    int error;       // 1 = non-finite mass defect (RuntimeError in the reference, F1:166-170)
};

// residual of (ph, mu_) with the old-level parts pre-combined in cphi, cmu; returns sum of squares
// over this thread's nodes; optionally stores -R into (rp, rm).
__device__ __forceinline__ double resid1(const Phys1 &P, int n, double a, double dt, const double *ph,
                                         const double *mu_, const double *cphi, const double *cmu, double *rp,
                                         double *rm, double *mass_defect_acc, double h) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += T1) {
        const double p = ph[i];
        const double Rp = (P.tau / dt) * p - 0.5 * P.kappa * lap1(ph, i, n, a) + P.c1 * reglog1(p) - 0.5 * mu_[i] + cphi[i];
        const double Rm = p / dt - 0.5 * lap1(mu_, i, n, a) + cmu[i];
        if (rp) { rp[i] = -Rp; rm[i] = -Rm; }
        if (mass_defect_acc) *mass_defect_acc += ((i == 0 || i == n - 1) ? 0.5 : 1.0) * h * Rm;
        acc += Rp * Rp + Rm * Rm;
    }
    return acc;
}

// One implicit time level (F1:139-235) for the trajectory owned by this workgroup.  On entry
// the old level is (S.phi, S.mu, S.w) and S.wnew holds w_new; on exit (S.phin, S.mun) hold the
// result.  hist (may be NULL) receives the residual norms.
__device__ void newton1(const Phys1 &P, int n, double h, double dt, int lvl, Scr1 &S, double *lds, double *s4,
                        double *hist, int hist_cap, Newton1Stats &ST) {
    const double a = 1.0 / (h * h);
    const int tid = threadIdx.x;
    // old-level parts of the residuals, initial guess (phi_old, mu_old) (F1:141-142)
    for (int i = tid; i < n; i += T1) {
        const double ph = S.phi[i];
        S.cphi[i] = -(P.tau / dt) * ph - 0.5 * P.kappa * lap1(S.phi, i, n, a) - 2.0 * P.c2 * ph - 0.5 * S.mu[i] -
                    0.5 * (S.wnew[i] + S.w[i]);
        S.cmu[i] = -ph / dt - 0.5 * lap1(S.mu, i, n, a);
        S.phin[i] = ph;
        S.mun[i] = S.mu[i];
    }
    __syncthreads();
    SysArgs A{S.phin, S.rp, S.rm, dt, a, P.tau, P.c1, P.c2, P.kappa, n};
    for (int k = 0; k < 50; ++k) {                                   // F1:144,156
        double md = 0.0;
        double ss = resid1(P, n, a, dt, S.phin, S.mun, S.cphi, S.cmu, S.rp, S.rm, (k % 10 == 0) ? &md : nullptr, h);
        const double nR = sqrt(block_red<0>(ss, s4));
        if (hist && tid == 0 && ST.iters < hist_cap) hist[ST.iters] = nR;
        ST.iters++;
        if (k % 10 == 0) {                                            // F1:166-170
            const double mdt = block_red<0>(md, s4);
            if (!isfinite(mdt)) { ST.error = 1; return; }
        }
        if (nR < 1e-6) return;                                        // F1:174
        __syncthreads();
        cr_solve<0>(A, lvl, lds, S.dphi, S.dmu);
        ST.solves++;
        // step ceiling (F1:194-212)
        double am = 1e300;
        for (int i = tid; i < n; i += T1) {
            const double d = S.dphi[i], ph = S.phin[i];
            if (d > 0.0) am = fmin(am, (1.0 - DSEP1 - ph) / d);
            else if (d < 0.0) am = fmin(am, (-1.0 + DSEP1 - ph) / d);
        }
        am = block_red<1>(am, s4);
        if (am >= 1e299) am = INFINITY;
        if (!isfinite(am) || am <= 0.0) am = 1.0;
        double alpha = fmin(1.0, 0.9 * am);
        bool accepted = false;
        for (int t = 0; t < 12; ++t) {                                // F1:216-226
            double mx = 0.0;
            for (int i = tid; i < n; i += T1) {
                const double pt = S.phin[i] + alpha * S.dphi[i];
                S.phit[i] = pt;
                S.mut[i] = S.mun[i] + alpha * S.dmu[i];
                mx = fmax(mx, fabs(pt));
            }
            mx = block_red<2>(mx, s4);            // (also orders the stores before the stencil reads)
            if (mx < 1.0 - DSEP1) {
                ST.trials++;
                const double st = resid1(P, n, a, dt, S.phit, S.mut, S.cphi, S.cmu, nullptr, nullptr, nullptr, h);
                const double nt = sqrt(block_red<0>(st, s4));
                if (nt <= (1.0 - 1e-3 * alpha) * nR) {
                    for (int i = tid; i < n; i += T1) { S.phin[i] = S.phit[i]; S.mun[i] = S.mut[i]; }
                    __syncthreads();
                    accepted = true;
                    break;
                }
            }
            alpha *= 0.5;
            __syncthreads();
        }
        if (!accepted) { ST.failed_ls = 1; return; }                  // F1:227-229
    }
}

// ---------------------------------------------------------------------------------
// persistent forward march (F1:286-386): grid = B workgroups
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(T1) void k1d_forward(Phys1 P, int n, double h, int lvl, int M, const double *__restrict__ dts,
                                                  const double *__restrict__ phi0, const double *u, int u_rows,
                                                  long u_stride, double *__restrict__ hist, long hist_stride,
                                                  double *scratch, int *__restrict__ stats /* [B][8] */,
                                                  const int *__restrict__ skip /* [B] or NULL */) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double s4[T1 / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (skip && skip[b]) return;            // line search already over for this trajectory (G1:98-108)
    Scr1 S = scr_of(scratch, b, n);
    const double a = 1.0 / (h * h);
    // phi = phi0, w = 0, mu = initialize_mu(phi, w) (F1:316-324), history rows 0 and 1 (F1:329-336)
    for (int i = tid; i < n; i += T1) {
        const double ph = phi0[(long)b * n + i];
        S.phi[i] = ph;
        S.w[i] = 0.0;
        if (hist) { hist[b * hist_stride + i] = ph; hist[b * hist_stride + n + i] = ph; }
    }
    __syncthreads();
    double ms = 0.0;
    for (int i = tid; i < n; i += T1) {
        const double ph = S.phi[i];
        S.mu[i] = -P.kappa * lap1(S.phi, i, n, a) + (P.c1 * reglog1(ph) - 2.0 * P.c2 * ph) - S.w[i];
        ms += ((i == 0 || i == n - 1) ? 0.5 : 1.0) * h * ph;
    }
    const double mass0 = block_red<0>(ms, s4);
    Newton1Stats ST{0, 0, 0, 0, 0};
    int nfail = 0;
    for (int step = 0; step < M; ++step) {
        const double dt = dts[step];
        const double gdt = P.gamma / dt;
        const double *un = nullptr, *u1 = nullptr;
        if (u) {                                                      // F1:347-353 (hold-last)
            un = u + b * u_stride + (long)step * n;
            u1 = step < u_rows - 1 ? un + n : un;
        }
        for (int i = tid; i < n; i += T1)
            S.wnew[i] = ((gdt - 0.5) * S.w[i] + 0.5 * ((u1 ? u1[i] : 0.0) + (un ? un[i] : 0.0))) / (gdt + 0.5);
        __syncthreads();
        ST.failed_ls = 0;
        newton1(P, n, h, dt, lvl, S, lds, s4, nullptr, 0, ST);
        nfail += ST.failed_ls;
        if (ST.error) break;
        __syncthreads();
        // clip, carry mu/w, uniform mass shift (F1:361-366), store
        double mc = 0.0;
        for (int i = tid; i < n; i += T1) {
            const double ph = fmin(fmax(S.phin[i], -1.0 + DSEP1), 1.0 - DSEP1);
            S.phi[i] = ph;
            S.mu[i] = S.mun[i];
            S.w[i] = S.wnew[i];
            mc += ((i == 0 || i == n - 1) ? 0.5 : 1.0) * h * ph;
        }
        const double shift = (block_red<0>(mc, s4) - mass0) / P.Lx;
        for (int i = tid; i < n; i += T1) {
            const double ph = S.phi[i] - shift;
            S.phi[i] = ph;
            if (hist) hist[b * hist_stride + (long)(step + 2) * n + i] = ph;
        }
        __syncthreads();
    }
    if (tid == 0) {
        stats[b * 8 + 0] = ST.iters; stats[b * 8 + 1] = ST.solves; stats[b * 8 + 2] = ST.trials;
        stats[b * 8 + 3] = nfail; stats[b * 8 + 4] = ST.error;
    }
}

// single Newton call (F1:139-235) for the parity tests: state preloaded in scratch (phi, mu, w, wnew)
__global__ __launch_bounds__(T1) void k1d_newton(Phys1 P, int n, double h, int lvl, double dt, double *scratch,
                                                 double *__restrict__ hist, int hist_cap, int *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double s4[T1 / 64];
    const int b = blockIdx.x;
    Scr1 S = scr_of(scratch, b, n);
    Newton1Stats ST{0, 0, 0, 0, 0};
    newton1(P, n, h, dt, lvl, S, lds, s4, hist ? hist + (long)b * hist_cap : nullptr, hist_cap, ST);
    if (threadIdx.x == 0) {
        stats[b * 8 + 0] = ST.iters; stats[b * 8 + 1] = ST.solves; stats[b * 8 + 2] = ST.trials;
        stats[b * 8 + 3] = ST.failed_ls; stats[b * 8 + 4] = ST.error;
    }
}

// stand-alone block-tridiagonal solves for the kernel-level parity tests
template <int SYS>
__global__ __launch_bounds__(T1) void k1d_solve(SysArgs A0, int lvl, const double *phi, const double *r0, const double *r1,
                                                double *x0, double *x1) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x;
    SysArgs A = A0;
    A.phi = phi + (long)b * A.n;
    A.r0 = r0 + (long)b * A.n;
    A.r1 = r1 ? r1 + (long)b * A.n : nullptr;
    cr_solve<SYS>(A, lvl, lds, x0 + (long)b * A.n, x1 + (long)b * A.n);
}

// ---------------------------------------------------------------------------------
// persistent adjoint sweep (B1:48-126): rows = M+2 history rows; parameters are the frozen
// defaults of B1:29-33 (passed in F).  p, q, r histories [B][rows][n] must be zero-initialised.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(T1) void k1d_backward(Phys1 F, int n, double h, int lvl, int rows,
                                                   const double *__restrict__ t, const double *__restrict__ phi,
                                                   const double *__restrict__ phiQ, const double *__restrict__ phiT,
                                                   double b1, double b2, double *p, double *q, double *r,
                                                   long hs, double *scratch) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double a = 1.0 / (h * h);
    double *rhs = scratch + (long)b * NSCR1 * n, *x1 = rhs + n;
    const double *ph = phi + b * hs;
    const double *pq = phiQ ? phiQ + b * hs : nullptr;
    double *pp = p + b * hs, *qq = q + b * hs, *rr = r + b * hs;
    const int last = rows - 1;
    // terminal: (I - tau L) p = b2 (phi_M - phi_T); q = -L p; r = 0 (B1:93-96)
    for (int i = tid; i < n; i += T1) rhs[i] = b2 * (ph[(long)last * n + i] - (phiT ? phiT[(long)b * n + i] : 0.0));
    __syncthreads();
    SysArgs A{ph + (long)last * n, rhs, nullptr, 0.0, a, F.tau, F.c1, F.c2, 0.0, n};
    cr_solve<1>(A, lvl, lds, pp + (long)last * n, x1);
    for (int i = tid; i < n; i += T1) qq[(long)last * n + i] = -lap1(pp + (long)last * n, i, n, a);
    __syncthreads();
    for (int k = last - 1; k >= 0; --k) {
        const double dt = t[k + 1] - t[k];
        if (dt <= 0.0) continue;                                       // B1:110
        const double *pn = pp + (long)(k + 1) * n, *qn = qq + (long)(k + 1) * n, *rn = rr + (long)(k + 1) * n;
        const double *f0 = ph + (long)k * n, *f1 = ph + (long)(k + 1) * n;
        // rhs = B(phi_{k+1}) p_{k+1} + src,  B p = p + (tau - dt/2 D+) q + dt/2 L q  with q = -L p  (B1:103-113)
        for (int i = tid; i < n; i += T1) {
            const double pc = fmin(fmax(f1[i], -1.0 + 1e-8), 1.0 - 1e-8);
            const double Dp = 2.0 * F.c1 / (1.0 - pc * pc) - 2.0 * F.c2;
            const double src = 0.5 * dt * b1 * ((f0[i] - (pq ? pq[(long)k * n + i] : 0.0)) +
                                               (f1[i] - (pq ? pq[(long)(k + 1) * n + i] : 0.0)));
            rhs[i] = pn[i] + (F.tau - 0.5 * dt * Dp) * qn[i] + 0.5 * dt * lap1(qn, i, n, a) + src;
        }
        __syncthreads();
        SysArgs Ak{f0, rhs, nullptr, dt, a, F.tau, F.c1, F.c2, 0.0, n};
        cr_solve<1>(Ak, lvl, lds, pp + (long)k * n, x1);
        const double gb = (F.gamma - 0.5 * dt) / (F.gamma + 0.5 * dt), gs = (dt * 0.5) / (F.gamma + 0.5 * dt);
        for (int i = tid; i < n; i += T1) {
            const double qk = -lap1(pp + (long)k * n, i, n, a);       // B1:120
            qq[(long)k * n + i] = qk;
            rr[(long)k * n + i] = gb * rn[i] + gs * (qk + qn[i]);     // B1:122-124
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// element-wise / reduction kernels: Laplacian, residuals, cost (C1:55-73), gradient+prox
// (C1:99,111, G1:68-70); grid = (rows or 1, B)
// ---------------------------------------------------------------------------------
__global__ void k1d_lap(int n, double a, const double *__restrict__ v, double *__restrict__ out) {
    const long o = (long)blockIdx.y * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[o + i] = lap1(v + o, i, n, a);
}

__global__ void k1d_residuals(Phys1 P, int n, double a, double dt, const double *pn, const double *po, const double *mn,
                              const double *mo, const double *wn, const double *wo, double *Rp, double *Rm) {
    const long o = (long)blockIdx.y * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Rp[o + i] = (P.tau * (pn[o + i] - po[o + i]) / dt) - 0.5 * P.kappa * (lap1(pn + o, i, n, a) + lap1(po + o, i, n, a)) +
                    (P.c1 * reglog1(pn[o + i]) + (-2.0 * P.c2 * po[o + i])) - 0.5 * (mn[o + i] + mo[o + i]) -
                    0.5 * (wn[o + i] + wo[o + i]);
        Rm[o + i] = (pn[o + i] - po[o + i]) / dt - 0.5 * (lap1(mn + o, i, n, a) + lap1(mo + o, i, n, a));
    }
}

// per (trajectory, row): sum wx (phi-phiQ)^2, sum wx (phi - phiT)^2 (last row), sum wx u^2, sum wx |u|
__global__ __launch_bounds__(T1) void k1d_cost(int n, int rows, const double *__restrict__ wx, const double *__restrict__ phi,
                                               const double *__restrict__ u, const double *__restrict__ pq,
                                               const double *__restrict__ pt, long hs, double *__restrict__ out) {
    __shared__ double s4[T1 / 64];
    const int row = blockIdx.x, b = blockIdx.y;
    const long o = b * hs + (long)row * n;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = threadIdx.x; i < n; i += T1) {
        const double w = wx[i], ph = phi[o + i], uu = u ? u[o + i] : 0.0;
        const double e = ph - (pq ? pq[o + i] : 0.0);
        a0 += w * (e * e);
        a2 += w * (uu * uu);
        a3 += w * fabs(uu);
        if (row == rows - 1) {
            const double e2 = ph - (pt ? pt[(long)b * n + i] : 0.0);
            a1 += w * (e2 * e2);
        }
    }
    a0 = block_red<0>(a0, s4); a1 = block_red<0>(a1, s4); a2 = block_red<0>(a2, s4); a3 = block_red<0>(a3, s4);
    if (threadIdx.x == 0) {
        double *q = out + ((long)b * rows + row) * 4;
        q[0] = a0; q[1] = a1; q[2] = a2; q[3] = a3;
    }
}

__global__ __launch_bounds__(T1) void k1d_grad_prox(int n, const double *__restrict__ u, const double *__restrict__ r, long hs,
                                                    const double *__restrict__ alpha, double b3, double ks, double umin,
                                                    double umax, double *__restrict__ uout, double *__restrict__ chg) {
    __shared__ double s4[T1 / 64];
    const int row = blockIdx.x, b = blockIdx.y, rows = gridDim.x;
    const long o = b * hs + (long)row * n;
    const double al = alpha[b];
    double d2 = 0, n2 = 0;
    for (int i = threadIdx.x; i < n; i += T1) {
        const double uu = u[o + i];
        const double v = uu - al * (r[o + i] + b3 * uu);                 // C1:99, C1:111
        const double sg = v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0);
        const double un = fmin(fmax(sg * fmax(fabs(v) - al * ks, 0.0), umin), umax);    // G1:68-70
        uout[o + i] = un;
        d2 += (un - uu) * (un - uu);
        n2 += uu * uu;
    }
    d2 = block_red<0>(d2, s4); n2 = block_red<0>(n2, s4);
    if (chg && threadIdx.x == 0) { chg[((long)b * rows + row) * 2] = d2; chg[((long)b * rows + row) * 2 + 1] = n2; }
}

// phi_Q = (1 - t/T) phi_initial + (t/T) phi_T, phi_initial = history row 0 (build_targets_1d, G1:238-241)
__global__ __launch_bounds__(T1) void k1d_ramp(int n, const double *__restrict__ tp, const double *__restrict__ phi_hist,
                                               const double *__restrict__ phiT, long hs, double *__restrict__ phiQ) {
    const int row = blockIdx.x, b = blockIdx.y;
    const double f = tp[row];
    for (int i = threadIdx.x; i < n; i += T1)
        phiQ[b * hs + (long)row * n + i] = (1.0 - f) * phi_hist[b * hs + i] + f * phiT[(long)b * n + i];
}

// free energy of every level of a history (F1:243-262): partial sums {sum dphi^2, sum W psi, sum W w phi}
__global__ __launch_bounds__(T1) void k1d_energy(int n, double c1, double c2, double eps, const double *__restrict__ phi,
                                                 const double *__restrict__ w, long hs, double *__restrict__ out) {
    __shared__ double s4[T1 / 64];
    const int row = blockIdx.x, b = blockIdx.y, rows = gridDim.x;
    const long o = b * hs + (long)row * n;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int i = threadIdx.x; i < n; i += T1) {
        const double a = phi[o + i];
        if (i + 1 < n) { const double d = phi[o + i + 1] - a; a0 += d * d; }
        const double wt = (i == 0 || i == n - 1) ? 0.5 : 1.0;
        const double p = fmin(fmax(a, -1.0 + eps), 1.0 - eps);
        a1 += wt * (c1 * ((1.0 + p) * log(1.0 + p) + (1.0 - p) * log(1.0 - p)) - c2 * (p * p));
        if (w) a2 += wt * (w[o + i] * a);
    }
    a0 = block_red<0>(a0, s4); a1 = block_red<0>(a1, s4); a2 = block_red<0>(a2, s4);
    if (threadIdx.x == 0) {
        double *q = out + ((long)b * rows + row) * 4;
        q[0] = a0; q[1] = a1; q[2] = a2; q[3] = 0.0;
    }
}

