// vch_kernels2d.h — hand-written HIP kernels of the 2D hot path (gfx950, wave64, fp64).
//
// All kernels are batched over B independent trajectories (blockIdx.z) and work on
// TX x TY tiles of one padded plane, staged through LDS with a mirrored (reflect) halo so
// that the Neumann boundary rows need no branches: the mirrored ghost v[-1] = v[1]
// reproduces the doubled off-diagonals of the reference's Laplacian (F2:119-121), and
// applying the same rule twice reproduces its L@L (B2:159).
//
// Per-trajectory control flow (Newton convergence, Armijo acceptance, linear-solver
// convergence) lives in a small TrajState record in device memory; stencil kernels read
// it and return early for trajectories that are not in the corresponding phase, tiny
// `fin` kernels (one workgroup per trajectory) update it from per-workgroup partials in a
// fixed order, so results are deterministic run to run.
#pragma once
#include "vch_common.h"

struct TrajState {
    // Newton iteration (F2:323-427)
    int slot;            // which of the two iterate buffers holds (phi+, mu+, R_phi, rhs, D)
    int newton_active;   // Newton loop still running for this trajectory
    int need_trial;      // an Armijo trial (or the initial residual) is pending
    int trial_no;        // halvings done in the current Armijo loop
    int force_accept;    // take the pending trial unconditionally (best-trial fallback F2:420-423)
    int iters;           // residual norms recorded (= len(hist))
    int nsolves;         // linear solves started
    int ntrials;         // Armijo residual evaluations
    int stuck;           // 12 trials failed and no trial improved: state can never change again
    int frozen;          // trajectory sits this march out (its line search has already accepted, G2:128-146)
    double normR;        // ||[R_phi;R_mu]||_2 of the current iterate
    double alpha;        // step of the pending trial
    double best_norm, best_alpha;
    double Dmin, Dmax, dbar;   // range of the Jacobian diagonal, preconditioner shift
    double rho;                // bound of the CG contraction per iteration from the spectrum of P^-1 A (cg_setup)
    double kT;                 // the bound of that spectrum itself: spec(P^-1 A) in [1, kT]
    long newton_total;         // residual norms recorded over the whole march
    // linear solve (preconditioned CG on the Schur system)
    int lin_active, lin_it;
    long lin_total;
    double lin_r0, lin_prev, lin_rel, lin_maxrel;
    // conjugate gradients in the weighted inner product (see k_schur_p)
    double cg_gamma, cg_gamma0, cg_alpha, cg_beta;
    int lin_budget;            // rigorous iteration bound from the spectrum of P^-1 A
    // inexact Newton (forward solves inside a march): relative tolerance of THIS solve, chosen in k_fin_residual so that
    // the Schur residual it leaves -- which is the nonlinear residual of the next iterate up to second-order terms --
    // is a small fraction of the Newton tolerance; lin_maxabs = worst (final relative residual x ||rhs||_2) of a solve
    double lin_reltol, lin_maxabs;
    // starting guess of the step's first / second Newton solve (k_guess): the sweeps of THIS solve start from x = x0 instead
    // of 0 and solve for the deflated right-hand side rhs - A x0; lin_rscale = ||rhs - A x0|| / ||rhs|| keeps lin_maxrel
    // relative to ||rhs||
    int x_primed, guess_pad;
    double lin_rscale;
    double guess_ratio;        // ||rhs - A x0|| / ||rhs|| of this step's guess (uncapped), for the host's choice of the order
    double guess_ratio2;       // the same for the guess of the step's SECOND Newton solve (0 = no such guess this step)
    // per time step, for the host's launch schedule: linear solves started and the longest of them
    int step_solves, step_lin_max;
    int step_lin[4];           // sweeps of the first four solves of the step
    double step_tol[4], step_kT[4];   // their relative tolerances and the bounds kappa_T of the spectrum of P^-1 A
    int step_chn[4];           // ... and the sweeps a Chebyshev solve of them needs (cheb_plan), whichever solver ran
    // reduction-free (Chebyshev) form of the forward solve (vch_fft.h, k_cheb_rows): spec(P^-1 A) in [theta - delta,
    // theta + delta] = [1, kT] and the number of sweeps after which the rigorous bound 1 / T_{n+1}(theta/delta) is below
    // the solve's tolerance -- all known before the solve starts, so no inner product steers it
    int cheb_n;
    int use_cheb;              // THIS solve takes the reduction-free form (its plan is short): decided per trajectory and per
                               // solve on the device, so a trajectory's arithmetic does not depend on its batch mates
    double cheb_theta, cheb_delta;
    int step_form[4];          // use_cheb of the step's first four solves (host: which launch sequences the next step needs)
    // CG form on a diagonal that spans a wide range (a few nodes near |phi| = 1 among ordinary ones): the solve runs on the
    // right-scaled system  P^-1 A S y = P^-1 rhs,  x = S y,  S = dbar / D  (see cg_scaled below)
    int scaled, scaled_pad;
    double step_R[4];          // residual norms of the step's first four iterates (R_0 .. R_3), for the host's statistics
    // ||R_1|| (the residual after the first Newton iteration) of the last three time steps, newest first; 0 = not known yet.
    // It is the quadratic remainder of the first iteration, three to four orders above the Newton tolerance in the bench
    // regime and smooth along a march (ratio from step to step 0.68 .. 1.03 at the 1 % / 99 % quantiles,
    // profiles/r03_first_solve_slack.txt): the first solve of the next step need not be more accurate than a small fraction of it
    double R1_hist[3];
    int lin_took, lin_unconv;  // adjoint: this solve started (its y is valid); solves the enqueued sweeps did not finish
    int cg_pending, cg_pbuf, cg_pad;   // forward CG: step (alpha, p[cg_pbuf]) computed but not yet added to x
    // forward CG, per-iteration state in two copies: iteration k's stencil kernel derives the step of
    // iteration k-1 itself (every workgroup, redundantly), reading copy (k-1)&1 while one workgroup
    // writes copy k&1 -- no workgroup ever reads a field that another one writes in the same launch
    int ci_active[2], ci_it[2];
    double ci_gamma[2];
    // mass fix (F2:565-577)
    double mass0, mass_err, Wint;
    // scratch for cost / change norms
    double aux[4];
};

// gate codes of the preconditioner kernels: 1 = lin_active, 2 / 3 = copy 0 / 1 of the forward CG's per-iteration flag
// 4 = the adjoint solve of this step took place (lin_took)
// 16 + j = the column pass in front of sweep j of a Chebyshev solve (the trajectory is solving and needs that sweep)
// 5 = solving in the CG form (lin_active and not use_cheb), 8 = solving in the reduction-free form
__device__ __forceinline__ bool gate_open(const TrajState &S, int gate) {
    if (gate >= 16) return S.lin_active != 0 && S.use_cheb != 0 && gate - 16 <= S.cheb_n;
    if (gate == 5) return S.lin_active != 0 && S.use_cheb == 0;
    if (gate == 8) return S.lin_active != 0 && S.use_cheb != 0;
    return gate == 1 ? S.lin_active != 0 : (gate == 4 ? S.lin_took != 0 : S.ci_active[gate - 2] != 0);
}

struct Phys {
    double tau, gamma, c1, c2, kappa;
    double LxLy;
};

// ---------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int refl(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return i < 0 ? 0 : i;
}

template <int H>
__device__ __forceinline__ void load_tile(double *s, const double *__restrict__ g, const Geom &G,
                                          int c0, int r0) {
    constexpr int W = TX + 2 * H, HT = TY + 2 * H;
    for (int e = threadIdx.x; e < W * HT; e += NTH) {
        int ly = e / W, lx = e - ly * W;
        int gr = refl(r0 - H + ly, G.ns), gc = refl(c0 - H + lx, G.nf);
        s[e] = g[(long)gr * G.pitch + gc];
    }
}

// s = a + alpha * d on the haloed tile
template <int H>
__device__ __forceinline__ void load_tile_axpy(double *s, const double *__restrict__ a,
                                               const double *__restrict__ d, double alpha,
                                               const Geom &G, int c0, int r0) {
    constexpr int W = TX + 2 * H, HT = TY + 2 * H;
    for (int e = threadIdx.x; e < W * HT; e += NTH) {
        int ly = e / W, lx = e - ly * W;
        int gr = refl(r0 - H + ly, G.ns), gc = refl(c0 - H + lx, G.nf);
        long o = (long)gr * G.pitch + gc;
        s[e] = a[o] + alpha * d[o];
    }
}

// 5-point mirrored-Neumann Laplacian at LDS position p of a tile with row stride W
template <int W>
__device__ __forceinline__ double lap_at(const double *s, int p, double ax, double ay) {
    double c = s[p];
    return ax * ((s[p - 1] + s[p + 1]) - 2.0 * c) + ay * ((s[p - W] + s[p + W]) - 2.0 * c);
}

__device__ __forceinline__ double reglog(double phi) {   // F2:86-102 with delta_sep = 1e-2
    const double eps = 0.5 * DELTA_SEP;                  // max(1e-8, delta_sep/2)
    double p = fmin(fmax(phi, -1.0 + eps), 1.0 - eps);
#ifdef VCH_FAKE_LOG                                      // timing experiment only (wrong numbers): how much of the evaluation
    return 2.0 * p * (1.0 + 0.33 * p * p);               // kernels' time is the logarithm and the quotient?
#else
    return log((1.0 + p) / (1.0 - p));
#endif
}

__device__ __forceinline__ double jac_diag(double phi, double tau_dt, double c1) {   // F2:243-244
    double psq = fmin(fmax(phi * phi, 0.0), 1.0 - DELTA_SEP * DELTA_SEP);
    return tau_dt + 2.0 * c1 / (1.0 - psq);
}

__device__ __forceinline__ double fpp_log(double phi, double c1, double c2) {   // B2:71-72
    double p = fmin(fmax(phi, -1.0 + 1e-8), 1.0 - 1e-8);
    return 2.0 * c1 / (1.0 - p * p) - 2.0 * c2;
}

// trapezoid weight of node (r, c) of the plane as the engine stores it; M = -L is self-adjoint
// in the inner product weighted by it (W L is symmetric)
__device__ __forceinline__ double wdev(int r, int c, const Geom &G) {
    return ((r == 0 || r == G.ns - 1) ? 0.5 : 1.0) * ((c == 0 || c == G.nf - 1) ? 0.5 : 1.0);
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

// Workgroup reduction of up to NPART values per thread (op: 0 sum, 1 min, 2 max); thread 0
// writes the results to part[0..n).  sred must hold NPART*4 doubles.
template <int N>
__device__ __forceinline__ void block_reduce_store(double (&v)[N], const int (&op)[N], double *sred,
                                                   double *part) {
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double r = op[k] == 0 ? wave_sum(v[k]) : (op[k] == 1 ? wave_min(v[k]) : wave_max(v[k]));
        if (lane == 0) sred[k * 4 + wv] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double a = sred[k * 4], b = sred[k * 4 + 1], c = sred[k * 4 + 2], d = sred[k * 4 + 3];
            part[k] = op[k] == 0 ? (a + b) + (c + d)
                                 : (op[k] == 1 ? fmin(fmin(a, b), fmin(c, d)) : fmax(fmax(a, b), fmax(c, d)));
        }
    }
}

// XCD-aware bijective remap of a 1-D block index: workgroups are dealt round-robin over the 8
// XCDs (blocks b and b+8 share an L2), so logically adjacent work items -- which here share
// 128-byte lines -- are given to the SAME XCD in contiguous chunks.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Tile of a stencil workgroup: the workgroups of one plane that land on the same XCD (dispatch index
// mod 8) are given a contiguous, column-major run of tiles, so vertically adjacent tiles -- whose 2-row
// halos are a quarter of a 64 x 16 tile -- meet in one L2; every XCD still gets an equal share of
// every trajectory.  blk doubles as the (bijective) index of the workgroup's reduction partial.
#define TILE_COORDS                                   \
    const int b = blockIdx.z;                         \
    const int nblk = gridDim.x * gridDim.y;           \
    const int blk = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, nblk); \
    const int c0 = (blk / (int)gridDim.y) * TX, r0 = (blk % (int)gridDim.y) * TY; \
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6; \
    (void)blk; (void)nblk; (void)lx; (void)ly0

// ---------------------------------------------------------------------------------
// out = L v                                                  (apply_laplacian, F2:140-152)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_lap(Geom G, const double *__restrict__ v, double *__restrict__ out) {
    TILE_COORDS;
    __shared__ double s[(TY + 2) * (TX + 2)];
    constexpr int W = TX + 2;
    load_tile<1>(s, v + b * G.plane, G, c0, r0);
    __syncthreads();
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf)
            out[b * G.plane + (long)r * G.pitch + c] = lap_at<W>(s, (ly + 1) * W + lx + 1, G.ax, G.ay);
    }
}

// ---------------------------------------------------------------------------------
// mu = -kappa L phi + c1 reglog(phi) - 2 c2 phi - w           (initialize_mu, F2:155-167)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_init_mu(Geom G, Phys P, const double *__restrict__ phi,
                                                 const double *__restrict__ w, double *__restrict__ mu) {
    TILE_COORDS;
    __shared__ double s[(TY + 2) * (TX + 2)];
    constexpr int W = TX + 2;
    load_tile<1>(s, phi + b * G.plane, G, c0, r0);
    __syncthreads();
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p = (ly + 1) * W + lx + 1;
            long o = b * G.plane + (long)r * G.pitch + c;
            double ph = s[p];
            mu[o] = -P.kappa * lap_at<W>(s, p, G.ax, G.ay) + (P.c1 * reglog(ph) - 2.0 * P.c2 * ph) - w[o];
        }
    }
}

// Start of a Newton call: arm the initial residual evaluation (done by one thread of k_prepare: the fields it sets are
// read by no other workgroup of that launch).
__device__ __forceinline__ void newton_begin(TrajState &S) {
    if (S.frozen) {
        S.newton_active = S.need_trial = S.lin_active = 0;
        return;
    }
    S.newton_active = 1;
    S.need_trial = 1;
    S.trial_no = 0;
    S.force_accept = 1;     // the initial residual is always "accepted"
    S.iters = 0;
    S.stuck = 0;
    S.alpha = 0.0;
    S.lin_active = 0;
    S.step_solves = 0;
    S.step_lin_max = 0;
    S.step_lin[0] = S.step_lin[1] = S.step_lin[2] = S.step_lin[3] = 0;
    S.step_chn[0] = S.step_chn[1] = S.step_chn[2] = S.step_chn[3] = 0;
    S.step_form[0] = S.step_form[1] = S.step_form[2] = S.step_form[3] = 0;
    S.use_cheb = 0;
    S.scaled = 0;
    S.x_primed = 0;
    S.lin_rscale = 1.0;
    S.guess_ratio = 1.0;
    S.guess_ratio2 = 0.0;
}

// ---------------------------------------------------------------------------------
// Start of a time step (F2:545-551, F2:350-351).  From the old level (phi, mu, w) and the
// control rows u_n, u_{n+1} (NULL = zeros) form
//   w_new = ((g-1/2) w + (u_n+u_{n+1})/2)/(g+1/2), g = gamma/dt          (solve_w, F2:180-181)
//   mu0   = -kappa L phi + c1 reglog(phi) - 2 c2 phi - w_new            (Newton initial guess)
//   c_phi = -(tau/dt) phi - kappa/2 L phi - 2 c2 phi - mu/2 - (w_new + w)/2   old-level part of R_phi
//   c_mu  = -phi/dt - L mu / 2                                                 old-level part of R_mu
// (wnew_in != NULL: w_new is given, as in a bare newton_raphson call, F2:323)
// so that R_phi = (tau/dt) phi+ - kappa/2 L phi+ + c1 reglog(phi+) - mu+/2 + c_phi and
// R_mu = phi+/dt - L mu+/2 + c_mu (F2:194-221 with the old-level terms pre-combined).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_prepare(Geom G, Phys P, TrajState *__restrict__ st,
                                                 long slot_stride, const double *__restrict__ phi_s,
                                                 const double *__restrict__ mu_s, const double *__restrict__ w,
                                                 const double *__restrict__ un, const double *__restrict__ unp1,
                                                 long u_stride, const double *__restrict__ wnew_in, double dt,
                                                 double *__restrict__ wnew,
                                                 double *__restrict__ mu0, double *__restrict__ cphi,
                                                 double *__restrict__ cmu) {
    TILE_COORDS;
    __shared__ double sp[(TY + 2) * (TX + 2)];
    __shared__ double sm[(TY + 2) * (TX + 2)];
    constexpr int W = TX + 2;
    if (blk == 0 && threadIdx.x == 0) newton_begin(st[b]);
    if (st[b].frozen) return;
    const int slot = st[b].slot;
    load_tile<1>(sp, phi_s + slot * slot_stride + b * G.plane, G, c0, r0);
    load_tile<1>(sm, mu_s + slot * slot_stride + b * G.plane, G, c0, r0);
    __syncthreads();
    const double gdt = P.gamma / dt;
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p = (ly + 1) * W + lx + 1;
            long o = (long)r * G.pitch + c;
            long ob = b * G.plane + o;
            double u0 = un ? un[b * u_stride + o] : 0.0, u1 = unp1 ? unp1[b * u_stride + o] : 0.0;
            double wo = w[ob];
            double wn = wnew_in ? wnew_in[ob] : ((gdt - 0.5) * wo + 0.5 * (u1 + u0)) / (gdt + 0.5);
            double ph = sp[p], lp = lap_at<W>(sp, p, G.ax, G.ay), lm = lap_at<W>(sm, p, G.ax, G.ay);
            wnew[ob] = wn;
            mu0[ob] = -P.kappa * lp + (P.c1 * reglog(ph) - 2.0 * P.c2 * ph) - wn;
            cphi[ob] = -(P.tau / dt) * ph - 0.5 * P.kappa * lp - 2.0 * P.c2 * ph - 0.5 * sm[p] - 0.5 * (wn + wo);
            cmu[ob] = -ph / dt - 0.5 * lm;
        }
    }
}

// ---------------------------------------------------------------------------------
// Residual of a Newton iterate / Armijo trial (F2:357-360, F2:399-408), fused with
// everything the next linear solve needs:
//   MODE 0 (initial residual): iterate = (phi[slot], mu0)      -> written to slot `slot`
//   MODE 1 (trial):            iterate = (phi, mu)[slot] + alpha (dphi, dmu) -> slot 1-slot
// Outputs per node: phi_t, mu_t, R_phi, D = tau/dt + 2c1/(1-clip(phi_t^2)) (F2:243-244) and
// the Schur right-hand side rhs = -R_mu + L R_phi; per workgroup: sum(R_phi^2 + R_mu^2),
// sum(rhs^2), min D, max D.
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NTH) void k_residual(Geom G, Phys P, const TrajState *__restrict__ st,
                                                  long slot_stride, double *__restrict__ phi_s,
                                                  double *__restrict__ mu_s, double *__restrict__ Rphi_s,
                                                  double *__restrict__ rhs_s, double *__restrict__ D_s,
                                                  const double *__restrict__ mu0, const double *__restrict__ dphi,
                                                  const double *__restrict__ dmu, const double *__restrict__ cphi,
                                                  const double *__restrict__ cmu, double dt,
                                                  double *__restrict__ part) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (!S.newton_active || !S.need_trial) return;
    __shared__ double sp[(TY + 4) * (TX + 4)];
    __shared__ double sm[(TY + 2) * (TX + 2)];
    __shared__ double sr[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W2 = TX + 4, W1 = TX + 2;
    const long pb = b * G.plane;
    const int src = S.slot, dst = MODE == 0 ? S.slot : 1 - S.slot;
    if (MODE == 0) {
        load_tile<2>(sp, phi_s + src * slot_stride + pb, G, c0, r0);
        load_tile<1>(sm, mu0 + pb, G, c0, r0);
    } else {
        load_tile_axpy<2>(sp, phi_s + src * slot_stride + pb, dphi + pb, S.alpha, G, c0, r0);
        load_tile_axpy<1>(sm, mu_s + src * slot_stride + pb, dmu + pb, S.alpha, G, c0, r0);
    }
    __syncthreads();
    const double tdt = P.tau / dt;
    // R_phi on the tile + halo 1
    for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
        int ly = e / W1, lxx = e - ly * W1;
        int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
        int p2 = (ly + 1) * W2 + lxx + 1;
        double ph = sp[p2];
        sr[e] = tdt * ph - 0.5 * P.kappa * lap_at<W2>(sp, p2, G.ax, G.ay) + P.c1 * reglog(ph) - 0.5 * sm[e] +
                cphi[pb + (long)gr * G.pitch + gc];
    }
    __syncthreads();
    double acc[4] = {0.0, 0.0, 1e300, -1e300};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
            long o = pb + (long)r * G.pitch + c;
            double ph = sp[p2], rp = sr[p1];
            double rm = ph / dt - 0.5 * lap_at<W1>(sm, p1, G.ax, G.ay) + cmu[o];
            double rh = -rm + lap_at<W1>(sr, p1, G.ax, G.ay);
            double d = jac_diag(ph, tdt, P.c1);
            long od = dst * slot_stride + o;
            if (MODE == 1) phi_s[od] = ph;
            mu_s[od] = sm[p1];
            Rphi_s[od] = rp;
            rhs_s[od] = rh;
            D_s[od] = d;
            acc[0] += rp * rp + rm * rm;
            acc[1] += rh * rh;
            acc[2] = fmin(acc[2], d);
            acc[3] = fmax(acc[3], d);
        }
    }
    const int op[4] = {0, 0, 1, 2};
    block_reduce_store<4>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

// The trial form for the reduction-free solves (cheb_solve): the back substitution dmu = 2 (K dphi + R_phi) (F2:241-253,
// second block row) happens HERE, on the haloed tile, from dphi, D and R_phi of the current iterate -- no kernel between
// the solve and the trial writes dmu (the step ceiling is taken by the solve's last row kernel, k_cheb_rows).  Same
// arithmetic, in the same order, as k_dmu_ceiling followed by k_residual<1>: bit-identical results.
// LDS: phi_t and dphi with halo 2, and ONE halo-1 buffer that holds mu_t first and R_phi afterwards (R_phi at a node needs
// mu_t at that node only, so it is formed in place once the Laplacians of mu_t have been taken).
__global__ __launch_bounds__(NTH) void k_residual2(Geom G, Phys P, const TrajState *__restrict__ st, long slot_stride,
                                                   double *__restrict__ phi_s, double *__restrict__ mu_s,
                                                   double *__restrict__ Rphi_s, double *__restrict__ rhs_s,
                                                   double *__restrict__ D_s, const double *__restrict__ dphi,
                                                   const double *__restrict__ cphi, const double *__restrict__ cmu, double dt,
                                                   double *__restrict__ part) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (!S.newton_active || !S.need_trial) return;
    __shared__ double sp[(TY + 4) * (TX + 4)];
    __shared__ double sd[(TY + 4) * (TX + 4)];
    __shared__ double sm[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W2 = TX + 4, W1 = TX + 2;
    const long pb = b * G.plane;
    const int src = S.slot, dst = 1 - S.slot;
    const double *phi_o = phi_s + src * slot_stride + pb, *mu_o = mu_s + src * slot_stride + pb;
    const double *D_o = D_s + src * slot_stride + pb, *R_o = Rphi_s + src * slot_stride + pb;
    for (int e = threadIdx.x; e < W2 * (TY + 4); e += NTH) {
        int ly = e / W2, lxx = e - ly * W2;
        int gr = refl(r0 - 2 + ly, G.ns), gc = refl(c0 - 2 + lxx, G.nf);
        long o = (long)gr * G.pitch + gc;
        const double d = dphi[pb + o];
        sd[e] = d;
        sp[e] = phi_o[o] + S.alpha * d;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
        int ly = e / W1, lxx = e - ly * W1;
        int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
        int p2 = (ly + 1) * W2 + lxx + 1;
        long o = (long)gr * G.pitch + gc;
        const double dm = 2.0 * ((-0.5 * P.kappa * lap_at<W2>(sd, p2, G.ax, G.ay) + D_o[o] * sd[p2]) + R_o[o]);
        sm[e] = mu_o[o] + S.alpha * dm;
    }
    __syncthreads();
    const double tdt = P.tau / dt;
    double rm[TY / 4], mt[TY / 4];
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        rm[k] = mt[k] = 0.0;
        if (r < G.ns && c < G.nf) {
            int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
            rm[k] = sp[p2] / dt - 0.5 * lap_at<W1>(sm, p1, G.ax, G.ay) + cmu[pb + (long)r * G.pitch + c];
            mt[k] = sm[p1];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
        int ly = e / W1, lxx = e - ly * W1;
        int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
        int p2 = (ly + 1) * W2 + lxx + 1;
        double ph = sp[p2];
        sm[e] = tdt * ph - 0.5 * P.kappa * lap_at<W2>(sp, p2, G.ax, G.ay) + P.c1 * reglog(ph) - 0.5 * sm[e] +
                cphi[pb + (long)gr * G.pitch + gc];
    }
    __syncthreads();
    double acc[4] = {0.0, 0.0, 1e300, -1e300};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
            long od = dst * slot_stride + pb + (long)r * G.pitch + c;
            double ph = sp[p2], rp = sm[p1];
            double rh = -rm[k] + lap_at<W1>(sm, p1, G.ax, G.ay);
            double d = jac_diag(ph, tdt, P.c1);
            phi_s[od] = ph;
            mu_s[od] = mt[k];
            Rphi_s[od] = rp;
            rhs_s[od] = rh;
            D_s[od] = d;
            acc[0] += rp * rp + rm[k] * rm[k];
            acc[1] += rh * rh;
            acc[2] = fmin(acc[2], d);
            acc[3] = fmax(acc[3], d);
        }
    }
    const int op[4] = {0, 0, 1, 2};
    block_reduce_store<4>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

// ---------------------------------------------------------------------------------
// Schur-reduced Newton operator  out = A x = x/dt + M (kappa/2 M x + D x),  M = -L   (13-point), the bare
// operator of vch2d_schur_apply (kernel-level tests); the solver's own application is fused into k_schur_p.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_schur(Geom G, Phys P, const TrajState *__restrict__ st,
                                               long slot_stride, const double *__restrict__ x,
                                               const double *__restrict__ D_s, double dt, double *__restrict__ out) {
    TILE_COORDS;
    const int slot = st ? st[b].slot : 0;
    __shared__ double sx[(TY + 4) * (TX + 4)];
    __shared__ double stt[(TY + 2) * (TX + 2)];
    constexpr int W2 = TX + 4, W1 = TX + 2;
    const long pb = b * G.plane;
    load_tile<2>(sx, x + pb, G, c0, r0);
    __syncthreads();
    const double *Dp = D_s + slot * slot_stride + pb;
    for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
        int ly = e / W1, lxx = e - ly * W1;
        int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
        int p2 = (ly + 1) * W2 + lxx + 1;
        stt[e] = -0.5 * P.kappa * lap_at<W2>(sx, p2, G.ax, G.ay) + Dp[(long)gr * G.pitch + gc] * sx[p2];
    }
    __syncthreads();
    const double idt = 1.0 / dt;
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
            out[pb + (long)r * G.pitch + c] = sx[p2] * idt - lap_at<W1>(stt, p1, G.ax, G.ay);
        }
    }
}

// ---------------------------------------------------------------------------------
// Starting guess for the first Newton solve of a time step.  Along a march the first Newton increments d_n vary smoothly
// from step to step: measured as the solver sees it, ||A (d_n - guess)|| / ||A d_n|| at 512^2, the previous increment
// alone leaves 1e-2, linear extrapolation 1e-4, quadratic 3e-6 and cubic 1e-7 (scripts/r2_extrap.py; the first dozen
// steps of the spinodal transient are the exception, there the order is kept low by the host, forward_core).  So the
// solve starts from x0 = sum_j c_j d_{n-j} (up to GUESS_ORD previous increments, coefficients = polynomial extrapolation of
// the increment rate over the step midpoints): this kernel, between k_residual<0> and k_fin_residual<0>, stores x0 and
// deflates the right-hand side,
//       rhs <- rhs - A x0        (A = the 13-point Schur operator of k_schur, D of the current iterate),
// and replaces the partial of sum rhs^2 (slot 1; the old one moves to slot 4) so that the forcing rule of the solve
// (newton_lin_tol) sees the deflated norm: the sweeps then reach the SAME absolute Schur residual target with fewer
// sweeps.  The solve itself is unchanged -- its first sweep takes x = x0 instead of 0 (TrajState::x_primed).
// Any x0 is valid; non-finite entries of a stale d plane are read as 0.
// ---------------------------------------------------------------------------------
constexpr int GUESS_RING = 8;     // increments kept (a power of two)
constexpr int GUESS_ORD = 8;      // at most this many planes enter one guess (forward: <= 6 increments; adjoint: every second level)
#ifndef VCH_GUESS_BMAX
#define VCH_GUESS_BMAX 32
#endif
constexpr int GUESS_BMAX = VCH_GUESS_BMAX;    // trajectories per context with coefficients of their own (beyond: one common set, row 0)
struct GuessArgs {
    const double *d[GUESS_ORD];   // first Newton increments of steps n-1 .. n-GUESS_ORD, [B][plane]
    // their coefficients (0 = plane not used), PER TRAJECTORY: the order of the extrapolation is chosen for every
    // trajectory from its own history (forward_core), so a trajectory's arithmetic does not depend on its batch mates
    double c[GUESS_BMAX][GUESS_ORD];
    int per_traj;                 // 0: row 0 holds for every trajectory
    __host__ __device__ const double *row(int b) const { return c[per_traj ? b : 0]; }
};
// bit b of the mask a fin kernel gets: trajectory b's right-hand side was deflated by a starting guess in this launch sequence
__host__ __device__ inline unsigned guess_bit(unsigned mask, int b) { return (mask >> (b & 31)) & 1u; }
// which = 0: the step's first solve (right-hand side of the initial residual, iterate slot S.slot);  which = 1: its second
// solve -- the kernel sits between a trial's k_residual<1> and k_fin_residual<1>, works on the slot the trial wrote
// (1 - S.slot) and only for trajectories whose first solve is done (iters == 1); a trial that is then rejected
// simply discards it (the next trial evaluates the right-hand side afresh and this kernel runs again).
__global__ __launch_bounds__(NTH) void k_guess(Geom G, Phys P, const TrajState *__restrict__ st, long slot_stride,
                                               GuessArgs ga, const double *__restrict__ D_s, double dt,
                                               double *__restrict__ rhs_s, double *__restrict__ x0, double *__restrict__ part,
                                               int which) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (!S.newton_active || !S.need_trial) return;
    if (which == 1 && S.iters != 1) return;
    const double *gcf = ga.row(b);
    if (gcf[0] == 0.0) return;             // no guess for this trajectory in this step
    const int slot = which == 1 ? 1 - S.slot : S.slot;
    __shared__ double sx[(TY + 4) * (TX + 4)];
    __shared__ double stt[(TY + 2) * (TX + 2)];
    __shared__ double sred[4];
    constexpr int W2 = TX + 4, W1 = TX + 2;
    const long pb = b * G.plane;
    for (int e = threadIdx.x; e < W2 * (TY + 4); e += NTH) {
        int ly = e / W2, lxx = e - ly * W2;
        int gr = refl(r0 - 2 + ly, G.ns), gc = refl(c0 - 2 + lxx, G.nf);
        long o = pb + (long)gr * G.pitch + gc;
        double v = gcf[0] * ga.d[0][o];
#pragma unroll
        for (int j = 1; j < GUESS_ORD; ++j)
            if (gcf[j] != 0.0) v += gcf[j] * ga.d[j][o];
        sx[e] = isfinite(v) ? v : 0.0;
    }
    __syncthreads();
    const double *Dp = D_s + slot * slot_stride + pb;
    for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
        int ly = e / W1, lxx = e - ly * W1;
        int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
        int p2 = (ly + 1) * W2 + lxx + 1;
        stt[e] = -0.5 * P.kappa * lap_at<W2>(sx, p2, G.ax, G.ay) + Dp[(long)gr * G.pitch + gc] * sx[p2];
    }
    __syncthreads();
    const double idt = 1.0 / dt;
    double acc = 0.0;
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
            long o = pb + (long)r * G.pitch + c, os = slot * slot_stride + o;
            const double rh = rhs_s[os] - (sx[p2] * idt - lap_at<W1>(stt, p1, G.ax, G.ay));
            rhs_s[os] = rh;
            x0[o] = sx[p2];
            acc += rh * rh;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double *pp = part + ((long)b * nblk + blk) * NPART;
        pp[4] = pp[1];
        pp[1] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    }
}

// ---------------------------------------------------------------------------------
// After the linear solve: dphi = x; dmu = 2 (K dphi + R_phi), K = kappa/2 M + D (back
// substitution of the Schur reduction); per workgroup min over nodes of the step-ceiling
// ratio ((+-(1-delta) - phi)/dphi, F2:381-387).
// ---------------------------------------------------------------------------------
// x_out = the finished dphi.  The kernel reads x (+ alpha p for the fused form of vch_fft.h, k_dmu_ceiling_fin, which also
// resolves the last reduction point of the solve) on the haloed tile and never writes what a neighbour reads.
__global__ __launch_bounds__(NTH) void k_dmu_ceiling(Geom G, Phys P, const TrajState *__restrict__ st,
                                                     long slot_stride, const double *__restrict__ x,
                                                     const double *__restrict__ phi_s,
                                                     const double *__restrict__ D_s,
                                                     const double *__restrict__ Rphi_s, double *__restrict__ dmu,
                                                     double *__restrict__ x_out, double *__restrict__ part) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (!S.newton_active || S.need_trial) return;
    __shared__ double sx[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W = TX + 2;
    const long pb = b * G.plane;
    load_tile<1>(sx, x + pb, G, c0, r0);
    __syncthreads();
    double acc[1] = {1e300};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p = (ly + 1) * W + lx + 1;
            long o = pb + (long)r * G.pitch + c, os = S.slot * slot_stride + o;
            double d = sx[p];
            x_out[o] = d;
            dmu[o] = 2.0 * ((-0.5 * P.kappa * lap_at<W>(sx, p, G.ax, G.ay) + D_s[os] * d) + Rphi_s[os]);
            double ph = phi_s[os];
            if (d > 0.0) acc[0] = fmin(acc[0], (1.0 - DELTA_SEP - ph) / d);
            else if (d < 0.0) acc[0] = fmin(acc[0], (-1.0 + DELTA_SEP - ph) / d);
        }
    }
    const int op[1] = {1};
    block_reduce_store<1>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

// (out_phi, out_mu) = J [dphi; dmu]  (F2:241-253), for the kernel-level parity test.
__global__ __launch_bounds__(NTH) void k_jac_apply(Geom G, Phys P, const double *__restrict__ phi,
                                                   const double *__restrict__ dphi, const double *__restrict__ dmu,
                                                   double dt, double *__restrict__ op_, double *__restrict__ om_) {
    TILE_COORDS;
    __shared__ double sa[(TY + 2) * (TX + 2)];
    __shared__ double sb[(TY + 2) * (TX + 2)];
    constexpr int W = TX + 2;
    const long pb = b * G.plane;
    load_tile<1>(sa, dphi + pb, G, c0, r0);
    load_tile<1>(sb, dmu + pb, G, c0, r0);
    __syncthreads();
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p = (ly + 1) * W + lx + 1;
            long o = pb + (long)r * G.pitch + c;
            double d = jac_diag(phi[o], P.tau / dt, P.c1);
            op_[o] = -0.5 * P.kappa * lap_at<W>(sa, p, G.ax, G.ay) + d * sa[p] - 0.5 * sb[p];
            om_[o] = sa[p] / dt - 0.5 * lap_at<W>(sb, p, G.ax, G.ay);
        }
    }
}

// Stand-alone (R_phi, R_mu) in the reference's own form (F2:194-221), for the parity test.
__global__ __launch_bounds__(NTH) void k_residual_plain(Geom G, Phys P, const double *__restrict__ pn,
                                                        const double *__restrict__ po, const double *__restrict__ mn,
                                                        const double *__restrict__ mo, const double *__restrict__ wn,
                                                        const double *__restrict__ wo, double dt,
                                                        double *__restrict__ Rp, double *__restrict__ Rm,
                                                        double *__restrict__ part) {
    TILE_COORDS;
    __shared__ double s[4][(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W = TX + 2;
    const long pb = b * G.plane;
    load_tile<1>(s[0], pn + pb, G, c0, r0);
    load_tile<1>(s[1], po + pb, G, c0, r0);
    load_tile<1>(s[2], mn + pb, G, c0, r0);
    load_tile<1>(s[3], mo + pb, G, c0, r0);
    __syncthreads();
    double acc[1] = {0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p = (ly + 1) * W + lx + 1;
            long o = pb + (long)r * G.pitch + c;
            double rp = (P.tau * (s[0][p] - s[1][p]) / dt) -
                        0.5 * P.kappa * (lap_at<W>(s[0], p, G.ax, G.ay) + lap_at<W>(s[1], p, G.ax, G.ay)) +
                        (P.c1 * reglog(s[0][p]) + (-2.0 * P.c2 * s[1][p])) - 0.5 * (s[2][p] + s[3][p]) -
                        0.5 * (wn[o] + wo[o]);
            double rm = (s[0][p] - s[1][p]) / dt - 0.5 * (lap_at<W>(s[2], p, G.ax, G.ay) + lap_at<W>(s[3], p, G.ax, G.ay));
            Rp[o] = rp;
            Rm[o] = rm;
            acc[0] += rp * rp + rm * rm;
        }
    }
    const int op[1] = {0};
    block_reduce_store<1>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

// ---------------------------------------------------------------------------------
// End of a time step (F2:562-577): clip to +-(1-delta), weighted mass and interior weight
// (pass 1), then subtract mass_error/W_int on the interior nodes (pass 2) and store the
// level into the history.  wts = hx*hy*outer(trapz_x, trapz_y) (F2:528-531), one shared plane.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_mass(Geom G, const TrajState *__restrict__ st, long slot_stride,
                                              const double *__restrict__ phi_s, const double *__restrict__ wts,
                                              int do_clip, double *__restrict__ part) {
    TILE_COORDS;
    __shared__ double sred[NPART * 4];
    if (st[b].frozen) return;
    const int slot = st[b].slot;
    const double hi = 1.0 - DELTA_SEP;
    double acc[2] = {0.0, 0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = (long)r * G.pitch + c;
            double ph = phi_s[slot * slot_stride + b * G.plane + o];
            if (do_clip) ph = fmin(fmax(ph, -hi), hi);
            double w = wts[o];
            acc[0] += w * ph;
            if (fabs(ph) < hi - 5e-3) acc[1] += w;
        }
    }
    const int op[2] = {0, 0};
    block_reduce_store<2>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

// k_post sums the per-workgroup partials of k_mass itself (every workgroup, the same numbers in the same order: the
// first wavefront strides over them like fin_reduce does), so no `fin` launch sits between the two passes.
// The same two pieces serve k_eval<0>, which applies the end of step n to the values it loads at the start of step n + 1
// (PostArgs): post_sums (first wavefront, then a barrier) and PostFix::apply (a pure function of the node's value).
__device__ __forceinline__ void post_sums(const double *__restrict__ part, int nblk, int b, double *sm2) {
    if (threadIdx.x < 64) {
        double a0 = 0.0, a1 = 0.0;
        for (int t = threadIdx.x; t < nblk; t += 64) {
            a0 += part[((long)b * nblk + t) * NPART];
            a1 += part[((long)b * nblk + t) * NPART + 1];
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (threadIdx.x == 0) {
            sm2[0] = a0;
            sm2[1] = a1;
        }
    }
}
struct PostFix {
    double shift, hi;
    bool fix, interior_ok;
    __device__ __forceinline__ PostFix(const double *sm2, double mass0, double LxLy) {
        const double mass_err = sm2[0] - mass0, Wint = sm2[1];
        hi = 1.0 - DELTA_SEP;
        fix = fabs(mass_err) > 1e-16;
        interior_ok = Wint > 0.0;
        shift = fix ? (interior_ok ? mass_err / Wint : mass_err / LxLy) : 0.0;
    }
    __device__ __forceinline__ double apply(double v) const {
        double ph = fmin(fmax(v, -hi), hi);
        if (fix) {
            if (interior_ok) {
                if (fabs(ph) < hi - 5e-3) ph -= shift;
            } else {
                ph = fmin(fmax(ph - shift, -hi), hi);
            }
        }
        return ph;
    }
};
struct PostArgs {
    const double *part;        // k_mass partials of the step that has just ended (own buffer), or NULL: nothing pending
    double *hist;              // history level of that step, or NULL
    long hist_stride;
};

__global__ __launch_bounds__(NTH) void k_post(Geom G, Phys P, const TrajState *__restrict__ st, long slot_stride,
                                              double *__restrict__ phi_s, double *__restrict__ hist_level,
                                              long hist_stride, const double *__restrict__ part) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (S.frozen) return;
    __shared__ double sm[2];
    post_sums(part, nblk, b, sm);
    __syncthreads();
    const PostFix pf(sm, S.mass0, P.LxLy);
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = (long)r * G.pitch + c;
            long os = S.slot * slot_stride + b * G.plane + o;
            const double ph = pf.apply(phi_s[os]);
            phi_s[os] = ph;
            if (hist_level) hist_level[b * hist_stride + o] = ph;
        }
    }
}

// ---------------------------------------------------------------------------------
// Adjoint sweep (B2:212-242).  With q = -L p carried from the previous step,
//   rhs = B(phi_{n+1}) p_{n+1} + src = p + (tau - dt/2 D+) q + dt/2 L q + src,
//   D+ = f''(phi_{n+1}),  src = dt/2 b1 ((phi_n - phiQ_n) + (phi_{n+1} - phiQ_{n+1})),
// and D_n = f''(phi_n) is stored for the A(phi_n) applications; per workgroup min/max D_n.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_adj_rhs(Geom G, Phys P, const double *__restrict__ p,
                                                 const double *__restrict__ q, const double *__restrict__ phin,
                                                 const double *__restrict__ phin1, const double *__restrict__ qn,
                                                 const double *__restrict__ qn1, long hist_stride, double dt,
                                                 double b1, double *__restrict__ rhs, double *__restrict__ Dn,
                                                 double *__restrict__ part) {
    TILE_COORDS;
    __shared__ double sq[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W = TX + 2;
    const long pb = b * G.plane;
    load_tile<1>(sq, q + pb, G, c0, r0);
    __syncthreads();
    double acc[3] = {1e300, -1e300, 0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int pp = (ly + 1) * W + lx + 1;
            long o = (long)r * G.pitch + c, oh = b * hist_stride + o;
            double f0 = phin[oh], f1 = phin1[oh];
            double t0 = qn ? qn[oh] : 0.0, t1 = qn1 ? qn1[oh] : 0.0;
            double src = 0.5 * dt * b1 * ((f0 - t0) + (f1 - t1));
            double dplus = fpp_log(f1, P.c1, P.c2), dn = fpp_log(f0, P.c1, P.c2);
            double v = p[pb + o] + (P.tau - 0.5 * dt * dplus) * sq[pp] + 0.5 * dt * lap_at<W>(sq, pp, G.ax, G.ay) + src;
            rhs[pb + o] = v;
            Dn[pb + o] = dn;
            acc[0] = fmin(acc[0], dn);
            acc[1] = fmax(acc[1], dn);
            acc[2] += v * v;
        }
    }
    const int op[3] = {1, 2, 0};
    block_reduce_store<3>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

// A(phi_n) x = x + (tau + dt/2 D) M x + dt/2 M M x  (= I - tau L + dt/2 L^2 - dt/2 D L, B2:198)
//   MODE 0: out = A x; MODE 1: out = rhs - A x with sum(out^2); MODE 2: out = B x (B2:203)
template <int MODE>
__global__ __launch_bounds__(NTH) void k_adj_op(Geom G, Phys P, const TrajState *__restrict__ st,
                                                const double *__restrict__ x, const double *__restrict__ Dn,
                                                const double *__restrict__ rhs, double dt,
                                                double *__restrict__ out, double *__restrict__ part) {
    TILE_COORDS;
    if (MODE == 1 && !st[b].lin_active) return;
    const double dbar = MODE == 1 ? st[b].dbar : 0.0;
    __shared__ double sx[(TY + 4) * (TX + 4)];
    __shared__ double stt[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W2 = TX + 4, W1 = TX + 2;
    const long pb = b * G.plane;
    load_tile<2>(sx, x + pb, G, c0, r0);
    __syncthreads();
    for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
        int ly = e / W1, lxx = e - ly * W1;
        stt[e] = -lap_at<W2>(sx, (ly + 1) * W2 + lxx + 1, G.ax, G.ay);      // t = M x
    }
    __syncthreads();
    double acc[2] = {0.0, 0.0};
    const double sgn = MODE == 2 ? -1.0 : 1.0;
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
            long o = pb + (long)r * G.pitch + c;
            double t = stt[p1], mt = -lap_at<W1>(stt, p1, G.ax, G.ay);
            double ax_ = sx[p2] + (P.tau + sgn * 0.5 * dt * Dn[o]) * t + sgn * 0.5 * dt * mt;
            if (MODE == 1) {
                double rr = rhs[o] - ax_;
                out[o] = rr;
                acc[0] += wdev(r, c, G) / (Dn[o] - dbar) * (rr * rr);
                acc[1] += rr * rr;
            } else {
                out[o] = ax_;
            }
        }
    }
    if (MODE == 1) {
        const int op[2] = {0, 0};
        block_reduce_store<2>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
    }
}

// p_n = x (already in place); q_n = -L p_n; r_n = gb r_{n+1} + gs (q_n + q_{n+1}) (B2:233-242).
__global__ __launch_bounds__(NTH) void k_adj_finish(Geom G, const double *__restrict__ p,
                                                    const double *__restrict__ q_old, double *__restrict__ q_new,
                                                    double *__restrict__ r_cur, double gb, double gs,
                                                    double *__restrict__ r_hist, double *__restrict__ p_hist,
                                                    double *__restrict__ q_hist, long hist_stride) {
    TILE_COORDS;
    __shared__ double s[(TY + 2) * (TX + 2)];
    constexpr int W = TX + 2;
    const long pb = b * G.plane;
    load_tile<1>(s, p + pb, G, c0, r0);
    __syncthreads();
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int pp = (ly + 1) * W + lx + 1;
            long o = (long)r * G.pitch + c, oh = b * hist_stride + o;
            double qn = -lap_at<W>(s, pp, G.ax, G.ay);
            double rn = q_old ? gb * r_cur[pb + o] + gs * (qn + q_old[pb + o]) : 0.0;
            q_new[pb + o] = qn;
            r_cur[pb + o] = rn;
            if (r_hist) r_hist[oh] = rn;
            if (p_hist) p_hist[oh] = s[pp];
            if (q_hist) q_hist[oh] = qn;
        }
    }
}

// out = a*(x - y) (terminal right-hand side b2 (phi_M - phi_T), B2:183); y may be NULL
__global__ __launch_bounds__(NTH) void k_scaled_diff(Geom G, const double *__restrict__ x, long xs,
                                                     const double *__restrict__ y, long ys, double a,
                                                     double *__restrict__ out, double *__restrict__ part) {
    TILE_COORDS;
    __shared__ double sred[NPART * 4];
    double acc[1] = {0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = (long)r * G.pitch + c;
            double v = a * (x[b * xs + o] - (y ? y[b * ys + o] : 0.0));
            out[b * G.plane + o] = v;
            acc[0] += v * v;
        }
    }
    const int op[1] = {0};
    block_reduce_store<1>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

__global__ __launch_bounds__(NTH) void k_copy_plane(Geom G, const double *__restrict__ src, long ss,
                                                    double *__restrict__ dst, long ds) {
    TILE_COORDS;
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = (long)r * G.pitch + c;
            dst[b * ds + o] = src[b * ss + o];
        }
    }
}

// ---------------------------------------------------------------------------------
// Cost integrands (C2:80-106): per (trajectory, level, tile) partial sums of
//   W (phi - phiQ)^2, W u^2, W |u| and, on the last level, W (phi - phiT)^2,
// W = product trapezoid weights built from the caller's x, y (np.trapz arithmetic).
// grid = (tiles, levels, B)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_cost(Geom G, int tiles_f, const double *__restrict__ phi,
                                              const double *__restrict__ u, const double *__restrict__ pq,
                                              const double *__restrict__ pt, const double *__restrict__ phi0,
                                              const double *__restrict__ tfrac, long hist_stride, int last_level,
                                              const double *__restrict__ W, double *__restrict__ part) {
    const int b = blockIdx.z, lvl = blockIdx.y;
    const int tf = blockIdx.x % tiles_f, ts = blockIdx.x / tiles_f;
    const int c0 = tf * TX, r0 = ts * TY;
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
    __shared__ double sred[NPART * 4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const long lb = b * hist_stride + (long)lvl * G.plane;
    const double tf_ = tfrac ? tfrac[lvl] : 0.0;
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = (long)r * G.pitch + c;
            double w = W[o], ph = phi[lb + o];
            double uu = u ? u[lb + o] : 0.0;
            double tq;
            if (pq) tq = pq[lb + o];
            else if (tfrac) tq = (1.0 - tf_) * phi0[b * G.plane + o] + tf_ * pt[b * G.plane + o];
            else tq = 0.0;
            double e = ph - tq;
            acc[0] += w * (e * e);
            acc[2] += w * (uu * uu);
            acc[3] += w * fabs(uu);
            if (lvl == last_level) {
                double e2 = ph - (pt ? pt[b * G.plane + o] : 0.0);
                acc[1] += w * (e2 * e2);
            }
        }
    }
    const int op[4] = {0, 0, 0, 0};
    block_reduce_store<4>(acc, op, sred, part + (((long)b * gridDim.y + lvl) * gridDim.x + blockIdx.x) * 4);
}

// ---------------------------------------------------------------------------------
// Free energy of every level of a history (F2:256-319), the array taken as the reference takes it:
// shape (A0, A1) row-major, forward differences along both axes, trapezoid weights outer(A0, A1).
// The engine stores that row-major block as ns rows of nf (flat index f -> (f / nf, f % nf), pitched).
// Per workgroup (1024 flat nodes) partials {sum d0^2, sum d1^2, sum W psi, sum W w phi}.
// grid = (ceil(A0*A1/1024), levels, B)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_energy(Geom G, int A0, int A1, double c1, double c2, double eps,
                                                const double *__restrict__ phi, const double *__restrict__ w,
                                                long hist_stride, double *__restrict__ part) {
    __shared__ double sred[NPART * 4];
    const int b = blockIdx.z, lvl = blockIdx.y;
    const long base = b * hist_stride + (long)lvl * G.plane;
    const double *ph = phi + base;
    const double *wp = w ? w + base : nullptr;
    auto at = [&](const double *p, int f) { return p[(long)(f / G.nf) * G.pitch + (f % G.nf)]; };
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int total = A0 * A1;
    for (int k = 0; k < 4; ++k) {
        const int f = blockIdx.x * 1024 + k * NTH + threadIdx.x;
        if (f < total) {
            const int i = f / A1, j = f - i * A1;
            const double a = at(ph, f);
            if (i + 1 < A0) { const double d = at(ph, f + A1) - a; acc[0] += d * d; }
            if (j + 1 < A1) { const double d = at(ph, f + 1) - a; acc[1] += d * d; }
            const double wt = ((i == 0 || i == A0 - 1) ? 0.5 : 1.0) * ((j == 0 || j == A1 - 1) ? 0.5 : 1.0);
            const double p = fmin(fmax(a, -1.0 + eps), 1.0 - eps);
            const double psi = c1 * ((1.0 + p) * log(1.0 + p) + (1.0 - p) * log(1.0 - p)) - c2 * (p * p);
            acc[2] += wt * psi;
            if (wp) acc[3] += wt * at(wp, f) * a;
        }
    }
    const int op[4] = {0, 0, 0, 0};
    block_reduce_store<4>(acc, op, sred, part + (((long)b * gridDim.y + lvl) * gridDim.x + blockIdx.x) * 4);
}

// sums the tile partials of k_cost: out[b][lvl][4]
__global__ void k_cost_fin(int ntiles, const double *__restrict__ part, double *__restrict__ out) {
    const long idx = (long)blockIdx.x;      // b * levels + lvl
    double a[4] = {0, 0, 0, 0};
    for (int t = threadIdx.x; t < ntiles; t += 64)
        for (int k = 0; k < 4; ++k) a[k] += part[(idx * ntiles + t) * 4 + k];
    for (int k = 0; k < 4; ++k) {
        double v = wave_sum(a[k]);
        if (threadIdx.x == 0) out[idx * 4 + k] = v;
    }
}

// ---------------------------------------------------------------------------------
// gradient + proximal step (C2:150, C2:191-198): g = r + b3 u; v = u - alpha g;
// s = sign(v) max(|v| - alpha kappa_s, 0); u+ = clip(s, u_min, u_max).  Also per-workgroup
// sum (u+ - u)^2 and sum u^2 for the relative-change stop rule (G2:375).
// grid = (tiles, levels, B)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void k_grad_prox(Geom G, int tiles_f, const double *__restrict__ u,
                                                   const double *__restrict__ r, long hist_stride,
                                                   const double *__restrict__ alpha, double b3, double ks,
                                                   double umin, double umax, double *__restrict__ uout,
                                                   double *__restrict__ part) {
    const int b = blockIdx.z, lvl = blockIdx.y;
    const int tf = blockIdx.x % tiles_f, ts = blockIdx.x / tiles_f;
    const int c0 = tf * TX, r0 = ts * TY;
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
    __shared__ double sred[NPART * 4];
    const double al = alpha[b];
    const long lb = b * hist_stride + (long)lvl * G.plane;
    double acc[2] = {0.0, 0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int rr = r0 + ly0 + 4 * k, c = c0 + lx;
        if (rr < G.ns && c < G.nf) {
            long o = lb + (long)rr * G.pitch + c;
            double uu = u[o];
            double v = uu - al * (r[o] + b3 * uu);
            double sgn = v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0);
            double s = sgn * fmax(fabs(v) - al * ks, 0.0);
            double un = fmin(fmax(s, umin), umax);
            uout[o] = un;
            acc[0] += (un - uu) * (un - uu);
            acc[1] += uu * uu;
        }
    }
    if (part) {
        const int op[2] = {0, 0};
        block_reduce_store<2>(acc, op, sred, part + (((long)b * gridDim.y + lvl) * gridDim.x + blockIdx.x) * 4);
    }
}

// =================================================================================
// `fin` kernels: one workgroup (64 threads) per trajectory; sum the per-workgroup partials
// in a fixed order and advance the trajectory's state machine.
// =================================================================================
__device__ __forceinline__ void fin_reduce(const double *part, int nblk, int b, double (&out)[NPART],
                                           const int (&op)[NPART], int n) {
    // all n slots in one pass over the workgroups' partials (the loads of a pass are independent), then the n wavefront
    // reductions side by side -- one slot after the other made this a chain of n x (loads + 6 shuffle steps)
    double a[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) a[k] = op[k] == 0 ? 0.0 : (op[k] == 1 ? 1e300 : -1e300);
    for (int t = threadIdx.x; t < nblk; t += 64) {
        const double *p = part + ((long)b * nblk + t) * NPART;
#pragma unroll
        for (int k = 0; k < NPART; ++k)
            if (k < n) {
                const double v = p[k];
                a[k] = op[k] == 0 ? a[k] + v : (op[k] == 1 ? fmin(a[k], v) : fmax(a[k], v));
            }
    }
#pragma unroll
    for (int k = 0; k < NPART; ++k)
        if (k < n) out[k] = op[k] == 0 ? wave_sum(a[k]) : (op[k] == 1 ? wave_min(a[k]) : wave_max(a[k]));
}

// The fin kernels work on a trajectory's record in LDS: the 64 lanes copy it in (and out) together, one lane advances it.
// (A per-thread copy `TrajState S = st[b]` lives in private memory -- the record has arrays indexed at run time -- and
// costs ~120 scratch instructions and 70 one-lane global loads / stores per launch; through a reference to global memory
// every field access is a round trip.)
#ifndef FIN_LDS
#define FIN_LDS 1
#endif
__device__ __forceinline__ void traj_copy_in(TrajState &dst, const TrajState *src) {
    const unsigned *s = reinterpret_cast<const unsigned *>(src);
    unsigned *d = reinterpret_cast<unsigned *>(&dst);
    for (int i = threadIdx.x; i < (int)(sizeof(TrajState) / 4); i += 64) d[i] = s[i];
    __syncthreads();
}
__device__ __forceinline__ void traj_copy_out(TrajState *dst, const TrajState &src) {
    __syncthreads();
    const unsigned *s = reinterpret_cast<const unsigned *>(&src);
    unsigned *d = reinterpret_cast<unsigned *>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(TrajState) / 4); i += 64) d[i] = s[i];
}

constexpr int HIST_CAP = 512;          // >= max_iter + 1 residual norms (F2:353)
constexpr double NEWTON_TOL = 1e-6;    // F2:353
constexpr int NEWTON_MAXIT = 500;      // F2:353
constexpr double ARMIJO_ETA = 1e-4;    // F2:394
constexpr int ARMIJO_TRIALS = 12;      // F2:398

// Mark the trajectories that skip the coming march (flags[b] != 0).
__global__ void k_set_frozen(TrajState *st, const int *__restrict__ flags, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) st[b].frozen = flags[b];
}

// After k_residual: the Armijo test (F2:411-419) or the bookkeeping of the initial residual,
// then the Newton stop test for the new iterate (F2:364, F2:356) and the setup of the next
// linear solve (preconditioner shift dbar = midpoint of the range of D).
// Preconditioner shift and CG iteration budget.  With A = P + M (D - dbar) (forward) or
// A = P + c (D - dbar) M (adjoint, c = dt/2), P the constant-coefficient operator with shift
// dbar < min D, the preconditioned operator is self-adjoint and positive in the inner product
// weighted by W (D - dbar) (forward, left preconditioning) resp. W / (D - dbar) (adjoint, right
// preconditioning), with spectrum in [1, kappa_T], kappa_T <= 1 + c (Dmax - dbar) / (c dbar + g0),
// g0 = lower bound of (P - c dbar M)/M.  CG therefore needs at most
// ln(2/tol) / -ln((sqrt(kappa_T)-1)/(sqrt(kappa_T)+1)) iterations.
__device__ __forceinline__ void cg_setup(TrajState &S, double g0, double cscale, double tol) {
    const double range = S.Dmax - S.Dmin;
    S.dbar = S.Dmin - fmax(1e-12, 0.05 * fmin(range, fabs(S.Dmin) + 1.0));
    const double den = cscale * S.dbar + g0;
    double kT = den > 0.0 ? 1.0 + cscale * (S.Dmax - S.dbar) / den : 1e12;
    const double sq = sqrt(kT);
    const double rate = (sq - 1.0) / (sq + 1.0);
    S.kT = kT;
    S.rho = rate;
    double k = rate > 1e-300 ? log(2.0 / tol) / -log(rate) : 1.0;
    S.lin_budget = (int)fmin(4000.0, ceil(k) + 2.0);
}

// Relative tolerance (Z-norm of the preconditioned residual) of a Newton linear solve inside a march.  The Schur
// residual s = rhs - A dphi that the solve leaves IS the second block row of the nonlinear residual of the next iterate
// (the first row is exact by back substitution), up to second-order terms, and P^-1 A has its spectrum in [1, ~1.04], so
// every component of s falls at the rate of the preconditioned residual.  eta = the absolute target for ||s||_2, a small
// fraction of the Newton tolerance (F2:353): the iteration ends when the reference's does.  A right-hand side above 1e8
// cannot be brought there by one fp64 solve (the 2-norm of s saturates near 1e-13..1e-10 ||rhs|| whatever the Z-norm
// does, which is why the reference's own first Newton step at 512^2 leaves ||R|| ~ 2 from 9e9): such a solve stops at
// 1e-12 and the next Newton iteration, which the reference needs as well, removes the rest.  eta = 0: lin_tol always.
__device__ __forceinline__ double newton_lin_tol(double r0, double lin_tol, double eta) {
    if (!(eta > 0.0) || !(r0 > 0.0)) return lin_tol;
    if (r0 > 1e8) return fmax(lin_tol, 1e-12);
    return fmin(fmax(lin_tol, eta / r0), 0.5);
}

// Options of the forward solves of a march, passed to the fin kernels that set a solve up.
struct SolveOpts {
    int cheb_max;              // plans of at most that many sweeps take the reduction-free form (-1: never, the CG form always)
    double scale_ratio;        // CG form: Dmax > scale_ratio * Dmin switches to the right-scaled system (0: never)
    double eta1_factor;        // first solve of a step: Schur-residual target max(eta, eta1_factor * min ||R_1|| of the last
                               // three steps) (0: eta always)
};

// Right-scaled CG form.  With S = dbar / D (a diagonal in (0, 1]) the Schur operator is
//     A S = (I/dt + kappa/2 M^2)(I - F) + dbar M (I - F) + M D S ... = P - (I/dt + kappa/2 M^2) F,     F = I - S = (D - dbar) / D,
// so  T = P^-1 A S = I - E F  with the spectral multiplier  E = (I/dt + kappa/2 M^2) P^-1  in (0, 1).  T is self-adjoint and
// positive in <x, y>_F = sum W F x y, spectrum in [dbar / Dmax, 1]: the same worst-case bound as the unscaled form, but a node
// whose D is a hundred times its neighbours' -- |phi| within 1e-2 of 1, F ~ 1 -- now contributes an eigenvalue near
// 1 - mean(E) ~ 0.3-0.5 instead of 1 + (D - dbar) mean(M P^-1) ~ 50, and CG, which adapts to the spectrum it meets, needs a
// tenth of the sweeps on the clipped-noise start of BASELINE config 5 (profiles/r03_config5.txt).
// In kernel terms, against the unscaled form  q = p + (M P^-1)((D - dbar) p):  weight (D - dbar) / D instead of D - dbar,
// multiplier -(1/dt + kappa/2 m^2) / P(m) instead of m / P(m), and x = (dbar / D) y at the end.
__device__ __forceinline__ double cg_weight(double D, double dbar, int scaled) {
    const double dl = D - dbar;
    return scaled ? dl / D : dl;
}

// Chebyshev plan of a forward solve: with spec(P^-1 A) in [1, kT] (cg_setup) the iteration
//     y_1 = b~ / theta,   y_{j+1} = y_j + rho_j rho_{j-1} (y_j - y_{j-1}) + (2 rho_j / delta) (b~ - P^-1 A y_j)
// (b~ = P^-1 rhs, rho_0 = delta / theta, rho_j = 1 / (2 theta / delta - rho_{j-1})) leaves, in the norm the stop test of the
// CG form uses, ||z_{n+1}||_Z <= ||z_0||_Z / T_{n+1}(theta / delta) after n sweeps: n is the smallest count for which that
// bound is below the solve's tolerance.  The first iterate costs no sweep (the preconditioner application alone).
constexpr int CHEB_NCAP = 200;
__device__ __forceinline__ void cheb_plan(TrajState &S, double tol) {
    const double kT = S.kT > 1.0 ? S.kT : 1.0;
    S.cheb_theta = 0.5 * (kT + 1.0);
    S.cheb_delta = 0.5 * (kT - 1.0);
    int n = 0;
    if (S.cheb_delta > 0.0) {
        const double sigma = S.cheb_theta / S.cheb_delta;
        double tkm1 = 1.0, tk = sigma;                       // T_0, T_1
        while (n < CHEB_NCAP && !(1.0 / tk <= tol)) {
            const double tn = 2.0 * sigma * tk - tkm1;
            tkm1 = tk;
            tk = tn;
            ++n;
        }
    }
    S.cheb_n = n;
}
// T_n(sigma) / T_{n+1}(sigma): what the bound promises for the last step of an n-sweep solve
__device__ __forceinline__ double cheb_last_factor(double theta, double delta, int n) {
    if (!(delta > 0.0)) return 0.0;
    const double sigma = theta / delta;
    double tkm1 = 1.0, tk = sigma;
    for (int k = 0; k < n; ++k) {
        const double tn = 2.0 * sigma * tk - tkm1;
        tkm1 = tk;
        tk = tn;
        if (!(tk < 1e280)) return 1.0 / (2.0 * sigma);      // far in the asymptotic range
    }
    return tkm1 / tk;
}

template <int MODE>
__device__ __forceinline__ void fin_residual_update(TrajState &S, const double (&v)[NPART], bool primed, double *__restrict__ hist,
                                                    double kappa, double dt, double lin_tol, double eta, SolveOpts so);

template <int MODE>
__global__ void k_fin_residual(TrajState *st, const double *__restrict__ part, int nblk,
                               double *__restrict__ hist, double kappa, double dt, double lin_tol, double eta, int guess,
                               SolveOpts so) {
    const int b = blockIdx.x;
#if FIN_LDS
    __shared__ TrajState S;
    traj_copy_in(S, st + b);
    if (MODE == 2) {               // after k_eval<0> without the fin step inside: the record has not been armed yet
        if (S.frozen) return;
        if (threadIdx.x == 0) {
            newton_begin(S);
            S.slot = 1 - S.slot;
        }
        __syncthreads();
    }
    if (!S.newton_active || !S.need_trial) return;
    double v[NPART];
    const int op[NPART] = {0, 0, 1, 2, 0, 0};
    // guess: k_guess has deflated the right-hand side (MODE 0: of the step's first solve; MODE 1: of its second solve, for
    // a trajectory whose first solve is done); slot 1 holds sum (rhs - A x0)^2, slot 4 sum rhs^2
    const bool primed = guess_bit((unsigned)guess, b) && (MODE != 1 || S.iters == 1);
    fin_reduce(part, nblk, b, v, op, primed ? 5 : 4);
    if (threadIdx.x == 0)
        fin_residual_update<(MODE == 1 ? 1 : 0)>(S, v, primed, hist + (long)b * HIST_CAP, kappa, dt, lin_tol, eta, so);
    traj_copy_out(st + b, S);
#else
    TrajState S = st[b];
    if (MODE == 2) {
        if (S.frozen) return;
        newton_begin(S);
        S.slot = 1 - S.slot;
    }
    if (!S.newton_active || !S.need_trial) return;
    double v[NPART];
    const int op[NPART] = {0, 0, 1, 2, 0, 0};
    const bool primed = guess_bit((unsigned)guess, b) && (MODE != 1 || S.iters == 1);
    fin_reduce(part, nblk, b, v, op, primed ? 5 : 4);
    if (threadIdx.x != 0) return;
    fin_residual_update<(MODE == 1 ? 1 : 0)>(S, v, primed, hist + (long)b * HIST_CAP, kappa, dt, lin_tol, eta, so);
    st[b] = S;
#endif
}

template <int MODE>
__device__ __forceinline__ void fin_residual_update(TrajState &S, const double (&v)[NPART], bool primed, double *__restrict__ hist,
                                                    double kappa, double dt, double lin_tol, double eta, SolveOpts so) {
    const double nt = sqrt(v[0]);
    bool accept;
    if (MODE == 0) {
        accept = true;
    } else {
        S.ntrials++;
        if (S.force_accept) {
            accept = true;
        } else {
            if (nt < S.best_norm) { S.best_norm = nt; S.best_alpha = S.alpha; }
            accept = nt <= (1.0 - ARMIJO_ETA * S.alpha) * S.normR;
        }
    }
    if (accept) {
        if (MODE == 1) S.slot = 1 - S.slot;
        S.normR = nt;
        S.need_trial = 0;
        S.force_accept = 0;
        if (S.iters < HIST_CAP) hist[S.iters] = nt;
        if (S.iters < 4) S.step_R[S.iters] = nt;
        if (S.iters == 1) {                     // the residual after this step's first Newton iteration
            S.R1_hist[2] = S.R1_hist[1];
            S.R1_hist[1] = S.R1_hist[0];
            S.R1_hist[0] = nt;
        }
        S.iters++;
        S.newton_total++;
        // F2:364: converged; F2:356: the loop body runs at most max_iter times
        if (!(nt >= NEWTON_TOL) || S.iters > NEWTON_MAXIT) {
            // note: a NaN norm also ends the loop here (the reference would spin on NaN)
            if (S.iters > NEWTON_MAXIT) S.iters = NEWTON_MAXIT;
            S.newton_active = 0;
            return;
        }
        // next linear solve
        S.Dmin = v[2];
        S.Dmax = v[3];
        S.lin_r0 = sqrt(v[1]);
        S.x_primed = primed ? 1 : 0;
        S.lin_rscale = (primed && v[4] > 0.0) ? fmin(1.0, sqrt(v[1] / v[4])) : 1.0;
        if (MODE == 0) S.guess_ratio = (primed && v[4] > 0.0) ? sqrt(v[1] / v[4]) : 1.0;
        else if (primed) S.guess_ratio2 = v[4] > 0.0 ? sqrt(v[1] / v[4]) : 1.0;
        // Forcing rule.  The Schur residual a solve leaves enters the next iterate's residual additively; what decides the
        // Newton loop is whether that residual is below the tolerance.  After the FIRST iteration of a step it is dominated by
        // the quadratic remainder ||R_1||, known from the previous steps to within a factor (R1_hist): a first solve that
        // leaves eta1_factor (1 %) of it changes ||R_1|| by 1 % -- the decision only if ||R_1|| is within 1 % of the tolerance,
        // and there (||R_1|| < 5e-6) the rule below gives eta itself, as for every later solve.
        double eta_eff = eta;
        if (MODE == 0 && eta > 0.0 && so.eta1_factor > 0.0 && S.R1_hist[0] > 0.0 && S.R1_hist[1] > 0.0 && S.R1_hist[2] > 0.0)
            eta_eff = fmax(eta, so.eta1_factor * fmin(S.R1_hist[0], fmin(S.R1_hist[1], S.R1_hist[2])));
        S.lin_reltol = newton_lin_tol(S.lin_r0, lin_tol, eta_eff);
        cg_setup(S, 2.0 * sqrt(0.5 * kappa / dt), 1.0, S.lin_reltol);
        cheb_plan(S, S.lin_reltol);
        S.use_cheb = (so.cheb_max >= 0 && S.cheb_n <= so.cheb_max) ? 1 : 0;
        S.scaled = (!S.use_cheb && so.scale_ratio > 0.0 && S.Dmax > so.scale_ratio * S.Dmin) ? 1 : 0;
        S.lin_active = 1;
        S.lin_it = 0;
        S.lin_prev = 1e300;
        S.lin_rel = 1.0;
        S.nsolves++;
        S.step_solves++;
        if (S.step_solves <= 4) {
            S.step_tol[S.step_solves - 1] = S.lin_reltol;
            S.step_kT[S.step_solves - 1] = S.kT;
            S.step_chn[S.step_solves - 1] = S.cheb_n;
            S.step_form[S.step_solves - 1] = S.use_cheb;
        }
        // (a zero right-hand side is not special-cased: the solve starts, zeroes x and ends at its first reduction point)
    } else {
        S.alpha *= 0.5;
        S.trial_no++;
        if (S.trial_no >= ARMIJO_TRIALS) {
            if (S.best_norm < S.normR) {     // F2:422-423: fall back to the best trial
                S.alpha = S.best_alpha;
                S.force_accept = 1;
            } else {
                // F2:424-425: state unchanged -> every later iteration repeats this one
                // bit for bit until max_iter; record that and stop.
                S.stuck = 1;
                S.need_trial = 0;
                while (S.iters < NEWTON_MAXIT) {
                    if (S.iters < HIST_CAP) hist[S.iters] = S.normR;
                    S.iters++;
                }
                S.newton_active = 0;
            }
        }
    }
}

// =================================================================================
// Fused evaluation kernels of a march on the stencil-free path (one launch where the separate form has three or four):
//   MODE 0  start of a time step:  k_prepare + k_residual<0> + k_guess(first solve) + k_fin_residual<0>
//   MODE 2  Armijo trial after a reduction-free solve:  k_residual2 + k_guess(second solve) + k_fin_residual<1>
// Same arithmetic on the same numbers in the same order as the separate kernels: bit-identical results.
//
// The `fin` step (sum the workgroups' partials, Armijo / Newton tests, set-up of the next solve) is done by the workgroup
// of a trajectory that finishes LAST: every workgroup stores its partials write-through (agent-scope stores), drains them
// (s_waitcnt vmcnt(0)) and adds 1 to the trajectory's counter (agent-scope atomic, value returned); the one whose add
// returns nblk - 1 reads all partials with agent-scope loads (never through this CU's L1 or a stale line of its XCD's L2:
// MI355X_MICROARCH.md, inter-workgroup visibility, 8-byte stores / loads, one lane signalling for its workgroup), sums them
// in the fixed order of fin_reduce -- whichever workgroup that is, the result has the same bits -- and resets the counter.
// =================================================================================
struct EvalFin {
    unsigned *counter;         // [B], zero between launches
    double *hist;              // [B][HIST_CAP] residual-norm histories
    double kappa, lin_tol, eta;
    SolveOpts so;
};

__device__ __forceinline__ void store_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_agent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// fin_reduce on partials handed over inside a launch (agent-scope loads); same order of operations
__device__ __forceinline__ void fin_reduce_agent(const double *part, int nblk, int b, double (&out)[NPART], const int (&op)[NPART], int n) {
    double a[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) a[k] = op[k] == 0 ? 0.0 : (op[k] == 1 ? 1e300 : -1e300);
    for (int t = threadIdx.x; t < nblk; t += 64) {
        const double *p = part + ((long)b * nblk + t) * NPART;
#pragma unroll
        for (int k = 0; k < NPART; ++k)
            if (k < n) {
                const double v = load_agent(p + k);
                a[k] = op[k] == 0 ? a[k] + v : (op[k] == 1 ? fmin(a[k], v) : fmax(a[k], v));
            }
    }
#pragma unroll
    for (int k = 0; k < NPART; ++k)
        if (k < n) out[k] = op[k] == 0 ? wave_sum(a[k]) : (op[k] == 1 ? wave_min(a[k]) : wave_max(a[k]));
}

#ifndef EVAL_MINBLK
#define EVAL_MINBLK 5          // workgroups per CU the register allocation must allow (LDS allows 5)
#endif
#ifndef EVAL_PREFETCH
#define EVAL_PREFETCH 1        // 1: every global operand of the later phases is loaded into registers BEFORE the first barrier,
                               // so a workgroup pays the global-memory latency once instead of once per phase (march at 8
                               // trajectories 0.533 -> 0.517 s, bench 5.17 -> 5.31; with the register cap lifted to fit them
                               // all, 122 VGPRs = 4 workgroups per CU: 0.551 s; profiles/r03_fused_ab.txt)
#endif
// FIN_INSIDE: the hand-off variant (VCH_FUSED=1).  The default variant ends with the partials and carries neither the record
// copy of the fin step (584 B of private memory per thread) nor the counter traffic.
template <int MODE, bool FIN_INSIDE>
__global__ __launch_bounds__(NTH, EVAL_MINBLK) void k_eval(Geom G, Phys P, TrajState *st, long slot_stride, double *phi_s, double *mu_s,
                                              double *Rphi_s, double *rhs_s, double *D_s, const double *dphi, double *cphi,
                                              double *cmu, double dt, double *part, const double *w, const double *un,
                                              const double *unp1, long u_stride, double *wnew, GuessArgs ga, double *x0,
                                              EvalFin fin, PostArgs post) {
    TILE_COORDS;
    // only the few fields the evaluation needs are read here; the record as a whole is read, advanced and written back by
    // one thread of the workgroup that finishes last (a per-thread copy of the record would live in private memory)
    const int S_slot = st[b].slot, S_iters = MODE == 0 ? 0 : st[b].iters;
    const double S_alpha = MODE == 0 ? 0.0 : st[b].alpha;
    if (MODE == 0) {
        if (st[b].frozen) {    // what newton_begin does for a trajectory that sits this march out
            if (blk == 0 && threadIdx.x == 0) st[b].newton_active = st[b].need_trial = st[b].lin_active = 0;
            return;
        }
    } else {
        if (!st[b].newton_active || !st[b].need_trial) return;
    }
    __shared__ double sp[(TY + 4) * (TX + 4)];
    __shared__ double sd[MODE == 2 ? (TY + 4) * (TX + 4) : 1];
    __shared__ double sm[(TY + 2) * (TX + 2)];
    __shared__ double sr[MODE == 0 ? (TY + 2) * (TX + 2) : 1];
    __shared__ double sred[NPART * 4];
    __shared__ int s_last;
    constexpr int W2 = TX + 4, W1 = TX + 2;
    const long pb = b * G.plane;
    // both modes write the evaluated iterate into the OTHER slot: in MODE 0 the new mu (the Newton start value) must not
    // land where a neighbouring workgroup may still be loading the halo of the old one
    const int src = S_slot, dst = 1 - S_slot;
    const double tdt = P.tau / dt;
    const double *gcf = ga.row(b);
    const bool do_guess = gcf[0] != 0.0 && (MODE == 0 || S_iters == 1);
    double rm[TY / 4], rh[TY / 4];
    double acc[5] = {0.0, 0.0, 1e300, -1e300, 0.0};
    if (MODE == 0) {
        // ---- k_prepare + k_residual<0> ----
        constexpr int N1 = (TY + 2) * W1, I1 = (N1 + NTH - 1) / NTH;      // halo-1 elements, per thread
        if (post.part) {
            // the end of the previous step (F2:562-577: clip, mass fix, history) applied to the level as it is loaded: the sums
            // of k_mass's partials by the first wavefront while everybody's loads are in flight, then the fix in registers
            __shared__ double s_post[2];
            constexpr int I2 = (W2 * (TY + 4) + NTH - 1) / NTH;
            double raw[I2];
#pragma unroll
            for (int i = 0; i < I2; ++i) {
                const int e = threadIdx.x + i * NTH;
                raw[i] = 0.0;
                if (e < W2 * (TY + 4)) {
                    int ly = e / W2, lxx = e - ly * W2;
                    int gr = refl(r0 - 2 + ly, G.ns), gc = refl(c0 - 2 + lxx, G.nf);
                    raw[i] = phi_s[src * slot_stride + pb + (long)gr * G.pitch + gc];
                }
            }
            post_sums(post.part, nblk, (int)b, s_post);
            __syncthreads();
            const PostFix pf(s_post, st[b].mass0, P.LxLy);
#pragma unroll
            for (int i = 0; i < I2; ++i) {
                const int e = threadIdx.x + i * NTH;
                if (e < W2 * (TY + 4)) sp[e] = pf.apply(raw[i]);
            }
        } else {
            load_tile<2>(sp, phi_s + src * slot_stride + pb, G, c0, r0);
        }
        load_tile<1>(sm, mu_s + src * slot_stride + pb, G, c0, r0);          // mu of the old level
#if EVAL_PREFETCH
        double vW[I1], vU0[I1], vU1[I1];
#pragma unroll
        for (int i = 0; i < I1; ++i) {
            const int e = threadIdx.x + i * NTH;
            vW[i] = vU0[i] = vU1[i] = 0.0;
            if (e < N1) {
                int ly = e / W1, lxx = e - ly * W1;
                int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
                long o = (long)gr * G.pitch + gc;
                vW[i] = w[pb + o];
                vU0[i] = un ? un[b * u_stride + o] : 0.0;
                vU1[i] = unp1 ? unp1[b * u_stride + o] : 0.0;
            }
        }
#endif
        __syncthreads();
        for (int k = 0; k < TY / 4; ++k) {                                   // c_mu needs the Laplacian of the old mu
            int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
            rm[k] = 0.0;
            if (r < G.ns && c < G.nf) {
                int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
                const double cm = -sp[p2] / dt - 0.5 * lap_at<W1>(sm, p1, G.ax, G.ay);
                cmu[pb + (long)r * G.pitch + c] = cm;
                rm[k] = cm;
            }
        }
        __syncthreads();
        const double gdt = P.gamma / dt;
#pragma unroll
        for (int i = 0; i < I1; ++i) {
            const int e = threadIdx.x + i * NTH;
            if (e >= N1) continue;
            int ly = e / W1, lxx = e - ly * W1;
            int rr = r0 - 1 + ly, cc = c0 - 1 + lxx;
            int gr = refl(rr, G.ns), gc = refl(cc, G.nf);
            int p2 = (ly + 1) * W2 + lxx + 1;
            long o = (long)gr * G.pitch + gc, ob = pb + o;
#if EVAL_PREFETCH
            const double u0 = vU0[i], u1 = vU1[i], wo = vW[i];
#else
            const double u0 = un ? un[b * u_stride + o] : 0.0, u1 = unp1 ? unp1[b * u_stride + o] : 0.0;
            const double wo = w[ob];
#endif
            const double wn = ((gdt - 0.5) * wo + 0.5 * (u1 + u0)) / (gdt + 0.5);
            const double ph = sp[p2], lp = lap_at<W2>(sp, p2, G.ax, G.ay);
            const double m0 = -P.kappa * lp + (P.c1 * reglog(ph) - 2.0 * P.c2 * ph) - wn;
            const double cp = -(P.tau / dt) * ph - 0.5 * P.kappa * lp - 2.0 * P.c2 * ph - 0.5 * sm[e] - 0.5 * (wn + wo);
            sr[e] = tdt * ph - 0.5 * P.kappa * lp + P.c1 * reglog(ph) - 0.5 * m0 + cp;
            sm[e] = m0;                                                      // mu of the old level is not needed any more
            if (ly >= 1 && ly <= TY && lxx >= 1 && lxx <= TX && rr < G.ns && cc < G.nf) {   // this workgroup's own nodes
                wnew[ob] = wn;
                cphi[ob] = cp;
            }
        }
        __syncthreads();
        for (int k = 0; k < TY / 4; ++k) {
            int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
            rh[k] = 0.0;
            if (r < G.ns && c < G.nf) {
                int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
                long od = dst * slot_stride + pb + (long)r * G.pitch + c;
                const double ph = sp[p2], rp = sr[p1];
                const double rmv = ph / dt - 0.5 * lap_at<W1>(sm, p1, G.ax, G.ay) + rm[k];
                const double rhv = -rmv + lap_at<W1>(sr, p1, G.ax, G.ay);
                const double d = jac_diag(ph, tdt, P.c1);
                phi_s[od] = ph;
                if (post.part && post.hist) post.hist[b * post.hist_stride + (long)r * G.pitch + c] = ph;
                mu_s[od] = sm[p1];
                Rphi_s[od] = rp;
                D_s[od] = d;
                rh[k] = rhv;
                acc[0] += rp * rp + rmv * rmv;
                acc[2] = fmin(acc[2], d);
                acc[3] = fmax(acc[3], d);
            }
        }
    } else {
        // ---- k_residual2 ----
        const double *phi_o = phi_s + src * slot_stride + pb, *mu_o = mu_s + src * slot_stride + pb;
        const double *D_o = D_s + src * slot_stride + pb, *R_o = Rphi_s + src * slot_stride + pb;
        constexpr int N1 = (TY + 2) * W1, I1 = (N1 + NTH - 1) / NTH;      // halo-1 elements, per thread
#if EVAL_PREFETCH
        // phase-1 loads first (the compiler's vmcnt then waits for them only), the later phases' operands right behind
        double pd[(W2 * (TY + 4) + NTH - 1) / NTH], pp[(W2 * (TY + 4) + NTH - 1) / NTH];
        {
            int i = 0;
#pragma unroll
            for (int e = threadIdx.x; e < W2 * (TY + 4); e += NTH, ++i) {
                int ly = e / W2, lxx = e - ly * W2;
                int gr = refl(r0 - 2 + ly, G.ns), gc = refl(c0 - 2 + lxx, G.nf);
                long o = (long)gr * G.pitch + gc;
                pd[i] = dphi[pb + o];
                pp[i] = phi_o[o];
            }
        }
        double vD[I1], vR[I1], vM[I1], vC[I1], vcm[TY / 4];
#pragma unroll
        for (int i = 0; i < I1; ++i) {
            const int e = threadIdx.x + i * NTH;
            vD[i] = vR[i] = vM[i] = vC[i] = 0.0;
            if (e < N1) {
                int ly = e / W1, lxx = e - ly * W1;
                int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
                long o = (long)gr * G.pitch + gc;
                vD[i] = D_o[o];
                vR[i] = R_o[o];
                vM[i] = mu_o[o];
                vC[i] = cphi[pb + o];
            }
        }
#pragma unroll
        for (int k = 0; k < TY / 4; ++k) {
            int r = r0 + ly0 + 4 * k, c = c0 + lx;
            vcm[k] = (r < G.ns && c < G.nf) ? cmu[pb + (long)r * G.pitch + c] : 0.0;
        }
        {
            int i = 0;
#pragma unroll
            for (int e = threadIdx.x; e < W2 * (TY + 4); e += NTH, ++i) {
                sd[e] = pd[i];
                sp[e] = pp[i] + S_alpha * pd[i];
            }
        }
#else
        for (int e = threadIdx.x; e < W2 * (TY + 4); e += NTH) {
            int ly = e / W2, lxx = e - ly * W2;
            int gr = refl(r0 - 2 + ly, G.ns), gc = refl(c0 - 2 + lxx, G.nf);
            long o = (long)gr * G.pitch + gc;
            const double d = dphi[pb + o];
            sd[e] = d;
            sp[e] = phi_o[o] + S_alpha * d;
        }
#endif
        __syncthreads();
#pragma unroll
        for (int i = 0; i < I1; ++i) {
            const int e = threadIdx.x + i * NTH;
            if (e < N1) {
                int ly = e / W1, lxx = e - ly * W1;
                int p2 = (ly + 1) * W2 + lxx + 1;
#if EVAL_PREFETCH
                const double Dv = vD[i], Rv = vR[i], Mv = vM[i];
#else
                int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
                long o = (long)gr * G.pitch + gc;
                const double Dv = D_o[o], Rv = R_o[o], Mv = mu_o[o];
#endif
                const double dm = 2.0 * ((-0.5 * P.kappa * lap_at<W2>(sd, p2, G.ax, G.ay) + Dv * sd[p2]) + Rv);
                sm[e] = Mv + S_alpha * dm;
            }
        }
        __syncthreads();
        double mt[TY / 4];
        for (int k = 0; k < TY / 4; ++k) {
            int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
            rm[k] = mt[k] = 0.0;
            if (r < G.ns && c < G.nf) {
                int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
#if EVAL_PREFETCH
                const double cm = vcm[k];
#else
                const double cm = cmu[pb + (long)r * G.pitch + c];
#endif
                rm[k] = sp[p2] / dt - 0.5 * lap_at<W1>(sm, p1, G.ax, G.ay) + cm;
                mt[k] = sm[p1];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < I1; ++i) {
            const int e = threadIdx.x + i * NTH;
            if (e < N1) {
                int ly = e / W1, lxx = e - ly * W1;
                int p2 = (ly + 1) * W2 + lxx + 1;
#if EVAL_PREFETCH
                const double cp = vC[i];
#else
                int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
                const double cp = cphi[pb + (long)gr * G.pitch + gc];
#endif
                double ph = sp[p2];
                sm[e] = tdt * ph - 0.5 * P.kappa * lap_at<W2>(sp, p2, G.ax, G.ay) + P.c1 * reglog(ph) - 0.5 * sm[e] + cp;
            }
        }
        __syncthreads();
        for (int k = 0; k < TY / 4; ++k) {
            int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
            rh[k] = 0.0;
            if (r < G.ns && c < G.nf) {
                int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
                long od = dst * slot_stride + pb + (long)r * G.pitch + c;
                double ph = sp[p2], rp = sm[p1];
                double rhv = -rm[k] + lap_at<W1>(sm, p1, G.ax, G.ay);
                double d = jac_diag(ph, tdt, P.c1);
                phi_s[od] = ph;
                mu_s[od] = mt[k];
                Rphi_s[od] = rp;
                D_s[od] = d;
                rh[k] = rhv;
                acc[0] += rp * rp + rm[k] * rm[k];
                acc[2] = fmin(acc[2], d);
                acc[3] = fmax(acc[3], d);
            }
        }
    }
    // ---- k_guess: x0 = sum c_j d_j, rhs -= A x0 (A with the D of the iterate just evaluated), sum rhs^2 before / after ----
    if (do_guess) {
        double *X = MODE == 0 ? sp : sd;          // x0 with halo 2 (MODE 0: phi is not needed once D has been taken)
        double *STT = MODE == 0 ? sr : sp;        // kappa/2 M x0 + D x0 with halo 1
        __syncthreads();                          // everybody is done with sm / sr (and sd)
        for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
            int ly = e / W1, lxx = e - ly * W1;
            sm[e] = jac_diag(sp[(ly + 1) * W2 + lxx + 1], tdt, P.c1);        // D on the tile + halo 1
        }
        __syncthreads();
        for (int e = threadIdx.x; e < W2 * (TY + 4); e += NTH) {
            int ly = e / W2, lxx = e - ly * W2;
            int gr = refl(r0 - 2 + ly, G.ns), gc = refl(c0 - 2 + lxx, G.nf);
            long o = pb + (long)gr * G.pitch + gc;
            double v = gcf[0] * ga.d[0][o];
#pragma unroll
            for (int j = 1; j < GUESS_ORD; ++j)
                if (gcf[j] != 0.0) v += gcf[j] * ga.d[j][o];
            X[e] = isfinite(v) ? v : 0.0;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {             // STT is neither X nor sm in either mode
            int ly = e / W1, lxx = e - ly * W1;
            int p2 = (ly + 1) * W2 + lxx + 1;
            STT[e] = -0.5 * P.kappa * lap_at<W2>(X, p2, G.ax, G.ay) + sm[e] * X[p2];
        }
        double xv[TY / 4];
        for (int k = 0; k < TY / 4; ++k) xv[k] = X[(ly0 + 4 * k + 2) * W2 + lx + 2];
        __syncthreads();
        const double idt = 1.0 / dt;
        for (int k = 0; k < TY / 4; ++k) {
            int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
            if (r < G.ns && c < G.nf) {
                int p1 = (ly + 1) * W1 + lx + 1;
                const double full = rh[k];
                const double defl = full - (xv[k] * idt - lap_at<W1>(STT, p1, G.ax, G.ay));
                x0[pb + (long)r * G.pitch + c] = xv[k];
                acc[4] += full * full;
                rh[k] = defl;
            }
        }
    }
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            rhs_s[dst * slot_stride + pb + (long)r * G.pitch + c] = rh[k];
            acc[1] += rh[k] * rh[k];
        }
    }
    // ---- per-workgroup partials, handed to the workgroup that finishes last ----
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int op5[5] = {0, 0, 1, 2, 0};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            double r_ = op5[k] == 0 ? wave_sum(acc[k]) : (op5[k] == 1 ? wave_min(acc[k]) : wave_max(acc[k]));
            if (lane == 0) sred[k * 4 + wv] = r_;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double *pp = part + ((long)b * nblk + blk) * NPART;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const double a = sred[k * 4], bb = sred[k * 4 + 1], cc = sred[k * 4 + 2], dd = sred[k * 4 + 3];
                const double v = op5[k] == 0 ? (a + bb) + (cc + dd)
                                             : (op5[k] == 1 ? fmin(fmin(a, bb), fmin(cc, dd)) : fmax(fmax(a, bb), fmax(cc, dd)));
                if (FIN_INSIDE) store_agent(pp + k, v);
                else pp[k] = v;
            }
            if (FIN_INSIDE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned old = __hip_atomic_fetch_add(fin.counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = old == (unsigned)(nblk - 1) ? 1 : 0;
            }
        }
        if (!FIN_INSIDE) return;                  // a k_fin_residual launch follows
        __syncthreads();
        if (!s_last || threadIdx.x >= 64) return;
    }
    // ---- k_fin_residual, by the first wavefront of the last workgroup ----
    if (FIN_INSIDE) {
        double v[NPART];
        const int op[NPART] = {0, 0, 1, 2, 0, 0};
        fin_reduce_agent(part, nblk, b, v, op, do_guess ? 5 : 4);
        if (threadIdx.x != 0) return;
        __hip_atomic_store(fin.counter + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        TrajState S = st[b];
        if (MODE == 0) {
            newton_begin(S);
            S.slot = dst;                         // the evaluated start iterate lives there (a trial flips the slot when accepted)
        }
        fin_residual_update<(MODE == 0 ? 0 : 1)>(S, v, do_guess, fin.hist + (long)b * HIST_CAP, fin.kappa, dt, fin.lin_tol, fin.eta,
                                                 fin.so);
        st[b] = S;
    }
}

// ---- CG scalar updates (one thread per trajectory does the arithmetic) ----
// gpart: partials written by the GEMM epilogue (gnblk per trajectory, 1 value each)
__device__ __forceinline__ double fin_sum1(const double *part, int n, int b, int stride, int k) {
    double a = 0.0;
    for (int t = threadIdx.x; t < n; t += 64) a += part[((long)b * n + t) * stride + k];
    return wave_sum(a);
}

// forward set-up: gamma0 = <z, z>_Z with z = P^-1 rhs (from the GEMM epilogue)
__global__ void k_fin_cg_init(TrajState *st, const double *__restrict__ gpart, int gnblk) {
    const int b = blockIdx.x;
    TrajState &S = st[b];
    if (threadIdx.x == 0) {
        S.cg_pending = 0;
        if (!S.lin_active) S.ci_active[0] = 0;
    }
    if (!S.lin_active) return;
    double g = fin_sum1(gpart, gnblk, b, 1, 0);
    if (threadIdx.x != 0) return;
    S.cg_gamma = S.cg_gamma0 = g;
    S.cg_beta = 0.0;
    S.lin_it = 0;
    S.lin_rel = 1.0;
    // (g == 0, a zero right-hand side: the first sweep still runs -- it zeroes x -- and the solve ends at its
    //  first reduction point through the breakdown branch of cg_next)
    S.ci_active[0] = S.lin_active;
    S.ci_it[0] = 0;
    S.ci_gamma[0] = g;
}

// Forward CG, the scalar step of one iteration from the three sums of its reduction point
// (see the header of the CG section).
struct CgNext {
    int active, breakdown, it;
    double alpha, beta, gamma, rel;
};
__device__ __forceinline__ CgNext cg_next(double pq, double qq, double gamma, double gamma0, int it_old, double tol,
                                          int maxit) {
    CgNext n;
    n.it = it_old;
    if (!(pq > 0.0) || !(gamma > 0.0)) {     // round-off level residual: stop here, no step
        n.active = 0; n.breakdown = 1; n.alpha = 0.0; n.beta = 0.0; n.gamma = fmax(gamma, 0.0);
        n.rel = gamma0 > 0.0 ? sqrt(fmax(gamma, 0.0) / gamma0) : 0.0;
        return n;
    }
    n.breakdown = 0;
    n.alpha = gamma / pq;
    double gn = n.alpha * n.alpha * qq - gamma;
    if (gn > 1e-13 * gamma) {
        n.beta = gn / gamma;
    } else {                                  // prediction lost in cancellation: restart the direction
        gn = 1e-13 * gamma;
        n.beta = 0.0;
    }
    n.gamma = gn;
    n.it = it_old + 1;
    n.rel = sqrt(gn / gamma0);
    n.active = (n.rel > tol && n.it < maxit) ? 1 : 0;
    return n;
}

// the three sums, one wavefront each (fixed order); all threads of the workgroup call it (>= 192 threads)
__device__ __forceinline__ void cg_sums(const double *__restrict__ gpart, const double *__restrict__ gpart2, int gnblk,
                                        const double *__restrict__ part, int nblk, int pslot, int direct, int b,
                                        double *s3 /* LDS [3] */) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wv < 3) {
        const double *src = wv == 0 ? gpart : (wv == 1 ? gpart2 : part + pslot);
        const int n = wv == 2 ? nblk : gnblk, stride = wv == 2 ? NPART : 1;
        double a = 0.0;
        if (wv < 2 || direct)
            for (int t = lane; t < n; t += 64) a += src[((long)b * n + t) * stride];
        a = wave_sum(a);
        if (lane == 0) s3[wv] = a;
    }
    __syncthreads();
}

// record the step in the trajectory's state (one thread); wr = copy of the per-iteration state to write
__device__ __forceinline__ void cg_record(TrajState &S, const CgNext &n, int wr) {
    S.cg_alpha = n.alpha;
    S.cg_beta = n.beta;
    S.cg_gamma = n.gamma;
    S.lin_rel = n.rel;
    if (!n.breakdown) {
        S.lin_it = n.it;
        S.lin_total++;
    }
    if (!n.active && n.rel * S.lin_rscale > S.lin_maxrel) S.lin_maxrel = n.rel * S.lin_rscale;
    if (!n.active) S.lin_maxabs = fmax(S.lin_maxabs, n.rel * S.lin_r0);
    if (n.it > S.step_lin_max) S.step_lin_max = n.it;
    if (S.step_solves >= 1 && S.step_solves <= 4 && n.it > S.step_lin[S.step_solves - 1]) S.step_lin[S.step_solves - 1] = n.it;
    S.ci_active[wr] = n.active;
    S.ci_it[wr] = n.it;
    S.ci_gamma[wr] = n.gamma;
}

// The reduction point of the LAST enqueued iteration (the earlier ones are resolved inside the next
// k_schur_p): reads copy rd of the per-iteration state, leaves the step pending for k_cg_finish and
// publishes lin_active for the kernels and the host code outside the CG loop.
__global__ void k_fin_cg_step(TrajState *st, const double *__restrict__ gpart, const double *__restrict__ gpart2,
                              int gnblk, const double *__restrict__ part, int nblk, int pslot, int direct, int pbuf,
                              int rd, double tol, int maxit) {
    const int b = blockIdx.x;
    TrajState &S = st[b];
    __shared__ double s3[3];
    if (!S.lin_active) return;
    if (!S.ci_active[rd]) {                  // converged earlier: its last step is already in x
        if (threadIdx.x == 0) { S.lin_active = 0; S.cg_pending = 0; }
        return;
    }
    cg_sums(gpart, gpart2, gnblk, part, nblk, pslot, direct, b, s3);
    if (threadIdx.x != 0) return;
    const CgNext n = cg_next(s3[0], s3[1], direct ? s3[2] : S.ci_gamma[rd], S.cg_gamma0, S.ci_it[rd], S.lin_reltol, maxit);
    cg_record(S, n, rd ^ 1);
    S.cg_pending = n.breakdown ? 0 : 1;
    S.cg_pbuf = pbuf;
    S.lin_active = n.active;
}

// lin_active <- copy rd of the per-iteration state (before the host looks at the state inside a long solve)
__global__ void k_cg_publish(TrajState *st, int rd, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && st[b].lin_active && !st[b].ci_active[rd]) st[b].lin_active = 0;
}

// alpha = gamma / <p, q>_Z ; src 0: GEMM partials (1 value), src 1: stencil partials (slot k)
__global__ void k_fin_cg_alpha(TrajState *st, const double *__restrict__ part, int n, int stride, int k) {
    const int b = blockIdx.x;
    TrajState &S = st[b];
    if (!S.lin_active) return;
    double pq = fin_sum1(part, n, b, stride, k);
    if (threadIdx.x != 0) return;
    if (!(pq > 0.0)) {                       // breakdown (round-off level residual): stop here
        S.lin_active = 0;
        S.cg_alpha = 0.0;
        if (S.lin_rel > S.lin_maxrel) S.lin_maxrel = S.lin_rel;
        return;
    }
    S.cg_alpha = S.cg_gamma / pq;
}

// beta = gamma_new / gamma; convergence test.  mode 0 (forward): relative decrease of the
// Z-norm of the preconditioned residual; mode 1 (adjoint): ||r||_2 / ||rhs||_2 (slot k+1).
__global__ void k_fin_cg_beta(TrajState *st, const double *__restrict__ part, int nblk, int mode, double tol,
                              int maxit, int init) {
    const int b = blockIdx.x;
    TrajState &S = st[b];
    if (!S.lin_active) return;
    double gn = fin_sum1(part, nblk, b, NPART, 0);
    double rr = mode == 1 ? fin_sum1(part, nblk, b, NPART, 1) : 0.0;
    if (threadIdx.x != 0) return;
    if (init) {
        S.cg_gamma = S.cg_gamma0 = gn;
        S.cg_beta = 0.0;
        S.lin_it = 0;
    } else {
        S.cg_beta = S.cg_gamma > 0.0 ? gn / S.cg_gamma : 0.0;
        S.cg_gamma = gn;
        S.lin_it++;
        S.lin_total++;
    }
    S.lin_rel = mode == 1 ? (S.lin_r0 > 0.0 ? sqrt(rr) / S.lin_r0 : 0.0)
                          : (S.cg_gamma0 > 0.0 ? sqrt(gn / S.cg_gamma0) : 0.0);
    if (!(S.lin_rel > tol) || S.lin_it >= maxit) {
        S.lin_active = 0;
        if (S.lin_rel > S.lin_maxrel) S.lin_maxrel = S.lin_rel;
    }
}

// After k_dmu_ceiling: the step ceiling (F2:383-391) and the start of the Armijo loop.
// strict: a solve that the enqueued sweeps did not finish is left as it is (lin_active stays set, no trial is armed), so the
// next solve slot of the schedule -- or the host's continuation loop -- runs it again with the budget it needs.
// fin_copy >= 0 (after k_dmu_ceiling_fin): the solve's last reduction point was resolved by that kernel into copy
// fin_copy of the per-sweep state, from which lin_active is taken over here.
// cheb.enq >= 0 (after a Chebyshev solve with cheb.enq sweeps enqueued): a trajectory whose plan asked for more sweeps has
// not finished (it is left as it is, like an unfinished CG solve under `strict`); for the others the solve's books are
// closed here: sweeps done, and the relative residual the solve left, estimated as the last directly summed
// ||z_n||_Z / ||z_0||_Z (partials of k_cheb_rows) times the factor T_n / T_{n+1} the bound promises for the last step.
struct ChebFin {
    int enq, gnblk;
    const double *g0, *gn;                  // [B][gnblk] partials of <z_0,z_0>_Z and of <z_n,z_n>_Z
    const double *cmin;                     // [B][gnblk] step-ceiling ratios taken by the solve's last row kernel
};
__global__ void k_fin_ceiling(TrajState *st, const double *__restrict__ part, int nblk, int strict, int fin_copy, ChebFin cheb) {
    const int b = blockIdx.x;
#if FIN_LDS
    __shared__ TrajState S;                 // the record in LDS, copied in and out by the whole wavefront (see k_fin_residual)
    traj_copy_in(S, st + b);
#else
    TrajState S = st[b];
#endif
    if (!S.newton_active || S.need_trial) return;
    if ((cheb.enq >= 0) != (S.use_cheb != 0)) return;      // the other form's launch sequence looks after this trajectory
    int lin_active = S.lin_active;          // (a value every lane keeps: the record itself is advanced by lane 0 only)
    if (cheb.enq >= 0 && lin_active) {
        if (S.cheb_n > cheb.enq) return;    // unfinished: taken up again by the next solve slot / the host's loop
        double rel = 1.0;
        if (S.cheb_n >= 1) {
            const double g0 = fin_sum1(cheb.g0, cheb.gnblk, b, 1, 0), gn = fin_sum1(cheb.gn, cheb.gnblk, b, 1, 0);
            rel = g0 > 0.0 ? sqrt(fmax(gn, 0.0) / g0) : 0.0;
        }
        if (threadIdx.x == 0) {
            rel *= cheb_last_factor(S.cheb_theta, S.cheb_delta, S.cheb_n);
            S.lin_rel = rel;
            S.lin_it = S.cheb_n;
            S.lin_total += S.cheb_n;
            if (rel * S.lin_rscale > S.lin_maxrel) S.lin_maxrel = rel * S.lin_rscale;
            S.lin_maxabs = fmax(S.lin_maxabs, rel * S.lin_r0);
            if (S.cheb_n > S.step_lin_max) S.step_lin_max = S.cheb_n;
            if (S.step_solves >= 1 && S.step_solves <= 4) S.step_lin[S.step_solves - 1] = S.cheb_n;
        }
        lin_active = 0;
    }
    if (fin_copy >= 0 && lin_active) {
        lin_active = fin_copy == 0 ? S.ci_active[0] : S.ci_active[1];
        if (strict && lin_active) {         // unfinished: only the flag goes back
            if (threadIdx.x == 0) st[b].lin_active = lin_active;
            return;
        }
    }
    if (strict && lin_active) return;
    double v[NPART];
    const int op[NPART] = {1, 0, 0, 0, 0, 0};
    if (cheb.enq >= 0 && cheb.cmin) {
        double a = 1e300;
        for (int t = threadIdx.x; t < cheb.gnblk; t += 64) a = fmin(a, cheb.cmin[(long)b * cheb.gnblk + t]);
        v[0] = wave_min(a);
    } else {
        fin_reduce(part, nblk, b, v, op, 1);
    }
    if (threadIdx.x == 0) {
        double amax = 2.0;
        if (v[0] < 1e299) amax = fmin(amax, 0.9 * v[0]);
        if (!isfinite(amax) || amax <= 0.0) amax = 1.0;
        S.alpha = fmin(1.0, amax);
        if (lin_active && S.lin_rel > S.lin_maxrel) S.lin_maxrel = S.lin_rel;     // the host's sweep budget ran out: go on
        S.lin_active = 0;                                                         // with an inexact step
        S.need_trial = 1;
        S.trial_no = 0;
        S.force_accept = 0;
        S.best_norm = 1e300;
        S.best_alpha = S.alpha;
    }
#if FIN_LDS
    traj_copy_out(st + b, S);
#else
    if (threadIdx.x == 0) st[b] = S;
#endif
}

__global__ void k_fin_mass(TrajState *st, const double *__restrict__ part, int nblk, int init) {
    const int b = blockIdx.x;
    TrajState &S = st[b];
    if (S.frozen) return;
    double v[NPART];
    const int op[NPART] = {0, 0, 0, 0, 0, 0};
    fin_reduce(part, nblk, b, v, op, 2);
    if (threadIdx.x != 0) return;
    if (init) {
        S.mass0 = v[0];
        S.mass_err = 0.0;
    } else {
        S.mass_err = v[0] - S.mass0;
    }
    S.Wint = v[1];
}

// Linear-solve setup outside the Newton loop.
//   mode 0: partial slots {min D, max D, sum rhs^2}, adjoint operator A(phi_n)  (B2:198)
//   mode 1: partial slot  {sum rhs^2}, constant-coefficient operator (D = 0)     (B2:184)
//   mode 2: partial slots {sum(.), sum rhs^2, min D, max D}, forward Schur operator
__global__ void k_fin_lin_begin(TrajState *st, const double *__restrict__ part, int nblk, int mode,
                                double tau, double kappa, double dt, double lin_tol) {
    const int b = blockIdx.x;
    TrajState &S = st[b];
    if (threadIdx.x == 0 && S.lin_active) {      // the previous solve of this trajectory ended on its sweep budget
        S.lin_unconv++;
        if (S.lin_rel > S.lin_maxrel) S.lin_maxrel = S.lin_rel;
    }
    double v[NPART];
    const int op0[NPART] = {1, 2, 0, 0, 0, 0};
    const int op1[NPART] = {0, 0, 0, 0, 0, 0};
    const int op2[NPART] = {0, 0, 1, 2, 0, 0};
    if (mode == 0) fin_reduce(part, nblk, b, v, op0, 3);
    else if (mode == 1) fin_reduce(part, nblk, b, v, op1, 1);
    else fin_reduce(part, nblk, b, v, op2, 4);
    if (threadIdx.x != 0) return;
    double dmin = 0.0, dmax = 0.0, r0;
    if (mode == 0) { dmin = v[0]; dmax = v[1]; r0 = sqrt(v[2]); }
    else if (mode == 1) { r0 = sqrt(v[0]); }
    else { dmin = v[2]; dmax = v[3]; r0 = sqrt(v[1]); }
    S.Dmin = dmin;
    S.Dmax = dmax;
    if (mode == 2) cg_setup(S, 2.0 * sqrt(0.5 * kappa / dt), 1.0, lin_tol);
    else cg_setup(S, tau + 2.0 * sqrt(0.5 * dt), 0.5 * dt, lin_tol);
    S.lin_r0 = r0;
    S.lin_active = (mode == 2 || r0 > 0.0) ? 1 : 0;
    S.lin_it = 0;
    S.lin_prev = 1e300;
    S.lin_rel = 1.0;
    S.lin_reltol = lin_tol;
    S.use_cheb = 0;
    S.scaled = 0;
    S.nsolves++;
}

// ---------------------------------------------------------------------------------
// Set-up kernels of the stand-alone linear-solve entry points (kernel-level parity tests).
// ---------------------------------------------------------------------------------
// J [dphi;dmu] = [a;b]  ->  Schur form: R_phi := -a, rhs := b - L a, D from phi (slot 0).
__global__ __launch_bounds__(NTH) void k_solve_setup(Geom G, Phys P, const double *__restrict__ a,
                                                     const double *__restrict__ bv, const double *__restrict__ phi,
                                                     double dt, double *__restrict__ Rphi, double *__restrict__ rhs,
                                                     double *__restrict__ D, double *__restrict__ part) {
    TILE_COORDS;
    __shared__ double s[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W = TX + 2;
    const long pb = b * G.plane;
    load_tile<1>(s, a + pb, G, c0, r0);
    __syncthreads();
    double acc[4] = {0.0, 0.0, 1e300, -1e300};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p = (ly + 1) * W + lx + 1;
            long o = pb + (long)r * G.pitch + c;
            double rh = bv[o] - lap_at<W>(s, p, G.ax, G.ay);
            double d = jac_diag(phi[o], P.tau / dt, P.c1);
            Rphi[o] = -s[p];
            rhs[o] = rh;
            D[o] = d;
            acc[1] += rh * rh;
            acc[2] = fmin(acc[2], d);
            acc[3] = fmax(acc[3], d);
        }
    }
    const int op[4] = {0, 0, 1, 2};
    block_reduce_store<4>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

// D = f''(phi) (or 0 when phi == NULL) and sum rhs^2, min/max D: slots {min D, max D, sum rhs^2}
__global__ __launch_bounds__(NTH) void k_adj_setup(Geom G, Phys P, const double *__restrict__ phi,
                                                   const double *__restrict__ rhs, double *__restrict__ Dn,
                                                   double *__restrict__ part) {
    TILE_COORDS;
    __shared__ double sred[NPART * 4];
    double acc[3] = {1e300, -1e300, 0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = b * G.plane + (long)r * G.pitch + c;
            double d = phi ? fpp_log(phi[o], P.c1, P.c2) : 0.0;
            Dn[o] = d;
            acc[0] = fmin(acc[0], d);
            acc[1] = fmax(acc[1], d);
            if (rhs) acc[2] += rhs[o] * rhs[o];
        }
    }
    const int op[3] = {1, 2, 0};
    block_reduce_store<3>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

__global__ __launch_bounds__(NTH) void k_solve_w(Geom G, const double *__restrict__ w, const double *__restrict__ un,
                                                 const double *__restrict__ unp1, double gdt, double *__restrict__ out) {
    TILE_COORDS;
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = b * G.plane + (long)r * G.pitch + c;
            double u0 = un ? un[o] : 0.0, u1 = unp1 ? unp1[o] : 0.0;
            out[o] = ((gdt - 0.5) * w[o] + 0.5 * (u1 + u0)) / (gdt + 0.5);
        }
    }
}

// Adjoint sweep, starting guess of the solve for p_n (the solve starts from the content of x): x holds p_{n+1}; it is
// saved into the ring slot `keep` and replaced by c_0 p_{n+1} + sum_{j>=1} c_j p_{n+1+j} over levels the ring holds.
// Crank-Nicolson does not damp the highest modes (amplification -> -1), so p carries a component that alternates from
// level to level and dominates the residual of a guess: p_{n+1} itself leaves ||rhs - A x0|| = 2 ||rhs||, p_{n+2} leaves
// 1e-3 and the extrapolation over levels of the SAME parity, n+2, n+4, ..., 5e-6 and below (scripts/r2_extrap_adj.py).
// backward_pass therefore sets c_0 = 0 and weights on j = 1, 3, 5, 7.
__global__ __launch_bounds__(NTH) void k_adj_guess(Geom G, double *__restrict__ x, GuessArgs ga, double *__restrict__ keep) {
    TILE_COORDS;
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            const long o = b * G.plane + (long)r * G.pitch + c;
            const double p1 = x[o];
            const double *gcf = ga.row(b);
            double v = gcf[0] * p1;
#pragma unroll
            for (int j = 1; j < GUESS_ORD; ++j)
                if (gcf[j] != 0.0) v += gcf[j] * ga.d[j][o];
            keep[o] = p1;
            x[o] = isfinite(v) ? v : p1;
        }
    }
}

// the longest solve since the host's last look (adjoint launch schedule)
__global__ void k_reset_longest(TrajState *st, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) st[b].step_lin_max = 0;
}

// zero-fill / constant fill of valid nodes
__global__ __launch_bounds__(NTH) void k_fill(Geom G, double *__restrict__ x, double v) {
    TILE_COORDS;
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) x[b * G.plane + (long)r * G.pitch + c] = v;
    }
}

// phi_Q[lvl] = (1 - t_lvl/T) phi_0 + (t_lvl/T) phi_T  (build_targets choice_q = 1, G2:221-222)
// grid = (tiles, levels, B)
__global__ __launch_bounds__(NTH) void k_ramp(Geom G, int tiles_f, const double *__restrict__ phi0,
                                              const double *__restrict__ phiT, const double *__restrict__ tfrac,
                                              long hist_stride, double *__restrict__ pq) {
    const int b = blockIdx.z, lvl = blockIdx.y;
    const int c0 = (blockIdx.x % tiles_f) * TX, r0 = (blockIdx.x / tiles_f) * TY;
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
    const double tp = tfrac[lvl];
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = (long)r * G.pitch + c;
            pq[b * hist_stride + (long)lvl * G.plane + o] = (1 - tp) * phi0[b * G.plane + o] + tp * phiT[b * G.plane + o];
        }
    }
}


// =================================================================================
// Conjugate gradients on the preconditioned Schur / adjoint systems.
//
// Forward (left preconditioning):  T = P^-1 A,  A = I/dt + M (kappa/2 M + D) = P + M Delta,
//   Delta = D - dbar > 0.  T is self-adjoint and positive in <x, y>_Z = sum W Delta x y, so
//   plain CG applies with z = P^-1 (rhs - A x) as residual.  One reduction point per iteration:
//       [k_schur_p]   x += alpha p; z' = z - alpha q; p' = z' + beta p; v = A p'; gamma = <z',z'>_Z
//       [3 DCT passes] q' = P^-1 v with <p',q'>_Z and <q',q'>_Z
//       [start of the next k_schur_p] alpha' = gamma / <p',q'>;  gamma' ~ alpha'^2 <q',q'> - gamma  (the value
//                     <z'',z''>_Z will take, by conjugacy);  beta' = gamma'/gamma;  stop test on gamma'
//   gamma itself is always the directly summed <z',z'>_Z of the previous line; the predicted
//   gamma' only steers beta and the stop test, and a prediction lost in cancellation
//   (gamma' < 1e-13 gamma) restarts the direction (beta = 0) instead.
// Adjoint (right preconditioning): A P^-1 is self-adjoint in <x, y>_Z' = sum W x y / Delta:
//       pv = P^-1 ph; q = A pv; alpha = <r,r>_Z' / <ph,q>_Z'; x += alpha pv; r -= alpha q;
//       beta = <r',r'>_Z' / <r,r>_Z'; ph = r' + beta ph.
// =================================================================================

// Iteration k of the forward CG (it = k; FIRST: k = 0, p_new = z, v = A p_new).  For k >= 1 every
// workgroup first resolves the reduction point of iteration k-1 from its partial sums (gpart/gpart2 from
// the preconditioner epilogue, part[., (k-1)&1] from the previous launch of this kernel) -- the same
// arithmetic on the same numbers in every workgroup; the workgroup with blk == 0 records the result --
// and then applies that step on the way:
//   x += alpha p_old; z_new = z - alpha q; p_new = z_new + beta p_old (z_new and p_new recomputed on
//   the halo, z_new written to its own buffer so that neighbours still read z), v = A p_new, and the
//   partial of <z_new, z_new>_Z goes to part[., k&1].  A trajectory that the step brings below the
//   tolerance only takes the step (x += alpha p_old).
template <int FIRST>
__global__ __launch_bounds__(NTH) void k_schur_p(Geom G, Phys P, TrajState *__restrict__ st,
                                                 long slot_stride, const double *__restrict__ z,
                                                 const double *__restrict__ q, const double *__restrict__ p_old,
                                                 const double *__restrict__ D_s, double dt,
                                                 double *__restrict__ x, double *__restrict__ z_new,
                                                 double *__restrict__ p_new, double *__restrict__ v,
                                                 double *__restrict__ part, const double *__restrict__ gpart,
                                                 const double *__restrict__ gpart2, int gnblk, int it, double tol,
                                                 int maxit) {
    TILE_COORDS;
    __shared__ double sx[(TY + 4) * (TX + 4)];
    __shared__ double stt[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W2 = TX + 4, W1 = TX + 2;
    const long pb = b * G.plane;
    const int rd = (it + 1) & 1, wr = it & 1;          // per-iteration state copies / partial slots
    const int slot = st[b].slot;
    const double dbar = st[b].dbar;
    double alpha = 0.0, beta = 0.0;
    if (FIRST) {
        if (!st[b].ci_active[0]) return;
        load_tile<2>(sx, z + pb, G, c0, r0);
    } else {
        const int was_active = st[b].ci_active[rd];
        if (!was_active) {
            if (blk == 0 && threadIdx.x == 0) {           // hand the (finished) state on to the other copy
                st[b].ci_active[wr] = 0;
                st[b].ci_it[wr] = st[b].ci_it[rd];
                st[b].ci_gamma[wr] = st[b].ci_gamma[rd];
            }
            return;
        }
        const double gamma0 = st[b].cg_gamma0, gamma_old = st[b].ci_gamma[rd];
        const int it_old = st[b].ci_it[rd];
        cg_sums(gpart, gpart2, gnblk, part, nblk, rd, it >= 2, b, sred);
        const CgNext n = cg_next(sred[0], sred[1], it >= 2 ? sred[2] : gamma_old, gamma0, it_old, st[b].lin_reltol, maxit);
        __syncthreads();                                   // sred is reused below
        if (blk == 0 && threadIdx.x == 0) cg_record(st[b], n, wr);
        alpha = n.alpha;
        beta = n.beta;
        if (!n.active) {                                   // converged (or broke down, alpha = 0): take the step only
            if (!n.breakdown)
                for (int k = 0; k < TY / 4; ++k) {
                    int r = r0 + ly0 + 4 * k, c = c0 + lx;
                    if (r < G.ns && c < G.nf) {
                        long o = pb + (long)r * G.pitch + c;
                        x[o] += alpha * p_old[o];
                    }
                }
            return;
        }
        for (int e = threadIdx.x; e < W2 * (TY + 4); e += NTH) {
            int ly = e / W2, lxx = e - ly * W2;
            int gr = refl(r0 - 2 + ly, G.ns), gc = refl(c0 - 2 + lxx, G.nf);
            long o = pb + (long)gr * G.pitch + gc;
            sx[e] = (z[o] - alpha * q[o]) + beta * p_old[o];
        }
    }
    __syncthreads();
    const double *Dp = D_s + slot * slot_stride + pb;
    for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
        int ly = e / W1, lxx = e - ly * W1;
        int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
        int p2 = (ly + 1) * W2 + lxx + 1;
        stt[e] = -0.5 * P.kappa * lap_at<W2>(sx, p2, G.ax, G.ay) + Dp[(long)gr * G.pitch + gc] * sx[p2];
    }
    __syncthreads();
    const double idt = 1.0 / dt;
    double acc[1] = {0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
            long o = pb + (long)r * G.pitch + c;
            if (!FIRST) {
                double po = p_old[o];
                double zn = z[o] - alpha * q[o];
                x[o] += alpha * po;
                z_new[o] = zn;
                acc[0] += wdev(r, c, G) * (Dp[(long)r * G.pitch + c] - dbar) * (zn * zn);
            } else {
                x[o] = 0.0;          // start of a solve (only for the trajectories that take part in it)
            }
            p_new[o] = sx[p2];
            v[o] = sx[p2] * idt - lap_at<W1>(stt, p1, G.ax, G.ay);
        }
    }
    if (!FIRST) {
        const int op[1] = {0};
        block_reduce_store<1>(acc, op, sred, part + ((long)b * nblk + blk) * NPART + wr);
    }
}

// End of a forward CG solve: add the step that is still pending, x += alpha p[cg_pbuf].
__global__ __launch_bounds__(NTH) void k_cg_finish(Geom G, const TrajState *__restrict__ st,
                                                   const double *__restrict__ p0, const double *__restrict__ p1,
                                                   double *__restrict__ x) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (!S.cg_pending) return;
    const double *p = S.cg_pbuf ? p1 : p0;
    for (int k = 0; k < TY / 4; ++k) {
        int r = r0 + ly0 + 4 * k, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            long o = b * G.plane + (long)r * G.pitch + c;
            x[o] += S.cg_alpha * p[o];
        }
    }
}

// ph = r + beta ph (FIRST: ph = r)
template <int FIRST>
__global__ __launch_bounds__(NTH) void k_cg_dir(Geom G, const TrajState *__restrict__ st,
                                                const double *__restrict__ r, double *__restrict__ ph) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (!S.lin_active) return;
    for (int k = 0; k < TY / 4; ++k) {
        int rr = r0 + ly0 + 4 * k, c = c0 + lx;
        if (rr < G.ns && c < G.nf) {
            long o = b * G.plane + (long)rr * G.pitch + c;
            ph[o] = FIRST ? r[o] : r[o] + S.cg_beta * ph[o];
        }
    }
}

// q = A(phi_n) pv with partial sum W ph q / (D_n - dbar)
__global__ __launch_bounds__(NTH) void k_adj_q(Geom G, Phys P, const TrajState *__restrict__ st,
                                               const double *__restrict__ pv, const double *__restrict__ Dn,
                                               const double *__restrict__ ph, double dt, double *__restrict__ q,
                                               double *__restrict__ part) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (!S.lin_active) return;
    __shared__ double sx[(TY + 4) * (TX + 4)];
    __shared__ double stt[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    constexpr int W2 = TX + 4, W1 = TX + 2;
    const long pb = b * G.plane;
    load_tile<2>(sx, pv + pb, G, c0, r0);
    __syncthreads();
    for (int e = threadIdx.x; e < (TY + 2) * W1; e += NTH) {
        int ly = e / W1, lxx = e - ly * W1;
        stt[e] = -lap_at<W2>(sx, (ly + 1) * W2 + lxx + 1, G.ax, G.ay);
    }
    __syncthreads();
    double acc[1] = {0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p1 = (ly + 1) * W1 + lx + 1, p2 = (ly + 2) * W2 + lx + 2;
            long o = pb + (long)r * G.pitch + c;
            double d = Dn[o];
            double qv = sx[p2] + (P.tau + 0.5 * dt * d) * stt[p1] - 0.5 * dt * lap_at<W1>(stt, p1, G.ax, G.ay);
            q[o] = qv;
            acc[0] += wdev(r, c, G) / (d - S.dbar) * (ph[o] * qv);
        }
    }
    const int op[1] = {0};
    block_reduce_store<1>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

// x += alpha pv; r -= alpha q; partials: sum W r^2/(D_n - dbar), sum r^2
__global__ __launch_bounds__(NTH) void k_cg_update_adj(Geom G, const TrajState *__restrict__ st,
                                                       const double *__restrict__ pv, const double *__restrict__ q,
                                                       const double *__restrict__ Dn, double *__restrict__ x,
                                                       double *__restrict__ r, double *__restrict__ part) {
    TILE_COORDS;
    const TrajState S = st[b];
    if (!S.lin_active) return;
    __shared__ double sred[NPART * 4];
    double acc[2] = {0.0, 0.0};
    for (int k = 0; k < TY / 4; ++k) {
        int rr = r0 + ly0 + 4 * k, c = c0 + lx;
        if (rr < G.ns && c < G.nf) {
            long o = b * G.plane + (long)rr * G.pitch + c;
            x[o] += S.cg_alpha * pv[o];
            double rn = r[o] - S.cg_alpha * q[o];
            r[o] = rn;
            acc[0] += wdev(rr, c, G) / (Dn[o] - S.dbar) * (rn * rn);
            acc[1] += rn * rn;
        }
    }
    const int op[2] = {0, 0};
    block_reduce_store<2>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}
