/* vch.h — C ABI of the MI355X-native viscous Cahn–Hilliard optimal-control engine.
 *
 * This is the drop-in boundary for the reference's hot path.  The
 * reference (a pure-Python research code) has no FFI layer: its seams are Python
 * function calls between sibling modules.  Each entry point below replaces one of
 * those functions; the file:line of the replaced interface is cited per function
 * (paths relative to the reference root):
 *   F2 = src/2D/Vch_control_2D/Forward2_solver.py    B2 = src/2D/Vch_control_2D/backward2_solver.py
 *   C2 = src/2D/Vch_control_2D/cost2_and_function.py G2 = src/2D/Vch_control_2D/GD2_configured.py
 *   K2 = src/2D/Vch_control_2D/config.py
 *   F1 = src/1D/Vch_control_1D/Forward_solver.py     B1 = src/1D/Vch_control_1D/backward_solver.py
 *   C1 = src/1D/Vch_control_1D/cost_and_function.py  G1 = src/1D/Vch_control_1D/GD_1D.py
 *
 * Conventions
 *  - plain C, no torch types; every array argument is a HOST pointer to caller-owned,
 *    C-contiguous float64 memory unless the name ends in `_dev`.  The engine never
 *    frees or keeps caller memory.
 *  - a context owns one GPU's device buffers, stream and batch of B independent
 *    trajectories; all calls on it are synchronous and must come from one host thread.
 *  - 2D fields are (Nx+1, Ny+1) row-major, y fastest (the reference's layout); batched
 *    arguments are [B][...]; histories are [B][rows][Nx+1][Ny+1].
 *  - return value: 0 on success, negative on error; vch_last_error() describes the last
 *    failure of the calling thread.  Newton non-convergence is NOT an error (F2:427).
 */
#ifndef VCH_H
#define VCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCH_OK 0
#define VCH_ERR_ARG (-1)      /* bad argument / shape (the reference raises ValueError / AssertionError) */
#define VCH_ERR_HIP (-2)      /* HIP runtime failure */
#define VCH_ERR_STATE (-3)    /* call order (e.g. backward before forward with resident history) */
#define VCH_ERR_NOMEM (-4)

const char *vch_last_error(void);
/* Library build/ABI version, and number of visible HIP devices (<0 on error). */
int vch_abi_version(void);
int vch_device_count(void);

/* ------------------------------------------------------------------ 2D ---- */

/* Physical parameters = the fields of ForwardSolverConfig (K2:103-113). */
typedef struct vch2d_params {
    int32_t Nx, Ny;
    double Lx, Ly;
    double tau, gamma, c1, c2, kappa;
} vch2d_params;

/* Weights/bounds = the fields of OptimizationConfig (K2:137-144). */
typedef struct vch_opt_params {
    double b1, b2, b3, kappa_sparsity;
    double alpha_max;
    int32_t max_iter;
    double u_min, u_max;
} vch_opt_params;

/* Solver statistics of one forward march / backward sweep (per trajectory sums). */
typedef struct vch_stats {
    int64_t newton_iters;      /* residual evaluations that entered the Newton loop (len(hist)) */
    int64_t linear_solves;     /* Newton linear solves (spsolve calls in the reference, F2:370) */
    int64_t linear_iters;      /* preconditioned CG iterations spent in them */
    int64_t armijo_trials;     /* residual evaluations in the Armijo loop (F2:398-419) */
    double  max_lin_relres;    /* worst final relative residual of a linear solve */
    double  seconds;           /* device time of the call (HIP events) */
    /* ABI version 2 */
    double  max_lin_absres;    /* worst (final relative residual x ||rhs||_2) of a Newton linear solve: the estimate of
                                  the Schur residual it leaves, which the inexact-Newton rule of a march keeps at
                                  5 % of the Newton tolerance */
    int64_t host_syncs;        /* blocking looks of the host at the device state during the call */
    int64_t launches;          /* kernel launches of the call */
    /* ABI version 3 (a caller must check vch_abi_version() >= 3 before passing a vch_stats: the library writes all fields) */
    int64_t unconverged_solves; /* adjoint solves that the speculative schedule's sweeps did not finish (each ended below
                                  10 x the solve tolerance, or the whole sweep was repeated with the rigorous budget) */
} vch_stats;

typedef struct vch2d_ctx vch2d_ctx;

/* Create a context for `batch` trajectories and marches of at most `max_steps` steps on
 * HIP device `device`.  History buffers are allocated lazily by the calls that need them. */
vch2d_ctx *vch2d_create(const vch2d_params *p, int batch, int max_steps, int device);
void vch2d_destroy(vch2d_ctx *ctx);
int vch2d_batch(const vch2d_ctx *ctx);
/* 1 when the DCT preconditioner runs as in-LDS FFTs (both Nx, Ny powers of two in 16..2048),
 * 0 when it runs as MFMA f64 matrix products (any grid size). */
int vch2d_uses_fft(const vch2d_ctx *ctx);

/* -- discrete operators (kernel-level parity tests; each replaces one reference helper) -- */

/* out = L v, mirrored-Neumann 5-point Laplacian incl. the reference's Kronecker-order
 * quirk for Nx != Ny.  Replaces apply_laplacian (F2:140-152).  v,out: [B][Nx+1][Ny+1]. */
int vch2d_apply_laplacian(vch2d_ctx *ctx, const double *v, double *out);
/* mu = -kappa L phi + c1 reglog(phi) - 2 c2 phi - w.  Replaces initialize_mu (F2:155-167). */
int vch2d_initialize_mu(vch2d_ctx *ctx, const double *phi, const double *w, double *mu_out);
/* Crank–Nicolson filter step.  Replaces solve_w (F2:170-181).  u_n/u_np1 may be NULL (zeros). */
int vch2d_solve_w(vch2d_ctx *ctx, const double *w_old, double dt, const double *u_n,
                  const double *u_np1, double *w_out);
/* [R_phi; R_mu].  Replaces solve_phi_residual + solve_mu_residual (F2:184-221). */
int vch2d_residuals(vch2d_ctx *ctx, const double *phi_new, const double *phi_old,
                    const double *mu_new, const double *mu_old, const double *w_new,
                    const double *w_old, double dt, double *Rphi_out, double *Rmu_out,
                    double *norm_out /* [B], may be NULL */);
/* (out_phi, out_mu) = J(phi_new) [dphi; dmu], matrix-free.  Replaces
 * assemble_jacobian(...) @ v (F2:224-253). */
int vch2d_jacobian_apply(vch2d_ctx *ctx, const double *phi_new, double dt, const double *dphi,
                         const double *dmu, double *out_phi, double *out_mu);
/* Solve J(phi_new) [dphi; dmu] = [rhs_phi; rhs_mu].  Replaces spsolve(J.tocsc(), -R)
 * (F2:370): Schur reduction to the scalar 13-point system + DCT-I preconditioned
 * conjugate gradients in the weighted inner product.  stats may be NULL. */
int vch2d_jacobian_solve(vch2d_ctx *ctx, const double *phi_new, double dt, const double *rhs_phi,
                         const double *rhs_mu, double *dphi, double *dmu, vch_stats *stats);
/* out = (I/dt + M(kappa/2 M + D)) x, the Schur-reduced Newton operator (M = -L). */
int vch2d_schur_apply(vch2d_ctx *ctx, const double *phi_new, double dt, const double *x,
                      double *out);
/* out = A(phi) v (which=0) or B(phi) v (which=1).  Replaces A_adjoint/B_adjoint @ v
 * (B2:195-203). */
int vch2d_adjoint_apply(vch2d_ctx *ctx, int which, const double *phi, double dt, const double *v,
                        double *out);
/* Solve A(phi_n) p = rhs.  Replaces spsolve(A, rhs) (B2:229); dt = 0 gives the terminal
 * solve (I - tau L) p = rhs (B2:184-185). */
int vch2d_adjoint_solve(vch2d_ctx *ctx, const double *phi_n, double dt, const double *rhs,
                        double *p_out, vch_stats *stats);
/* out = (c0 + m (c1 + c2 m))^{-1} v by fast diagonalisation (DCT-I), m = eigenvalues of
 * -L: the preconditioner itself, exposed for tests. */
int vch2d_spectral_solve(vch2d_ctx *ctx, double c0, double c1, double c2, const double *v,
                         double *out);

/* One implicit time level.  Replaces newton_raphson (F2:323-427): same initial guess,
 * stop rule (||R||_2 < 1e-6, <= 500 its), step ceiling and Armijo rule.
 * hist: [B][hist_cap] residual norms (return_residual_history), n_hist: [B]; may be NULL. */
int vch2d_newton_raphson(vch2d_ctx *ctx, const double *phi_old, const double *mu_old,
                         const double *w_old, const double *w_new, double dt, double *phi_new,
                         double *mu_new, double *hist, int hist_cap, int32_t *n_hist,
                         vch_stats *stats);

/* -- the time march and the adjoint sweep -- */

/* Forward march.  Replaces run_main_simulation (F2:489-596) for B trajectories at once.
 *   phi0   [B][Nx+1][Ny+1]  initial states (the reference hard-codes init_phi_random(seed=42), F2:517)
 *   u      [B][u_rows][Nx+1][Ny+1] control or NULL; rows (step, step+1) are used while
 *          step < u_rows-1, zeros afterwards (F2:545-548).  u == VCH_RESIDENT uses the
 *          control currently resident in the context (PGD state).
 *   dt     [M] step sizes, as produced by the reference's accumulated-time rule (F2:542-543)
 *   phi_hist_out [B][M+1][Nx+1][Ny+1] or NULL (history stays resident on the device)
 * The history (and mu, w at the final level) stays resident for vch2d_backward/vch2d_cost. */
int vch2d_forward(vch2d_ctx *ctx, const double *phi0, const double *u, int u_rows,
                  const double *dt, int M, double *phi_hist_out, vch_stats *stats);

/* Adjoint sweep.  Replaces run_backward (B2:75-246).
 *   phi_hist [B][M+1][..] or NULL = use the resident history of the last forward call
 *   t_hist [M+1]; phi_Q [B][M+1][..] or NULL (zeros, B2:167); phi_T [B][..] or NULL
 *   hx, hy = x[1]-x[0], y[1]-y[0] as the reference takes them (B2:154-155)
 *   p_out, q_out, r_out [B][M+1][..], each may be NULL (r stays resident for the prox). */
int vch2d_backward(vch2d_ctx *ctx, const double *phi_hist, int M, const double *t_hist, double hx,
                   double hy, double b1, double b2, const double *phi_Q, const double *phi_T,
                   double *p_out, double *q_out, double *r_out, vch_stats *stats);

/* J1..J4 and their sum.  Replaces calculate_cost (C2:19-120).  NULL array arguments use the
 * resident state (history of the last forward, resident control and targets).
 *   x [Nx+1], y [Ny+1], t_hist [M+1]; J_out [B][5] = {J1, J2, J3, J4, J}. */
int vch2d_cost(vch2d_ctx *ctx, const double *phi_hist, const double *u, const double *phi_Q,
               const double *phi_T, int M, const double *x, const double *y, const double *t_hist,
               const vch_opt_params *opt, double *J_out);

/* u_out = clip(soft_threshold(u - alpha (r + b3 u), alpha kappa_s), u_min, u_max).
 * Replaces calculate_gradient + proximal_step (C2:123-200).  rows = leading dimension of the
 * histories; alpha [B]. */
int vch2d_grad_prox(vch2d_ctx *ctx, const double *u, const double *r, int rows, const double *alpha,
                    const vch_opt_params *opt, double *u_out);

/* -- device-resident proximal-gradient loop (G2:291-382) -- */

/* Load the optimisation problem: initial states, targets, time grid; u^0 = 0; runs the
 * uncontrolled forward march and evaluates J(u^0) (G2:255-292).
 *   phi_Q == NULL and ramp != 0: phi_Q = (1 - t/T) phi0 + (t/T) phi_T evaluated on the
 *   fly with t/T from the config T (build_targets choice_q = 1, G2:221-222). */
int vch2d_pgd_init(vch2d_ctx *ctx, const double *phi0, const double *phi_T, const double *phi_Q,
                   int ramp, double T, const double *t_hist, int M, const double *x, const double *y,
                   const vch_opt_params *opt, double *J0_out /* [B][5] */);
/* Run n_iters PGD iterations for every trajectory of the batch (optimistic step,
 * backtracking from 0.8 alpha_prev with beta 0.8 and <= 10 trials, alpha growth / plateau
 * rule, stop rule; G2:295-382).  Outputs [B][n_iters] unless noted; any may be NULL.
 *   cost_out      accepted cost after each iteration
 *   alpha_out     step used
 *   attempts_out  backtracking forwards (0 = optimistic step accepted)
 *   change_out    relative control change
 *   seconds_out   [5] time buckets {backward, grad+prox, optimistic forward, cost, backtracking}
 * Returns the number of iterations performed (>= 0) or a negative error. */
int vch2d_pgd_iterate(vch2d_ctx *ctx, int n_iters, double *cost_out, double *alpha_out,
                      int32_t *attempts_out, double *change_out, double *seconds_out);
/* Error metrics the driver appends after every iteration (G2:336-363), for the iterations of the LAST
 * vch2d_pgd_iterate call ([B][n_iters], n_iters as passed there; NaN where no iteration ran):
 *   tracking = ||phi - phi_Q||_L2(Q) / ||phi_Q||_L2(Q)  (denominator sqrt(|Omega| T) when ||phi_Q|| < 1e-9 of it)
 *   terminal = ||phi(T) - phi_T||_L2(Omega) / ||phi_T||_L2(Omega)
 * both from the cost kernel's weighted sums (nested trapezoid rule in y, x, t).  Either output may be NULL. */
int vch2d_pgd_errors(vch2d_ctx *ctx, int n_iters, double *tracking_out, double *terminal_out);
/* Copy resident PGD arrays to the host: what = 0 control u, 1 state history, 2 adjoint r,
 * 3 phi_Q.  out [B][M+1][Nx+1][Ny+1]. */
int vch2d_pgd_get(vch2d_ctx *ctx, int what, double *out);
/* Per-trajectory cost scalars {J1,J2,J3,J4,J} of the current iterate on the DEVICE
 * (5*B doubles), for the caller's RCCL all-reduce; returns a device pointer via *ptr_dev. */
int vch2d_pgd_cost_dev(vch2d_ctx *ctx, double **ptr_dev);

/* Kernel launches and blocking looks of the host at the device state since the context was created: out[0], out[1]. */
int vch2d_counters(vch2d_ctx *ctx, int64_t *out /* [2] */);

/* -- multi-GPU: the one collective of the path (no reference counterpart) --
 * Trajectories are independent, so ranks share nothing on the data path; per PGD iteration there is ONE
 * all-reduce (sum) of the cost scalars {J1,J2,J3,J4,J} over all trajectories of all ranks, done by RCCL over
 * xGMI on a 5-double device buffer.  The per-trajectory values stay on the device: each context keeps the last
 * 64 iterations' scalars in HBM, a one-workgroup kernel adds them up, RCCL reduces in place, 5 doubles come back.
 * RCCL is bound with dlopen("librccl.so.1") at the first call; the library has no link-time dependency on it.
 *   vch_comm_unique_id   rank 0 only: ncclGetUniqueId; the caller hands the 128 bytes to the other ranks
 *   vch_comm_create      ncclCommInitRank on `device` (collective over all ranks); NULL on failure
 *   vch_comm_allreduce_cost  sum over the trajectories of the nctx contexts of this rank (all on the
 *                        communicator's device) and over all ranks of the cost scalars of PGD iteration
 *                        `iteration` (0-based count since vch2d_pgd_init; < 0: the current iterate);
 *                        J_sum_out [5] on the host.  Every rank must call it, in the same order. */
#define VCH_COMM_ID_BYTES 128
typedef struct vch_comm vch_comm;
int vch_comm_unique_id(unsigned char *id_out /* [VCH_COMM_ID_BYTES] */);
vch_comm *vch_comm_create(const unsigned char *id, int rank, int world, int device);
void vch_comm_destroy(vch_comm *comm);
int vch_comm_allreduce_cost(vch_comm *comm, vch2d_ctx *const *ctxs, int nctx, long iteration, double *J_sum_out);

#define VCH_RESIDENT ((const double *)(uintptr_t)1)

/* Replaces free_energy (F2:256-319) for every level of a history at once (the mass/energy
 * invariants of the reference's tests, T2f:252-279): phi_hist [B][rows][Nx+1][Ny+1] or
 * VCH_RESIDENT (the state history of the last march), w_hist the coupling field (same shape) or
 * NULL, eps <= 0 -> 1e-8.  The array is taken as the reference takes it: axis 0 with hy, axis 1
 * with hx.  E_out [B][rows]. */
int vch2d_free_energy(vch2d_ctx *ctx, const double *phi_hist, int rows, const double *w_hist,
                      double hx, double hy, double eps, double *E_out);

/* -- in-situ kernel timing (used by bench.py for the roofline figure; no reference counterpart) --
 * Between _begin and _end every launch of the profiled kernel classes is bracketed by a HIP event
 * pair on the engine's stream (at most max_launches pairs).  _end returns, per class, the summed
 * elapsed milliseconds and the number of launches:
 *   0 Newton stencil SpMV (k_schur_p, fused with the CG updates)  1 DCT as MFMA GEMM (non-power-of-two grids)
 *   2 Newton residual  3 adjoint operator  4 CG vector update  5 adjoint right-hand side  6 cost integrands
 *   7 gradient+prox  8 DCT row pass (forward)  9 DCT column pass (forward, multiplier, inverse)
 *   10 DCT row pass (inverse, with the CG dot products)  11 first sweep of a solve (k_schur_p<1>)
 *   12 first pass of a CG sweep (k_cg_rows_fwd: CG vector updates + Delta p + forward row DCT)  13 the same, first sweep
 *   14 an empty kernel launched 256 times by _begin: the cost of an event pair itself
 *   15 starting guess of a step's first Newton solve (k_guess)  16 starting guess of an adjoint solve (k_adj_guess)
 *   17 row kernel of a reduction-free sweep (k_cheb_rows: inverse row DCT + Chebyshev update + forward row DCT)
 *   18 the same, first kernel of a solve (b~ = P^-1 rhs, y_1)
 *   19 start of a time step (k_eval<0>: old-level terms + initial residual + starting guess; or k_residual<0>)
 *   (class 2 is the Armijo trial: k_eval<2> / k_residual2 / k_residual<1>)
 * _spans (after _end) returns every recorded launch: its class and the elapsed milliseconds of its event pair, in launch
 * order, at most cap entries; the return value is the number recorded.  With these the caller separates launches whose
 * trajectories were all gated off (they last as long as the empty kernel of class 14) from live ones. */
#define VCH_PROF_CLASSES 20
int vch2d_prof_begin(vch2d_ctx *ctx, int max_launches);
int vch2d_prof_end(vch2d_ctx *ctx, double *ms_out, int64_t *count_out, int ncls);
int vch2d_prof_spans(vch2d_ctx *ctx, int32_t *cls_out, float *ms_out, int cap);

/* ------------------------------------------------------------------ 1D ---- */

typedef struct vch1d_params {          /* ForwardSolverConfig of the 1D code (K1:93-102) */
    int32_t N;
    double Lx;
    double tau, gamma, c1, c2, kappa;
} vch1d_params;

typedef struct vch1d_ctx vch1d_ctx;

vch1d_ctx *vch1d_create(const vch1d_params *p, int batch, int max_steps, int device);
void vch1d_destroy(vch1d_ctx *ctx);

/* out = L v (F1:64-80).  v,out [B][N+1]. */
int vch1d_apply_laplacian(vch1d_ctx *ctx, const double *v, double *out);
/* Replaces solve_phi_residual/solve_mu_residual (F1:93-109). */
int vch1d_residuals(vch1d_ctx *ctx, const double *phi_new, const double *phi_old,
                    const double *mu_new, const double *mu_old, const double *w_new,
                    const double *w_old, double dt, double *Rphi_out, double *Rmu_out);
/* Solve the 2(N+1) Newton system.  Replaces np.linalg.solve(J, -R) (F1:185): block
 * (2x2) tridiagonal cyclic reduction in LDS, one workgroup per trajectory (N <= 4096). */
int vch1d_jacobian_solve(vch1d_ctx *ctx, const double *phi_new, double dt, const double *rhs_phi,
                         const double *rhs_mu, double *dphi, double *dmu);
/* Solve A(phi_n) p = rhs with the frozen default parameters (B1:29-33, B1:116). */
int vch1d_adjoint_solve(vch1d_ctx *ctx, const double *phi_n, double dt, const double *rhs,
                        double *p_out);
/* Replaces newton_raphson (F1:139-235). */
int vch1d_newton_raphson(vch1d_ctx *ctx, const double *phi_old, const double *mu_old,
                         const double *w_old, const double *w_new, double dt, double *phi_new,
                         double *mu_new, double *hist, int hist_cap, int32_t *n_hist);
/* Replaces run_main_simulation (F1:286-386): history has M+2 rows (duplicated t=0).
 *   u [B][u_rows][N+1] or NULL; hold-last rule of F1:347-353 (u_rows < M errors like the
 *   reference's IndexError).  phi_hist_out [B][M+2][N+1] or NULL. */
int vch1d_forward(vch1d_ctx *ctx, const double *phi0, const double *u, int u_rows, const double *dt,
                  int M, double *phi_hist_out, vch_stats *stats);
/* Replaces run_backward (B1:48-126); rows = M+2, t_hist [rows]. */
int vch1d_backward(vch1d_ctx *ctx, const double *phi_hist, int rows, const double *t_hist, double h,
                   double b1, double b2, const double *phi_Q, const double *phi_T, double *p_out,
                   double *q_out, double *r_out);
/* Replaces calculate_cost (C1:26-84); J_out [B][5]. */
int vch1d_cost(vch1d_ctx *ctx, const double *phi_hist, const double *u, const double *phi_Q,
               const double *phi_T, int rows, const double *x, const double *t_hist,
               const vch_opt_params *opt, double *J_out);
/* Replaces calculate_gradient + perform_gradient_step + perform_proximal_and_projection
 * (C1:86-112, G1:56-71). */
int vch1d_grad_prox(vch1d_ctx *ctx, const double *u, const double *r, int rows, const double *alpha,
                    const vch_opt_params *opt, double *u_out);

/* Replaces free_energy (F1:243-262) for every level of a history; E_out [B][rows]. */
int vch1d_free_energy(vch1d_ctx *ctx, const double *phi_hist, int rows, const double *w_hist, double h,
                      double eps, double *E_out);

/* Device-resident PGD loop of the 1D driver (the __main__ block of GD_1D.py, G1:333-477, with
 * perform_backtracking_line_search G1:73-113): control, state history, adjoint and targets stay in
 * HBM between iterations.  rows = M+2 (duplicated t = 0 row), t_hist [rows], dt [M], x [N+1].
 *   init: uncontrolled march from phi0 (G1:341), u = 0, targets phi_T [B][N+1] and phi_Q
 *   [B][rows][N+1] or NULL = the ramp (1 - t/T) phi_hist[0] + (t/T) phi_T of build_targets_1d
 *   (G1:238-241); J0_out [B][5] = {J1,J2,J3,J4,J} of the start.
 *   iterate: optimistic step with alpha_prev, else backtracking from alpha_prev with beta 0.8 and
 *   <= 5 trials (the first of which repeats the optimistic step and is not recomputed), alpha growth
 *   1.2 / plateau rule (10 x |dJ| < 1e-7 -> 2.0), stop rule (relative control change < 1e-5 after
 *   k > 10; the control is taken, state and cost keep the previous iterate, G1:462-465).
 *   Outputs [B][n_iters], any may be NULL; trials_out as the reference counts them (1 = optimistic
 *   step accepted); seconds_out [3] = {adjoint sweep, optimistic round, backtracking rounds}.
 *   Returns the number of iterations performed or a negative error.
 *   get: what = 0 control u, 1 state history, 2 adjoint r, 3 phi_Q; out [B][rows][N+1]. */
int vch1d_pgd_init(vch1d_ctx *ctx, const double *phi0, const double *phi_T, const double *phi_Q,
                   const double *x, const double *t_hist, int rows, const double *dt,
                   const vch_opt_params *opt, double *J0_out);
int vch1d_pgd_iterate(vch1d_ctx *ctx, int n_iters, double *cost_out, double *alpha_out,
                      int32_t *trials_out, double *change_out, double *seconds_out);
int vch1d_pgd_get(vch1d_ctx *ctx, int what, double *out);
/* Relative tracking / terminal errors of the iterations of the last vch1d_pgd_iterate call (G1:425-450);
 * same conventions as vch2d_pgd_errors. */
int vch1d_pgd_errors(vch1d_ctx *ctx, int n_iters, double *tracking_out, double *terminal_out);

#ifdef __cplusplus
}
#endif
#endif /* VCH_H */
