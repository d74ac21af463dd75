// vch_engine1d.hip — host side of the 1D engine (C ABI vch1d_* of include/vch.h).
// Fields are [B][N+1], histories [B][rows][N+1] (rows = M+2: the reference's duplicated t=0 row,
// F1:329-336), contiguous on host and device alike.
#include "vch_common.h"
#include "vch_kernels1d.h"
#include <algorithm>
#include <cmath>
#include <vector>

struct vch1d_ctx {
    vch1d_params prm;
    int B, Mmax, device, n, lvl;
    double h;
    Phys1 P, F;                    // run-time parameters; frozen defaults for the adjoint (B1:29-33)
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    double *scratch;               // [B][NSCR1][n]
    double *tmp[8];                // [B][n]
    double *phi_hist, *u_hist, *u_trial, *phiQ, *p_hist, *q_hist, *r_hist;   // [B][Mmax+2][n], lazy
    double *phiT, *dts, *tgrid, *wx, *alpha_dev, *cost_lvl, *hist_dev;
    double *cost_host;
    int *stats_dev, *stats_host;
    size_t lds_bytes;
    int rows_res;
    // device-resident PGD (vch1d_pgd_*)
    bool pgd_ready = false;
    int pgd_rows = 0;
    vch_opt_params opt;
    double *phi0_dev = nullptr, *phi_trial = nullptr, *chg_dev = nullptr, *tp_dev = nullptr;
    int *skip_dev = nullptr;
    std::vector<double> t_host, chg_host, pgd_cost, pgd_alpha_prev;
    std::vector<std::vector<double>> pgd_cost_hist;
    std::vector<int> pgd_plateau, pgd_k, pgd_done;
    // error metrics of the driver loop (G1:425-450): squared target norms, RMS fallback scale, histories of the last iterate call
    std::vector<double> pgd_denQ2, pgd_denT2, pgd_trk, pgd_trm;
    double pgd_rms = 1.0;
    int pgd_err_n = 0;
};

#define LAUNCH1(kern, grid, block, lds, ...)                                       \
    do {                                                                           \
        hipLaunchKernelGGL(kern, grid, block, lds, c->stream, __VA_ARGS__);        \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess)                                                      \
            return vch_fail(VCH_ERR_HIP, "launch %s: %s", #kern, hipGetErrorString(e_)); \
    } while (0)
#define CTXCHK1(c)                                                        \
    do {                                                                  \
        if (!(c)) return vch_fail(VCH_ERR_ARG, "%s: NULL context", __func__); \
        HIPCHK(hipSetDevice((c)->device));                                \
    } while (0)
#define ARGCHK1(cond, msg)                                            \
    do {                                                              \
        if (!(cond)) return vch_fail(VCH_ERR_ARG, "%s: %s", __func__, msg); \
    } while (0)

static int dalloc1(double **p, size_t n, hipStream_t s) {
    *p = nullptr;
    HIPCHK(hipMalloc((void **)p, n * sizeof(double)));
    HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(double), s));
    return 0;
}
static inline long hs1(const vch1d_ctx *c) { return (long)(c->Mmax + 2) * c->n; }
static int ensure1(vch1d_ctx *c, double **p) {
    if (*p) return 0;
    return dalloc1(p, (size_t)c->B * hs1(c), c->stream);
}
static int up(vch1d_ctx *c, double *dev, const double *host, size_t n) {
    HIPCHK(hipMemcpyAsync(dev, host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}
static int down(vch1d_ctx *c, double *host, const double *dev, size_t n) {
    HIPCHK(hipMemcpyAsync(host, dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
// host [B][rows][n] <-> device [B][Mmax+2][n]
static int up_hist(vch1d_ctx *c, double *dev, const double *host, int rows) {
    for (int b = 0; b < c->B; ++b) VCHCHK(up(c, dev + b * hs1(c), host + (size_t)b * rows * c->n, (size_t)rows * c->n));
    return 0;
}
static int down_hist(vch1d_ctx *c, double *host, const double *dev, int rows) {
    for (int b = 0; b < c->B; ++b)
        HIPCHK(hipMemcpyAsync(host + (size_t)b * rows * c->n, dev + b * hs1(c), sizeof(double) * rows * c->n,
                              hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" vch1d_ctx *vch1d_create(const vch1d_params *p, int batch, int max_steps, int device) {
    if (!p || p->N < 2 || batch < 1 || max_steps < 1 || !(p->Lx > 0)) {
        vch_fail(VCH_ERR_ARG, "vch1d_create: bad arguments");
        return nullptr;
    }
    int lvl = 0;
    while ((p->N) / (1 << lvl) + 1 > NR_MAX && lvl < 3) ++lvl;
    if (lvl > 2) {
        vch_fail(VCH_ERR_ARG, "vch1d_create: N = %d exceeds the supported 4096 (cyclic-reduction rows kept in LDS)", p->N);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();          // do not leave the sticky error for the next launch check
        vch_fail(VCH_ERR_HIP, "hipSetDevice(%d) failed", device);
        return nullptr;
    }
    vch1d_ctx *c = new vch1d_ctx();
    c->prm = *p;
    c->B = batch;
    c->Mmax = max_steps;
    c->device = device;
    c->n = p->N + 1;
    c->lvl = lvl;
    c->h = p->Lx / p->N;
    c->P = Phys1{p->tau, p->gamma, p->c1, p->c2, p->kappa, p->Lx};
    c->F = Phys1{0.05, 10.0, 0.75, 1.0, 0.03 * 0.03, 1.0};      // K1:95-102 defaults, frozen at import in B1:29-33
    c->rows_res = 0;
    auto fail = [&](const char *what) {
        vch_fail(VCH_ERR_HIP, "vch1d_create: %s failed: %s", what, hipGetErrorString(hipGetLastError()));
        return (vch1d_ctx *)nullptr;
    };
    if (hipStreamCreate(&c->stream) != hipSuccess) return fail("hipStreamCreate");
    hipEventCreate(&c->ev0);
    hipEventCreate(&c->ev1);
    const size_t bn = (size_t)batch * c->n;
    if (dalloc1(&c->scratch, bn * NSCR1, c->stream)) return fail("hipMalloc");
    for (auto &t : c->tmp)
        if (dalloc1(&t, bn, c->stream)) return fail("hipMalloc");
    if (dalloc1(&c->phiT, bn, c->stream) || dalloc1(&c->dts, max_steps + 2, c->stream) ||
        dalloc1(&c->tgrid, max_steps + 2, c->stream) || dalloc1(&c->wx, c->n, c->stream) ||
        dalloc1(&c->alpha_dev, batch, c->stream) || dalloc1(&c->cost_lvl, (size_t)batch * (max_steps + 2) * 4, c->stream) ||
        dalloc1(&c->hist_dev, (size_t)batch * 64, c->stream))
        return fail("hipMalloc");
    if (hipMalloc((void **)&c->stats_dev, sizeof(int) * 8 * batch) != hipSuccess) return fail("hipMalloc");
    if (hipHostMalloc((void **)&c->stats_host, sizeof(int) * 8 * batch) != hipSuccess) return fail("hipHostMalloc");
    if (hipHostMalloc((void **)&c->cost_host, sizeof(double) * batch * (max_steps + 2) * 4) != hipSuccess) return fail("hipHostMalloc");
    c->phi_hist = c->u_hist = c->u_trial = c->phiQ = c->p_hist = c->q_hist = c->r_hist = nullptr;
    c->lds_bytes = sizeof(double) * 14 * NR_MAX;
    // > 64 KiB of dynamic LDS needs the opt-in attribute
    hipFuncSetAttribute((const void *)k1d_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
    hipFuncSetAttribute((const void *)k1d_newton, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
    hipFuncSetAttribute((const void *)k1d_backward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
    hipFuncSetAttribute((const void *)k1d_solve<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
    hipFuncSetAttribute((const void *)k1d_solve<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail("hipStreamSynchronize");
    return c;
}

extern "C" void vch1d_destroy(vch1d_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    double *all[] = {c->scratch, c->tmp[0], c->tmp[1], c->tmp[2], c->tmp[3], c->tmp[4], c->tmp[5], c->tmp[6], c->tmp[7],
                     c->phiT, c->dts, c->tgrid, c->wx, c->alpha_dev, c->cost_lvl, c->hist_dev, c->phi_hist, c->u_hist,
                     c->u_trial, c->phiQ, c->p_hist, c->q_hist, c->r_hist, c->phi0_dev, c->phi_trial, c->chg_dev, c->tp_dev};
    for (double *q : all)
        if (q) hipFree(q);
    hipFree(c->stats_dev);
    if (c->skip_dev) hipFree(c->skip_dev);
    hipHostFree(c->stats_host);
    hipHostFree(c->cost_host);
    hipEventDestroy(c->ev0);
    hipEventDestroy(c->ev1);
    hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int vch1d_apply_laplacian(vch1d_ctx *c, const double *v, double *out) {
    CTXCHK1(c);
    ARGCHK1(v && out, "NULL array");
    const size_t bn = (size_t)c->B * c->n;
    VCHCHK(up(c, c->tmp[0], v, bn));
    LAUNCH1(k1d_lap, dim3((c->n + 255) / 256, c->B), dim3(256), 0, c->n, 1.0 / (c->h * c->h), (const double *)c->tmp[0], c->tmp[1]);
    return down(c, out, c->tmp[1], bn);
}

extern "C" int vch1d_residuals(vch1d_ctx *c, const double *pn, const double *po, const double *mn, const double *mo,
                               const double *wn, const double *wo, double dt, double *Rp, double *Rm) {
    CTXCHK1(c);
    ARGCHK1(pn && po && mn && mo && wn && wo && Rp && Rm && dt > 0, "NULL array or dt <= 0");
    const size_t bn = (size_t)c->B * c->n;
    const double *src[6] = {pn, po, mn, mo, wn, wo};
    for (int k = 0; k < 6; ++k) VCHCHK(up(c, c->tmp[k], src[k], bn));
    LAUNCH1(k1d_residuals, dim3((c->n + 255) / 256, c->B), dim3(256), 0, c->P, c->n, 1.0 / (c->h * c->h), dt,
            (const double *)c->tmp[0], (const double *)c->tmp[1], (const double *)c->tmp[2], (const double *)c->tmp[3],
            (const double *)c->tmp[4], (const double *)c->tmp[5], c->tmp[6], c->tmp[7]);
    VCHCHK(down(c, Rp, c->tmp[6], bn));
    return down(c, Rm, c->tmp[7], bn);
}

extern "C" int vch1d_jacobian_solve(vch1d_ctx *c, const double *phi_new, double dt, const double *rhs_phi,
                                    const double *rhs_mu, double *dphi, double *dmu) {
    CTXCHK1(c);
    ARGCHK1(phi_new && rhs_phi && rhs_mu && dphi && dmu && dt > 0, "NULL array or dt <= 0");
    const size_t bn = (size_t)c->B * c->n;
    VCHCHK(up(c, c->tmp[0], phi_new, bn));
    VCHCHK(up(c, c->tmp[1], rhs_phi, bn));
    VCHCHK(up(c, c->tmp[2], rhs_mu, bn));
    SysArgs A{nullptr, nullptr, nullptr, dt, 1.0 / (c->h * c->h), c->P.tau, c->P.c1, c->P.c2, c->P.kappa, c->n};
    LAUNCH1((k1d_solve<0>), dim3(c->B), dim3(T1), c->lds_bytes, A, c->lvl, (const double *)c->tmp[0], (const double *)c->tmp[1],
            (const double *)c->tmp[2], c->tmp[3], c->tmp[4]);
    VCHCHK(down(c, dphi, c->tmp[3], bn));
    return down(c, dmu, c->tmp[4], bn);
}

extern "C" int vch1d_adjoint_solve(vch1d_ctx *c, const double *phi_n, double dt, const double *rhs, double *p_out) {
    CTXCHK1(c);
    ARGCHK1(rhs && p_out && dt >= 0 && (phi_n || dt == 0), "NULL array or dt < 0");
    const size_t bn = (size_t)c->B * c->n;
    if (phi_n) VCHCHK(up(c, c->tmp[0], phi_n, bn));
    else HIPCHK(hipMemsetAsync(c->tmp[0], 0, bn * sizeof(double), c->stream));
    VCHCHK(up(c, c->tmp[1], rhs, bn));
    SysArgs A{nullptr, nullptr, nullptr, dt, 1.0 / (c->h * c->h), c->F.tau, c->F.c1, c->F.c2, 0.0, c->n};
    LAUNCH1((k1d_solve<1>), dim3(c->B), dim3(T1), c->lds_bytes, A, c->lvl, (const double *)c->tmp[0], (const double *)c->tmp[1],
            (const double *)nullptr, c->tmp[3], c->tmp[4]);
    return down(c, p_out, c->tmp[3], bn);
}

extern "C" int vch1d_newton_raphson(vch1d_ctx *c, const double *phi_old, const double *mu_old, const double *w_old,
                                    const double *w_new, double dt, double *phi_new, double *mu_new, double *hist,
                                    int hist_cap, int32_t *n_hist) {
    CTXCHK1(c);
    ARGCHK1(phi_old && mu_old && w_old && w_new && phi_new && mu_new && dt > 0, "NULL array or dt <= 0");
    const int n = c->n;
    for (int b = 0; b < c->B; ++b) {          // scratch layout: phi, mu, w, wnew are the first four arrays
        double *q = c->scratch + (size_t)b * NSCR1 * n;
        VCHCHK(up(c, q, phi_old + (size_t)b * n, n));
        VCHCHK(up(c, q + n, mu_old + (size_t)b * n, n));
        VCHCHK(up(c, q + 2 * n, w_old + (size_t)b * n, n));
        VCHCHK(up(c, q + 3 * n, w_new + (size_t)b * n, n));
    }
    LAUNCH1(k1d_newton, dim3(c->B), dim3(T1), c->lds_bytes, c->P, n, c->h, c->lvl, dt, c->scratch, c->hist_dev, 64, c->stats_dev);
    HIPCHK(hipMemcpyAsync(c->stats_host, c->stats_dev, sizeof(int) * 8 * c->B, hipMemcpyDeviceToHost, c->stream));
    for (int b = 0; b < c->B; ++b) {
        double *q = c->scratch + (size_t)b * NSCR1 * n;
        VCHCHK(down(c, phi_new + (size_t)b * n, q + 4 * n, n));
        VCHCHK(down(c, mu_new + (size_t)b * n, q + 5 * n, n));
    }
    std::vector<double> hh((size_t)c->B * 64);
    VCHCHK(down(c, hh.data(), c->hist_dev, hh.size()));
    for (int b = 0; b < c->B; ++b) {
        if (c->stats_host[b * 8 + 4]) return vch_fail(VCH_ERR_STATE, "Non-finite mass_defect; check phi bounds/log regularization.");
        const int k = std::min(c->stats_host[b * 8 + 0], 64);
        if (n_hist) n_hist[b] = k;
        if (hist)
            for (int j = 0; j < std::min(k, hist_cap); ++j) hist[(size_t)b * hist_cap + j] = hh[(size_t)b * 64 + j];
    }
    return 0;
}

extern "C" int vch1d_forward(vch1d_ctx *c, const double *phi0, const double *u, int u_rows, const double *dt, int M,
                             double *phi_hist_out, vch_stats *stats) {
    CTXCHK1(c);
    ARGCHK1(phi0 && dt && M >= 1 && M <= c->Mmax, "NULL array or M out of range (1..max_steps)");
    // F1:347-353 indexes control_input[step] for every step: fewer than M rows is an IndexError there
    if (u) ARGCHK1(u_rows >= M && u_rows <= c->Mmax + 2, "control rows: need M <= rows <= max_steps+2 (IndexError in the reference)");
    VCHCHK(ensure1(c, &c->phi_hist));
    if (u) {
        VCHCHK(ensure1(c, &c->u_hist));
        VCHCHK(up_hist(c, c->u_hist, u, u_rows));
    }
    VCHCHK(up(c, c->tmp[0], phi0, (size_t)c->B * c->n));
    VCHCHK(up(c, c->dts, dt, M));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    LAUNCH1(k1d_forward, dim3(c->B), dim3(T1), c->lds_bytes, c->P, c->n, c->h, c->lvl, M, (const double *)c->dts,
            (const double *)c->tmp[0], (const double *)(u ? c->u_hist : nullptr), u_rows, hs1(c), c->phi_hist, hs1(c),
            c->scratch, c->stats_dev, (const int *)nullptr);
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipMemcpyAsync(c->stats_host, c->stats_dev, sizeof(int) * 8 * c->B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->rows_res = M + 2;
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (int b = 0; b < c->B; ++b) {
            stats->newton_iters += c->stats_host[b * 8 + 0];
            stats->linear_solves += c->stats_host[b * 8 + 1];
            stats->armijo_trials += c->stats_host[b * 8 + 2];
            stats->linear_iters += c->stats_host[b * 8 + 3];      // 1D: steps left through the line-search-failure return
        }
        stats->seconds = ms * 1e-3;
    }
    for (int b = 0; b < c->B; ++b)
        if (c->stats_host[b * 8 + 4]) return vch_fail(VCH_ERR_STATE, "Non-finite mass_defect; check phi bounds/log regularization.");
    if (phi_hist_out) VCHCHK(down_hist(c, phi_hist_out, c->phi_hist, M + 2));
    return 0;
}

extern "C" int vch1d_backward(vch1d_ctx *c, const double *phi_hist, int rows, const double *t_hist, double h, double b1,
                              double b2, const double *phi_Q, const double *phi_T, double *p_out, double *q_out,
                              double *r_out) {
    CTXCHK1(c);
    ARGCHK1(t_hist && rows >= 2 && rows <= c->Mmax + 2, "NULL t_hist or rows out of range");
    ARGCHK1(std::fabs(h - c->h) <= 1e-12 * c->h, "grid spacing differs from the context's Lx/N");
    if (phi_hist) {
        VCHCHK(ensure1(c, &c->phi_hist));
        VCHCHK(up_hist(c, c->phi_hist, phi_hist, rows));
        c->rows_res = rows;
    } else if (c->rows_res != rows) {
        return vch_fail(VCH_ERR_STATE, "vch1d_backward: no resident history with %d rows", rows);
    }
    if (phi_Q) {
        VCHCHK(ensure1(c, &c->phiQ));
        VCHCHK(up_hist(c, c->phiQ, phi_Q, rows));
    }
    if (phi_T) VCHCHK(up(c, c->phiT, phi_T, (size_t)c->B * c->n));
    VCHCHK(ensure1(c, &c->p_hist));
    VCHCHK(ensure1(c, &c->q_hist));
    VCHCHK(ensure1(c, &c->r_hist));
    const size_t hb = (size_t)c->B * hs1(c) * sizeof(double);
    HIPCHK(hipMemsetAsync(c->p_hist, 0, hb, c->stream));
    HIPCHK(hipMemsetAsync(c->q_hist, 0, hb, c->stream));
    HIPCHK(hipMemsetAsync(c->r_hist, 0, hb, c->stream));
    VCHCHK(up(c, c->tgrid, t_hist, rows));
    LAUNCH1(k1d_backward, dim3(c->B), dim3(T1), c->lds_bytes, c->F, c->n, c->h, c->lvl, rows, (const double *)c->tgrid,
            (const double *)c->phi_hist, (const double *)(phi_Q ? c->phiQ : nullptr),
            (const double *)(phi_T ? c->phiT : nullptr), b1, b2, c->p_hist, c->q_hist, c->r_hist, hs1(c), c->scratch);
    HIPCHK(hipStreamSynchronize(c->stream));
    if (p_out) VCHCHK(down_hist(c, p_out, c->p_hist, rows));
    if (q_out) VCHCHK(down_hist(c, q_out, c->q_hist, rows));
    if (r_out) VCHCHK(down_hist(c, r_out, c->r_hist, rows));
    return 0;
}

extern "C" int vch1d_cost(vch1d_ctx *c, const double *phi_hist, const double *u, const double *phi_Q, const double *phi_T,
                          int rows, const double *x, const double *t_hist, const vch_opt_params *o, double *J_out) {
    CTXCHK1(c);
    ARGCHK1(phi_hist && x && t_hist && o && J_out && rows >= 2 && rows <= c->Mmax + 2, "NULL argument or rows out of range");
    VCHCHK(ensure1(c, &c->phi_hist));
    VCHCHK(up_hist(c, c->phi_hist, phi_hist, rows));
    c->rows_res = rows;
    if (u) { VCHCHK(ensure1(c, &c->u_hist)); VCHCHK(up_hist(c, c->u_hist, u, rows)); }
    if (phi_Q) { VCHCHK(ensure1(c, &c->phiQ)); VCHCHK(up_hist(c, c->phiQ, phi_Q, rows)); }
    if (phi_T) VCHCHK(up(c, c->phiT, phi_T, (size_t)c->B * c->n));
    std::vector<double> wx(c->n, 0.0);         // np.trapezoid weights from the caller's grid (C1:57)
    for (int i = 0; i + 1 < c->n; ++i) {
        const double d = x[i + 1] - x[i];
        wx[i] += 0.5 * d;
        wx[i + 1] += 0.5 * d;
    }
    VCHCHK(up(c, c->wx, wx.data(), c->n));
    LAUNCH1(k1d_cost, dim3(rows, c->B), dim3(T1), 0, c->n, rows, (const double *)c->wx, (const double *)c->phi_hist,
            (const double *)(u ? c->u_hist : nullptr), (const double *)(phi_Q ? c->phiQ : nullptr),
            (const double *)(phi_T ? c->phiT : nullptr), hs1(c), c->cost_lvl);
    VCHCHK(down(c, c->cost_host, c->cost_lvl, (size_t)c->B * rows * 4));
    for (int b = 0; b < c->B; ++b) {
        const double *s = c->cost_host + (size_t)b * rows * 4;
        double i1 = 0, i3 = 0, i4 = 0;
        for (int k = 0; k + 1 < rows; ++k) {
            const double d = t_hist[k + 1] - t_hist[k];
            i1 += d * (s[(k + 1) * 4 + 0] + s[k * 4 + 0]) / 2.0;
            i3 += d * (s[(k + 1) * 4 + 2] + s[k * 4 + 2]) / 2.0;
            i4 += d * (s[(k + 1) * 4 + 3] + s[k * 4 + 3]) / 2.0;
        }
        double *J = J_out + 5 * b;
        J[0] = (o->b1 / 2.0) * i1;
        J[1] = (o->b2 / 2.0) * s[(rows - 1) * 4 + 1];
        J[2] = (o->b3 / 2.0) * i3;
        J[3] = o->kappa_sparsity * i4;
        J[4] = J[0] + J[1] + J[2] + J[3];
    }
    return 0;
}

extern "C" int vch1d_grad_prox(vch1d_ctx *c, const double *u, const double *r, int rows, const double *alpha,
                               const vch_opt_params *o, double *u_out) {
    CTXCHK1(c);
    ARGCHK1(u && r && alpha && o && u_out && rows >= 1 && rows <= c->Mmax + 2, "NULL argument or rows out of range");
    VCHCHK(ensure1(c, &c->u_hist));
    VCHCHK(ensure1(c, &c->r_hist));
    VCHCHK(ensure1(c, &c->u_trial));
    VCHCHK(up_hist(c, c->u_hist, u, rows));
    VCHCHK(up_hist(c, c->r_hist, r, rows));
    VCHCHK(up(c, c->alpha_dev, alpha, c->B));
    LAUNCH1(k1d_grad_prox, dim3(rows, c->B), dim3(T1), 0, c->n, (const double *)c->u_hist, (const double *)c->r_hist, hs1(c),
            (const double *)c->alpha_dev, o->b3, o->kappa_sparsity, o->u_min, o->u_max, c->u_trial, (double *)nullptr);
    return down_hist(c, u_out, c->u_trial, rows);
}

extern "C" int vch1d_free_energy(vch1d_ctx *c, const double *phi_hist, int rows, const double *w_hist, double h, double eps,
                                 double *E_out) {
    CTXCHK1(c);
    ARGCHK1(phi_hist && E_out && rows >= 1 && rows <= c->Mmax + 2 && h > 0, "NULL argument, rows out of range or h <= 0");
    VCHCHK(ensure1(c, &c->phi_hist));
    VCHCHK(up_hist(c, c->phi_hist, phi_hist, rows));
    c->rows_res = rows;
    if (w_hist) {
        VCHCHK(ensure1(c, &c->u_trial));
        VCHCHK(up_hist(c, c->u_trial, w_hist, rows));
    }
    LAUNCH1(k1d_energy, dim3(rows, c->B), dim3(T1), 0, c->n, c->P.c1, c->P.c2, eps > 0 ? eps : 1e-8, (const double *)c->phi_hist,
            (const double *)(w_hist ? c->u_trial : nullptr), hs1(c), c->cost_lvl);
    VCHCHK(down(c, c->cost_host, c->cost_lvl, (size_t)c->B * rows * 4));
    for (long k = 0; k < (long)c->B * rows; ++k) {
        const double *s = c->cost_host + 4 * k;
        double E = (c->P.kappa / (2.0 * h)) * s[0] + h * s[1];
        if (w_hist) E -= h * s[2];
        E_out[k] = E;
    }
    return 0;
}

// ------------------------------------------------------------------------------------
// device-resident PGD loop (G1:333-477): control, state history, adjoint and targets stay in HBM
// ------------------------------------------------------------------------------------
static void trapz_x(const vch1d_ctx *c, const double *x, std::vector<double> &wx) {
    wx.assign(c->n, 0.0);                      // np.trapezoid weights from the caller's grid (C1:57)
    for (int i = 0; i + 1 < c->n; ++i) {
        const double d = x[i + 1] - x[i];
        wx[i] += 0.5 * d;
        wx[i + 1] += 0.5 * d;
    }
}

// J[b][5] of (phi, u) on the device; wx already uploaded
static int cost1_core(vch1d_ctx *c, const double *phi_dev, const double *u_dev, int rows, double *J_out,
                      double *raw_out = nullptr /* [B][2] = {int int (phi - phi_Q)^2, int (phi_end - phi_T)^2} */) {
    const vch_opt_params *o = &c->opt;
    LAUNCH1(k1d_cost, dim3(rows, c->B), dim3(T1), 0, c->n, rows, (const double *)c->wx, phi_dev, u_dev,
            (const double *)c->phiQ, (const double *)c->phiT, hs1(c), c->cost_lvl);
    VCHCHK(down(c, c->cost_host, c->cost_lvl, (size_t)c->B * rows * 4));
    const double *t = c->t_host.data();
    for (int b = 0; b < c->B; ++b) {
        const double *s = c->cost_host + (size_t)b * rows * 4;
        double i1 = 0, i3 = 0, i4 = 0;
        for (int k = 0; k + 1 < rows; ++k) {
            const double d = t[k + 1] - t[k];
            i1 += d * (s[(k + 1) * 4 + 0] + s[k * 4 + 0]) / 2.0;
            i3 += d * (s[(k + 1) * 4 + 2] + s[k * 4 + 2]) / 2.0;
            i4 += d * (s[(k + 1) * 4 + 3] + s[k * 4 + 3]) / 2.0;
        }
        double *J = J_out + 5 * b;
        J[0] = (o->b1 / 2.0) * i1;
        J[1] = (o->b2 / 2.0) * s[(rows - 1) * 4 + 1];
        J[2] = (o->b3 / 2.0) * i3;
        J[3] = o->kappa_sparsity * i4;
        J[4] = J[0] + J[1] + J[2] + J[3];
        if (raw_out) {
            raw_out[2 * b] = i1;
            raw_out[2 * b + 1] = s[(rows - 1) * 4 + 1];
        }
    }
    return 0;
}

// int_t int_x a^2 (rows > 1) or int_x a^2 (rows == 1) per trajectory with the cost's weights; arr [B][stride]
static int l2sq1_core(vch1d_ctx *c, const double *arr, long stride, int rows, double *out) {
    LAUNCH1(k1d_cost, dim3(rows, c->B), dim3(T1), 0, c->n, rows, (const double *)c->wx, arr, (const double *)nullptr,
            (const double *)nullptr, (const double *)nullptr, stride, c->cost_lvl);
    VCHCHK(down(c, c->cost_host, c->cost_lvl, (size_t)c->B * rows * 4));
    const double *t = c->t_host.data();
    for (int b = 0; b < c->B; ++b) {
        const double *s = c->cost_host + (size_t)b * rows * 4;
        if (rows == 1) { out[b] = s[0]; continue; }
        double acc = 0;
        for (int k = 0; k + 1 < rows; ++k) acc += (t[k + 1] - t[k]) * (s[(k + 1) * 4] + s[k * 4]) / 2.0;
        out[b] = acc;
    }
    return 0;
}

static int fwd1_core(vch1d_ctx *c, const double *u_dev, int rows, double *hist_dev, const int *skip_dev) {
    const int M = rows - 2;
    LAUNCH1(k1d_forward, dim3(c->B), dim3(T1), c->lds_bytes, c->P, c->n, c->h, c->lvl, M, (const double *)c->dts,
            (const double *)c->phi0_dev, u_dev, rows, hs1(c), hist_dev, hs1(c), c->scratch, c->stats_dev, skip_dev);
    HIPCHK(hipMemcpyAsync(c->stats_host, c->stats_dev, sizeof(int) * 8 * c->B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int b = 0; b < c->B; ++b)
        if (c->stats_host[b * 8 + 4]) return vch_fail(VCH_ERR_STATE, "Non-finite mass_defect; check phi bounds/log regularization.");
    return 0;
}

extern "C" int vch1d_pgd_init(vch1d_ctx *c, const double *phi0, const double *phi_T, const double *phi_Q, const double *x,
                              const double *t_hist, int rows, const double *dt, const vch_opt_params *opt, double *J0_out) {
    CTXCHK1(c);
    ARGCHK1(phi0 && phi_T && x && t_hist && dt && opt, "NULL argument");
    ARGCHK1(rows >= 3 && rows <= c->Mmax + 2, "rows out of range (3..max_steps+2)");
    const int B = c->B, M = rows - 2;
    for (int k = 0; k < M; ++k) ARGCHK1(dt[k] > 0, "dt must be positive");
    c->opt = *opt;
    c->pgd_rows = rows;
    c->t_host.assign(t_hist, t_hist + rows);
    double **hs[] = {&c->phi_hist, &c->u_hist, &c->u_trial, &c->phiQ, &c->p_hist, &c->q_hist, &c->r_hist, &c->phi_trial};
    for (auto p : hs) VCHCHK(ensure1(c, p));
    if (!c->phi0_dev) VCHCHK(dalloc1(&c->phi0_dev, (size_t)B * c->n, c->stream));
    if (!c->chg_dev) VCHCHK(dalloc1(&c->chg_dev, (size_t)B * (c->Mmax + 2) * 2, c->stream));
    if (!c->tp_dev) VCHCHK(dalloc1(&c->tp_dev, c->Mmax + 2, c->stream));
    if (!c->skip_dev) HIPCHK(hipMalloc((void **)&c->skip_dev, sizeof(int) * B));
    c->chg_host.assign((size_t)B * rows * 2, 0.0);
    VCHCHK(up(c, c->phi0_dev, phi0, (size_t)B * c->n));
    VCHCHK(up(c, c->phiT, phi_T, (size_t)B * c->n));
    VCHCHK(up(c, c->dts, dt, M));
    VCHCHK(up(c, c->tgrid, t_hist, rows));
    std::vector<double> wx;
    trapz_x(c, x, wx);
    VCHCHK(up(c, c->wx, wx.data(), c->n));
    HIPCHK(hipStreamSynchronize(c->stream));          // wx is a local
    // uncontrolled march (G1:341), u = 0
    VCHCHK(fwd1_core(c, nullptr, rows, c->phi_hist, nullptr));
    c->rows_res = rows;
    HIPCHK(hipMemsetAsync(c->u_hist, 0, sizeof(double) * B * hs1(c), c->stream));
    if (phi_Q) {
        VCHCHK(up_hist(c, c->phiQ, phi_Q, rows));
    } else {
        std::vector<double> tp(rows);
        const double Tend = t_hist[rows - 1] > 0 ? t_hist[rows - 1] : 1.0;
        for (int k = 0; k < rows; ++k) tp[k] = t_hist[k] / Tend;
        VCHCHK(up(c, c->tp_dev, tp.data(), rows));
        HIPCHK(hipStreamSynchronize(c->stream));
        LAUNCH1(k1d_ramp, dim3(rows, B), dim3(T1), 0, c->n, (const double *)c->tp_dev, (const double *)c->phi_hist,
                (const double *)c->phiT, hs1(c), c->phiQ);
    }
    std::vector<double> J(5 * B);
    VCHCHK(cost1_core(c, c->phi_hist, c->u_hist, rows, J.data()));
    c->pgd_denQ2.assign(B, 0.0);
    c->pgd_denT2.assign(B, 0.0);
    VCHCHK(l2sq1_core(c, c->phiQ, hs1(c), rows, c->pgd_denQ2.data()));
    VCHCHK(l2sq1_core(c, c->phiT, c->n, 1, c->pgd_denT2.data()));
    c->pgd_rms = std::sqrt(std::max(x[c->n - 1] - x[0], 1e-30) * std::max(t_hist[rows - 1] - t_hist[0], 1e-30));
    c->pgd_err_n = 0;
    c->pgd_cost.assign(B, 0.0);
    for (int b = 0; b < B; ++b) c->pgd_cost[b] = J[5 * b + 4];
    if (J0_out) memcpy(J0_out, J.data(), sizeof(double) * 5 * B);
    c->pgd_alpha_prev.assign(B, opt->alpha_max);
    c->pgd_plateau.assign(B, 0);
    c->pgd_k.assign(B, 0);
    c->pgd_done.assign(B, 0);
    c->pgd_cost_hist.assign(B, std::vector<double>());
    for (int b = 0; b < B; ++b) c->pgd_cost_hist[b].push_back(c->pgd_cost[b]);
    c->pgd_ready = true;
    return 0;
}

static int copy_traj1(vch1d_ctx *c, double *dst, const double *src, int b, int rows) {
    HIPCHK(hipMemcpyAsync(dst + b * hs1(c), src + b * hs1(c), sizeof(double) * rows * c->n, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

extern "C" int vch1d_pgd_iterate(vch1d_ctx *c, int n_iters, double *cost_out, double *alpha_out, int32_t *trials_out,
                                 double *change_out, double *seconds_out) {
    CTXCHK1(c);
    if (!c->pgd_ready) return vch_fail(VCH_ERR_STATE, "vch1d_pgd_iterate: call vch1d_pgd_init first");
    ARGCHK1(n_iters >= 1, "n_iters must be >= 1");
    const int B = c->B, rows = c->pgd_rows;
    const vch_opt_params &O = c->opt;
    constexpr int MAX_LS = 5;                 // G1:74
    constexpr double LS_BETA = 0.8;
    double sec[3] = {0, 0, 0};                // backward, optimistic round, backtracking rounds
    auto tick = [&](hipEvent_t e) { return hipEventRecord(e, c->stream); };
    auto lap = [&]() {
        float ms = 0;
        hipEventSynchronize(c->ev1);
        hipEventElapsedTime(&ms, c->ev0, c->ev1);
        return (double)ms * 1e-3;
    };
    std::vector<double> alpha(B), Jt(5 * B), raw(2 * B);
    std::vector<int> accepted(B), trials(B);
    const size_t hb = (size_t)B * hs1(c) * sizeof(double);
    c->pgd_err_n = n_iters;
    c->pgd_trk.assign((size_t)B * n_iters, std::nan(""));
    c->pgd_trm.assign((size_t)B * n_iters, std::nan(""));
    int done_iters = 0;
    for (int it = 0; it < n_iters; ++it) {
        bool all_done = true;
        for (int b = 0; b < B; ++b) all_done &= (c->pgd_done[b] != 0);
        if (all_done) break;
        // --- adjoint sweep (G1:356); only r is consumed
        HIPCHK(tick(c->ev0));
        HIPCHK(hipMemsetAsync(c->p_hist, 0, hb, c->stream));
        HIPCHK(hipMemsetAsync(c->q_hist, 0, hb, c->stream));
        HIPCHK(hipMemsetAsync(c->r_hist, 0, hb, c->stream));
        LAUNCH1(k1d_backward, dim3(B), dim3(T1), c->lds_bytes, c->F, c->n, c->h, c->lvl, rows, (const double *)c->tgrid,
                (const double *)c->phi_hist, (const double *)c->phiQ, (const double *)c->phiT, O.b1, O.b2, c->p_hist, c->q_hist,
                c->r_hist, hs1(c), c->scratch);
        HIPCHK(tick(c->ev1));
        sec[0] += lap();
        for (int b = 0; b < B; ++b) {
            alpha[b] = c->pgd_alpha_prev[b];
            accepted[b] = c->pgd_done[b] ? 1 : 0;
            trials[b] = 0;
        }
        // round 0: optimistic step with alpha_prev (G1:365-372), which is also the first trial of the line
        // search (alpha_init = alpha_prev, G1:383) -- that repetition is not recomputed; rounds 1..4: alpha *= 0.8
        for (int round = 0; round < MAX_LS; ++round) {
            HIPCHK(tick(c->ev0));
            VCHCHK(up(c, c->alpha_dev, alpha.data(), B));
            HIPCHK(hipMemcpyAsync(c->skip_dev, accepted.data(), sizeof(int) * B, hipMemcpyHostToDevice, c->stream));
            LAUNCH1(k1d_grad_prox, dim3(rows, B), dim3(T1), 0, c->n, (const double *)c->u_hist, (const double *)c->r_hist, hs1(c),
                    (const double *)c->alpha_dev, O.b3, O.kappa_sparsity, O.u_min, O.u_max, c->u_trial, c->chg_dev);
            VCHCHK(fwd1_core(c, c->u_trial, rows, c->phi_trial, c->skip_dev));
            VCHCHK(cost1_core(c, c->phi_trial, c->u_trial, rows, Jt.data(), raw.data()));
            VCHCHK(down(c, c->chg_host.data(), c->chg_dev, (size_t)B * rows * 2));
            HIPCHK(tick(c->ev1));
            sec[round == 0 ? 1 : 2] += lap();
            bool pending = false;
            for (int b = 0; b < B; ++b) {
                if (accepted[b]) continue;
                trials[b] = round + 1;
                const bool ok = Jt[5 * b + 4] < c->pgd_cost[b];
                const bool last = (round == MAX_LS - 1);
                if (!ok && !last) {
                    alpha[b] *= LS_BETA;
                    pending = true;
                    continue;
                }
                accepted[b] = 1;
                const double a_k = ok ? alpha[b] : alpha[b] * LS_BETA;     // G1:112-113 returns the once-more reduced step
                const double c_n = Jt[5 * b + 4];
                double d2 = 0, n2 = 0;
                for (int r = 0; r < rows; ++r) {
                    d2 += c->chg_host[((size_t)b * rows + r) * 2];
                    n2 += c->chg_host[((size_t)b * rows + r) * 2 + 1];
                }
                const double change = std::sqrt(d2) / (std::sqrt(n2) + 1e-9);
                {   // relative tracking / terminal errors of the accepted state (G1:438-450)
                    double denQ = std::sqrt(std::max(c->pgd_denQ2[b], 0.0));
                    if (denQ < 1e-9 * c->pgd_rms) denQ = c->pgd_rms;
                    c->pgd_trk[(size_t)b * n_iters + it] = std::sqrt(std::max(raw[2 * b], 0.0)) / (denQ + 1e-12);
                    c->pgd_trm[(size_t)b * n_iters + it] =
                        std::sqrt(std::max(raw[2 * b + 1], 0.0)) / (std::sqrt(std::max(c->pgd_denT2[b], 0.0)) + 1e-12);
                }
                const int k = c->pgd_k[b];
                auto &ch = c->pgd_cost_hist[b];
                ch.push_back(c_n);
                if (k > 0 && std::fabs(ch[ch.size() - 1] - ch[ch.size() - 2]) < 1e-7) c->pgd_plateau[b]++;
                else c->pgd_plateau[b] = 0;
                if (c->pgd_plateau[b] >= 10) {
                    c->pgd_alpha_prev[b] = std::min(O.alpha_max, a_k * 2.0);
                    c->pgd_plateau[b] = 0;
                } else {
                    c->pgd_alpha_prev[b] = std::min(O.alpha_max, a_k * 1.2);
                }
                VCHCHK(copy_traj1(c, c->u_hist, c->u_trial, b, rows));
                if (change < 1e-5 && k > 10) {
                    c->pgd_done[b] = 1;                 // G1:462-465: u_k taken, state and cost keep the previous iterate
                } else {
                    VCHCHK(copy_traj1(c, c->phi_hist, c->phi_trial, b, rows));
                    c->pgd_cost[b] = c_n;
                }
                c->pgd_k[b] = k + 1;
                if (cost_out) cost_out[(long)b * n_iters + it] = c_n;
                if (alpha_out) alpha_out[(long)b * n_iters + it] = a_k;
                if (trials_out) trials_out[(long)b * n_iters + it] = trials[b];
                if (change_out) change_out[(long)b * n_iters + it] = change;
            }
            HIPCHK(hipStreamSynchronize(c->stream));
            if (!pending) break;
        }
        done_iters = it + 1;
    }
    if (seconds_out) memcpy(seconds_out, sec, sizeof(sec));
    return done_iters;
}

extern "C" int vch1d_pgd_errors(vch1d_ctx *c, int n_iters, double *tracking_out, double *terminal_out) {
    CTXCHK1(c);
    if (!c->pgd_ready) return vch_fail(VCH_ERR_STATE, "vch1d_pgd_errors: call vch1d_pgd_init first");
    ARGCHK1(n_iters == c->pgd_err_n && n_iters >= 1, "n_iters differs from the last vch1d_pgd_iterate call");
    const size_t n = (size_t)c->B * n_iters;
    if (tracking_out) memcpy(tracking_out, c->pgd_trk.data(), n * sizeof(double));
    if (terminal_out) memcpy(terminal_out, c->pgd_trm.data(), n * sizeof(double));
    return 0;
}

extern "C" int vch1d_pgd_get(vch1d_ctx *c, int what, double *out) {
    CTXCHK1(c);
    ARGCHK1(out && what >= 0 && what <= 3, "NULL out or what not in 0..3");
    if (!c->pgd_ready) return vch_fail(VCH_ERR_STATE, "vch1d_pgd_get: call vch1d_pgd_init first");
    const double *src = what == 0 ? c->u_hist : what == 1 ? c->phi_hist : what == 2 ? c->r_hist : c->phiQ;
    return down_hist(c, out, src, c->pgd_rows);
}

