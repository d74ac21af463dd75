// vch_engine1d.hip — host side of the 1D engine (C ABI vch1d_* of include/vch.h).
// Fields are [B][N+1], histories [B][rows][N+1] (rows = M+2: the reference's duplicated t=0 row,
// F1:329-336), contiguous on host and device alike.
#include "vch_common.h"
#include "vch_kernels1d.h"
#include <algorithm>
#include <cmath>

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
                     c->u_trial, c->phiQ, c->p_hist, c->q_hist, c->r_hist};
    for (double *q : all)
        if (q) hipFree(q);
    hipFree(c->stats_dev);
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
            c->scratch, c->stats_dev);
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
