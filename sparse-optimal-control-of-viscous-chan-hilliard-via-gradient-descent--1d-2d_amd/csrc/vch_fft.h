// vch_fft.h — in-LDS FFT implementation of the fast-diagonalisation preconditioner for
// power-of-two grids (N = 16 ... 2048 intervals per axis).
//
// The DCT-I (REDFT00)  E(x)_k = x_0 + (-1)^k x_N + 2 sum_{0<j<N} x_j cos(pi j k / N)  diagonalises the
// mirrored-Neumann Laplacian and is its own inverse up to 2N, so
//     (c0 + m (c1 + c2 m))^-1 v  =  E2( mult o E2(v) ) / (4 Nf Ns),     E2 = E along both axes.
// E of a real sequence is the DFT of its even extension (length L = 2N), which is real; two real
// rows a, b are therefore transformed by ONE complex FFT of a + i b: FFT = E(a) + i E(b).
//
// A workgroup of C/8 threads owns an LDS image of C complex doubles = NFFT = C/L transforms
// (2 NFFT rows or columns of the plane); C = 1024 for L <= 1024, i.e. TWO wavefronts and 16 KiB
// of LDS per workgroup, so ~10 independent workgroups (20 waves) share a CU (C = 2048 / 4096 for L = 2048 / 4096).
// The FFT is a Stockham autosort done in place: every thread reads its operands, barrier, writes them back
// transposed, barrier.  Measured on MI355X at 512^2 x 8 (profiles/r01_c_fft_variants.txt, r01_i_fft_phases.txt) a
// pass kernel is bound by this chain of barrier-separated phases, so the plans minimise their number:
//   * lengths 512 / 1024 / 2048 start with radix-8 passes (one butterfly per thread): 8x8x8, 8x8x4x4, 8x8x8x4;
//     other lengths radix-4 (plus one radix-2 pass when log2 L is odd);
//   * the first pass can take its inputs from global memory (Ingest) and the last pass can hand its outputs over
//     in registers (Emit) instead of going through the image.
// Twiddles come from a table of exp(-2 pi i m / L) computed in long double on the host.
//
//   k_dct_rows : E along the fast (contiguous) axis for 2 NFFT rows; optional epilogue
//                (weighted dot partial for the CG scalars)
//   k_dct_cols : E along the slow axis for 2 NFFT columns, spectral multiplier, E again (the
//                forward and the backward slow-axis transforms fused: one load, one store)
#pragma once
#include "vch_common.h"
#include "vch_kernels2d.h"
#include "vch_gemm.h"        // SpecArgs

#ifdef VCH_FFT_TIMING
// phase timing of the DCT kernels (tuning builds only, scripts/fft_phases.py): per workgroup
// {start, after load, after FFT(s), end} in s_memrealtime ticks (100 MHz)
__device__ long long *g_fft_dbg = nullptr;
#define FFT_STAMP(i) do { if (g_fft_dbg && threadIdx.x == 0) g_fft_dbg[(((long)blockIdx.z * gridDim.x + blockIdx.x) * 4 + (i))] = wall_clock64(); } while (0)
#else
#define FFT_STAMP(i) do { } while (0)
#endif

struct FftAxis {
    int N, L, logL;          // intervals, FFT length 2N, log2 L
    const double2 *tw;       // exp(-2 pi i m / L), m = 0..L-1
};


__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

#ifndef FFT_RCP
#define FFT_RCP 1             // spectral multiplier of a column pair from ONE v_rcp_f64 of the product of the two
                              // denominators + two Newton steps instead of two IEEE divisions
#endif
// scale * (mult_m ? m : 1) / (c0 + m (c1 + c2 m)) for the two columns of a pair.  The denominators are positive and of
// moderate magnitude (1/dt ... c2 m^2), so the operand scaling and fix-up of an IEEE division buy nothing: one hardware
// reciprocal of pa*pb refined by two Newton steps, then 1/pa = r pb, 1/pb = r pa (within 2 ulp of the divisions).
struct SpecArgs;
template <class SP>
__device__ __forceinline__ void spec_mult2(const SP &sp, double c1, double scale, double ma, double mb, double &fa, double &fb,
                                           int mode) {
    const double pa = sp.c0 + ma * (c1 + sp.c2 * ma), pb = sp.c0 + mb * (c1 + sp.c2 * mb);
    // numerator: 1 (inverse of P), m (M P^-1), or mode 2 = -(c0 + c2 m^2) (the multiplier -E of the right-scaled CG form)
    const double na = scale * (mode == 2 ? -(sp.c0 + sp.c2 * ma * ma) : (mode ? ma : 1.0));
    const double nb = scale * (mode == 2 ? -(sp.c0 + sp.c2 * mb * mb) : (mode ? mb : 1.0));
#if FFT_RCP
    const double d = pa * pb;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    fa = na * (r * pb);
    fb = nb * (r * pa);
#else
    fa = na / pa;
    fb = nb / pb;
#endif
}

// LDS image addressing.  A wavefront's ds_*_b128 is served 8 lanes (one 128-byte row of banks) per clock.
// Reads of every pass take consecutive complex numbers (conflict-free); the Stockham writes of the first two
// radix-4 passes do not: lane j writes index 4j + r (pass 0: 8 lanes fall on 2 of the 8 slots of a row) and
// 16 (j >> 2) + (j & 3) + 4r (pass 1: 4 of 8).  XOR-ing the slot (index bits 0..2) with (i3, i4, i4) makes
// both patterns -- and still every aligned run of 8 consecutive indices -- hit 8 distinct slots.
// FFT_SWZ=0 restores the plain image, FFT_PAD=1 the padded one (profiles/r01_c_fft_variants.txt).
#ifndef FFT_PAD
#define FFT_PAD 0
#endif
#ifndef FFT_SWZ
#define FFT_SWZ 1
#endif
#if FFT_PAD
#define PADC(c) ((c) + ((c) >> 2))
#elif FFT_SWZ
#define PADC(c) ((c) ^ ((((c) >> 3) & 1) | ((((c) >> 4) & 1) * 6)))
#else
#define PADC(c) (c)
#endif
// Lengths 512 / 1024 / 2048 (grids 256^2, 512^2, 1024^2) start with radix-8 passes (FFT_R8): plans 8x8x8, 8x8x4x4 and
// 8x8x8x4, i.e. 3 / 4 / 4 LDS exchanges instead of 5 / 5 / 6.  Those passes write index 8j + r, 64 (j >> 3) + (j & 7) + 8r,
// 512 (j >> 6) + (j & 63) + 64r, for which the slot is XOR-ed with (i3, i4, i5).
#ifndef FFT_R8
#define FFT_R8 1
#endif
#ifndef FFT_ROWS_INGEST
#define FFT_ROWS_INGEST 1
#endif
#ifndef FFT_COLS_PAIR
#define FFT_COLS_PAIR 1       // column pass: 16-byte accesses of a column pair (-1.3 % on the march, r02_fft_variants.txt)
#endif
#ifndef FFT_COLS_FUSE
#define FFT_COLS_FUSE 1       // column pass: the last pass of the first transform hands its outputs to the first pass of the
                              // second in registers (same eight indices per thread), multiplier applied on the way
#endif
#ifndef CG_ROWS_STAGED
#define CG_ROWS_STAGED 1      // first pass of a CG sweep: visit every node once and stage the even extension in LDS (0: feed the
                              // first FFT pass from global memory, which reads each operand twice; -5 % on the march, r02_fft_variants.txt)
#endif
#ifndef FFT_COLS_INGEST
#define FFT_COLS_INGEST 0
#endif
#ifndef FFT_COLS_EMIT
#define FFT_COLS_EMIT 1
#endif
// FFT_W64 (off): the 1024-point FFT (512^2 grid) done by ONE wavefront with the plan 16 x 8 x 8: a radix-16 butterfly per
// lane fed from global memory, then two radix-8 passes with two butterflies per lane -- two LDS exchanges instead of
// three, no workgroup barrier that ever waits for another wave, the last pass emitted from registers.  Its radix-16
// pass writes index 16 j + r, for which the 16-byte slot within a 256-byte row of banks is XOR-ed with index bits 4..7.
#ifndef FFT_W64
#define FFT_W64 0      // measured slower than the two-wave 8x8x4x4 plan (profiles/r02_fft_variants.txt): opt-in A/B knob
#endif
template <int LOGL>
__device__ __forceinline__ int swz(int c) {
#if FFT_R8 && FFT_SWZ && !FFT_PAD
    if (LOGL == 10 && FFT_W64) return c ^ ((c >> 4) & 15);
    if (LOGL >= 9 && LOGL <= 11) return c ^ ((c >> 3) & 7);
#endif
    return PADC(c);
}
template <int C>
struct FftLds {
    static constexpr int SIZE = FFT_PAD ? C + C / 4 + 4 : C;
};

// In-place FFT of the NFFT = C/L sequences stored back to back in buf (C/16 threads).
// threads per workgroup for an image of C complex doubles (tuning knob: C/16 = one butterfly
// quartet per thread per pass; fewer butterflies per thread = more wavefronts per CU at equal LDS)
#ifndef FFT_BPT
#define FFT_BPT 2
#endif
template <int C, int LOGL = 0>
struct FftThreads {
    static constexpr int T = (LOGL == 10 && C == 1024 && FFT_W64 && FFT_R8) ? 64 : C / (4 * FFT_BPT);
};

// LOGL > 0: log2 of the FFT length is a compile-time constant (all index arithmetic folds and the
// pass loop unrolls; the kernels are VALU-issue bound, so this matters); LOGL = 0: run-time length.
//
// Emit: what happens to the outputs of the LAST pass (compile-time lengths 512 / 1024 / 2048 only).  Default: they go
// back into the image like those of every other pass.  Otherwise emit(index in the image, value) is called for each
// output instead -- the kernels store the wanted half of the spectrum straight to global memory (Emit::TO_LDS = 0: no
// barriers at all around the last pass) or write it back scaled (TO_LDS = 1: the spectral multiplier costs no pass).
struct FftNoEmit {
    static constexpr int ACTIVE = 0, TO_LDS = 1;
    __device__ __forceinline__ void operator()(int, double2) const {}
};

// Ingest: where the FIRST pass takes its inputs from (same lengths).  Default: the image.  Otherwise ingest(index in the
// image) supplies them -- the kernels read global memory directly, so the image is never staged (no fill loop, no barrier
// before the first pass).
struct FftNoIngest {
    enum { ACTIVE = 0 };
    __device__ __forceinline__ double2 operator()(int) const { return make_double2(0.0, 0.0); }
};

// ---- one-wavefront 1024-point FFT (plan 16 x 8 x 8) ----
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// 8-point DFT, natural order in and out (o_m = sum_n a_n exp(-2 pi i n m / 8))
__device__ __forceinline__ void dft8(const double2 (&a)[8], double2 (&o)[8]) {
    constexpr double RH = 0.70710678118654752440;
    double2 bb[4], cc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        bb[n] = cadd(a[n], a[n + 4]);
        cc[n] = csub(a[n], a[n + 4]);
    }
    cc[1] = make_double2(RH * (cc[1].x + cc[1].y), RH * (cc[1].y - cc[1].x));      // * (1 - i)/sqrt 2
    cc[2] = make_double2(cc[2].y, -cc[2].x);                                       // * (-i)
    cc[3] = make_double2(RH * (cc[3].y - cc[3].x), -RH * (cc[3].x + cc[3].y));     // * (-1 - i)/sqrt 2
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double2 *v = h ? cc : bb;
        const double2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]);
        const double2 t3 = make_double2(v[1].y - v[3].y, -(v[1].x - v[3].x));      // -i (v1 - v3)
        o[h + 0] = cadd(t0, t2);
        o[h + 2] = cadd(t1, t3);
        o[h + 4] = csub(t0, t2);
        o[h + 6] = csub(t1, t3);
    }
}

// 16-point DFT, natural order in and out: b_n = a_n + a_{n+8}, c_n = (a_n - a_{n+8}) W16^n, then DFT8(b) gives the even
// and DFT8(c) the odd outputs
__device__ __forceinline__ void dft16(const double2 (&a)[16], double2 (&o)[16]) {
    constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173, RH = 0.70710678118654752440;
    constexpr double WR[8] = {1.0, C1, RH, S1, 0.0, -S1, -RH, -C1};
    constexpr double WI[8] = {0.0, -S1, -RH, -C1, -1.0, -C1, -RH, -S1};
    double2 b[8], c[8], e[8], f[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        b[n] = cadd(a[n], a[n + 8]);
        const double2 d = csub(a[n], a[n + 8]);
        c[n] = make_double2(d.x * WR[n] - d.y * WI[n], d.x * WI[n] + d.y * WR[n]);
    }
    dft8(b, e);
    dft8(c, f);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        o[2 * m] = e[m];
        o[2 * m + 1] = f[m];
    }
}

// One wavefront (threadIdx.x & 63 = lane; the workgroup IS that wavefront) transforms the 1024 complex entries of buf.
// Stockham autosort: pass A radix 16 (Ns = 1), passes B, C radix 8 (Ns = 16, 128), two butterflies per lane.
template <bool EMIT, bool EMIT_TO_LDS, bool INGEST, class Emit, class Ingest>
__device__ __forceinline__ void fft1024_wave(double2 *buf, const FftAxis ax, Emit emit, Ingest ingest) {
    constexpr int LG = 10;
    const int lane = threadIdx.x & 63;
    {   // pass A: inputs x[lane + 64 r], outputs y[16 lane + r]
        double2 a[16], o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = INGEST ? ingest(lane + 64 * r) : buf[swz<LG>(lane + 64 * r)];
        dft16(a, o);
        if (!INGEST) __syncthreads();             // every lane has read the image
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[swz<LG>(16 * lane + r)] = o[r];
        __syncthreads();
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {             // passes B (Ns = 16) and C (Ns = 128)
        const int lNs = ps ? 7 : 4, Ns = 1 << lNs;
        double2 o[2][8];
        int wb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = lane + 64 * i, k = j & (Ns - 1);
            double2 a[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = buf[swz<LG>(j + 128 * r)];
            const double2 w1 = ax.tw[k << (LG - 3 - lNs)];      // exp(-2 pi i k / (8 Ns))
            double2 w = w1;
            a[1] = cmul(a[1], w);
#pragma unroll
            for (int r = 2; r < 8; ++r) {
                w = cmul(w, w1);
                a[r] = cmul(a[r], w);
            }
            dft8(a, o[i]);
            wb[i] = ((j >> lNs) << (lNs + 3)) + k;
        }
        if (EMIT && ps == 1) {
            if (EMIT_TO_LDS) __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) emit(wb[i] + r * Ns, o[i][r]);
            if (EMIT_TO_LDS) __syncthreads();
        } else {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) buf[swz<LG>(wb[i] + r * Ns)] = o[i][r];
            __syncthreads();
        }
    }
}

// REGMODE (two transforms back to back, k_dct_cols): 1 = the last pass leaves its outputs in xr[0..8) instead of the image,
// 2 = the first pass takes its inputs from xr.  With one butterfly octet per thread the first radix-8 pass reads the
// indices j + s L/8 (s = 0..7) of its transform, and the last pass of the plan produces exactly those: a radix-8 pass
// with Ns = L/8 writes j + r L/8, a radix-4 pass with Ns = L/4 and two butterflies (j, j + L/8) per thread writes
// j + (i + 2 r) L/8.  FftRegOk says for which images that holds.
template <int C, int LOGL>
struct FftRegOk {
    static constexpr bool V = FFT_R8 && !FFT_W64 && (LOGL == 9 || ((LOGL == 10 || LOGL == 11) && C == (1 << LOGL)));
};
template <int C, int LOGL, class Emit = FftNoEmit, class Ingest = FftNoIngest, int REGMODE = 0>
__device__ __forceinline__ void fft_lds(double2 *buf, const FftAxis ax, Emit emit = Emit(), Ingest ingest = Ingest(),
                                        double2 *xr = nullptr) {
    static_assert(REGMODE == 0 || FftRegOk<C, LOGL>::V, "register hand-over needs a compile-time radix-8 plan");
    constexpr bool EMIT = Emit::ACTIVE && LOGL >= 9 && LOGL <= 11 && FFT_R8;
    constexpr bool INGEST = Ingest::ACTIVE && LOGL >= 9 && LOGL <= 11 && FFT_R8;
    constexpr int T = FftThreads<C, LOGL>::T;
    constexpr int NB4 = C / (4 * T), NB2 = C / (2 * T);
    const int tid = threadIdx.x;
    const int logL = LOGL ? LOGL : ax.logL, L = 1 << logL;
    int logNs = 0;
#if FFT_R8 && FFT_W64
    if constexpr (LOGL == 10 && T == 64 && C == 1024) {
        fft1024_wave<Emit::ACTIVE != 0, Emit::TO_LDS != 0, Ingest::ACTIVE != 0>(buf, ax, emit, ingest);
        return;
    }
#endif
#if FFT_R8
    if (LOGL >= 9 && LOGL <= 11) {
        // leading radix-8 passes (Ns = 1, 8[, 64]): C/8 butterflies = one per thread (T = C/8)
        static_assert(LOGL < 9 || LOGL > 11 || C == 8 * T || (LOGL == 10 && T == 64), "radix-8 plan: one butterfly per thread");
        constexpr int NR8 = LOGL == 10 ? 2 : 3;
        constexpr double RH = 0.70710678118654752440;
        const int f = tid >> (logL - 3), j = tid & ((L >> 3) - 1), fb = f * L;
#pragma unroll
        for (int ps = 0; ps < NR8; ++ps) {
            const int lNs = 3 * ps, Ns = 1 << lNs, k = j & (Ns - 1);
            double2 a[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (REGMODE == 2 && ps == 0) a[r] = xr[r];
                else a[r] = (INGEST && ps == 0) ? ingest(fb + j + r * (L >> 3)) : buf[swz<LOGL>(fb + j + r * (L >> 3))];
            }
            if (ps) {                                   // inputs r = 1..7 times w^r, w = exp(-2 pi i k / (8 Ns))
                const double2 w1 = ax.tw[k << (logL - 3 - lNs)];
                double2 w = w1;
                a[1] = cmul(a[1], w);
#pragma unroll
                for (int r = 2; r < 8; ++r) {
                    w = cmul(w, w1);
                    a[r] = cmul(a[r], w);
                }
            }
            // 8-point DFT: b_n = a_n + a_{n+4}; c_n = (a_n - a_{n+4}) W8^n; then two 4-point DFTs (even / odd outputs)
            double2 bb[4], cc[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bb[n] = make_double2(a[n].x + a[n + 4].x, a[n].y + a[n + 4].y);
                cc[n] = make_double2(a[n].x - a[n + 4].x, a[n].y - a[n + 4].y);
            }
            cc[1] = make_double2(RH * (cc[1].x + cc[1].y), RH * (cc[1].y - cc[1].x));      // * (1 - i)/sqrt 2
            cc[2] = make_double2(cc[2].y, -cc[2].x);                                       // * (-i)
            cc[3] = make_double2(RH * (cc[3].y - cc[3].x), -RH * (cc[3].x + cc[3].y));     // * (-1 - i)/sqrt 2
            double2 o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2 *v = h ? cc : bb;
                const double2 t0 = make_double2(v[0].x + v[2].x, v[0].y + v[2].y), t1 = make_double2(v[0].x - v[2].x, v[0].y - v[2].y);
                const double2 t2 = make_double2(v[1].x + v[3].x, v[1].y + v[3].y);
                const double2 t3 = make_double2(v[1].y - v[3].y, -(v[1].x - v[3].x));      // -i (v1 - v3)
                o[h + 0] = make_double2(t0.x + t2.x, t0.y + t2.y);
                o[h + 2] = make_double2(t1.x + t3.x, t1.y + t3.y);
                o[h + 4] = make_double2(t0.x - t2.x, t0.y - t2.y);
                o[h + 6] = make_double2(t1.x - t3.x, t1.y - t3.y);
            }
            const int wb = fb + ((j >> lNs) << (lNs + 3)) + k;
            if (REGMODE == 1 && 3 * (ps + 1) == LOGL) {  // last pass of the 8x8x8 plan, outputs stay with the thread
#pragma unroll
                for (int r = 0; r < 8; ++r) xr[r] = o[r];
            } else if (EMIT && 3 * (ps + 1) == LOGL) {   // last pass of the 8x8x8 plan
                if (Emit::TO_LDS) __syncthreads();
#pragma unroll
                for (int r = 0; r < 8; ++r) emit(wb + r * Ns, o[r]);
                if (Emit::TO_LDS) __syncthreads();
            } else {
                if (!(INGEST && ps == 0)) __syncthreads();          // nobody has read the image yet in an ingesting first pass
#pragma unroll
                for (int r = 0; r < 8; ++r) buf[swz<LOGL>(wb + r * Ns)] = o[r];
                __syncthreads();
            }
        }
        logNs = 3 * NR8;
    }
#endif
    // radix-4 passes: C/4 butterflies = 4 per thread
    const int logQ = logL - 2, Q = L >> 2;
#pragma unroll
    for (; logNs + 2 <= logL; logNs += 2) {
        const int Ns = 1 << logNs;
        double2 v[NB4][4];
        int wbase[NB4];
        // the twiddles depend on k = j mod Ns only; for Ns <= T all four butterflies of a thread
        // share them (j = tid + i T), so they are loaded once
        const bool shared_tw = Ns <= T;
        double2 w1 = make_double2(1.0, 0.0), w2 = w1, w3 = w1;
        if (logNs > 0 && shared_tw) {
            const int ti = (tid & (Ns - 1)) << (logL - logNs - 2);
            w1 = ax.tw[ti];
            w2 = ax.tw[2 * ti];
            w3 = ax.tw[3 * ti];
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int idx = tid + i * T;
            const int f = idx >> logQ, j = idx & (Q - 1), k = j & (Ns - 1);
            const int fb = f * L;
            double2 a = buf[swz<LOGL>(fb + j)], b = buf[swz<LOGL>(fb + j + Q)], c = buf[swz<LOGL>(fb + j + 2 * Q)],
                    d = buf[swz<LOGL>(fb + j + 3 * Q)];
            if (logNs > 0) {
                if (!shared_tw) {
                    const int ti = k << (logL - logNs - 2);
                    w1 = ax.tw[ti];
                    w2 = ax.tw[2 * ti];
                    w3 = ax.tw[3 * ti];
                }
                b = cmul(b, w1);
                c = cmul(c, w2);
                d = cmul(d, w3);
            }
            double2 t0 = make_double2(a.x + c.x, a.y + c.y), t1 = make_double2(a.x - c.x, a.y - c.y);
            double2 t2 = make_double2(b.x + d.x, b.y + d.y);
            double2 t3 = make_double2(b.y - d.y, -(b.x - d.x));            // -i (b - d)
            v[i][0] = make_double2(t0.x + t2.x, t0.y + t2.y);
            v[i][1] = make_double2(t1.x + t3.x, t1.y + t3.y);
            v[i][2] = make_double2(t0.x - t2.x, t0.y - t2.y);
            v[i][3] = make_double2(t1.x - t3.x, t1.y - t3.y);
            wbase[i] = f * L + ((j >> logNs) << (logNs + 2)) + k;
        }
        if (REGMODE == 1 && logNs + 2 == logL) {
            static_assert(REGMODE != 1 || LOGL == 9 || NB4 == 2, "two radix-4 butterflies per thread");
#pragma unroll
            for (int i = 0; i < NB4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[(i + 2 * r) & 7] = v[i][r];
        } else if (EMIT && logNs + 2 == logL) {
            if (Emit::TO_LDS) __syncthreads();
#pragma unroll
            for (int i = 0; i < NB4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) emit(wbase[i] + r * Ns, v[i][r]);
            if (Emit::TO_LDS) __syncthreads();
        } else {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NB4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) buf[swz<LOGL>(wbase[i] + r * Ns)] = v[i][r];
            __syncthreads();
        }
    }
    if (logNs < logL) {          // one radix-2 pass: C/2 butterflies = 8 per thread
        const int Ns = 1 << logNs, H = L >> 1, logH = logL - 1;
        double2 v[NB2][2];
        int wbase[NB2];
#pragma unroll
        for (int i = 0; i < NB2; ++i) {
            const int idx = tid + i * T;
            const int f = idx >> logH, j = idx & (H - 1), k = j & (Ns - 1);
            const int fb = f * L;
            double2 a = buf[swz<LOGL>(fb + j)], b = cmul(buf[swz<LOGL>(fb + j + H)], ax.tw[k << (logL - logNs - 1)]);
            v[i][0] = make_double2(a.x + b.x, a.y + b.y);
            v[i][1] = make_double2(a.x - b.x, a.y - b.y);
            wbase[i] = f * L + ((j >> logNs) << (logNs + 1)) + k;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NB2; ++i) {
            buf[swz<LOGL>(wbase[i])] = v[i][0];
            buf[swz<LOGL>(wbase[i] + Ns)] = v[i][1];
        }
        __syncthreads();
    }
}

// E along the fast axis.  EPI 0: out = scale * E(in);  EPI 3: same + per-workgroup partials of
// sum W (D[slot] - dbar) (other ? other : out) * out  into sp.gpart[b * gridDim.x + blockIdx.x] and of
// sum W (D[slot] - dbar) out * out into sp.gpart2[same];  EPI 4 (last pass of a CG sweep, see k_cg_rows_fwd):
// out = other + scale * E(in) with the partials of sum W (D - dbar) other * out and sum W (D - dbar) out * out;
// EPI 5 (last pass of an adjoint CG sweep, k_adj_rows_fwd): out = other + epi_c (D - dbar) scale E(in), partials with
// the weight W / (D - dbar); there D is the plane sp.Dslot itself (d_slot_stride = 0).
template <int EPI, int C, int LOGL>
__global__ __launch_bounds__((FftThreads<C, LOGL>::T)) void k_dct_rows(Geom G, FftAxis ax, const double *__restrict__ in,
                                                  long in_slot_stride, double *__restrict__ out, double scale,
                                                  SpecArgs sp, const TrajState *__restrict__ st, int gate) {
    const int b = blockIdx.z;
    if (gate && !gate_open(st[b], gate)) return;
    __shared__ double2 buf[FftLds<C>::SIZE];
    constexpr int T = FftThreads<C, LOGL>::T;
    const int tid = threadIdx.x;
    const int logL = LOGL ? LOGL : ax.logL, L = 1 << logL, N = L >> 1, nfft = C >> logL;
    const int row0 = blockIdx.x * 2 * nfft;
    const double *ib = in + b * G.plane + (in_slot_stride ? st[b].slot * in_slot_stride : 0);
    const int n1 = N + 1;
    const float inv_n1 = 1.0f / (float)n1;       // idx < 2^14: the float quotient is exact enough
    FFT_STAMP(0);
    constexpr bool DIRECT = FFT_R8 && LOGL >= 9 && LOGL <= 11;
    if (!(DIRECT && FFT_ROWS_INGEST)) {
        for (int idx = tid; idx < nfft * n1; idx += T) {
            const int f = nfft == 1 ? 0 : (int)(((float)idx + 0.5f) * inv_n1), j = idx - f * n1;
            const int ra = row0 + 2 * f, rb = ra + 1;
            double2 v = make_double2(ra < G.ns ? ib[(long)ra * G.pitch + j] : 0.0,
                                     rb < G.ns ? ib[(long)rb * G.pitch + j] : 0.0);
            buf[swz<LOGL>(f * L + j)] = v;
            if (j > 0 && j < N) buf[swz<LOGL>(f * L + L - j)] = v;
        }
        __syncthreads();
    }
    FFT_STAMP(1);
    double dot = 0.0, dot2 = 0.0, dbar = 0.0;
    const double *Dp = nullptr, *Ob = nullptr;
    int scaled = 0;
    if (EPI >= 3) {
        dbar = st[b].dbar;
        scaled = (EPI != 5) ? st[b].scaled : 0;
        Dp = sp.Dslot + st[b].slot * sp.d_slot_stride + b * G.plane;
        Ob = sp.other ? sp.other + b * G.plane : nullptr;
    }
    double *ob = out + b * G.plane;
    auto put = [&](int row, int k, double e) {
        double v = scale * e;
        const long o = (long)row * G.pitch + k;
        if (EPI == 4) v += Ob[o];
        if (EPI == 5) {
            const double dl = Dp[o] - dbar, pv = Ob[o];
            v = pv + sp.epi_c * dl * v;
            ob[o] = v;
            const double wd = wdev(row, k, G) / dl;
            dot += wd * (pv * v);
            dot2 += wd * (v * v);
            return;
        }
        ob[o] = v;
        if (EPI >= 3) {
            const double wd = wdev(row, k, G) * cg_weight(Dp[o], dbar, scaled);
            dot += wd * ((Ob ? Ob[o] : v) * v);
            dot2 += wd * (v * v);
        }
    };
    if (DIRECT) {
        // the last pass hands its outputs over in registers: entry k <= N of transform f is (E(a)_k, E(b)_k) of rows
        // row0 + 2f, row0 + 2f + 1 (the upper half of the spectrum is its mirror image and is dropped)
        struct RowEmit {
            enum { ACTIVE = 1, TO_LDS = 0 };
            decltype(put) &put_;
            int row0_, ns_;
            __device__ __forceinline__ void operator()(int idx, double2 v) const {
                constexpr int LL = 1 << (LOGL ? LOGL : 1);
                const int f = idx >> (LOGL ? LOGL : 1), k = idx & (LL - 1);
                if (k <= LL / 2) {
                    const int ra = row0_ + 2 * f;
                    if (ra < ns_) put_(ra, k, v.x);
                    if (ra + 1 < ns_) put_(ra + 1, k, v.y);
                }
            }
        };
        // ... and the first pass takes its inputs (entry i of the even extension of rows row0 + 2f, + 1) from global memory
        struct RowIngest {
            enum { ACTIVE = 1 };
            const double *ib_;
            long pitch_;
            int row0_, ns_;
            __device__ __forceinline__ double2 operator()(int idx) const {
                constexpr int LL = 1 << (LOGL ? LOGL : 1);
                const int f = idx >> (LOGL ? LOGL : 1), i = idx & (LL - 1), m = i <= LL / 2 ? i : LL - i;
                const int ra = row0_ + 2 * f;
                const double *p = ib_ + (long)ra * pitch_ + m;
                return make_double2(ra < ns_ ? p[0] : 0.0, ra + 1 < ns_ ? p[pitch_] : 0.0);
            }
        };
        if (FFT_ROWS_INGEST) fft_lds<C, LOGL>(buf, ax, RowEmit{put, row0, G.ns}, RowIngest{ib, (long)G.pitch, row0, G.ns});
        else fft_lds<C, LOGL>(buf, ax, RowEmit{put, row0, G.ns});
        FFT_STAMP(2);
    } else {
        fft_lds<C, LOGL>(buf, ax);
        FFT_STAMP(2);
        for (int idx = tid; idx < 2 * nfft * n1; idx += T) {
            const int rr = nfft == 1 ? (idx >= n1 ? 1 : 0) : (int)(((float)idx + 0.5f) * inv_n1), k = idx - rr * n1;
            const int row = row0 + rr;
            if (row < G.ns) {
                const double2 c = buf[swz<LOGL>((rr >> 1) * L + k)];
                put(row, k, (rr & 1) ? c.y : c.x);
            }
        }
    }
    FFT_STAMP(3);
    if (EPI >= 3) {
        dot = wave_sum(dot);
        dot2 = wave_sum(dot2);
        __syncthreads();
        double *sred = reinterpret_cast<double *>(buf);
        if ((tid & 63) == 0) {
            sred[tid >> 6] = dot;
            sred[T / 64 + (tid >> 6)] = dot2;
        }
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0, tot2 = 0.0;
            for (int w = 0; w < T / 64; ++w) {
                tot += sred[w];
                tot2 += sred[T / 64 + w];
            }
            sp.gpart[(long)b * gridDim.x + blockIdx.x] = tot;
            sp.gpart2[(long)b * gridDim.x + blockIdx.x] = tot2;
        }
    }
}

// =====================================================================================================
// First pass of a forward CG sweep (power-of-two grids).  A = P + M Delta (Delta = D - dbar, P the constant-
// coefficient operator the DCT inverts), so the preconditioned operator needs no stencil at all:
//       q = P^-1 A p = p + (P^-1 M)(Delta p)
// i.e. a point-wise multiply, the transform pair with the spectral multiplier m / P(m), and an add.  A sweep is
//   [k_cg_rows_fwd]  resolve the reduction point of the previous sweep (every workgroup, redundantly, from the partial
//                    sums; workgroup 0 of a trajectory records it), then on the way into the row transform
//                    x += alpha p_old;  z' = z - alpha q;  p' = z' + beta p_old;  partial of <z',z'>_Z;  E_rows(Delta p')
//   [k_dct_cols]     E_cols, multiplier m / P(m), E_cols
//   [k_dct_rows<4>]  E_rows, q' = p' + (.), partials of <p',q'>_Z and <q',q'>_Z
// three launches, 120 B per node, and the image A p' of the 13-point operator never exists.  (The GEMM-DCT path of
// non-power-of-two grids keeps the stencil form, k_schur_p.)
// =====================================================================================================
struct CgSweepArgs {
    const double *z, *q, *p_old;          // [B][plane]
    double *x, *z_new, *p_new;
    const double *Dslot;                  // slot-indexed D planes
    long d_slot_stride;
    const double *gpart, *gpart2;         // [B][gnblk] partials of <p,q>_Z, <q,q>_Z from the previous sweep's last pass
    double *gpart3;                       // [2][B][gnblk] partials of <z',z'>_Z, copy = sweep parity
    int it, maxit, nbatch;
    const double *x0;                     // FIRST: starting guess of a primed solve (TrajState::x_primed), else unused
};

// the three sums of a reduction point by the waves of the workgroup (fixed order); s3 = LDS [3]
template <int T>
__device__ __forceinline__ void cg_sums3(const double *__restrict__ g1, const double *__restrict__ g2,
                                         const double *__restrict__ g3, int n, int b, double *s3) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wv; k < 3; k += T / 64) {
        const double *src = k == 0 ? g1 : (k == 1 ? g2 : g3);
        double a = 0.0;
        if (src)
            for (int t = lane; t < n; t += 64) a += src[(long)b * n + t];
        a = wave_sum(a);
        if (lane == 0) s3[k] = a;
    }
    __syncthreads();
}

// End of a stencil-free forward solve, fused with the back substitution (cf. k_dmu_ceiling): every workgroup resolves the
// reduction point of the last enqueued sweep (copy rd -> rd ^ 1, recorded by workgroup 0; lin_active itself is taken
// over by k_fin_ceiling), forms dphi = x + alpha p_last on its haloed tile, stores it to x_out, and goes on with
// dmu = 2 (K dphi + R_phi) and the step-ceiling ratio.
struct FinSolveArgs {
    const double *gpart, *gpart2, *gpart3_rd;
    int gnblk, direct, rd, maxit;
    const double *p_last;                 // the direction of the last sweep
    int cheb_enq;                         // >= 0: after a Chebyshev solve with that many sweeps enqueued (x is final: no
                                          // reduction point to resolve; a trajectory whose plan asked for more has no x yet)
};
__global__ __launch_bounds__(NTH) void k_dmu_ceiling_fin(Geom G, Phys P, TrajState *__restrict__ st, long slot_stride,
                                                         const double *__restrict__ x, FinSolveArgs f,
                                                         const double *__restrict__ phi_s, const double *__restrict__ D_s,
                                                         const double *__restrict__ Rphi_s, double *__restrict__ dmu,
                                                         double *__restrict__ x_out, double *__restrict__ part,
                                                         double *__restrict__ x_keep, double *__restrict__ x_keep2) {
    TILE_COORDS;
    if (!st[b].newton_active || st[b].need_trial) return;
    if (st[b].use_cheb) return;           // solved in the reduction-free form: its last row kernel did this kernel's work
    __shared__ double sx[(TY + 2) * (TX + 2)];
    __shared__ double sred[NPART * 4];
    __shared__ double s3[3];
    // the step's first / second Newton increment, for the next steps' k_guess
    double *const keepp = st[b].iters == 1 ? x_keep : (st[b].iters == 2 ? x_keep2 : nullptr);
    const bool keep = keepp != nullptr;
    constexpr int W = TX + 2;
    const long pb = b * G.plane;
    const int slot = st[b].slot;
    double alpha = 0.0;
    if (f.cheb_enq >= 0) {
        if (st[b].lin_active && st[b].cheb_n > f.cheb_enq) return;
    } else if (st[b].lin_active) {
        if (!st[b].ci_active[f.rd]) {                      // converged in an earlier sweep: its last step is in x already
            if (blk == 0 && threadIdx.x == 0) st[b].ci_active[f.rd ^ 1] = 0;
        } else {
            const double gamma0 = st[b].cg_gamma0, gamma_old = st[b].ci_gamma[f.rd], tol = st[b].lin_reltol;
            const int it_old = st[b].ci_it[f.rd];
            cg_sums3<NTH>(f.gpart, f.gpart2, f.direct ? f.gpart3_rd : nullptr, f.gnblk, b, s3);
            const CgNext n = cg_next(s3[0], s3[1], f.direct ? s3[2] : gamma_old, gamma0, it_old, tol, f.maxit);
            if (blk == 0 && threadIdx.x == 0) cg_record(st[b], n, f.rd ^ 1);
            alpha = n.breakdown ? 0.0 : n.alpha;
        }
    }
    if (alpha != 0.0) load_tile_axpy<1>(sx, x + pb, f.p_last + pb, alpha, G, c0, r0);
    else load_tile<1>(sx, x + pb, G, c0, r0);
    if (st[b].scaled) {                   // right-scaled CG form: the iterate was y, dphi = (dbar / D) y (halo included)
        __syncthreads();
        const double dbar = st[b].dbar;
        const double *Dp = D_s + slot * slot_stride + pb;
        for (int e = threadIdx.x; e < W * (TY + 2); e += NTH) {
            int ly = e / W, lxx = e - ly * W;
            int gr = refl(r0 - 1 + ly, G.ns), gc = refl(c0 - 1 + lxx, G.nf);
            sx[e] *= dbar / Dp[(long)gr * G.pitch + gc];
        }
    }
    __syncthreads();
    double acc[1] = {1e300};
    for (int k = 0; k < TY / 4; ++k) {
        int ly = ly0 + 4 * k, r = r0 + ly, c = c0 + lx;
        if (r < G.ns && c < G.nf) {
            int p = (ly + 1) * W + lx + 1;
            long o = pb + (long)r * G.pitch + c, os = slot * slot_stride + o;
            double d = sx[p];
            x_out[o] = d;
            if (keep) keepp[o] = d;
            dmu[o] = 2.0 * ((-0.5 * P.kappa * lap_at<W>(sx, p, G.ax, G.ay) + D_s[os] * d) + Rphi_s[os]);
            double ph = phi_s[os];
            if (d > 0.0) acc[0] = fmin(acc[0], (1.0 - DELTA_SEP - ph) / d);
            else if (d < 0.0) acc[0] = fmin(acc[0], (-1.0 + DELTA_SEP - ph) / d);
        }
    }
    const int op[1] = {1};
    block_reduce_store<1>(acc, op, sred, part + ((long)b * nblk + blk) * NPART);
}

template <int FIRST, int C, int LOGL>
__global__ __launch_bounds__((FftThreads<C, LOGL>::T)) void k_cg_rows_fwd(Geom G, FftAxis ax, CgSweepArgs a, double *__restrict__ out,
                                                                         TrajState *__restrict__ st) {
    const int b = blockIdx.z;
    __shared__ double2 buf[FftLds<C>::SIZE];
    __shared__ double s3[4];
    constexpr int T = FftThreads<C, LOGL>::T;
    const int tid = threadIdx.x;
    const int logL = LOGL ? LOGL : ax.logL, L = 1 << logL, N = L >> 1, nfft = C >> logL;
    const int row0 = blockIdx.x * 2 * nfft, n1 = N + 1;
    const int rd = (a.it + 1) & 1, wr = a.it & 1;
    const long pb = b * G.plane;
    double alpha = 0.0, beta = 0.0;
    if (FIRST) {
        // start of a solve (what k_fin_cg_init does on the GEMM path): workgroup 0 of the trajectory sums
        // gamma0 = <z0,z0>_Z from the partials of the transform that produced z0 and arms the per-sweep state
        const int active = st[b].lin_active && !st[b].use_cheb;
        if (blockIdx.x == 0 && tid < 64) {
            double g = 0.0;
            if (active)
                for (int t = tid; t < (int)gridDim.x; t += 64) g += a.gpart[(long)b * gridDim.x + t];
            g = wave_sum(g);
            if (tid == 0) {
                TrajState &S = st[b];
                S.cg_pending = 0;
                if (!active) {
                    S.ci_active[0] = 0;
                } else {
                    S.cg_gamma = S.cg_gamma0 = g;
                    S.cg_beta = 0.0;
                    S.lin_it = 0;
                    S.lin_rel = 1.0;
                    S.ci_active[0] = 1;
                    S.ci_it[0] = 0;
                    S.ci_gamma[0] = g;
                }
            }
        }
        if (!active) return;
    } else {
        if (!st[b].ci_active[rd]) {
            if (blockIdx.x == 0 && tid == 0) {             // hand the (finished) state on to the other copy
                st[b].ci_active[wr] = 0;
                st[b].ci_it[wr] = st[b].ci_it[rd];
                st[b].ci_gamma[wr] = st[b].ci_gamma[rd];
            }
            return;
        }
        const double gamma0 = st[b].cg_gamma0, gamma_old = st[b].ci_gamma[rd], tol = st[b].lin_reltol;
        const int it_old = st[b].ci_it[rd], n = gridDim.x;
        cg_sums3<T>(a.gpart, a.gpart2, a.it >= 2 ? a.gpart3 + (long)rd * a.nbatch * n : nullptr, n, b, s3);
        const CgNext nx = cg_next(s3[0], s3[1], a.it >= 2 ? s3[2] : gamma_old, gamma0, it_old, tol, a.maxit);
        if (blockIdx.x == 0 && tid == 0) cg_record(st[b], nx, wr);
        alpha = nx.alpha;
        beta = nx.beta;
        if (!nx.active) {                                  // converged (or broke down, alpha = 0): take the step only
            if (!nx.breakdown)
                for (int idx = tid; idx < 2 * nfft * n1; idx += T) {
                    const int rr = idx / n1, k = idx - rr * n1, row = row0 + rr;
                    if (row < G.ns) {
                        const long o = pb + (long)row * G.pitch + k;
                        a.x[o] += alpha * a.p_old[o];
                    }
                }
            return;
        }
    }
    const double dbar = st[b].dbar;
    const int scaled = st[b].scaled;
    const double *Dp = a.Dslot + st[b].slot * a.d_slot_stride + pb;
    const bool primed = FIRST && a.x0 && st[b].x_primed;
    double acc = 0.0;
    // value fed into the transform at node (row, m); the OWNER visit (each node exactly once) also applies the step
    auto node = [&](int row, int m, bool owner) -> double {
        if (row >= G.ns) return 0.0;
        const long o = (long)row * G.pitch + m;
        const double dl = cg_weight(Dp[o], dbar, scaled);
        double pn;
        if (FIRST) {
            pn = a.z[pb + o];
            if (owner) {
                // the iterate lives in y = S^-1 x on the right-scaled system: the starting guess x0 becomes (D / dbar) x0
                a.x[pb + o] = primed ? (scaled ? a.x0[pb + o] * (Dp[o] / dbar) : a.x0[pb + o]) : 0.0;
                a.p_new[pb + o] = pn;
            }
        } else {
            const double po = a.p_old[pb + o];
            const double zn = a.z[pb + o] - alpha * a.q[pb + o];
            pn = zn + beta * po;
            if (owner) {
                a.x[pb + o] += alpha * po;
                a.z_new[pb + o] = zn;
                a.p_new[pb + o] = pn;
                acc += wdev(row, m, G) * dl * (zn * zn);
            }
        }
        return dl * pn;
    };
    double *ob = out + pb;
    auto put = [&](int row, int k, double e) { ob[(long)row * G.pitch + k] = e; };
    constexpr bool DIRECT = FFT_R8 && LOGL >= 9 && LOGL <= 11;
    if (DIRECT) {
        struct Emit {
            enum { ACTIVE = 1, TO_LDS = 0 };
            decltype(put) &put_;
            int row0_, ns_;
            __device__ __forceinline__ void operator()(int idx, double2 v) const {
                constexpr int LL = 1 << (LOGL ? LOGL : 1);
                const int f = idx >> (LOGL ? LOGL : 1), k = idx & (LL - 1);
                if (k <= LL / 2) {
                    const int ra = row0_ + 2 * f;
                    if (ra < ns_) put_(ra, k, v.x);
                    if (ra + 1 < ns_) put_(ra + 1, k, v.y);
                }
            }
        };
        struct Ingest {
            enum { ACTIVE = 1 };
            decltype(node) &node_;
            int row0_;
            __device__ __forceinline__ double2 operator()(int idx) const {
                constexpr int LL = 1 << (LOGL ? LOGL : 1);
                const int f = idx >> (LOGL ? LOGL : 1), i = idx & (LL - 1);
                const bool owner = i <= LL / 2;
                const int m = owner ? i : LL - i, ra = row0_ + 2 * f;
                return make_double2(node_(ra, m, owner), node_(ra + 1, m, owner));
            }
        };
#if CG_ROWS_STAGED
        // every node once (no second visit for the mirror half of the even extension): stage the image, then transform
        for (int idx = tid; idx < nfft * n1; idx += T) {
            const int f = nfft == 1 ? 0 : idx / n1, j = idx - f * n1;
            const int ra = row0 + 2 * f;
            const double2 v = make_double2(node(ra, j, true), node(ra + 1, j, true));
            buf[swz<LOGL>(f * L + j)] = v;
            if (j > 0 && j < N) buf[swz<LOGL>(f * L + L - j)] = v;
        }
        __syncthreads();
        fft_lds<C, LOGL>(buf, ax, Emit{put, row0, G.ns});
#else
        fft_lds<C, LOGL>(buf, ax, Emit{put, row0, G.ns}, Ingest{node, row0});
#endif
    } else {
        const float inv_n1 = 1.0f / (float)n1;
        for (int idx = tid; idx < nfft * n1; idx += T) {
            const int f = nfft == 1 ? 0 : (int)(((float)idx + 0.5f) * inv_n1), j = idx - f * n1;
            const int ra = row0 + 2 * f;
            const double2 v = make_double2(node(ra, j, true), node(ra + 1, j, true));
            buf[swz<LOGL>(f * L + j)] = v;
            if (j > 0 && j < N) buf[swz<LOGL>(f * L + L - j)] = v;
        }
        __syncthreads();
        fft_lds<C, LOGL>(buf, ax);
        for (int idx = tid; idx < 2 * nfft * n1; idx += T) {
            const int rr = nfft == 1 ? (idx >= n1 ? 1 : 0) : (int)(((float)idx + 0.5f) * inv_n1), k = idx - rr * n1;
            const int row = row0 + rr;
            if (row < G.ns) {
                const double2 c = buf[swz<LOGL>((rr >> 1) * L + k)];
                put(row, k, (rr & 1) ? c.y : c.x);
            }
        }
    }
    if (!FIRST) {
        acc = wave_sum(acc);
        __syncthreads();
        double *sred = reinterpret_cast<double *>(buf);
        if ((tid & 63) == 0) sred[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < T / 64; ++w) tot += sred[w];
            a.gpart3[((long)wr * a.nbatch + b) * gridDim.x + blockIdx.x] = tot;
        }
    }
}

// =====================================================================================================
// Reduction-free (Chebyshev) forward solve, power-of-two grids.  spec(P^-1 A) lies in [1, kT] with kT known before the
// solve (cg_setup), so the Chebyshev semi-iteration needs no inner product: its coefficients and its length follow from
// the bound (cheb_plan).  In the three-term form on y (x = x0 + y, b~ = P^-1 rhs, y_0 = 0):
//       y_1 = b~ / theta,      z_j = b~ - y_j - (P^-1 M)(Delta y_j),      y_{j+1} = y_j + c_j (y_j - y_{j-1}) + e_j z_j
// With nothing to wait for between the inverse row transform of sweep j and the forward row transform of sweep j + 1,
// the two are ONE kernel on the same rows (two in-LDS transforms back to back, as in k_dct_cols):
//   [k_dct_rows<0>]   E_rows(rhs)                                    }  once per solve
//   [k_dct_cols]      E_cols, multiplier 1 / P(m), E_cols            }
//   [k_cheb_rows j=0] b~ = E_rows(.), y_1 = b~ / theta, E_rows(Delta y_1)
//   [k_dct_cols]      E_cols, multiplier m / P(m), E_cols            }  per sweep: 2 launches, 72 B per node
//   [k_cheb_rows j]   g = E_rows(.), z_j, y_{j+1}, E_rows(Delta y_{j+1})  }  (CG form: 3 launches, 120 B)
// A trajectory's LAST kernel (j = its cheb_n) stores x = x0 + y_{j+1} and transforms nothing; later kernels of the
// launch sequence skip it.  Each kernel leaves the partial of <z_j,z_j>_Z behind (one sweep late, for the books only).
// =====================================================================================================
struct ChebSweepArgs {
    const double *bt_in;                  // b~ [B][plane], read by the sweeps j >= 1
    double *bt_out;                       // ... written by sweep 0
    const double *y_cur, *y_prev;         // y_j, y_{j-1} (j >= 2)
    double *y_new;                        // y_{j+1}: may be the buffer of y_{j-1} (same node, read before written)
    double *x_out;                        // the finished increment of a trajectory's last sweep
    const double *x0;                     // starting guess of a primed solve (TrajState::x_primed) or NULL
    const double *Dslot;                  // slot-indexed D planes
    long d_slot_stride;
    double *gpart;                        // [B][gridDim.x] partials of <z_j,z_j>_Z
    int j;
    double scale;                         // 1 / (4 Nf Ns) of the transform pair
    // a trajectory's last sweep also takes the step ceiling (F2:381-387: min over the nodes of (+-(1 - delta) - phi) / dphi,
    // per-workgroup minima into cmin) and keeps the step's first / second Newton increment for the next steps' guesses
    const double *phi_s;                  // slot-indexed iterate
    double *cmin;                         // [B][gridDim.x]
    double *keep1, *keep2;                // ring planes of this step (or NULL)
};

// FIRST = 1: the kernel with j = 0 (a separate instantiation: its branches fold, and profiles tell the two apart)
#ifndef CHEB_MINW
#define CHEB_MINW 1            // minimum waves per SIMD the register allocation of k_cheb_rows must allow (A/B knob)
#endif
template <int C, int LOGL, int FIRST>
__global__ __launch_bounds__((FftThreads<C, LOGL>::T), CHEB_MINW) void k_cheb_rows(Geom G, FftAxis ax, ChebSweepArgs a_, const double *__restrict__ in,
                                                                       double *__restrict__ out, const TrajState *__restrict__ st) {
    const int b = blockIdx.z;
    ChebSweepArgs a = a_;
    if (FIRST) a.j = 0;
    if (!st[b].lin_active || !st[b].use_cheb || a.j > st[b].cheb_n) return;
    const bool last = a.j == st[b].cheb_n;
    __shared__ double2 buf[FftLds<C>::SIZE];
    constexpr int T = FftThreads<C, LOGL>::T;
    const int tid = threadIdx.x;
    const int logL = LOGL ? LOGL : ax.logL, L = 1 << logL, N = L >> 1, nfft = C >> logL;
    const int row0 = blockIdx.x * 2 * nfft, n1 = N + 1;
    const long pb = b * G.plane;
    // coefficients of this sweep: e_j = 2 rho_j / delta = 2 / (2 theta - delta rho_{j-1}), c_j = rho_j rho_{j-1}
    const double theta = st[b].cheb_theta, delta = st[b].cheb_delta;
    double cj = 0.0, ej = 1.0 / theta;
    {
        double rho_prev = delta / theta;
        for (int i = 1; i <= a.j; ++i) {
            ej = 2.0 / (2.0 * theta - delta * rho_prev);
            const double rho = 0.5 * delta * ej;
            cj = rho * rho_prev;
            rho_prev = rho;
        }
    }
    const double dbar = st[b].dbar;
    const double *Dp = a.Dslot + st[b].slot * a.d_slot_stride + pb;
    const bool primed = a.x0 && st[b].x_primed;
    const bool need_norm = !(last && a.j == 0);            // a solve without sweeps keeps no books
    double acc = 0.0, cmin = 1e300;
    const double *php = a.phi_s + st[b].slot * a.d_slot_stride + pb;
    double *const keepp = st[b].iters == 1 ? a.keep1 : (st[b].iters == 2 ? a.keep2 : nullptr);
    // node (row, k): g = the transform's output there; returns Delta y_{j+1}, the next transform's input
    auto upd = [&](int row, int k, double g) -> double {
        const long o = (long)row * G.pitch + k;
        const double gg = a.scale * g;
        double z, yn, dl = 0.0;
        if (need_norm || !last) dl = Dp[o] - dbar;
        if (a.j == 0) {
            z = gg;
            yn = ej * z;
            if (!last) a.bt_out[pb + o] = gg;
        } else {
            const double yc = a.y_cur[pb + o], yp = a.j >= 2 ? a.y_prev[pb + o] : 0.0;
            z = a.bt_in[pb + o] - yc - gg;
            yn = yc + cj * (yc - yp) + ej * z;
        }
        if (need_norm) acc += wdev(row, k, G) * dl * (z * z);
        if (last) {
            const double d = primed ? a.x0[pb + o] + yn : yn, ph = php[o];
            a.x_out[pb + o] = d;
            if (keepp) keepp[pb + o] = d;
            if (d > 0.0) cmin = fmin(cmin, (1.0 - DELTA_SEP - ph) / d);
            else if (d < 0.0) cmin = fmin(cmin, (-1.0 + DELTA_SEP - ph) / d);
        } else {
            a.y_new[pb + o] = yn;
        }
        return dl * yn;
    };
    const double *ib = in + pb;
    double *ob = out + pb;
    auto put = [&](int row, int k, double e) { ob[(long)row * G.pitch + k] = e; };
    constexpr bool DIRECT = FFT_R8 && LOGL >= 9 && LOGL <= 11;
    if (DIRECT) {
        constexpr int LL = 1 << (LOGL ? LOGL : 1), LG = LOGL ? LOGL : 1;
        struct RowIngest {
            enum { ACTIVE = 1 };
            const double *ib_;
            long pitch_;
            int row0_, ns_;
            __device__ __forceinline__ double2 operator()(int idx) const {
                const int f = idx >> LG, i = idx & (LL - 1), m = i <= LL / 2 ? i : LL - i;
                const int ra = row0_ + 2 * f;
                const double *p = ib_ + (long)ra * pitch_ + m;
                return make_double2(ra < ns_ ? p[0] : 0.0, ra + 1 < ns_ ? p[pitch_] : 0.0);
            }
        };
        // the first transform's last pass hands its outputs to the update; what the update returns goes back into the
        // image as the even extension of the next transform's input (entry k and its mirror L - k from the same thread)
        struct UpdEmit {
            enum { ACTIVE = 1, TO_LDS = 1 };
            decltype(upd) &upd_;
            double2 *buf_;
            int row0_, ns_;
            bool last_;
            __device__ __forceinline__ void operator()(int idx, double2 v) const {
                const int f = idx >> LG, k = idx & (LL - 1);
                if (k > LL / 2) return;
                const int ra = row0_ + 2 * f;
                double2 w = make_double2(0.0, 0.0);
                if (ra < ns_) w.x = upd_(ra, k, v.x);
                if (ra + 1 < ns_) w.y = upd_(ra + 1, k, v.y);
                if (!last_) {
                    buf_[swz<LOGL>(idx)] = w;
                    if (k > 0 && k < LL / 2) buf_[swz<LOGL>(f * LL + LL - k)] = w;
                }
            }
        };
        struct RowEmit {
            enum { ACTIVE = 1, TO_LDS = 0 };
            decltype(put) &put_;
            int row0_, ns_;
            __device__ __forceinline__ void operator()(int idx, double2 v) const {
                const int f = idx >> LG, k = idx & (LL - 1);
                if (k <= LL / 2) {
                    const int ra = row0_ + 2 * f;
                    if (ra < ns_) put_(ra, k, v.x);
                    if (ra + 1 < ns_) put_(ra + 1, k, v.y);
                }
            }
        };
        fft_lds<C, LOGL>(buf, ax, UpdEmit{upd, buf, row0, G.ns, last}, RowIngest{ib, (long)G.pitch, row0, G.ns});
        if (!last) fft_lds<C, LOGL>(buf, ax, RowEmit{put, row0, G.ns});
    } else {
        const float inv_n1 = 1.0f / (float)n1;
        for (int idx = tid; idx < nfft * n1; idx += T) {
            const int f = nfft == 1 ? 0 : (int)(((float)idx + 0.5f) * inv_n1), j = idx - f * n1;
            const int ra = row0 + 2 * f;
            const double2 v = make_double2(ra < G.ns ? ib[(long)ra * G.pitch + j] : 0.0,
                                           ra + 1 < G.ns ? ib[(long)(ra + 1) * G.pitch + j] : 0.0);
            buf[swz<LOGL>(f * L + j)] = v;
            if (j > 0 && j < N) buf[swz<LOGL>(f * L + L - j)] = v;
        }
        __syncthreads();
        fft_lds<C, LOGL>(buf, ax);
        // entry k <= N of a transform is read and rewritten by ONE thread, the entries above N are only written: no barrier
        // between the reads and the writes of this loop
        for (int idx = tid; idx < nfft * n1; idx += T) {
            const int f = nfft == 1 ? 0 : (int)(((float)idx + 0.5f) * inv_n1), k = idx - f * n1;
            const int ra = row0 + 2 * f;
            const double2 v = buf[swz<LOGL>(f * L + k)];
            double2 w = make_double2(0.0, 0.0);
            if (ra < G.ns) w.x = upd(ra, k, v.x);
            if (ra + 1 < G.ns) w.y = upd(ra + 1, k, v.y);
            if (!last) {
                buf[swz<LOGL>(f * L + k)] = w;
                if (k > 0 && k < N) buf[swz<LOGL>(f * L + L - k)] = w;
            }
        }
        if (!last) {
            __syncthreads();
            fft_lds<C, LOGL>(buf, ax);
            for (int idx = tid; idx < 2 * nfft * n1; idx += T) {
                const int rr = nfft == 1 ? (idx >= n1 ? 1 : 0) : (int)(((float)idx + 0.5f) * inv_n1), k = idx - rr * n1;
                const int row = row0 + rr;
                if (row < G.ns) {
                    const double2 c = buf[swz<LOGL>((rr >> 1) * L + k)];
                    put(row, k, (rr & 1) ? c.y : c.x);
                }
            }
        }
    }
    if (need_norm || last) {
        acc = wave_sum(acc);
        cmin = wave_min(cmin);
        __syncthreads();
        double *sred = reinterpret_cast<double *>(buf);
        if ((tid & 63) == 0) {
            sred[tid >> 6] = acc;
            sred[T / 64 + (tid >> 6)] = cmin;
        }
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0, mn = 1e300;
            for (int w = 0; w < T / 64; ++w) {
                tot += sred[w];
                mn = fmin(mn, sred[T / 64 + w]);
            }
            if (need_norm) a.gpart[(long)b * gridDim.x + blockIdx.x] = tot;
            if (last) a.cmin[(long)b * gridDim.x + blockIdx.x] = mn;
        }
    }
}

// =====================================================================================================
// Adjoint CG sweep in the same stencil-free form.  A(phi_n) = P + c Delta M (c = dt/2, Delta = D_n - dbar,
// P = I + (tau + c dbar) M + c M^2), right preconditioning: with x = x0 + P^-1 y the operator on y is
//       A P^-1 = I + c Delta (M P^-1),
// self-adjoint and positive in <a,b>_Z' = sum W a b / Delta.  CG on y with ONE reduction point per sweep:
//   [k_adj_rows_fwd]  resolve the previous sweep's reduction point (alpha = gamma / <ph,q>, predicted
//                     gamma' = alpha^2 <q,q> - gamma, beta = gamma'/gamma, stop test), then on the way into the row
//                     transform  y += alpha ph_old;  r' = r - alpha q;  ph' = r' + beta ph_old;  partials of
//                     <r',r'>_Z' and ||r'||_2^2;  E_rows(ph')
//   [k_dct_cols]      E_cols, multiplier m / P(m), E_cols
//   [k_dct_rows<5>]   E_rows, q' = ph' + c Delta (.), partials of <ph',q'>_Z' and <q',q'>_Z'
// and after the last sweep x = x0 + P^-1 y (one more transform pair, k_dct_rows<4> adds x0).  The stop test is the
// reference's accuracy class, ||r||_2 <= tol ||rhs||_2, evaluated as the last directly summed ||r||_2 times the
// predicted decrease of the Z' norm.
// =====================================================================================================
struct AdjSweepArgs {
    const double *r, *q, *p_old;          // [B][plane]
    double *y, *r_new, *p_new;
    const double *Dn;                     // f''(phi_n) plane
    const double *gpart, *gpart2;         // [B][gnblk] partials of <ph,q>_Z', <q,q>_Z' of the previous sweep
    double *gpart3;                       // [2][B][gnblk][2] partials of <r',r'>_Z' and ||r'||^2, copy = sweep parity
    int it, maxit, nbatch;
    const double *part;                   // first sweep: partials {<r0,r0>_Z', ||r0||^2} of k_adj_op<1>, stencil-tile layout
    int nblk;
    double tol;
};

struct CgNextAdj {
    int active, breakdown, it;
    double alpha, beta, gamma, rel;
};
__device__ __forceinline__ CgNextAdj cg_next_adj(double pq, double qq, double gamma, double r2, double rhs_norm, int it_old,
                                                 double tol, int maxit) {
    CgNextAdj n;
    n.it = it_old;
    const double rel_now = rhs_norm > 0.0 ? sqrt(fmax(r2, 0.0)) / rhs_norm : 0.0;
    if (!(pq > 0.0) || !(gamma > 0.0)) {     // round-off level residual: stop here, no step
        n.active = 0; n.breakdown = 1; n.alpha = 0.0; n.beta = 0.0; n.gamma = fmax(gamma, 0.0); n.rel = rel_now;
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
    n.rel = sqrt(gn / gamma) * rel_now;
    n.active = (n.rel > tol && n.it < maxit) ? 1 : 0;
    return n;
}
__device__ __forceinline__ void adj_record(TrajState &S, const CgNextAdj &n, int wr) {
    S.cg_alpha = n.alpha;
    S.cg_beta = n.beta;
    S.cg_gamma = n.gamma;
    S.lin_rel = n.rel;
    if (!n.breakdown) {
        S.lin_it = n.it;
        S.lin_total++;
    }
    if (!n.active && n.rel > S.lin_maxrel) S.lin_maxrel = n.rel;
    S.ci_active[wr] = n.active;
    S.ci_it[wr] = n.it;
    S.ci_gamma[wr] = n.gamma;
}

// the four sums of an adjoint reduction point (fixed order); g3 holds pairs {Z' norm^2, 2-norm^2}; s4 = LDS [4]
template <int T>
__device__ __forceinline__ void adj_sums4(const double *__restrict__ g1, const double *__restrict__ g2,
                                          const double *__restrict__ g3, int n, int b, double *s4) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wv; k < 4; k += T / 64) {
        double a = 0.0;
        if (k < 2) {
            const double *src = k == 0 ? g1 : g2;
            for (int t = lane; t < n; t += 64) a += src[(long)b * n + t];
        } else if (g3) {
            for (int t = lane; t < n; t += 64) a += g3[((long)b * n + t) * 2 + (k - 2)];
        }
        a = wave_sum(a);
        if (lane == 0) s4[k] = a;
    }
    __syncthreads();
}

// The reduction point of the LAST enqueued adjoint sweep; leaves the step pending for k_cg_finish (on y).
__global__ void k_fin_adj_step(TrajState *st, const double *__restrict__ gpart, const double *__restrict__ gpart2,
                               const double *__restrict__ gpart3_rd, int gnblk, int direct, int pbuf, int rd, int maxit) {
    const int b = blockIdx.x;
    TrajState &S = st[b];
    __shared__ double s4[4];
    if (!S.lin_active) return;
    if (!S.ci_active[rd]) {
        if (threadIdx.x == 0) { S.lin_active = 0; S.cg_pending = 0; }
        return;
    }
    adj_sums4<192>(gpart, gpart2, direct ? gpart3_rd : nullptr, gnblk, b, s4);
    if (threadIdx.x != 0) return;
    const CgNextAdj n = cg_next_adj(s4[0], s4[1], direct ? s4[2] : S.ci_gamma[rd], direct ? s4[3] : S.aux[0], S.lin_r0,
                                    S.ci_it[rd], S.lin_reltol, maxit);
    adj_record(S, n, rd ^ 1);
    if (n.it > S.step_lin_max) S.step_lin_max = n.it;
    S.cg_pending = n.breakdown ? 0 : 1;
    S.cg_pbuf = pbuf;
    S.lin_active = n.active;      // still set: the enqueued sweeps did not reach the tolerance (k_fin_lin_begin counts it)
}

template <int FIRST, int C, int LOGL>
__global__ __launch_bounds__((FftThreads<C, LOGL>::T)) void k_adj_rows_fwd(Geom G, FftAxis ax, AdjSweepArgs a, double *__restrict__ out,
                                                                          TrajState *__restrict__ st) {
    const int b = blockIdx.z;
    __shared__ double2 buf[FftLds<C>::SIZE];
    __shared__ double s4[4];
    constexpr int T = FftThreads<C, LOGL>::T;
    const int tid = threadIdx.x;
    const int logL = LOGL ? LOGL : ax.logL, L = 1 << logL, N = L >> 1, nfft = C >> logL;
    const int row0 = blockIdx.x * 2 * nfft, n1 = N + 1;
    const int rd = (a.it + 1) & 1, wr = a.it & 1;
    const long pb = b * G.plane;
    double alpha = 0.0, beta = 0.0;
    if (FIRST) {
        // start of an adjoint solve: every workgroup sums gamma0 = <r0,r0>_Z' and ||r0||^2 (the partials of k_adj_op<1>),
        // so all of them know whether the start value already solves the system; workgroup 0 records the state
        const int pending = st[b].lin_active;
        const double rhsn = st[b].lin_r0;
        double g = 0.0, r2 = 0.0;
        if (pending) {
            if (tid < 64) {
                for (int t = tid; t < a.nblk; t += 64) {
                    g += a.part[((long)b * a.nblk + t) * NPART];
                    r2 += a.part[((long)b * a.nblk + t) * NPART + 1];
                }
                g = wave_sum(g);
                r2 = wave_sum(r2);
                if (tid == 0) {
                    s4[0] = g;
                    s4[1] = r2;
                }
            }
            __syncthreads();
            g = s4[0];
            r2 = s4[1];
        }
        const double rel = rhsn > 0.0 ? sqrt(r2) / rhsn : 0.0;
        const int start = pending && (rel > a.tol) && (g > 0.0);
        if (blockIdx.x == 0 && tid == 0) {
            TrajState &S = st[b];
            S.cg_pending = 0;
            if (pending) {
                S.cg_gamma = S.cg_gamma0 = g;
                S.cg_beta = 0.0;
                S.lin_it = 0;
                S.lin_reltol = a.tol;
                S.aux[0] = r2;
                S.lin_rel = rel;
                S.guess_ratio = rel;             // ||rhs - A x0|| / ||rhs||: what the starting guess left (backward_pass)
                if (!start) {
                    S.lin_active = 0;
                    if (rel > S.lin_maxrel) S.lin_maxrel = rel;
                }
                S.ci_it[0] = 0;
                S.ci_gamma[0] = g;
            }
            S.ci_active[0] = start;
            S.lin_took = start;
        }
        if (!start) return;
    } else {
        if (!st[b].ci_active[rd]) {
            if (blockIdx.x == 0 && tid == 0) {
                st[b].ci_active[wr] = 0;
                st[b].ci_it[wr] = st[b].ci_it[rd];
                st[b].ci_gamma[wr] = st[b].ci_gamma[rd];
            }
            return;
        }
        const double gamma_old = st[b].ci_gamma[rd], tol = st[b].lin_reltol, r0sq = st[b].aux[0], rhsn = st[b].lin_r0;
        const int it_old = st[b].ci_it[rd], n = gridDim.x;
        adj_sums4<T>(a.gpart, a.gpart2, a.it >= 2 ? a.gpart3 + (long)rd * a.nbatch * n * 2 : nullptr, n, b, s4);
        const CgNextAdj nx = cg_next_adj(s4[0], s4[1], a.it >= 2 ? s4[2] : gamma_old, a.it >= 2 ? s4[3] : r0sq, rhsn, it_old, tol,
                                         a.maxit);
        if (blockIdx.x == 0 && tid == 0) {
            adj_record(st[b], nx, wr);
            if (nx.it > st[b].step_lin_max) st[b].step_lin_max = nx.it;
        }
        alpha = nx.alpha;
        beta = nx.beta;
        if (!nx.active) {
            if (!nx.breakdown)
                for (int idx = tid; idx < 2 * nfft * n1; idx += T) {
                    const int rr = idx / n1, k = idx - rr * n1, row = row0 + rr;
                    if (row < G.ns) {
                        const long o = pb + (long)row * G.pitch + k;
                        a.y[o] += alpha * a.p_old[o];
                    }
                }
            return;
        }
    }
    const double dbar = st[b].dbar;
    const double *Dp = a.Dn + pb;
    double acc = 0.0, acc2 = 0.0;
    auto node = [&](int row, int m, bool owner) -> double {
        if (row >= G.ns) return 0.0;
        const long o = pb + (long)row * G.pitch + m;
        double pn;
        if (FIRST) {
            pn = a.r[o];
            if (owner) {
                a.y[o] = 0.0;
                a.p_new[o] = pn;
            }
        } else {
            const double po = a.p_old[o];
            const double rn = a.r[o] - alpha * a.q[o];
            pn = rn + beta * po;
            if (owner) {
                a.y[o] += alpha * po;
                a.r_new[o] = rn;
                a.p_new[o] = pn;
                acc += wdev(row, m, G) / (Dp[o - pb] - dbar) * (rn * rn);
                acc2 += rn * rn;
            }
        }
        return pn;
    };
    double *ob = out + pb;
    auto put = [&](int row, int k, double e) { ob[(long)row * G.pitch + k] = e; };
    constexpr bool DIRECT = FFT_R8 && LOGL >= 9 && LOGL <= 11;
    if (DIRECT) {
        struct Emit {
            enum { ACTIVE = 1, TO_LDS = 0 };
            decltype(put) &put_;
            int row0_, ns_;
            __device__ __forceinline__ void operator()(int idx, double2 v) const {
                constexpr int LL = 1 << (LOGL ? LOGL : 1);
                const int f = idx >> (LOGL ? LOGL : 1), k = idx & (LL - 1);
                if (k <= LL / 2) {
                    const int ra = row0_ + 2 * f;
                    if (ra < ns_) put_(ra, k, v.x);
                    if (ra + 1 < ns_) put_(ra + 1, k, v.y);
                }
            }
        };
        struct Ingest {
            enum { ACTIVE = 1 };
            decltype(node) &node_;
            int row0_;
            __device__ __forceinline__ double2 operator()(int idx) const {
                constexpr int LL = 1 << (LOGL ? LOGL : 1);
                const int f = idx >> (LOGL ? LOGL : 1), i = idx & (LL - 1);
                const bool owner = i <= LL / 2;
                const int m = owner ? i : LL - i, ra = row0_ + 2 * f;
                return make_double2(node_(ra, m, owner), node_(ra + 1, m, owner));
            }
        };
#if CG_ROWS_STAGED
        // every node once (no second visit for the mirror half of the even extension): stage the image, then transform
        for (int idx = tid; idx < nfft * n1; idx += T) {
            const int f = nfft == 1 ? 0 : idx / n1, j = idx - f * n1;
            const int ra = row0 + 2 * f;
            const double2 v = make_double2(node(ra, j, true), node(ra + 1, j, true));
            buf[swz<LOGL>(f * L + j)] = v;
            if (j > 0 && j < N) buf[swz<LOGL>(f * L + L - j)] = v;
        }
        __syncthreads();
        fft_lds<C, LOGL>(buf, ax, Emit{put, row0, G.ns});
#else
        fft_lds<C, LOGL>(buf, ax, Emit{put, row0, G.ns}, Ingest{node, row0});
#endif
    } else {
        const float inv_n1 = 1.0f / (float)n1;
        for (int idx = tid; idx < nfft * n1; idx += T) {
            const int f = nfft == 1 ? 0 : (int)(((float)idx + 0.5f) * inv_n1), j = idx - f * n1;
            const int ra = row0 + 2 * f;
            const double2 v = make_double2(node(ra, j, true), node(ra + 1, j, true));
            buf[swz<LOGL>(f * L + j)] = v;
            if (j > 0 && j < N) buf[swz<LOGL>(f * L + L - j)] = v;
        }
        __syncthreads();
        fft_lds<C, LOGL>(buf, ax);
        for (int idx = tid; idx < 2 * nfft * n1; idx += T) {
            const int rr = nfft == 1 ? (idx >= n1 ? 1 : 0) : (int)(((float)idx + 0.5f) * inv_n1), k = idx - rr * n1;
            const int row = row0 + rr;
            if (row < G.ns) {
                const double2 c = buf[swz<LOGL>((rr >> 1) * L + k)];
                put(row, k, (rr & 1) ? c.y : c.x);
            }
        }
    }
    if (!FIRST) {
        acc = wave_sum(acc);
        acc2 = wave_sum(acc2);
        __syncthreads();
        double *sred = reinterpret_cast<double *>(buf);
        if ((tid & 63) == 0) {
            sred[tid >> 6] = acc;
            sred[T / 64 + (tid >> 6)] = acc2;
        }
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0, tot2 = 0.0;
            for (int w = 0; w < T / 64; ++w) {
                tot += sred[w];
                tot2 += sred[T / 64 + w];
            }
            double *dst = a.gpart3 + (((long)wr * a.nbatch + b) * gridDim.x + blockIdx.x) * 2;
            dst[0] = tot;
            dst[1] = tot2;
        }
    }
}

// E along the slow axis, spectral multiplier 1/(c0 + m (c1 + c2 m)) (m = ms[k] + mf[col]), E again.
template <int C, int LOGL>
__global__ __launch_bounds__((FftThreads<C, LOGL>::T)) void k_dct_cols(Geom G, FftAxis ax, const double *__restrict__ in,
                                                  double *__restrict__ out, double scale, SpecArgs sp,
                                                  const TrajState *__restrict__ st, int gate) {
    const int b = blockIdx.z;
    if (gate && !gate_open(st[b], gate)) return;
    __shared__ double2 buf[FftLds<C>::SIZE];
    constexpr int T = FftThreads<C, LOGL>::T;
    const int tid = threadIdx.x;
    const int logL = LOGL ? LOGL : ax.logL, L = 1 << logL, N = L >> 1, nfft = C >> logL, ncol = 2 * nfft;
    const int col0 = xcd_remap(blockIdx.x, gridDim.x) * ncol;
    const double *ib = in + b * G.plane;
    double *sb = reinterpret_cast<double *>(buf);
    const int n1 = N + 1;
    int lc = 0;                                   // log2(ncol)
    while ((1 << lc) < ncol) ++lc;
    constexpr bool DIRECT = FFT_R8 && LOGL >= 9 && LOGL <= 11;
    constexpr int LL = 1 << (LOGL ? LOGL : 1), LG = LOGL ? LOGL : 1;
    if (!(DIRECT && FFT_COLS_INGEST && (FFT_COLS_EMIT & 1))) {
#if FFT_COLS_PAIR
        // a column pair is 16 contiguous, 16-byte aligned bytes of a row (col0 even, pitch a multiple of 8): one b128 load
        // and one b128 LDS store per row instead of two 8-byte ones from two lanes
        for (int idx = tid; idx < n1 * nfft; idx += T) {
            const int r = nfft == 1 ? idx : idx / nfft, f = nfft == 1 ? 0 : idx - r * nfft;
            const int ca = col0 + 2 * f;
            const double *p = ib + (long)r * G.pitch + ca;
            double2 v = make_double2(0.0, 0.0);
            if (ca + 1 < G.nf) v = *reinterpret_cast<const double2 *>(p);
            else if (ca < G.nf) v.x = p[0];
            buf[swz<LOGL>(f * L + r)] = v;
            if (r > 0 && r < N) buf[swz<LOGL>(f * L + L - r)] = v;
        }
        if (false)
#endif
        for (int idx = tid; idx < n1 * ncol; idx += T) {
            const int r = idx >> lc, cc = idx & (ncol - 1);
            const int col = col0 + cc;
            const double v = col < G.nf ? ib[(long)r * G.pitch + col] : 0.0;
            const int f = cc >> 1, comp = cc & 1;
            sb[2 * swz<LOGL>(f * L + r) + comp] = v;
            if (r > 0 && r < N) sb[2 * swz<LOGL>(f * L + L - r) + comp] = v;
        }
        __syncthreads();
    }
    // first pass of the forward transform fed from global memory: entry i of the even extension of columns col0 + 2f, + 1
    struct ColIngest {
        enum { ACTIVE = 1 };
        const double *ib_;
        long pitch_;
        int col0_, nf_;
        __device__ __forceinline__ double2 operator()(int idx) const {
            const int f = idx >> LG, i = idx & (LL - 1), m = i <= LL / 2 ? i : LL - i;
            const int ca = col0_ + 2 * f;
            const double *p = ib_ + (long)m * pitch_ + ca;
            return make_double2(ca < nf_ ? p[0] : 0.0, ca + 1 < nf_ ? p[1] : 0.0);
        }
    };
    const double c1 = sp.c1a + sp.c1b * st[b].dbar;
    const int mmode = (sp.mult_m && st[b].scaled) ? 2 : sp.mult_m;      // right-scaled CG form of this trajectory's solve
    double *ob = out + b * G.plane;
    // forward transform; with a compile-time plan its last pass writes the outputs back already multiplied by the
    // spectral multiplier (FFT_COLS_EMIT & 1), otherwise a separate pass over the image does
    struct ScaleEmit {
        enum { ACTIVE = 1, TO_LDS = 1 };
        double2 *buf_;
        const SpecArgs &sp_;
        double c1_, scale_;
        int col0_, nf_, mode_;
        __device__ __forceinline__ void operator()(int idx, double2 v) const {
            const int f = idx >> LG, k = idx & (LL - 1), ks = k <= LL / 2 ? k : LL - k;
            const int ca = col0_ + 2 * f, cb = ca + 1;
            const double msk = sp_.ms[ks];
            const double ma = msk + sp_.mf[ca < nf_ ? ca : nf_ - 1], mb = msk + sp_.mf[cb < nf_ ? cb : nf_ - 1];
            double fa, fb;
            spec_mult2(sp_, c1_, scale_, ma, mb, fa, fb, mode_);
            v.x *= fa;
            v.y *= fb;
            buf_[swz<LOGL>(idx)] = v;
        }
    };
    constexpr bool FUSE = FFT_COLS_FUSE && DIRECT && FftRegOk<C, LOGL>::V && !FFT_COLS_INGEST && !(FFT_COLS_EMIT & 2);
    if constexpr (FUSE) {
        // first transform, its last pass kept in registers; multiplier; second transform fed from those registers: one
        // LDS round trip and one barrier pair less than ScaleEmit, same arithmetic
        double2 xr[8];
        fft_lds<C, LOGL, FftNoEmit, FftNoIngest, 1>(buf, ax, FftNoEmit(), FftNoIngest(), xr);
        const int f = tid >> (LG - 3), j = tid & ((LL >> 3) - 1);
        const int ca = col0 + 2 * f, cb = ca + 1;
        const double mfa = sp.mf[ca < G.nf ? ca : G.nf - 1], mfb = sp.mf[cb < G.nf ? cb : G.nf - 1];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = j + s * (LL >> 3), ks = k <= LL / 2 ? k : LL - k;
            const double msk = sp.ms[ks];
            double fa, fb;
            spec_mult2(sp, c1, scale, msk + mfa, msk + mfb, fa, fb, mmode);
            xr[s].x *= fa;
            xr[s].y *= fb;
        }
        fft_lds<C, LOGL, FftNoEmit, FftNoIngest, 2>(buf, ax, FftNoEmit(), FftNoIngest(), xr);
    } else if (DIRECT && (FFT_COLS_EMIT & 1)) {
        if (FFT_COLS_INGEST)
            fft_lds<C, LOGL>(buf, ax, ScaleEmit{buf, sp, c1, scale, col0, G.nf, mmode}, ColIngest{ib, (long)G.pitch, col0, G.nf});
        else
            fft_lds<C, LOGL>(buf, ax, ScaleEmit{buf, sp, c1, scale, col0, G.nf, mmode});
    } else {
        fft_lds<C, LOGL>(buf, ax);
        for (int idx = tid; idx < nfft * L; idx += T) {
            const int f = idx >> logL, k = idx & (L - 1);
            const int ks = k <= N ? k : L - k;
            const int ca = col0 + 2 * f, cb = ca + 1;
            const double msk = sp.ms[ks];
            double2 v = buf[swz<LOGL>(idx)];
            double ma = msk + sp.mf[ca < G.nf ? ca : G.nf - 1], mb = msk + sp.mf[cb < G.nf ? cb : G.nf - 1];
            double fa, fb;
            spec_mult2(sp, c1, scale, ma, mb, fa, fb, mmode);
            v.x *= fa;
            v.y *= fb;
            buf[swz<LOGL>(idx)] = v;
        }
        __syncthreads();
    }
    // second transform; the wanted half of its last pass's outputs can go straight to global memory (FFT_COLS_EMIT & 2)
    struct ColEmit {
        enum { ACTIVE = 1, TO_LDS = 0 };
        double *ob_;
        long pitch_;
        int col0_, nf_;
        __device__ __forceinline__ void operator()(int idx, double2 v) const {
            const int f = idx >> LG, k = idx & (LL - 1);
            if (k <= LL / 2) {
                const int ca = col0_ + 2 * f;
                double *po = ob_ + (long)k * pitch_ + ca;
                if (ca < nf_) po[0] = v.x;
                if (ca + 1 < nf_) po[1] = v.y;
            }
        }
    };
    if (DIRECT && (FFT_COLS_EMIT & 2)) {
        fft_lds<C, LOGL>(buf, ax, ColEmit{ob, (long)G.pitch, col0, G.nf});
        return;
    }
    if constexpr (!FUSE) fft_lds<C, LOGL>(buf, ax);
#if FFT_COLS_PAIR
    for (int idx = tid; idx < n1 * nfft; idx += T) {
        const int r = nfft == 1 ? idx : idx / nfft, f = nfft == 1 ? 0 : idx - r * nfft;
        const int ca = col0 + 2 * f;
        const double2 v = buf[swz<LOGL>(f * L + r)];
        double *p = ob + (long)r * G.pitch + ca;
        if (ca + 1 < G.nf) *reinterpret_cast<double2 *>(p) = v;
        else if (ca < G.nf) p[0] = v.x;
    }
    return;
#endif
    for (int idx = tid; idx < n1 * ncol; idx += T) {
        const int r = idx >> lc, cc = idx & (ncol - 1);
        const int col = col0 + cc;
        if (col < G.nf) ob[(long)r * G.pitch + col] = sb[2 * swz<LOGL>((cc >> 1) * L + r) + (cc & 1)];
    }
}

// =====================================================================================================
// Half-size DCT-I for N = 512 intervals (the 512^2 grid), OPT-IN with VCH_DCT_HALF=1 (parity-tested, but not faster on
// MI355X at the batch sizes of the bench: the passes are latency-bound there, see profiles/r01_c_fft_variants.txt).
// E of a real sequence x_0..x_N needs only a
// real FFT of length N (instead of the complex FFT of its even extension, length 2N, shared by two rows):
//     y_n = (x_n + x_{N-n})/2 - sin(pi n/N) (x_n - x_{N-n}),  n = 0..N-1;     Y = FFT_N(y)
//     E_{2k}   = 2 Re Y_k,                                      k = 0..N/2
//     E_{2k+1} = 2 (X_1 - sum_{j=1..k} Im Y_j),                 X_1 = (x_0 - x_N)/2 + sum_{0<n<N} x_n cos(pi n/N)
// Two rows a, b share ONE complex FFT of length N (z = y_a + i y_b, untangled afterwards), so a workgroup
// image of 1024 complex doubles now carries FOUR rows: half the butterflies and LDS passes per row.  Each
// wavefront owns one transform in the pre- and post-passes (64 lanes x 8 points), so the running sum of the
// odd outputs is a wavefront scan and X_1 a wavefront sum.  Accuracy against the long transform: 2e-15 of the
// largest coefficient (tests/test_gpu_2d.py::test_spectral_solve...; the true residual of a converged solve
// moves from 5e-16 to 6e-15), irrelevant for a preconditioner.
// =====================================================================================================
constexpr int HN = 512, HLOGN = 9, HC = 1024, HT = 128;

__device__ __forceinline__ double wave_bcast_sum(double v) { return __shfl(wave_sum(v), 0, 64); }
__device__ __forceinline__ double wave_excl_scan(double v) {
    const int lane = threadIdx.x & 63;
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    return inc - v;
}

// pre-pass of one transform from registers: xa/xb[j] = x_n, ma/mb[j] = x_{N-n} for n = l + 64 j; writes the
// packed sequence into image f and returns the two X_1 sums
__device__ __forceinline__ void dcth_pre(double2 *buf, const FftAxis &axL, int f, int l, const double (&xa)[8],
                                         const double (&ma)[8], const double (&xb)[8], const double (&mb)[8],
                                         double &x1a, double &x1b) {
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = l + 64 * j;
        const double2 t = axL.tw[n];                     // exp(-i pi n / N): cos = t.x, sin = -t.y
        const double ya = 0.5 * (xa[j] + ma[j]) + t.y * (xa[j] - ma[j]);
        const double yb = 0.5 * (xb[j] + mb[j]) + t.y * (xb[j] - mb[j]);
        buf[swz<HLOGN>(f * HN + n)] = make_double2(ya, yb);
        if (n == 0) {
            sa += 0.5 * (xa[j] - ma[j]);
            sb += 0.5 * (xb[j] - mb[j]);
        } else {
            sa += t.x * xa[j];
            sb += t.x * xb[j];
        }
    }
    x1a = wave_bcast_sum(sa);
    x1b = wave_bcast_sum(sb);
}

// post-pass of one transform: lane l gets E_{8l .. 8l+7} of both rows (ea, eb) and lane 63 also E_N (eNa, eNb)
__device__ __forceinline__ void dcth_post(const double2 *buf, int f, int l, double x1a, double x1b, double (&ea)[8],
                                          double (&eb)[8], double &eNa, double &eNb) {
    double da[4], db[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 4 * l + j;
        const double2 zk = buf[swz<HLOGN>(f * HN + k)], zm = buf[swz<HLOGN>(f * HN + ((HN - k) & (HN - 1)))];
        ea[2 * j] = 0.5 * (zk.x + zm.x);                 // Re Y_a
        eb[2 * j] = 0.5 * (zk.y + zm.y);                 // Re Y_b
        da[j] = k == 0 ? 0.0 : -0.5 * (zk.y - zm.y);     // -Im Y_a
        db[j] = k == 0 ? 0.0 : 0.5 * (zk.x - zm.x);      // -Im Y_b
    }
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        da[j] += da[j - 1];
        db[j] += db[j - 1];
    }
    const double oa = x1a + wave_excl_scan(da[3]), ob = x1b + wave_excl_scan(db[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ea[2 * j + 1] = oa + da[j];
        eb[2 * j + 1] = ob + db[j];
    }
    const double2 zh = buf[swz<HLOGN>(f * HN + HN / 2)];
    eNa = zh.x;
    eNb = zh.y;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ea[j] *= 2.0;
        eb[j] *= 2.0;
    }
    eNa *= 2.0;
    eNb *= 2.0;
}

// E along the fast axis, 4 rows per workgroup.  EPI as in k_dct_rows.
template <int EPI>
__global__ __launch_bounds__(HT) void k_dcth_rows(Geom G, FftAxis axL, FftAxis axN, const double *__restrict__ in,
                                                  long in_slot_stride, double *__restrict__ out, double scale, SpecArgs sp,
                                                  const TrajState *__restrict__ st, int gate) {
    const int b = blockIdx.z;
    if (gate && !gate_open(st[b], gate)) return;
    __shared__ double2 buf[HC];
    const int tid = threadIdx.x, f = tid >> 6, l = tid & 63;
    const int ra = blockIdx.x * 4 + 2 * f, rb = ra + 1;
    const bool va = ra < G.ns, vb = rb < G.ns;
    const double *ib = in + b * G.plane + (in_slot_stride ? st[b].slot * in_slot_stride : 0);
    const double *pa = ib + (long)ra * G.pitch, *pb = ib + (long)rb * G.pitch;
    double xa[8], ma[8], xb[8], mb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = l + 64 * j;
        xa[j] = va ? pa[n] : 0.0;
        ma[j] = va ? pa[HN - n] : 0.0;
        xb[j] = vb ? pb[n] : 0.0;
        mb[j] = vb ? pb[HN - n] : 0.0;
    }
    double x1a, x1b;
    dcth_pre(buf, axL, f, l, xa, ma, xb, mb, x1a, x1b);
    __syncthreads();
    fft_lds<HC, HLOGN>(buf, axN);
    double ea[8], eb[8], eNa, eNb;
    dcth_post(buf, f, l, x1a, x1b, ea, eb, eNa, eNb);
    double dot = 0.0, dot2 = 0.0, dbar = 0.0;
    const double *Dp = nullptr, *Ob = nullptr;
    if (EPI == 3) {
        dbar = st[b].dbar;
        Dp = sp.Dslot + st[b].slot * sp.d_slot_stride + b * G.plane;
        Ob = sp.other ? sp.other + b * G.plane : nullptr;
    }
    double *ob = out + b * G.plane;
    auto put = [&](int row, int k, double e) {
        const double v = scale * e;
        const long o = (long)row * G.pitch + k;
        ob[o] = v;
        if (EPI == 3) {
            const double wd = wdev(row, k, G) * (Dp[o] - dbar);
            dot += wd * ((Ob ? Ob[o] : v) * v);
            dot2 += wd * (v * v);
        }
    };
    // lane l holds outputs 8l..8l+7: turned through the (wave-private) image so that consecutive lanes
    // store consecutive elements (and the epilogue reads D / other coalesced)
    double *ra_ = reinterpret_cast<double *>(buf) + (long)f * 2 * HN, *rb_ = ra_ + HN;
    __syncthreads();                               // all post-pass reads of the image are done
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ra_[8 * l + j] = ea[j];
        rb_[8 * l + j] = eb[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = l + 64 * j;
        if (va) put(ra, k, ra_[k]);
        if (vb) put(rb, k, rb_[k]);
    }
    if (l == 63) {
        if (va) put(ra, HN, eNa);
        if (vb) put(rb, HN, eNb);
    }
    if (EPI == 3) {
        dot = wave_sum(dot);
        dot2 = wave_sum(dot2);
        __syncthreads();
        double *sred = reinterpret_cast<double *>(buf);
        if (l == 0) {
            sred[f] = dot;
            sred[2 + f] = dot2;
        }
        __syncthreads();
        if (tid == 0) {
            sp.gpart[(long)b * gridDim.x + blockIdx.x] = sred[0] + sred[1];
            sp.gpart2[(long)b * gridDim.x + blockIdx.x] = sred[2] + sred[3];
        }
    }
}

// E along the slow axis for 4 columns, spectral multiplier, E again (one load, one store).
__global__ __launch_bounds__(HT) void k_dcth_cols(Geom G, FftAxis axL, FftAxis axN, const double *__restrict__ in,
                                                  double *__restrict__ out, double scale, SpecArgs sp,
                                                  const TrajState *__restrict__ st, int gate) {
    const int b = blockIdx.z;
    if (gate && !gate_open(st[b], gate)) return;
    __shared__ double2 buf[HC];
    const int tid = threadIdx.x, f = tid >> 6, l = tid & 63;
    const int ca = xcd_remap(blockIdx.x, gridDim.x) * 4 + 2 * f, cb = ca + 1;
    const bool va = ca < G.nf, vb = cb < G.nf;
    const double *ib = in + b * G.plane;
    double xa[8], ma[8], xb[8], mb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = l + 64 * j;
        const double *pn = ib + (long)n * G.pitch, *pm = ib + (long)(HN - n) * G.pitch;
        xa[j] = va ? pn[ca] : 0.0;
        ma[j] = va ? pm[ca] : 0.0;
        xb[j] = vb ? pn[cb] : 0.0;
        mb[j] = vb ? pm[cb] : 0.0;
    }
    double x1a, x1b;
    dcth_pre(buf, axL, f, l, xa, ma, xb, mb, x1a, x1b);
    __syncthreads();
    fft_lds<HC, HLOGN>(buf, axN);
    double ea[8], eb[8], eNa, eNb;
    dcth_post(buf, f, l, x1a, x1b, ea, eb, eNa, eNb);
    // spectral multiplier on E (mode kk along the slow axis, column along the fast axis)
    const double c1 = sp.c1a + sp.c1b * st[b].dbar;
    const double mfa = sp.mf[va ? ca : G.nf - 1], mfb = sp.mf[vb ? cb : G.nf - 1];
    auto mul = [&](int kk, double mfc, double e) {
        const double m = sp.ms[kk] + mfc;
        return e * (scale / (sp.c0 + m * (c1 + sp.c2 * m)));
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ea[j] = mul(8 * l + j, mfa, ea[j]);
        eb[j] = mul(8 * l + j, mfb, eb[j]);
    }
    eNa = mul(HN, mfa, eNa);
    eNb = mul(HN, mfb, eNb);
    // the scaled spectrum becomes the input of the second transform: through the (wave-private) image as two
    // real arrays of N entries each; entry N travels by shuffle from lane 63
    double *ra_ = reinterpret_cast<double *>(buf) + (long)f * 2 * HN, *rb_ = ra_ + HN;
    __syncthreads();                               // all post-pass reads of the image are done
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ra_[8 * l + j] = ea[j];
        rb_[8 * l + j] = eb[j];
    }
    const double xNa = __shfl(eNa, 63, 64), xNb = __shfl(eNb, 63, 64);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = l + 64 * j;
        xa[j] = ra_[n];
        xb[j] = rb_[n];
        ma[j] = n == 0 ? xNa : ra_[HN - n];
        mb[j] = n == 0 ? xNb : rb_[HN - n];
    }
    __syncthreads();                               // reads done before the image is overwritten
    dcth_pre(buf, axL, f, l, xa, ma, xb, mb, x1a, x1b);
    __syncthreads();
    fft_lds<HC, HLOGN>(buf, axN);
    dcth_post(buf, f, l, x1a, x1b, ea, eb, eNa, eNb);
    double *ob = out + b * G.plane;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double *po = ob + (long)(8 * l + j) * G.pitch;
        if (va) po[ca] = ea[j];
        if (vb) po[cb] = eb[j];
    }
    if (l == 63) {
        double *po = ob + (long)HN * G.pitch;
        if (va) po[ca] = eNa;
        if (vb) po[cb] = eNb;
    }
}

