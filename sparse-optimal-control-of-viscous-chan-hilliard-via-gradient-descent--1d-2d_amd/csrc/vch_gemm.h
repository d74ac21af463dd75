// vch_gemm.h — batched fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64) for the
// fast-diagonalisation preconditioner.
//
// The constant-coefficient operator  c0 + m (c1 + c2 m)  (m = eigenvalues of M = -L) is
// diagonal in the DCT-I basis of the mirrored-Neumann Laplacian (eigenvectors
// cos(pi j k / N), F2:115-122).  For any grid size the two 1-D transforms are applied as
// dense matrix products with precomputed matrices Q1 = S C^ and Q2 = C^ S^-1 (C^ the
// orthonormal symmetric DCT-I matrix, S = diag(1/sqrt2,1,...,1,1/sqrt2)):
//     z = Q2s^T [ mult o (Q1s^T (g Q1f)) ] Q2f            (g = plane viewed as ns x nf)
// i.e. four GEMMs per application, the spectral multiplier fused into the second one and the
// weighted dot product that CG needs next fused into the last one (EPI 3).
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 wavefronts as 2x2, each 32x32 =
// 2x2 MFMA tiles of 16x16), K advanced 16 at a time through LDS.  MFMA operand layout for
// 16x16x4 f64: lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; the four
// accumulator values of lane l are C[row = (l>>4) + 4*reg][col = l&15].
// LDS images are padded so that the operand reads are bank-conflict free:
//   [k][i] images use a row stride of 80 doubles (lanes l and l+16 land 32 banks apart),
//   [i][k] images use a row stride of 17 doubles.
#pragma once
#include "vch_common.h"
#include "vch_kernels2d.h"

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int GM = 64, GN = 64, GK = 16;
constexpr int LDK = 80;     // row stride of [k][*] LDS images
constexpr int LDI = 17;     // row stride of [i][k] LDS images

struct SpecArgs {           // multiplier = 1 / (c0 + m (c1a + c1b*dbar[b] + c2 m)), m = ms[row] + mf[col]
    double c0, c1a, c1b, c2;
    const double *ms, *mf;
    // EPI 3: per-workgroup partial of  sum W (D - dbar) other * C  (other == NULL: C * C)
    const double *other;     // [B][plane]
    const double *Dslot;     // slot-indexed D planes
    long d_slot_stride;
    double *gpart;           // [B][gridDim.x*gridDim.y]
    double *gpart2;          // same layout: partial of sum W (D - dbar) C * C
    int mult_m;              // FFT column pass: multiplier m / (c0 + ...) instead of 1 / (c0 + ...)
    double epi_c;            // EPI 5 (adjoint sweep): out = other + epi_c (D - dbar) E(in)
};

// EPI: 0 store, 1 store * spectral multiplier, 2 accumulate (C += A B), 3 store + weighted dot partial
// A_KMAJOR: A is stored [K][M] (lda = row length M-side), else [M][K].
// gate: 0 none, 1 only trajectories with lin_active, 2 / 3 with ci_active[0] / [1] (gate_open)
template <bool A_KMAJOR, int EPI>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, const double *__restrict__ A, long lda,
                                               long sA, long a_slot_stride, const double *__restrict__ Bm,
                                               long ldb, long sB, double *__restrict__ C, long ldc, long sC,
                                               SpecArgs sp, const TrajState *__restrict__ st, int gate) {
    const int b = blockIdx.z;
    if (gate && !gate_open(st[b], gate)) return;
    __shared__ double As[GK * LDK];     // 1280 doubles; the [i][k] image needs 64*17 = 1088
    __shared__ double Bs[GK * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    const double *Ab = A + b * sA + (a_slot_stride ? st[b].slot * a_slot_stride : 0);
    const double *Bb = Bm + b * sB;
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < K; k0 += GK) {
        if (!A_KMAJOR) {
            const int row = tid >> 2, kq = (tid & 3) * 4;
            const int m = m0 + row;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int k = k0 + kq + q;
                As[row * LDI + kq + q] = (m < M && k < K) ? Ab[(long)m * lda + k] : 0.0;
            }
        } else {
            const int kr = tid >> 4, iq = (tid & 15) * 4;
            const int k = k0 + kr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int m = m0 + iq + q;
                As[kr * LDK + iq + q] = (m < M && k < K) ? Ab[(long)k * lda + m] : 0.0;
            }
        }
        {
            const int kr = tid >> 4, jq = (tid & 15) * 4;
            const int k = k0 + kr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int n = n0 + jq + q;
                Bs[kr * LDK + jq + q] = (n < N && k < K) ? Bb[(long)k * ldb + n] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            const int kl = kk * 4 + (lane >> 4), il = lane & 15;
            double af[2], bf[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                af[mi] = A_KMAJOR ? As[kl * LDK + wm * 32 + mi * 16 + il] : As[(wm * 32 + mi * 16 + il) * LDI + kl];
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) bf[nj] = Bs[kl * LDK + wn * 32 + nj * 16 + il];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
                    acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
        }
        __syncthreads();
    }
    double *Cb = C + b * sC;
    double c1 = 0.0, dot = 0.0, dot2 = 0.0, dbar = 0.0;
    const double *Dp = nullptr, *Ob = nullptr;
    if (EPI == 1) c1 = sp.c1a + sp.c1b * st[b].dbar;
    if (EPI == 3) {
        dbar = st[b].dbar;
        Dp = sp.Dslot + st[b].slot * sp.d_slot_stride + b * sC;
        Ob = sp.other ? sp.other + b * sC : nullptr;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                int row = m0 + wm * 32 + mi * 16 + (lane >> 4) + 4 * rg;
                int col = n0 + wn * 32 + nj * 16 + (lane & 15);
                if (row < M && col < N) {
                    double v = acc[mi][nj][rg];
                    long o = (long)row * ldc + col;
                    if (EPI == 1) {
                        double m = sp.ms[row] + sp.mf[col];
                        v = v / (sp.c0 + m * (c1 + sp.c2 * m));
                    } else if (EPI == 2) {
                        v += Cb[o];
                    } else if (EPI == 3) {
                        double wgt = ((row == 0 || row == M - 1) ? 0.5 : 1.0) * ((col == 0 || col == N - 1) ? 0.5 : 1.0);
                        const double wd = wgt * (Dp[o] - dbar);
                        dot += wd * ((Ob ? Ob[o] : v) * v);
                        dot2 += wd * (v * v);
                    }
                    Cb[o] = v;
                }
            }
    if (EPI == 3) {
        dot = wave_sum(dot);
        dot2 = wave_sum(dot2);
        __syncthreads();
        if (lane == 0) {
            As[wv] = dot;
            As[4 + wv] = dot2;
        }
        __syncthreads();
        if (tid == 0) {
            const long go = (long)b * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x;
            sp.gpart[go] = (As[0] + As[1]) + (As[2] + As[3]);
            sp.gpart2[go] = (As[4] + As[5]) + (As[6] + As[7]);
        }
    }
}
