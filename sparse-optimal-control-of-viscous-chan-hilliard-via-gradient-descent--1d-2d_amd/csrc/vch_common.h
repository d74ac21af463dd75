// vch_common.h — shared host/device definitions of the HIP engine (gfx950 / MI355X).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include "../../include/vch.h"

// ----------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------
extern thread_local char g_vch_err[512];
static inline int vch_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_vch_err, sizeof(g_vch_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (void)hipGetLastError();     /* reset the sticky error for later launch checks */ \
            return vch_fail(VCH_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                            \
        }                                                                                   \
    } while (0)
#define VCHCHK(call)                   \
    do {                               \
        int r_ = (call);               \
        if (r_ < 0) return r_;         \
    } while (0)

// ----------------------------------------------------------------------------------
// geometry of one padded plane in HBM
//
// A field of the reference is an (Nx+1, Ny+1) row-major array.  Its Laplacian matrix is
// kron(I_{Ny+1}, Lx) + kron(Ly, I_{Nx+1}) applied to the y-fastest ravel (F2:125-152), i.e.
// the operator acts on the flat memory reinterpreted as `ns = Ny+1` rows of `nf = Nx+1`
// entries: hx-stencil along the fast axis, hy-stencil along the slow axis.  The engine stores
// exactly that reinterpretation, rows padded to `pitch` doubles (multiple of 8 => every row
// starts on a 64-byte boundary).
// ----------------------------------------------------------------------------------
struct Geom {
    int nf, ns;        // fast / slow extents
    int pitch;         // padded fast extent (doubles)
    int tiles_f, tiles_s;
    long plane;        // ns * pitch (doubles)
    double ax, ay;     // 1/hx^2 (fast axis), 1/hy^2 (slow axis)
};

constexpr int TX = 64;      // tile extent along the fast axis (one wavefront wide)
#ifndef VCH_TY
#define VCH_TY 16
#endif
constexpr int TY = VCH_TY;  // tile extent along the slow axis (A/B knob: 8 or 16; every thread owns TY / 4 rows)
constexpr int NTH = 256;    // threads per workgroup = 4 wavefronts
constexpr int NPART = 6;    // partial-reduction slots per workgroup

constexpr double DELTA_SEP = 1e-2;   // F2:510
