// kernels_sweep.hpp -- the per-sweep kernels (gfx950, wave64, FP64).
//
// All of them compute one weighted-Jacobi sweep exactly as the reference's
// updateX_SOR (Deff2DGPU/Deff2D.cuh:69-92): sigma accumulates W, E, S(row+1),
// N(row-1) in that order, each term only when its coefficient is non-zero
// (cuh:77), then xNew = (1-w)*x + (w/A0)*(b - sigma) (cuh:89).  With w = 1 the
// expression equals updateX_V1's 1/A0*(b - sigma) (cuh:96-118) bit for bit for
// finite x.  The library is compiled with -ffp-contract=off, so no product is
// fused into an add: results are bit-identical to the CPU oracle.
//
// FMA = false is that arithmetic (the default).  FMA = true is the SAME expression as a compiler
// contracts it when allowed to (gcc -mfma -ffp-contract=fast on the reference's source; nvcc's
// default -fmad=true is the same kind of transformation): every `sigma += a*x` becomes
// fma(a, x, sigma) and the final sum becomes fma(1-w, x, (w/A0)*(b - sigma)).  It is bit-identical
// to the oracle's "fma" build (oracle/Makefile), reproduces the second Deff value the survey
// recorded for the reference and needs 7 instead of 11 FP64 instructions
// per cell.  Opt-in: deff_set_tuning(ctx, "fma", 1).
//
// Neighbour addressing is the reference's linear one (x[p-1], x[p+1],
// x[p+nx], x[p-nx]); a neighbour whose linear index falls outside [0, n) is
// never dereferenced (the reference relies on the zero coefficient there).
//
// Kernels:
//   k_sweep_scalar    1 cell / thread, SoA coefficients (the simplest form; kept selectable as a cross-check)
//   k_sweep_explicit  2 x R cells / thread (16-B accesses), SoA coefficient
//                     streams, all loads of a register tile issued up front:
//                     64 B/cell/sweep of HBM
//   k_sweep_matfree   coefficients looked up in the row dictionary (LDS) by a 16-bit
//                     code per cell, persistent workgroups: 18 B/cell/sweep of HBM
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lut_layout.hpp"

namespace deff {

struct CoefConst {
    const double *c0, *aW, *aE, *aS, *aN, *b;
};

template <bool FMA>
__device__ __forceinline__ double mul_add(double a, double x, double acc)
{
    if constexpr (FMA) return __builtin_fma(a, x, acc);
    else return acc + a * x;
}

template <bool FMA>
__device__ __forceinline__ double jacobi_cell(double c0, double aW, double aE, double aS, double aN,
                                              double b, double xc, double xw, double xe, double xs,
                                              double xn, double omw)
{
    double sigma = 0;
    if (aW != 0) sigma = mul_add<FMA>(aW, xw, sigma);
    if (aE != 0) sigma = mul_add<FMA>(aE, xe, sigma);
    if (aS != 0) sigma = mul_add<FMA>(aS, xs, sigma);
    if (aN != 0) sigma = mul_add<FMA>(aN, xn, sigma);
    if constexpr (FMA) return __builtin_fma(omw, xc, c0 * (b - sigma));
    else return omw * xc + c0 * (b - sigma);
}

// Workgroup -> tile map.  Workgroups are dealt round-robin over the 8 XCDs, so
// blocks id and id+8 share an L2.  Give every XCD a contiguous run of tiles in
// column-major order (tiles above/below each other, which share halo rows of
// x, then land in the same L2).  Placement only affects speed, never results.
//
// flip = 1 walks every XCD's run backwards.  Alternating it from sweep to sweep
// ("serpentine") makes a sweep start on the rows the previous sweep wrote last,
// i.e. on the part of the field most likely still in L2 / Infinity Cache.
__device__ __forceinline__ bool xcd_tile(int gx, int gy, int flip, int &bx, int &by)
{
    const unsigned total = (unsigned)gx * (unsigned)gy;
    const unsigned per = (total + 7u) / 8u;
    const unsigned id = blockIdx.x;
    if ((id >> 3) >= per) return false;
    const unsigned k = flip ? per - 1u - (id >> 3) : (id >> 3);
    const unsigned t = (id & 7u) * per + k;
    if (t >= total) return false;
    bx = (int)(t / (unsigned)gy);
    by = (int)(t % (unsigned)gy);
    return true;
}

// ------------------------------------------------------------- scalar -----

// NT = stream the coefficient planes with non-temporal loads: they are read
// once per sweep and are 6x the size of x, so keeping them out of the caches
// leaves L2 / Infinity Cache to the field, which is re-read by neighbours and by
// the next sweep.
template <bool NT>
__device__ __forceinline__ double ldc(const double *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool NT, bool FMA>
__global__ __launch_bounds__(256) void k_sweep_scalar(CoefConst c, const double *__restrict__ x,
                                                      double *__restrict__ xnew, int nx, size_t n,
                                                      size_t n_img, const uint8_t *__restrict__ active,
                                                      double omw)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    if (active && !active[p / n_img]) return;       // frozen image of a batch
    const double xw = (p >= 1) ? x[p - 1] : 0.0;
    const double xe = (p + 1 < n) ? x[p + 1] : 0.0;
    const double xs = (p + nx < n) ? x[p + nx] : 0.0;
    const double xn = (p >= (size_t)nx) ? x[p - nx] : 0.0;
    xnew[p] = jacobi_cell<FMA>(ldc<NT>(c.c0 + p), ldc<NT>(c.aW + p), ldc<NT>(c.aE + p), ldc<NT>(c.aS + p),
                               ldc<NT>(c.aN + p), ldc<NT>(c.b + p), x[p], xw, xe, xs, xn, omw);
}

// ----------------------------------------------------------- explicit -----

__device__ __forceinline__ double2 ld2(const double *p)
{
    return *reinterpret_cast<const double2 *>(p);
}
__device__ __forceinline__ void st2(double *p, double2 v)
{
    *reinterpret_cast<double2 *>(p) = v;
}
template <bool NT>
__device__ __forceinline__ double2 ldc2(const double *p)
{
    if constexpr (NT) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
        return make_double2(v.x, v.y);
    } else {
        return ld2(p);
    }
}

// Requires nx even (16-B aligned row starts).  Tile = 512 columns x R rows; every
// thread owns a 2 x R register tile and issues all of its loads before the
// first use, so (R+2) x-rows and 6R coefficient vectors are in flight per lane.
// (A version that marched down many rows with a rolling 3-row window was
// latency-bound: one dependent global round trip per row.)
// Batches (see kernels_setup.hpp): `ny` rows per image, `rows` stacked rows, `cpi` row
// tiles per image (tiles never straddle two images); `active` (may be null) marks the
// images still being iterated.
template <int R, bool NT, bool FMA>
__global__ __launch_bounds__(256) void k_sweep_explicit(CoefConst c, const double *__restrict__ x,
                                                        double *__restrict__ xnew, int nx, int ny,
                                                        int rows, int cpi,
                                                        const uint8_t *__restrict__ active, int gx,
                                                        int gy, int flip, double omw)
{
    int bx, by;
    if (!xcd_tile(gx, gy, flip, bx, by)) return;
    const int col = (bx * 256 + (int)threadIdx.x) * 2;
    if (col >= nx) return;
    const int img = by / cpi;
    if (active && !active[img]) return;
    const int r0 = img * ny + (by - img * cpi) * R;
    const int rlim = (img + 1) * ny;
    const size_t n = (size_t)nx * rows;
    const size_t p0 = (size_t)r0 * nx + col;
    const double2 zero = make_double2(0.0, 0.0);

    double2 xr[R + 2];
    double xw[R], xe[R];
    xr[0] = (p0 >= (size_t)nx) ? ld2(x + p0 - nx) : zero;
#pragma unroll
    for (int k = 0; k <= R; ++k) {
        const size_t p = p0 + (size_t)k * nx;
        xr[k + 1] = (p < n) ? ld2(x + p) : zero;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const size_t p = p0 + (size_t)k * nx;
        xw[k] = (p >= 1 && p < n) ? x[p - 1] : 0.0;
        xe[k] = (p + 2 < n) ? x[p + 2] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (r0 + k >= rlim) break;
        const size_t p = p0 + (size_t)k * nx;
        const double2 c0 = ldc2<NT>(c.c0 + p), aW = ldc2<NT>(c.aW + p), aE = ldc2<NT>(c.aE + p);
        const double2 aS = ldc2<NT>(c.aS + p), aN = ldc2<NT>(c.aN + p), b = ldc2<NT>(c.b + p);
        const double2 xm = xr[k], xc = xr[k + 1], xp = xr[k + 2];
        double2 o;
        o.x = jacobi_cell<FMA>(c0.x, aW.x, aE.x, aS.x, aN.x, b.x, xc.x, xw[k], xc.y, xp.x, xm.x, omw);
        o.y = jacobi_cell<FMA>(c0.y, aW.y, aE.y, aS.y, aN.y, b.y, xc.y, xc.x, xe[k], xp.y, xm.y, omw);
        st2(xnew + p, o);
    }
}

// -------------------------------------------------------- matrix-free -----

// One cell from the row dictionary (lut_layout.hpp); `off` is the cell's code = byte offset of
// its row inside a plane.
template <bool FMA>
__device__ __forceinline__ double jacobi_cell_lut(const double *lut, unsigned off, double xc, double xw,
                                                  double xe, double xs, double xn, double omw)
{
    const char *base = reinterpret_cast<const char *>(lut) + off;
    constexpr int PS = LUT_PLANE_STRIDE * 8;
    return jacobi_cell<FMA>(*reinterpret_cast<const double *>(base), *reinterpret_cast<const double *>(base + PS),
                            *reinterpret_cast<const double *>(base + 2 * PS), *reinterpret_cast<const double *>(base + 3 * PS),
                            *reinterpret_cast<const double *>(base + 4 * PS), *reinterpret_cast<const double *>(base + 5 * PS),
                            xc, xw, xe, xs, xn, omw);
}

// Copy the first `nrows` rows of every plane of the dictionary into LDS (workgroup-wide).
// (NTHREADS is a compile-time constant on purpose: with blockDim.x in the loop the streaming temporally blocked kernel
// came out 22 VGPRs heavier -- 177 instead of 155 at T = 8, one wave per SIMD less.)
template <int NTHREADS = 256>
__device__ __forceinline__ void load_lut(double *lut, const double *__restrict__ lut_g, int nrows)
{
    for (int k = threadIdx.x; k < LUT_PLANES * nrows; k += NTHREADS) {
        const int pl = k / nrows, r = k - pl * nrows;
        lut[pl * LUT_PLANE_STRIDE + r] = lut_g[pl * LUT_PLANE_STRIDE + r];
    }
    __syncthreads();
}

// Tile -> XCD map for persistent grids: workgroup `wg` of `nwg` walks tiles
// t = wg, wg + nwg, ...; tiles are numbered so that the tiles of one XCD (wg & 7)
// form a contiguous column-major run (same idea as xcd_tile()).
__device__ __forceinline__ void tile_coords(unsigned t, int gy, int &bx, int &by)
{
    bx = (int)(t / (unsigned)gy);
    by = (int)(t % (unsigned)gy);
}

// VEC = 2 (the form in use: array rows are always even, an odd mesh width is padded); VEC = 1 is the
// one-cell-per-thread form, no longer instantiated.  Tile = 256*VEC columns x R
// rows.  Position classes are folded into the codes at assembly, the kernel only looks rows up.  Persistent: the grid is a few workgroups per CU, each loads the tables
// into LDS once and then walks its share of the tiles.
template <int VEC, int R, bool FMA>
__global__ __launch_bounds__(256) void k_sweep_matfree(const double *__restrict__ lut_g,
                                                       const uint16_t *__restrict__ code,
                                                       const double *__restrict__ x,
                                                       double *__restrict__ xnew, int nx, int ny,
                                                       int rows, int cpi,
                                                       const uint8_t *__restrict__ active, int gx,
                                                       int gy, int flip, int nrows, double omw)
{
    __shared__ double lut[LUT_DOUBLES];
    load_lut(lut, lut_g, nrows);

    const size_t n = (size_t)nx * rows;
    const unsigned total = (unsigned)gx * (unsigned)gy;
    const unsigned per = (total + 7u) / 8u;              // tiles per XCD
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned nper = gridDim.x >> 3;                // workgroups per XCD (grid is a multiple of 8)
    for (unsigned k = blockIdx.x >> 3; k < per; k += nper) {
        const unsigned t = xcd * per + (flip ? per - 1u - k : k);
        if (t >= total) continue;
        int bx, by;
        tile_coords(t, gy, bx, by);
        const int col = (bx * 256 + (int)threadIdx.x) * VEC;
        if (col >= nx) continue;
        const int img = by / cpi;
        if (active && !active[img]) continue;
        const int row_lo = img * ny;                      // first stacked row of this image
        const int r0 = row_lo + (by - img * cpi) * R;
        const int rlim = row_lo + ny;
        const size_t p0 = (size_t)r0 * nx + col;

        if constexpr (VEC == 2) {
            const double2 zero = make_double2(0.0, 0.0);
            double2 xr[R + 2];
            double xw[R], xe[R];
            unsigned cc[R];
            xr[0] = (p0 >= (size_t)nx) ? ld2(x + p0 - nx) : zero;
#pragma unroll
            for (int q = 0; q <= R; ++q) {
                const size_t p = p0 + (size_t)q * nx;
                xr[q + 1] = (p < n) ? ld2(x + p) : zero;
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const size_t p = p0 + (size_t)q * nx;
                xw[q] = (p >= 1 && p < n) ? x[p - 1] : 0.0;
                xe[q] = (p + 2 < n) ? x[p + 2] : 0.0;
                cc[q] = (p < n) ? *reinterpret_cast<const uint32_t *>(code + p) : 0u;   // two 16-bit codes
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int r = r0 + q;
                if (r >= rlim) break;
                const size_t p = p0 + (size_t)q * nx;
                const unsigned i0 = cc[q] & 0xFFFFu, i1 = cc[q] >> 16;
                const double2 xm = xr[q], xc = xr[q + 1], xp = xr[q + 2];
                double2 o;
                o.x = jacobi_cell_lut<FMA>(lut, i0, xc.x, xw[q], xc.y, xp.x, xm.x, omw);
                o.y = jacobi_cell_lut<FMA>(lut, i1, xc.y, xc.x, xe[q], xp.y, xm.y, omw);
                st2(xnew + p, o);
            }
        } else {
            double xr[R + 2], xw[R], xe[R];
            unsigned cc[R];
            xr[0] = (p0 >= (size_t)nx) ? x[p0 - nx] : 0.0;
#pragma unroll
            for (int q = 0; q <= R; ++q) {
                const size_t p = p0 + (size_t)q * nx;
                xr[q + 1] = (p < n) ? x[p] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const size_t p = p0 + (size_t)q * nx;
                xw[q] = (p >= 1 && p < n) ? x[p - 1] : 0.0;
                xe[q] = (p + 1 < n) ? x[p + 1] : 0.0;
                cc[q] = (p < n) ? code[p] : 0u;
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int r = r0 + q;
                if (r >= rlim) break;
                const size_t p = p0 + (size_t)q * nx;
                const unsigned i0 = cc[q];
                xnew[p] = jacobi_cell_lut<FMA>(lut, i0, xr[q + 1], xw[q], xe[q], xr[q + 2], xr[q], omw);
            }
        }
    }
}

}  // namespace deff
