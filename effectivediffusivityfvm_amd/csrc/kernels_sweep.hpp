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
// Neighbour addressing is the reference's linear one (x[p-1], x[p+1],
// x[p+nx], x[p-nx]); a neighbour whose linear index falls outside [0, n) is
// never dereferenced (the reference relies on the zero coefficient there).
//
// Kernels:
//   k_sweep_scalar    1 cell / thread, SoA coefficients, any nx
//   k_sweep_explicit  2 cells / thread (16-B accesses), SoA coefficient
//                     streams, each workgroup marches down `rows` rows keeping
//                     the three x rows in registers: 64 B/cell/sweep of HBM
//   k_sweep_matfree   coefficients looked up in an LDS table from a 1-byte
//                     phase code: 17 B/cell/sweep of HBM
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace deff {

struct CoefConst {
    const double *c0, *aW, *aE, *aS, *aN, *b;
};

__device__ __forceinline__ double jacobi_cell(double c0, double aW, double aE, double aS, double aN,
                                              double b, double xc, double xw, double xe, double xs,
                                              double xn, double omw)
{
    double sigma = 0;
    if (aW != 0) sigma += aW * xw;
    if (aE != 0) sigma += aE * xe;
    if (aS != 0) sigma += aS * xs;
    if (aN != 0) sigma += aN * xn;
    return omw * xc + c0 * (b - sigma);
}

// Workgroup -> tile map.  Workgroups are dealt round-robin over the 8 XCDs, so
// blocks id and id+8 share an L2.  Give every XCD a contiguous run of tiles in
// column-major order (tiles above/below each other, which share halo rows of
// x, then land in the same L2).  Placement only affects speed, never results.
__device__ __forceinline__ bool xcd_tile(int gx, int gy, int &bx, int &by)
{
    const unsigned total = (unsigned)gx * (unsigned)gy;
    const unsigned per = (total + 7u) / 8u;
    const unsigned id = blockIdx.x;
    const unsigned t = (id & 7u) * per + (id >> 3);
    if ((id >> 3) >= per || t >= total) return false;
    bx = (int)(t / (unsigned)gy);
    by = (int)(t % (unsigned)gy);
    return true;
}

// ------------------------------------------------------------- scalar -----

__global__ __launch_bounds__(256) void k_sweep_scalar(CoefConst c, const double *__restrict__ x,
                                                      double *__restrict__ xnew, int nx, size_t n,
                                                      double omw)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const double xw = (p >= 1) ? x[p - 1] : 0.0;
    const double xe = (p + 1 < n) ? x[p + 1] : 0.0;
    const double xs = (p + nx < n) ? x[p + nx] : 0.0;
    const double xn = (p >= (size_t)nx) ? x[p - nx] : 0.0;
    xnew[p] = jacobi_cell(c.c0[p], c.aW[p], c.aE[p], c.aS[p], c.aN[p], c.b[p], x[p], xw, xe, xs, xn, omw);
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

// Requires nx even (16-B aligned row starts).  Tile = 512 columns x `rows` rows.
__global__ __launch_bounds__(256) void k_sweep_explicit(CoefConst c, const double *__restrict__ x,
                                                        double *__restrict__ xnew, int nx, int ny,
                                                        int rows, int gx, int gy, double omw)
{
    int bx, by;
    if (!xcd_tile(gx, gy, bx, by)) return;
    const int col = (bx * 256 + (int)threadIdx.x) * 2;
    if (col >= nx) return;
    const int r0 = by * rows;
    const int r1 = min(r0 + rows, ny);
    const size_t n = (size_t)nx * ny;
    size_t p = (size_t)r0 * nx + col;

    const double2 zero = make_double2(0.0, 0.0);
    double2 xm = (p >= (size_t)nx) ? ld2(x + p - nx) : zero;
    double2 xc = ld2(x + p);
    for (int r = r0; r < r1; ++r, p += nx) {
        const double2 xp = (p + nx < n) ? ld2(x + p + nx) : zero;
        const double xw = (p >= 1) ? x[p - 1] : 0.0;
        const double xe = (p + 2 < n) ? x[p + 2] : 0.0;
        const double2 c0 = ld2(c.c0 + p), aW = ld2(c.aW + p), aE = ld2(c.aE + p);
        const double2 aS = ld2(c.aS + p), aN = ld2(c.aN + p), b = ld2(c.b + p);
        double2 o;
        o.x = jacobi_cell(c0.x, aW.x, aE.x, aS.x, aN.x, b.x, xc.x, xw, xc.y, xp.x, xm.x, omw);
        o.y = jacobi_cell(c0.y, aW.y, aE.y, aS.y, aN.y, b.y, xc.y, xc.x, xe, xp.y, xm.y, omw);
        st2(xnew + p, o);
        xm = xc;
        xc = xp;
    }
}

// -------------------------------------------------------- matrix-free -----

// Lookup tables: 6 planes (c0, aW, aE, aS, aN, b) x 9 position classes
// (ypos*3 + xpos) x 32 phase codes.  A 32-entry x 8-B group is exactly one
// 256-B LDS bank row, so ds_read_b64 with per-lane codes is conflict-free
// whenever all lanes of a half-wave share the position class.
constexpr int LUT_CODES = 32;
constexpr int LUT_CLASSES = 9;
constexpr int LUT_PLANES = 6;
constexpr int LUT_PLANE_STRIDE = LUT_CLASSES * LUT_CODES;           // doubles
constexpr int LUT_DOUBLES = LUT_PLANES * LUT_PLANE_STRIDE;          // 1728

__device__ __forceinline__ double jacobi_cell_lut(const double *lut, int idx, double xc, double xw,
                                                  double xe, double xs, double xn, double omw)
{
    return jacobi_cell(lut[idx], lut[idx + LUT_PLANE_STRIDE], lut[idx + 2 * LUT_PLANE_STRIDE],
                       lut[idx + 3 * LUT_PLANE_STRIDE], lut[idx + 4 * LUT_PLANE_STRIDE],
                       lut[idx + 5 * LUT_PLANE_STRIDE], xc, xw, xe, xs, xn, omw);
}

// VEC = 2 needs nx even; VEC = 1 handles any nx.  Tile = 256*VEC columns x rows.
template <int VEC>
__global__ __launch_bounds__(256) void k_sweep_matfree(const double *__restrict__ lut_g,
                                                       const uint8_t *__restrict__ code,
                                                       const double *__restrict__ x,
                                                       double *__restrict__ xnew, int nx, int ny,
                                                       int rows, int gx, int gy, double omw)
{
    __shared__ double lut[LUT_DOUBLES];
    for (int k = threadIdx.x; k < LUT_DOUBLES; k += 256) lut[k] = lut_g[k];
    __syncthreads();

    int bx, by;
    if (!xcd_tile(gx, gy, bx, by)) return;
    const int col = (bx * 256 + (int)threadIdx.x) * VEC;
    if (col >= nx) return;
    const int r0 = by * rows;
    const int r1 = min(r0 + rows, ny);
    const size_t n = (size_t)nx * ny;
    size_t p = (size_t)r0 * nx + col;

    if constexpr (VEC == 2) {
        const int xcls0 = (col == 0) ? 1 : 0;                 // cell 0 can only be the first column
        const int xcls1 = (col + 1 == nx - 1) ? 2 : 0;        // cell 1 can only be the last column
        const double2 zero = make_double2(0.0, 0.0);
        double2 xm = (p >= (size_t)nx) ? ld2(x + p - nx) : zero;
        double2 xc = ld2(x + p);
        for (int r = r0; r < r1; ++r, p += nx) {
            const double2 xp = (p + nx < n) ? ld2(x + p + nx) : zero;
            const double xw = (p >= 1) ? x[p - 1] : 0.0;
            const double xe = (p + 2 < n) ? x[p + 2] : 0.0;
            const unsigned cc = *reinterpret_cast<const uint16_t *>(code + p);
            const int ycls = (r == 0) ? 1 : (r == ny - 1 ? 2 : 0);
            const int i0 = (ycls * 3 + xcls0) * LUT_CODES + (int)(cc & 31u);
            const int i1 = (ycls * 3 + xcls1) * LUT_CODES + (int)((cc >> 8) & 31u);
            double2 o;
            o.x = jacobi_cell_lut(lut, i0, xc.x, xw, xc.y, xp.x, xm.x, omw);
            o.y = jacobi_cell_lut(lut, i1, xc.y, xc.x, xe, xp.y, xm.y, omw);
            st2(xnew + p, o);
            xm = xc;
            xc = xp;
        }
    } else {
        const int xcls = (col == 0) ? 1 : (col == nx - 1 ? 2 : 0);
        double xm = (p >= (size_t)nx) ? x[p - nx] : 0.0;
        double xc = x[p];
        for (int r = r0; r < r1; ++r, p += nx) {
            const double xp = (p + nx < n) ? x[p + nx] : 0.0;
            const double xw = (p >= 1) ? x[p - 1] : 0.0;
            const double xe = (p + 1 < n) ? x[p + 1] : 0.0;
            const int ycls = (r == 0) ? 1 : (r == ny - 1 ? 2 : 0);
            const int i0 = (ycls * 3 + xcls) * LUT_CODES + (int)(code[p] & 31u);
            xnew[p] = jacobi_cell_lut(lut, i0, xc, xw, xe, xp, xm, omw);
            xm = xc;
            xc = xp;
        }
    }
}

}  // namespace deff
