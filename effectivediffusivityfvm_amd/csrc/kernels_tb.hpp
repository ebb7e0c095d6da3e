// kernels_tb.hpp -- matrix-free sweeps with temporal blocking: T weighted-Jacobi
// sweeps per pass over HBM (gfx950, wave64, FP64).
//
// Legal because the reference only inspects the field every 10 000 sweeps
// (Deff2DGPU/Deff2D.cuh:1243); each of the T sweeps is the same updateX_SOR
// arithmetic (cuh:69-92) as the single-sweep kernels, so results stay
// bit-identical -- a cell's value after sweep k does not depend on which kernel
// produced it.
//
// Structure: every WAVE is independent.  A wave owns a strip of 128 columns
// (2 per lane) and streams down the rows of its chunk.  Per input row it
//   level 0   loads the row of x (16 B per lane, coalesced) and its phase codes,
//   level t   (t = 1..T) computes row r-t of sweep t from the three newest rows
//             of sweep t-1, all held in registers (a 3-row window per level);
//             W/E neighbours come from the adjacent lanes by DPP wave shifts,
//             N/S from the window; coefficients from the LDS lookup tables,
//   level T   is stored (16 B per lane).
// After t sweeps the outermost t columns/rows of a strip are stale, so a strip
// produces 128 - 2T valid columns and needs T extra rows above and below its
// chunk: neighbouring strips overlap by 2T and recompute the overlap instead of
// synchronising.  No barrier, no inter-wave traffic inside the row loop.
//
// HBM traffic per cell per sweep: (8 + 1) / T / efficiency read + 8 / T
// written -- ~5 B at T = 4 against 17 B for the single-sweep matrix-free
// kernel and 64 B for explicit coefficients; the kernel is VALU/LDS-bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_sweep.hpp"

namespace deff {

// Tables for this kernel: 6 planes x 16 position classes (ycls*4 + xcls, class 3
// = outside the mesh: all zeros, so such cells stay exactly 0) x 32 codes.
// The plane stride is padded to 520 doubles on purpose: with a stride that is a
// multiple of 512 B hipcc fuses the six per-cell lookups into ds_read2st64_b64,
// which banks on 32 banks (2-way conflicts on a 32-entry x 8-B group, 16 LDS
// cycles per instruction, measured as THE bottleneck of this kernel); with 4160 B
// neither ds_read2 form can encode the offset, the lookups stay plain
// ds_read_b64 (64 banks: a 256-B group is conflict-free, 2 cycles each).
constexpr int TB_CLASSES = 16;
constexpr int TB_PLANE_STRIDE = TB_CLASSES * LUT_CODES + 8;    // 520 doubles
constexpr int TB_LUT_DOUBLES = LUT_PLANES * TB_PLANE_STRIDE;   // 3120 doubles = 24.4 KiB

// lane i <- lane i-1 (lane 0 <- 0.0)
__device__ __forceinline__ double from_lane_below(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);   // bound_ctrl: lane 0 <- 0
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lane i <- lane i+1 (lane 63 <- 0.0)
__device__ __forceinline__ double from_lane_above(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// One cell.  `off` is the BYTE offset of the cell's entry in plane 0 (position
// class group + pre-scaled phase code); the other planes sit at fixed strides.
// GUARD = reference's non-zero test on every link (needed when a phase has zero
// diffusivity: links are -0.0 and neighbours may hold NaN/Inf, cuh:77).  Without
// it a zero link multiplies a finite value and adds +-0, which leaves sigma
// unchanged, so both forms give the same bits.  WALL = the strip contains the
// first or last column; everywhere else b is identically +0.0 and its lookup is
// skipped (0.0 - sigma is still evaluated as a subtraction, so the bits match).
template <bool GUARD, bool WALL>
__device__ __forceinline__ double tb_cell(const double *lut, unsigned off, double xc, double xw, double xe,
                                          double xs, double xn, double omw)
{
    const char *base = reinterpret_cast<const char *>(lut) + off;
    constexpr int PS = TB_PLANE_STRIDE * 8;
    const double c0 = *reinterpret_cast<const double *>(base);
    const double aW = *reinterpret_cast<const double *>(base + PS);
    const double aE = *reinterpret_cast<const double *>(base + 2 * PS);
    const double aS = *reinterpret_cast<const double *>(base + 3 * PS);
    const double aN = *reinterpret_cast<const double *>(base + 4 * PS);
    double b = 0.0;
    if constexpr (WALL) b = *reinterpret_cast<const double *>(base + 5 * PS);
    if constexpr (GUARD) {
        return jacobi_cell(c0, aW, aE, aS, aN, b, xc, xw, xe, xs, xn, omw);
    } else {
        // the reference starts from sigma = 0 (cuh:74); 0 + p differs from p only in the sign of a
        // zero, which b - sigma cannot see (b is +0 or non-zero), so the leading add is dropped
        double sigma = aW * xw;
        sigma += aE * xe;
        sigma += aS * xs;
        sigma += aN * xn;
        return omw * xc + c0 * (b - sigma);
    }
}

__device__ __forceinline__ int tb_ycls(int r, int ny)
{
    return (r < 0 || r >= ny) ? 3 : (r == 0 ? 1 : (r == ny - 1 ? 2 : 0));
}

// One strip x chunk: the whole row pipeline of a wave.  CPL = cells per lane: the
// strip is 64*CPL columns wide and produces 64*CPL - 2T valid columns.  Written for
// any even CPL; only CPL = 2 is instantiated (4 was measured slower, see deff_amd.hip).
template <int T, int CPL, bool GUARD, bool WALL>
__device__ __forceinline__ void tb_strip(const double *lut, const uint8_t *__restrict__ code,
                                         const double *__restrict__ x, double *__restrict__ xnew, int nx,
                                         int ny, int tx, int ry0, int LY, int lane, double omw)
{
    constexpr int NP = CPL / 2;                    // 16-B pairs per lane
    constexpr int WOUT = 64 * CPL - 2 * T;
    const int cx0 = tx * WOUT;                     // first output column of the strip
    const int col = cx0 - T + CPL * lane;          // this lane's first column (even)
    const int ry1 = min(ry0 + LY, ny);
    const int r_begin = ry0 - T, r_end = ry1 + T;  // input rows [r_begin, r_end)
    bool in_x[NP], st_x[NP];
    unsigned xoff[CPL];                            // byte offset of each cell's x position class group
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int c = col + 2 * q;
        in_x[q] = (c >= 0) && (c < nx);            // nx even => c+1 < nx too
        st_x[q] = in_x[q] && (c >= cx0) && (c < cx0 + WOUT);
        xoff[2 * q] = (unsigned)(!in_x[q] ? 3 : (c == 0 ? 1 : 0)) * (LUT_CODES * 8);
        xoff[2 * q + 1] = (unsigned)(!in_x[q] ? 3 : (c + 1 == nx - 1 ? 2 : 0)) * (LUT_CODES * 8);
    }
    const double2 zero = make_double2(0.0, 0.0);

    double w[T][3][CPL];                           // w[t][slot]: 3 newest rows of sweep t
    unsigned cw[T + 1][NP];                        // cw[t]: codes of row rr-t (2 per 16-bit pair)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int q = 0; q < CPL; ++q) w[t][sl][q] = 0.0;
#pragma unroll
    for (int t = 0; t <= T; ++t)
#pragma unroll
        for (int q = 0; q < NP; ++q) cw[t][q] = 0u;

    double2 nx_x[3][NP];
    unsigned nx_c[3][NP];
    auto fetch = [&](int rr, double2 *vx, unsigned *vc) {
        const bool row_ok = rr >= 0 && rr < ny && rr < r_end;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const bool ok = row_ok && in_x[q];
            const size_t p = (size_t)(ok ? rr : 0) * nx + (ok ? col + 2 * q : 0);
            vx[q] = ok ? ld2(x + p) : zero;
            vc[q] = ok ? (unsigned)*reinterpret_cast<const uint16_t *>(code + p) : 0u;
        }
    };
    // prefetch the first group of three rows
#pragma unroll
    for (int k = 0; k < 3; ++k) fetch(r_begin + k, nx_x[k], nx_c[k]);

    // (A skewed variant -- sweep t working from the previous step's rows so that the T updates
    // of a step are independent -- was measured: hipcc hoists every lookup, 198 VGPRs at T = 4,
    // 2 waves per SIMD, no faster; constrained to 128 VGPRs it spills.  Not kept.)
    for (int r = r_begin; r < r_end; r += 3) {
        double2 cur_x[3][NP];
        unsigned cur_c[3][NP];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < NP; ++q) { cur_x[k][q] = nx_x[k][q]; cur_c[k][q] = nx_c[k][q]; }
        // issue the next group's loads before working on this one
#pragma unroll
        for (int k = 0; k < 3; ++k) fetch(r + 3 + k, nx_x[k], nx_c[k]);
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int rr = r + ph;                 // input row of this step
            // after this step's level-(t-1) write: newest = slot ph, previous = (ph+2)%3, oldest = (ph+1)%3
            const int sN = (ph + 1) % 3, sC = (ph + 2) % 3, sS = ph;
#pragma unroll
            for (int t = T; t >= 1; --t)
#pragma unroll
                for (int q = 0; q < NP; ++q) cw[t][q] = cw[t - 1][q];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                cw[0][q] = cur_c[ph][q];
                w[0][sS][2 * q] = cur_x[ph][q].x;
                w[0][sS][2 * q + 1] = cur_x[ph][q].y;
            }
#pragma unroll
            for (int t = 1; t <= T; ++t) {
                const int rt = rr - t;             // row produced by sweep t in this step
                const double *vN = w[t - 1][sN], *vC = w[t - 1][sC], *vS = w[t - 1][sS];
                const double xw_first = from_lane_below(vC[CPL - 1]);
                const double xe_last = from_lane_above(vC[0]);
                // byte offset = class group (multiple of 256) | pre-scaled code (< 256)
                const unsigned ybase = (unsigned)tb_ycls(rt, ny) * (4 * LUT_CODES * 8);
                double o[CPL];
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const unsigned cbits = (q & 1) ? (cw[t][q >> 1] >> 8) : cw[t][q >> 1];
                    const unsigned off = (cbits & 0xF8u) | (ybase + xoff[q]);
                    const double xw = (q == 0) ? xw_first : vC[q - 1];
                    const double xe = (q == CPL - 1) ? xe_last : vC[q + 1];
                    o[q] = tb_cell<GUARD, WALL>(lut, off, vC[q], xw, xe, vS[q], vN[q], omw);
                }
                if (t < T) {
#pragma unroll
                    for (int q = 0; q < CPL; ++q) w[t][sS][q] = o[q];
                } else if (rt >= ry0 && rt < ry1) {
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        if (st_x[q]) st2(xnew + (size_t)rt * nx + col + 2 * q, make_double2(o[2 * q], o[2 * q + 1]));
                }
            }
        }
    }
}

// grid: persistent workgroups of 4 waves; wave w of block-tile (btx, bty) owns
// strip tx = btx*4 + w and rows [bty*LY, bty*LY + LY).  nx must be even, T even.
template <int T, int CPL, bool GUARD>
__global__ __launch_bounds__(256) void k_sweep_matfree_tb(const double *__restrict__ lut_g,
                                                          const uint8_t *__restrict__ code,
                                                          const double *__restrict__ x,
                                                          double *__restrict__ xnew, int nx, int ny,
                                                          int LY, int ntx, int gx, int gy, int flip,
                                                          double omw)
{
    static_assert(T >= 1 && T <= 8 && (CPL == 2 || CPL == 4), "unsupported T / CPL");
    __shared__ double lut[TB_LUT_DOUBLES];
    for (int k = threadIdx.x; k < TB_LUT_DOUBLES; k += 256) lut[k] = lut_g[k];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned total = (unsigned)gx * (unsigned)gy;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned nper = gridDim.x >> 3;

    for (unsigned kk = blockIdx.x >> 3; kk < per; kk += nper) {
        const unsigned bt = xcd * per + (flip ? per - 1u - kk : kk);
        if (bt >= total) continue;
        const int btx = (int)(bt / (unsigned)gy), bty = (int)(bt % (unsigned)gy);
        const int tx = btx * 4 + wave;
        if (tx >= ntx) continue;                       // wave-uniform

        if (tx == 0 || tx == ntx - 1)
            tb_strip<T, CPL, GUARD, true>(lut, code, x, xnew, nx, ny, tx, bty * LY, LY, lane, omw);
        else
            tb_strip<T, CPL, GUARD, false>(lut, code, x, xnew, nx, ny, tx, bty * LY, LY, lane, omw);
    }
}

}  // namespace deff
