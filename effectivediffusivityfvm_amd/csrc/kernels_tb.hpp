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
//   level 0   loads the row of x (16 B per lane, coalesced) and its row codes (4 B per lane),
//   level t   (t = 1..T) computes row r-t of sweep t from the three newest rows
//             of sweep t-1, all held in registers (a 3-row window per level);
//             W/E neighbours come from the adjacent lanes by DPP wave shifts,
//             N/S from the window; coefficients from the row dictionary in LDS,
//   level T   is stored (16 B per lane).
// After t sweeps the outermost t columns/rows of a strip are stale, so a strip
// produces 128 - 2T valid columns and needs T extra rows above and below its
// chunk: neighbouring strips overlap by 2T and recompute the overlap instead of
// synchronising.  No barrier, no inter-wave traffic inside the row loop.
//
// HBM traffic per cell per sweep: (8 + 2) / T / efficiency read + 8 / T
// written -- 2.7 B measured at T = 8 against 18 B for the single-sweep
// matrix-free kernel and 64 B for explicit coefficients; the kernel is bound by
// how many waves of a SIMD are ready to issue FP64 (SQ counters: VALU ~50-60 % busy, LDS ~40 %;
// see the notes at tb_strip and DESIGN.md section 4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_sweep.hpp"

namespace deff {

constexpr int TB_COLS = 128;                                   // columns per wave strip (2 per lane)
template <bool V> struct TbTag { static constexpr bool value = V; };   // compile-time flag for generic lambdas

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

// One cell.  `off` is the cell's code: the BYTE offset of its row in plane 0 of the row
// dictionary (lut_layout.hpp); the other planes sit at fixed strides.
// GUARD = reference's non-zero test on every link (needed when a phase has zero
// diffusivity: links are -0.0 and neighbours may hold NaN/Inf, cuh:77).  Without
// it a zero link multiplies a finite value and adds +-0, which leaves sigma
// unchanged, so both forms give the same bits.  WALL = the strip contains the
// first or last column; everywhere else b is identically +0.0 and its lookup is
// skipped (0.0 - sigma is still evaluated as a subtraction, so the bits match).
template <bool GUARD, bool WALL, bool FMA>
__device__ __forceinline__ double tb_cell(const double *lut, unsigned off, double xc, double xw, double xe,
                                          double xs, double xn, double omw)
{
    const char *base = reinterpret_cast<const char *>(lut) + off;
    constexpr int PS = LUT_PLANE_STRIDE * 8;
    const double c0 = *reinterpret_cast<const double *>(base);
    const double aW = *reinterpret_cast<const double *>(base + PS);
    const double aE = *reinterpret_cast<const double *>(base + 2 * PS);
    const double aS = *reinterpret_cast<const double *>(base + 3 * PS);
    const double aN = *reinterpret_cast<const double *>(base + 4 * PS);
    double b = 0.0;
    if constexpr (WALL) b = *reinterpret_cast<const double *>(base + 5 * PS);
    if constexpr (GUARD) {
        return jacobi_cell<FMA>(c0, aW, aE, aS, aN, b, xc, xw, xe, xs, xn, omw);
    } else {
        // the reference starts from sigma = 0 (cuh:74); 0 + p differs from p only in the sign of a
        // zero, which b - sigma cannot see (b is +0 or non-zero), so the leading add is dropped
        // (in the contracted form the first term is fma(aW, xw, 0) = the rounded product, likewise)
        double sigma = aW * xw;
        sigma = mul_add<FMA>(aE, xe, sigma);
        sigma = mul_add<FMA>(aS, xs, sigma);
        sigma = mul_add<FMA>(aN, xn, sigma);
        if constexpr (FMA) return __builtin_fma(omw, xc, c0 * (b - sigma));
        else return omw * xc + c0 * (b - sigma);
    }
}

// The two cells of a lane TOGETHER, one arithmetic stage at a time: cell by cell (two tb_cell() calls) hipcc emits each
// cell's seven-deep dependent chain back to back, so that a wave has a single chain in flight; written stage-wise the two
// chains interleave and every FP64 instruction has another between itself and its consumer.  Same operations, same order
// per cell, same bits.
// the matrix rows of a lane's two cells
struct TbCoef {
    double c0[2], aW[2], aE[2], aS[2], aN[2], b[2];
};
template <bool WALL>
__device__ __forceinline__ void tb_lookup(const double *lut, unsigned o0, unsigned o1, TbCoef &k)
{
    constexpr int PS = LUT_PLANE_STRIDE * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const char *base = reinterpret_cast<const char *>(lut) + (h ? o1 : o0);
        k.aW[h] = *reinterpret_cast<const double *>(base + PS);
        k.aE[h] = *reinterpret_cast<const double *>(base + 2 * PS);
        k.aS[h] = *reinterpret_cast<const double *>(base + 3 * PS);
        k.aN[h] = *reinterpret_cast<const double *>(base + 4 * PS);
        k.c0[h] = *reinterpret_cast<const double *>(base);
        if constexpr (WALL) k.b[h] = *reinterpret_cast<const double *>(base + 5 * PS);
        else k.b[h] = 0.0;
    }
}
template <bool FMA>
__device__ __forceinline__ double2 tb_apply(const TbCoef &k, double2 vC, double xw0, double xe1, double2 vS, double2 vN, double omw)
{
    double s0 = k.aW[0] * xw0, s1 = k.aW[1] * vC.x;
    s0 = mul_add<FMA>(k.aE[0], vC.y, s0); s1 = mul_add<FMA>(k.aE[1], xe1, s1);
    s0 = mul_add<FMA>(k.aS[0], vS.x, s0); s1 = mul_add<FMA>(k.aS[1], vS.y, s1);
    s0 = mul_add<FMA>(k.aN[0], vN.x, s0); s1 = mul_add<FMA>(k.aN[1], vN.y, s1);
    s0 = k.b[0] - s0; s1 = k.b[1] - s1;
    double2 o;
    if constexpr (FMA) {
        s0 = k.c0[0] * s0; s1 = k.c0[1] * s1;
        o.x = __builtin_fma(omw, vC.x, s0); o.y = __builtin_fma(omw, vC.y, s1);
    } else {
        const double m0 = omw * vC.x, m1 = omw * vC.y;
        s0 = k.c0[0] * s0; s1 = k.c0[1] * s1;
        o.x = m0 + s0; o.y = m1 + s1;
    }
    return o;
}
template <bool GUARD, bool WALL, bool FMA>
__device__ __forceinline__ double2 tb_pair(const double *lut, unsigned o0, unsigned o1, double2 vC, double xw0, double xe1,
                                           double2 vS, double2 vN, double omw)
{
    if constexpr (GUARD) {
        double2 o;
        o.x = tb_cell<GUARD, WALL, FMA>(lut, o0, vC.x, xw0, vC.y, vS.x, vN.x, omw);
        o.y = tb_cell<GUARD, WALL, FMA>(lut, o1, vC.y, vC.x, xe1, vS.y, vN.y, omw);
        return o;
    } else {
        TbCoef k;
        tb_lookup<WALL>(lut, o0, o1, k);
        return tb_apply<FMA>(k, vC, xw0, xe1, vS, vN, omw);
    }
}


// One strip x chunk: the whole row pipeline of a wave (2 cells per lane, 128 columns).
// Geometry (array rows): the mesh of this image is rows [row_lo, row_lo+ny) -- rows outside it
// are "outside the mesh" even if another image of a batch lives there, and row_lo is negative
// for a row slab whose array is a window into a taller image; the chunk computes rows
// [ry0, min(ry0+LY, own_hi)), own_hi being the end of the rows this launch owns (the image in a
// batch, the slab's own rows -- without its halo -- in a multi-GPU run).
// The code is deliberately written with double2 values and named slots: an array-of-scalars formulation of the same
// dataflow made hipcc hoist the lookups to 204 VGPRs.  Everything that was measured on this function and dropped --
// pipelined lookups, 4 cells per lane, skewed levels, cooperating strips, one lookup per face, LDS-DMA prefetch, fences
// elsewhere, buffer addressing, a pair of waves per tile -- is recorded with its numbers in DESIGN.md section 4 and
// profiles/r03_tb_ab_kbench.log; the variants themselves are tools/experiments/r03_tb_variants.patch.
template <int T, bool GUARD, bool WALL, bool FMA>
__device__ __forceinline__ void tb_strip(const double *lut, const uint16_t *__restrict__ code,
                                         const double *__restrict__ x, double *__restrict__ xnew, int nx,
                                         int ny, int row_lo, int own_hi, int tx, int ntx, int shift, int ry0, int LY,
                                         int lane, double omw)
{
    // column halo rounded up to even so that odd T keeps the 16-B alignment of a lane's pair
    constexpr int HW = (T + 1) & ~1;
    constexpr int WOUT = TB_COLS - 2 * HW;
    // Strip tx reads columns [tx*WOUT, tx*WOUT + 128) and owns the outputs [out_lo, out_hi): its
    // window minus HW stale columns on each side that borders another strip -- a side that is a
    // wall of the mesh needs no halo, so the first strip starts at column 0 and an image of up to
    // 128 columns is a single strip with nothing recomputed.
    // (shift = HW selects the older placement with a halo also outside the first column; kept as
    // a tuning switch so that the two can be compared inside one process.)
    const int col = tx * WOUT - shift + 2 * lane;  // this lane's first column (even)
    const bool in_x = col >= 0 && col < nx;        // nx even => col+1 < nx too
    const int out_lo = (tx == 0) ? 0 : tx * WOUT - shift + HW;
    const int out_hi = (tx == ntx - 1) ? nx : tx * WOUT - shift + TB_COLS - HW;
    const int row_hi = row_lo + ny;
    const int ry1 = min(ry0 + LY, own_hi);
    // input rows [r_begin, r_end): T rows of halo above and below the chunk, except that nothing
    // lies above the first row of the mesh (the steps for rows below the last one still run: they
    // drain the pipeline)
    const int r_begin = max(ry0 - T, row_lo), r_end = ry1 + T;
    const bool st_x = in_x && (col >= out_lo) && (col < out_hi);
    const double2 zero = make_double2(0.0, 0.0);

    double2 w[T][3];                               // w[t]: 3 newest rows of sweep t
    unsigned cw[T + 1];                            // cw[t]: the two 16-bit codes of row rr-t
#pragma unroll
    for (int t = 0; t < T; ++t) { w[t][0] = zero; w[t][1] = zero; w[t][2] = zero; }
#pragma unroll
    for (int t = 0; t <= T; ++t) cw[t] = 0u;

    // prefetch the first group of three rows
    double2 nx_x[3];
    unsigned nx_c[3];
    auto fetch = [&](const int rr, double2 &vx_out, unsigned &vc_out) __attribute__((always_inline)) {
        const bool ok = in_x && rr >= row_lo && rr < row_hi && rr < r_end;
        const size_t p = (size_t)(ok ? rr : 0) * nx + (ok ? col : 0);
        // the loads are issued UNCONDITIONALLY (the address is clamped into the array) and the value is
        // selected afterwards: with `ok ? load : 0` hipcc branches around the loads, no longer knows how
        // many are in flight, and waits with vmcnt(0) right after issuing the next group's loads --
        // i.e. no prefetch at all (found by removing the loads / the store: +23 % / +30 %)
        const double2 vx = ld2(x + p);
        const unsigned vc = *reinterpret_cast<const uint32_t *>(code + p);
        vx_out = ok ? vx : zero;
        vc_out = ok ? vc : 0u;                                               // rows / lanes outside the mesh: zero row
    };
#pragma unroll
    for (int k = 0; k < 3; ++k) fetch(r_begin + k, nx_x[k], nx_c[k]);

    // (Measured with the tile time stamps, tools/tb_stamps.py: waves sharing a SIMD are served oldest-first, so identical
    // tiles end at 71 / 89 / 108 us of one T = 8 launch at 4096^2, by wave slot.  Evening that out with s_setprio -- a
    // rotating priority per group of steps -- brought +2...4 %: served in turn the three waves issue less in total than served
    // oldest-first.  What is kept is the other way round: the service order stays and the TILES differ, the oldest wave of a
    // SIMD getting the tallest chunk -- `dealt` below, deal_ranked_tiles in api_solve.hip: +6...8 %.)
    // One group of three steps (input rows r, r+1, r+2).  TRIM = the group may contain levels whose
    // output row this chunk does not need: sweep t needs rows from max(mesh top, ry0 - (T - t)) on, and
    // produces row rr - t at the step that reads row rr, so for the first 2T steps of a chunk (T at the
    // top of the mesh) part of the levels would only compute rows nothing reads -- 72 of 528 level
    // steps at T = 8 with 50-row chunks.  The trimmed groups skip them behind wave-uniform branches;
    // the steady-state loop below stays branch-free (a branch around every level of every step was
    // measured 8 % slower).  A skipped level leaves its window slot as it was: finite values that are
    // read only through zero links or not at all.
    auto group = [&](const int r, auto trim_tag) __attribute__((always_inline)) {
        constexpr bool TRIM = decltype(trim_tag)::value;
        double2 cur_x[3];
        unsigned cur_c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { cur_x[k] = nx_x[k]; cur_c[k] = nx_c[k]; }
        // issue the next group's loads before working on this one
#pragma unroll
        for (int k = 0; k < 3; ++k) fetch(r + 3 + k, nx_x[k], nx_c[k]);      // unconditional loads, see above
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int rr = r + ph;                 // input row of this step
            // after this step's level-(t-1) write: newest = slot ph, previous = (ph+2)%3, oldest = (ph+1)%3
            const int sN = (ph + 1) % 3, sC = (ph + 2) % 3, sS = ph;
#pragma unroll
            for (int t = T; t >= 1; --t) cw[t] = cw[t - 1];
            cw[0] = cur_c[ph];
            w[0][sS] = cur_x[ph];
#pragma unroll
            for (int t = 1; t <= T; ++t) {
                const int rt = rr - t;             // row produced by sweep t in this step
                if constexpr (TRIM) {
                    if (rt < row_lo || rt - t < ry0 - T) {   // wave-uniform: above the mesh / above this level's halo
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                }
                const double2 vN = w[t - 1][sN], vC = w[t - 1][sC], vS = w[t - 1][sS];
                const double xw0 = from_lane_below(vC.y);
                const double xe1 = from_lane_above(vC.x);
                const unsigned o0 = cw[t] & 0xFFFFu, o1 = cw[t] >> 16;
                const double2 o = tb_pair<GUARD, WALL, FMA>(lut, o0, o1, vC, xw0, xe1, vS, vN, omw);
                if (t == T && ph == 2) {
                    // CDNA counts loads and stores in one vmcnt and lets stores complete out of order, so
                    // "the prefetched rows have arrived" can only be expressed as vmcnt(0), which also
                    // waits for every store in flight.  Ask for the prefetched group HERE, before the
                    // group's last store goes out: the two stores then in flight are a step old (acked),
                    // whereas at the top of the next group the wait would sit right behind a fresh store
                    // (measured by deleting the store: +21..30 %).
                    asm volatile("" :: "v"(nx_x[0].x), "v"(nx_x[0].y), "v"(nx_x[1].x), "v"(nx_x[1].y), "v"(nx_x[2].x),
                                 "v"(nx_x[2].y), "v"(nx_c[0]), "v"(nx_c[1]), "v"(nx_c[2]));
                }
                if (t < T) {
                    w[t][sS] = o;
                } else if (st_x && rt >= ry0 && rt < ry1) {
                    st2(xnew + (size_t)rt * nx + col, o);
                }
                // keep the scheduler from pulling the next sweeps' table lookups up here: left
                // alone it hoists them all (180-250 VGPRs, 1-2 waves per SIMD); with the fence a
                // step keeps ~120 VGPRs and 4 waves per SIMD hide the LDS latency instead
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    constexpr int TRIMMED = ((2 * T + 2) / 3) * 3;   // steps that may hold unneeded levels, rounded up to whole groups
    int r = r_begin;
    for (; r < r_begin + TRIMMED && r < r_end; r += 3) group(r, TbTag<true>{});
    for (; r < r_end; r += 3) group(r, TbTag<false>{});
}

// grid: persistent workgroups of 4 waves.  Wave tiles (strip tx, chunk ty) are numbered
// strip-major (wt = tx*gy + ty) and dealt 4 per workgroup, so the 4 waves of a workgroup
// normally hold 4 vertically adjacent chunks of one strip (shared halo rows stay in L1/L2)
// and no wave idles because the strip count is not a multiple of 4.  `gx` = number of
// workgroup tiles = ceil(ntx*gy / 4); gy = nimg * cpi chunks.  nx must be even.
// Image k of a stack: mesh rows [dom_lo + k*img_stride, ... + ny), owned rows
// [own_lo + k*img_stride, ... + own_h).  Batch: dom_lo = own_lo = 0, own_h = img_stride = ny.
// Row slab (one image): dom_lo = -(first array row's global index), ny = global height,
// own_lo = halo depth, own_h = rows owned by this rank.
template <int T, bool FMA, bool GUARD>
__global__ __launch_bounds__(256, (T >= 6 ? 3 : 1)) void k_sweep_matfree_tb(const double *__restrict__ lut_g,
                                                          const uint16_t *__restrict__ code,
                                                          const double *__restrict__ x,
                                                          double *__restrict__ xnew, int nx, int ny,
                                                          int img_stride, int dom_lo, int own_lo,
                                                          int own_h, int cpi,
                                                          const uint8_t *__restrict__ active,
                                                          int LY, int ntx, int nbt, int gy, int flip,
                                                          int xmajor, int allb, int nrows, int shift,
                                                          double omw, unsigned long long *__restrict__ stamps,
                                                          const int4 *__restrict__ dealt)
{
    static_assert(T >= 1 && T <= 8, "unsupported T");
    __shared__ double lut[LUT_DOUBLES];
    load_lut(lut, lut_g, nrows);

    const int lane = threadIdx.x & 63;
    // readfirstlane makes the wave index (and everything derived from it: strip, chunk, row
    // classes, loop bounds) provably wave-uniform, i.e. SGPR/SALU work; left as a VGPR it
    // costs ~80 VGPRs of per-lane copies of scalars
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned total = (unsigned)nbt;              // workgroup tiles
    const unsigned wtiles = (unsigned)ntx * (unsigned)gy;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned nper = gridDim.x >> 3;

    // diagnostics only (tools/tb_stamps.py): wall-clock start / end of every wave tile; the buffer is
    // written by lane 0 after the tile and read by nobody on the device
    const unsigned long long t_begin = stamps ? wall_clock64() : 0ull;
    if (dealt) {
        // Dealt tiles (one tile per wave; plan_streaming / deal_ranked_tiles in api_solve.hip): wave `wave` of workgroup
        // blockIdx.x runs the chunk the host wrote at dealt[4 * blockIdx.x + wave] = (strip | image << 16, first row, rows, stamp index) --
        // chunk heights then follow the order in which a SIMD serves its waves.  rows = 0: nothing for this wave.
        const int4 d = dealt[(size_t)blockIdx.x * 4u + (unsigned)wave];
        if (d.z <= 0) return;
        // the deal rests on an observation -- workgroup (blockIdx >> 3) / (CUs per XCD) of an XCD lands in wave slot 0, 1, 2 of its
        // SIMDs -- and watches it: a wave that finds itself in another slot than its tile was cut for counts itself in the word
        // behind the table; the host reads the count after the first launches of a plan and goes back to equal chunks if the
        // chip is not dispatching that way (another process on the GPU).  d.w = stamp index | slot << 30.
        if (lane == 0 && __builtin_amdgcn_s_getreg(0x1804) != ((unsigned)d.w >> 30))
            atomicAdd(reinterpret_cast<unsigned *>(const_cast<int4 *>(dealt) + (size_t)gridDim.x * 4u), 1u);
        const int tx = d.x & 0xFFFF, img = d.x >> 16;      // image of a stack
        if (active && !active[img]) return;
        const int row_lo = dom_lo + img * img_stride, own_hi = own_lo + img * img_stride + own_h;
        if (allb || tx == 0 || tx == ntx - 1)
            tb_strip<T, GUARD, true, FMA>(lut, code, x, xnew, nx, ny, row_lo, own_hi, tx, ntx, shift, d.y, d.z, lane, omw);
        else
            tb_strip<T, GUARD, false, FMA>(lut, code, x, xnew, nx, ny, row_lo, own_hi, tx, ntx, shift, d.y, d.z, lane, omw);
        if (stamps && lane == 0) {
            const unsigned long long where = (unsigned long long)(__builtin_amdgcn_s_getreg(0xF804) & 0xFFFFu) |
                                             ((unsigned long long)(__builtin_amdgcn_s_getreg(0xF814) & 0xFu) << 16);
            const size_t si = (size_t)(d.w & 0x3FFFFFFF);
            stamps[2 * si] = t_begin;
            stamps[2 * si + 1] = ((wall_clock64() - t_begin) & 0xFFFFFFFFull) | (where << 32);
        }
        return;
    }
    for (unsigned kk = blockIdx.x >> 3; kk < per; kk += nper) {
        const unsigned bt = xcd * per + (flip ? per - 1u - kk : kk);
        if (bt >= total) continue;
        const unsigned wt = bt * 4u + (unsigned)wave;
        if (wt >= wtiles) continue;                    // wave-uniform
        // strip-major: 4 waves = 4 stacked chunks of one strip; x-major: 4 neighbouring strips
        const int tx = xmajor ? (int)(wt % (unsigned)ntx) : (int)(wt / (unsigned)gy);
        const int bty = xmajor ? (int)(wt / (unsigned)ntx) : (int)(wt % (unsigned)gy);
        const int img = bty / cpi;                     // cpi chunks per image, gy = nimg * cpi
        if (active && !active[img]) continue;          // frozen image of a batch
        const int row_lo = dom_lo + img * img_stride;
        const int own0 = own_lo + img * img_stride;
        const int ry0 = own0 + (bty - img * cpi) * LY;

        // b is read only where it can be non-zero: strips holding a wall column, or everywhere for a
        // harvested dictionary whose right-hand side is not confined to the walls
        if (allb || tx == 0 || tx == ntx - 1)
            tb_strip<T, GUARD, true, FMA>(lut, code, x, xnew, nx, ny, row_lo, own0 + own_h, tx, ntx, shift, ry0, LY, lane, omw);
        else
            tb_strip<T, GUARD, false, FMA>(lut, code, x, xnew, nx, ny, row_lo, own0 + own_h, tx, ntx, shift, ry0, LY, lane, omw);
        if (stamps && lane == 0) {
            // end stamp: low 32 bits = duration in 10-ns ticks, bits 32...47 = HW_ID (wave slot, SIMD, CU, SE), 48...51 = XCC
            const unsigned long long where = (unsigned long long)(__builtin_amdgcn_s_getreg(0xF804) & 0xFFFFu) |
                                             ((unsigned long long)(__builtin_amdgcn_s_getreg(0xF814) & 0xFu) << 16);
            stamps[2 * (size_t)wt] = t_begin;
            stamps[2 * (size_t)wt + 1] = ((wall_clock64() - t_begin) & 0xFFFFFFFFull) | (where << 32);
        }
    }
}

}  // namespace deff
