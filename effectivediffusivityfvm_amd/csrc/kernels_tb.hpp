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

#ifndef TB_TOUCH_PH
#define TB_TOUCH_PH 2
#endif
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
// per cell, same bits.  (TB_PAIR = 0 restores the cell-by-cell form for A/B runs.)
#ifndef TB_PAIR
#define TB_PAIR 1
#endif
// A/B switches of round 3 (tools/build_variant.sh):
//   TB_FENCE 0 = scheduling fence behind every level (the default since round 1), 1 = in the MIDDLE of every level, behind
//   the four sigma stages: the next level's lookups and lane shifts may then be hoisted over this level's tail (b - sigma,
//   the two products, the sum) into the registers of the links that have just died, 2 = both.
#ifndef TB_FENCE
#define TB_FENCE 0
#endif
//   TB_BUF 1 = rows are addressed as buffer base (per wave tile, SGPRs) + lane offset (one VGPR) + row offset (an SGPR)
//   instead of 64-bit pointers computed per row in VALU (10 VALU instructions per step, 12 VGPRs of addresses).
//   TB_SPLIT 1 = the two 16-bit codes of a lane's cells travel down the levels as two registers, split once per input row,
//   instead of one register split at every level (2 VALU instructions per level saved, T + 1 VGPRs spent).
//   Measured at 4096^2, one process per comparison (profiles/r03_tb_ab_kbench.log; all of them bit-exact on the parity suite):
//   fence in the middle 1 109-1 115 against 1 112-1 116 G; TB_BUF 1 +0.7 %; TB_BUF 1 + TB_SPLIT + fence in the middle +1.7 %
//   with 4.4 % fewer VALU instructions per launch (profiles/r03_tb_sq_counters_fewer_valu_midfence.json).  (A further form that
//   addressed everything not to be read or written OUT OF RANGE of the buffer -- loads return 0, stores are dropped, no branch
//   around the store -- measured -1...+3 % and failed the row-slab parity test: removed.)
//   None is the default: the kernel is not short of issue slots, and the buffer forms need the wave
//   tile's window to stay below 2 GiB, which the pointer form does not.
#ifndef TB_BUF
#define TB_BUF 0
#endif
#ifndef TB_SPLIT
#define TB_SPLIT 0
#endif
//   TB_FAKE 1 = TIMING ONLY, results are garbage: rows come from and go to LDS instead of HBM (one ds_read_b128 / ds_write_b128
//   per lane and row, codes made up per row), i.e. the level pipeline of a wave with no global memory at all -- what one half
//   of a temporally split pair of waves (levels 1..T/2 | T/2+1..T, rows handed over through an LDS ring) could do at best.
#ifndef TB_FAKE
#define TB_FAKE 0
#endif
//   TB_PIPE 1 = in the steady-state groups the LINK lookups of level t + 1 are issued in the middle of level t, right behind
//   its four sigma stages, into the registers of level t's links (dead by then); c0 / b of level t are issued at its head and
//   needed at its tail.  The wave then waits for a level's first coefficients once per step instead of once per level.
//   Same registers (168 VGPRs at T = 8), bit-exact, and within the run-to-run noise of the default: 1 148-1 191 against
//   1 145-1 157 G at T = 8, 1 130-1 141 against 1 118-1 157 at T = 6, 1 014-1 063 against 1 006-1 052 at T = 4.
#ifndef TB_PIPE
#define TB_PIPE 0
#endif
#if TB_BUF
typedef unsigned int tb_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tb_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
#endif
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
#if TB_FENCE == 3
    // TB_FENCE 3 (default arithmetic only): everything above needs rows of EARLIER steps only, and so does the product of the
    // N term; the row this step's previous level has just produced (vS) enters below.  With the fence here the head of level
    // t + 1 -- lookups, lane shifts, two of its four terms and the N product -- may overlap the tail of level t.  The order of
    // the additions is the reference's (W, E, S, N): same bits.
    if constexpr (!FMA) {
        const double pn0 = k.aN[0] * vN.x, pn1 = k.aN[1] * vN.y;
        __builtin_amdgcn_sched_barrier(0);
        s0 = s0 + k.aS[0] * vS.x; s1 = s1 + k.aS[1] * vS.y;
        s0 = s0 + pn0; s1 = s1 + pn1;
    } else {
        __builtin_amdgcn_sched_barrier(0);
        s0 = mul_add<FMA>(k.aS[0], vS.x, s0); s1 = mul_add<FMA>(k.aS[1], vS.y, s1);
        s0 = mul_add<FMA>(k.aN[0], vN.x, s0); s1 = mul_add<FMA>(k.aN[1], vN.y, s1);
    }
#else
    s0 = mul_add<FMA>(k.aS[0], vS.x, s0); s1 = mul_add<FMA>(k.aS[1], vS.y, s1);
    s0 = mul_add<FMA>(k.aN[0], vN.x, s0); s1 = mul_add<FMA>(k.aN[1], vN.y, s1);
#endif
#if TB_FENCE == 1 || TB_FENCE == 2
    __builtin_amdgcn_sched_barrier(0);
#endif
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
    if constexpr (GUARD || !TB_PAIR) {
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
// (Variants measured and dropped: software-pipelining the table lookups one sweep level ahead of
// the arithmetic -- 152/172/190 VGPRs at T=4/6/8, +4 % at T=4, -8 % / -4 % at T=6 / T=8: once more
// instruction-level parallelism bought with occupancy is a wash; a 3-row prefetch ring refilled slot by slot instead of the
// group-of-three double buffer -- fewer VGPRs (94/118 at T=4/6) but 5-8 % slower, the batched
// loads matter; T = 8 squeezed to 128 VGPRs for 4 waves/SIMD -- spills, 45 % slower;
// 4 cells per lane -- 244 VGPRs, 2 waves/SIMD, 20 % slower;
// a skewed pipeline whose T updates per step are independent -- 198 VGPRs, no faster;
// one lookup per FACE instead of per cell (a symmetric dictionary has aE(i) = aW(i+1), aN(r) =
// aS(r-1): take aE from the lane's other cell / the next lane by DPP and carry aS down one step;
// 6 instead of 10 ds_read_b64 per lane and level) -- +4 % at T=4, +5 % at T=6, 0 at T=8 where the
// carried values cost a wave of occupancy: the LDS (CDNA4: 2 clocks per ds_read_b64) is ~45 % busy,
// the lookups' latency matters, not their number; the four links a level needs first (aW, aE of both
// cells) looked up one level ahead inside the occupancy budget (+8 VGPRs: 158/136/114 at T=8/6/4) --
// 0-3 %.  What bounds the kernel (tools/ubench): an FP64 instruction with VGPR-pair sources issues once
// per 8 clocks from ONE wave whatever the ILP, and reaches the 4-clock rate only with two waves of
// the SIMD ready at once -- with 3-4 resident waves that are parked on LDS part of the time, the
// SIMD sees one or two.  Only more resident waves would help, and the registers are spent;
// cooperating strips -- the 4 waves of a workgroup on 4 adjacent windows WITHOUT overlap, the edge
// columns of every level handed over through an LDS mailbox with one s_barrier per step, so that
// only the outer sides of a 512-column super-strip carry a halo (33 instead of 37 strips of work at
// 4096 columns) -- 856 against 1 138 G cells*iter/s at T=8 (168 VGPRs): the lock-step of waves that
// sit on four different SIMDs and the mailbox read at the head of every level's dependency chain
// cost far more than the 11 % of work saved.  Independence of the waves is worth its redundancy.  The
// code is deliberately written with double2 values and named slots: an array-of-scalars
// formulation of the same dataflow made hipcc hoist the lookups to 204 VGPRs.
// Round 2, all A/B in one process at 4096^2 with the two cells stage-wise (tb_pair): lookups one level ahead again, now
// that __launch_bounds__ holds 3 waves per SIMD -- T = 6 168 VGPRs + 36 B scratch 1 104 against 1 127 G, T = 8 180 B of
// scratch 700 G, T = 4 152 VGPRs (3 waves instead of 4) 999 against 1 003; b - sigma folded into the next multiply as a
// negated source where b is identically 0 (20 instead of 22 FP64 instructions; differs from the reference only in the
// sign of an exact zero when the caller's field holds -0.0) -- T = 8 +-0, T = 6 +2 %: not taken; non-temporal stores of
// the result row -- +-0; T = 6 forced to 128 VGPRs for 4 waves per SIMD -- 96 B of scratch, 1 003 against 1 160 G;
// a fence after every 2nd / 4th level instead of every level -- +-0.  Fewer FP64 instructions and earlier lookups change
// nothing: the kernel is bound by how many waves are READY, not by what they execute.
// The row prefetch without the compiler's waits (hipcc waits vmcnt(0) for the youngest prefetched row, i.e. also for the
// ACK of the store issued one step earlier; gfx950 retires vector-memory operations in issue order, so vmcnt(3) behind
// three stores is enough): (a) asm loads into registers + a hand-counted wait -- hipcc copied / reused the destination
// registers before the wait (an asm load's destination counts as written at the end of the statement): wrong values and,
// once a reused register held an address, a memory fault; AGPR destinations halve the VGPR budget; (b) LDS-DMA
// (global_load_lds into a 3 840-B staging area per wave, rows read back by ds_read after `s_waitcnt vmcnt(3)`): correct
// (every parity test bit-exact: the in-order retirement holds), 158 VGPRs, and NO faster -- 1 204 against 1 233 G at T = 8,
// 979 against 1 084 at T = 4; with the wait removed altogether (wrong results, timing only) 1 300 G: +8 % is ALL that
// waiting for global memory costs this kernel.  Not kept.)
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
#if TB_SPLIT
    unsigned cw1[T + 1];                           // ... split: cw = first cell's code, cw1 = second cell's
#endif
#pragma unroll
    for (int t = 0; t < T; ++t) { w[t][0] = zero; w[t][1] = zero; w[t][2] = zero; }
#pragma unroll
    for (int t = 0; t <= T; ++t) {
        cw[t] = 0u;
#if TB_SPLIT
        cw1[t] = 0u;
#endif
    }

    // prefetch the first group of three rows
    double2 nx_x[3];
    unsigned nx_c[3];
#if TB_FAKE
    __shared__ double2 fake_io[2][256];
    auto fetch = [&](const int rr, double2 &vx_out, unsigned &vc_out) __attribute__((always_inline)) {
        vx_out = fake_io[0][(threadIdx.x + rr) & 255];
        vc_out = (unsigned)((((lane + rr) & 31) * 8 + 8) * 0x10001);
    };
#elif TB_BUF
    // the wave tile's window of the arrays as buffers: base = first input row (clamped into the array), offsets of the rows
    // it touches stay far below 2^32 whatever the size of the context
    const int rbase = max(r_begin, 0);
    const unsigned span = (unsigned)(r_end + 3 - rbase) * (unsigned)nx;       // cells from rbase on that may be addressed
    const __amdgpu_buffer_rsrc_t bx = tb_rsrc(x + (size_t)rbase * nx, span * 8u);
    const __amdgpu_buffer_rsrc_t bc = tb_rsrc(code + (size_t)rbase * nx, span * 2u);
    const __amdgpu_buffer_rsrc_t bo = tb_rsrc(xnew + (size_t)rbase * nx, span * 8u);
    const unsigned vcol = (unsigned)(in_x ? col : 0);
    auto fetch = [&](const int rr, double2 &vx_out, unsigned &vc_out) __attribute__((always_inline)) {
        const bool rok = rr >= row_lo && rr < row_hi && rr < r_end;          // wave-uniform
        const unsigned ro = (unsigned)((rok ? rr : rbase) - rbase) * (unsigned)nx;
        const tb_u4 v = __builtin_amdgcn_raw_buffer_load_b128(bx, (int)(vcol * 8u), (int)(ro * 8u), 0);
        const unsigned vc = __builtin_amdgcn_raw_buffer_load_b32(bc, (int)(vcol * 2u), (int)(ro * 2u), 0);
        double2 vx;
        __builtin_memcpy(&vx, &v, 16);
        const bool ok = in_x && rok;
        vx_out = ok ? vx : zero;
        vc_out = ok ? vc : 0u;
    };
#else
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
#endif
#pragma unroll
    for (int k = 0; k < 3; ++k) fetch(r_begin + k, nx_x[k], nx_c[k]);

    // (Measured with the tile time stamps, tools/tb_stamps.py: waves sharing a SIMD are served
    // oldest-first, so identical tiles finish between 82 and 128 us inside one T = 8 launch.  A
    // self-balancing s_setprio -- each wave lowering its priority as it advances -- narrowed that to
    // 88..119 us but left the back-to-back launch rate unchanged (the SIMDs are throughput-bound;
    // the early finishers' share goes to the rest), so it is not kept.)
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
#if TB_SPLIT
#pragma unroll
            for (int t = T; t >= 1; --t) { cw[t] = cw[t - 1]; cw1[t] = cw1[t - 1]; }
            cw[0] = cur_c[ph] & 0xFFFFu;
            cw1[0] = cur_c[ph] >> 16;
#else
#pragma unroll
            for (int t = T; t >= 1; --t) cw[t] = cw[t - 1];
            cw[0] = cur_c[ph];
#endif
            w[0][sS] = cur_x[ph];
#if TB_PIPE && !TB_SPLIT && !TB_FAKE
            if constexpr (!TRIM && !GUARD) {
                constexpr int PS = LUT_PLANE_STRIDE * 8;
                TbCoef k;
                auto links = [&](const unsigned cwt) __attribute__((always_inline)) {
                    const char *b0 = reinterpret_cast<const char *>(lut) + (cwt & 0xFFFFu);
                    const char *b1 = reinterpret_cast<const char *>(lut) + (cwt >> 16);
                    k.aW[0] = *reinterpret_cast<const double *>(b0 + PS);     k.aW[1] = *reinterpret_cast<const double *>(b1 + PS);
                    k.aE[0] = *reinterpret_cast<const double *>(b0 + 2 * PS); k.aE[1] = *reinterpret_cast<const double *>(b1 + 2 * PS);
                    k.aS[0] = *reinterpret_cast<const double *>(b0 + 3 * PS); k.aS[1] = *reinterpret_cast<const double *>(b1 + 3 * PS);
                    k.aN[0] = *reinterpret_cast<const double *>(b0 + 4 * PS); k.aN[1] = *reinterpret_cast<const double *>(b1 + 4 * PS);
                };
                links(cw[1]);
#pragma unroll
                for (int t = 1; t <= T; ++t) {
                    const int rt = rr - t;
                    const double2 vN = w[t - 1][sN], vC = w[t - 1][sC], vS = w[t - 1][sS];
                    const double xw0 = from_lane_below(vC.y);
                    const double xe1 = from_lane_above(vC.x);
                    {
                        const char *b0 = reinterpret_cast<const char *>(lut) + (cw[t] & 0xFFFFu);
                        const char *b1 = reinterpret_cast<const char *>(lut) + (cw[t] >> 16);
                        k.c0[0] = *reinterpret_cast<const double *>(b0); k.c0[1] = *reinterpret_cast<const double *>(b1);
                        if constexpr (WALL) { k.b[0] = *reinterpret_cast<const double *>(b0 + 5 * PS); k.b[1] = *reinterpret_cast<const double *>(b1 + 5 * PS); }
                        else { k.b[0] = 0.0; k.b[1] = 0.0; }
                    }
                    double s0 = k.aW[0] * xw0, s1 = k.aW[1] * vC.x;
                    s0 = mul_add<FMA>(k.aE[0], vC.y, s0); s1 = mul_add<FMA>(k.aE[1], xe1, s1);
                    s0 = mul_add<FMA>(k.aS[0], vS.x, s0); s1 = mul_add<FMA>(k.aS[1], vS.y, s1);
                    s0 = mul_add<FMA>(k.aN[0], vN.x, s0); s1 = mul_add<FMA>(k.aN[1], vN.y, s1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t < T) links(cw[t + 1]);                               // into the registers of the links just used
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
                    if (t == T && ph == TB_TOUCH_PH)
                        asm volatile("" :: "v"(nx_x[0].x), "v"(nx_x[0].y), "v"(nx_x[1].x), "v"(nx_x[1].y), "v"(nx_x[2].x),
                                     "v"(nx_x[2].y), "v"(nx_c[0]), "v"(nx_c[1]), "v"(nx_c[2]));
                    if (t < T) {
                        w[t][sS] = o;
                    } else if (st_x && rt >= ry0 && rt < ry1) {
#if TB_BUF
                        tb_u4 ov;
                        __builtin_memcpy(&ov, &o, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(ov, bo, (int)(vcol * 8u), (int)((unsigned)(rt - rbase) * (unsigned)nx * 8u), 0);
#else
                        st2(xnew + (size_t)rt * nx + col, o);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;                                                      // next step of the group
            }
#endif
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
#if TB_SPLIT
                const unsigned o0 = cw[t], o1 = cw1[t];
#else
                const unsigned o0 = cw[t] & 0xFFFFu, o1 = cw[t] >> 16;
#endif
                const double2 o = tb_pair<GUARD, WALL, FMA>(lut, o0, o1, vC, xw0, xe1, vS, vN, omw);
                if (t == T && ph == TB_TOUCH_PH) {
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
#if TB_FAKE
                    fake_io[1][threadIdx.x] = o;
#elif TB_BUF
                    tb_u4 ov;
                    __builtin_memcpy(&ov, &o, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(ov, bo, (int)(vcol * 8u), (int)((unsigned)(rt - rbase) * (unsigned)nx * 8u), 0);
#else
                    st2(xnew + (size_t)rt * nx + col, o);
#endif
                }
                // keep the scheduler from pulling the next sweeps' table lookups up here: left
                // alone it hoists them all (180-250 VGPRs, 1-2 waves per SIMD); with the fence a
                // step keeps ~120 VGPRs and 4 waves per SIMD hide the LDS latency instead
#if TB_FENCE != 1 && TB_FENCE != 3
                __builtin_amdgcn_sched_barrier(0);
#endif
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
                                                          double omw, unsigned long long *__restrict__ stamps)
{
    static_assert(T >= 1 && T <= 8, "unsupported T");
    __shared__ double lut[LUT_DOUBLES];
#ifdef TB_EXTRA_LDS
    // occupancy experiment (tools/build_variant.sh occ3 -DTB_EXTRA_LDS=16384): LDS nobody uses, so that fewer workgroups fit a CU
    __shared__ char occupancy_pad[TB_EXTRA_LDS];
    if (nrows < 0) occupancy_pad[threadIdx.x] = 1;
#endif
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
            stamps[2 * (size_t)wt] = t_begin;
            stamps[2 * (size_t)wt + 1] = wall_clock64();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Temporal split across a PAIR of waves (round 3; tuning tb_impl = 3, T = 8 only): wave A streams the
// rows of a tile from HBM through levels 1..T/2 and hands every level-T/2 row (with the row's codes) to wave B through a
// ring in LDS; wave B runs levels T/2+1..T and stores.  Each wave holds half the windows (~120 VGPRs: four waves per SIMD),
// A's vmcnt counts loads only and B never waits for memory at all; a SIMD carries two tiles (two pairs) instead of three,
// so the chunks are taller and recompute less.  No barrier: the ring is a single-producer / single-consumer queue with two
// counters (rows produced / rows consumed), LDS operations of one wave execute in order, and both waves run the same
// number of steps over the same tiles, so every wait is for a row the partner is certain to produce / consume.
// Same arithmetic per cell as tb_strip -- the levels are just dealt to two waves -- so the same bits.
constexpr int TB2_RING = 8;                                    // level-T/2 rows in flight between the two waves of a pair
struct Tb2Shared {
    double2 x[4][TB2_RING][64];
    unsigned c[4][TB2_RING][64];
    int prod[4], cons[4];
};

// ROLE 1 = wave A (levels 1..T/2, reads HBM, writes the ring), 2 = wave B (reads the ring, levels T/2+1..T, writes HBM).
template <int T, bool GUARD, bool WALL, bool FMA, int ROLE>
__device__ __forceinline__ void tb_strip_half(const double *lut, const uint16_t *__restrict__ code,
                                              const double *__restrict__ x, double *__restrict__ xnew, int nx,
                                              int ny, int row_lo, int own_hi, int tx, int ntx, int shift, int ry0, int LY,
                                              int lane, double omw, double2 (*ring_x)[64], unsigned (*ring_c)[64],
                                              int *prod, int *cons, int &k, int &seen)
{
    static_assert(T % 2 == 0 && T >= 2, "the split needs an even T");
    constexpr int H = T / 2;
    constexpr int HW = (T + 1) & ~1;
    constexpr int WOUT = TB_COLS - 2 * HW;
    constexpr int TLO = ROLE == 1 ? 1 : H + 1, THI = ROLE == 1 ? H : T;
    const int col = tx * WOUT - shift + 2 * lane;
    const bool in_x = col >= 0 && col < nx;
    const int out_lo = (tx == 0) ? 0 : tx * WOUT - shift + HW;
    const int out_hi = (tx == ntx - 1) ? nx : tx * WOUT - shift + TB_COLS - HW;
    const int row_hi = row_lo + ny;
    const int ry1 = min(ry0 + LY, own_hi);
    const int r_begin = max(ry0 - T, row_lo), r_end = ry1 + T;
    const bool st_x = in_x && (col >= out_lo) && (col < out_hi);
    const double2 zero = make_double2(0.0, 0.0);

    double2 w[T][3];                               // w[t]: 3 newest rows of sweep t (A uses 0..H-1, B uses H..T-1)
    unsigned cw[T + 1];                            // cw[t]: the two 16-bit codes of row rr-t (A: 0..H, B: H..T)
#pragma unroll
    for (int t = 0; t < T; ++t) { w[t][0] = zero; w[t][1] = zero; w[t][2] = zero; }
#pragma unroll
    for (int t = 0; t <= T; ++t) cw[t] = 0u;

    double2 nx_x[3] = {zero, zero, zero};
    unsigned nx_c[3] = {0u, 0u, 0u};
    auto fetch = [&](const int rr, double2 &vx_out, unsigned &vc_out) __attribute__((always_inline)) {
        const bool ok = in_x && rr >= row_lo && rr < row_hi && rr < r_end;
        const size_t p = (size_t)(ok ? rr : 0) * nx + (ok ? col : 0);
        const double2 vx = ld2(x + p);                                       // unconditional, see tb_strip
        const unsigned vc = *reinterpret_cast<const uint32_t *>(code + p);
        vx_out = ok ? vx : zero;
        vc_out = ok ? vc : 0u;
    };
    if constexpr (ROLE == 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) fetch(r_begin + q, nx_x[q], nx_c[q]);
    }
    // The two waves meet once per GROUP of three steps, at its head (a wait inside the unrolled steps -- a loop in the middle of
    // the level code -- cost wave A 66 VGPRs: 196 instead of 130): A starts a group when the ring has room for its three rows,
    // B when its three rows are there; with 8 slots the two conditions cannot both fail.
    auto meet = [&]() __attribute__((always_inline)) {
        if constexpr (ROLE == 1) {
            while (k + 3 - seen > TB2_RING) {
                seen = *reinterpret_cast<volatile int *>(cons);
                if (k + 3 - seen > TB2_RING) __builtin_amdgcn_s_sleep(1);
            }
        } else {
            while (seen - (k + 3) < 0) {
                seen = *reinterpret_cast<volatile int *>(prod);
                if (seen - (k + 3) < 0) __builtin_amdgcn_s_sleep(1);
            }
            *reinterpret_cast<volatile int *>(cons) = k;           // the rows of the groups before this one are in registers
        }
    };
    // A: hand row k to B
    auto emit = [&](const double2 o, const unsigned c) __attribute__((always_inline)) {
        const int slot = k & (TB2_RING - 1);
        ring_x[slot][lane] = o;
        ring_c[slot][lane] = c;
        asm volatile("" ::: "memory");                 // the counter goes out behind the row (LDS executes a wave's operations in order)
        *reinterpret_cast<volatile int *>(prod) = ++k;
    };
    // B: take row k
    auto take = [&](double2 &o, unsigned &c) __attribute__((always_inline)) {
        const int slot = k & (TB2_RING - 1);
        o = ring_x[slot][lane];
        c = ring_c[slot][lane];
        ++k;
    };

    auto group = [&](const int r, auto trim_tag) __attribute__((always_inline)) {
        constexpr bool TRIM = decltype(trim_tag)::value;
        double2 cur_x[3];
        unsigned cur_c[3];
        meet();
        if constexpr (ROLE == 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { cur_x[q] = nx_x[q]; cur_c[q] = nx_c[q]; }
#pragma unroll
            for (int q = 0; q < 3; ++q) fetch(r + 3 + q, nx_x[q], nx_c[q]);
        } else {
            // the group's three rows are there (meet): read them all now, use them step by step
#pragma unroll
            for (int q = 0; q < 3; ++q) take(cur_x[q], cur_c[q]);
        }
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int rr = r + ph;
            const int sN = (ph + 1) % 3, sC = (ph + 2) % 3, sS = ph;
            if constexpr (ROLE == 1) {
#pragma unroll
                for (int t = H; t >= 1; --t) cw[t] = cw[t - 1];
                cw[0] = cur_c[ph];
                w[0][sS] = cur_x[ph];
            } else {
#pragma unroll
                for (int t = T; t >= H + 1; --t) cw[t] = cw[t - 1];
                cw[H] = cur_c[ph];
                w[H][sS] = cur_x[ph];
            }
#pragma unroll
            for (int t = TLO; t <= THI; ++t) {
                const int rt = rr - t;
                if constexpr (TRIM) {
                    if (rt < row_lo || rt - t < ry0 - T) {   // wave-uniform: above the mesh / above this level's halo
                        if (ROLE == 1 && t == H) emit(zero, 0u);               // B skips its levels of this step too; it still takes a row
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                }
                const double2 vN = w[t - 1][sN], vC = w[t - 1][sC], vS = w[t - 1][sS];
                const double xw0 = from_lane_below(vC.y);
                const double xe1 = from_lane_above(vC.x);
                const unsigned o0 = cw[t] & 0xFFFFu, o1 = cw[t] >> 16;
                const double2 o = tb_pair<GUARD, WALL, FMA>(lut, o0, o1, vC, xw0, xe1, vS, vN, omw);
                if (ROLE == 1 && t == H && ph == TB_TOUCH_PH) {
                    // A's vmcnt holds loads only: ask for the prefetched group once per group, late (see tb_strip)
                    asm volatile("" :: "v"(nx_x[0].x), "v"(nx_x[0].y), "v"(nx_x[1].x), "v"(nx_x[1].y), "v"(nx_x[2].x),
                                 "v"(nx_x[2].y), "v"(nx_c[0]), "v"(nx_c[1]), "v"(nx_c[2]));
                }
                if (ROLE == 1 && t == H) {
                    emit(o, cw[H]);
                } else if (t < T) {
                    w[t][sS] = o;
                } else if (st_x && rt >= ry0 && rt < ry1) {
                    st2(xnew + (size_t)rt * nx + col, o);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    constexpr int TRIMMED = ((2 * T + 2) / 3) * 3;
    int r = r_begin;
    for (; r < r_begin + TRIMMED && r < r_end; r += 3) group(r, TbTag<true>{});
    for (; r < r_end; r += 3) group(r, TbTag<false>{});
}

// grid: persistent workgroups of 8 waves = 4 pairs; wave tiles are numbered and dealt exactly as in k_sweep_matfree_tb (4 per
// workgroup), wave w and wave w + 4 share tile w (A = the lower one).
template <int T, bool FMA, bool GUARD>
__global__ __launch_bounds__(512, 4) void k_sweep_matfree_tb2(const double *__restrict__ lut_g,
                                                          const uint16_t *__restrict__ code,
                                                          const double *__restrict__ x,
                                                          double *__restrict__ xnew, int nx, int ny,
                                                          int img_stride, int dom_lo, int own_lo,
                                                          int own_h, int cpi,
                                                          const uint8_t *__restrict__ active,
                                                          int LY, int ntx, int nbt, int gy, int flip,
                                                          int xmajor, int allb, int nrows, int shift,
                                                          double omw, unsigned long long *__restrict__ stamps)
{
    __shared__ double lut[LUT_DOUBLES];
    __shared__ Tb2Shared ring;
    if (threadIdx.x < 4) { ring.prod[threadIdx.x] = 0; ring.cons[threadIdx.x] = 0; }
    load_lut<512>(lut, lut_g, nrows);                              // (ends in a barrier: the counters are zero for everybody)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pair = wave & 3, role = wave >> 2;
    const unsigned total = (unsigned)nbt;
    const unsigned wtiles = (unsigned)ntx * (unsigned)gy;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned nper = gridDim.x >> 3;
    int k = 0, seen = 0;                                           // rows handed over so far / the partner's counter as last read
    for (unsigned kk = blockIdx.x >> 3; kk < per; kk += nper) {
        const unsigned bt = xcd * per + (flip ? per - 1u - kk : kk);
        if (bt >= total) continue;
        const unsigned wt = bt * 4u + (unsigned)pair;
        if (wt >= wtiles) continue;                    // the same for both waves of the pair
        const int tx = xmajor ? (int)(wt % (unsigned)ntx) : (int)(wt / (unsigned)gy);
        const int bty = xmajor ? (int)(wt / (unsigned)ntx) : (int)(wt % (unsigned)gy);
        const int img = bty / cpi;
        if (active && !active[img]) continue;
        const int row_lo = dom_lo + img * img_stride;
        const int own0 = own_lo + img * img_stride;
        const int ry0 = own0 + (bty - img * cpi) * LY;
        const bool wall = allb || tx == 0 || tx == ntx - 1;
#define TB2_CALL(WALL_, ROLE_)                                                                                     \
    tb_strip_half<T, GUARD, WALL_, FMA, ROLE_>(lut, code, x, xnew, nx, ny, row_lo, own0 + own_h, tx, ntx, shift, ry0, LY, lane, \
                                               omw, ring.x[pair], ring.c[pair], &ring.prod[pair], &ring.cons[pair], k, seen)
        if (role == 0) { if (wall) TB2_CALL(true, 1); else TB2_CALL(false, 1); }
        else { if (wall) TB2_CALL(true, 2); else TB2_CALL(false, 2); }
#undef TB2_CALL
    }
}

}  // namespace deff
