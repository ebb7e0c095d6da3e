// kernels_wgtile.hpp -- matrix-free sweeps with temporal blocking, workgroup-tile form:
// T weighted-Jacobi sweeps per pass over HBM on a tile that stays IN REGISTERS for the whole pass
// (gfx950, wave64, FP64).
//
// Same arithmetic as every other sweep kernel (updateX_SOR, Deff2DGPU/Deff2D.cuh:69-92, through
// tb_cell()), same legality argument as kernels_tb.hpp (the reference inspects the field only every
// 10 000 sweeps, cuh:1243), bit-identical results.
//
// Why a second form.  In the streaming kernel (kernels_tb.hpp) ONE wave carries a tile through all T
// levels, a row at a time: every level of a step depends on the level before it, so a tile is a
// chain of (rows + 2T) x T dependent level-steps.  That is fine when there are far more tiles than
// wave slots (4096^2 and up), and slow when there are not: ONE 1024^2 image is ~2 300 tiles of 4 rows
// whose 8 halo rows, first-load latency and ~600-clock level-steps bound a pass at ~10 us whatever
// the chip could do (BASELINE config #2 ran at 24 % of the large-image rate).
//
// Here a WORKGROUP of 8 waves owns a tile of 8R rows x 128 columns (halo included).  Wave w holds rows
// [wR, wR + R) in registers (2 cells per lane, like the streaming kernel: W/E neighbours by DPP).  A
// sweep updates all rows in place; only the first and the last row of every wave are needed by another
// wave, and they travel through an LDS mailbox (double-buffered: ONE s_barrier per sweep, reached after the
// wave's interior rows are done, so that waiting for the slowest wave overlaps useful work).  The R
// rows of a wave are independent within a sweep, so a wave has instruction-level parallelism where the
// streaming kernel has a dependency chain, and it loads its rows in one burst and never reloads them.
// And because a row stays with its wave for the whole pass, so do its MATRIX ROWS: the coefficients of the
// wave's cells are looked up in the dictionary ONCE per pass and kept in registers (20 VGPRs per tile row),
// so a sweep is pure arithmetic -- 22 FP64 + 4 DPP per tile row, no LDS lookup (tools/ubench: a sweep level
// costs 109 clocks with the coefficients in registers against 151 with ten ds_read_b64 in front of it).
// The price is the barrier: the waves of a tile advance in lock-step once per sweep.
//
// (Tried for images with more tiles than resident workgroups: the NEXT tile's rows prefetched into LDS by LDS-DMA while
// the current tile is swept -- correct, and no faster: 1536^2 573 against 613 G, 2048^2 645 against 680 G.  Between tiles
// the direct loads are L2-warm; the 2.7 us in front of a launch's first tile is the launch ramp, not the rows.)
//
// Halo: after t sweeps the outermost t rows / columns of a tile are stale, so a tile produces
// up to 8R - 2T rows x (128 - 2*HW) columns; rows that no owned cell depends on any more are skipped sweep by
// sweep (wave-uniform tests), columns cannot be (they are lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_tb.hpp"

namespace deff {

constexpr int WGT_WAVES = 8;                                   // waves per workgroup tile (512 threads, 2 per SIMD)

// most rows a tile of 8 waves x R rows per wave can OWN after a pass of T sweeps
constexpr int wgt_rows_owned(int T, int R) { return WGT_WAVES * R - 2 * T; }

// The matrix rows of a lane's two cells (c0 = w/A0 and the four links; b separately: it is zero away from the walls).
struct WgtCoef {
    double c0[2], aW[2], aE[2], aS[2], aN[2];
};

template <bool WALL>
__device__ __forceinline__ void wgt_lookup(const double *lut, unsigned codes, WgtCoef &k, double2 &b)
{
    constexpr int PS = LUT_PLANE_STRIDE * 8;
    double bb[2] = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned off = h ? (codes >> 16) : (codes & 0xFFFFu);
        const char *base = reinterpret_cast<const char *>(lut) + off;
        k.c0[h] = *reinterpret_cast<const double *>(base);
        k.aW[h] = *reinterpret_cast<const double *>(base + PS);
        k.aE[h] = *reinterpret_cast<const double *>(base + 2 * PS);
        k.aS[h] = *reinterpret_cast<const double *>(base + 3 * PS);
        k.aN[h] = *reinterpret_cast<const double *>(base + 4 * PS);
        if constexpr (WALL) bb[h] = *reinterpret_cast<const double *>(base + 5 * PS);
    }
    b = make_double2(bb[0], bb[1]);
}

// NROWS tile rows (2 cells per lane each) updated TOGETHER, one arithmetic stage at a time over all of them: the
// 2*NROWS cells are independent, so every FP64 instruction has 2*NROWS - 1 others between itself and the one that
// consumes its result.  (Written cell by cell, hipcc emits each cell's seven-deep chain back to back -- measured 13
// clocks per FP64 instruction on a SIMD holding two such waves.)  The arithmetic per cell is tb_cell()'s
// (kernels_tb.hpp), i.e. updateX_SOR's (cuh:69-92), operation for operation.
template <int NROWS, bool GUARD, bool FMA>
__device__ __forceinline__ void wgt_rows(const WgtCoef *const (&k)[NROWS], const double2 (&b)[NROWS],
                                         const double2 (&n_)[NROWS], const double2 (&c_)[NROWS],
                                         const double2 (&s_)[NROWS], double omw, double2 (&out)[NROWS])
{
    constexpr int G = 2 * NROWS;
    double xw[G], xe[G], xc[G], xs[G], xn[G], sg[G];
#pragma unroll
    for (int q = 0; q < NROWS; ++q) {
        xw[2 * q] = from_lane_below(c_[q].y);  xe[2 * q] = c_[q].y;
        xw[2 * q + 1] = c_[q].x;               xe[2 * q + 1] = from_lane_above(c_[q].x);
        xc[2 * q] = c_[q].x; xc[2 * q + 1] = c_[q].y;
        xs[2 * q] = s_[q].x; xs[2 * q + 1] = s_[q].y;
        xn[2 * q] = n_[q].x; xn[2 * q + 1] = n_[q].y;
    }
    if constexpr (GUARD) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const WgtCoef &kk = *k[g >> 1];
            const int h = g & 1;
            sg[g] = jacobi_cell<FMA>(kk.c0[h], kk.aW[h], kk.aE[h], kk.aS[h], kk.aN[h], h ? b[g >> 1].y : b[g >> 1].x,
                                     xc[g], xw[g], xe[g], xs[g], xn[g], omw);
        }
    } else {
        double om[G];
        // the leading `0 +` of the reference's sigma is bit-neutral here, see tb_cell()
#pragma unroll
        for (int g = 0; g < G; ++g) sg[g] = k[g >> 1]->aW[g & 1] * xw[g];
#pragma unroll
        for (int g = 0; g < G; ++g) sg[g] = mul_add<FMA>(k[g >> 1]->aE[g & 1], xe[g], sg[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) sg[g] = mul_add<FMA>(k[g >> 1]->aS[g & 1], xs[g], sg[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) sg[g] = mul_add<FMA>(k[g >> 1]->aN[g & 1], xn[g], sg[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) sg[g] = ((g & 1) ? b[g >> 1].y : b[g >> 1].x) - sg[g];
        if constexpr (!FMA) {
#pragma unroll
            for (int g = 0; g < G; ++g) om[g] = omw * xc[g];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) sg[g] = k[g >> 1]->c0[g & 1] * sg[g];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if constexpr (FMA) sg[g] = __builtin_fma(omw, xc[g], sg[g]);
            else sg[g] = om[g] + sg[g];
        }
    }
#pragma unroll
    for (int q = 0; q < NROWS; ++q) out[q] = make_double2(sg[2 * q], sg[2 * q + 1]);
}

// The T sweeps of a wave's R rows, in place, from coefficients held in registers.  Per sweep: publish the first and last
// row; update the interior rows 1 .. R-2 (they need nothing from another wave), two at a time; barrier; update rows 0 and
// R-1 from the neighbours' edge rows.  A wave some of whose rows no owned cell depends on any more (the tile's outermost
// waves) takes the row-by-row path with a wave-uniform test per row.  Every wave of the workgroup must call this (one
// __syncthreads per sweep); `par` is the mailbox parity, toggled every sweep (also across tiles and passes).
template <int T, int R, bool FMA, bool GUARD, bool WALL>
__device__ __forceinline__ void wgt_sweeps(double2 (&xr)[R], const WgtCoef (&k)[R], const double2 (&bb)[WALL ? R : 1],
                                           double2 (&edge)[2][WGT_WAVES][2][64], int &par, const int wave, const int lane,
                                           const int w0, const int ry0, const int ry1, const int row_lo, const int row_hi,
                                           const double omw, unsigned long long *st)
{
    constexpr int NW = WGT_WAVES;
    const double2 zero = make_double2(0.0, 0.0);
    auto one = [&](const int r, const double2 n_, const double2 c_, const double2 s_) __attribute__((always_inline)) {
        const WgtCoef *const kp[1] = {&k[r]};
        const double2 b1[1] = {WALL ? bb[WALL ? r : 0] : zero};
        const double2 n1[1] = {n_}, c1[1] = {c_}, s1[1] = {s_};
        double2 o[1];
        wgt_rows<1, GUARD, FMA>(kp, b1, n1, c1, s1, omw, o);
        return o[0];
    };
    auto two = [&](const int ra, const int rb, const double2 na, const double2 ca, const double2 sa, const double2 nb,
                   const double2 cb, const double2 sb, double2 &oa, double2 &ob) __attribute__((always_inline)) {
        const WgtCoef *const kp[2] = {&k[ra], &k[rb]};
        const double2 b2[2] = {WALL ? bb[WALL ? ra : 0] : zero, WALL ? bb[WALL ? rb : 0] : zero};
        const double2 n2[2] = {na, nb}, c2[2] = {ca, cb}, s2[2] = {sa, sb};
        double2 o[2];
        wgt_rows<2, GUARD, FMA>(kp, b2, n2, c2, s2, omw, o);
        oa = o[0];
        ob = o[1];
    };
#pragma unroll 1
    for (int t = 1; t <= T; ++t) {
        // level t is needed on [ry0 - (T - t), ry1 + (T - t)) inside the mesh
        const int need_lo = max(ry0 - (T - t), row_lo), need_hi = min(ry1 + (T - t), row_hi);
        edge[par][wave][0][lane] = xr[0];
        edge[par][wave][1][lane] = xr[R - 1];
        const bool any = w0 + R > need_lo && w0 < need_hi;
        const bool full = w0 >= need_lo && w0 + R <= need_hi;
        if (full) {
            const double2 old1 = xr[1], oldp = xr[R - 2];
            double2 prev = xr[0];
#pragma unroll
            for (int r = 1; r + 1 <= R - 2; r += 2) {
                const double2 ca = xr[r], cb = xr[r + 1];
                two(r, r + 1, prev, ca, cb, ca, cb, xr[r + 2], xr[r], xr[r + 1]);
                prev = cb;
            }
            if constexpr ((R - 2) % 2 == 1) xr[R - 2] = one(R - 2, prev, xr[R - 2], xr[R - 1]);
            __syncthreads();
            const double2 top = (wave > 0) ? edge[par][wave - 1][1][lane] : zero;
            const double2 bot = (wave < NW - 1) ? edge[par][wave + 1][0][lane] : zero;
            two(0, R - 1, top, xr[0], old1, oldp, xr[R - 1], bot, xr[0], xr[R - 1]);
        } else if (any) {
            const double2 old1 = xr[1], oldp = xr[R - 2];
            double2 prev = xr[0];
#pragma unroll
            for (int r = 1; r <= R - 2; ++r) {
                const double2 cur = xr[r];
                const int rr = w0 + r;
                if (rr >= need_lo && rr < need_hi) xr[r] = one(r, prev, cur, xr[r + 1]);
                prev = cur;
            }
            __syncthreads();
            const double2 top = (wave > 0) ? edge[par][wave - 1][1][lane] : zero;
            const double2 bot = (wave < NW - 1) ? edge[par][wave + 1][0][lane] : zero;
            if (w0 >= need_lo && w0 < need_hi) xr[0] = one(0, top, xr[0], old1);
            if (w0 + R - 1 >= need_lo && w0 + R - 1 < need_hi) xr[R - 1] = one(R - 1, oldp, xr[R - 1], bot);
        } else {
            __syncthreads();
        }
        par ^= 1;
        if (st) st[2 + t] = wall_clock64();
    }
}

// grid: persistent workgroups of 8 waves; tiles (strip tx, row tile ty) numbered like the streaming
// kernel's wave tiles (x-major: neighbouring strips adjacent; per XCD a contiguous run).
// Geometry parameters as k_sweep_matfree_tb: image k of a stack has its mesh rows at
// [dom_lo + k*img_stride, ... + ny) and its owned rows at [own_lo + k*img_stride, ... + own_h).
// ly = rows a tile owns (<= 8R - 2T; the planner spreads an image's rows evenly over its cpi row tiles).
template <int T, int R, bool FMA, bool GUARD>
__global__ __launch_bounds__(WGT_WAVES * 64, 2) void k_sweep_wgtile(const double *__restrict__ lut_g,
                                                                    const uint16_t *__restrict__ code,
                                                                    const double *__restrict__ x,
                                                                    double *__restrict__ xnew, int nx, int ny,
                                                                    int img_stride, int dom_lo, int own_lo,
                                                                    int own_h, int cpi, int ly,
                                                                    const uint8_t *__restrict__ active,
                                                                    int ntx, int gy, int xmajor, int allb,
                                                                    int nrows, int shift, double omw,
                                                                    unsigned long long *__restrict__ stamps)
{
    constexpr int NW = WGT_WAVES;
    static_assert(T >= 1 && T <= 8 && R >= 4, "unsupported tile");
    static_assert(LUT_PLANES * LUT_MAX_ROWS <= 6 * WGT_WAVES * 64, "dictionary fetch assumes <= 6 doubles per thread");
    static_assert(wgt_rows_owned(T, R) >= 1, "tile owns no row");
    constexpr int HW = (T + 1) & ~1;               // column halo, even (16-B lane pairs), as tb_strip
    constexpr int WOUT = TB_COLS - 2 * HW;

    __shared__ double lut[LUT_DOUBLES];
    __shared__ double2 edge[2][NW][2][64];         // [parity][wave][first / last row][lane]: 32 KiB
    // diagnostics only (tools/wgt_stamps.py): 100 MHz wall clock at kernel entry, dictionary loaded, rows loaded, after every
    // sweep, after the stores -- T + 4 values per tile, written by one lane, read by nobody on the device
    const unsigned long long t_entry = stamps ? wall_clock64() : 0ull;
    unsigned long long t_lut = 0ull;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned total = (unsigned)ntx * (unsigned)gy;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned nper = gridDim.x >> 3;
    const double2 zero = make_double2(0.0, 0.0);
    int par = 0;                                   // mailbox parity, toggled every sweep (also across tiles)

    // Per-tile state.  The loop below is rotated: the rows of tile i+1 are requested right after the stores of tile i, and
    // the dictionary is loaded BETWEEN the first tile's row requests and its first use, so that the two global round trips
    // overlap (one 1024^2 image is a single tile per workgroup: ~1 us of ~12).
    unsigned bt = 0;
    int row_lo = 0, row_hi = 0, ry0 = 0, ry1 = 0, w0 = 0, col = 0;
    bool st_x = false, wall = false;
    double2 xr[R];
    unsigned cc[R];
    // geometry of tile kk and the burst of loads for its rows: R rows of x (16 B per lane) and their codes (4 B per lane),
    // issued unconditionally at a clamped address and selected afterwards (see tb_strip).  false: no such tile / frozen image.
    auto open_tile = [&](const unsigned kk) __attribute__((always_inline)) -> bool {
        bt = xcd * per + kk;
        if (bt >= total) return false;             // workgroup-uniform, like every test up to the sweeps
        const int tx = xmajor ? (int)(bt % (unsigned)ntx) : (int)(bt / (unsigned)gy);
        const int bty = xmajor ? (int)(bt / (unsigned)ntx) : (int)(bt % (unsigned)gy);
        const int img = bty / cpi;
        if (active && !active[img]) return false;  // frozen image of a batch
        row_lo = dom_lo + img * img_stride;
        row_hi = row_lo + ny;
        const int own0 = own_lo + img * img_stride;
        ry0 = own0 + (bty - img * cpi) * ly;
        ry1 = min(ry0 + ly, own0 + own_h);
        w0 = ry0 - T + wave * R;                   // this wave's first array row
        // rows worth loading: the tile's own rows + T above and below, inside the mesh
        const int ld_lo = max(ry0 - T, row_lo), ld_hi = min(ry1 + T, row_hi);
        // columns: exactly tb_strip's placement
        col = tx * WOUT - shift + 2 * lane;
        const bool in_x = col >= 0 && col < nx;
        const int out_lo = (tx == 0) ? 0 : tx * WOUT - shift + HW;
        const int out_hi = (tx == ntx - 1) ? nx : tx * WOUT - shift + TB_COLS - HW;
        st_x = in_x && (col >= out_lo) && (col < out_hi);
        wall = allb || tx == 0 || tx == ntx - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int rr = w0 + r;
            const bool ok = in_x && rr >= ld_lo && rr < ld_hi;
            const size_t p = (size_t)(ok ? rr : 0) * nx + (ok ? col : 0);
            const double2 vx = ld2(x + p);
            const unsigned vc = *reinterpret_cast<const uint32_t *>(code + p);
            xr[r] = ok ? vx : zero;
            cc[r] = ok ? vc : 0u;                  // outside the mesh: the zero row
        }
        return true;
    };

    unsigned kk = blockIdx.x >> 3;
    bool valid = kk < per && open_tile(kk);
    load_lut<NW * 64>(lut, lut_g, nrows);
    t_lut = stamps ? wall_clock64() : 0ull;

    while (kk < per) {
        if (valid) {
        unsigned long long *st = (stamps && threadIdx.x == 0) ? stamps + (size_t)bt * (T + 4) : nullptr;
        if (st) {
            st[0] = t_entry;
            st[1] = t_lut;
            asm volatile("" :: "v"(xr[0].x), "v"(xr[R - 1].y), "v"(cc[R - 1]));    // wait for the first wave's rows
            st[2] = wall_clock64();
        }

        // the T sweeps of this wave's R rows, from coefficients looked up once (wgt_sweeps)
        auto tile = [&](auto wall_tag) __attribute__((always_inline)) {
            constexpr bool WALL = decltype(wall_tag)::value;
            WgtCoef k[R];
            double2 bb[WALL ? R : 1];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double2 b_;
                wgt_lookup<WALL>(lut, cc[r], k[r], b_);
                if constexpr (WALL) bb[r] = b_;
            }
            wgt_sweeps<T, R, FMA, GUARD, WALL>(xr, k, bb, edge, par, wave, lane, w0, ry0, ry1, row_lo, row_hi, omw, st);
        };
        if (wall) tile(TbTag<true>{});
        else tile(TbTag<false>{});

#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int rr = w0 + r;
            if (st_x && rr >= ry0 && rr < ry1) st2(xnew + (size_t)rr * nx + col, xr[r]);
        }
        if (st) st[T + 3] = wall_clock64();
        }
        kk += nper;
        valid = kk < per && open_tile(kk);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tall tiles (resident form only): 16 waves per workgroup (4 per SIMD, 128 VGPRs), R rows per wave of which only the
// FIELD lives in registers (4 VGPRs per tile row instead of 24); the rows' 16-bit codes sit in LDS (256 B per tile row)
// and the matrix rows are looked up in the dictionary in every sweep, like the streaming kernel does (tb_pair) -- four
// waves per SIMD hide those lookups.  A tile is 16R rows x 128
// columns (R = 12: 192 rows, 176 x 112 owned = 80 % of the cells it sweeps; the 8-wave tiles own 62 %), which is what puts
// images between 1024^2 and ~2300^2 (R = 4 ... 14) -- too many 56-row tiles for residency, too few cells for tall streaming chunks --
// on the chip all at once.  Same mailbox, one barrier per sweep, rows updated in place top to bottom, interior rows before
// the barrier as in the 8-wave form (barrier first and no saved copies of rows 1 and R-2: 4-8 VGPRs less, measured 8 %
// SLOWER at R = 6...12 -- the interior rows do cover the wait; the lookups of row r + 1 issued before the arithmetic of
// row r where R <= 8 leaves 24 VGPRs for them: +-0, as in the streaming kernel).
constexpr int WGL_WAVES = 16;
constexpr int wgl_rows_owned(int T, int R) { return WGL_WAVES * R - 2 * T; }

// Is the system behind (dictionary, codes) link-symmetric the way wgl_sweeps' short-cut needs it?  For every cell: the E link
// of an even column equals, bit for bit, the W link of the odd column next to it (a lane's two cells), and the N link of a
// row equals the S link of the row above it in the same image.  The native assemblies are (fvm_row: a face has one
// harmonic mean); a dictionary harvested from somebody's matrix need not be.  Raises *flag on the first mismatch.
__global__ __launch_bounds__(256) void k_links_symmetric(const double *__restrict__ lut_g, const uint16_t *__restrict__ code,
                                                         int nx, int rows, int ny, int nrows, unsigned *flag)
{
    __shared__ double lut[LUT_DOUBLES];
    load_lut(lut, lut_g, nrows);
    constexpr int PS = LUT_PLANE_STRIDE * 8;
    const size_t n = (size_t)nx * rows;
    bool bad = false;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (size_t)gridDim.x * 256) {
        const int r = (int)(p / nx), c = (int)(p - (size_t)r * nx);
        const char *me = reinterpret_cast<const char *>(lut) + code[p];
        if (!(c & 1) && c + 1 < nx) {
            const char *east = reinterpret_cast<const char *>(lut) + code[p + 1];
            bad |= __double_as_longlong(*reinterpret_cast<const double *>(me + 2 * PS)) !=
                   __double_as_longlong(*reinterpret_cast<const double *>(east + PS));
        }
        if (r % ny != 0) {
            const char *north = reinterpret_cast<const char *>(lut) + code[p - nx];
            bad |= __double_as_longlong(*reinterpret_cast<const double *>(me + 4 * PS)) !=
                   __double_as_longlong(*reinterpret_cast<const double *>(north + 3 * PS));
        }
    }
    if (bad) atomicOr(flag, 1u);
}

// SYM = the system is link-symmetric (compile-time: both row loops in one kernel cost the tall tiles their registers --
// 2048^2 fell from 950 to 470 G with a run-time flag).
template <int T, int R, bool FMA, bool GUARD, bool WALL, bool SYM>
__device__ __forceinline__ void wgl_sweeps(double2 (&xr)[R], const unsigned *codes, const double *lut,
                                           double2 (&edge)[2][WGL_WAVES][2][64], int &par, const int wave, const int lane,
                                           const int w0, const int ry0, const int ry1, const int row_lo, const int row_hi,
                                           const double omw)
{
    constexpr int NW = WGL_WAVES;
    const double2 zero = make_double2(0.0, 0.0);
    const double *lut_t = lut;
    // One row: lookups + arithmetic, FINISHED before the next row starts.  The empty asm pins the result where it is
    // computed: without it hipcc sinks the arithmetic of every row below the sweep's barrier (nothing above needs the
    // results) while the lookups stay above it -- all 10R coefficients live at once, 1 KiB of scratch per lane at R = 12
    // and 50 us per sweep.
    auto row = [&](const int r, const double2 n_, const double2 c_, const double2 s_) __attribute__((always_inline)) {
        const double xw0 = from_lane_below(c_.y), xe1 = from_lane_above(c_.x);
        const unsigned cw = codes[r * 64];                     // this lane's two 16-bit codes of tile row r (LDS)
        double2 o;
        if constexpr (R >= 13) {
            // the two cells one after the other: half the coefficients live at a time -- what lets R = 14 stay out of
            // scratch in this loop (2304^2: 966 G against 563 with the two cells stage-wise; at R = 8 stage-wise is 8 % faster)
            o.x = tb_cell<GUARD, WALL, FMA>(lut_t, cw & 0xFFFFu, c_.x, xw0, c_.y, s_.x, n_.x, omw);
            asm volatile("" : "+v"(o.x));
            o.y = tb_cell<GUARD, WALL, FMA>(lut_t, cw >> 16, c_.y, c_.x, xe1, s_.y, n_.y, omw);
            asm volatile("" : "+v"(o.y));
        } else {
            o = tb_pair<GUARD, WALL, FMA>(lut_t, cw & 0xFFFFu, cw >> 16, c_, xw0, xe1, s_, n_, omw);
            asm volatile("" : "+v"(o.x), "+v"(o.y));
        }
        return o;
    };
#pragma unroll 1
    for (int t = 1; t <= T; ++t) {
        // The lookups of a row do not change from sweep to sweep, and hipcc knows: left alone it hoists all 10R of them out
        // of this loop, i.e. rebuilds the 8-wave form's registers (20 per tile row) inside a 128-VGPR budget -- 1 KiB of
        // scratch per lane at R = 12.  An offset the compiler cannot see through (always 0) ties them to their sweep.
        unsigned salt = 0;
        asm volatile("" : "+s"(salt));
        lut_t = reinterpret_cast<const double *>(reinterpret_cast<const char *>(lut) + salt);
        const int need_lo = max(ry0 - (T - t), row_lo), need_hi = min(ry1 + (T - t), row_hi);
        edge[par][wave][0][lane] = xr[0];
        edge[par][wave][1][lane] = xr[R - 1];
        const bool any = w0 + R > need_lo && w0 < need_hi;
        const bool full = w0 >= need_lo && w0 + R <= need_hi;
        if (full) {
            const double2 old1 = xr[1], oldp = xr[R - 2];
            double2 prev = xr[0];
            constexpr bool SHORT = SYM && !GUARD;
            if constexpr (SHORT) {
                {
                    // A link-symmetric system (k_links_symmetric) needs 7 lookups per row instead of 10: the W link of a
                    // lane's second cell IS the E link of its first, and the N links of a row ARE the S links of the row the
                    // wave has just finished (4 VGPRs carried).  Same values bit for bit, 30 % less LDS traffic:
                    // 1280^2 +4 %, 1792^2 +5 %, 2048^2 +12 %, 256 x 128^2 +7 %.
                    constexpr int PS = LUT_PLANE_STRIDE * 8;
                    double aSp0 = 0.0, aSp1 = 0.0;
                    // the row's own coefficients (7 lookups; aW[1] and aN come from the symmetry)
                    auto lookup7 = [&](const unsigned cw, TbCoef &k) __attribute__((always_inline)) {
                        const char *b0 = reinterpret_cast<const char *>(lut_t) + (cw & 0xFFFFu);
                        const char *b1 = reinterpret_cast<const char *>(lut_t) + (cw >> 16);
                        k.c0[0] = *reinterpret_cast<const double *>(b0);
                        k.c0[1] = *reinterpret_cast<const double *>(b1);
                        k.aW[0] = *reinterpret_cast<const double *>(b0 + PS);
                        k.aE[0] = *reinterpret_cast<const double *>(b0 + 2 * PS);
                        k.aE[1] = *reinterpret_cast<const double *>(b1 + 2 * PS);
                        k.aW[1] = k.aE[0];
                        k.aS[0] = *reinterpret_cast<const double *>(b0 + 3 * PS);
                        k.aS[1] = *reinterpret_cast<const double *>(b1 + 3 * PS);
                        if constexpr (WALL) {
                            k.b[0] = *reinterpret_cast<const double *>(b0 + 5 * PS);
                            k.b[1] = *reinterpret_cast<const double *>(b1 + 5 * PS);
                        } else {
                            k.b[0] = 0.0;
                            k.b[1] = 0.0;
                        }
                    };
                    // The codes of row r + 1 are fetched from LDS while row r is computed (one VGPR): without it every row starts with
                    // two dependent LDS round trips, code -> coefficients.  +2 % at 1152^2, +0.5...2 % at 1536^2...2048^2, one process
                    // per build.  (Fetching the next row's COEFFICIENTS ahead as well -- 14 VGPRs, affordable up to R = 5 -- measured
                    // -1 %: at four waves per SIMD the lookups' latency is already covered.)
                    unsigned cw_next = codes[64];
#pragma unroll
                    for (int r = 1; r <= R - 2; ++r) {
                        const double2 cur = xr[r];
                        const unsigned cw = cw_next;
                        if (r < R - 2) cw_next = codes[(r + 1) * 64];
                        TbCoef k;
                        lookup7(cw, k);
                        if (r == 1) {
                            k.aN[0] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lut_t) + (cw & 0xFFFFu) + 4 * PS);
                            k.aN[1] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lut_t) + (cw >> 16) + 4 * PS);
                        } else {
                            k.aN[0] = aSp0;
                            k.aN[1] = aSp1;
                        }
                        aSp0 = k.aS[0];
                        aSp1 = k.aS[1];
                        const double xw0 = from_lane_below(cur.y), xe1 = from_lane_above(cur.x);
                        double2 o = tb_apply<FMA>(k, cur, xw0, xe1, xr[r + 1], prev, omw);
                        asm volatile("" : "+v"(o.x), "+v"(o.y));
                        xr[r] = o;
                        prev = cur;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (!SHORT) {
#pragma unroll
                for (int r = 1; r <= R - 2; ++r) {
                    const double2 cur = xr[r];
                    xr[r] = row(r, prev, cur, xr[r + 1]);
                    prev = cur;
                    __builtin_amdgcn_sched_barrier(0);         // keep the next rows' lookups where they are (see tb_strip)
                }
            }
            __syncthreads();
            const double2 top = (wave > 0) ? edge[par][wave - 1][1][lane] : zero;
            const double2 bot = (wave < NW - 1) ? edge[par][wave + 1][0][lane] : zero;
            xr[0] = row(0, top, xr[0], old1);
            __builtin_amdgcn_sched_barrier(0);
            xr[R - 1] = row(R - 1, oldp, xr[R - 1], bot);
        } else if (any) {
            const double2 old1 = xr[1], oldp = xr[R - 2];
            double2 prev = xr[0];
#pragma unroll
            for (int r = 1; r <= R - 2; ++r) {
                const double2 cur = xr[r];
                const int rr = w0 + r;
                if (rr >= need_lo && rr < need_hi) xr[r] = row(r, prev, cur, xr[r + 1]);
                prev = cur;
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            const double2 top = (wave > 0) ? edge[par][wave - 1][1][lane] : zero;
            const double2 bot = (wave < NW - 1) ? edge[par][wave + 1][0][lane] : zero;
            if (w0 >= need_lo && w0 < need_hi) xr[0] = row(0, top, xr[0], old1);
            __builtin_amdgcn_sched_barrier(0);
            if (w0 + R - 1 >= need_lo && w0 + R - 1 < need_hi) xr[R - 1] = row(R - 1, oldp, xr[R - 1], bot);
        } else {
            __syncthreads();
        }
        par ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Link-symmetric tiles (resident form only): the 8-wave form's idea -- the matrix rows of a wave's cells stay in registers,
// a sweep is pure arithmetic -- at 14 instead of 20 VGPRs per tile row, which is what lets THREE waves per SIMD (12 per
// workgroup, 168 VGPRs) hold a tile.  A link belongs to a face: the W link of a lane's second cell IS the E link of its
// first, and the N links of a row ARE the S links of the row above it, which the same wave holds (row 0 of a wave keeps
// its own N links: 4 VGPRs per wave).  Same values, so the same bits -- provided the system at hand really is
// link-symmetric, which k_links_symmetric verifies on the device per assembly (api_solve.hip, plan key tb_sym), exactly as
// for the tall tiles' 7-lookup rows.  What it buys: an FP64 instruction issues every ~5.0 clocks from three waves of a SIMD
// against ~6.1 from two (tools/ubench), and a tile of 12 x R rows has the shape of an 8-wave tile of 1.5 R rows:
// one 1024^2 image is 9 x 24 tiles of 12 x 5 rows (44 owned) instead of 9 x 26 of 8 x 7 (40 owned).
constexpr int WGS_WAVES = 12;
constexpr int wgs_rows_owned(int T, int R) { return WGS_WAVES * R - 2 * T; }

template <int R>
struct WgsCoef {
    double2 c0[R];      // w/A0 of the lane's two cells
    double aW0[R];      // W link of the first cell (the second cell's W link is aE[].x)
    double2 aE[R];      // E links
    double2 aS[R];      // S links (row r + 1 takes them as its N links)
    double2 aN0;        // N links of the wave's first row
};

// A row's N links are taken from the S links of the row above it in the wave (k.aS[r - 1]; the wave's first row keeps its own:
// k.aN0).  When row r - 1 lies outside the image its code is 0, i.e. the dictionary's row 0, and that row is all zeros on the
// device BY CONSTRUCTION, whatever was harvested: upload_lut() (api_core.hip) fills the table from k = 1 on over a zeroed
// buffer.  The row's own N link is 0 there too (top row of an image: no N face, fvm_row.hpp), and the value it multiplies is
// the 0 that rows outside the image hold, so the product is the same +0.
template <int R, bool WALL>
__device__ __forceinline__ void wgs_lookup(const double *lut, const unsigned (&cc)[R], WgsCoef<R> &k, double2 (&bb)[WALL ? R : 1])
{
    constexpr int PS = LUT_PLANE_STRIDE * 8;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const char *b0 = reinterpret_cast<const char *>(lut) + (cc[r] & 0xFFFFu);
        const char *b1 = reinterpret_cast<const char *>(lut) + (cc[r] >> 16);
        k.c0[r] = make_double2(*reinterpret_cast<const double *>(b0), *reinterpret_cast<const double *>(b1));
        k.aW0[r] = *reinterpret_cast<const double *>(b0 + PS);
        k.aE[r] = make_double2(*reinterpret_cast<const double *>(b0 + 2 * PS), *reinterpret_cast<const double *>(b1 + 2 * PS));
        k.aS[r] = make_double2(*reinterpret_cast<const double *>(b0 + 3 * PS), *reinterpret_cast<const double *>(b1 + 3 * PS));
        if (r == 0) k.aN0 = make_double2(*reinterpret_cast<const double *>(b0 + 4 * PS), *reinterpret_cast<const double *>(b1 + 4 * PS));
        if constexpr (WALL) bb[r] = make_double2(*reinterpret_cast<const double *>(b0 + 5 * PS), *reinterpret_cast<const double *>(b1 + 5 * PS));
    }
}

// NROWS tile rows together, stage-wise (see wgt_rows); per cell tb_apply()'s operations in tb_apply()'s order.
template <int NROWS, bool FMA>
__device__ __forceinline__ void wgs_rows(const double2 (&c0)[NROWS], const double (&aW0)[NROWS], const double2 (&aE)[NROWS],
                                         const double2 (&aS)[NROWS], const double2 (&aN)[NROWS], const double2 (&b)[NROWS],
                                         const double2 (&n_)[NROWS], const double2 (&c_)[NROWS], const double2 (&s_)[NROWS],
                                         double omw, double2 (&out)[NROWS])
{
    double s0[NROWS], s1[NROWS], m0[NROWS], m1[NROWS];
#pragma unroll
    for (int q = 0; q < NROWS; ++q) { s0[q] = aW0[q] * from_lane_below(c_[q].y); s1[q] = aE[q].x * c_[q].x; }
#pragma unroll
    for (int q = 0; q < NROWS; ++q) { s0[q] = mul_add<FMA>(aE[q].x, c_[q].y, s0[q]); s1[q] = mul_add<FMA>(aE[q].y, from_lane_above(c_[q].x), s1[q]); }
#pragma unroll
    for (int q = 0; q < NROWS; ++q) { s0[q] = mul_add<FMA>(aS[q].x, s_[q].x, s0[q]); s1[q] = mul_add<FMA>(aS[q].y, s_[q].y, s1[q]); }
#pragma unroll
    for (int q = 0; q < NROWS; ++q) { s0[q] = mul_add<FMA>(aN[q].x, n_[q].x, s0[q]); s1[q] = mul_add<FMA>(aN[q].y, n_[q].y, s1[q]); }
#pragma unroll
    for (int q = 0; q < NROWS; ++q) { s0[q] = b[q].x - s0[q]; s1[q] = b[q].y - s1[q]; }
    if constexpr (!FMA) {
#pragma unroll
        for (int q = 0; q < NROWS; ++q) { m0[q] = omw * c_[q].x; m1[q] = omw * c_[q].y; }
    }
#pragma unroll
    for (int q = 0; q < NROWS; ++q) { s0[q] = c0[q].x * s0[q]; s1[q] = c0[q].y * s1[q]; }
#pragma unroll
    for (int q = 0; q < NROWS; ++q) {
        if constexpr (FMA) out[q] = make_double2(__builtin_fma(omw, c_[q].x, s0[q]), __builtin_fma(omw, c_[q].y, s1[q]));
        else out[q] = make_double2(m0[q] + s0[q], m1[q] + s1[q]);
    }
}

// The T sweeps of a wave's R rows, in place: wgt_sweeps' schedule (publish the edge rows, interior rows, barrier, edge rows)
// on the symmetric coefficient set; every wave of the workgroup must call this.
// (Measured and dropped: handing every row to the rim store as soon as its value of the pass's LAST sweep exists, so that the
// stores travel while the rest of that sweep is computed.  The flag of a pass waits for the acknowledgement of the LAST store,
// and the rows finished last -- each wave's two edge rows, behind the sweep's barrier -- all hold rim cells: "rim stored +
// flag raised" fell from 1.9 to 1.0 us per pass at 1024^2, the sweeps grew from 7.0 to 7.7 us, the pass stayed at 9.9-10.1 us.)
template <int T, int R, int NW, bool FMA, bool WALL>
__device__ __forceinline__ void wgs_sweeps(double2 (&xr)[R], const WgsCoef<R> &k, const double2 (&bb)[WALL ? R : 1],
                                           double2 (&edge)[2][NW][2][64], int &par, const int wave, const int lane,
                                           const int w0, const int ry0, const int ry1, const int row_lo, const int row_hi,
                                           const double omw)
{
    const double2 zero = make_double2(0.0, 0.0);
    auto b_of = [&](const int r) __attribute__((always_inline)) { return WALL ? bb[WALL ? r : 0] : zero; };
    auto one = [&](const int r, const double2 n_, const double2 c_, const double2 s_) __attribute__((always_inline)) {
        const double2 c01[1] = {k.c0[r]}, aE1[1] = {k.aE[r]}, aS1[1] = {k.aS[r]}, aN1[1] = {r == 0 ? k.aN0 : k.aS[r == 0 ? 0 : r - 1]};
        const double aW1[1] = {k.aW0[r]};
        const double2 b1[1] = {b_of(r)};
        const double2 n1[1] = {n_}, c1[1] = {c_}, s1[1] = {s_};
        double2 o[1];
        wgs_rows<1, FMA>(c01, aW1, aE1, aS1, aN1, b1, n1, c1, s1, omw, o);
        return o[0];
    };
    auto two = [&](const int ra, const int rb, const double2 na, const double2 ca, const double2 sa, const double2 nb,
                   const double2 cb, const double2 sb, double2 &oa, double2 &ob) __attribute__((always_inline)) {
        const double2 c02[2] = {k.c0[ra], k.c0[rb]}, aE2[2] = {k.aE[ra], k.aE[rb]}, aS2[2] = {k.aS[ra], k.aS[rb]};
        const double2 aN2[2] = {ra == 0 ? k.aN0 : k.aS[ra == 0 ? 0 : ra - 1], k.aS[rb - 1]};
        const double aW2[2] = {k.aW0[ra], k.aW0[rb]};
        const double2 b2[2] = {b_of(ra), b_of(rb)};
        const double2 n2[2] = {na, nb}, c2[2] = {ca, cb}, s2[2] = {sa, sb};
        double2 o[2];
        wgs_rows<2, FMA>(c02, aW2, aE2, aS2, aN2, b2, n2, c2, s2, omw, o);
        oa = o[0];
        ob = o[1];
    };
#pragma unroll 1
    for (int t = 1; t <= T; ++t) {
        // level t is needed on [ry0 - (T - t), ry1 + (T - t)) inside the mesh
        const int need_lo = max(ry0 - (T - t), row_lo), need_hi = min(ry1 + (T - t), row_hi);
        edge[par][wave][0][lane] = xr[0];
        edge[par][wave][1][lane] = xr[R - 1];
        const bool any = w0 + R > need_lo && w0 < need_hi;
        const bool full = w0 >= need_lo && w0 + R <= need_hi;
        if (full) {
            const double2 old1 = xr[1], oldp = xr[R - 2];
            double2 prev = xr[0];
#pragma unroll
            for (int r = 1; r + 1 <= R - 2; r += 2) {
                const double2 ca = xr[r], cb = xr[r + 1];
                two(r, r + 1, prev, ca, cb, ca, cb, xr[r + 2], xr[r], xr[r + 1]);
                prev = cb;
            }
            if constexpr ((R - 2) % 2 == 1) xr[R - 2] = one(R - 2, prev, xr[R - 2], xr[R - 1]);
            __syncthreads();
            const double2 top = (wave > 0) ? edge[par][wave - 1][1][lane] : zero;
            const double2 bot = (wave < NW - 1) ? edge[par][wave + 1][0][lane] : zero;
            two(0, R - 1, top, xr[0], old1, oldp, xr[R - 1], bot, xr[0], xr[R - 1]);
        } else if (any) {
            const double2 old1 = xr[1], oldp = xr[R - 2];
            double2 prev = xr[0];
#pragma unroll
            for (int r = 1; r <= R - 2; ++r) {
                const double2 cur = xr[r];
                const int rr = w0 + r;
                if (rr >= need_lo && rr < need_hi) xr[r] = one(r, prev, cur, xr[r + 1]);
                prev = cur;
            }
            __syncthreads();
            const double2 top = (wave > 0) ? edge[par][wave - 1][1][lane] : zero;
            const double2 bot = (wave < NW - 1) ? edge[par][wave + 1][0][lane] : zero;
            if (w0 >= need_lo && w0 < need_hi) xr[0] = one(0, top, xr[0], old1);
            if (w0 + R - 1 >= need_lo && w0 + R - 1 < need_hi) xr[R - 1] = one(R - 1, oldp, xr[R - 1], bot);
        } else {
            __syncthreads();
        }
        par ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Resident form: when ALL tiles of the context are on the chip at once (one tile per workgroup, every workgroup
// resident: one 1024^2 image is 234 tiles on 256 CUs), the launch need not end after T sweeps.  A workgroup keeps its
// tile's MATRIX ROWS and its OWNED CELLS in registers over `npass` passes and exchanges only the halo with its (up to 8)
// neighbours, so a pass costs neither a launch gap (~1.7 us), nor the dictionary fetch, nor the 70 lookups per lane of
// its first sweep, nor a re-read of the tile.  No grid-wide barrier; per pass p (reads buffer p&1, writes the other one):
//   - after its T sweeps a tile stores the RIM of its owned cells (the T outermost rows / HW outermost columns: all a
//     neighbour ever reads; the last pass of a launch stores every owned cell), waits until those stores are
//     acknowledged, and publishes flags[tile] = base + p + 1;
//   - before pass p >= 1 it polls until each neighbour's flag has reached base + p and then reads its halo cells.
//   Write-after-read is covered by the same flags: a tile overwrites buffer p&1 at the end of pass p + 1, which it only
//   starts once its neighbours have finished pass p, i.e. are done reading that buffer.  Pass 0 needs no wait: the launch
//   boundary orders it after everything before.  `base` grows by npass from launch to launch, so stale flags are always
//   smaller than anything waited for.
// Forward progress: the host launches at most as many workgroups as the occupancy query says are co-resident and never
// two resident kernels of one process on one device at a time (api_solve.hip), and every wait is bounded -- a lane that
// has polled for WGR_TIMEOUT of the 100 MHz wall clock, or that sees *abort_flag set, raises *abort_flag and its workgroup
// returns; so does, within one poll, every workgroup waiting anywhere.  The host checks the flag at its next
// synchronisation, restores the field the interval started from and redoes it with one launch per pass (resident_check).
// (stall_tile >= 0 makes that tile leave without publishing -- the tests' way to exercise this path.)
// (hipLaunchCooperativeKernel, which would make the runtime vouch for co-residency, was offered as tb_launch = 2 in round 2
// and is gone: with several host threads launching cooperatively the process died in libhsa-runtime64 under the HIP
// runtime's own exit handler, deff2d --devices 0,0,0, while the plain launch that this library bounds itself exited
// cleanly -- the library's own objects were not involved, the cooperative path bypasses the event chain -- and with the
// fallback above the plain launch needs no such guarantee.)
// Same arithmetic, same tiles, same results bit for bit as npass launches of k_sweep_wgtile.
// (Round 3, measured and dropped -- profiles/r03_resident_tile_stamps.log: the exchange costs ~2.9 of a 1024^2 pass's 9.9 us
// as a chain of three device-coherent round trips -- rim stores acknowledged 1.1-1.9, flag seen 1.0, halo read 0.6.  A tagged
// mailbox -- every cell as one 16-byte element {value, pass tag}, no flag, no acknowledgement, readers fetch until all tags
// are fresh -- polls with every wave instead of eight lanes and was SLOWER (wait 4.4-6.8 us, growing with the rows per wave);
// with all loads of a poll in flight together one 150 001-sweep solve came out WRONG on a quiet GPU while the short parity
// tests passed: whether a 16-byte sc1 access is single-copy atomic against another XCD is nothing this library can
// establish, so nothing here rests on it.  Flags after acknowledged stores stay.)
constexpr unsigned long long WGR_TIMEOUT = 200000000ull;       // 2 s
constexpr int WGR_FLAG_STRIDE = 64;                            // unsigneds between two tiles' flags: one 256-byte block each, so that
                                                               // ~2 000 polling lanes do not queue on a handful of cache lines

// The field between passes of one launch travels through device-coherent accesses: 16-byte buffer loads / stores with
// sc1 set (what an agent-scope relaxed atomic compiles to on gfx942/gfx950: the store is written through, the load is
// served from the coherence point, neither needs a cache-wide write-back or invalidate).  Measured on one 1024^2 image:
// plain accesses + agent-scope release / acquire fences (buffer_wbl2 / buffer_inv sc1 by every wave) 63 us per pass;
// 8-byte sc1 accesses of whole tiles 16.6 us (2.6 us of it the re-read: twice 8 MB from beyond L2); halo-only 16-byte
// accesses: see DESIGN.md.
typedef unsigned int wgr_u4 __attribute__((ext_vector_type(4)));
constexpr int WGR_SC1 = 16;                                    // aux bit 4 of the raw buffer intrinsics on gfx94x/gfx950
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wgr_rsrc(double *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000);
}
// address = base + voff (per lane) + soff (wave-uniform, an SGPR: the R row addresses of a wave cost ONE VGPR)
__device__ __forceinline__ double2 wgr_ld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const wgr_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, WGR_SC1);
    double2 d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}
__device__ __forceinline__ void wgr_st2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double2 d)
{
    wgr_u4 v;
    __builtin_memcpy(&v, &d, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, WGR_SC1);
}

// Forms of a resident tile; the exchange protocol is the same code for all of them.
constexpr int WGF_COEF = 0;     // 8 waves, full matrix rows in registers (20 VGPRs per tile row): any dictionary
constexpr int WGF_TALL = 1;     // 16 waves, field in registers, codes in LDS, matrix rows looked up in every sweep
constexpr int WGF_SYM = 2;      // NW waves, link-symmetric matrix rows in registers (14 VGPRs per tile row), unguarded systems
// (lut, edge, codes_lds: the workgroup's LDS -- dictionary, mailbox, and for tall tiles the codes of every tile row --, declared
// by the kernel; first: the wave's first row within the tile, wave * R unless the rows are dealt by age, k_sweep_wgage)
template <int T, int R, int NW, bool FMA, bool GUARD, int FORM, bool SYM>
__device__ __forceinline__ void wgres_body(double *lut, double2 (&edge)[2][NW][2][64], unsigned *codes_lds, const int first,
                                                                   const double *__restrict__ lut_g,
                                                                   const uint16_t *__restrict__ code, double *xa,
                                                                   double *xb, int nx, int ny, int img_stride,
                                                                   int dom_lo, int own_lo, int own_h, int cpi, int ly,
                                                                   const uint8_t *__restrict__ active, int ntx, int gy,
                                                                   int xmajor, int allb, int nrows, int shift,
                                                                   double omw, int npass, unsigned *flags,
                                                                   unsigned base, unsigned *abort_flag,
                                                                   unsigned xbytes, int stall_tile,
                                                                   unsigned long long *__restrict__ stamps)
{
    constexpr bool TALL = FORM == WGF_TALL;
    static_assert(T >= 1 && T <= 8 && R >= 3, "unsupported tile");
    static_assert(NW * R - 2 * T >= 1, "tile owns no row");
    static_assert(FORM != WGF_SYM || !GUARD, "the symmetric form has no guarded variant");
    static_assert(LUT_PLANES * LUT_MAX_ROWS <= 6 * NW * 64, "dictionary fetch assumes <= 6 doubles per thread");
    constexpr int HW = (T + 1) & ~1;
    constexpr int WOUT = TB_COLS - 2 * HW;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned total = (unsigned)ntx * (unsigned)gy;
    const unsigned per = (total + 7u) / 8u;
    const double2 zero = make_double2(0.0, 0.0);
    int par = 0;

    // ONE tile per workgroup, numbered like k_sweep_wgtile's first round
    const unsigned kk = blockIdx.x >> 3;
    const unsigned bt = (blockIdx.x & 7u) * per + kk;
    if (kk >= per || bt >= total) return;                          // workgroup-uniform
    const int tx = xmajor ? (int)(bt % (unsigned)ntx) : (int)(bt / (unsigned)gy);
    const int bty = xmajor ? (int)(bt / (unsigned)ntx) : (int)(bt % (unsigned)gy);
    const int img = bty / cpi;
    if (active && !active[img]) return;                            // frozen image: all its tiles return, nobody waits for them
    const int row_lo = dom_lo + img * img_stride, row_hi = row_lo + ny;
    const int own0 = own_lo + img * img_stride;
    const int ry0 = own0 + (bty - img * cpi) * ly;
    const int ry1 = min(ry0 + ly, own0 + own_h);
    // a tall tile keeps no rows above the first row of its image (there is no halo beyond a wall): its 16R rows then reach
    // T rows further down, which is what lets ONE tile hold a whole 128-row image (api_solve.hip, wgl_row_tiles)
    const int w0 = (TALL ? max(ry0 - T, row_lo) : ry0 - T) + first;
    const int ld_lo = max(ry0 - T, row_lo), ld_hi = min(ry1 + T, row_hi);
    const int col = tx * WOUT - shift + 2 * lane;
    const bool in_x = col >= 0 && col < nx;
    const int out_lo = (tx == 0) ? 0 : tx * WOUT - shift + HW;
    const int out_hi = (tx == ntx - 1) ? nx : tx * WOUT - shift + TB_COLS - HW;
    const bool st_x = in_x && (col >= out_lo) && (col < out_hi);
    const bool wall = allb || tx == 0 || tx == ntx - 1;

    // diagnostics (tools/wgr_stamps.py): per tile 12 clocks = entry, then for passes 0..2 {neighbours seen, halo in, swept,
    // rim stored and flag raised}; pass 0 has nobody to wait for (slot 0 = entry) and the last pass raises no flag
    unsigned long long *st = (stamps && threadIdx.x == 0) ? stamps + (size_t)bt * 12 : nullptr;
    if (st) st[0] = wall_clock64();
    // the neighbour this lane polls between passes (lanes 0..7 of wave 0; -1: none -- a wall, or another image)
    int nb = -1;
    if (threadIdx.x < 8) {
        const int q = (int)threadIdx.x < 4 ? (int)threadIdx.x : (int)threadIdx.x + 1;    // 0..8 without the centre
        const int tx2 = tx + q % 3 - 1, by2 = bty + q / 3 - 1;
        if (tx2 >= 0 && tx2 < ntx && by2 >= img * cpi && by2 < (img + 1) * cpi)
            nb = xmajor ? by2 * ntx + tx2 : tx2 * gy + by2;
    }

    // a tile that is a whole image (one strip, one row tile: a 128^2 image of a stack on a tall tile) has nobody to
    // exchange with: it neither stores its rim nor raises a flag between passes
    const bool alone = !__syncthreads_or(nb >= 0);

    double2 xr[R];
    unsigned cc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rr = w0 + r;
        const bool ok = in_x && rr >= ld_lo && rr < ld_hi;
        const size_t p = (size_t)(ok ? rr : 0) * nx + (ok ? col : 0);
        // the field is read through device-coherent loads from the first pass on (see wgr_ld2): no line of it is ever
        // brought into this XCD's L2 by a plain load that a later coherent load could find there
        const double2 vx = wgr_ld2(wgr_rsrc(xa, xbytes), (unsigned)(ok ? col : 0) * 8u,
                                   (unsigned)((rr >= ld_lo && rr < ld_hi ? rr : 0) * nx) * 8u);
        const unsigned vc = *reinterpret_cast<const uint32_t *>(code + p);
        xr[r] = ok ? vx : zero;
        cc[r] = ok ? vc : 0u;
        if constexpr (TALL) codes_lds[(first + r) * 64 + lane] = cc[r];         // read back by the same lane only
    }
    load_lut<NW * 64>(lut, lut_g, nrows);

    auto passes = [&](auto wall_tag) __attribute__((always_inline)) {
        constexpr bool WALL = decltype(wall_tag)::value;
        // the coefficient-resident forms look their matrix rows up ONCE per launch
        WgtCoef k[FORM == WGF_COEF ? R : 1];
        WgsCoef<FORM == WGF_SYM ? R : 1> ks;
        double2 bb[(WALL && !TALL) ? R : 1];
        if constexpr (FORM == WGF_COEF) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double2 b_;
                wgt_lookup<WALL>(lut, cc[r], k[r], b_);
                if constexpr (WALL) bb[r] = b_;
            }
        }
        if constexpr (FORM == WGF_SYM) wgs_lookup<R, WALL>(lut, cc, ks, bb);
        const __amdgpu_buffer_rsrc_t ra = wgr_rsrc(xa, xbytes), rb = wgr_rsrc(xb, xbytes);
        // what a tile exchanges with its neighbours: its halo (read) and the rim of its owned cells (written) -- the owned
        // cells themselves never leave the registers between two passes of a launch
        const bool halo_x = in_x && !st_x;                         // this lane's two columns belong to a neighbouring strip
        const bool rim_x = st_x && (col < out_lo + HW || col >= out_hi - HW);
        // byte offset of this lane's cell pair within a row; the row's offset is wave-uniform (rows above the array and lanes
        // left of it are never addressed: they fail `ok` / `st_x`; the buffer's range check sees voff + soff unwrapped)
        const unsigned voff = (unsigned)col * 8u;
#pragma unroll 1
        for (int p = 0; p < npass; ++p) {
            const __amdgpu_buffer_rsrc_t src = (p & 1) ? rb : ra, dst = (p & 1) ? ra : rb;
            const bool last = p + 1 == npass;
            if (p > 0) {
                int bad = 0;
                if (nb >= 0) {
                    const unsigned want = base + (unsigned)p;
                    const unsigned long long t0 = wall_clock64();
                    unsigned polls = 0;
                    while ((int)(__hip_atomic_load(flags + (size_t)nb * WGR_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                        __builtin_amdgcn_s_sleep(1);
                        if ((++polls & 31u) == 0u &&
                            (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                             wall_clock64() - t0 > WGR_TIMEOUT)) {
                            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bad = 1;
                            break;
                        }
                    }
                }
                if (__syncthreads_or(bad)) return;                 // workgroup-uniform: decided by wave 0's polling lanes
                if (st && p < 3) st[4 * p] = wall_clock64();
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int rr = w0 + r;
                    const bool ok = in_x && rr >= ld_lo && rr < ld_hi;
                    if (ok && (halo_x || rr < ry0 || rr >= ry1))
                        xr[r] = wgr_ld2(src, voff, (unsigned)(rr * nx) * 8u);
                }
            }
            if (st && p < 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st[4 * p + 1] = wall_clock64();
            }
            if constexpr (TALL) wgl_sweeps<T, R, FMA, GUARD, WALL, SYM>(xr, codes_lds + first * 64 + lane, lut, edge, par, wave, lane, w0, ry0, ry1, row_lo, row_hi, omw);
            else if constexpr (FORM == WGF_SYM) wgs_sweeps<T, R, NW, FMA, WALL>(xr, ks, bb, edge, par, wave, lane, w0, ry0, ry1, row_lo, row_hi, omw);
            else wgt_sweeps<T, R, FMA, GUARD, WALL>(xr, k, bb, edge, par, wave, lane, w0, ry0, ry1, row_lo, row_hi, omw, nullptr);
            if (st && p < 3) st[4 * p + 2] = wall_clock64();
            if (alone && !last) continue;                          // workgroup-uniform
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int rr = w0 + r;
                if (st_x && rr >= ry0 && rr < ry1 && (last || rim_x || rr < ry0 + T || rr >= ry1 - T))
                    wgr_st2(dst, voff, (unsigned)(rr * nx) * 8u, xr[r]);
            }
            if (!last) {                                           // the last pass is published by the end of the launch
                if ((int)bt == stall_tile) return;                 // test hook (tuning "tb_debug_stall"): a tile that never publishes
                // Every wave waits until ITS rim stores are acknowledged (written through to the coherence point), then the
                // barrier, then the flag.  The wait must be spelled out: a workgroup-scope barrier does not wait for global
                // stores on gfx950 (hipcc emits s_waitcnt lgkmcnt(0) only) -- without it the flag can overtake the data
                // (seen once, as a 1e-10 deviation of one Deff, with three contexts loading the memory system).
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0)
                    __hip_atomic_store(flags + (size_t)bt * WGR_FLAG_STRIDE, base + (unsigned)p + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (st && p < 3) st[4 * p + 3] = wall_clock64();
            }
        }
    };
    if (wall) passes(TbTag<true>{});
    else passes(TbTag<false>{});
}

// TALL = the 16-wave form (matrix rows looked up in every sweep); otherwise the 8-wave form with the matrix rows in registers.
template <int T, int R, bool FMA, bool GUARD, bool TALL = false, bool SYM = false>
__global__ __launch_bounds__((TALL ? WGL_WAVES : WGT_WAVES) * 64, (TALL ? 4 : 2)) void k_sweep_wgres(const double *__restrict__ lut_g,
                                                                   const uint16_t *__restrict__ code, double *xa,
                                                                   double *xb, int nx, int ny, int img_stride,
                                                                   int dom_lo, int own_lo, int own_h, int cpi, int ly,
                                                                   const uint8_t *__restrict__ active, int ntx, int gy,
                                                                   int xmajor, int allb, int nrows, int shift,
                                                                   double omw, int npass, unsigned *flags,
                                                                   unsigned base, unsigned *abort_flag,
                                                                   unsigned xbytes, int stall_tile,
                                                                   unsigned long long *__restrict__ stamps)
{
    static_assert(R >= 4, "unsupported tile");
    constexpr int NW = TALL ? WGL_WAVES : WGT_WAVES;
    __shared__ double lut[LUT_DOUBLES];
    __shared__ double2 edge[2][NW][2][64];
    __shared__ unsigned codes_lds[TALL ? NW * R * 64 : 1];         // tall tiles: the codes of every tile row
    wgres_body<T, R, NW, FMA, GUARD, (TALL ? WGF_TALL : WGF_COEF), SYM>(
        lut, edge, codes_lds, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * R, lut_g, code, xa, xb, nx, ny, img_stride, dom_lo, own_lo, own_h, cpi, ly, active, ntx, gy, xmajor, allb, nrows, shift, omw,
        npass, flags, base, abort_flag, xbytes, stall_tile, stamps);
}

// The link-symmetric form: 12 waves per workgroup = 3 per SIMD (168 VGPRs), R rows per wave.
template <int T, int R, bool FMA>
__global__ __launch_bounds__(WGS_WAVES * 64, 3) void k_sweep_wgsym(const double *__restrict__ lut_g,
                                                                   const uint16_t *__restrict__ code, double *xa,
                                                                   double *xb, int nx, int ny, int img_stride,
                                                                   int dom_lo, int own_lo, int own_h, int cpi, int ly,
                                                                   const uint8_t *__restrict__ active, int ntx, int gy,
                                                                   int xmajor, int allb, int nrows, int shift,
                                                                   double omw, int npass, unsigned *flags,
                                                                   unsigned base, unsigned *abort_flag,
                                                                   unsigned xbytes, int stall_tile,
                                                                   unsigned long long *__restrict__ stamps)
{
    __shared__ double lut[LUT_DOUBLES];
    __shared__ double2 edge[2][WGS_WAVES][2][64];
    __shared__ unsigned codes_lds[1];
    wgres_body<T, R, WGS_WAVES, FMA, false, WGF_SYM, true>(
        lut, edge, codes_lds, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * R, lut_g, code, xa, xb, nx, ny, img_stride, dom_lo, own_lo, own_h, cpi, ly, active, ntx, gy, xmajor, allb, nrows, shift, omw,
        npass, flags, base, abort_flag, xbytes, stall_tile, stamps);
}

// Tall tiles with the rows dealt by age.  A tile's waves meet at a barrier in every sweep and the SIMD serves its four waves oldest
// first (wave w of the workgroup: SIMD w % 4, age w / 4): with R rows each they arrive after 3 150 / 4 050 / 5 040 / 6 350 clocks
// at 2048^2 (R = 11, profiles/r04_tile_wave_service_order.log).  Here the waves of age 0..3 hold RA >= RB >= RC >= RD rows
// (RA + RB + RC + RD = 4R: the same tile, the same halo, the same number of tiles) and each age runs the pass loop instantiated
// for ITS row count -- the row count must be a compile-time constant: a run-time one cost the sweeps their schedule and the
// field its registers (tools/experiments/r04_tall_rows_by_age_runtime.patch).  All four bodies pass the same barriers.
template <int T, int RA, int RB, int RC, int RD, bool FMA, bool GUARD, bool SYM>
__global__ __launch_bounds__(WGL_WAVES * 64, 4) void k_sweep_wgage(const double *__restrict__ lut_g,
                                                                   const uint16_t *__restrict__ code, double *xa,
                                                                   double *xb, int nx, int ny, int img_stride,
                                                                   int dom_lo, int own_lo, int own_h, int cpi, int ly,
                                                                   const uint8_t *__restrict__ active, int ntx, int gy,
                                                                   int xmajor, int allb, int nrows, int shift,
                                                                   double omw, int npass, unsigned *flags,
                                                                   unsigned base, unsigned *abort_flag,
                                                                   unsigned xbytes, int stall_tile,
                                                                   unsigned long long *__restrict__ stamps)
{
    static_assert(RA >= RB && RB >= RC && RC >= RD && RD >= 3, "rows by age: oldest first");
    __shared__ double lut[LUT_DOUBLES];
    __shared__ double2 edge[2][WGL_WAVES][2][64];
    __shared__ unsigned codes_lds[4 * (RA + RB + RC + RD) * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int age = wave >> 2, nth = wave & 3;
#define WGAGE_BODY(R_, FIRST_)                                                                                              \
    wgres_body<T, R_, WGL_WAVES, FMA, GUARD, WGF_TALL, SYM>(lut, edge, codes_lds, (FIRST_) + nth * R_, lut_g, code, xa, xb, nx, ny,   \
        img_stride, dom_lo, own_lo, own_h, cpi, ly, active, ntx, gy, xmajor, allb, nrows, shift, omw, npass, flags, base,   \
        abort_flag, xbytes, stall_tile, stamps)
    if (age == 0) WGAGE_BODY(RA, 0);
    else if (age == 1) WGAGE_BODY(RB, 4 * RA);
    else if (age == 2) WGAGE_BODY(RC, 4 * (RA + RB));
    else WGAGE_BODY(RD, 4 * (RA + RB + RC));
#undef WGAGE_BODY
}

// The link-symmetric form with the rows dealt by age (three waves per SIMD: ages 0..2), as k_sweep_wgage.
template <int T, int RA, int RB, int RC, bool FMA>
__global__ __launch_bounds__(WGS_WAVES * 64, 3) void k_sweep_wgsage(const double *__restrict__ lut_g,
                                                                   const uint16_t *__restrict__ code, double *xa,
                                                                   double *xb, int nx, int ny, int img_stride,
                                                                   int dom_lo, int own_lo, int own_h, int cpi, int ly,
                                                                   const uint8_t *__restrict__ active, int ntx, int gy,
                                                                   int xmajor, int allb, int nrows, int shift,
                                                                   double omw, int npass, unsigned *flags,
                                                                   unsigned base, unsigned *abort_flag,
                                                                   unsigned xbytes, int stall_tile,
                                                                   unsigned long long *__restrict__ stamps)
{
    static_assert(RA >= RB && RB >= RC && RC >= 3, "rows by age: oldest first");
    __shared__ double lut[LUT_DOUBLES];
    __shared__ double2 edge[2][WGS_WAVES][2][64];
    __shared__ unsigned codes_lds[1];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int age = wave >> 2, nth = wave & 3;
#define WGSAGE_BODY(R_, FIRST_)                                                                                             \
    wgres_body<T, R_, WGS_WAVES, FMA, false, WGF_SYM, true>(lut, edge, codes_lds, (FIRST_) + nth * R_, lut_g, code, xa, xb, nx, ny,   \
        img_stride, dom_lo, own_lo, own_h, cpi, ly, active, ntx, gy, xmajor, allb, nrows, shift, omw, npass, flags, base,   \
        abort_flag, xbytes, stall_tile, stamps)
    if (age == 0) WGSAGE_BODY(RA, 0);
    else if (age == 1) WGSAGE_BODY(RB, 4 * RA);
    else WGSAGE_BODY(RC, 4 * (RA + RB));
#undef WGSAGE_BODY
}

}  // namespace deff
