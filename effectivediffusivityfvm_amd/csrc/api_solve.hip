// api_solve.hip -- launch plans, sweeps and the solve loops of libdeff_amd.so: the host side of
// JacobiGPU (Deff2DGPU/Deff2D.cuh:1163-1314) for one image, a stack of images and a stream of
// images through a stack.  See ctx.hpp for the file map, DESIGN.md section 4-6 for the design.
#include "ctx.hpp"
#include "kernels_dict.hpp"
#include "kernels_sweep.hpp"
#include "kernels_tb.hpp"
#include "kernels_wgtile.hpp"
#include <array>
#include <mutex>

// ----------------------------------------------------------- sweeps -------

// Workgroups of the temporally blocked kernel that are resident at once on this device.
template <int T, bool F, bool G>
static int tb_occ(int *per_cu)
{
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_sweep_matfree_tb<T, F, G>, 256, 0));
    return DEFF_OK;
}

// Only 2 cells per lane are instantiated: 4 per lane (twice the work per wave, 244 VGPRs,
// 2 waves per SIMD) measured 20 % slower at 4096^2 -- the kernel needs the wave-level
// parallelism more than it needs the smaller strip overlap.
#define TB_DISPATCH_FG(TC_, F_, G_, CALL)                                                   \
    switch (((F_) ? 2 : 0) + ((G_) ? 1 : 0)) {                                              \
    case 1: { CALL(TC_, false, true); } break;  case 2: { CALL(TC_, true, false); } break; \
    case 3: { CALL(TC_, true, true); } break;   default: { CALL(TC_, false, false); } break; \
    }
#define TB_DISPATCH(T_, F_, G_, CALL)                                                       \
    do {                                                                                    \
        switch (T_) {                                                                       \
        case 1: TB_DISPATCH_FG(1, F_, G_, CALL); break;                                     \
        case 2: TB_DISPATCH_FG(2, F_, G_, CALL); break;                                     \
        case 4: TB_DISPATCH_FG(4, F_, G_, CALL); break;                                     \
        case 6: TB_DISPATCH_FG(6, F_, G_, CALL); break;                                     \
        default: TB_DISPATCH_FG(8, F_, G_, CALL); break;                                    \
        }                                                                                   \
    } while (0)

static int tb_resident_blocks(const deff_ctx *c, int T, bool fma, bool guard, int *resident)
{
    int per_cu = 0, cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
#define OCC_CALL(T_, C_, G_) TRY((tb_occ<T_, C_, G_>(&per_cu)))
    TB_DISPATCH(T, fma, guard, OCC_CALL);
#undef OCC_CALL
    if (per_cu < 1) per_cu = 1;
    *resident = per_cu * cus;
    return DEFF_OK;
}

// ---- workgroup-tile form (kernels_wgtile.hpp): instantiated for T in {4, 8} x R in {4, 6, 7} rows per wave
// (R = 8 needs 256 VGPRs + 148 B of scratch per lane and ran 40 % SLOWER than R = 6: the spills sit in the sweep loop)
#define WGT_DISPATCH_FG(T_, R_, F_, G_, CALL)                                                   \
    switch (((F_) ? 2 : 0) + ((G_) ? 1 : 0)) {                                                 \
    case 1: { CALL(T_, R_, false, true); } break;  case 2: { CALL(T_, R_, true, false); } break; \
    case 3: { CALL(T_, R_, true, true); } break;   default: { CALL(T_, R_, false, false); } break; \
    }
#define WGT_DISPATCH_R(T_, R_, F_, G_, CALL)                                                    \
    if ((R_) == 4) { WGT_DISPATCH_FG(T_, 4, F_, G_, CALL) }                                     \
    else if ((R_) == 6) { WGT_DISPATCH_FG(T_, 6, F_, G_, CALL) }                                \
    else { WGT_DISPATCH_FG(T_, 7, F_, G_, CALL) }
#define WGT_DISPATCH(T_, R_, F_, G_, CALL)                                                      \
    do {                                                                                       \
        if ((T_) == 4) { WGT_DISPATCH_R(4, R_, F_, G_, CALL) } else { WGT_DISPATCH_R(8, R_, F_, G_, CALL) } \
    } while (0)

template <int T, int R, bool F, bool G>
static int wgt_occ(int *per_cu)
{
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_sweep_wgtile<T, R, F, G>, WGT_WAVES * 64, 0));
    return DEFF_OK;
}

static int wgt_resident_blocks(const deff_ctx *c, int T, int R, bool fma, bool guard, int *resident)
{
    int per_cu = 0, cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
#define OCC_CALL(T_, R_, C_, G_) TRY((wgt_occ<T_, R_, C_, G_>(&per_cu)))
    WGT_DISPATCH(T, R, fma, guard, OCC_CALL);
#undef OCC_CALL
    if (per_cu < 1) per_cu = 1;
    *resident = per_cu * cus;
    return DEFF_OK;
}

// tall resident tiles (kernels_wgtile.hpp: 16 waves, matrix rows looked up in every sweep): T = 8, R in WGL_ROWS
#define WGL_DISPATCH(R_, F_, G_, CALL)                                                          \
    do {                                                                                       \
        if ((R_) == 4) { WGT_DISPATCH_FG(8, 4, F_, G_, CALL) }                                  \
        else if ((R_) == 5) { WGT_DISPATCH_FG(8, 5, F_, G_, CALL) }                             \
        else if ((R_) == 6) { WGT_DISPATCH_FG(8, 6, F_, G_, CALL) }                             \
        else if ((R_) == 7) { WGT_DISPATCH_FG(8, 7, F_, G_, CALL) }                             \
        else if ((R_) == 8) { WGT_DISPATCH_FG(8, 8, F_, G_, CALL) }                             \
        else if ((R_) == 9) { WGT_DISPATCH_FG(8, 9, F_, G_, CALL) }                             \
        else if ((R_) == 10) { WGT_DISPATCH_FG(8, 10, F_, G_, CALL) }                           \
        else if ((R_) == 11) { WGT_DISPATCH_FG(8, 11, F_, G_, CALL) }                           \
        else if ((R_) == 12) { WGT_DISPATCH_FG(8, 12, F_, G_, CALL) }                           \
        else if ((R_) == 13) { WGT_DISPATCH_FG(8, 13, F_, G_, CALL) }                           \
        else { WGT_DISPATCH_FG(8, 14, F_, G_, CALL) }                                           \
    } while (0)
// (R = 16 -- 256-row tiles, images up to ~2600^2 -- spills inside the sweep loop: 9.9 us per sweep, slower than streaming)
static const int WGL_ROWS[] = {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14};
// Row tiles a tall-tile image of own_h rows needs at R rows per wave.  A tall tile carries no halo rows beyond a wall of
// the mesh (its rows start at the image's first row, kernels_wgtile.hpp), so ONE tile holds 16R rows, two tiles 16R - T
// each, three or more 16R - 2T (the inner ones).  A 128^2 image of a stack is one tile of 16 x 8 rows: nothing recomputed,
// nobody to wait for.
static int wgl_row_tiles(int own_h, int R, int T)
{
    const int rows = WGL_WAVES * R;
    if (own_h <= rows) return 1;
    if (own_h <= 2 * (rows - T)) return 2;
    const int lymax = rows - 2 * T;
    return (own_h + lymax - 1) / lymax;
}
static bool wgl_has_R(int R) { for (int r : WGL_ROWS) if (r == R) return true; return false; }

template <int T, int R, bool F, bool G>
static int wgl_occ(int *per_cu)
{
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_sweep_wgres<T, R, F, G, true>, WGL_WAVES * 64, 0));
    return DEFF_OK;
}

static int wgl_resident_blocks(const deff_ctx *c, int R, bool fma, bool guard, int *resident)
{
    // asked for every candidate R by every plan: remembered per (device, R, arithmetic, guard)
    static std::mutex mu;
    static int cache[64][16][4];
    const int d = c->device, key = (fma ? 2 : 0) + (guard ? 1 : 0);
    if (d >= 0 && d < 64 && R >= 0 && R < 16) {
        std::lock_guard<std::mutex> lock(mu);
        if (cache[d][R][key] > 0) { *resident = cache[d][R][key]; return DEFF_OK; }
    }
    int per_cu = 0, cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
#define OCC_CALL(T_, R_, C_, G_) TRY((wgl_occ<T_, R_, C_, G_>(&per_cu)))
    WGL_DISPATCH(R, fma, guard, OCC_CALL);
#undef OCC_CALL
    *resident = per_cu * cus;
    if (d >= 0 && d < 64 && R >= 0 && R < 16 && *resident > 0) {
        std::lock_guard<std::mutex> lock(mu);
        cache[d][R][key] = *resident;
    }
    return DEFF_OK;
}

// link-symmetric 12-wave tiles (kernels_wgtile.hpp, k_sweep_wgsym): T = 8, 6 or 4, R in WGS_ROWS
#define WGS_DISPATCH_T(T_, R_, F_, CALL)                                                        \
    do {                                                                                       \
        if ((R_) == 4) { if (F_) { CALL(T_, 4, true); } else { CALL(T_, 4, false); } }          \
        else { if (F_) { CALL(T_, 5, true); } else { CALL(T_, 5, false); } }                    \
    } while (0)
#define WGS_DISPATCH(T_, R_, F_, CALL)                                                          \
    do {                                                                                       \
        if ((T_) == 6) WGS_DISPATCH_T(6, R_, F_, CALL);                                        \
        else if ((T_) == 4) WGS_DISPATCH_T(4, R_, F_, CALL);                                   \
        else WGS_DISPATCH_T(8, R_, F_, CALL);                                                  \
    } while (0)
// (R = 6 -- 72-row tiles, images up to ~1230^2 -- needs 168 VGPRs + ~100 B of scratch, which lands in the halo exchange: 1152^2
// 652 G against 704 G on tall tiles: not instantiated;
// R = 3 -- 36-row tiles -- is no faster than 8 waves x 4 rows, see plan_blocked_pass)
static const int WGS_ROWS[] = {4, 5};
static bool wgs_has_R(int R) { for (int r : WGS_ROWS) if (r == R) return true; return false; }

template <int T, int R, bool F>
static int wgs_occ(int *per_cu)
{
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_sweep_wgsym<T, R, F>, WGS_WAVES * 64, 0));
    return DEFF_OK;
}

static int wgs_resident_blocks(const deff_ctx *c, int T, int R, bool fma, int *resident)
{
    static std::mutex mu;
    static int cache[64][8][2][3];
    const int d = c->device, t6 = T == 6 ? 1 : T == 4 ? 2 : 0;
    if (d >= 0 && d < 64 && R >= 0 && R < 8) {
        std::lock_guard<std::mutex> lock(mu);
        if (cache[d][R][fma][t6] > 0) { *resident = cache[d][R][fma][t6]; return DEFF_OK; }
    }
    int per_cu = 0, cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
#define OCC_CALL(T_, R_, C_) TRY((wgs_occ<T_, R_, C_>(&per_cu)))
    WGS_DISPATCH(T, R, fma, OCC_CALL);
#undef OCC_CALL
    *resident = per_cu * cus;
    if (d >= 0 && d < 64 && R >= 0 && R < 8 && *resident > 0) {
        std::lock_guard<std::mutex> lock(mu);
        cache[d][R][fma][t6] = *resident;
    }
    return DEFF_OK;
}

template <int T, int R, bool F, bool G>
static int wgr_occ(int *per_cu)
{
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_sweep_wgres<T, R, F, G>, WGT_WAVES * 64, 0));
    return DEFF_OK;
}

static int wgr_resident_blocks(const deff_ctx *c, int T, int R, bool fma, bool guard, int *resident)
{
    int per_cu = 0, cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
#define OCC_CALL(T_, R_, C_, G_) TRY((wgr_occ<T_, R_, C_, G_>(&per_cu)))
    WGT_DISPATCH(T, R, fma, guard, OCC_CALL);
#undef OCC_CALL
    *resident = per_cu * cus;
    return DEFF_OK;
}

// Resident launches of this process, chained per device: a resident kernel must have all its workgroups on the chip to
// make progress, so two of them (two contexts on one GPU: deff2d --devices 0,0, a thread pool) must never be dispatched
// side by side -- each waits for the previous one's end.  Finite kernels of other streams only delay a resident launch.
static std::mutex g_res_mu;
static hipEvent_t g_res_ev[64];
static bool g_res_has[64];
static int g_res_users[64];          // contexts alive per device: the chain's event goes with the last of them

void resident_chain_ctx_created(int device)
{
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lock(g_res_mu);
    ++g_res_users[device];
}

// (called by deff_destroy with the device current and the context's stream drained)
void resident_chain_ctx_destroyed(int device)
{
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lock(g_res_mu);
    if (--g_res_users[device] > 0 || !g_res_has[device]) return;
    (void)hipEventDestroy(g_res_ev[device]);     // nobody is left to wait on it; the next context of this device starts a new chain
    g_res_has[device] = false;
}

static hipError_t resident_chain_begin(const deff_ctx *c)
{
    const int d = c->device;
    if (d < 0 || d >= 64) return hipSuccess;
    if (!g_res_has[d]) {
        hipError_t e = hipEventCreateWithFlags(&g_res_ev[d], hipEventDisableTiming);
        if (e != hipSuccess) return e;
        g_res_has[d] = true;
        return hipSuccess;                                          // nothing to wait for yet
    }
    return hipStreamWaitEvent(c->stream, g_res_ev[d], 0);
}

static hipError_t resident_chain_end(const deff_ctx *c)
{
    const int d = c->device;
    if (d < 0 || d >= 64 || !g_res_has[d]) return hipSuccess;
    return hipEventRecord(g_res_ev[d], c->stream);
}

template <class Kernel>
static hipError_t launch_resident(deff_ctx *c, const SweepPlan &pl, Kernel kernel, int threads, double *xa, double *xb, int npass,
                                  unsigned base)
{
    unsigned long long *stamps = c->tb_stamps;
    unsigned xbytes = (unsigned)(c->n * sizeof(double));
    int stall_tile = c->tb_debug_stall - 1;
    if (stall_tile >= 0 && c->tb_debug_stall_skip > 0) { --c->tb_debug_stall_skip; stall_tile = -1; }   // tests: a LATER launch stalls
    const double *lut = c->lut;
    const uint16_t *code = c->code;
    int nx = c->nx, ny = c->mesh_ny, img_stride = c->ny, dom_lo = c->dom_lo, own_lo = pl.own_lo, own_h = pl.own_h;
    int cpi = pl.tcpi, ly = pl.LY, ntx = pl.ntx, gy = pl.tgy, xmajor = c->tb_xmajor;
    int allb = (c->lut_allb || c->nx != c->nxt) ? 1 : 0, nrows = c->lut_nrows, shift = pl.shift;
    const uint8_t *mask = c->masked ? c->active : nullptr;
    double omw = pl.omw;
    unsigned *flags = c->res_flags, *abort_flag = c->res_abort;
    std::lock_guard<std::mutex> lock(g_res_mu);
    hipError_t e = resident_chain_begin(c);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(pl.tblocks), dim3(threads), 0, c->stream, lut, code, xa,
                       xb, nx, ny, img_stride, dom_lo, own_lo, own_h, cpi, ly, mask, ntx, gy, xmajor, allb, nrows, shift,
                       omw, npass, flags, base, abort_flag, xbytes, stall_tile, stamps);
    e = hipPeekAtLastError();
    if (e != hipSuccess) return e;
    return resident_chain_end(c);
}

template <int T, int R, bool F, bool G, bool TALL = false, bool SYM = false>
static hipError_t launch_wgres(deff_ctx *c, const SweepPlan &pl, double *xa, double *xb, int npass, unsigned base)
{
    return launch_resident(c, pl, k_sweep_wgres<T, R, F, G, TALL, SYM>, (TALL ? WGL_WAVES : WGT_WAVES) * 64, xa, xb, npass, base);
}

template <int RA, int RB, int RC, int RD, bool F, bool G, bool SYM>
static hipError_t launch_wgage(deff_ctx *c, const SweepPlan &pl, double *xa, double *xb, int npass, unsigned base)
{
    return launch_resident(c, pl, k_sweep_wgage<8, RA, RB, RC, RD, F, G, SYM>, WGL_WAVES * 64, xa, xb, npass, base);
}

// Rows by age for the tall tiles of R rows per wave (k_sweep_wgage): the 4R rows of a SIMD's four waves, oldest first.  Measured,
// not derived (profiles/r04_tall_rows_by_age_kbench.log: four candidate sets per R, one process, against equal rows): what wins
// gives the youngest wave about half its share and keeps the three older ones level; bodies of 9 and more rows spill, which is why
// R = 7 stops at 8 rows, R = 9 deals one row only, and R = 13 found no set that beats equal rows -- four equal bodies in this kernel
// run 3-4 % behind the one-body kernel, which is what every set has to earn first (14 x 4 has nothing to deal).  Unguarded systems,
// link-symmetric (7 lookups per row) or not (10: the 3-phase assembly with impermeable solid), both arithmetics.
#define WGAGE_SETS(X) X(5, 6, 6, 5, 3) X(6, 8, 8, 5, 3) X(7, 8, 8, 8, 4) X(8, 9, 9, 9, 5) X(9, 10, 9, 9, 8) X(10, 12, 12, 10, 6) X(11, 13, 13, 11, 7) X(12, 13, 13, 13, 9)
static bool wgage_has(int R)
{
#define X(R_, A_, B_, C_, D_) if (R == R_) return true;
    WGAGE_SETS(X)
#undef X
    return false;
}

template <int T, int R, bool F>
static hipError_t launch_wgsym(deff_ctx *c, const SweepPlan &pl, double *xa, double *xb, int npass, unsigned base)
{
    return launch_resident(c, pl, k_sweep_wgsym<T, R, F>, WGS_WAVES * 64, xa, xb, npass, base);
}

// Did a resident launch give up?  Reads the flag (one 4-byte copy + a stream synchronisation) only when such a launch was
// enqueued since the last look.  A raised flag means some tile stopped updating -- another process's kernels held part of
// the chip, a CU mask -- and what the buffers hold is not a Jacobi iterate.  Nothing is lost: the field the interval started
// from was copied aside in front of its first resident launch (enqueue_sweeps), so the interval is redone from that copy
// with one launch per pass, and the context keeps launching that way (the condition that starved the tiles is not ours
// to lift).  The caller sees the same bits it would have seen; deff_get_plan("tb_fallbacks") counts the occurrences.
int resident_check(deff_ctx *c)
{
    if (!c->res_pending) return DEFF_OK;
    unsigned h = 0;
    HIP_TRY(hipMemcpyAsync(&h, c->res_abort, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->res_pending = false;
    const int64_t redo = c->res_redo;
    c->res_redo = 0;
    if (!h) return DEFF_OK;
    HIP_TRY(hipMemsetAsync(c->res_abort, 0, sizeof h, c->stream));
    c->tb_resident = 0;
    ++c->res_fallbacks;
    c->tb_debug_stall = 0;
    if (!c->res_backup) {
        c->have_field = false;
        return fail(DEFF_EHIP, "resident passes aborted and no restart copy exists (the field is invalid)");
    }
    HIP_TRY(hipMemcpyAsync(c->x[c->res_backup_cur], c->res_backup, sizeof(double) * c->n, hipMemcpyDeviceToDevice, c->stream));
    c->cur = c->res_backup_cur;
    SweepPlan pl;
    TRY(plan_sweeps(c, c->res_omega, &pl));
    const int64_t launches = c->last_launches;
    TRY(enqueue_sweeps(c, pl, redo));
    c->last_launches = launches;                                   // (the redone launches are not the caller's)
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DEFF_OK;
}

static int pick_R(int requested, int dflt)
{
    const int r = requested ? requested : dflt;
    return r >= 8 ? 8 : r >= 4 ? 4 : r >= 2 ? 2 : 1;
}

static void tile_grid(const deff_ctx *c, int cols_per_block, int rows, SweepPlan *pl)
{
    pl->rows = rows;
    pl->gx = (c->nx + cols_per_block - 1) / cols_per_block;
    pl->cpi = (c->ny + rows - 1) / rows;               // row tiles never straddle two images
    pl->gy = pl->cpi * c->nimg;
    const unsigned total = (unsigned)pl->gx * (unsigned)pl->gy;
    pl->blocks = (int)(((total + 7u) / 8u) * 8u);      // see xcd_tile()
}

// Which form of the temporally blocked pass a context gets when the caller does not say (tb_impl = 0): workgroup tiles
// (kernels_wgtile.hpp) below 4 Mi cells in the context -- one image or a stack --, where the streaming kernel has too few
// tiles to fill the chip and a tile's dependency chain sets the time of a pass; everything larger streams.  Measured,
// G cells*iter/s, streaming / workgroup tiles: one image 512^2 106 / 231, 1024^2 316 / 556, 1536^2 455 / 613, 2048^2
// 682 / 678, 4096^2 1 128 / 742; stacks 16 x 128^2 117 / 239, 64 x 128^2 364 / 683, 200 x 128^2 589 / 652, 16 x 256^2
// 322 / 418, 48 x 256^2 535 / 535, 12 x 512^2 509 / 618, 2 x 1024^2 426 / 569, 3 x 1024^2 577 / 623 (and 1 024 x 128^2,
// 16 Mi cells, whole images per wave with no halo: 1 222 streaming).
// Since the resident forms (k_sweep_wgres) the numbers above are those of ONE LAUNCH PER PASS; with all tiles on the chip
// one image runs at 512^2 306, 1024^2 840-855 (8-wave tiles), 1536^2 835-866, 2048^2 933-958, 2304^2 958-966 (tall tiles,
// which plan_sweeps() also takes just above this threshold when they fit), 2 x 1024^2 958, 16 x 512^2 998.
// Keyed on tb_ref_cells for slabs, so that every slab of an image takes the same decision (and the same T).
int default_tb_impl(const deff_ctx *c)
{
    const size_t cells = c->tb_ref_cells ? c->tb_ref_cells : c->n;
    return cells < ((size_t)1 << 22) ? 2 : 1;
}

int default_tb_T(const deff_ctx *c)
{
    // workgroup tiles: 8 sweeps per pass amortise the launch gap and the first-load latency (1024^2: T = 8 556, T = 4 457)
    if ((c->tb_impl ? c->tb_impl : default_tb_impl(c)) == 2) return 8;
    // streaming: below 4 Mi cells the launch is latency-bound and T = 4 wins; above, T = 8 everywhere (with the
    // prefetch really in flight, kernels_tb.hpp, stacks no longer prefer T = 6: 1 024 x 128^2 1 222 vs
    // 1 125 G cells*iter/s, 64 x 1024^2 1 258 vs 1 156, 16 x 1024^2 1 106 vs 1 064)
    const size_t cells = c->tb_ref_cells ? c->tb_ref_cells : c->n;
    return cells < ((size_t)1 << 22) ? 4 : 8;
}

// the instantiated sweeps-per-pass: 1, 2, 4, 6, 8 (one helper for the planner and deff_last_launches)
int clamp_tb_T(int T) { return T >= 8 ? 8 : T >= 6 ? 6 : T >= 4 ? 4 : T >= 2 ? 2 : 1; }

// Harvest the row dictionary of the explicit system (kernels_dict.hpp).  On success the context
// also has a matrix-free form (codes + tables); when the system has too many distinct rows it
// simply keeps running on the explicit kernels.
static int try_dict(deff_ctx *c)
{
    c->dict_tried = true;
    if (!c->have_explicit) return DEFF_OK;
    const size_t S = DICT_SLOTS;
    const size_t bytes = S * (8 + 4 + 8) + 16 + S * 2 + (size_t)LUT_MAX_ROWS * (8 + 48);
    TRY(ensure_scratch(c, bytes));
    char *base = (char *)c->scratch;
    DictTable t;
    t.key = (unsigned long long *)base;
    t.rep = (unsigned long long *)(base + S * 8);
    t.count = (unsigned int *)(base + S * 16);
    t.flags = (unsigned int *)(base + S * 20);
    uint16_t *d_slot2code = (uint16_t *)(base + S * 20 + 16);
    unsigned long long *d_cells = (unsigned long long *)(base + S * 22 + 16);
    double *d_rows = (double *)(base + S * 22 + 16 + (size_t)LUT_MAX_ROWS * 8);
    HIP_TRY(hipMemsetAsync(base, 0, S * 20 + 16, c->stream));
    const CoefSoA planes = soa_of(c);
    hipLaunchKernelGGL(k_dict_insert, dim3(grid_for(c->n, 4096)), dim3(256), 0, c->stream, planes, c->n, t);
    HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> key(S), rp(S);
    std::vector<unsigned int> cnt(S);
    unsigned int flags[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(key.data(), t.key, S * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(rp.data(), t.rep, S * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(cnt.data(), t.count, S * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(flags, t.flags, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flags[0]) return DEFF_OK;                                 // table overflow: far too many rows
    struct Ent { unsigned int count; unsigned long long cell; unsigned slot; };
    std::vector<Ent> ents;
    for (unsigned sl = 0; sl < S; ++sl)
        if (key[sl]) ents.push_back({cnt[sl], rp[sl] - 1, sl});
    if (ents.empty() || (int)ents.size() + 1 > LUT_MAX_ROWS) return DEFF_OK;
    // most populous rows first: the 32 commonest rows then share one conflict-free LDS bank row
    std::sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) {
        return a.count != b.count ? a.count > b.count : a.cell < b.cell;
    });
    std::vector<uint16_t> slot2code(S, 0xFFFFu);
    std::vector<unsigned long long> cells(ents.size());
    for (size_t k = 0; k < ents.size(); ++k) {
        slot2code[ents[k].slot] = (uint16_t)((k + 1) * 8);
        cells[k] = ents[k].cell;
    }
    HIP_TRY(hipMemcpyAsync(d_slot2code, slot2code.data(), S * 2, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_cells, cells.data(), cells.size() * 8, hipMemcpyHostToDevice, c->stream));
    const int nrows = (int)ents.size();
    hipLaunchKernelGGL(k_dict_gather, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, planes, d_cells, nrows, d_rows);
    TRY(dev_alloc(&c->code, c->n));
    hipLaunchKernelGGL(k_dict_encode, dim3(grid_for(c->n, 4096)), dim3(256), 0, c->stream, planes, c->n, c->nx, c->nxt, t,
                       d_slot2code, c->code);
    HIP_TRY(hipGetLastError());
    std::vector<double> rows((size_t)nrows * 6);
    HIP_TRY(hipMemcpyAsync(rows.data(), d_rows, rows.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(flags, t.flags, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flags[1]) return DEFF_OK;                                 // hash collision (astronomically unlikely): stay explicit
    c->lut_nrows = nrows + 1;
    c->lut_rows.assign((size_t)c->lut_nrows * 6, 0.0);
    memcpy(&c->lut_rows[6], rows.data(), rows.size() * 8);
    c->lut_allb = flags[2] != 0;
    c->lut_omega = NAN;
    c->have_matfree = true;
    c->links_sym = 0;
    return DEFF_OK;
}

// Runs k_links_symmetric on the current (dictionary, codes) unless that was done since they last changed; the answer is
// c->links_sym (1 yes, 2 no).  Needs c->res_abort (its flag word) and synchronises the stream.
static int check_links_symmetric(deff_ctx *c)
{
    if (c->links_sym != 0) return DEFF_OK;
    unsigned h = 1;
    TRY(resident_check(c));                                        // the abort word doubles as this kernel's flag: read it first
    if (!c->res_abort) {
        TRY(dev_alloc(&c->res_abort, 1));
        HIP_TRY(hipMemsetAsync(c->res_abort, 0, sizeof(unsigned), c->stream));
    }
    unsigned *flag = c->res_abort;
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_links_symmetric, dim3(grid_for(c->n, 2048)), dim3(256), 0, c->stream, c->lut, c->code, c->nx, c->rows,
                       c->ny, c->lut_nrows, flag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h, flag, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->links_sym = h ? 2 : 1;
    return DEFF_OK;
}

// ---- the plan of a temporally blocked pass (T sweeps), piece by piece ------------------------------------------------

// Strips of 128 columns overlapping by 2*HW; a mesh wall needs no halo (kernels_tb.hpp).
// Placement A: every strip carries its halo, also outside the first column; placement B: no halo outside a wall.  B needs
// fewer strips for narrow images (a 128-column image is ONE strip: 2x on dataset batches); where the counts tie, A measured
// equal or up to 5 % faster in one process (T = 8 at 4096^2), so B is used only when it wins.
static void plan_strips(const deff_ctx *c, int T, SweepPlan *pl)
{
    const int hw = (T + 1) & ~1, wout = TB_COLS - 2 * hw;
    const int ntx_a = (c->nx + wout - 1) / wout;
    const int ntx_b = c->nx <= TB_COLS ? 1 : (c->nx - TB_COLS + wout - 1) / wout + 1;
    const bool use_b = c->tb_wall_halo == 0 ? true : (c->tb_wall_halo == 1 ? false : ntx_b < ntx_a);
    pl->shift = use_b ? 0 : hw;
    pl->ntx = use_b ? ntx_b : ntx_a;
}

// flags (one 256-byte block per tile) and the abort word of the resident launches.  A flag holds the number of passes its
// tile has completed since the array was last cleared (c->res_epoch, compared through a signed difference in the kernel):
// a new array starts a new count, and so does an old one before the count could wrap (launch_resident_passes).
static int ensure_resident_buffers(deff_ctx *c, long tiles)
{
    if (c->res_flags_n < (size_t)tiles) {
        TRY(resident_check(c));                                     // nothing resident may still be using the old array
        if (c->res_flags) { HIP_TRY(hipFree(c->res_flags)); c->res_flags = nullptr; }
        TRY(dev_alloc(&c->res_flags, (size_t)tiles * WGR_FLAG_STRIDE));
        HIP_TRY(hipMemsetAsync(c->res_flags, 0, sizeof(unsigned) * tiles * WGR_FLAG_STRIDE, c->stream));
        c->res_flags_n = (size_t)tiles;
        c->res_epoch = 0;
    }
    if (!c->res_abort) {
        TRY(dev_alloc(&c->res_abort, 1));
        HIP_TRY(hipMemsetAsync(c->res_abort, 0, sizeof(unsigned), c->stream));
    }
    return DEFF_OK;
}

// Can this plan run resident at all?  A whole context (a slab's halo rows change between passes from outside), not a band
// of one, 32-bit buffer offsets, and the caller has not asked for one launch per pass.
static bool resident_allowed(const deff_ctx *c, const SweepPlan *pl)
{
    return c->tb_resident && !c->slab && pl->band_h <= 0 && !pl->T_override && c->n * sizeof(double) < ((size_t)1 << 31);
}

// Tall resident tiles (16 waves x R rows, kernels_wgtile.hpp): the smallest R in WGL_ROWS whose tiles all fit the chip -- or
// whose tiles are whole images, which wait for nobody and may queue for the CUs in any number --, or 0.  T = 8 only; not when
// the caller shapes the 8-wave tiles (tb_R, tb_LY) or insists on them (tb_NW = 8).
static int choose_tall_R(const deff_ctx *c, const SweepPlan *pl, int T, int own_h, int *tall_R)
{
    *tall_R = 0;
    if (T != 8 || !resident_allowed(c, pl) || c->tb_NW == WGT_WAVES || c->tb_NW == WGS_WAVES) return DEFF_OK;
    if (c->tb_NW != WGL_WAVES && (c->tb_R != 0 || c->tb_LY != 0)) return DEFF_OK;
    for (int R : WGL_ROWS) {
        if (c->tb_NW == WGL_WAVES && wgl_has_R(c->tb_R) && R != c->tb_R) continue;
        const int row_tiles = wgl_row_tiles(own_h, R, T);
        const long tiles = (long)pl->ntx * row_tiles * c->nimg;
        int res = 0;
        TRY(wgl_resident_blocks(c, R, pl->fma, c->lut_guard, &res));
        const bool whole_images = pl->ntx == 1 && row_tiles == 1;
        if (((tiles + 7) / 8) * 8 <= res || whole_images) { *tall_R = R; break; }
    }
    return DEFF_OK;
}

// 8-wave tiles (matrix rows in registers): rows per wave, rows per tile, grid; resident when all tiles fit the chip.
static int plan_tiles8(deff_ctx *c, SweepPlan *pl, int T, int own_h)
{
    pl->impl = 2;
    pl->NW = WGT_WAVES;
    pl->guard = c->lut_guard;
    int resident = c->tb_wg;
    if (c->tb_R == 4 || c->tb_R == 6 || c->tb_R == 7) {
        pl->R = c->tb_R;
    } else {
        // rows per wave: the fewest (shortest sweeps) whose tiles are all resident at once; if none is, 6
        // (7 needs 256 VGPRs and a few spilled registers: fine for one round, slower over several)
        pl->R = 6;
        for (int R : {4, 6, 7}) {
            const int lymax = wgt_rows_owned(T, R);
            if (lymax < 1) continue;
            int res = resident;
            if (!res) TRY(wgt_resident_blocks(c, T, R, pl->fma, c->lut_guard, &res));
            const long tiles = (long)pl->ntx * ((own_h + lymax - 1) / lymax) * c->nimg;
            if (tiles <= res) { pl->R = R; break; }
        }
    }
    // rows a tile owns: at most 8R - 2T; spread the image's rows evenly over its row tiles
    const int lymax = wgt_rows_owned(T, pl->R);
    int cpi = (own_h + lymax - 1) / lymax;
    if (c->tb_LY > 0 && c->tb_LY < lymax) cpi = (own_h + c->tb_LY - 1) / c->tb_LY;
    pl->LY = (own_h + cpi - 1) / cpi;
    pl->tcpi = (own_h + pl->LY - 1) / pl->LY;
    pl->tgy = pl->tcpi * c->nimg;
    const long tiles = (long)pl->ntx * pl->tgy;
    if (!resident) TRY(wgt_resident_blocks(c, T, pl->R, pl->fma, c->lut_guard, &resident));
    pl->tgx = (int)tiles;
    pl->tblocks = (int)(((tiles + 7) / 8) * 8);
    if (pl->tblocks > resident) pl->tblocks = resident >= 8 ? resident / 8 * 8 : 8;
    // Resident passes (k_sweep_wgres): every tile on the chip at once, tiles at least T rows tall (a tile's halo must end
    // inside its immediate neighbours: they are the ones it waits for)
    pl->resident = false;
    if (resident_allowed(c, pl) && (pl->LY >= T || pl->tcpi == 1)) {
        int res = 0;
        TRY(wgr_resident_blocks(c, T, pl->R, pl->fma, c->lut_guard, &res));
        if ((long)((tiles + 7) / 8) * 8 <= res) {
            pl->resident = true;
            pl->tblocks = (int)(((tiles + 7) / 8) * 8);
            TRY(ensure_resident_buffers(c, tiles));
        }
    }
    return DEFF_OK;
}

// Tall tiles with R rows per wave: always resident; the 7-lookup short-cut when the system is verified link-symmetric.
static int plan_tall(deff_ctx *c, SweepPlan *pl, int T, int own_h, int R)
{
    pl->impl = 2;
    pl->NW = WGL_WAVES;
    pl->R = R;
    pl->guard = c->lut_guard;
    const int cpi = wgl_row_tiles(own_h, R, T);
    pl->LY = (own_h + cpi - 1) / cpi;
    pl->tcpi = (own_h + pl->LY - 1) / pl->LY;
    pl->tgy = pl->tcpi * c->nimg;
    const long tiles = (long)pl->ntx * pl->tgy;
    pl->tgx = (int)tiles;
    pl->tblocks = (int)(((tiles + 7) / 8) * 8);
    pl->resident = true;
    TRY(ensure_resident_buffers(c, tiles));
    if (c->tb_sym != 2) TRY(check_links_symmetric(c));            // once per (codes, dictionary): one pass over the codes
    pl->sym = c->tb_sym != 2 && c->links_sym == 1;
    pl->aged = c->tb_tall_deal && wgage_has(R) && !pl->guard;     // rows dealt by age: k_sweep_wgage (unguarded systems, symmetric or not)
    return DEFF_OK;
}

// Link-symmetric 12-wave tiles (k_sweep_wgsym): matrix rows in registers at 3 waves per SIMD.  A tile of 12 x R rows has the
// shape of an 8-wave tile of 1.5 R rows and sweeps it faster (three waves of a SIMD issue FP64 every ~5 clocks, two every ~6),
// so wherever the system is verified link-symmetric and unguarded this form replaces the 8-wave tiles: the fewest rows per
// wave whose tiles all fit the chip.  *R = 0: not applicable (not symmetric, guarded, too many tiles, caller insists on
// another form).  T = 8, resident launches only.
// Shapes of a 12-wave tile: the rows of the waves of age 0 / 1 / 2 (wave >> 2: a SIMD serves its three waves oldest first, and
// a tile's waves meet at a barrier in every sweep -- see k_sweep_wgage).  Equal rows (k_sweep_wgsym<T, R>) for every T; for T = 8
// also the shapes that give the younger waves a row less (k_sweep_wgsage): 5 / 5 / 4 is a 56-row tile that sweeps ~9 % faster
// than 5 / 5 / 5 and owns 40 rows instead of 44 -- one 1024^2 image: 234 tiles instead of 216, 853 -> 901 G; 4 / 4 / 3 and 5 / 4 / 4
// likewise (704^2 ... 992^2: +6 ... 11 %, profiles/r04_sym_shapes_kbench.log).  The planner takes the
// first shape of this list (fewest rows first) whose tiles all fit the chip.  Encoded for the callers as R | a << 8 | b << 16 |
// c << 24 (R = the most rows a wave holds; a = 0: equal rows).
struct SymShape { int a, b, c; bool aged; };
static const SymShape SYM_SHAPES_T8[] = {{4, 4, 3, true}, {4, 4, 4, false}, {5, 4, 4, true}, {5, 5, 4, true}, {5, 5, 5, false}};
static const SymShape SYM_SHAPES[] = {{4, 4, 4, false}, {5, 5, 5, false}};
static int sym_shape_rows(int enc) { return (enc >> 8) ? 4 * (((enc >> 8) & 0xFF) + ((enc >> 16) & 0xFF) + ((enc >> 24) & 0xFF)) : WGS_WAVES * (enc & 0xFF); }

static int choose_sym_R(deff_ctx *c, const SweepPlan *pl, int T, int own_h, int *sym_R)
{
    *sym_R = 0;
    if ((T != 8 && T != 6 && T != 4) || !resident_allowed(c, pl) || c->lut_guard || c->tb_sym == 2) return DEFF_OK;
    if (c->tb_NW != 0 && c->tb_NW != WGS_WAVES) return DEFF_OK;
    if (T != 8 && c->tb_T && c->tb_NW != WGS_WAVES) return DEFF_OK;  // a caller's T = 4 / 6 means these tiles only together with tb_NW = 12
    if (c->tb_NW != WGS_WAVES && (c->tb_R != 0 || c->tb_LY != 0)) return DEFF_OK;
    int found = 0;
    const bool t8 = T == 8 && c->tb_sym_age;
    const SymShape *shapes = t8 ? SYM_SHAPES_T8 : SYM_SHAPES;
    const int nshapes = t8 ? (int)(sizeof SYM_SHAPES_T8 / sizeof SYM_SHAPES_T8[0]) : (int)(sizeof SYM_SHAPES / sizeof SYM_SHAPES[0]);
    for (int k = 0; k < nshapes; ++k) {
        const SymShape &sh = shapes[k];
        const int R = sh.a;
        if (c->tb_NW == WGS_WAVES && wgs_has_R(c->tb_R) && (R != c->tb_R || sh.aged)) continue;   // a caller's R: equal rows of that many
        if (t8 && c->tb_sym_shape && k + 1 != c->tb_sym_shape) continue;                          // tests: this shape of SYM_SHAPES_T8 or none
        const int enc = sh.aged ? (R | sh.a << 8 | sh.b << 16 | sh.c << 24) : R;
        const int lymax = sym_shape_rows(enc) - 2 * T;
        const int cpi = (own_h + lymax - 1) / lymax;
        const int LY = (own_h + cpi - 1) / cpi;
        if (LY < T && cpi > 1) continue;                            // a tile's halo must end inside its immediate neighbours
        const long tiles = (long)pl->ntx * cpi * c->nimg;
        int res = 0;
        TRY(wgs_resident_blocks(c, T, R, pl->fma, &res));
        if (((tiles + 7) / 8) * 8 <= res) { found = enc; break; }
    }
    if (!found) return DEFF_OK;
    TRY(check_links_symmetric(c));                                  // once per (codes, dictionary): one pass over the codes
    if (c->links_sym == 1) *sym_R = found;
    return DEFF_OK;
}

static int plan_sym(deff_ctx *c, SweepPlan *pl, int T, int own_h, int enc)
{
    pl->impl = 2;
    pl->NW = WGS_WAVES;
    pl->R = enc & 0xFF;
    pl->rows3 = enc >> 8;
    pl->guard = false;
    const int lymax = sym_shape_rows(enc) - 2 * T;
    const int cpi = (own_h + lymax - 1) / lymax;
    pl->LY = (own_h + cpi - 1) / cpi;
    pl->tcpi = (own_h + pl->LY - 1) / pl->LY;
    pl->tgy = pl->tcpi * c->nimg;
    const long tiles = (long)pl->ntx * pl->tgy;
    pl->tgx = (int)tiles;
    pl->tblocks = (int)(((tiles + 7) / 8) * 8);
    pl->resident = true;
    pl->sym = true;
    TRY(ensure_resident_buffers(c, tiles));
    return DEFF_OK;
}

// Streaming form: rows per chunk.  Workgroups are persistent, so a pass takes `rounds` tiles per wave slot (one round = as
// many wave tiles as are resident at once), and a tile costs its LY rows + T steps that drain the pipeline + T rows of halo
// above it unless it starts at the top wall of its image + a fixed start-up (first loads, measured ~8 row steps).  Pick the
// chunks per image minimising rounds x tile cost; for k rounds only the largest chunk count that fits matters.  (Stacks of
// small images: 3 072 x 128^2 as whole-image tiles 1 266 G cells*iter/s against 1 107 G for the 4 x 32-row tiles a
// halo-blind model picks.)
// Chunk heights by service order.  A SIMD serves the waves it holds oldest first (tools/tb_stamps.py: with equal chunks the
// three waves of a SIMD end at 71 / 89 / 108 us of a 4096^2 pass -- the SIMD runs on two waves, then on one, for a third of the
// launch), and which wave is the oldest is known beforehand: workgroups go to the XCDs round-robin and fill an XCD's CUs
// once around before any CU gets its second one (observed on every SIMD of the chip: workgroup (blockIdx >> 3) / 32 of an XCD
// = wave slot 0, 1, 2).  So the chunks need not be equal: the oldest rank gets the tallest, the youngest the shortest, in
// proportion to the speeds the ranks run at (tb_rank_w), and all three end together.  Each strip is cut into nq chunks
// per rank -- the oldest rank's at the top, the youngest's at the bottom --, a workgroup's four waves hold four stacked
// chunks of one rank, and an XCD's workgroups hold neighbouring strips (its L2 sees the shared halo rows).  The result
// is written as a table the kernel reads (k_sweep_matfree_tb, `dealt`); every row is still covered once, so the bits
// cannot change -- if the dispatch order is ever different (another process on the GPU), only the balance is lost.
static int deal_ranked_tiles(deff_ctx *c, SweepPlan *pl, int T, int own_lo, int own_h, int resident, bool *dealt)
{
    *dealt = false;
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    const int xcds = 8, per_xcd = cus / xcds, occ = cus > 0 ? resident / cus : 0;
    if (cus % xcds != 0 || occ != 3 || resident != occ * cus) return DEFF_OK;       // three waves per SIMD is what was measured
    const int slots = cus * 4;                                                       // waves per rank
    const int ntx = pl->ntx, cap = own_h / (3 * T);
    const int ncol = ntx * c->nimg;                      // columns to cut into chunks: every strip of every image of a stack
    if (ntx > 0xFFFF || c->nimg > 0x7FFF || ncol > slots) return DEFF_OK;
    const int nq = std::min(slots / ncol, cap);
    if (nq < 1 || c->tb_rank_wall < 1000 || c->tb_rank_w[0] < 1 || c->tb_rank_w[1] < 1 || c->tb_rank_w[2] < 1) return DEFF_OK;
    const std::vector<int> key = {T, own_lo, own_h, ntx, c->nimg, pl->shift, resident, c->tb_rank_w[0], c->tb_rank_w[1], c->tb_rank_w[2], c->tb_rank_wall};
    auto fill_plan = [&]() {
        pl->dealt = c->tb_dealt;
        pl->LY = c->tb_dealt_LY;
        pl->tcpi = 3 * c->tb_dealt_nmax;                                             // (the stamps' numbering: rank, image, strip, chunk < nmax)
        pl->tgy = pl->tcpi * c->nimg;
        pl->tgx = (ntx * pl->tgy + 3) / 4;
        pl->tblocks = resident;
        *dealt = true;
    };
    if (c->tb_dealt && key == c->tb_dealt_key) { fill_plan(); return DEFF_OK; }
    const double v[3] = {(double)c->tb_rank_w[0], (double)c->tb_rank_w[1], (double)c->tb_rank_w[2]};
    const int K = T * (T - 1) + 8;                        // level steps a chunk costs on top of T per row: halo triangles + fill
    // A wall strip's waves look up b as well and run 10-15 % longer per row (mean end of a 4096^2 pass by strip, equal chunk
    // counts: 107 us in the first strip against 94...100 in the others): its rows count tb_rank_wall per mille.  A last strip
    // that is partly outside the mesh (4096 columns: half of it) moves fewer cache lines: the surcharge in proportion.
    // (the surcharge differs by rank: the oldest wave runs at the pace of its own dependency chain and pays every extra lookup --
    // its wall chunks ended 12-14 % after the rank's mean --, the youngest waits for issue slots anyway: 6 %)
    std::vector<std::array<double, 3>> speed(ncol, std::array<double, 3>{v[0], v[1], v[2]});
    const bool walls = ntx >= 3;
    if (walls) {
        const int hw = (T + 1) & ~1, wout = TB_COLS - 2 * hw;
        const int last_cols = c->nx - ((ntx - 1) * wout - pl->shift);                // columns of the last strip inside the mesh
        const double extra = c->tb_rank_wall / 1000.0 - 1.0, by_rank[3] = {2.5, 1.5, 1.0};
        const double fill = (double)std::min(last_cols, TB_COLS) / TB_COLS;
        for (int img = 0; img < c->nimg; ++img)
            for (int r = 0; r < 3; ++r) {
                speed[(size_t)img * ntx][r] = v[r] / (1.0 + extra * by_rank[r]);
                speed[(size_t)img * ntx + ntx - 1][r] = v[r] / (1.0 + extra * by_rank[r] * fill);
            }
    }
    // chunks per strip and rank: nq each, then the wave slots left over go one by one to the strip that would end last
    // (time of a strip = its level steps over the speed of its waves: (T own_h + K sum n_r) / sum n_r v_r)
    struct Strip { int n[3]; int ly[3]; };
    std::vector<Strip> st(ncol);
    for (auto &q : st) q.n[0] = q.n[1] = q.n[2] = nq;
    int spare[3] = {slots - ncol * nq, slots - ncol * nq, slots - ncol * nq};
    auto strip_time = [&](int tx) {
        const Strip &q = st[tx];
        const std::array<double, 3> &u = speed[tx];
        return ((double)T * own_h + (double)K * (q.n[0] + q.n[1] + q.n[2])) / (q.n[0] * u[0] + q.n[1] * u[1] + q.n[2] * u[2]);
    };
    for (int it = 0; it < 3 * slots; ++it) {
        int worst = -1;
        double tw = 0;
        for (int tx = 0; tx < ncol; ++tx) {
            const double t = strip_time(tx);
            if (t > tw) { tw = t; worst = tx; }
        }
        int r = -1;
        for (int k = 0; k < 3; ++k)
            if (spare[k] > 0 && st[worst].n[k] < cap && (r < 0 || spare[k] > spare[r])) r = k;
        if (r < 0) break;
        ++st[worst].n[r];
        --spare[r];
    }
    // chunk heights: a rank's chunk gets the rows its waves finish in the strip's time; the oldest rank takes the rounding
    int nmax = 0;
    for (int tx = 0; tx < ncol; ++tx) {
        Strip &q = st[tx];
        const double t = strip_time(tx);
        int left = own_h;
        for (int r = 2; r >= 1; --r) {
            int ly = (int)((t * speed[tx][r] - K) / T);
            if (ly < T) ly = T;
            q.ly[r] = ly;
            left -= q.n[r] * ly;
        }
        q.ly[0] = (left + q.n[0] - 1) / q.n[0];
        if (q.ly[0] < T) return DEFF_OK;
        nmax = std::max(nmax, std::max(q.n[0], std::max(q.n[1], q.n[2])));
    }
    if ((long)3 * c->nimg * ntx * nmax >= (1L << 30)) return DEFF_OK;
    const size_t entries = (size_t)resident * 4 + 1;              // + the word the waves count their misplacements in (kernels_tb.hpp)
    {
        std::vector<int4> &tab = c->tb_dealt_host;
        tab.assign(entries, make_int4(0, 0, 0, 0));
        auto tile = [&](int r, int col, int q) {
            const Strip &sp = st[col];
            const int img = col / ntx, tx = col % ntx;
            const int own0 = own_lo + img * c->ny;                                  // (c->ny: the row pitch of a stack's images)
            const int own_hi = own0 + own_h;
            int ry0 = own0;
            for (int k = 0; k < r; ++k) ry0 += sp.n[k] * sp.ly[k];
            ry0 += q * sp.ly[r];
            int rows_here = std::min(sp.ly[r], own_hi - ry0);
            if (r == 2 && q == sp.n[2] - 1) rows_here = own_hi - ry0;              // the youngest rank's last chunk takes what rounding left over
            return make_int4(tx | (img << 16), ry0, rows_here > 0 ? rows_here : 0, (int)((unsigned)(((r * c->nimg + img) * ntx + tx) * nmax + q) | ((unsigned)r << 30)));
        };
        for (int r = 0; r < 3; ++r) {
            // workgroup m of rank r: XCD m / per_xcd, the (m % per_xcd)-th of that XCD's workgroups of this rank
            auto slot = [&](int m, int w) { return ((((size_t)(r * per_xcd + m % per_xcd) << 3) | (size_t)(m / per_xcd)) * 4 + w); };
            std::vector<char> used((size_t)cus * 4, 0);
            // the wall strips' chunks first, one per workgroup and spread over the chip (twelve waves looking up b on one CU
            // were the last to end by 5 us), each rank starting elsewhere
            std::vector<int4> wall_tiles;
            if (walls)
                for (int img = 0; img < c->nimg; ++img)
                    for (int tx : {0, ntx - 1})
                        for (int q = 0; q < st[(size_t)img * ntx + tx].n[r]; ++q) wall_tiles.push_back(tile(r, img * ntx + tx, q));
            const int nw = (int)wall_tiles.size();
            for (int k = 0; k < nw; ++k) {
                int m = (int)(((long)k * cus) / std::max(nw, 1) + (long)r * cus / 3) % cus, w = 0;
                while (used[(size_t)m * 4 + w]) { if (++w == 4) { w = 0; m = (m + 1) % cus; } }
                used[(size_t)m * 4 + w] = 1;
                tab[slot(m, w)] = wall_tiles[k];
            }
            // the inner strips in order (chunk index fastest): a workgroup's waves hold stacked chunks, an XCD neighbouring strips
            int m = 0, w = 0;
            for (int col = 0; col < ncol; ++col) {
                if (walls && (col % ntx == 0 || col % ntx == ntx - 1)) continue;
                for (int q = 0; q < st[col].n[r]; ++q) {
                    while (m < cus && used[(size_t)m * 4 + w]) { if (++w == 4) { w = 0; ++m; } }
                    if (m >= cus) return fail(DEFF_ESTATE, "dealt tiles: more chunks than waves (rank %d)", r);
                    used[(size_t)m * 4 + w] = 1;
                    tab[slot(m, w)] = tile(r, col, q);
                }
            }
        }
        if (c->tb_dealt_cap < entries) {
            TRY(resident_check(c));
            if (c->tb_dealt) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->tb_dealt)); c->tb_dealt = nullptr; }
            HIP_TRY(hipMalloc((void **)&c->tb_dealt, entries * sizeof(int4)));
            c->tb_dealt_cap = entries;
        }
        HIP_TRY(hipMemcpyAsync(c->tb_dealt, tab.data(), entries * sizeof(int4), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));                                   // (pageable source; once per plan change)
        c->tb_dealt_key = key;
        c->tb_dealt_LY = st[ntx / 2].ly[1];
        c->tb_dealt_nmax = nmax;
        c->tb_dealt_miss_at = (size_t)resident * 4;
        c->tb_dealt_waves = 0;
        c->tb_dealt_looks = 0;
    }
    fill_plan();
    return DEFF_OK;
}

static int plan_streaming(deff_ctx *c, SweepPlan *pl, int T, int own_lo, int own_h)
{
    pl->impl = 1;
    pl->resident = false;
    pl->guard = c->lut_guard;                          // the reference's non-zero link test matters only when a phase cannot diffuse
    int resident = c->tb_wg;
    if (!resident) {
        TRY(tb_resident_blocks(c, T, pl->fma, c->lut_guard, &resident));
    }
    pl->dealt = nullptr;
    // (T = 8 only: the ranks' speeds were measured there; with them T = 6 gains 3 % at 4096^2 and loses 4 % at 8192^2)
    if (c->tb_ranked && !c->tb_rank_lost && !c->tb_LY && !c->tb_wg && pl->band_h == 0 && !c->slab && T == 8 && !c->masked) {
        bool dealt = false;
        TRY(deal_ranked_tiles(c, pl, T, own_lo, own_h, resident, &dealt));
        if (dealt) return DEFF_OK;
    }
    int LY = c->tb_LY;
    if (!LY) {
        long best_cost = -1;
        const bool top_wall = own_lo == 0;              // not a slab with rows above it
        for (int k = 1; k <= 8; ++k) {
            const int cpi_max = (int)(((long)k * resident * 4) / ((long)pl->ntx * c->nimg));
            if (cpi_max < 1) continue;
            int ly = (own_h + cpi_max - 1) / cpi_max;
            // chunks shorter than the pipeline is deep lose more to fill/drain than the model says (1024^2, T=4: 3-row
            // chunks 254 G, 4..6-row chunks 295 G cells*iter/s)
            if (ly < T) ly = T;
            const int cpi = (own_h + ly - 1) / ly;
            const long cost = (long)k * (ly + T + ((cpi > 1 || !top_wall) ? T : 0) + 8);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; LY = ly; }
        }
        if (!LY) LY = own_h;
    }
    if (LY > own_h) LY = own_h;
    pl->LY = LY;
    pl->tcpi = (own_h + LY - 1) / LY;
    pl->tgy = pl->tcpi * c->nimg;
    pl->tgx = (int)(((long)pl->ntx * pl->tgy + 3) / 4);           // workgroup tiles (4 wave tiles each)
    const unsigned total = (unsigned)pl->tgx;
    pl->tblocks = (int)(((total + 7u) / 8u) * 8u);
    if (pl->tblocks > resident) pl->tblocks = resident >= 8 ? resident / 8 * 8 : 8;
    return DEFF_OK;
}

// what deff_get_plan() reports: the plan of whole passes of the context (not a slab's T = 1 remainder plan, not a band)
static void record_plan(deff_ctx *c, const SweepPlan *pl)
{
    if (pl->band_h > 0 || (pl->impl != 2 && pl->T_override)) return;
    c->plan_T = pl->T; c->plan_LY = pl->LY; c->plan_ntx = pl->ntx; c->plan_cpi = pl->tcpi;
    c->plan_blocks = pl->tblocks; c->plan_impl = pl->impl;
    c->plan_R = pl->impl == 2 ? pl->R : 0;
    c->plan_NW = pl->impl == 2 ? pl->NW : 0;
    c->plan_resident = pl->impl == 2 && pl->resident ? 1 : 0;
    c->plan_ranked = pl->impl == 1 && pl->dealt ? 1 : 0;
    c->plan_aged = pl->impl == 2 && ((pl->NW == WGL_WAVES && pl->aged) || (pl->NW == WGS_WAVES && pl->rows3)) ? 1 : 0;
}

// The form of a blocked pass, in this order ("What the planner picks"): 8-wave tiles when they are all
// resident; tall tiles when those are; else 8-wave tiles with one launch per pass below 4 Mi cells and streaming above.
static int plan_blocked_pass(deff_ctx *c, SweepPlan *pl)
{
    // sweeps per pass (measured, G cells*iter/s: 4096^2 T=4 926, T=6 1063, T=8 1106; stacks of 16 x 1024^2 peak at T=6;
    // 1024^2 alone at T=4)
    const int T = clamp_tb_T(pl->T_override ? pl->T_override : (c->tb_T ? c->tb_T : default_tb_T(c)));
    pl->T = T;
    // rows this plan updates: the context's owned rows, or a band of them (row slabs split a pass into the bands the
    // neighbours wait for and the interior, api_slab.hip)
    const int own_lo = pl->band_h > 0 ? pl->band_lo : c->own_lo;
    const int own_h = pl->band_h > 0 ? pl->band_h : c->own_h;
    pl->own_lo = own_lo;
    pl->own_h = own_h;
    plan_strips(c, T, pl);
    int tall_R = 0, sym_R = 0;
    TRY(choose_tall_R(c, pl, T, own_h, &tall_R));
    int want_impl = c->tb_impl ? c->tb_impl : default_tb_impl(c);
    // a context just above the 4 Mi cells where the streaming form takes over still runs faster on tall tiles when they fit
    if (!c->tb_impl && want_impl == 1 && tall_R) want_impl = 2;
    // workgroup tiles exist for T = 4 and 8 (the 12-wave link-symmetric form also for T = 6, on request); slabs' T = 1 remainder
    // passes and the other T stay on the streaming kernel
    if (want_impl == 2 && (T == 4 || T == 8 || (T == 6 && c->tb_NW == WGS_WAVES)) && !pl->T_override) {
        const bool sym_only = c->tb_NW == WGS_WAVES && (T == 6 || T == 4) && c->tb_T;   // the caller asked for 12-wave tiles at this T
        // images that are ONE tall tile each (a stack of 128^2 images) recompute nothing and wait for nobody: nothing beats that
        const bool tall_whole = tall_R && pl->ntx == 1 && wgl_row_tiles(own_h, tall_R, T) == 1;
        if (!tall_whole) TRY(choose_sym_R(c, pl, T, own_h, &sym_R));
        bool planned = false;
        if (tall_whole && c->tb_NW != WGT_WAVES) {
            TRY(plan_tall(c, pl, T, own_h, tall_R));
            planned = true;
        } else if (sym_R) {
            // Both coefficient-resident forms may fit the chip: a sweep costs a SIMD its share of the tile's rows times the
            // clocks a row takes at that occupancy -- measured (tools/wgr_stamps.py) ~160 at two waves per SIMD, ~142 at
            // three.  512^2 / 640^2 stay on 8 waves x 4 rows (303-476 G against 310-481 G on 12 x 3, which is therefore
            // not instantiated), 768^2 ... 1100^2 go to 12 waves (559 against 511 G, 746 against 688 G, 850 against 808 G).
            bool take_sym = true;
            if (c->tb_NW != WGS_WAVES) {
                SweepPlan alt = *pl;
                TRY(plan_tiles8(c, &alt, T, own_h));
                if (alt.resident && 2 * alt.R * 160 <= 3 * (sym_R & 0xFF) * 142) { *pl = alt; take_sym = false; }
            }
            if (take_sym) TRY(plan_sym(c, pl, T, own_h, sym_R));
            planned = true;
        }
        if (!planned && sym_only) return fail(DEFF_EINVAL, "tb_T = %d on 12-wave tiles: the tiles are not co-resident or the system is not link-symmetric", T);
        if (!planned) {
            TRY(plan_tiles8(c, pl, T, own_h));
            // Images a little too large for the 12-wave tiles at T = 8 (1101 ... 1172 columns: 1152^2 is 297 tiles of 44 x 112
            // owned cells) fit with passes of SIX sweeps -- 48 x 116 owned cells per tile, 240 tiles at 1152^2 -- and a sweep
            // then costs (6 x 5 rows x 3 waves x ~142 clocks + the exchange) / 6 = ~3 300 clocks against ~4 000 ... 4 800 on
            // tall tiles (lookups in every sweep, 4 waves per SIMD): taken whenever it fits and the caller has fixed neither T
            // nor the form.
            // Passes of FOUR (52 x 120 owned cells: up to 1208 columns x 1248 rows) come after that: ~4 200 clocks per sweep,
            // still ahead of the tall tiles' ~4 900 where those need R = 5.
            bool shorter = false;
            if (!pl->resident && T == 8 && !c->tb_T && (c->tb_NW == 0 || c->tb_NW == WGS_WAVES) && !c->tb_R && !c->tb_LY) {
                for (int Ts : {6, 4}) {
                    SweepPlan alt = *pl;
                    alt.T = Ts;
                    plan_strips(c, Ts, &alt);
                    int rs = 0;
                    TRY(choose_sym_R(c, &alt, Ts, own_h, &rs));
                    if (Ts == 4 && rs && (rs & 0xFF) < 5 && tall_R && tall_R <= 4) rs = 0;      // (4 x 4 rows per sweep: no better than tall R = 4)
                    if (rs) { TRY(plan_sym(c, &alt, Ts, own_h, rs)); *pl = alt; shorter = true; break; }
                }
            }
            if (!shorter && tall_R && (!pl->resident || c->tb_NW == WGL_WAVES)) TRY(plan_tall(c, pl, T, own_h, tall_R));
        }
    } else {
        TRY(plan_streaming(c, pl, T, own_lo, own_h));
    }
    record_plan(c, pl);
    return DEFF_OK;
}

int plan_sweeps(deff_ctx *c, double omega, SweepPlan *pl)
{
    if (!c->have_field) return fail(DEFF_ESTATE, "no field: call deff_init_linear() or deff_set_field()");
    // an explicit system (host-assembled, 3-phase, ImpSolid) with few distinct rows also runs matrix-free
    if (!c->have_matfree && c->have_explicit && !c->dict_tried && c->dict_enabled && !c->wrap_links &&
        (c->kernel == DEFF_KERNEL_AUTO || c->kernel == DEFF_KERNEL_MATFREE || c->kernel == DEFF_KERNEL_MATFREE_TB))
        TRY(try_dict(c));
    TRY(resolve_kernel(c, &pl->kernel));
    pl->fma = c->fma != 0;
    pl->omega = omega;
    pl->omw = 1.0 - omega;                              // cuh:89 evaluates (1.0 - w) in double
    if (pl->kernel == DEFF_KERNEL_MATFREE || pl->kernel == DEFF_KERNEL_MATFREE_TB) {
        TRY(upload_lut(c, omega));
        if (pl->kernel == DEFF_KERNEL_MATFREE_TB) TRY(plan_blocked_pass(c, pl));
        // single sweeps (the first sweep and the n mod T remainder): 4 rows per tile and up to 8 192 workgroups (measured at
        // 4096^2: 52.0 us = 5.8 TB/s against 59-61 us for 8 rows x 2 048 persistent workgroups; 16384^2: 960-990 us = 4.9-5.0
        // TB/s either way -- above what a plain copy kernel gets from HBM for this read / write mix, tools/ubench mem: 4.7 TB/s)
        tile_grid(c, 256 * 2, pick_R(c->rows_matfree, c->n >= ((size_t)1 << 21) ? 4 : 2), pl);
        // persistent grid: workgroups walk the tiles (tables loaded once each)
        const int cap = c->wg_matfree ? c->wg_matfree : 256 * 32;
        if (pl->blocks > cap) pl->blocks = cap;
    } else {
        TRY(explicit_from_image(c));
        if (c->c0_omega != omega) {
            hipLaunchKernelGGL(k_make_c0, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->a0, omega, c->c0,
                               c->n);
            HIP_TRY(hipGetLastError());
            c->c0_omega = omega;
        }
        if (pl->kernel == DEFF_KERNEL_EXPLICIT)
            tile_grid(c, 512, pick_R(c->rows_explicit, 1), pl);
    }
    return DEFF_OK;
}

// Enqueue one sweep x[cur] -> x[cur^1] and flip (the reference copies instead, cuh:1281).
void enqueue_sweep(deff_ctx *c, const SweepPlan &pl)
{
    const double *xin = c->x[c->cur];
    double *xout = c->x[c->cur ^ 1];
    const CoefConst cf{c->c0, c->aW, c->aE, c->aS, c->aN, c->b};
    const int flip = c->serpentine ? c->cur : 0;
    const uint8_t *mask = c->masked ? c->active : nullptr;
    switch (pl.kernel) {
    case DEFF_KERNEL_SCALAR: {
#define LAUNCH_SCALAR(NT_, F_)                                                                              \
    hipLaunchKernelGGL((k_sweep_scalar<NT_, F_>), dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0,       \
                       c->stream, cf, xin, xout, c->nx, c->n, c->n_img, mask, pl.omw)
        if (c->nt_explicit) { if (pl.fma) LAUNCH_SCALAR(true, true); else LAUNCH_SCALAR(true, false); }
        else { if (pl.fma) LAUNCH_SCALAR(false, true); else LAUNCH_SCALAR(false, false); }
#undef LAUNCH_SCALAR
        break;
    }
    case DEFF_KERNEL_EXPLICIT: {
#define LAUNCH_EXPLICIT_(R_, NT_, F_)                                                                        \
    hipLaunchKernelGGL((k_sweep_explicit<R_, NT_, F_>), dim3(pl.blocks), dim3(256), 0, c->stream, cf, xin,   \
                       xout, c->nx, c->ny, c->rows, pl.cpi, mask, pl.gx, pl.gy, flip, pl.omw)
#define LAUNCH_EXPLICIT(R_)                                                                                  \
    do {                                                                                                    \
        if (c->nt_explicit) { if (pl.fma) LAUNCH_EXPLICIT_(R_, true, true); else LAUNCH_EXPLICIT_(R_, true, false); } \
        else { if (pl.fma) LAUNCH_EXPLICIT_(R_, false, true); else LAUNCH_EXPLICIT_(R_, false, false); }   \
    } while (0)
        switch (pl.rows) {
        case 1: LAUNCH_EXPLICIT(1); break;
        case 2: LAUNCH_EXPLICIT(2); break;
        case 4: LAUNCH_EXPLICIT(4); break;
        default: LAUNCH_EXPLICIT(8); break;
        }
#undef LAUNCH_EXPLICIT
#undef LAUNCH_EXPLICIT_
        break;
    }
    default: {
#define LAUNCH_MATFREE_(V_, R_, F_)                                                                          \
    hipLaunchKernelGGL((k_sweep_matfree<V_, R_, F_>), dim3(pl.blocks), dim3(256), 0, c->stream, c->lut,      \
                       c->code, xin, xout, c->nx, c->ny, c->rows, pl.cpi, mask, pl.gx, pl.gy, flip,          \
                       c->lut_nrows, pl.omw)
#define LAUNCH_MATFREE(V_, R_)                                                                               \
    do { if (pl.fma) LAUNCH_MATFREE_(V_, R_, true); else LAUNCH_MATFREE_(V_, R_, false); } while (0)
        switch (pl.rows) {
        case 1: LAUNCH_MATFREE(2, 1); break;
        case 2: LAUNCH_MATFREE(2, 2); break;
        case 4: LAUNCH_MATFREE(2, 4); break;
        default: LAUNCH_MATFREE(2, 8); break;
        }
#undef LAUNCH_MATFREE
#undef LAUNCH_MATFREE_
        break;
    }
    }
    c->cur ^= 1;
}

// Is the chip dispatching the way the dealt tiles assume?  Called where the stream has just been synchronised, for the first
// three such points after a table was built: the waves that found themselves in another slot than their tile was cut for have
// counted themselves (kernels_tb.hpp).  More than a quarter of them misplaced -- somebody else's kernels on the GPU, another
// dispatch order -- and the context goes back to equal chunks; the results are the same bits either way.
int dealt_watch(deff_ctx *c)
{
    if (!c->tb_dealt || c->tb_dealt_looks >= 3 || c->tb_dealt_waves == 0) return DEFF_OK;
    unsigned miss = 0;
    HIP_TRY(hipMemcpyAsync(&miss, reinterpret_cast<const char *>(c->tb_dealt + c->tb_dealt_miss_at), sizeof miss, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ++c->tb_dealt_looks;
    c->tb_rank_misses = (int)std::min<int64_t>(miss, INT32_MAX);
    if ((int64_t)miss * 4 > c->tb_dealt_waves) c->tb_rank_lost = 1;
    return DEFF_OK;
}

// One temporally blocked pass: T sweeps, x[cur] -> x[cur^1].
int enqueue_tb_pass(deff_ctx *c, const SweepPlan &pl)
{
    TRY(launch_tb_pass(c, pl));
    c->cur ^= 1;
    return DEFF_OK;
}

// The launch of a pass (or of one band of it: pl.own_lo / pl.own_h) without the buffer flip.
int launch_tb_pass(deff_ctx *c, const SweepPlan &pl)
{
    const double *xin = c->x[c->cur];
    double *xout = c->x[c->cur ^ 1];
    const int flip = c->serpentine ? c->cur : 0;
    const uint8_t *mask = c->masked ? c->active : nullptr;
    if (pl.impl == 2 && pl.NW != WGT_WAVES)
        return fail(DEFF_ESTATE, "tiles of %d waves only exist as resident launches (plan again with tb_launch = 1)", pl.NW);
    if (pl.impl == 2) {
#define LAUNCH_WGT(T_, R_, C_, G_)                                                                             \
    hipLaunchKernelGGL((k_sweep_wgtile<T_, R_, C_, G_>), dim3(pl.tblocks), dim3(WGT_WAVES * 64), 0, c->stream, c->lut, \
                       c->code, xin, xout, c->nx, c->mesh_ny, c->ny, c->dom_lo, pl.own_lo, pl.own_h, pl.tcpi, pl.LY, \
                       mask, pl.ntx, pl.tgy, c->tb_xmajor, (c->lut_allb || c->nx != c->nxt) ? 1 : 0,         \
                       c->lut_nrows, pl.shift, pl.omw, c->tb_stamps)
        WGT_DISPATCH(pl.T, pl.R, pl.fma, pl.guard, LAUNCH_WGT);
#undef LAUNCH_WGT
        HIP_TRY(hipPeekAtLastError());
        return DEFF_OK;
    }
#define LAUNCH_TB(T_, C_, G_)                                                                                  \
    hipLaunchKernelGGL((k_sweep_matfree_tb<T_, C_, G_>), dim3(pl.tblocks), dim3(256), 0, c->stream, c->lut,    \
                       c->code, xin, xout, c->nx, c->mesh_ny, c->ny, c->dom_lo, pl.own_lo, pl.own_h, pl.tcpi, \
                       mask, pl.LY, pl.ntx, pl.tgx, pl.tgy, flip, c->tb_xmajor,                              \
                       (c->lut_allb || c->nx != c->nxt) ? 1 : 0, /* padded: the wall column may not be in the last strip */ \
                       c->lut_nrows, pl.shift, pl.omw, c->tb_stamps, pl.dealt)
    TB_DISPATCH(pl.T, pl.fma, pl.guard, LAUNCH_TB);
#undef LAUNCH_TB
    HIP_TRY(hipPeekAtLastError());
    if (pl.dealt) c->tb_dealt_waves += (int64_t)pl.tblocks * 4;
    return DEFF_OK;
}

// All whole passes of n sweeps as resident launches of up to 4 096 passes (tens of milliseconds each); *n is reduced by
// the sweeps enqueued.  In front of the first resident launch since the abort flag was last looked at, the field is copied
// aside: the restart point if a launch gives up (resident_check).
static int launch_resident_passes(deff_ctx *c, const SweepPlan &pl, int64_t *n)
{
    int64_t np = *n / pl.T;
    if (np > 0 && !c->res_pending) {
        TRY(dev_alloc(&c->res_backup, c->n));
        HIP_TRY(hipMemcpyAsync(c->res_backup, c->x[c->cur], sizeof(double) * c->n, hipMemcpyDeviceToDevice, c->stream));
        c->res_backup_cur = c->cur;
        c->res_redo = 0;
        c->res_omega = pl.omega;
    }
    // a resident launch holds the whole chip until it ends: keep one to ~25 ms (a pass of T sweeps takes about n * T / 0.9e12 s
    // on these forms), between 64 and 4 096 passes -- the reference's 10 000-sweep interval is one launch up to ~1500^2
    const double pass_s = (double)c->n * pl.T / 0.9e12;
    const int64_t cap = std::max<int64_t>(64, std::min<int64_t>(4096, (int64_t)(25e-3 / pass_s)));
    while (np > 0) {
        const int chunk = (int)(np < cap ? np : cap);
        if (c->res_epoch > (1u << 30)) {
            // the flags count passes since they were last cleared and are compared through a signed difference: start a
            // new count long before it could wrap (stream-ordered: every earlier launch has finished with them)
            HIP_TRY(hipMemsetAsync(c->res_flags, 0, sizeof(unsigned) * c->res_flags_n * WGR_FLAG_STRIDE, c->stream));
            c->res_epoch = 0;
        }
        hipError_t e = hipSuccess;
        if (pl.NW == WGS_WAVES && pl.rows3) {
            e = hipErrorInvalidConfiguration;                       // (stays if SYM_SHAPES_T8 names a shape without a kernel)
#define SAGE(A_, B_, C_)                                                                                                    \
            if (pl.T == 8 && pl.rows3 == (A_ | B_ << 8 | C_ << 16)) {                                                        \
                if (pl.fma) e = launch_resident(c, pl, k_sweep_wgsage<8, A_, B_, C_, true>, WGS_WAVES * 64, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch);  \
                else e = launch_resident(c, pl, k_sweep_wgsage<8, A_, B_, C_, false>, WGS_WAVES * 64, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch);        \
            }
            SAGE(4, 4, 3) SAGE(5, 4, 4) SAGE(5, 5, 4)
#undef SAGE
        } else if (pl.NW == WGS_WAVES) {
#define LAUNCH_WGS(T_, R_, C_) e = launch_wgsym<T_, R_, C_>(c, pl, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch)
            WGS_DISPATCH(pl.T, pl.R, pl.fma, LAUNCH_WGS);
#undef LAUNCH_WGS
        } else if (pl.NW == WGL_WAVES && pl.aged) {
            e = hipErrorInvalidConfiguration;                       // (stays if the table has no set for pl.R: plan_tall asked wgage_has)
#define X(R_, A_, B_, C_, D_)                                                                                               \
            if (pl.R == R_) {                                                                                                \
                if (pl.sym) {                                                                                                \
                    if (pl.fma) e = launch_wgage<A_, B_, C_, D_, true, false, true>(c, pl, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch);  \
                    else e = launch_wgage<A_, B_, C_, D_, false, false, true>(c, pl, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch);        \
                } else {                                                                                                     \
                    if (pl.fma) e = launch_wgage<A_, B_, C_, D_, true, false, false>(c, pl, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch); \
                    else e = launch_wgage<A_, B_, C_, D_, false, false, false>(c, pl, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch);       \
                }                                                                                                            \
            }
            WGAGE_SETS(X)
#undef X
        } else if (pl.NW == WGL_WAVES) {
            // (the symmetric short-cut exists in the unguarded kernels only: the guarded one branches on every link anyway)
#define LAUNCH_WGL(T_, R_, C_, G_) e = launch_wgres<T_, R_, C_, G_, true, false>(c, pl, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch)
#define LAUNCH_WGLS(T_, R_, C_, G_) e = launch_wgres<T_, R_, C_, false, true, true>(c, pl, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch)
            if (pl.sym && !pl.guard) { WGL_DISPATCH(pl.R, pl.fma, false, LAUNCH_WGLS); }
            else { WGL_DISPATCH(pl.R, pl.fma, pl.guard, LAUNCH_WGL); }
#undef LAUNCH_WGL
#undef LAUNCH_WGLS
        } else {
#define LAUNCH_WGR(T_, R_, C_, G_) e = launch_wgres<T_, R_, C_, G_>(c, pl, c->x[c->cur], c->x[c->cur ^ 1], chunk, c->res_epoch)
            WGT_DISPATCH(pl.T, pl.R, pl.fma, pl.guard, LAUNCH_WGR);
#undef LAUNCH_WGR
        }
        if (e != hipSuccess) return fail(DEFF_EHIP, "resident launch failed: %s", hipGetErrorString(e));
        c->res_epoch += (unsigned)chunk;
        c->res_pending = true;
        c->cur ^= (chunk & 1);
        np -= chunk;
        *n -= (int64_t)chunk * pl.T;
        c->res_redo += (int64_t)chunk * pl.T;
        ++c->last_launches;
    }
    return DEFF_OK;
}

// n sweeps: as many T-sweep passes as fit, the rest one at a time.  Stops at the first launch that fails.
int enqueue_sweeps(deff_ctx *c, const SweepPlan &pl, int64_t n)
{
    if (pl.kernel == DEFF_KERNEL_MATFREE_TB && pl.impl == 2 && pl.resident && !c->tb_resident) {
        // the context fell back to one launch per pass (resident_check) after this plan was made
        SweepPlan again;
        TRY(plan_sweeps(c, pl.omega, &again));
        if (again.resident) return fail(DEFF_ESTATE, "internal: plan still resident after the fallback");
        return enqueue_sweeps(c, again, n);
    }
    // (while resident launches are in flight unchecked, whatever follows them is part of what a fallback must redo)
    if (pl.kernel == DEFF_KERNEL_MATFREE_TB && pl.impl == 2 && pl.resident && n >= (pl.NW != WGT_WAVES ? 1 : 2) * pl.T)
        TRY(launch_resident_passes(c, pl, &n));
    if (pl.kernel == DEFF_KERNEL_MATFREE_TB && !(pl.impl == 2 && pl.NW != WGT_WAVES)) {
        while (n >= pl.T) { TRY(enqueue_tb_pass(c, pl)); n -= pl.T; ++c->last_launches; if (c->res_pending) c->res_redo += pl.T; }
    }
    for (; n > 0; --n) { enqueue_sweep(c, pl); ++c->last_launches; if (c->res_pending) ++c->res_redo; }
    HIP_TRY(hipPeekAtLastError());
    return DEFF_OK;
}

extern "C" int deff_sweeps(deff_ctx *c, int64_t nsweeps, double omega, float *ms)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    if (nsweeps < 0) return fail(DEFF_EINVAL, "negative sweep count");
    TRY(use_device(c));
    SweepPlan pl;
    TRY(plan_sweeps(c, omega, &pl));
    TRY(consolidate(c));
    c->last_launches = 0;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    TRY(enqueue_sweeps(c, pl, nsweeps));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    if (ms) HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    TRY(resident_check(c));
    TRY(dealt_watch(c));
    return DEFF_OK;
}
DEFF_API_CATCH

// Wall fluxes of the current field (cuh:1256-1257) for every stacked row, brought to the
// pinned host buffer: mf_host[0..rows) left wall, mf_host[rows..2*rows) right wall.
int flux_rows(deff_ctx *c, bool need_rows)
{
    if (!c->have_walls)
        return fail(DEFF_ESTATE, "wall diffusivities unknown: pass D to deff_set_system() or assemble on the device");
    TRY(resident_check(c));
    hipLaunchKernelGGL(k_wall_flux, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, c->x[c->cur], c->Dl,
                       c->Dr, c->nx, c->nxt, c->rows, c->dx, c->CL, c->CR, c->mf);
    HIP_TRY(hipGetLastError());
    c->q_valid = false;
    if (c->flux_reduce && !c->slab) {
        // sums on the device (kernels_setup.hpp: k_flux_sum): the check moves 16 B per image
        TRY(dev_alloc(&c->q, (size_t)2 * c->nimg));
        if (!c->q_host) HIP_TRY(hipHostMalloc((void **)&c->q_host, sizeof(double) * 2 * c->nimg));
        const dim3 grid((c->nimg + 3) / 4);
        if (c->flux_reduce == 2) hipLaunchKernelGGL(k_flux_sum<true>, grid, dim3(256), 0, c->stream, c->mf, c->rows, c->ny, c->nimg, c->q);
        else hipLaunchKernelGGL(k_flux_sum<false>, grid, dim3(256), 0, c->stream, c->mf, c->rows, c->ny, c->nimg, c->q);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->q_host, c->q, sizeof(double) * 2 * c->nimg, hipMemcpyDeviceToHost, c->stream));
        c->q_valid = true;
        if (!need_rows) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            return dealt_watch(c);
        }
    }
    HIP_TRY(hipMemcpyAsync(c->mf_host, c->mf, sizeof(double) * 2 * c->rows, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return dealt_watch(c);
}

// Deff of image k from its wall fluxes, summed in row order like the reference (cuh:1258-1263): on the host from
// mf_host, or from the device's sums when flux_rows() produced them.
static double deff_of_image(const deff_ctx *c, int k)
{
    double Q1 = 0, Q2 = 0;
    if (c->q_valid) {
        Q1 = c->q_host[2 * k];
        Q2 = c->q_host[2 * k + 1];
    } else {
        const double *L = c->mf_host + (size_t)k * c->ny, *R = c->mf_host + c->rows + (size_t)k * c->ny;
        for (int j = 0; j < c->ny; ++j) {
            Q1 += L[j];
            Q2 += R[j];
        }
    }
    const double qAvg = (Q1 + Q2) / (2.0 * c->ny);
    return qAvg / ((c->CR - c->CL));
}

static void copy_fluxes(const deff_ctx *c, double *MFL, double *MFR)
{
    if (MFL) memcpy(MFL, c->mf_host, sizeof(double) * c->rows);
    if (MFR) memcpy(MFR, c->mf_host + c->rows, sizeof(double) * c->rows);
}

// deff_raw: nimg values (one per stacked image); MFL/MFR: rows values each, may be NULL.
extern "C" int deff_flux(deff_ctx *c, double *deff_raw, double *MFL, double *MFR)
try {
    if (!c || !deff_raw) return fail(DEFF_EINVAL, "NULL argument");
    if (!c->have_field) return fail(DEFF_ESTATE, "no field");
    TRY(use_device(c));
    TRY(consolidate(c));
    TRY(flux_rows(c, MFL || MFR));
    for (int k = 0; k < c->nimg; ++k) deff_raw[k] = deff_of_image(c, k);
    copy_fluxes(c, MFL, MFR);
    return DEFF_OK;
}
DEFF_API_CATCH

// JacobiGPU's loop, cuh:1232-1290, with the sweeps between two checks enqueued without host
// round trips.  `iter` counts completed sweeps; the sweep with 0-based index k is followed by a
// check iff k % check_every == 0 (cuh:1243).  All images of a batch start together, so their
// checks coincide; each image carries its own deffOld / change and drops out (is frozen in the
// buffer it is in) as soon as ITS stopping rule fires -- exactly what a one-image-at-a-time run
// of the reference's loop would do.
extern "C" int deff_solve_batch(deff_ctx *c, double omega, double tol, int64_t max_iter, int64_t check_every,
                                deff_result *out, double *MFL, double *MFR)
try {
    if (!c || !out) return fail(DEFF_EINVAL, "NULL argument");
    if (check_every < 1) return fail(DEFF_EINVAL, "check_every must be >= 1");
    TRY(use_device(c));
    SweepPlan pl;
    TRY(plan_sweeps(c, omega, &pl));
    if (!c->have_walls) return fail(DEFF_ESTATE, "wall diffusivities unknown (needed for Deff)");
    TRY(consolidate(c));                                             // x is in/out: warm start from x[cur]
    reset_batch_state(c);                                            // every image iterates again

    const int B = c->nimg;
    std::vector<double> deffNew(B, 1.0), deffOld(B, 5.0), change(B, 100.0), conv(B, 0.0);   // cuh:1171-1173
    std::vector<int64_t> iters(B, 0), checks(B, 0);
    int n_active = (max_iter > 0 && tol < 100.0) ? B : 0;           // cuh:1232 with change = 100
    if (n_active == 0) c->active_h.assign(B, 0);
    int64_t iter = 0;
    c->last_launches = 0;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    while (iter < max_iter && n_active > 0) {                        // cuh:1232
        const int64_t next_check = ((iter + check_every - 1) / check_every) * check_every;
        const bool do_check = next_check < max_iter;
        const int64_t batch = do_check ? next_check - iter + 1 : max_iter - iter;
        TRY(enqueue_sweeps(c, pl, batch));
        iter += batch;
        for (int k = 0; k < B; ++k)
            if (c->active_h[k]) { iters[k] = iter; c->buf_of[k] = (uint8_t)c->cur; }
        if (do_check) {
            TRY(flux_rows(c, MFL || MFR));
            bool froze = false;
            for (int k = 0; k < B; ++k) {
                if (!c->active_h[k]) continue;
                deffNew[k] = deff_of_image(c, k);
                change[k] = (deffOld[k] - deffNew[k]) / (deffOld[k]);           // cuh:1265
                deffOld[k] = deffNew[k];
                conv[k] = change[k];                                             // cuh:1275
                ++checks[k];
                if (B == 1 && c->progress) c->progress(next_check, deffNew[k], change[k], c->progress_user);
                if (!(tol < fabs(change[k]))) { c->active_h[k] = 0; --n_active; froze = true; }
            }
            if (B == 1) copy_fluxes(c, MFL, MFR);
            else {
                // keep, per image, the fluxes of ITS last check
                for (int k = 0; k < B; ++k)
                    if (checks[k] && iters[k] == iter) {
                        if (MFL) memcpy(MFL + (size_t)k * c->ny, c->mf_host + (size_t)k * c->ny, sizeof(double) * c->ny);
                        if (MFR) memcpy(MFR + (size_t)k * c->ny, c->mf_host + c->rows + (size_t)k * c->ny,
                                        sizeof(double) * c->ny);
                    }
            }
            if (froze && n_active > 0) {
                TRY(dev_alloc(&c->active, (size_t)B));
                HIP_TRY(hipMemcpyAsync(c->active, c->active_h.data(), (size_t)B, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                c->masked = true;
            }
        }
    }
    TRY(resident_check(c));                                          // a solve that ends between two checks (MAX_ITER)
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    for (int k = 0; k < B; ++k) {
        out[k].iters = iters[k];
        out[k].checks = checks[k];
        out[k].deff_raw = deffNew[k];                                // cuh:1309: value at the last check
        out[k].conv = conv[k];
        out[k].loop_ms = ms;                                         // the batch shares one loop
    }
    return DEFF_OK;
}
DEFF_API_CATCH

// ---- streaming batch ---------------------------------------------------------------------
//
// Dataset generation with images that converge after very different numbers of sweeps: a plain
// batch drains (its slots empty one by one, measured 376 of ~1000 G cells*iter/s end to end on 512
// images of 128^2).  Here a slot whose image has finished is REFILLED with the next image.  To keep
// every image on the reference's schedule -- checks after its own sweeps 1, C+1, 2C+1, ... -- new
// images enter exactly one sweep before a check of the running ones: that sweep is their sweep 1,
// so all slots share the check points for ever.  Per image the arithmetic and the stopping rule are
// those of a one-image run (cuh:1232-1290).

// put image `slot`'s pixels / codes / wall data / linear guess in place (2-phase native system)
static int stream_load_slot(deff_ctx *c, int slot, const uint8_t *pix_host)
{
    TRY(resident_check(c));                            // the restart copy of a resident interval predates this slot's new image
    const size_t npix = (size_t)c->W * c->H;
    uint8_t *dpix = c->pix + (size_t)slot * npix;
    uint16_t *dcode = c->code + (size_t)slot * c->n_img;
    HIP_TRY(hipMemcpyAsync(dpix, pix_host, npix, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));        // pix_host is the caller's scratch
    hipLaunchKernelGGL(k_phase_codes, dim3(grid_for(c->n_img)), dim3(256), 0, c->stream, dpix, c->W, c->ampX, c->ampY,
                       c->nx, c->nxt, c->ny, c->ny, 0, c->ny, dcode);
    hipLaunchKernelGGL(k_wall_D_2phase, dim3((c->ny + 255) / 256), dim3(256), 0, c->stream, dpix, c->W, c->ampX, c->ampY,
                       c->nxt, c->ny, c->ny, c->Df, c->Ds, c->Dl + (size_t)slot * c->ny, c->Dr + (size_t)slot * c->ny);
    hipLaunchKernelGGL(k_init_linear, dim3(grid_for(c->n_img)), dim3(256), 0, c->stream,
                       c->x[c->cur] + (size_t)slot * c->n_img, c->nx, c->nxt, c->ny, c->CL, c->CR, c->fma);
    HIP_TRY(hipGetLastError());
    c->buf_of[slot] = (uint8_t)c->cur;
    c->links_sym = 0;                                  // new codes in this slot
    return DEFF_OK;
}

static int stream_push_mask(deff_ctx *c, int n_active)
{
    TRY(resident_check(c));                            // ... and the mask the interval ran under
    bool all = true;
    for (int k = 0; k < c->nimg; ++k) all = all && c->active_h[k];
    c->masked = !all && n_active > 0;
    if (c->masked) {
        TRY(dev_alloc(&c->active, (size_t)c->nimg));
        HIP_TRY(hipMemcpyAsync(c->active, c->active_h.data(), (size_t)c->nimg, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return DEFF_OK;
}

extern "C" int deff_get_slot_field(deff_ctx *c, int slot, double *x)
try {
    if (!c || !x || slot < 0 || slot >= c->nimg) return fail(DEFF_EINVAL, "bad slot");
    TRY(use_device(c));
    TRY(rows_d2h(c, x, (const double *)(c->x[(c->masked || c->in_stream) ? c->buf_of[slot] : c->cur] + (size_t)slot * c->n_img),
                 (size_t)c->ny));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_solve_stream(deff_ctx *c, int W, int H, int ampX, int ampY, double Ds, double Df, double CL,
                                 double CR, double omega, double tol, int64_t max_iter, int64_t check_every,
                                 deff_next_image_fn next, deff_image_done_fn done, void *user)
try {
    if (!c || !next || !done) return fail(DEFF_EINVAL, "NULL argument");
    if (check_every < 1) return fail(DEFF_EINVAL, "check_every must be >= 1");
    if (c->slab) return fail(DEFF_EINVAL, "not for slab contexts");
    TRY(use_device(c));
    TRY(image_shape(c, W, H, ampX, ampY));
    TRY(ensure_walls(c));
    TRY(dev_alloc(&c->code, c->n));
    const int B = c->nimg;
    c->CL = CL; c->CR = CR; c->Ds = Ds; c->Df = Df;
    c->phase_mode = 2; c->phase_D[0] = Df; c->phase_D[1] = Ds; c->phase_D[2] = 0.0;
    build_lut_rows(c, Ds, Df, CL, CR);
    c->have_image = true; c->have_walls = true; c->have_matfree = true; c->have_explicit = false;
    c->links_sym = 0;
    c->dict_tried = false; c->have_field = true;
    HIP_TRY(hipMemsetAsync(c->code, 0, sizeof(uint16_t) * c->n, c->stream));      // empty slots: zero rows
    HIP_TRY(hipMemsetAsync(c->x[0], 0, sizeof(double) * c->n, c->stream));
    HIP_TRY(hipMemsetAsync(c->x[1], 0, sizeof(double) * c->n, c->stream));
    reset_batch_state(c);

    struct Slot { bool live = false; int64_t id = -1, iters = 0, checks = 0; double deffNew = 1, deffOld = 5, change = 100, conv = 0; };
    std::vector<Slot> S(B);
    std::vector<uint8_t> pixbuf((size_t)W * H);
    bool more = true;
    int n_active = 0;
    auto refill = [&]() -> int {                       // fill every free slot while images remain
        for (int k = 0; k < B && more; ++k) {
            if (S[k].live) continue;
            int64_t id = -1;
            const int got = next(user, k, pixbuf.data(), &id);
            if (got < 0) return fail(DEFF_EINVAL, "image source reported an error");
            if (got == 0) { more = false; break; }
            TRY(stream_load_slot(c, k, pixbuf.data()));
            S[k] = Slot();
            S[k].live = true; S[k].id = id;
            c->active_h[k] = 1;
            ++n_active;
        }
        return DEFF_OK;
    };
    auto retire = [&](int k, float ms) {
        deff_result r;
        r.iters = S[k].iters; r.checks = S[k].checks; r.deff_raw = S[k].deffNew; r.conv = S[k].conv; r.loop_ms = ms;
        c->active_h[k] = 0;
        --n_active;
        done(user, S[k].id, k, &r);                    // the slot's field is still readable (deff_get_slot_field)
        S[k].live = false;
    };
    auto advance = [&](SweepPlan &pl, int64_t nsw) -> int {
        if (nsw <= 0) return DEFF_OK;
        TRY(enqueue_sweeps(c, pl, nsw));
        for (int k = 0; k < B; ++k)
            if (S[k].live) { S[k].iters += nsw; c->buf_of[k] = (uint8_t)c->cur; }
        return DEFF_OK;
    };

    for (int k = 0; k < B; ++k) c->active_h[k] = 0;
    c->in_stream = true;
    struct Leave { deff_ctx *c; ~Leave() { c->in_stream = false; } } leave{c};
    TRY(refill());
    SweepPlan pl;
    TRY(plan_sweeps(c, omega, &pl));
    c->last_launches = 0;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (!(max_iter > 0 && tol < 100.0)) {               // cuh:1232 with change = 100: no sweep at all
        for (;;) {
            for (int k = 0; k < B; ++k) if (S[k].live) retire(k, 0.f);
            if (!more) break;
            TRY(refill());
            if (n_active == 0) break;
        }
    }
    // phase of max_iter inside a check interval: live images sit at iters = j*C + 1 after a check
    while (n_active > 0) {
        TRY(stream_push_mask(c, n_active));
        // one sweep (the first of the newly loaded images, sweep j*C + 1 of the others), then the check
        TRY(advance(pl, 1));
        TRY(flux_rows(c, false));                      // a stream reports no per-row fluxes
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        HIP_TRY(hipEventSynchronize(c->ev1));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        for (int k = 0; k < B; ++k) {
            if (!S[k].live) continue;
            Slot &s = S[k];
            s.deffNew = deff_of_image(c, k);
            s.change = (s.deffOld - s.deffNew) / (s.deffOld);                     // cuh:1265
            s.deffOld = s.deffNew;
            s.conv = s.change;
            ++s.checks;
            if (!(tol < fabs(s.change)) || s.iters >= max_iter) retire(k, ms);
        }
        if (n_active == 0 && !more) break;
        // up to the next check: C - 1 sweeps, split where images run into max_iter (they all carry
        // iters = j*C + 1 with their own j, so each reaches max_iter the same distance after a check)
        int64_t left = check_every - 1;
        while (left > 0 && n_active > 0) {
            int64_t seg = left;
            for (int k = 0; k < B; ++k)
                if (S[k].live && max_iter - S[k].iters < seg) seg = max_iter - S[k].iters;
            if (seg > 0) {
                TRY(stream_push_mask(c, n_active));
                TRY(advance(pl, seg));
                left -= seg;
            }
            bool hit = false;
            for (int k = 0; k < B; ++k)
                if (S[k].live && S[k].iters >= max_iter) { hit = true; }
            if (hit) {
                TRY(resident_check(c));                // the retired images' fields are handed out now: settle the interval first
                HIP_TRY(hipEventRecord(c->ev1, c->stream));
                HIP_TRY(hipEventSynchronize(c->ev1));
                float ms2 = 0;
                HIP_TRY(hipEventElapsedTime(&ms2, c->ev0, c->ev1));
                for (int k = 0; k < B; ++k)
                    if (S[k].live && S[k].iters >= max_iter) retire(k, ms2);   // MAX_ITER reached between checks, cuh:1232
            }
        }
        TRY(refill());                                 // newcomers start with the sweep that precedes the next check
        // new images, new codes: the symmetric short-cut of the tall tiles is re-verified, not carried over
        if (pl.impl == 2 && pl.NW == WGL_WAVES && c->tb_sym != 2 && c->links_sym == 0) {
            TRY(check_links_symmetric(c));
            pl.sym = c->links_sym == 1;
        } else if (pl.impl == 2 && pl.NW == WGS_WAVES && c->links_sym == 0) {
            TRY(check_links_symmetric(c));
            if (c->links_sym != 1) { pl = SweepPlan(); TRY(plan_sweeps(c, omega, &pl)); }   // (never for the native assembly)
        }
    }
    TRY(resident_check(c));                            // nothing unchecked outlives the stream's mask and slots
    c->masked = false;
    reset_batch_state(c);
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_solve(deff_ctx *c, double omega, double tol, int64_t max_iter, int64_t check_every,
                          deff_result *out, double *MFL, double *MFR)
try {
    if (c && c->nimg != 1) return fail(DEFF_EINVAL, "context holds %d images: use deff_solve_batch()", c->nimg);
    return deff_solve_batch(c, omega, tol, max_iter, check_every, out, MFL, MFR);
}
DEFF_API_CATCH

// Diagnostics: time-stamp every wave tile of ONE temporally blocked pass (100 MHz wall clock ticks).
// out[2*k], out[2*k+1] = start, end of wave tile k; *ntiles = number of tiles (call with out = NULL
// to size the buffer).  Advances the field by one pass.
extern "C" int deff_debug_tb_stamps(deff_ctx *c, double omega, unsigned long long *out, int *ntiles)
try {
    if (!c || !ntiles) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    SweepPlan pl;
    TRY(plan_sweeps(c, omega, &pl));
    if (pl.kernel != DEFF_KERNEL_MATFREE_TB) return fail(DEFF_ESTATE, "not on the temporally blocked kernel");
    TRY(consolidate(c));
    // streaming form: 2 stamps per wave tile; workgroup-tile form: T + 4 per tile, flattened -- *ntiles is always
    // the number of PAIRS the buffer must hold
    // (resident launches stamp 12 clocks per tile whatever T: entry + 3 passes x {neighbours seen, halo in, swept, published})
    const bool res_stamps = pl.impl == 2 && pl.resident;
    const int n = res_stamps ? (pl.ntx * pl.tgy * 12 + 1) / 2 : pl.impl == 2 ? (pl.ntx * pl.tgy * (pl.T + 4) + 1) / 2 : pl.ntx * pl.tgy;
    *ntiles = n;
    if (!out) return DEFF_OK;
    HIP_TRY(hipMalloc((void **)&c->tb_stamps, sizeof(unsigned long long) * 2 * n));
    HIP_TRY(hipMemsetAsync(c->tb_stamps, 0, sizeof(unsigned long long) * 2 * n, c->stream));
    int rc = DEFF_OK;
    if (res_stamps) rc = enqueue_sweeps(c, pl, 3 * pl.T);                                  // k_sweep_wgres / wgsym: 12 stamps per tile, 3 passes
    else rc = enqueue_tb_pass(c, pl);
    if (rc != DEFF_OK) { (void)hipFree(c->tb_stamps); c->tb_stamps = nullptr; return rc; }
    hipError_t e = hipMemcpyAsync(out, c->tb_stamps, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(c->tb_stamps);
    c->tb_stamps = nullptr;
    if (e != hipSuccess) return fail(DEFF_EHIP, "stamp readback failed: %s", hipGetErrorString(e));
    TRY(resident_check(c));
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_set_progress(deff_ctx *c, deff_progress_fn fn, void *user)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    c->progress = fn;
    c->progress_user = user;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_last_launches(const deff_ctx *c, int64_t *launches, int *sweeps_per_pass)
try {
    if (!c || !launches) return fail(DEFF_EINVAL, "NULL argument");
    *launches = c->last_launches;
    if (sweeps_per_pass) {
        int k = 0;
        *sweeps_per_pass = 1;
        if (resolve_kernel(c, &k) == DEFF_OK && k == DEFF_KERNEL_MATFREE_TB) {
            *sweeps_per_pass = c->plan_T ? c->plan_T : clamp_tb_T(c->tb_T ? c->tb_T : default_tb_T(c));   // (the planner may take T = 6)
        }
    }
    return DEFF_OK;
}
DEFF_API_CATCH
