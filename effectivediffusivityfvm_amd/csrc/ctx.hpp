// ctx.hpp -- what the translation units of libdeff_amd.so share: the solver context, the error
// helpers and the internal entry points that cross files.
//   api_core.hip   library, context lifecycle, image, assembly (native / from D / imported), field
//   api_solve.hip  row dictionary, launch plans, sweeps, wall fluxes, the solve loops (one image,
//                  batch, streaming batch)
//   api_slab.hip   one image over several GPUs: row slabs (peer copies in one process, RCCL or a
//                  caller-supplied transport with one process per GPU)
// The library is built with -fvisibility=hidden; only the C ABI of include/deff_amd.h is exported.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/deff_amd.h"
#pragma GCC visibility pop
#include "fvm_row.hpp"
#include "kernels_setup.hpp"
#include "lut_layout.hpp"

using namespace deff;

// ------------------------------------------------------------- errors -----

extern thread_local char g_err[512];                 // message of this thread's last failure (api_core.hip)
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(e_ == hipErrorOutOfMemory ? DEFF_ENOMEM : DEFF_EHIP,             \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                       \
    } while (0)

#define TRY(expr)                  \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != DEFF_OK) return rc_; \
    } while (0)

// ------------------------------------------------------------ context -----

struct deff_ctx {
    int device = 0;
    int nx = 0, ny = 0;             // mesh of ONE image
    int nimg = 1;                   // images stacked in this context (batch), see kernels_setup.hpp
    int rows = 0;                   // nimg * ny
    size_t n_img = 0;               // cells per image
    size_t n = 0;                   // cells in the stack
    double dx = 0, dy = 0;
    // nx is the width of every device array (even); nxt <= nx the width of the mesh -- an odd mesh
    // width is padded by one column of cells outside the mesh (kernels_setup.hpp, "Row pitch")
    int nxt = 0;
    // Row slab of a taller image (multi-GPU split of one image): the arrays hold
    // `halo` rows above and below the `own_h` rows this context updates; array row li is mesh row
    // li - dom_lo of a mesh_ny-row mesh.  Plain contexts: dom_lo = 0, mesh_ny = own_h = ny, halo = 0.
    bool slab = false;
    int dom_lo = 0, mesh_ny = 0, own_lo = 0, own_h = 0, halo = 0;
    // cells the default sweeps-per-pass is keyed on: 0 = this context's own n; for a slab a figure derived from
    // (nx, NY, number of slabs) alone, so that every slab / rank of one image plans the SAME T whatever its own
    // row count (slabs differ by one row; a different T per slab would desynchronise passes and exchanges)
    size_t tb_ref_cells = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // image (pixels as decoded, W x H, before mesh amplification)
    uint8_t *pix = nullptr;
    int W = 0, H = 0, ampX = 1, ampY = 1;
    bool have_image = false;

    // explicit SoA system
    double *a0 = nullptr, *c0 = nullptr, *aW = nullptr, *aE = nullptr, *aS = nullptr, *aN = nullptr,
           *b = nullptr;
    bool have_explicit = false;
    double c0_omega = NAN;          // omega the c0 plane was built for

    // matrix-free system: one 16-bit code per cell + the dictionary of distinct rows (lut_layout.hpp)
    uint16_t *code = nullptr;
    double *lut = nullptr;          // device tables [LUT_PLANES][LUT_PLANE_STRIDE]
    std::vector<double> lut_rows;   // host: rows [nrows][6] = A0, aW, aE, aS, aN, b (row 0 = zeros)
    int lut_nrows = 0;
    bool lut_allb = false;          // some row away from the walls has b != 0 (harvested dictionaries only)
    bool lut_guard = false;         // some c0 = w/A0 is not finite: the reference's non-zero link test matters
    bool have_matfree = false;
    bool dict_tried = false;        // a dictionary was already looked for in the current explicit system
    int dict_enabled = 1;
    // an imported system (deff_set_system) with a non-zero W link in the first column or E link in the last:
    // the reference's kernel addresses neighbours linearly (x[p-1], x[p+1], cuh:80-83), so such a link reads the
    // neighbouring ROW's end cell.  Only the explicit / scalar kernels keep that addressing; the dictionary and the
    // temporally blocked kernel treat a wall column's outer neighbour as absent -- such a system stays explicit.
    bool wrap_links = false;
    double lut_omega = NAN;
    double Ds = 0, Df = 0;          // phase diffusivities of the native 2-phase system
    // what deff_residual() needs to know about a system assembled from the image: 2 / 3 pixel classes (0 = the system came
    // from a caller's D plane or matrix) and their diffusivities {fluid, solid, gas}
    int phase_mode = 0;
    double phase_D[3] = {0, 0, 0};
    double *resid = nullptr;        // device: partial sums of the residual reduction + one sum per image
    size_t resid_cap = 0;
    int res_kt = 0;                 // tuning: tiles of 8 rows a wave of the residual kernel streams through (0 = planner)

    // wall data for the flux evaluation
    double *Dl = nullptr, *Dr = nullptr;
    double CL = 0, CR = 0;
    bool have_walls = false;
    double *mf = nullptr;           // device: 2*ny fluxes
    double *mf_host = nullptr;      // pinned
    // where the fluxes are summed: 0 = on the host, in row order (default); 1 = on the device in the same order (bit-identical;
    // a check then moves 16 B per image instead of 16 B per row); 2 = on the device by a wave-level tree (fixed order, not the
    // reference's: ~1e-16 relative).  Row slabs always sum on the host (the rows of one image live on several devices).
    int flux_reduce = 0;
    double *q = nullptr;            // device: {Q1, Q2} per image
    double *q_host = nullptr;       // pinned
    bool q_valid = false;           // q_host holds the sums of the last flux_rows()

    // field, ping-pong
    double *x[2] = {nullptr, nullptr};
    int cur = 0;
    bool have_field = false;
    // batch bookkeeping: images still iterating (device mask is only bound while some are
    // frozen) and, per image, the ping-pong buffer that holds its newest field
    uint8_t *active = nullptr;
    std::vector<uint8_t> active_h, buf_of;
    bool masked = false;
    bool in_stream = false;         // inside deff_solve_stream: buf_of[] is current for every slot

    // scratch for chunked uploads (AoS import, D upload)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;

    deff_progress_fn progress = nullptr;
    void *progress_user = nullptr;

    int kernel = DEFF_KERNEL_AUTO;
    int rows_explicit = 0, rows_matfree = 0;     // rows per register tile, 0 = default
    int wg_matfree = 0;                          // persistent workgroups of the matrix-free kernel, 0 = default
    int nt_explicit = 1;                         // non-temporal coefficient loads in the explicit kernels
    int serpentine = 1;                          // alternate the tile walk direction from sweep to sweep
    int tb_T = 0, tb_LY = 0, tb_wg = 0;          // temporal blocking: sweeps per pass, rows per chunk, workgroups
    unsigned long long *tb_stamps = nullptr;     // diagnostics: per wave-tile start/end clocks (deff_debug_tb_stamps)
    int tb_wall_halo = 2;                        // strip placement: 1 = halo also outside the walls, 0 = not, 2 = whichever needs fewer strips
    int plan_T = 0, plan_LY = 0, plan_ntx = 0, plan_cpi = 0, plan_blocks = 0;   // last temporally blocked plan
    int plan_impl = 0, plan_R = 0, plan_NW = 0;
    // form of a temporally blocked pass: 0 = chosen by the planner, 1 = streaming (kernels_tb.hpp: one wave per tile),
    // 2 = workgroup tiles (kernels_wgtile.hpp: 8 waves per tile, rows resident in registers); tb_R = rows per wave there
    int tb_impl = 0, tb_R = 0, tb_NW = 0;
    // resident passes (kernels_wgtile.hpp, k_sweep_wgres): when every tile of the context is on the chip at once, all the
    // passes between two checks are ONE launch whose tiles keep their matrix rows and owned cells in registers and wait
    // for their neighbours only.  Tuning key "tb_launch": 0 = resident passes whenever the tiles are co-resident (default:
    // the grid fits the chip by the occupancy query, resident launches of one process are chained per device,
    // api_solve.hip, so that two of them never share the chip, and every wait is bounded); 1 = never (one launch per pass).
    // What the process cannot rule out -- another PROCESS on the same GPU, a CU mask -- ends in a bounded wait running
    // out; the solve then restores the field it had when the interval's first resident launch was enqueued, redoes the
    // interval with one launch per pass and stays in that mode (res_fallbacks counts these; deff_get_plan "tb_fallbacks").
    int tb_resident = 1;
    // is the matrix-free system link-symmetric (k_links_symmetric)?  0 = not looked at since the codes / dictionary last
    // changed, 1 = yes (tall tiles then do 7 lookups per row instead of 10), 2 = no
    int links_sym = 0;
    int tb_sym = 0;                              // tuning: 2 = never use the symmetric short-cut (A/B, tests)
    int tb_debug_stall = 0;                      // tests: tile (index + 1) that leaves a resident launch without publishing
    int tb_debug_stall_skip = 0;                 // tests: ... after this many further resident launches
    int plan_resident = 0;                       // the last plan used resident passes
    unsigned *res_flags = nullptr;               // per tile: passes completed (epoch counter)
    size_t res_flags_n = 0;
    unsigned res_epoch = 0;
    unsigned *res_abort = nullptr;               // raised by a workgroup whose wait for a neighbour ran out
    bool res_pending = false;                    // a resident launch was enqueued since the flag was last read
    // restart point of the resident launches in flight: a copy of x[res_backup_cur] taken in front of the first resident
    // launch after a synchronised look at the abort flag, and the sweeps enqueued since (with their omega)
    double *res_backup = nullptr;
    int res_backup_cur = 0;
    int64_t res_redo = 0;
    double res_omega = 0;
    int res_fallbacks = 0;                       // aborted resident intervals redone with one launch per pass
    bool chain_counted = false;                  // this context is counted among its device's users of the resident chain's event
    int fma = 0;                                 // contracted arithmetic (kernels_sweep.hpp), opt-in
    int tb_xmajor = 1;                           // wave-tile numbering of the temporally blocked kernel
    // Streaming kernel, chunk heights by service order (deal_ranked_tiles, api_solve.hip): 0 = equal chunks, 1 = dealt tiles.
    // tb_rank_w: the relative speed (per mille) of a SIMD's oldest / second / youngest wave, tb_rank_wall: what a row of a wall strip
    // costs, per mille of an inner strip's (its waves look up b as well)
    int tb_tall_deal = 1;                        // tall tiles: rows dealt by the waves' age where a kernel for it exists (api_solve.hip, WGAGE_SETS)
    int plan_aged = 0;
    int tb_sym_age = 1;                          // 12-wave tiles: the shapes with a row less for the youngest waves (api_solve.hip, SYM_SHAPES_T8)
    int tb_sym_shape = 0;                        // tests: 1-based index into SYM_SHAPES_T8 (0: the planner's choice)
    int tb_ranked = 1;
    int tb_rank_w[3] = {460, 325, 215};
    int tb_rank_wall = 1100;
    int4 *tb_dealt = nullptr;                    // device table: (strip, first row, rows, stamp index) per wave of the grid
    size_t tb_dealt_cap = 0;
    std::vector<int4> tb_dealt_host;
    std::vector<int> tb_dealt_key;               // what the table was built for
    size_t tb_dealt_miss_at = 0;                 // index of the table's last element: the waves' count of misplacements
    int64_t tb_dealt_waves = 0;                  // waves launched on the current table
    int tb_dealt_looks = 0;                      // times the count was read (dealt_watch: the first three synchronisations)
    int tb_rank_misses = 0, tb_rank_lost = 0;    // last count read; 1 = the dispatch order is not the assumed one: equal chunks from now on
    int tb_dealt_LY = 0, tb_dealt_nmax = 0;      // ... and what deff_get_plan reports of it (an inner strip's middle rank; most chunks per rank)
    int plan_ranked = 0;
    int64_t last_launches = 0;                   // sweep-kernel launches of the last deff_sweeps()/deff_solve()
};

static inline int use_device(const deff_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device));
    return DEFF_OK;
}

template <typename T>
static inline int dev_alloc(T **p, size_t count)
{
    if (*p) return DEFF_OK;
    HIP_TRY(hipMalloc((void **)p, count * sizeof(T)));
    return DEFF_OK;
}

static inline int ensure_scratch(deff_ctx *c, size_t bytes)
{
    if (c->scratch_bytes >= bytes) return DEFF_OK;
    if (c->scratch) { HIP_TRY(hipFree(c->scratch)); c->scratch = nullptr; c->scratch_bytes = 0; }
    HIP_TRY(hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return DEFF_OK;
}

static inline int ensure_explicit(deff_ctx *c)
{
    TRY(dev_alloc(&c->a0, c->n)); TRY(dev_alloc(&c->c0, c->n));
    TRY(dev_alloc(&c->aW, c->n)); TRY(dev_alloc(&c->aE, c->n));
    TRY(dev_alloc(&c->aS, c->n)); TRY(dev_alloc(&c->aN, c->n));
    TRY(dev_alloc(&c->b, c->n));
    return DEFF_OK;
}

static inline int ensure_walls(deff_ctx *c)
{
    TRY(dev_alloc(&c->Dl, (size_t)c->rows));
    TRY(dev_alloc(&c->Dr, (size_t)c->rows));
    TRY(dev_alloc(&c->mf, (size_t)2 * c->rows));
    if (!c->mf_host) HIP_TRY(hipHostMalloc((void **)&c->mf_host, sizeof(double) * 2 * c->rows));
    return DEFF_OK;
}

static inline int grid_for(size_t n, int cap = 16384)
{
    size_t g = (n + 255) / 256;
    return (int)(g < (size_t)cap ? (g ? g : 1) : (size_t)cap);
}

// rows of nxt elements on the host <-> rows of nx elements on the device
template <typename T>
static inline int rows_h2d(deff_ctx *c, T *dst, const T *src, size_t rows)
{
    if (c->nx == c->nxt)
        HIP_TRY(hipMemcpyAsync(dst, src, sizeof(T) * c->nx * rows, hipMemcpyHostToDevice, c->stream));
    else
        HIP_TRY(hipMemcpy2DAsync(dst, sizeof(T) * c->nx, src, sizeof(T) * c->nxt, sizeof(T) * c->nxt, rows,
                                 hipMemcpyHostToDevice, c->stream));
    return DEFF_OK;
}
template <typename T>
static inline int rows_d2h(deff_ctx *c, T *dst, const T *src, size_t rows)
{
    if (c->nx == c->nxt)
        HIP_TRY(hipMemcpyAsync(dst, src, sizeof(T) * c->nx * rows, hipMemcpyDeviceToHost, c->stream));
    else
        HIP_TRY(hipMemcpy2DAsync(dst, sizeof(T) * c->nxt, src, sizeof(T) * c->nx, sizeof(T) * c->nxt, rows,
                                 hipMemcpyDeviceToHost, c->stream));
    return DEFF_OK;
}

static inline CoefSoA soa_of(deff_ctx *c) { return CoefSoA{c->a0, c->aW, c->aE, c->aS, c->aN, c->b}; }

// No C++ exception may unwind through the C ABI: every `extern "C" int` entry point is a
// function-try-block ending in DEFF_API_CATCH, which turns std::bad_alloc (host vectors sized by the
// caller's image) and anything else into an error code + message.
static inline int api_exception() noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(DEFF_ENOMEM, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(DEFF_EINVAL, "internal error: %s", e.what());
    } catch (...) {
        return fail(DEFF_EINVAL, "internal error: unknown exception");
    }
}
#define DEFF_API_CATCH catch (...) { return api_exception(); }

// ------------------------------------------------- shared internals -------

struct SweepPlan {
    int kernel = 0;
    double omw = 0;
    int rows = 0, cpi = 0, gx = 0, gy = 0, blocks = 0;   // single-sweep kernels (cpi: row tiles per image)
    // temporally blocked kernel
    bool fma = false;
    int T = 0, LY = 0, tcpi = 0, ntx = 0, tgx = 0, tgy = 0, tblocks = 0;
    int shift = 0;                                        // column shift of the strips (0: no halo outside the walls)
    int T_override = 0;                                   // slab mode plans a T = 1 pass for remainders
    bool guard = false;
    double omega = 0;                                     // as given to plan_sweeps (omw = 1 - omega)
    int impl = 1, R = 0, NW = 8;                          // 1 = streaming kernel, 2 = workgroup tiles of NW waves x R rows
    const int4 *dealt = nullptr;                          // streaming kernel: dealt tiles (chunk heights by service order), or none
    bool resident = false;                                // impl 2 only: all passes of a batch in one launch (k_sweep_wgres)
    bool sym = false;                                     // tall tiles: the system is link-symmetric (7 lookups per row)
    bool aged = false;                                    // tall tiles: rows dealt by the waves' age (k_sweep_wgage)
    int rows3 = 0;                                        // 12-wave tiles: rows by age, a | b << 8 | c << 16 (0: R each)
    // rows the plan updates: band_h > 0 restricts it to the band [band_lo, band_lo + band_h) of the context's owned rows
    // (input); own_lo / own_h are what the planner resolved (output, passed to the kernels)
    int band_lo = 0, band_h = 0, own_lo = 0, own_h = 0;
};

// api_core.hip
void reset_batch_state(deff_ctx *c);
int resolve_kernel(const deff_ctx *c, int *k);
int image_shape(deff_ctx *c, int W, int H, int ampX, int ampY);
void build_lut_rows(deff_ctx *c, double Ds, double Df, double CL, double CR);
int upload_lut(deff_ctx *c, double omega);
int consolidate(deff_ctx *c);
int explicit_from_image(deff_ctx *c);
// api_solve.hip
int default_tb_T(const deff_ctx *c);
int clamp_tb_T(int T);
int default_tb_impl(const deff_ctx *c);
int plan_sweeps(deff_ctx *c, double omega, SweepPlan *pl);
void enqueue_sweep(deff_ctx *c, const SweepPlan &pl);
int enqueue_tb_pass(deff_ctx *c, const SweepPlan &pl);
int launch_tb_pass(deff_ctx *c, const SweepPlan &pl);    // the same launch without flipping x[cur]
int dealt_watch(deff_ctx *c);                            // after a synchronisation: is the dispatch order the dealt tiles assume?
int enqueue_sweeps(deff_ctx *c, const SweepPlan &pl, int64_t n);   // stops at the first launch that fails
// did a resident launch give up waiting?  (synchronises if one is pending; on an abort the interval is redone with one
// launch per pass and the context stays in that mode)
int resident_check(deff_ctx *c);
void resident_chain_ctx_created(int device);
void resident_chain_ctx_destroyed(int device);
int flux_rows(deff_ctx *c, bool need_rows = true);
