// deff_amd.hip -- solver context and C ABI (include/deff_amd.h) of the
// MI355X-native effective-diffusivity hot path.  Host side of the path the
// reference implements in Deff2DGPU/Deff2D.cuh: DiscretizeMatrix2D (cuh:815-902),
// initializeGPU/unInitializeGPU (cuh:904-1021), JacobiGPU (cuh:1163-1314).
//
// Design: one context per GPU owns every buffer and a private
// HIP stream; the image is uploaded as bytes (1 B/pixel) and everything else is
// produced on the device; sweeps are enqueued back to back with pointer
// ping-pong (no per-sweep sync or D2D copy, unlike cuh:1239/cuh:1281); a
// convergence check moves 16*ny bytes, not the field (cuh:1245).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/deff_amd.h"
#include "driver/jpeg_gray.hpp"
#include "flood_fill.hpp"
#include "fvm_row.hpp"
#include "kernels_setup.hpp"
#include "kernels_dict.hpp"
#include "kernels_sweep.hpp"
#include "kernels_tb.hpp"
#include "lut_layout.hpp"

using namespace deff;

// ------------------------------------------------------------- errors -----

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

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
    // Row slab of a taller image (multi-GPU split of one image): the arrays hold
    // `halo` rows above and below the `own_h` rows this context updates; array row li is mesh row
    // li - dom_lo of a mesh_ny-row mesh.  Plain contexts: dom_lo = 0, mesh_ny = own_h = ny, halo = 0.
    bool slab = false;
    int dom_lo = 0, mesh_ny = 0, own_lo = 0, own_h = 0, halo = 0;
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
    double lut_omega = NAN;
    double Ds = 0, Df = 0;          // phase diffusivities of the native 2-phase system

    // wall data for the flux evaluation
    double *Dl = nullptr, *Dr = nullptr;
    double CL = 0, CR = 0;
    bool have_walls = false;
    double *mf = nullptr;           // device: 2*ny fluxes
    double *mf_host = nullptr;      // pinned

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
    int tb_xmajor = 1;                           // wave-tile numbering of the temporally blocked kernel
    int64_t last_launches = 0;                   // sweep-kernel launches of the last deff_sweeps()/deff_solve()
};

static int use_device(const deff_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device));
    return DEFF_OK;
}

template <typename T>
static int dev_alloc(T **p, size_t count)
{
    if (*p) return DEFF_OK;
    HIP_TRY(hipMalloc((void **)p, count * sizeof(T)));
    return DEFF_OK;
}

static int ensure_scratch(deff_ctx *c, size_t bytes)
{
    if (c->scratch_bytes >= bytes) return DEFF_OK;
    if (c->scratch) { HIP_TRY(hipFree(c->scratch)); c->scratch = nullptr; c->scratch_bytes = 0; }
    HIP_TRY(hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return DEFF_OK;
}

static int ensure_explicit(deff_ctx *c)
{
    TRY(dev_alloc(&c->a0, c->n)); TRY(dev_alloc(&c->c0, c->n));
    TRY(dev_alloc(&c->aW, c->n)); TRY(dev_alloc(&c->aE, c->n));
    TRY(dev_alloc(&c->aS, c->n)); TRY(dev_alloc(&c->aN, c->n));
    TRY(dev_alloc(&c->b, c->n));
    return DEFF_OK;
}

static int ensure_walls(deff_ctx *c)
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

static CoefSoA soa_of(deff_ctx *c) { return CoefSoA{c->a0, c->aW, c->aE, c->aS, c->aN, c->b}; }

// No C++ exception may unwind through the C ABI: every `extern "C" int` entry point is a
// function-try-block ending in DEFF_API_CATCH, which turns std::bad_alloc (host vectors sized by the
// caller's image) and anything else into an error code + message.
static int api_exception() noexcept
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

// ---------------------------------------------------------- library -------

extern "C" const char *deff_version(void) { return "deff_amd 0.1 (gfx950)"; }
extern "C" const char *deff_last_error(void) { return g_err; }
extern "C" const char *deff_error_string(int code)
{
    switch (code) {
    case DEFF_OK: return "ok";
    case DEFF_EINVAL: return "invalid argument";
    case DEFF_EHIP: return "HIP runtime error";
    case DEFF_ENOMEM: return "out of memory";
    case DEFF_ENODEV: return "no usable device";
    case DEFF_ESTATE: return "system or field not set";
    case DEFF_ECOMM: return "RCCL error";
    default: return "unknown error";
    }
}

extern "C" int deff_device_count(int *count)
try {
    if (!count) return fail(DEFF_EINVAL, "count is NULL");
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) { *count = 0; return fail(DEFF_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = k;
    return DEFF_OK;
}
DEFF_API_CATCH

// -------------------------------------------------------- lifecycle -------

extern "C" int deff_create(int device, int nx, int ny, deff_ctx **out)
try {
    return deff_create_batch(device, nx, ny, 1, out);
}
DEFF_API_CATCH

extern "C" int deff_create_batch(int device, int nx, int ny, int nimg, deff_ctx **out)
try {
    if (!out) return fail(DEFF_EINVAL, "out is NULL");
    *out = nullptr;
    if (nx < 2 || ny < 2) return fail(DEFF_EINVAL, "mesh must be at least 2x2 (got %dx%d)", nx, ny);
    if (nimg < 1) return fail(DEFF_EINVAL, "batch size must be >= 1 (got %d)", nimg);
    if ((size_t)nx * (size_t)ny * (size_t)nimg > (size_t)1 << 31 || (long long)ny * nimg > (1ll << 30))
        return fail(DEFF_EINVAL, "%d image(s) of %dx%d exceed 2^31 cells", nimg, nx, ny);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(DEFF_ENODEV, "no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(DEFF_EINVAL, "device %d out of range [0,%d)", device, count);

    deff_ctx *c = new (std::nothrow) deff_ctx();
    if (!c) return fail(DEFF_ENOMEM, "host allocation failed");
    c->device = device;
    c->nx = nx; c->ny = ny; c->nimg = nimg; c->rows = nimg * ny;
    c->n_img = (size_t)nx * ny; c->n = c->n_img * nimg;
    c->active_h.assign(nimg, 1); c->buf_of.assign(nimg, 0);
    c->mesh_ny = ny; c->own_h = ny;
    c->dx = 1.0 / nx;           // cuh:1910-1911: the domain is always the unit square
    c->dy = 1.0 / ny;
    int rc = DEFF_OK;
    do {
        if ((rc = use_device(c)) != DEFF_OK) break;
        hipError_t he;
        if ((he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
            (he = hipEventCreate(&c->ev0)) != hipSuccess || (he = hipEventCreate(&c->ev1)) != hipSuccess) {
            rc = fail(DEFF_EHIP, "stream/event creation failed: %s", hipGetErrorString(he));
            break;
        }
        if ((rc = dev_alloc(&c->x[0], c->n)) != DEFF_OK) break;
        if ((rc = dev_alloc(&c->x[1], c->n)) != DEFF_OK) break;
        // the reference zero-fills its device arrays (cuh:946-973)
        if (hipMemsetAsync(c->x[0], 0, sizeof(double) * c->n, c->stream) != hipSuccess ||
            hipMemsetAsync(c->x[1], 0, sizeof(double) * c->n, c->stream) != hipSuccess) {
            rc = fail(DEFF_EHIP, "memset failed");
            break;
        }
    } while (0);
    if (rc != DEFF_OK) { deff_destroy(c); return rc; }
    *out = c;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_destroy(deff_ctx *c)
try {
    if (!c) return DEFF_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *bufs[] = {c->pix, c->a0, c->c0, c->aW, c->aE, c->aS, c->aN, c->b, c->code, c->lut,
                    c->Dl, c->Dr, c->mf, c->x[0], c->x[1], c->scratch, c->active};
    for (void *p : bufs) if (p) (void)hipFree(p);
    if (c->mf_host) (void)hipHostFree(c->mf_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;                      // no hipDeviceReset (the reference resets per image, cuh:1015)
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_mesh(const deff_ctx *c, int *nx, int *ny, double *dx, double *dy)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    if (nx) *nx = c->nx;
    if (ny) *ny = c->ny;
    if (dx) *dx = c->dx;
    if (dy) *dy = c->dy;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_batch_size(const deff_ctx *c, int *nimg)
try {
    if (!c || !nimg) return fail(DEFF_EINVAL, "NULL argument");
    *nimg = c->nimg;
    return DEFF_OK;
}
DEFF_API_CATCH

// All images iterate again and their newest field is in x[cur] (after a new guess / system).
static void reset_batch_state(deff_ctx *c)
{
    c->active_h.assign(c->nimg, 1);
    c->buf_of.assign(c->nimg, (uint8_t)c->cur);
    c->masked = false;
}

extern "C" int deff_set_kernel(deff_ctx *c, int kernel)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    if (kernel < DEFF_KERNEL_AUTO || kernel > DEFF_KERNEL_MATFREE_TB)
        return fail(DEFF_EINVAL, "unknown kernel id %d", kernel);
    c->kernel = kernel;
    return DEFF_OK;
}
DEFF_API_CATCH

// Which kernel a sweep will use, given what has been assembled.
static int resolve_kernel(const deff_ctx *c, int *k)
{
    int want = c->kernel;
    if (want == DEFF_KERNEL_AUTO) want = c->have_matfree ? DEFF_KERNEL_MATFREE_TB : DEFF_KERNEL_EXPLICIT;
    if (want == DEFF_KERNEL_MATFREE || want == DEFF_KERNEL_MATFREE_TB) {
        if (!c->have_matfree) {
            // an explicit system without a usable row dictionary stays on the explicit kernels
            if (!c->have_explicit) return fail(DEFF_ESTATE, "no system assembled");
            want = (c->nx & 1) ? DEFF_KERNEL_SCALAR : DEFF_KERNEL_EXPLICIT;
            *k = want;
            return DEFF_OK;
        }
        // the temporally blocked kernel needs 16-B aligned strips (even nx) and a few rows to stream
        if (want == DEFF_KERNEL_MATFREE_TB && ((c->nx & 1) || c->ny < 8)) want = DEFF_KERNEL_MATFREE;
    } else {
        if (!c->have_explicit && !c->have_matfree) return fail(DEFF_ESTATE, "no system assembled");
        if (want == DEFF_KERNEL_EXPLICIT && (c->nx & 1)) want = DEFF_KERNEL_SCALAR;   // 16-B rows need even nx
    }
    *k = want;
    return DEFF_OK;
}

extern "C" int deff_get_kernel(const deff_ctx *c, int *k)
try {
    if (!c || !k) return fail(DEFF_EINVAL, "NULL argument");
    return resolve_kernel(c, k);
}
DEFF_API_CATCH

extern "C" int deff_set_tuning(deff_ctx *c, const char *key, int value)
try {
    if (!c || !key) return fail(DEFF_EINVAL, "NULL argument");
    if (value < 0) return fail(DEFF_EINVAL, "tuning value must be >= 0");
    if (!strcmp(key, "rows_explicit")) c->rows_explicit = value;
    else if (!strcmp(key, "rows_matfree")) c->rows_matfree = value;
    else if (!strcmp(key, "wg_matfree")) c->wg_matfree = (value + 7) / 8 * 8;
    else if (!strcmp(key, "nt_explicit")) c->nt_explicit = value ? 1 : 0;
    else if (!strcmp(key, "serpentine")) c->serpentine = value ? 1 : 0;
    else if (!strcmp(key, "tb_T")) c->tb_T = value;
    else if (!strcmp(key, "tb_LY")) c->tb_LY = value;
    else if (!strcmp(key, "dict")) c->dict_enabled = value ? 1 : 0;
    else if (!strcmp(key, "tb_xmajor")) c->tb_xmajor = value ? 1 : 0;
    else if (!strcmp(key, "tb_wall_halo")) c->tb_wall_halo = value > 2 ? 2 : value;
    else if (!strcmp(key, "tb_wg")) c->tb_wg = (value + 7) / 8 * 8;
    else return fail(DEFF_EINVAL, "unknown tuning key '%s'", key);
    return DEFF_OK;
}
DEFF_API_CATCH

// What the last plan of the temporally blocked kernel chose (0 before any sweep ran on it).
extern "C" int deff_get_plan(deff_ctx *c, const char *key, int *value)
try {
    if (!c || !key || !value) return fail(DEFF_EINVAL, "NULL argument");
    if (!strcmp(key, "tb_T")) *value = c->plan_T;
    else if (!strcmp(key, "tb_LY")) *value = c->plan_LY;
    else if (!strcmp(key, "tb_strips")) *value = c->plan_ntx;
    else if (!strcmp(key, "tb_chunks_per_image")) *value = c->plan_cpi;
    else if (!strcmp(key, "tb_blocks")) *value = c->plan_blocks;
    else return fail(DEFF_EINVAL, "unknown plan key '%s'", key);
    return DEFF_OK;
}
DEFF_API_CATCH

// ------------------------------------------------------------ image -------

static int image_shape(deff_ctx *c, int W, int H, int ampX, int ampY)
{
    if (W < 1 || H < 1 || ampX < 1 || ampY < 1)            // cuh:1901-1904
        return fail(DEFF_EINVAL, "image %dx%d / mesh amplification %dx%d invalid", W, H, ampX, ampY);
    if ((long long)W * ampX != c->nx || (long long)H * ampY != c->ny)
        return fail(DEFF_EINVAL, "image %dx%d x amp %dx%d does not match mesh %dx%d", W, H, ampX, ampY,
                    c->nx, c->ny);
    if (c->pix && (c->W != W || c->H != H)) { HIP_TRY(hipFree(c->pix)); c->pix = nullptr; }
    TRY(dev_alloc(&c->pix, (size_t)W * H * c->nimg));
    c->W = W; c->H = H; c->ampX = ampX; c->ampY = ampY;
    return DEFF_OK;
}

extern "C" int deff_set_image(deff_ctx *c, const uint8_t *pix, int W, int H, int ampX, int ampY)
try {
    if (!c || !pix) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    TRY(image_shape(c, W, H, ampX, ampY));
    HIP_TRY(hipMemcpyAsync(c->pix, pix, (size_t)W * H * c->nimg, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_image = true;
    c->have_matfree = false;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_synth_image(deff_ctx *c, uint64_t seed, uint64_t img)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    TRY(use_device(c));
    TRY(image_shape(c, c->nx, c->ny, 1, 1));
    // stacked rows continue the per-pixel key, so a batch holds images img, img+1, ...
    hipLaunchKernelGGL(k_synth_mask, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->pix, c->nx, c->rows,
                       seed, img);
    HIP_TRY(hipGetLastError());
    c->have_image = true;
    c->have_matfree = false;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_get_image(deff_ctx *c, uint8_t *pix)
try {
    if (!c || !pix) return fail(DEFF_EINVAL, "NULL argument");
    if (!c->have_image) return fail(DEFF_ESTATE, "no image set");
    TRY(use_device(c));
    HIP_TRY(hipMemcpyAsync(pix, c->pix, (size_t)c->W * c->H * c->nimg, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DEFF_OK;
}
DEFF_API_CATCH

// --------------------------------------------------------- assembly -------

// Row dictionary of the native 2-phase system, enumerated a priori: for every position class and
// every own/W/E/S/N phase pattern the matrix row fvm_row() produces -- built on the host with the
// same routine the device assembly uses, so it holds the very same doubles.  Row 0 = zeros.
static void build_lut_rows(deff_ctx *c, double Ds, double Df, double CL, double CR)
{
    c->lut_nrows = LUT_NATIVE_ROWS;
    c->lut_rows.assign((size_t)c->lut_nrows * 6, 0.0);
    for (int ycls = 0; ycls < 3; ++ycls)
        for (int xcls = 0; xcls < 3; ++xcls)
            for (int code = 0; code < 32; ++code) {
                auto D = [&](int bit) { return ((code >> bit) & 1) ? Ds : Df; };
                const FvmRow r = fvm_row(D(0), D(1), D(2), D(3), D(4), xcls, ycls, c->dx, c->dy, CL, CR);
                double *row = &c->lut_rows[(size_t)(1 + (ycls * 3 + xcls) * 32 + code) * 6];
                row[0] = r.a0; row[1] = r.aW; row[2] = r.aE; row[3] = r.aS; row[4] = r.aN; row[5] = r.b;
            }
    c->lut_allb = false;
    c->lut_omega = NAN;
}

// Device tables for a given omega: plane 0 holds c0 = omega / A0 (the reference divides w by
// A[p*5+0] first, cuh:89), the other planes the links and b.
static int upload_lut(deff_ctx *c, double omega)
{
    if (c->lut_omega == omega) return DEFF_OK;
    std::vector<double> t(LUT_DOUBLES, 0.0);
    bool guard = false;
    for (int k = 1; k < c->lut_nrows; ++k) {
        const double *row = &c->lut_rows[(size_t)k * 6];
        const double c0 = omega / row[0];
        if (!std::isfinite(c0)) guard = true;
        t[k] = c0;
        for (int pl = 1; pl < LUT_PLANES; ++pl) t[(size_t)pl * LUT_PLANE_STRIDE + k] = row[pl];
    }
    TRY(dev_alloc(&c->lut, (size_t)LUT_DOUBLES));
    HIP_TRY(hipMemcpyAsync(c->lut, t.data(), sizeof(double) * LUT_DOUBLES, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // `t` goes out of scope
    // With every c0 finite the field stays finite, a zero link then contributes exactly +-0 and the
    // reference's `A != 0` test (cuh:77) cannot change a bit; otherwise (a phase that cannot
    // diffuse made a singular row) the guarded kernels keep its skip semantics.
    c->lut_guard = guard;
    c->lut_omega = omega;
    return DEFF_OK;
}

extern "C" int deff_assemble_2phase(deff_ctx *c, double Ds, double Df, double CL, double CR)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    if (!c->have_image) return fail(DEFF_ESTATE, "deff_assemble_2phase needs an image");
    TRY(use_device(c));
    TRY(ensure_walls(c));
    c->CL = CL; c->CR = CR;
    hipLaunchKernelGGL(k_wall_D_2phase, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, c->pix, c->W,
                       c->ampX, c->ampY, c->nx, c->ny, c->rows, Df, Ds, c->Dl, c->Dr);
    HIP_TRY(hipGetLastError());
    c->have_walls = true;

    // matrix-free form: 1 byte per cell + lookup tables
    TRY(dev_alloc(&c->code, c->n));
    c->dict_tried = false;
    hipLaunchKernelGGL(k_phase_codes, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->pix, c->W, c->ampX,
                       c->ampY, c->nx, c->ny, c->rows, c->dom_lo, c->mesh_ny, c->code);
    HIP_TRY(hipGetLastError());
    build_lut_rows(c, Ds, Df, CL, CR);
    c->have_matfree = true;

    // the explicit SoA planes are built on demand (explicit_from_image)
    c->Ds = Ds; c->Df = Df;
    c->have_explicit = false;
    return DEFF_OK;
}
DEFF_API_CATCH

// Explicit SoA planes for the native 2-phase system: D from the pixels
// (cuh:1988-2000), then the general assembly.  Only needed when an explicit
// kernel is selected or the coefficients are exported.
static int explicit_from_image(deff_ctx *c)
{
    if (c->have_explicit) return DEFF_OK;
    if (!c->have_matfree) return fail(DEFF_ESTATE, "no system assembled");
    TRY(ensure_explicit(c));
    TRY(ensure_scratch(c, sizeof(double) * c->n));
    double *D = (double *)c->scratch;
    hipLaunchKernelGGL(k_fill_D_2phase, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->pix, c->W, c->ampX,
                       c->ampY, c->nx, c->ny, c->rows, c->Df, c->Ds, D);
    hipLaunchKernelGGL(k_assemble_from_D, dim3(grid_for(c->n)), dim3(256), 0, c->stream, D,
                       (const unsigned int *)nullptr, c->nx, c->ny, c->rows, c->dx, c->dy, c->CL, c->CR,
                       soa_of(c));
    HIP_TRY(hipGetLastError());
    c->have_explicit = true;
    c->c0_omega = NAN;
    return DEFF_OK;
}

// 3-phase system (SingleSim3Phase cuh:1509-1535 / cuh:1558-1586): D from the three pixel
// classes on the device, then DiscretizeMatrix2D_ImpSolid with the caller's Grid (the output of
// deff_flood_fill on `pixel > 200`), or plain DiscretizeMatrix2D when Grid is NULL.  Explicit
// coefficient planes: identity rows and zero-diffusivity links need the guarded general kernel.
extern "C" int deff_assemble_3phase(deff_ctx *c, double Ds, double Df, double Dg, const unsigned int *Grid,
                                    double CL, double CR)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    if (!c->have_image) return fail(DEFF_ESTATE, "deff_assemble_3phase needs an image");
    TRY(use_device(c));
    TRY(ensure_explicit(c));
    TRY(ensure_walls(c));
    const size_t bytes = sizeof(double) * c->n + (Grid ? sizeof(unsigned int) * c->n : 0);
    TRY(ensure_scratch(c, bytes));
    double *dD = (double *)c->scratch;
    unsigned int *dG = Grid ? (unsigned int *)((char *)c->scratch + sizeof(double) * c->n) : nullptr;
    if (Grid) HIP_TRY(hipMemcpyAsync(dG, Grid, sizeof(unsigned int) * c->n, hipMemcpyHostToDevice, c->stream));
    c->CL = CL; c->CR = CR;
    hipLaunchKernelGGL(k_fill_D_3phase, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->pix, c->W, c->ampX,
                       c->ampY, c->nx, c->ny, c->rows, Df, Ds, Dg, dD);
    hipLaunchKernelGGL(k_assemble_from_D, dim3(grid_for(c->n)), dim3(256), 0, c->stream, dD, dG, c->nx,
                       c->ny, c->rows, c->dx, c->dy, CL, CR, soa_of(c));
    hipLaunchKernelGGL(k_wall_D_from_D, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, dD, c->nx,
                       c->rows, c->Dl, c->Dr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));        // Grid may be freed by the caller
    c->have_explicit = true; c->c0_omega = NAN; c->have_walls = true;
    c->have_matfree = false; c->dict_tried = false;
    return DEFF_OK;
}
DEFF_API_CATCH

// FloodFill, cuh:557-713 (host, see flood_fill.hpp).  Grid: 1 = solid on entry; unreachable
// non-solid cells are set to 2; *path_flag receives PathFlag.
extern "C" int deff_flood_fill(unsigned int *Grid, int nx, int ny, int *path_flag)
try {
    if (!Grid || nx < 1 || ny < 1) return fail(DEFF_EINVAL, "bad flood-fill arguments");
    const int flag = flood_fill(Grid, nx, ny);
    if (path_flag) *path_flag = flag;
    return DEFF_OK;
}
DEFF_API_CATCH

// Grayscale JPEG -> bytes, the reference's readImage (cuh:327-345: stbi_load(name,&w,&h,&n,1)).
// *pix is malloc'ed (release with deff_free); *nChannels is the file's component count and is set
// even when the call fails because the image is not single-channel (the reference's check).
extern "C" int deff_load_jpeg_gray(const char *path, uint8_t **pix, int *W, int *H, int *nChannels)
try {
    if (!path || !pix || !W || !H) return fail(DEFF_EINVAL, "NULL argument");
    std::vector<uint8_t> buf;
    std::string err;
    int n = 0;
    *pix = nullptr;
    const bool ok = jpeg::load_gray(path, buf, *W, *H, n, err);
    if (nChannels) *nChannels = n;
    if (!ok) return fail(DEFF_EINVAL, "%s: %s", path, err.c_str());
    *pix = (uint8_t *)malloc(buf.size());
    if (!*pix) return fail(DEFF_ENOMEM, "host allocation failed");
    memcpy(*pix, buf.data(), buf.size());
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" void deff_free(void *p) { free(p); }

// Host -> device in bounded chunks through the scratch buffer.
static const size_t CHUNK_CELLS = (size_t)1 << 22;   // 4 Mi cells

extern "C" int deff_assemble_from_D(deff_ctx *c, const double *D, const unsigned int *Grid, double CL,
                                    double CR)
try {
    if (!c || !D) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    TRY(ensure_explicit(c));
    TRY(ensure_walls(c));
    const size_t bytes = sizeof(double) * c->n + (Grid ? sizeof(unsigned int) * c->n : 0);
    TRY(ensure_scratch(c, bytes));
    double *dD = (double *)c->scratch;
    unsigned int *dG = Grid ? (unsigned int *)((char *)c->scratch + sizeof(double) * c->n) : nullptr;
    HIP_TRY(hipMemcpyAsync(dD, D, sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
    if (Grid) HIP_TRY(hipMemcpyAsync(dG, Grid, sizeof(unsigned int) * c->n, hipMemcpyHostToDevice, c->stream));
    c->CL = CL; c->CR = CR;
    hipLaunchKernelGGL(k_assemble_from_D, dim3(grid_for(c->n)), dim3(256), 0, c->stream, dD, dG, c->nx,
                       c->ny, c->rows, c->dx, c->dy, CL, CR, soa_of(c));
    hipLaunchKernelGGL(k_wall_D_from_D, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, dD, c->nx,
                       c->rows, c->Dl, c->Dr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_explicit = true; c->c0_omega = NAN; c->have_walls = true;
    c->have_matfree = false; c->dict_tried = false;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_set_system(deff_ctx *c, const double *A, const double *b, const double *D, double CL,
                               double CR)
try {
    if (!c || !A || !b) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    TRY(ensure_explicit(c));
    TRY(ensure_scratch(c, sizeof(double) * 5 * CHUNK_CELLS));
    for (size_t first = 0; first < c->n; first += CHUNK_CELLS) {
        const size_t cnt = (c->n - first < CHUNK_CELLS) ? c->n - first : CHUNK_CELLS;
        HIP_TRY(hipMemcpyAsync(c->scratch, A + first * 5, sizeof(double) * 5 * cnt, hipMemcpyHostToDevice,
                               c->stream));
        hipLaunchKernelGGL(k_import_aos, dim3(grid_for(cnt)), dim3(256), 0, c->stream,
                           (const double *)c->scratch, first, cnt, soa_of(c));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    HIP_TRY(hipMemcpyAsync(c->b, b, sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
    c->CL = CL; c->CR = CR;
    c->have_walls = false;
    if (D) {
        TRY(ensure_walls(c));
        // only the first and last column of D are ever read (cuh:1256-1257)
        for (int i = 0; i < c->rows; ++i) {
            c->mf_host[i] = D[(size_t)i * c->nx];
            c->mf_host[c->rows + i] = D[(size_t)(i + 1) * c->nx - 1];
        }
        HIP_TRY(hipMemcpyAsync(c->Dl, c->mf_host, sizeof(double) * c->rows, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->Dr, c->mf_host + c->rows, sizeof(double) * c->rows, hipMemcpyHostToDevice,
                               c->stream));
        c->have_walls = true;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_explicit = true; c->c0_omega = NAN;
    c->have_matfree = false; c->dict_tried = false;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_get_system(deff_ctx *c, double *A, double *b)
try {
    if (!c || !A || !b) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    if (c->have_explicit) {
        TRY(ensure_scratch(c, sizeof(double) * 5 * CHUNK_CELLS));
        for (size_t first = 0; first < c->n; first += CHUNK_CELLS) {
            const size_t cnt = (c->n - first < CHUNK_CELLS) ? c->n - first : CHUNK_CELLS;
            hipLaunchKernelGGL(k_export_aos, dim3(grid_for(cnt)), dim3(256), 0, c->stream, (double *)c->scratch,
                               first, cnt, soa_of(c));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(A + first * 5, c->scratch, sizeof(double) * 5 * cnt, hipMemcpyDeviceToHost,
                                   c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        HIP_TRY(hipMemcpyAsync(b, c->b, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return DEFF_OK;
    }
    if (c->have_matfree) {
        // expand the codes through the row dictionary (what the matrix-free kernels "see")
        std::vector<uint16_t> code(c->n);
        HIP_TRY(hipMemcpyAsync(code.data(), c->code, sizeof(uint16_t) * c->n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (size_t p = 0; p < c->n; ++p) {
            const double *row = &c->lut_rows[(size_t)(code[p] >> 3) * 6];
            for (int k = 0; k < 5; ++k) A[p * 5 + k] = row[k];
            b[p] = row[5];
        }
        return DEFF_OK;
    }
    return fail(DEFF_ESTATE, "no system assembled");
}
DEFF_API_CATCH

// ------------------------------------------------------------ field -------

extern "C" int deff_init_linear(deff_ctx *c, double CL, double CR)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    TRY(use_device(c));
    hipLaunchKernelGGL(k_init_linear, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->x[c->cur], c->nx,
                       c->rows, CL, CR);
    HIP_TRY(hipGetLastError());
    c->have_field = true;
    reset_batch_state(c);
    return DEFF_OK;
}
DEFF_API_CATCH

// After a batch solve the images that converged earlier sit frozen in whichever ping-pong
// buffer was current at that moment; bring every image's newest field into x[cur].
static int consolidate(deff_ctx *c)
{
    if (!c->masked) return DEFF_OK;                  // nothing frozen: every image is current in x[cur]
    for (int k = 0; k < c->nimg; ++k)
        if (c->buf_of[k] != (uint8_t)c->cur) {
            const size_t off = (size_t)k * c->n_img;
            HIP_TRY(hipMemcpyAsync(c->x[c->cur] + off, c->x[c->buf_of[k]] + off, sizeof(double) * c->n_img,
                                   hipMemcpyDeviceToDevice, c->stream));
            c->buf_of[k] = (uint8_t)c->cur;
        }
    c->active_h.assign(c->nimg, 1);
    c->masked = false;
    return DEFF_OK;
}

extern "C" int deff_set_field(deff_ctx *c, const double *x)
try {
    if (!c || !x) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    HIP_TRY(hipMemcpyAsync(c->x[c->cur], x, sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_field = true;
    reset_batch_state(c);
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_get_field(deff_ctx *c, double *x)
try {
    if (!c || !x) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    TRY(consolidate(c));
    HIP_TRY(hipMemcpyAsync(x, c->x[c->cur], sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_device_field(deff_ctx *c, void **d_x, size_t *pitch)
try {
    if (!c || !d_x) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    TRY(consolidate(c));
    *d_x = c->x[c->cur];
    if (pitch) *pitch = sizeof(double) * c->nx;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_synchronize(deff_ctx *c)
try {
    if (!c) return fail(DEFF_EINVAL, "ctx is NULL");
    TRY(use_device(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DEFF_OK;
}
DEFF_API_CATCH

// ----------------------------------------------------------- sweeps -------

struct SweepPlan {
    int kernel = 0;
    double omw = 0;
    int rows = 0, cpi = 0, gx = 0, gy = 0, blocks = 0;   // single-sweep kernels (cpi: row tiles per image)
    // temporally blocked kernel
    int T = 0, CPL = 2, LY = 0, tcpi = 0, ntx = 0, tgx = 0, tgy = 0, tblocks = 0;
    int shift = 0;                                        // column shift of the strips (0: no halo outside the walls)
    int T_override = 0;                                   // slab mode plans a T = 1 pass for remainders
    bool guard = false;
};

// Workgroups of the temporally blocked kernel that are resident at once on this device.
template <int T, int CPL, bool G>
static int tb_occ(int *per_cu)
{
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_sweep_matfree_tb<T, CPL, G>, 256, 0));
    return DEFF_OK;
}

// Only 2 cells per lane are instantiated: 4 per lane (twice the work per wave, 244 VGPRs,
// 2 waves per SIMD) measured 20 % slower at 4096^2 -- the kernel needs the wave-level
// parallelism more than it needs the smaller strip overlap.
#define TB_DISPATCH(T_, CPL_, G_, CALL)                                                     \
    do {                                                                                    \
        const int key_ = (T_) * 10 + ((G_) ? 1 : 0);                                        \
        switch (key_) {                                                                     \
        case 10: { CALL(1, 2, false); } break; case 11: { CALL(1, 2, true); } break;       \
        case 20: { CALL(2, 2, false); } break; case 21: { CALL(2, 2, true); } break;       \
        case 40: { CALL(4, 2, false); } break; case 41: { CALL(4, 2, true); } break;       \
        case 60: { CALL(6, 2, false); } break; case 61: { CALL(6, 2, true); } break;       \
        default: { CALL(8, 2, false); } break; case 81: { CALL(8, 2, true); } break;       \
        }                                                                                   \
    } while (0)

static int tb_resident_blocks(const deff_ctx *c, int T, int CPL, bool guard, int *resident)
{
    int per_cu = 0, cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
#define OCC_CALL(T_, C_, G_) TRY((tb_occ<T_, C_, G_>(&per_cu)))
    TB_DISPATCH(T, CPL, guard, OCC_CALL);
#undef OCC_CALL
    if (per_cu < 1) per_cu = 1;
    *resident = per_cu * cus;
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

static int default_tb_T(const deff_ctx *c)
{
    if (c->n < ((size_t)1 << 22)) return 4;
    return (c->nimg == 1 && !c->slab && c->n >= ((size_t)1 << 24)) ? 8 : 6;
}

// Harvest the row dictionary of the explicit system (kernels_dict.hpp).  On success the context
// also has a matrix-free form (codes + tables); when the system has too many distinct rows it
// simply keeps running on the explicit kernels.
static int try_dict(deff_ctx *c)
{
    c->dict_tried = true;
    if (!c->have_explicit || (c->nx & 1)) return DEFF_OK;
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
    hipLaunchKernelGGL(k_dict_encode, dim3(grid_for(c->n, 4096)), dim3(256), 0, c->stream, planes, c->n, c->nx, t,
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
    return DEFF_OK;
}

static int plan_sweeps(deff_ctx *c, double omega, SweepPlan *pl)
{
    if (!c->have_field) return fail(DEFF_ESTATE, "no field: call deff_init_linear() or deff_set_field()");
    // an explicit system (host-assembled, 3-phase, ImpSolid) with few distinct rows also runs matrix-free
    if (!c->have_matfree && c->have_explicit && !c->dict_tried && c->dict_enabled &&
        (c->kernel == DEFF_KERNEL_AUTO || c->kernel == DEFF_KERNEL_MATFREE || c->kernel == DEFF_KERNEL_MATFREE_TB))
        TRY(try_dict(c));
    TRY(resolve_kernel(c, &pl->kernel));
    pl->omw = 1.0 - omega;                              // cuh:89 evaluates (1.0 - w) in double
    if (pl->kernel == DEFF_KERNEL_MATFREE || pl->kernel == DEFF_KERNEL_MATFREE_TB) {
        TRY(upload_lut(c, omega));
        if (pl->kernel == DEFF_KERNEL_MATFREE_TB) {
            // sweeps per pass (measured, G cells*iter/s: 4096^2 T=4 926, T=6 1063, T=8 1106; stacks of
            // 16 x 1024^2 peak at T=6; 1024^2 alone at T=4)
            int T = pl->T_override ? pl->T_override : (c->tb_T ? c->tb_T : default_tb_T(c));
            T = T >= 8 ? 8 : T >= 6 ? 6 : T >= 4 ? 4 : T >= 2 ? 2 : 1;
            pl->T = T;
            pl->CPL = 2;
            // strips of 128 columns overlapping by 2*HW; a mesh wall needs no halo (kernels_tb.hpp)
            const int hw = (T + 1) & ~1, wout = 64 * pl->CPL - 2 * hw;
            // placement A: every strip carries its halo, also outside the first column; placement B:
            // no halo outside a wall (kernels_tb.hpp).  B needs fewer strips for narrow images
            // (a 128-column image is ONE strip: 2x on dataset batches); where the counts tie, A measured
            // equal or up to 5 % faster in one process (T = 8 at 4096^2), so B is used only when it wins.
            const int ntx_a = (c->nx + wout - 1) / wout;
            const int ntx_b = c->nx <= 64 * pl->CPL ? 1 : (c->nx - 64 * pl->CPL + wout - 1) / wout + 1;
            const bool use_b = c->tb_wall_halo == 0 ? true : (c->tb_wall_halo == 1 ? false : ntx_b < ntx_a);
            pl->shift = use_b ? 0 : hw;
            pl->ntx = use_b ? ntx_b : ntx_a;
            // Rows per chunk.  Workgroups are persistent, so a pass takes `rounds` tiles per wave slot
            // (one round = as many wave tiles as are resident at once), and a tile costs its LY rows
            // + T steps that drain the pipeline + T rows of halo above it unless it starts at the top
            // wall of its image + a fixed start-up (first loads, measured ~8 row steps).  Pick the
            // chunks per image minimising rounds x tile cost; for k rounds only the largest chunk
            // count that fits matters.  (Stacks of small images: 3 072 x 128^2 as whole-image tiles
            // 1 266 G cells*iter/s against 1 107 G for the 4 x 32-row tiles a halo-blind model picks.)
            int resident = c->tb_wg;
            if (!resident) TRY(tb_resident_blocks(c, T, pl->CPL, c->lut_guard, &resident));
            int LY = c->tb_LY;
            if (!LY) {
                long best_cost = -1;
                const bool top_wall = c->own_lo == 0;              // not a slab with rows above it
                for (int k = 1; k <= 8; ++k) {
                    const int cpi_max = (int)(((long)k * resident * 4) / ((long)pl->ntx * c->nimg));
                    if (cpi_max < 1) continue;
                    int ly = (c->own_h + cpi_max - 1) / cpi_max;
                    // chunks shorter than the pipeline is deep lose more to fill/drain than the model
                    // says (1024^2, T=4: 3-row chunks 254 G, 4..6-row chunks 295 G cells*iter/s)
                    if (ly < T) ly = T;
                    const int cpi = (c->own_h + ly - 1) / ly;
                    const long cost = (long)k * (ly + T + ((cpi > 1 || !top_wall) ? T : 0) + 8);
                    if (best_cost < 0 || cost < best_cost) { best_cost = cost; LY = ly; }
                }
                if (!LY) LY = c->own_h;
            }
            if (LY > c->own_h) LY = c->own_h;
            pl->LY = LY;
            pl->tcpi = (c->own_h + LY - 1) / LY;
            pl->tgy = pl->tcpi * c->nimg;
            pl->tgx = (int)(((long)pl->ntx * pl->tgy + 3) / 4);       // workgroup tiles (4 wave tiles each)
            const unsigned total = (unsigned)pl->tgx;
            pl->tblocks = (int)(((total + 7u) / 8u) * 8u);
            if (pl->tblocks > resident) pl->tblocks = resident >= 8 ? resident / 8 * 8 : 8;
            c->plan_T = pl->T; c->plan_LY = pl->LY; c->plan_ntx = pl->ntx; c->plan_cpi = pl->tcpi;
            c->plan_blocks = pl->tblocks;
            // the reference's non-zero link test matters only when a phase cannot diffuse
            pl->guard = c->lut_guard;
        }
        const int vec = (c->nx & 1) ? 1 : 2;
        tile_grid(c, 256 * vec, pick_R(c->rows_matfree, c->n >= ((size_t)1 << 21) ? 8 : 2), pl);
        // persistent grid: a few workgroups per CU walk the tiles (tables loaded once each)
        const int cap = c->wg_matfree ? c->wg_matfree : 256 * 8;
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
static inline void enqueue_sweep(deff_ctx *c, const SweepPlan &pl)
{
    const double *xin = c->x[c->cur];
    double *xout = c->x[c->cur ^ 1];
    const CoefConst cf{c->c0, c->aW, c->aE, c->aS, c->aN, c->b};
    const int flip = c->serpentine ? c->cur : 0;
    const uint8_t *mask = c->masked ? c->active : nullptr;
    switch (pl.kernel) {
    case DEFF_KERNEL_SCALAR:
        if (c->nt_explicit)
            hipLaunchKernelGGL(k_sweep_scalar<true>, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream,
                               cf, xin, xout, c->nx, c->n, c->n_img, mask, pl.omw);
        else
            hipLaunchKernelGGL(k_sweep_scalar<false>, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream,
                               cf, xin, xout, c->nx, c->n, c->n_img, mask, pl.omw);
        break;
    case DEFF_KERNEL_EXPLICIT: {
#define LAUNCH_EXPLICIT(R_)                                                                                  \
    do {                                                                                                    \
        if (c->nt_explicit)                                                                                 \
            hipLaunchKernelGGL((k_sweep_explicit<R_, true>), dim3(pl.blocks), dim3(256), 0, c->stream, cf,  \
                               xin, xout, c->nx, c->ny, c->rows, pl.cpi, mask, pl.gx, pl.gy, flip, pl.omw); \
        else                                                                                                \
            hipLaunchKernelGGL((k_sweep_explicit<R_, false>), dim3(pl.blocks), dim3(256), 0, c->stream, cf, \
                               xin, xout, c->nx, c->ny, c->rows, pl.cpi, mask, pl.gx, pl.gy, flip, pl.omw); \
    } while (0)
        switch (pl.rows) {
        case 1: LAUNCH_EXPLICIT(1); break;
        case 2: LAUNCH_EXPLICIT(2); break;
        case 4: LAUNCH_EXPLICIT(4); break;
        default: LAUNCH_EXPLICIT(8); break;
        }
#undef LAUNCH_EXPLICIT
        break;
    }
    default: {
#define LAUNCH_MATFREE(V_, R_)                                                                               \
    hipLaunchKernelGGL((k_sweep_matfree<V_, R_>), dim3(pl.blocks), dim3(256), 0, c->stream, c->lut, c->code, \
                       xin, xout, c->nx, c->ny, c->rows, pl.cpi, mask, pl.gx, pl.gy, flip, c->lut_nrows, pl.omw)
        if (c->nx & 1) {
            switch (pl.rows) {
            case 1: LAUNCH_MATFREE(1, 1); break;
            case 2: LAUNCH_MATFREE(1, 2); break;
            case 4: LAUNCH_MATFREE(1, 4); break;
            default: LAUNCH_MATFREE(1, 8); break;
            }
        } else {
            switch (pl.rows) {
            case 1: LAUNCH_MATFREE(2, 1); break;
            case 2: LAUNCH_MATFREE(2, 2); break;
            case 4: LAUNCH_MATFREE(2, 4); break;
            default: LAUNCH_MATFREE(2, 8); break;
            }
        }
#undef LAUNCH_MATFREE
        break;
    }
    }
    c->cur ^= 1;
}

// One temporally blocked pass: T sweeps, x[cur] -> x[cur^1].
static inline void enqueue_tb_pass(deff_ctx *c, const SweepPlan &pl)
{
    const double *xin = c->x[c->cur];
    double *xout = c->x[c->cur ^ 1];
    const int flip = c->serpentine ? c->cur : 0;
    const uint8_t *mask = c->masked ? c->active : nullptr;
#define LAUNCH_TB(T_, C_, G_)                                                                                  \
    hipLaunchKernelGGL((k_sweep_matfree_tb<T_, C_, G_>), dim3(pl.tblocks), dim3(256), 0, c->stream, c->lut,    \
                       c->code, xin, xout, c->nx, c->mesh_ny, c->ny, c->dom_lo, c->own_lo, c->own_h, pl.tcpi, \
                       mask, pl.LY, pl.ntx, pl.tgx, pl.tgy, flip, c->tb_xmajor, c->lut_allb ? 1 : 0,         \
                       c->lut_nrows, pl.shift, pl.omw, c->tb_stamps)
    TB_DISPATCH(pl.T, pl.CPL, pl.guard, LAUNCH_TB);
#undef LAUNCH_TB
    c->cur ^= 1;
}

// n sweeps: as many T-sweep passes as fit, the rest one at a time.
static inline void enqueue_sweeps(deff_ctx *c, const SweepPlan &pl, int64_t n)
{
    if (pl.kernel == DEFF_KERNEL_MATFREE_TB) {
        while (n >= pl.T) { enqueue_tb_pass(c, pl); n -= pl.T; ++c->last_launches; }
    }
    for (; n > 0; --n) { enqueue_sweep(c, pl); ++c->last_launches; }
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
    enqueue_sweeps(c, pl, nsweeps);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    if (ms) HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return DEFF_OK;
}
DEFF_API_CATCH

// Wall fluxes of the current field (cuh:1256-1257) for every stacked row, brought to the
// pinned host buffer: mf_host[0..rows) left wall, mf_host[rows..2*rows) right wall.
static int flux_rows(deff_ctx *c)
{
    if (!c->have_walls)
        return fail(DEFF_ESTATE, "wall diffusivities unknown: pass D to deff_set_system() or assemble on the device");
    hipLaunchKernelGGL(k_wall_flux, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, c->x[c->cur], c->Dl,
                       c->Dr, c->nx, c->rows, c->dx, c->CL, c->CR, c->mf);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->mf_host, c->mf, sizeof(double) * 2 * c->rows, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DEFF_OK;
}

// Deff of image k from its rows of mf_host, summed in row order like the reference (cuh:1258-1263).
static double deff_of_image(const deff_ctx *c, int k)
{
    const double *L = c->mf_host + (size_t)k * c->ny, *R = c->mf_host + c->rows + (size_t)k * c->ny;
    double Q1 = 0, Q2 = 0;
    for (int j = 0; j < c->ny; ++j) {
        Q1 += L[j];
        Q2 += R[j];
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
    TRY(flux_rows(c));
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
        enqueue_sweeps(c, pl, batch);
        HIP_TRY(hipGetLastError());
        iter += batch;
        for (int k = 0; k < B; ++k)
            if (c->active_h[k]) { iters[k] = iter; c->buf_of[k] = (uint8_t)c->cur; }
        if (do_check) {
            TRY(flux_rows(c));
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
    const size_t npix = (size_t)c->W * c->H;
    uint8_t *dpix = c->pix + (size_t)slot * npix;
    uint16_t *dcode = c->code + (size_t)slot * c->n_img;
    HIP_TRY(hipMemcpyAsync(dpix, pix_host, npix, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));        // pix_host is the caller's scratch
    hipLaunchKernelGGL(k_phase_codes, dim3(grid_for(c->n_img)), dim3(256), 0, c->stream, dpix, c->W, c->ampX, c->ampY,
                       c->nx, c->ny, c->ny, 0, c->ny, dcode);
    hipLaunchKernelGGL(k_wall_D_2phase, dim3((c->ny + 255) / 256), dim3(256), 0, c->stream, dpix, c->W, c->ampX, c->ampY,
                       c->nx, c->ny, c->ny, c->Df, c->Ds, c->Dl + (size_t)slot * c->ny, c->Dr + (size_t)slot * c->ny);
    hipLaunchKernelGGL(k_init_linear, dim3(grid_for(c->n_img)), dim3(256), 0, c->stream,
                       c->x[c->cur] + (size_t)slot * c->n_img, c->nx, c->ny, c->CL, c->CR);
    HIP_TRY(hipGetLastError());
    c->buf_of[slot] = (uint8_t)c->cur;
    return DEFF_OK;
}

static int stream_push_mask(deff_ctx *c, int n_active)
{
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
    HIP_TRY(hipMemcpyAsync(x, c->x[(c->masked || c->in_stream) ? c->buf_of[slot] : c->cur] + (size_t)slot * c->n_img,
                           sizeof(double) * c->n_img, hipMemcpyDeviceToHost, c->stream));
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
    build_lut_rows(c, Ds, Df, CL, CR);
    c->have_image = true; c->have_walls = true; c->have_matfree = true; c->have_explicit = false;
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
        enqueue_sweeps(c, pl, nsw);
        HIP_TRY(hipGetLastError());
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
        TRY(flux_rows(c));
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
                HIP_TRY(hipEventRecord(c->ev1, c->stream));
                HIP_TRY(hipEventSynchronize(c->ev1));
                float ms2 = 0;
                HIP_TRY(hipEventElapsedTime(&ms2, c->ev0, c->ev1));
                for (int k = 0; k < B; ++k)
                    if (S[k].live && S[k].iters >= max_iter) retire(k, ms2);   // MAX_ITER reached between checks, cuh:1232
            }
        }
        TRY(refill());                                 // newcomers start with the sweep that precedes the next check
    }
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
    const int n = pl.ntx * pl.tgy;
    *ntiles = n;
    if (!out) return DEFF_OK;
    HIP_TRY(hipMalloc((void **)&c->tb_stamps, sizeof(unsigned long long) * 2 * n));
    HIP_TRY(hipMemsetAsync(c->tb_stamps, 0, sizeof(unsigned long long) * 2 * n, c->stream));
    enqueue_tb_pass(c, pl);
    hipError_t e = hipMemcpyAsync(out, c->tb_stamps, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(c->tb_stamps);
    c->tb_stamps = nullptr;
    if (e != hipSuccess) return fail(DEFF_EHIP, "stamp readback failed: %s", hipGetErrorString(e));
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
            int T = c->tb_T ? c->tb_T : default_tb_T(c);
            *sweeps_per_pass = T >= 8 ? 8 : T >= 6 ? 6 : T >= 4 ? 4 : 2;
        }
    }
    return DEFF_OK;
}
DEFF_API_CATCH

// ====================================================================== row slabs ==
//
// One image split into N contiguous row slabs, one per GPU (BASELINE config #4).
// The reference has nothing like it (cudaSetDevice(0), cuh:908).  Each slab is an ordinary
// context whose arrays carry SLAB_HALO extra rows above and below its own rows; a temporally
// blocked pass of T <= SLAB_HALO sweeps needs exactly T valid halo rows, so ONE exchange per
// pass (not per sweep) refreshes them: every slab sends its first and last SLAB_HALO own rows
// to its neighbours.  No arithmetic changes, so the assembled field is bit-identical to the
// one-GPU field, and the wall fluxes are summed on the host in global row order, so Deff and the
// stopping decision are too.
//
// This group drives all slabs from one host thread (one process, N devices; copies between
// devices are hipMemcpyPeerAsync over xGMI, ordered by events) -- which is also what lets the
// whole path be exercised with N slabs on a single GPU.  The process-per-GPU variant only swaps
// the transport (RCCL send/recv of the same row blocks + an all-gather of the fluxes).

static const int SLAB_HALO = 8;
extern "C" int deff_slab_group_destroy(deff_slab_group *g);

struct deff_slab_group {
    int n = 0, nx = 0, NY = 0;
    std::vector<deff_ctx *> ctx;
    std::vector<int> g0, own;                 // first global row and row count of every slab
    std::vector<hipEvent_t> done;             // "pass finished" per slab
    std::vector<double> mfl, mfr;             // global wall fluxes of the last check
};

static int slab_create_ctx(int device, int nx, int NY, int g0, int own, deff_ctx **out)
{
    const int rows = own + 2 * SLAB_HALO;
    TRY(deff_create_batch(device, nx, rows, 1, out));
    deff_ctx *c = *out;
    c->slab = true;
    c->halo = SLAB_HALO;
    c->dom_lo = SLAB_HALO - g0;               // array row of mesh row 0
    c->mesh_ny = NY;
    c->own_lo = SLAB_HALO;
    c->own_h = own;
    c->dy = 1.0 / NY;                         // the mesh is the whole image, cuh:1911
    c->kernel = DEFF_KERNEL_MATFREE_TB;
    return DEFF_OK;
}

extern "C" int deff_slab_group_create(int nslabs, const int *devices, int nx, int NY, deff_slab_group **out)
try {
    if (!out || nslabs < 1) return fail(DEFF_EINVAL, "bad slab group arguments");
    *out = nullptr;
    if (nx < 2 || (nx & 1)) return fail(DEFF_EINVAL, "row-slab mode needs an even nx >= 2 (got %d)", nx);
    if (NY / nslabs < SLAB_HALO) return fail(DEFF_EINVAL, "%d rows over %d slabs: fewer than %d rows per slab", NY, nslabs, SLAB_HALO);
    deff_slab_group *g = new (std::nothrow) deff_slab_group();
    if (!g) return fail(DEFF_ENOMEM, "host allocation failed");
    g->n = nslabs; g->nx = nx; g->NY = NY;
    g->mfl.assign(NY, 0.0); g->mfr.assign(NY, 0.0);
    int rc = DEFF_OK;
    for (int r = 0; r < nslabs && rc == DEFF_OK; ++r) {
        const int a = (int)((long long)NY * r / nslabs), b = (int)((long long)NY * (r + 1) / nslabs);
        deff_ctx *c = nullptr;
        rc = slab_create_ctx(devices ? devices[r] : 0, nx, NY, a, b - a, &c);
        if (rc != DEFF_OK) break;
        g->ctx.push_back(c); g->g0.push_back(a); g->own.push_back(b - a);
        hipEvent_t ev = nullptr;
        if (hipSetDevice(c->device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            rc = fail(DEFF_EHIP, "event creation failed");
        g->done.push_back(ev);
    }
    if (rc != DEFF_OK) { deff_slab_group_destroy(g); return rc; }
    // direct xGMI copies between neighbouring slabs' devices (staged through the host otherwise)
    for (int r = 0; r + 1 < nslabs; ++r) {
        const int a = g->ctx[r]->device, b = g->ctx[r + 1]->device;
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
            (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0);
            (void)hipSetDevice(b); (void)hipDeviceEnablePeerAccess(a, 0);
            (void)hipGetLastError();                 // "already enabled" is fine
        }
    }
    *out = g;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_destroy(deff_slab_group *g)
try {
    if (!g) return DEFF_OK;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        if (r < g->done.size() && g->done[r]) { (void)hipSetDevice(g->ctx[r]->device); (void)hipEventDestroy(g->done[r]); }
        deff_destroy(g->ctx[r]);
    }
    delete g;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_layout(const deff_slab_group *g, int *first_row, int *row_count)
try {
    if (!g) return fail(DEFF_EINVAL, "group is NULL");
    for (int r = 0; r < g->n; ++r) {
        if (first_row) first_row[r] = g->g0[r];
        if (row_count) row_count[r] = g->own[r];
    }
    return DEFF_OK;
}
DEFF_API_CATCH

// Array rows [lo, hi) of slab r as mesh rows, clipped to the mesh.
static void slab_window(const deff_slab_group *g, int r, int *mesh_first, int *array_first, int *count)
{
    const deff_ctx *c = g->ctx[r];
    int a = -c->dom_lo, b = a + c->rows;        // mesh rows covered by the array
    int ar = 0;
    if (a < 0) { ar = -a; a = 0; }
    if (b > g->NY) b = g->NY;
    *mesh_first = a; *array_first = ar; *count = b - a;
}

// pix: the whole image, NY x nx bytes (mesh amplification is not supported in slab mode)
extern "C" int deff_slab_group_set_image(deff_slab_group *g, const uint8_t *pix)
try {
    if (!g || !pix) return fail(DEFF_EINVAL, "NULL argument");
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        TRY(image_shape(c, c->nx, c->ny, 1, 1));
        HIP_TRY(hipMemsetAsync(c->pix, 0, c->n, c->stream));
        int m0, a0, cnt;
        slab_window(g, r, &m0, &a0, &cnt);
        HIP_TRY(hipMemcpyAsync(c->pix + (size_t)a0 * c->nx, pix + (size_t)m0 * c->nx, (size_t)cnt * c->nx,
                               hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->have_image = true; c->have_matfree = false;
    }
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_synth_image(deff_slab_group *g, uint64_t seed, uint64_t img)
try {
    if (!g) return fail(DEFF_EINVAL, "group is NULL");
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        TRY(image_shape(c, c->nx, c->ny, 1, 1));
        HIP_TRY(hipMemsetAsync(c->pix, 0, c->n, c->stream));
        int m0, a0, cnt;
        slab_window(g, r, &m0, &a0, &cnt);
        // the generator's key is seed*K + img*NY*nx + global cell index: start it at mesh row m0
        const uint64_t base_img_cells = img * (uint64_t)g->NY * (uint64_t)g->nx + (uint64_t)m0 * (uint64_t)g->nx;
        hipLaunchKernelGGL(k_synth_mask_at, dim3(grid_for((size_t)cnt * c->nx)), dim3(256), 0, c->stream,
                           c->pix + (size_t)a0 * c->nx, (size_t)cnt * c->nx, seed, base_img_cells);
        HIP_TRY(hipGetLastError());
        c->have_image = true; c->have_matfree = false;
    }
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_assemble_2phase(deff_slab_group *g, double Ds, double Df, double CL, double CR)
try {
    if (!g) return fail(DEFF_EINVAL, "group is NULL");
    for (int r = 0; r < g->n; ++r) TRY(deff_assemble_2phase(g->ctx[r], Ds, Df, CL, CR));
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_init_linear(deff_slab_group *g, double CL, double CR)
try {
    if (!g) return fail(DEFF_EINVAL, "group is NULL");
    for (int r = 0; r < g->n; ++r) TRY(deff_init_linear(g->ctx[r], CL, CR));     // a function of the column only
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_set_field(deff_slab_group *g, const double *x)
try {
    if (!g || !x) return fail(DEFF_EINVAL, "NULL argument");
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        HIP_TRY(hipMemsetAsync(c->x[c->cur], 0, sizeof(double) * c->n, c->stream));
        int m0, a0, cnt;
        slab_window(g, r, &m0, &a0, &cnt);
        HIP_TRY(hipMemcpyAsync(c->x[c->cur] + (size_t)a0 * c->nx, x + (size_t)m0 * c->nx,
                               sizeof(double) * (size_t)cnt * c->nx, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->have_field = true;
        reset_batch_state(c);
    }
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_get_field(deff_slab_group *g, double *x)
try {
    if (!g || !x) return fail(DEFF_EINVAL, "NULL argument");
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        HIP_TRY(hipMemcpyAsync(x + (size_t)g->g0[r] * c->nx, c->x[c->cur] + (size_t)c->own_lo * c->nx,
                               sizeof(double) * (size_t)g->own[r] * c->nx, hipMemcpyDeviceToHost, c->stream));
    }
    for (int r = 0; r < g->n; ++r) { TRY(use_device(g->ctx[r])); HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream)); }
    return DEFF_OK;
}
DEFF_API_CATCH

// After a pass: every slab's halo rows of the NEW field are stale; refill them from the
// neighbours' own rows.  Copies run on the receiver's stream once the sender's pass is done.
static int slab_exchange(deff_slab_group *g)
{
    const size_t blk = (size_t)SLAB_HALO * g->nx;                      // doubles per halo block
    for (int r = 0; r < g->n; ++r) {
        TRY(use_device(g->ctx[r]));
        HIP_TRY(hipEventRecord(g->done[r], g->ctx[r]->stream));
    }
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        if (r > 0) {                                                   // top halo <- last own rows of slab r-1
            deff_ctx *u = g->ctx[r - 1];
            HIP_TRY(hipStreamWaitEvent(c->stream, g->done[r - 1], 0));
            const double *src = u->x[u->cur] + (size_t)(u->own_lo + u->own_h - SLAB_HALO) * g->nx;
            HIP_TRY(hipMemcpyPeerAsync(c->x[c->cur], c->device, src, u->device, sizeof(double) * blk, c->stream));
        }
        if (r + 1 < g->n) {                                            // bottom halo <- first own rows of slab r+1
            deff_ctx *d = g->ctx[r + 1];
            HIP_TRY(hipStreamWaitEvent(c->stream, g->done[r + 1], 0));
            const double *src = d->x[d->cur] + (size_t)d->own_lo * g->nx;
            HIP_TRY(hipMemcpyPeerAsync(c->x[c->cur] + (size_t)(c->own_lo + c->own_h) * g->nx, c->device, src,
                                       d->device, sizeof(double) * blk, c->stream));
        }
    }
    // a slab must not start its next pass (which overwrites x[cur^1] ... and whose result the
    // neighbours will read) before the neighbours have taken their copies of this one
    for (int r = 0; r < g->n; ++r) {
        TRY(use_device(g->ctx[r]));
        HIP_TRY(hipEventRecord(g->done[r], g->ctx[r]->stream));
    }
    for (int r = 0; r < g->n; ++r) {
        TRY(use_device(g->ctx[r]));
        if (r > 0) HIP_TRY(hipStreamWaitEvent(g->ctx[r]->stream, g->done[r - 1], 0));
        if (r + 1 < g->n) HIP_TRY(hipStreamWaitEvent(g->ctx[r]->stream, g->done[r + 1], 0));
    }
    return DEFF_OK;
}

// n sweeps on every slab: blocked passes of T, remainder as T = 1 passes, one exchange per pass.
static int slab_sweeps(deff_slab_group *g, std::vector<SweepPlan> &plT, std::vector<SweepPlan> &pl1, int64_t n)
{
    const int T = plT[0].T;
    while (n > 0) {
        const bool big = n >= T;
        for (int r = 0; r < g->n; ++r) {
            TRY(use_device(g->ctx[r]));
            enqueue_tb_pass(g->ctx[r], big ? plT[r] : pl1[r]);
            ++g->ctx[r]->last_launches;
        }
        HIP_TRY(hipGetLastError());
        TRY(slab_exchange(g));
        n -= big ? T : 1;
    }
    return DEFF_OK;
}

static int slab_plans(deff_slab_group *g, double omega, std::vector<SweepPlan> &plT, std::vector<SweepPlan> &pl1)
{
    plT.assign(g->n, SweepPlan()); pl1.assign(g->n, SweepPlan());
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        if (c->tb_T > SLAB_HALO) return fail(DEFF_EINVAL, "tb_T exceeds the slab halo depth %d", SLAB_HALO);
        TRY(plan_sweeps(c, omega, &plT[r]));
        if (plT[r].kernel != DEFF_KERNEL_MATFREE_TB) return fail(DEFF_ESTATE, "row-slab mode needs the temporally blocked kernel");
        pl1[r].T_override = 1;
        TRY(plan_sweeps(c, omega, &pl1[r]));
        c->last_launches = 0;
    }
    return DEFF_OK;
}

extern "C" int deff_slab_group_sweeps(deff_slab_group *g, int64_t n, double omega, float *ms)
try {
    if (!g || n < 0) return fail(DEFF_EINVAL, "bad arguments");
    std::vector<SweepPlan> plT, pl1;
    TRY(slab_plans(g, omega, plT, pl1));
    deff_ctx *c0 = g->ctx[0];
    TRY(use_device(c0));
    HIP_TRY(hipEventRecord(c0->ev0, c0->stream));
    TRY(slab_sweeps(g, plT, pl1, n));
    for (int r = 0; r < g->n; ++r) { TRY(use_device(g->ctx[r])); HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream)); }
    TRY(use_device(c0));
    HIP_TRY(hipEventRecord(c0->ev1, c0->stream));
    HIP_TRY(hipEventSynchronize(c0->ev1));
    if (ms) HIP_TRY(hipEventElapsedTime(ms, c0->ev0, c0->ev1));
    return DEFF_OK;
}
DEFF_API_CATCH

// Wall fluxes of every slab's own rows -> the group's global arrays -> Deff (cuh:1252-1263),
// summed in global row order exactly like the one-GPU path.
static int slab_flux(deff_slab_group *g, double *deff_raw)
{
    for (int r = 0; r < g->n; ++r) { TRY(use_device(g->ctx[r])); TRY(flux_rows(g->ctx[r])); }
    for (int r = 0; r < g->n; ++r) {
        const deff_ctx *c = g->ctx[r];
        memcpy(&g->mfl[g->g0[r]], c->mf_host + c->own_lo, sizeof(double) * g->own[r]);
        memcpy(&g->mfr[g->g0[r]], c->mf_host + c->rows + c->own_lo, sizeof(double) * g->own[r]);
    }
    double Q1 = 0, Q2 = 0;
    for (int j = 0; j < g->NY; ++j) { Q1 += g->mfl[j]; Q2 += g->mfr[j]; }
    const deff_ctx *c = g->ctx[0];
    const double qAvg = (Q1 + Q2) / (2.0 * g->NY);
    *deff_raw = qAvg / ((c->CR - c->CL));
    return DEFF_OK;
}

extern "C" int deff_slab_group_flux(deff_slab_group *g, double *deff_raw, double *MFL, double *MFR)
try {
    if (!g || !deff_raw) return fail(DEFF_EINVAL, "NULL argument");
    TRY(slab_flux(g, deff_raw));
    if (MFL) memcpy(MFL, g->mfl.data(), sizeof(double) * g->NY);
    if (MFR) memcpy(MFR, g->mfr.data(), sizeof(double) * g->NY);
    return DEFF_OK;
}
DEFF_API_CATCH

// JacobiGPU's loop (cuh:1232-1290) over the slabs; same stopping rule as deff_solve.
extern "C" int deff_slab_group_solve(deff_slab_group *g, double omega, double tol, int64_t max_iter,
                                     int64_t check_every, deff_result *out, double *MFL, double *MFR)
try {
    if (!g || !out) return fail(DEFF_EINVAL, "NULL argument");
    if (check_every < 1) return fail(DEFF_EINVAL, "check_every must be >= 1");
    std::vector<SweepPlan> plT, pl1;
    TRY(slab_plans(g, omega, plT, pl1));
    deff_ctx *c0 = g->ctx[0];
    int64_t iter = 0, checks = 0;
    double deffNew = 1, deffOld = 5, change = 100.0, conv = 0;      // cuh:1171-1173
    TRY(use_device(c0));
    HIP_TRY(hipEventRecord(c0->ev0, c0->stream));
    while (iter < max_iter && tol < fabs(change)) {                  // cuh:1232
        const int64_t next_check = ((iter + check_every - 1) / check_every) * check_every;
        const bool do_check = next_check < max_iter;
        const int64_t batch = do_check ? next_check - iter + 1 : max_iter - iter;
        TRY(slab_sweeps(g, plT, pl1, batch));
        iter += batch;
        if (do_check) {
            TRY(slab_flux(g, &deffNew));
            change = (deffOld - deffNew) / (deffOld);                // cuh:1265
            deffOld = deffNew;
            conv = change;
            ++checks;
        }
    }
    for (int r = 0; r < g->n; ++r) { TRY(use_device(g->ctx[r])); HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream)); }
    TRY(use_device(c0));
    HIP_TRY(hipEventRecord(c0->ev1, c0->stream));
    HIP_TRY(hipEventSynchronize(c0->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c0->ev0, c0->ev1));
    out->iters = iter; out->checks = checks; out->deff_raw = deffNew; out->conv = conv; out->loop_ms = ms;
    if (MFL) memcpy(MFL, g->mfl.data(), sizeof(double) * g->NY);
    if (MFR) memcpy(MFR, g->mfr.data(), sizeof(double) * g->NY);
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_set_tuning(deff_slab_group *g, const char *key, int value)
try {
    if (!g) return fail(DEFF_EINVAL, "group is NULL");
    for (int r = 0; r < g->n; ++r) TRY(deff_set_tuning(g->ctx[r], key, value));
    return DEFF_OK;
}
DEFF_API_CATCH

// ------------------------------------------------- row slabs, one process per GPU (RCCL) --
//
// Same slab contexts and the same loop as the group above; only the transport differs: the halo
// blocks travel by grouped ncclSend/ncclRecv between neighbouring ranks on the context's stream
// (point-to-point over one xGMI link per neighbour pair; 8 rows x nx doubles, 1 MiB at nx =
// 16384, once per blocked pass), and the per-row wall fluxes are all-gathered so that every rank
// sums them in global row order and takes the same stop/continue decision.

struct deff_slab_rank {
    deff_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, nx = 0, NY = 0, maxown = 0;
    std::vector<int> g0, own;
    double *d_pack = nullptr, *d_all = nullptr;      // [2*maxown], [nranks*2*maxown]
    std::vector<double> h_all, mfl, mfr;
    // host-staged custom transport (deff_slab_rank_create_custom): the same loop, the blocks go
    // through host buffers and the caller's callbacks instead of RCCL
    deff_host_exchange_fn xchg = nullptr;
    deff_host_allgather_fn gather = nullptr;
    void *user = nullptr;
    std::vector<double> h_send_up, h_send_dn, h_recv_up, h_recv_dn, h_pack;
};

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(DEFF_ECOMM, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

extern "C" int deff_rccl_unique_id(char *id128)
try {
    if (!id128) return fail(DEFF_EINVAL, "id buffer is NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_rank_destroy(deff_slab_rank *s)
try {
    if (!s) return DEFF_OK;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    if (s->d_pack) (void)hipFree(s->d_pack);
    if (s->d_all) (void)hipFree(s->d_all);
    if (s->comm) (void)ncclCommDestroy(s->comm);
    deff_destroy(s->ctx);
    delete s;
    return DEFF_OK;
}
DEFF_API_CATCH

static int slab_rank_create_impl(int device, int nx, int NY, int rank, int nranks, const char *id128,
                                 deff_host_exchange_fn xchg, deff_host_allgather_fn gather, void *user,
                                 deff_slab_rank **out);

extern "C" int deff_slab_rank_create(int device, int nx, int NY, int rank, int nranks, const char *id128,
                                     deff_slab_rank **out)
try {
    if (!id128) return fail(DEFF_EINVAL, "RCCL id is NULL");
    return slab_rank_create_impl(device, nx, NY, rank, nranks, id128, nullptr, nullptr, nullptr, out);
}
DEFF_API_CATCH

// Same slabs and loop with a caller-supplied transport: after every pass the two 8-row blocks are
// copied to the host and handed to `exchange`, the fluxes to `allgather` (both collective over the
// ranks).  Slow (host staged) but runs anywhere -- e.g. two processes sharing one GPU under gloo,
// which is how the per-rank loop is tested across real process boundaries.
extern "C" int deff_slab_rank_create_custom(int device, int nx, int NY, int rank, int nranks,
                                            deff_host_exchange_fn exchange, deff_host_allgather_fn allgather,
                                            void *user, deff_slab_rank **out)
try {
    if (!exchange || !allgather) return fail(DEFF_EINVAL, "transport callbacks are NULL");
    return slab_rank_create_impl(device, nx, NY, rank, nranks, nullptr, exchange, allgather, user, out);
}
DEFF_API_CATCH

static int slab_rank_create_impl(int device, int nx, int NY, int rank, int nranks, const char *id128,
                                 deff_host_exchange_fn xchg, deff_host_allgather_fn gather, void *user,
                                 deff_slab_rank **out)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return fail(DEFF_EINVAL, "bad slab rank arguments");
    *out = nullptr;
    if (nx < 2 || (nx & 1)) return fail(DEFF_EINVAL, "row-slab mode needs an even nx >= 2 (got %d)", nx);
    if (NY / nranks < SLAB_HALO) return fail(DEFF_EINVAL, "%d rows over %d ranks: fewer than %d rows per slab", NY, nranks, SLAB_HALO);
    deff_slab_rank *s = new (std::nothrow) deff_slab_rank();
    if (!s) return fail(DEFF_ENOMEM, "host allocation failed");
    s->rank = rank; s->nranks = nranks; s->nx = nx; s->NY = NY;
    for (int r = 0; r < nranks; ++r) {
        const int a = (int)((long long)NY * r / nranks), b = (int)((long long)NY * (r + 1) / nranks);
        s->g0.push_back(a); s->own.push_back(b - a);
        if (b - a > s->maxown) s->maxown = b - a;
    }
    s->mfl.assign(NY, 0.0); s->mfr.assign(NY, 0.0);
    s->h_all.assign((size_t)nranks * 2 * s->maxown, 0.0);
    s->xchg = xchg; s->gather = gather; s->user = user;
    const size_t blk = (size_t)SLAB_HALO * nx;
    if (xchg) {
        s->h_send_up.assign(blk, 0.0); s->h_send_dn.assign(blk, 0.0);
        s->h_recv_up.assign(blk, 0.0); s->h_recv_dn.assign(blk, 0.0);
        s->h_pack.assign((size_t)2 * s->maxown, 0.0);
    }
    int rc = slab_create_ctx(device, nx, NY, s->g0[rank], s->own[rank], &s->ctx);
    if (rc == DEFF_OK) {
        ncclUniqueId id;
        if (id128) memcpy(&id, id128, sizeof id);
        hipError_t he;
        ncclResult_t nr;
        if ((he = hipSetDevice(device)) != hipSuccess) rc = fail(DEFF_EHIP, "hipSetDevice: %s", hipGetErrorString(he));
        else if (id128 && (nr = ncclCommInitRank(&s->comm, nranks, id, rank)) != ncclSuccess)
            rc = fail(DEFF_ECOMM, "ncclCommInitRank: %s", ncclGetErrorString(nr));
        else if ((he = hipMalloc((void **)&s->d_pack, sizeof(double) * 2 * s->maxown)) != hipSuccess ||
                 (he = hipMalloc((void **)&s->d_all, sizeof(double) * 2 * s->maxown * nranks)) != hipSuccess)
            rc = fail(DEFF_ENOMEM, "hipMalloc: %s", hipGetErrorString(he));
        else if ((he = hipMemset(s->d_pack, 0, sizeof(double) * 2 * s->maxown)) != hipSuccess)
            rc = fail(DEFF_EHIP, "hipMemset: %s", hipGetErrorString(he));
    }
    if (rc != DEFF_OK) { deff_slab_rank_destroy(s); return rc; }
    *out = s;
    return DEFF_OK;
}

extern "C" int deff_slab_rank_layout(const deff_slab_rank *s, int *first_row, int *row_count)
try {
    if (!s) return fail(DEFF_EINVAL, "slab is NULL");
    if (first_row) *first_row = s->g0[s->rank];
    if (row_count) *row_count = s->own[s->rank];
    return DEFF_OK;
}
DEFF_API_CATCH

// the context behind the slab, for deff_set_tuning / deff_assemble_2phase / deff_init_linear
extern "C" int deff_slab_rank_context(deff_slab_rank *s, deff_ctx **ctx)
try {
    if (!s || !ctx) return fail(DEFF_EINVAL, "NULL argument");
    *ctx = s->ctx;
    return DEFF_OK;
}
DEFF_API_CATCH

// window = the rows of the whole image this rank's arrays cover (own rows + halo, clipped to the
// mesh): *first_row, *row_count; the image upload below takes exactly those rows.
extern "C" int deff_slab_rank_window(const deff_slab_rank *s, int *first_row, int *row_count)
try {
    if (!s) return fail(DEFF_EINVAL, "slab is NULL");
    const deff_ctx *c = s->ctx;
    int a = -c->dom_lo, b = a + c->rows;
    if (a < 0) a = 0;
    if (b > s->NY) b = s->NY;
    if (first_row) *first_row = a;
    if (row_count) *row_count = b - a;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_rank_set_image_window(deff_slab_rank *s, const uint8_t *pix_window)
try {
    if (!s || !pix_window) return fail(DEFF_EINVAL, "NULL argument");
    deff_ctx *c = s->ctx;
    TRY(use_device(c));
    TRY(image_shape(c, c->nx, c->ny, 1, 1));
    int a = 0, cnt = 0;
    TRY(deff_slab_rank_window(s, &a, &cnt));
    HIP_TRY(hipMemsetAsync(c->pix, 0, c->n, c->stream));
    HIP_TRY(hipMemcpyAsync(c->pix + (size_t)(a + c->dom_lo) * c->nx, pix_window, (size_t)cnt * c->nx,
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_image = true; c->have_matfree = false;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_rank_synth_image(deff_slab_rank *s, uint64_t seed, uint64_t img)
try {
    if (!s) return fail(DEFF_EINVAL, "slab is NULL");
    deff_ctx *c = s->ctx;
    TRY(use_device(c));
    TRY(image_shape(c, c->nx, c->ny, 1, 1));
    int a = 0, cnt = 0;
    TRY(deff_slab_rank_window(s, &a, &cnt));
    HIP_TRY(hipMemsetAsync(c->pix, 0, c->n, c->stream));
    const uint64_t first = img * (uint64_t)s->NY * (uint64_t)s->nx + (uint64_t)a * (uint64_t)s->nx;
    hipLaunchKernelGGL(k_synth_mask_at, dim3(grid_for((size_t)cnt * c->nx)), dim3(256), 0, c->stream,
                       c->pix + (size_t)(a + c->dom_lo) * c->nx, (size_t)cnt * c->nx, seed, first);
    HIP_TRY(hipGetLastError());
    c->have_image = true; c->have_matfree = false;
    return DEFF_OK;
}
DEFF_API_CATCH

// own rows of the current field -> host
extern "C" int deff_slab_rank_get_field(deff_slab_rank *s, double *x_own)
try {
    if (!s || !x_own) return fail(DEFF_EINVAL, "NULL argument");
    deff_ctx *c = s->ctx;
    TRY(use_device(c));
    HIP_TRY(hipMemcpyAsync(x_own, c->x[c->cur] + (size_t)c->own_lo * c->nx, sizeof(double) * (size_t)c->own_h * c->nx,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DEFF_OK;
}
DEFF_API_CATCH

static int rank_exchange(deff_slab_rank *s)
{
    deff_ctx *c = s->ctx;
    const size_t blk = (size_t)SLAB_HALO * s->nx;
    double *x = c->x[c->cur];
    if (s->xchg) {                                               // host-staged custom transport
        const bool up = s->rank > 0, dn = s->rank + 1 < s->nranks;
        double *top_own = x + (size_t)c->own_lo * s->nx, *bot_own = x + (size_t)(c->own_lo + c->own_h - SLAB_HALO) * s->nx;
        if (up) HIP_TRY(hipMemcpyAsync(s->h_send_up.data(), top_own, sizeof(double) * blk, hipMemcpyDeviceToHost, c->stream));
        if (dn) HIP_TRY(hipMemcpyAsync(s->h_send_dn.data(), bot_own, sizeof(double) * blk, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (s->xchg(s->user, up ? s->h_send_up.data() : nullptr, up ? s->h_recv_up.data() : nullptr,
                    dn ? s->h_send_dn.data() : nullptr, dn ? s->h_recv_dn.data() : nullptr, blk) != 0)
            return fail(DEFF_ECOMM, "custom halo exchange failed");
        if (up) HIP_TRY(hipMemcpyAsync(x, s->h_recv_up.data(), sizeof(double) * blk, hipMemcpyHostToDevice, c->stream));
        if (dn) HIP_TRY(hipMemcpyAsync(x + (size_t)(c->own_lo + c->own_h) * s->nx, s->h_recv_dn.data(), sizeof(double) * blk,
                                       hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));                // the host buffers are reused by the next pass
        return DEFF_OK;
    }
    NCCL_TRY(ncclGroupStart());
    if (s->rank > 0) {
        NCCL_TRY(ncclSend(x + (size_t)c->own_lo * s->nx, blk, ncclDouble, s->rank - 1, s->comm, c->stream));
        NCCL_TRY(ncclRecv(x, blk, ncclDouble, s->rank - 1, s->comm, c->stream));
    }
    if (s->rank + 1 < s->nranks) {
        NCCL_TRY(ncclSend(x + (size_t)(c->own_lo + c->own_h - SLAB_HALO) * s->nx, blk, ncclDouble, s->rank + 1, s->comm,
                          c->stream));
        NCCL_TRY(ncclRecv(x + (size_t)(c->own_lo + c->own_h) * s->nx, blk, ncclDouble, s->rank + 1, s->comm, c->stream));
    }
    NCCL_TRY(ncclGroupEnd());
    return DEFF_OK;
}

static int rank_sweeps(deff_slab_rank *s, const SweepPlan &plT, const SweepPlan &pl1, int64_t n)
{
    deff_ctx *c = s->ctx;
    while (n > 0) {
        const bool big = n >= plT.T;
        enqueue_tb_pass(c, big ? plT : pl1);
        ++c->last_launches;
        HIP_TRY(hipGetLastError());
        if (s->nranks > 1) TRY(rank_exchange(s));
        n -= big ? plT.T : 1;
    }
    return DEFF_OK;
}

static int rank_plans(deff_slab_rank *s, double omega, SweepPlan *plT, SweepPlan *pl1)
{
    deff_ctx *c = s->ctx;
    TRY(use_device(c));
    if (c->tb_T > SLAB_HALO) return fail(DEFF_EINVAL, "tb_T exceeds the slab halo depth %d", SLAB_HALO);
    TRY(plan_sweeps(c, omega, plT));
    if (plT->kernel != DEFF_KERNEL_MATFREE_TB) return fail(DEFF_ESTATE, "row-slab mode needs the temporally blocked kernel");
    pl1->T_override = 1;
    TRY(plan_sweeps(c, omega, pl1));
    c->last_launches = 0;
    return DEFF_OK;
}

extern "C" int deff_slab_rank_sweeps(deff_slab_rank *s, int64_t n, double omega, float *ms)
try {
    if (!s || n < 0) return fail(DEFF_EINVAL, "bad arguments");
    SweepPlan plT, pl1;
    TRY(rank_plans(s, omega, &plT, &pl1));
    deff_ctx *c = s->ctx;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    TRY(rank_sweeps(s, plT, pl1, n));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    if (ms) HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return DEFF_OK;
}
DEFF_API_CATCH

__global__ void k_pack_own_flux(const double *__restrict__ mf, int rows, int own_lo, int own_h, int maxown,
                                double *__restrict__ pack)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= own_h) return;
    pack[i] = mf[own_lo + i];
    pack[maxown + i] = mf[rows + own_lo + i];
}

static int rank_flux(deff_slab_rank *s, double *deff_raw)
{
    deff_ctx *c = s->ctx;
    if (!c->have_walls) return fail(DEFF_ESTATE, "wall diffusivities unknown");
    hipLaunchKernelGGL(k_wall_flux, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, c->x[c->cur], c->Dl, c->Dr,
                       c->nx, c->rows, c->dx, c->CL, c->CR, c->mf);
    hipLaunchKernelGGL(k_pack_own_flux, dim3((c->own_h + 255) / 256), dim3(256), 0, c->stream, c->mf, c->rows, c->own_lo,
                       c->own_h, s->maxown, s->d_pack);
    HIP_TRY(hipGetLastError());
    if (s->gather) {
        HIP_TRY(hipMemcpyAsync(s->h_pack.data(), s->d_pack, sizeof(double) * 2 * s->maxown, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (s->gather(s->user, s->h_pack.data(), s->h_all.data(), (size_t)2 * s->maxown) != 0)
            return fail(DEFF_ECOMM, "custom flux all-gather failed");
    } else {
        NCCL_TRY(ncclAllGather(s->d_pack, s->d_all, (size_t)2 * s->maxown, ncclDouble, s->comm, c->stream));
        HIP_TRY(hipMemcpyAsync(s->h_all.data(), s->d_all, sizeof(double) * s->h_all.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (int r = 0; r < s->nranks; ++r) {
        const double *blk = s->h_all.data() + (size_t)r * 2 * s->maxown;
        memcpy(&s->mfl[s->g0[r]], blk, sizeof(double) * s->own[r]);
        memcpy(&s->mfr[s->g0[r]], blk + s->maxown, sizeof(double) * s->own[r]);
    }
    double Q1 = 0, Q2 = 0;
    for (int j = 0; j < s->NY; ++j) { Q1 += s->mfl[j]; Q2 += s->mfr[j]; }      // global row order, cuh:1258-1259
    const double qAvg = (Q1 + Q2) / (2.0 * s->NY);
    *deff_raw = qAvg / ((c->CR - c->CL));
    return DEFF_OK;
}

// Collective over the ranks of the communicator: every rank calls it with the same arguments
// and gets the same result (iters, Deff, conv); MFL/MFR receive the GLOBAL fluxes (NY each).
extern "C" int deff_slab_rank_solve(deff_slab_rank *s, double omega, double tol, int64_t max_iter, int64_t check_every,
                                    deff_result *out, double *MFL, double *MFR)
try {
    if (!s || !out) return fail(DEFF_EINVAL, "NULL argument");
    if (check_every < 1) return fail(DEFF_EINVAL, "check_every must be >= 1");
    SweepPlan plT, pl1;
    TRY(rank_plans(s, omega, &plT, &pl1));
    deff_ctx *c = s->ctx;
    int64_t iter = 0, checks = 0;
    double deffNew = 1, deffOld = 5, change = 100.0, conv = 0;      // cuh:1171-1173
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    while (iter < max_iter && tol < fabs(change)) {                  // cuh:1232
        const int64_t next_check = ((iter + check_every - 1) / check_every) * check_every;
        const bool do_check = next_check < max_iter;
        const int64_t batch = do_check ? next_check - iter + 1 : max_iter - iter;
        TRY(rank_sweeps(s, plT, pl1, batch));
        iter += batch;
        if (do_check) {
            TRY(rank_flux(s, &deffNew));
            change = (deffOld - deffNew) / (deffOld);                // cuh:1265
            deffOld = deffNew;
            conv = change;
            ++checks;
        }
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    out->iters = iter; out->checks = checks; out->deff_raw = deffNew; out->conv = conv; out->loop_ms = ms;
    if (MFL) memcpy(MFL, s->mfl.data(), sizeof(double) * s->NY);
    if (MFR) memcpy(MFR, s->mfr.data(), sizeof(double) * s->NY);
    return DEFF_OK;
}
DEFF_API_CATCH
