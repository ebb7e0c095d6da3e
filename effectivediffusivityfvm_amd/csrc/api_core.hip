// api_core.hip -- solver context and C ABI (include/deff_amd.h) of the
// MI355X-native effective-diffusivity hot path.  Host side of the path the
// reference implements in Deff2DGPU/Deff2D.cuh: DiscretizeMatrix2D (cuh:815-902),
// initializeGPU/unInitializeGPU (cuh:904-1021), JacobiGPU (cuh:1163-1314).
//
// Design: one context per GPU owns every buffer and a private
// HIP stream; the image is uploaded as bytes (1 B/pixel) and everything else is
// produced on the device; sweeps are enqueued back to back with pointer
// ping-pong (no per-sweep sync or D2D copy, unlike cuh:1239/cuh:1281); a
// convergence check moves 16*ny bytes, not the field (cuh:1245).
#include "ctx.hpp"
#include <algorithm>
#include "driver/jpeg_gray.hpp"
#include "flood_fill.hpp"

// ------------------------------------------------------------- errors -----

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

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
    if ((size_t)((nx + 1) & ~1) * (size_t)ny * (size_t)nimg > (size_t)1 << 31 || (long long)ny * nimg > (1ll << 30))
        return fail(DEFF_EINVAL, "%d image(s) of %dx%d exceed 2^31 cells", nimg, nx, ny);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(DEFF_ENODEV, "no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(DEFF_EINVAL, "device %d out of range [0,%d)", device, count);

    deff_ctx *c = new (std::nothrow) deff_ctx();
    if (!c) return fail(DEFF_ENOMEM, "host allocation failed");
    struct Guard { deff_ctx *c; ~Guard() { if (c) deff_destroy(c); } } guard{c};   // until handed to the caller
    c->device = device;
    c->nxt = nx;
    c->nx = (nx + 1) & ~1;          // arrays are padded to an even width (16-byte rows)
    c->ny = ny; c->nimg = nimg; c->rows = nimg * ny;
    c->n_img = (size_t)c->nx * ny; c->n = c->n_img * nimg;
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
    if (rc != DEFF_OK) return rc;
    resident_chain_ctx_created(device);
    c->chain_counted = true;
    guard.c = nullptr;
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
    if (c->q) (void)hipFree(c->q);
    if (c->resid) (void)hipFree(c->resid);
    if (c->tb_dealt) (void)hipFree(c->tb_dealt);
    if (c->res_flags) (void)hipFree(c->res_flags);
    if (c->res_abort) (void)hipFree(c->res_abort);
    if (c->res_backup) (void)hipFree(c->res_backup);
    if (c->q_host) (void)hipHostFree(c->q_host);
    if (c->chain_counted) resident_chain_ctx_destroyed(c->device);
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
    if (nx) *nx = c->nxt;
    if (ny) *ny = c->ny;
    if (dx) *dx = c->dx;
    if (dy) *dy = c->dy;
    return DEFF_OK;
}
DEFF_API_CATCH

// How many images of an nx x ny mesh a stack context (deff_create_batch + deff_solve_stream) should hold on `device` when
// `images` of them are to be solved -- the planner's knowledge, so that callers need not copy it:
//   - an image that is ONE tall tile (one 128-column strip, at most 16 x 14 rows: kernels_wgtile.hpp) -- one image per CU:
//     every image stays in the registers of its workgroup between two checks, nothing is recomputed, nobody waits
//     (256 x 128^2: 1 472 G cells*iter/s against 1 277 for 1 024 slots and 1 347 for 4 096 on the streaming kernel);
//   - otherwise a stack of ~16 Mi cells for the streaming kernel, ~64 Mi cells when the images outnumber such a stack
//     several times (whole images per wave, nothing recomputed; a shorter run would spend its time draining).
extern "C" int deff_recommended_batch(int device, int nx, int ny, int64_t images, int *slots)
try {
    if (!slots || nx < 2 || ny < 2 || images < 1) return fail(DEFF_EINVAL, "bad argument");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(DEFF_ENODEV, "no such device");
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    const int nxp = (nx + 1) & ~1;
    int64_t want;
    if (nxp <= 128 && ny <= 16 * 14 && cus >= 8) {
        want = cus / 8 * 8;
    } else {
        const int64_t cells = (int64_t)nxp * ny;
        const int64_t big = ((int64_t)64 << 20) / cells, small = std::max<int64_t>(1, ((int64_t)16 << 20) / cells);
        want = (big >= 1 && images >= 3 * big) ? big : small;
    }
    want = std::min<int64_t>(std::min<int64_t>(want, images), 4096);
    *slots = (int)std::max<int64_t>(1, want);
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
void reset_batch_state(deff_ctx *c)
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
int resolve_kernel(const deff_ctx *c, int *k)
{
    int want = c->kernel;
    if (want == DEFF_KERNEL_AUTO) want = c->have_matfree ? DEFF_KERNEL_MATFREE_TB : DEFF_KERNEL_EXPLICIT;
    if (want == DEFF_KERNEL_MATFREE || want == DEFF_KERNEL_MATFREE_TB) {
        if (!c->have_matfree) {
            // an explicit system without a usable row dictionary stays on the explicit kernels
            if (!c->have_explicit) return fail(DEFF_ESTATE, "no system assembled");
            want = DEFF_KERNEL_EXPLICIT;
            *k = want;
            return DEFF_OK;
        }
        // the temporally blocked kernel needs a few rows to stream (rows are always 16-B aligned:
        // an odd mesh width is padded, ctx.hpp)
        if (want == DEFF_KERNEL_MATFREE_TB && c->ny < 8) want = DEFF_KERNEL_MATFREE;
    } else {
        if (!c->have_explicit && !c->have_matfree) return fail(DEFF_ESTATE, "no system assembled");
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
    if (c->res_pending) { TRY(use_device(c)); TRY(resident_check(c)); }   // before arithmetic / plan / test hooks change
    if (!strcmp(key, "rows_explicit")) c->rows_explicit = value;
    else if (!strcmp(key, "rows_matfree")) c->rows_matfree = value;
    else if (!strcmp(key, "wg_matfree")) c->wg_matfree = (value + 7) / 8 * 8;
    else if (!strcmp(key, "nt_explicit")) c->nt_explicit = value ? 1 : 0;
    else if (!strcmp(key, "serpentine")) c->serpentine = value ? 1 : 0;
    else if (!strcmp(key, "tb_T")) c->tb_T = value;
    else if (!strcmp(key, "tb_LY")) c->tb_LY = value;
    else if (!strcmp(key, "tb_impl")) {
        if (value > 2) return fail(DEFF_EINVAL, "tb_impl takes 0 (planner's choice), 1 (streaming) or 2 (workgroup tiles)");
        c->tb_impl = value;
    }
    else if (!strcmp(key, "tb_R")) c->tb_R = value;
    else if (!strcmp(key, "tb_debug_stall")) c->tb_debug_stall = value;
    else if (!strcmp(key, "tb_debug_stall_skip")) c->tb_debug_stall_skip = value;
    else if (!strcmp(key, "res_kt")) c->res_kt = value;
    else if (!strcmp(key, "tb_sym")) c->tb_sym = value;
    else if (!strcmp(key, "tb_launch")) {            // 0: resident passes where possible; 1: one launch per pass
        if (value > 1) return fail(DEFF_EINVAL, "tb_launch takes 0 (resident passes where the tiles fit the chip) or 1 (one launch per pass)");
        c->tb_resident = value == 1 ? 0 : 1;
    }
    else if (!strcmp(key, "flux_reduce")) c->flux_reduce = value > 2 ? 0 : value;
    else if (!strcmp(key, "tb_NW")) c->tb_NW = value;
    else if (!strcmp(key, "dict")) c->dict_enabled = value ? 1 : 0;
    else if (!strcmp(key, "tb_xmajor")) c->tb_xmajor = value ? 1 : 0;
    else if (!strcmp(key, "fma")) c->fma = value ? 1 : 0;
    else if (!strcmp(key, "tb_wall_halo")) c->tb_wall_halo = value > 2 ? 2 : value;
    else if (!strcmp(key, "tb_wg")) c->tb_wg = (value + 7) / 8 * 8;
    else if (!strcmp(key, "tb_ranked")) { c->tb_ranked = value ? 1 : 0; c->tb_rank_lost = 0; }
    else if (!strcmp(key, "tb_tall_deal")) c->tb_tall_deal = value ? 1 : 0;
    else if (!strcmp(key, "tb_sym_age")) c->tb_sym_age = value ? 1 : 0;
    else if (!strcmp(key, "tb_sym_shape")) c->tb_sym_shape = value;
    else if (!strcmp(key, "tb_rank_w0")) c->tb_rank_w[0] = value;
    else if (!strcmp(key, "tb_rank_w1")) c->tb_rank_w[1] = value;
    else if (!strcmp(key, "tb_rank_w2")) c->tb_rank_w[2] = value;
    else if (!strcmp(key, "tb_rank_wall")) c->tb_rank_wall = value;
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
    else if (!strcmp(key, "tb_impl")) *value = c->plan_impl;
    else if (!strcmp(key, "tb_R")) *value = c->plan_R;
    else if (!strcmp(key, "tb_resident")) *value = c->plan_resident;
    else if (!strcmp(key, "tb_sym")) *value = c->links_sym;
    else if (!strcmp(key, "tb_NW")) *value = c->plan_NW;
    else if (!strcmp(key, "tb_fallbacks")) *value = c->res_fallbacks;
    else if (!strcmp(key, "tb_ranked")) *value = c->plan_ranked;
    else if (!strcmp(key, "tb_aged")) *value = c->plan_aged;
    else if (!strcmp(key, "tb_rank_misses")) *value = c->tb_rank_misses;
    else if (!strcmp(key, "tb_rank_lost")) *value = c->tb_rank_lost;
    else return fail(DEFF_EINVAL, "unknown plan key '%s'", key);
    return DEFF_OK;
}
DEFF_API_CATCH

// ------------------------------------------------------------ image -------

int image_shape(deff_ctx *c, int W, int H, int ampX, int ampY)
{
    if (W < 1 || H < 1 || ampX < 1 || ampY < 1)            // cuh:1901-1904
        return fail(DEFF_EINVAL, "image %dx%d / mesh amplification %dx%d invalid", W, H, ampX, ampY);
    if ((long long)W * ampX != c->nxt || (long long)H * ampY != c->ny)
        return fail(DEFF_EINVAL, "image %dx%d x amp %dx%d does not match mesh %dx%d", W, H, ampX, ampY,
                    c->nxt, c->ny);
    if (c->pix && (c->W != W || c->H != H)) { HIP_TRY(hipFree(c->pix)); c->pix = nullptr; }
    TRY(dev_alloc(&c->pix, (size_t)W * H * c->nimg));
    c->W = W; c->H = H; c->ampX = ampX; c->ampY = ampY;
    return DEFF_OK;
}

extern "C" int deff_set_image(deff_ctx *c, const uint8_t *pix, int W, int H, int ampX, int ampY)
try {
    if (!c || !pix) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    TRY(resident_check(c));                          // an unchecked resident interval is settled under the OLD field / system
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
    TRY(resident_check(c));                          // an unchecked resident interval is settled under the OLD field / system
    TRY(image_shape(c, c->nxt, c->ny, 1, 1));
    // stacked rows continue the per-pixel key, so a batch holds images img, img+1, ...
    hipLaunchKernelGGL(k_synth_mask, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->pix, c->nxt, c->ny,
                       c->rows, seed, img);
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
void build_lut_rows(deff_ctx *c, double Ds, double Df, double CL, double CR)
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
int upload_lut(deff_ctx *c, double omega)
{
    if (c->lut_omega == omega) return DEFF_OK;
    std::vector<double> t(LUT_DOUBLES, 0.0);
    bool guard = false;
    // (k starts at 1: row 0 -- the code of every cell outside the mesh -- stays all zeros in every plane, c0 included; the
    // symmetric tile forms rely on it, kernels_wgtile.hpp: wgs_lookup)
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
    TRY(resident_check(c));                          // an unchecked resident interval is settled under the OLD field / system
    TRY(ensure_walls(c));
    c->CL = CL; c->CR = CR;
    hipLaunchKernelGGL(k_wall_D_2phase, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, c->pix, c->W,
                       c->ampX, c->ampY, c->nxt, c->ny, c->rows, Df, Ds, c->Dl, c->Dr);
    HIP_TRY(hipGetLastError());
    c->have_walls = true;

    // matrix-free form: 1 byte per cell + lookup tables
    TRY(dev_alloc(&c->code, c->n));
    c->dict_tried = false;
    hipLaunchKernelGGL(k_phase_codes, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->pix, c->W, c->ampX,
                       c->ampY, c->nx, c->nxt, c->ny, c->rows, c->dom_lo, c->mesh_ny, c->code);
    HIP_TRY(hipGetLastError());
    build_lut_rows(c, Ds, Df, CL, CR);
    c->have_matfree = true;
    c->links_sym = 0;

    // the explicit SoA planes are built on demand (explicit_from_image)
    c->Ds = Ds; c->Df = Df;
    c->phase_mode = 2; c->phase_D[0] = Df; c->phase_D[1] = Ds; c->phase_D[2] = 0.0;
    c->have_explicit = false;
    return DEFF_OK;
}
DEFF_API_CATCH

// Explicit SoA planes for the native 2-phase system: D from the pixels
// (cuh:1988-2000), then the general assembly.  Only needed when an explicit
// kernel is selected or the coefficients are exported.
int explicit_from_image(deff_ctx *c)
{
    if (c->have_explicit) return DEFF_OK;
    if (!c->have_matfree) return fail(DEFF_ESTATE, "no system assembled");
    TRY(ensure_explicit(c));
    TRY(ensure_scratch(c, sizeof(double) * c->n));
    double *D = (double *)c->scratch;
    hipLaunchKernelGGL(k_fill_D_2phase, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->pix, c->W, c->ampX,
                       c->ampY, c->nx, c->nxt, c->ny, c->rows, c->Df, c->Ds, D);
    hipLaunchKernelGGL(k_assemble_from_D, dim3(grid_for(c->n)), dim3(256), 0, c->stream, D,
                       (const unsigned int *)nullptr, c->nx, c->nxt, c->ny, c->rows, c->dom_lo, c->mesh_ny, c->dx,
                       c->dy, c->CL, c->CR, soa_of(c));
    HIP_TRY(hipGetLastError());
    c->have_explicit = true;
    c->c0_omega = NAN;
    c->wrap_links = false;
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
    TRY(resident_check(c));                          // an unchecked resident interval is settled under the OLD field / system
    TRY(ensure_explicit(c));
    TRY(ensure_walls(c));
    const size_t bytes = sizeof(double) * c->n + (Grid ? sizeof(unsigned int) * c->n : 0);
    TRY(ensure_scratch(c, bytes));
    double *dD = (double *)c->scratch;
    unsigned int *dG = Grid ? (unsigned int *)((char *)c->scratch + sizeof(double) * c->n) : nullptr;
    if (Grid) {
        if (c->nx != c->nxt) HIP_TRY(hipMemsetAsync(dG, 0, sizeof(unsigned int) * c->n, c->stream));
        TRY(rows_h2d(c, dG, Grid, (size_t)c->rows));
    }
    c->CL = CL; c->CR = CR;
    hipLaunchKernelGGL(k_fill_D_3phase, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->pix, c->W, c->ampX,
                       c->ampY, c->nx, c->nxt, c->ny, c->rows, Df, Ds, Dg, dD);
    hipLaunchKernelGGL(k_assemble_from_D, dim3(grid_for(c->n)), dim3(256), 0, c->stream, dD, dG, c->nx, c->nxt,
                       c->ny, c->rows, c->dom_lo, c->mesh_ny, c->dx, c->dy, CL, CR, soa_of(c));
    hipLaunchKernelGGL(k_wall_D_from_D, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, dD, c->nx, c->nxt,
                       c->rows, c->Dl, c->Dr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));        // Grid may be freed by the caller
    c->have_explicit = true; c->c0_omega = NAN; c->have_walls = true;
    c->phase_mode = 3; c->phase_D[0] = Df; c->phase_D[1] = Ds; c->phase_D[2] = Dg;
    c->have_matfree = false; c->dict_tried = false; c->wrap_links = false;
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
    TRY(resident_check(c));                          // an unchecked resident interval is settled under the OLD field / system
    TRY(ensure_explicit(c));
    TRY(ensure_walls(c));
    const size_t bytes = sizeof(double) * c->n + (Grid ? sizeof(unsigned int) * c->n : 0);
    TRY(ensure_scratch(c, bytes));
    double *dD = (double *)c->scratch;
    unsigned int *dG = Grid ? (unsigned int *)((char *)c->scratch + sizeof(double) * c->n) : nullptr;
    if (c->nx != c->nxt) HIP_TRY(hipMemsetAsync(c->scratch, 0, bytes, c->stream));
    TRY(rows_h2d(c, dD, D, (size_t)c->rows));
    if (Grid) TRY(rows_h2d(c, dG, Grid, (size_t)c->rows));
    c->CL = CL; c->CR = CR;
    hipLaunchKernelGGL(k_assemble_from_D, dim3(grid_for(c->n)), dim3(256), 0, c->stream, dD, dG, c->nx, c->nxt,
                       c->ny, c->rows, c->dom_lo, c->mesh_ny, c->dx, c->dy, CL, CR, soa_of(c));
    hipLaunchKernelGGL(k_wall_D_from_D, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, dD, c->nx, c->nxt,
                       c->rows, c->Dl, c->Dr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_explicit = true; c->c0_omega = NAN; c->have_walls = true;
    c->phase_mode = 0;
    c->have_matfree = false; c->dict_tried = false; c->wrap_links = false;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_set_system(deff_ctx *c, const double *A, const double *b, const double *D, double CL,
                               double CR)
try {
    if (!c || !A || !b) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    TRY(resident_check(c));                          // an unchecked resident interval is settled under the OLD field / system
    // W link of a first-column cell / E link of a last-column cell: never produced by the reference's assembly
    // (cuh:849-864), but a caller-built A may hold one; see deff_ctx::wrap_links
    bool wrap = false;
    for (int i = 0; i < c->rows && !wrap; ++i)
        wrap = A[((size_t)i * c->nxt) * 5 + 1] != 0 || A[((size_t)(i + 1) * c->nxt - 1) * 5 + 2] != 0;
    if (wrap && c->nx != c->nxt)
        return fail(DEFF_EINVAL, "the system links a wall column to the neighbouring row (A[.][1] != 0 in column 0 or A[.][2] != 0 "
                                 "in the last column): not supported for an odd mesh width (%d)", c->nxt);
    TRY(ensure_explicit(c));
    TRY(ensure_scratch(c, sizeof(double) * 5 * CHUNK_CELLS));
    const size_t n_mesh = (size_t)c->nxt * c->rows;       // cells of the caller's arrays
    for (size_t first = 0; first < n_mesh; first += CHUNK_CELLS) {
        const size_t cnt = (n_mesh - first < CHUNK_CELLS) ? n_mesh - first : CHUNK_CELLS;
        HIP_TRY(hipMemcpyAsync(c->scratch, A + first * 5, sizeof(double) * 5 * cnt, hipMemcpyHostToDevice,
                               c->stream));
        hipLaunchKernelGGL(k_import_aos, dim3(grid_for(cnt)), dim3(256), 0, c->stream,
                           (const double *)c->scratch, first, cnt, c->nx, c->nxt, soa_of(c));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    TRY(rows_h2d(c, c->b, b, (size_t)c->rows));
    if (c->nx != c->nxt) {
        hipLaunchKernelGGL(k_pad_identity, dim3((c->rows + 255) / 256), dim3(256), 0, c->stream, c->nx, c->nxt,
                           c->rows, soa_of(c));
        HIP_TRY(hipGetLastError());
    }
    c->phase_mode = 0;
    c->CL = CL; c->CR = CR;
    c->have_walls = false;
    if (D) {
        TRY(ensure_walls(c));
        // only the first and last column of D are ever read (cuh:1256-1257)
        for (int i = 0; i < c->rows; ++i) {
            c->mf_host[i] = D[(size_t)i * c->nxt];
            c->mf_host[c->rows + i] = D[(size_t)(i + 1) * c->nxt - 1];
        }
        HIP_TRY(hipMemcpyAsync(c->Dl, c->mf_host, sizeof(double) * c->rows, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->Dr, c->mf_host + c->rows, sizeof(double) * c->rows, hipMemcpyHostToDevice,
                               c->stream));
        c->have_walls = true;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_explicit = true; c->c0_omega = NAN;
    c->have_matfree = false; c->dict_tried = false;
    c->wrap_links = wrap;
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_get_system(deff_ctx *c, double *A, double *b)
try {
    if (!c || !A || !b) return fail(DEFF_EINVAL, "NULL argument");
    TRY(use_device(c));
    if (c->have_explicit) {
        TRY(ensure_scratch(c, sizeof(double) * 5 * CHUNK_CELLS));
        const size_t n_mesh = (size_t)c->nxt * c->rows;
        for (size_t first = 0; first < n_mesh; first += CHUNK_CELLS) {
            const size_t cnt = (n_mesh - first < CHUNK_CELLS) ? n_mesh - first : CHUNK_CELLS;
            hipLaunchKernelGGL(k_export_aos, dim3(grid_for(cnt)), dim3(256), 0, c->stream, (double *)c->scratch,
                               first, cnt, c->nx, c->nxt, soa_of(c));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(A + first * 5, c->scratch, sizeof(double) * 5 * cnt, hipMemcpyDeviceToHost,
                                   c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        TRY(rows_d2h(c, b, c->b, (size_t)c->rows));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return DEFF_OK;
    }
    if (c->have_matfree) {
        // expand the codes through the row dictionary (what the matrix-free kernels "see")
        std::vector<uint16_t> code(c->n);
        HIP_TRY(hipMemcpyAsync(code.data(), c->code, sizeof(uint16_t) * c->n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int i = 0; i < c->rows; ++i)
            for (int j = 0; j < c->nxt; ++j) {
                const size_t p = (size_t)i * c->nxt + j;
                const double *row = &c->lut_rows[(size_t)(code[(size_t)i * c->nx + j] >> 3) * 6];
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
    TRY(resident_check(c));                          // an unchecked resident interval is settled under the OLD field / system
    hipLaunchKernelGGL(k_init_linear, dim3(grid_for(c->n)), dim3(256), 0, c->stream, c->x[c->cur], c->nx,
                       c->nxt, c->rows, CL, CR, c->fma);
    HIP_TRY(hipGetLastError());
    c->have_field = true;
    reset_batch_state(c);
    return DEFF_OK;
}
DEFF_API_CATCH

// After a batch solve the images that converged earlier sit frozen in whichever ping-pong
// buffer was current at that moment; bring every image's newest field into x[cur].
int consolidate(deff_ctx *c)
{
    TRY(resident_check(c));
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
    TRY(resident_check(c));                          // an unchecked resident interval is settled under the OLD field / system
    TRY(rows_h2d(c, c->x[c->cur], x, (size_t)c->rows));
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
    if (!c->have_field) return fail(DEFF_ESTATE, "no field: call deff_init_linear() or deff_set_field()");
    TRY(rows_d2h(c, x, (const double *)c->x[c->cur], (size_t)c->rows));
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
    TRY(resident_check(c));
    return DEFF_OK;
}
DEFF_API_CATCH
