// api_residual.hip -- deff_residual / deff_residual_D: the reference's Residual() (Deff2DGPU/Deff2D.cuh:451-494) of the
// context's current field, reduced on the device (kernels_residual.hpp).
#include "ctx.hpp"
#include "kernels_residual.hpp"
#include <algorithm>
#include <vector>

// (dy/dx) * H per class pair, the wall conductance and the "no face" zero -- the reference's expressions, evaluated once on
// the host (this file is built with -ffp-contract=off like the rest of the library)
static ResTable residual_table(const deff_ctx *c)
{
    ResTable t;
    const double dx = c->dx, dy = c->dy;
    for (int k = 0; k < 3; ++k) {
        for (int q = 0; q < 3; ++q) t.g[k][q] = dy / (dx) * whm(dx / 2, dx / 2, c->phase_D[k], c->phase_D[q]);   // cuh:469
        t.g[k][3] = dy / (dx / 2) * c->phase_D[k];                                                              // cuh:466
        for (int q = 4; q < 8; ++q) t.g[k][q] = 0.0;
    }
    return t;
}

static int residual_buffers(deff_ctx *c, size_t partials)
{
    const size_t want = partials + (size_t)c->nimg;
    if (c->resid_cap < want) {
        if (c->resid) { HIP_TRY(hipFree(c->resid)); c->resid = nullptr; c->resid_cap = 0; }
        HIP_TRY(hipMalloc((void **)&c->resid, want * sizeof(double)));
        c->resid_cap = want;
    }
    return DEFF_OK;
}

// sums -> means: R / (numCols * numRows), cuh:491
static int residual_finish(deff_ctx *c, size_t per_img, int nimg, double *r, float *ms)
{
    double *out = c->resid + per_img * nimg;
    hipLaunchKernelGGL(k_residual_final, dim3(nimg), dim3(1024), 0, c->stream, c->resid, per_img, out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    std::vector<double> sums(nimg);
    HIP_TRY(hipMemcpyAsync(sums.data(), out, sizeof(double) * nimg, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (ms) HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    const double cells = (double)((int64_t)c->nxt * (int64_t)c->ny);
    for (int k = 0; k < nimg; ++k) r[k] = sums[k] / cells;
    return DEFF_OK;
}

static int residual_common(deff_ctx *c, const double *r)
{
    if (!c || !r) return fail(DEFF_EINVAL, "NULL argument");
    if (c->slab) return fail(DEFF_EINVAL, "deff_residual: not for row-slab contexts (the rows of the image live on several devices)");
    if (!c->have_field) return fail(DEFF_ESTATE, "no field");
    if (c->nxt < 2 || c->ny < 2) return fail(DEFF_EINVAL, "the residual needs a mesh of at least 2 x 2 cells");
    TRY(use_device(c));
    return DEFF_OK;
}

// the class kernel over `nimg` stacked images starting at field `x` / pixels `pix`; sums -> c->resid (after the partials)
static int residual_classes(deff_ctx *c, const double *x, const uint8_t *pix, int nimg, double *r, float *ms)
{
    if (!c->phase_mode || !c->have_image)
        return fail(DEFF_ESTATE, "deff_residual needs a system assembled from the image (deff_assemble_2phase / _3phase); "
                                 "pass the diffusivities to deff_residual_D() otherwise");
    if ((uint64_t)c->ny * (uint64_t)c->nx * 8u >= ((uint64_t)1 << 32))
        return fail(DEFF_EINVAL, "deff_residual: an image of more than 512 Mi cells (use deff_residual_D)");
    // work items: column strips of 128, cut into runs of kt tiles of RES_ROWS rows that one wave streams top to bottom (the two
    // halo rows of a tile are then loaded once per run).  kt: about 4 096 items in the launch -- one round of waves on 256 CUs;
    // measured at 4096^2, reduction included: kt = 1 42.0 us, 2 34.4, 4 32.0, 8 31.7 -- and at most 16 384 items per image
    // (k_residual_final adds an image's partial sums with ONE workgroup); tuning "res_kt" overrides
    const int ntx = (c->nxt + RES_COLS - 1) / RES_COLS, tiles_y = (c->ny + RES_ROWS - 1) / RES_ROWS;
    int kt = c->res_kt;
    if (kt <= 0) {
        kt = (int)std::min<size_t>(64, std::max<size_t>(1, (size_t)ntx * tiles_y * nimg / 4096));
        kt = std::max(kt, (int)(((size_t)ntx * tiles_y + 16383) / 16384));
    }
    if (kt > tiles_y) kt = tiles_y;
    const int cpi = (tiles_y + kt - 1) / kt;
    const size_t per_img = (size_t)ntx * cpi, tiles = per_img * nimg;
    TRY(residual_buffers(c, tiles));
    const ResTable tab = residual_table(c);
    const bool fast = c->ampX == 1 && c->ampY == 1 && (c->W % 2) == 0;
    const dim3 grid((unsigned)((tiles + 3) / 4));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
#define LAUNCH_RES(P_, F_)                                                                                              \
    hipLaunchKernelGGL((k_residual_classes<P_, F_>), grid, dim3(256), 0, c->stream, x, pix, c->W, c->ampX,              \
                       c->ampY, c->nx, c->nxt, c->ny, nimg, ntx, cpi, kt, c->CL, c->CR, tab, c->resid)
    if (c->phase_mode == 2) { if (fast) LAUNCH_RES(2, true); else LAUNCH_RES(2, false); }
    else { if (fast) LAUNCH_RES(3, true); else LAUNCH_RES(3, false); }
#undef LAUNCH_RES
    HIP_TRY(hipGetLastError());
    return residual_finish(c, per_img, nimg, r, ms);
}

extern "C" int deff_residual(deff_ctx *c, double *r, float *ms)
try {
    TRY(residual_common(c, r));
    TRY(consolidate(c));                                           // every image's newest field in x[cur]
    return residual_classes(c, c->x[c->cur], c->pix, c->nimg, r, ms);
}
DEFF_API_CATCH

// One image of a stack, wherever its newest field lives (a frozen image of a batch solve, a slot of a running stream: callable
// from the deff_image_done_fn callback, like deff_get_slot_field).
extern "C" int deff_residual_slot(deff_ctx *c, int slot, double *r)
try {
    TRY(residual_common(c, r));
    if (slot < 0 || slot >= c->nimg) return fail(DEFF_EINVAL, "bad slot");
    const double *x = c->x[(c->masked || c->in_stream) ? c->buf_of[slot] : c->cur] + (size_t)slot * c->n_img;
    return residual_classes(c, x, c->pix + (size_t)slot * c->W * c->H, 1, r, nullptr);
}
DEFF_API_CATCH

extern "C" int deff_residual_D(deff_ctx *c, const double *D, double CL, double CR, double *r, float *ms)
try {
    TRY(residual_common(c, r));
    if (!D) return fail(DEFF_EINVAL, "D is NULL");
    TRY(consolidate(c));
    const int segs = (c->nxt + 255) / 256;
    const size_t per_img = (size_t)c->ny * segs;
    TRY(residual_buffers(c, per_img * c->nimg));
    TRY(ensure_scratch(c, sizeof(double) * c->n));
    double *dD = (double *)c->scratch;
    if (c->nx != c->nxt) HIP_TRY(hipMemsetAsync(dD, 0, sizeof(double) * c->n, c->stream));
    TRY(rows_h2d(c, dD, D, (size_t)c->rows));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(k_residual_plane, dim3((unsigned)(c->rows * segs)), dim3(256), 0, c->stream, c->x[c->cur], dD, c->nx,
                       c->nxt, c->ny, c->nimg, segs, c->dx, c->dy, CL, CR, c->resid);
    HIP_TRY(hipGetLastError());
    return residual_finish(c, per_img, c->nimg, r, ms);
}
DEFF_API_CATCH
