// api_slab.hip -- one image over several GPUs as row slabs (BASELINE config #4).
// See ctx.hpp for the file map.
#include <rccl/rccl.h>

#include "ctx.hpp"

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
    int n = 0, nx = 0, nxt = 0, NY = 0;       // nx: width of the device rows (even), nxt: width of the image
    std::vector<deff_ctx *> ctx;
    std::vector<int> g0, own;                 // first global row and row count of every slab
    std::vector<hipEvent_t> done;             // "pass finished" per slab
    // overlap of the exchange with the interior of the pass: per slab a second stream for the copies and two events --
    // bnd: "the rows my neighbours wait for (first / last SLAB_HALO owned rows) are written", halo: "my halo rows hold
    // the neighbours' new rows"
    std::vector<hipStream_t> xs;
    std::vector<hipEvent_t> bnd, halo;
    int overlap = 1;
    std::vector<double> mfl, mfr;             // global wall fluxes of the last check
};

// The launch plans of one slab's pass: the whole slab in one launch, or -- so that the exchange can start while most of
// the slab is still being swept -- the two SLAB_HALO-row bands the neighbours wait for, then the interior.
struct SlabPass {
    SweepPlan whole, top, bot, mid;
    bool split = false;
};

// overlap: 0 = never split, 1 = split when it pays (a slab of >= 16 Mi cells: three launches + the stream hand-overs
// cost ~10 us of host time per pass, which a small slab does not have -- 4 slabs of 4096 x 1024 on one GPU: 226 us per
// pass split against 185 us unsplit, whereas 4 slabs of 16384 x 4096 hide the exchange completely), 2 = always
static int slab_pass_plans(deff_ctx *c, double omega, int T_override, int overlap, SlabPass *sp)
{
    sp->whole = SweepPlan();
    sp->whole.T_override = T_override;
    TRY(plan_sweeps(c, omega, &sp->whole));
    if (sp->whole.kernel != DEFF_KERNEL_MATFREE_TB) return fail(DEFF_ESTATE, "row-slab mode needs the temporally blocked kernel");
    // keyed on tb_ref_cells -- a figure of (nx, NY, slab count) alone -- so that every slab / rank of an image splits alike
    const size_t ref = c->tb_ref_cells ? c->tb_ref_cells : c->n;
    sp->split = (overlap == 2 || (overlap == 1 && ref >= ((size_t)1 << 24))) && c->own_h >= 3 * SLAB_HALO;
    if (!sp->split) return DEFF_OK;
    const int lo[3] = {c->own_lo, c->own_lo + c->own_h - SLAB_HALO, c->own_lo + SLAB_HALO};
    const int h[3] = {SLAB_HALO, SLAB_HALO, c->own_h - 2 * SLAB_HALO};
    SweepPlan *pl[3] = {&sp->top, &sp->bot, &sp->mid};
    for (int k = 0; k < 3; ++k) {
        *pl[k] = SweepPlan();
        pl[k]->T_override = sp->whole.T;       // the same sweeps per pass as the whole-slab plan
        pl[k]->band_lo = lo[k];
        pl[k]->band_h = h[k];
        TRY(plan_sweeps(c, omega, pl[k]));
    }
    return DEFF_OK;
}

// Enqueue one pass of a slab on its stream: the bands first (then `bnd`), the interior behind them; flips x[cur].
static int slab_enqueue_pass(deff_ctx *c, const SlabPass &sp, hipEvent_t bnd)
{
    if (sp.split) {
        TRY(launch_tb_pass(c, sp.top));
        TRY(launch_tb_pass(c, sp.bot));
        HIP_TRY(hipEventRecord(bnd, c->stream));
        TRY(launch_tb_pass(c, sp.mid));
        c->last_launches += 3;
    } else {
        TRY(launch_tb_pass(c, sp.whole));
        HIP_TRY(hipEventRecord(bnd, c->stream));
        ++c->last_launches;
    }
    c->cur ^= 1;
    HIP_TRY(hipGetLastError());
    return DEFF_OK;
}


static int slab_create_ctx(int device, int nx, int NY, int nslabs, int g0, int own, deff_ctx **out)
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
    // one T per IMAGE: keyed on the largest slab's array, a function of (nx, NY, nslabs) only
    c->tb_ref_cells = (size_t)c->nx * (size_t)((NY + nslabs - 1) / nslabs + 2 * SLAB_HALO);
    return DEFF_OK;
}

extern "C" int deff_slab_group_create(int nslabs, const int *devices, int nx, int NY, deff_slab_group **out)
try {
    if (!out || nslabs < 1) return fail(DEFF_EINVAL, "bad slab group arguments");
    *out = nullptr;
    if (nx < 2) return fail(DEFF_EINVAL, "row-slab mode needs nx >= 2 (got %d)", nx);
    if (NY / nslabs < SLAB_HALO) return fail(DEFF_EINVAL, "%d rows over %d slabs: fewer than %d rows per slab", NY, nslabs, SLAB_HALO);
    deff_slab_group *g = new (std::nothrow) deff_slab_group();
    if (!g) return fail(DEFF_ENOMEM, "host allocation failed");
    g->n = nslabs; g->nxt = nx; g->nx = (nx + 1) & ~1; g->NY = NY;
    g->mfl.assign(NY, 0.0); g->mfr.assign(NY, 0.0);
    int rc = DEFF_OK;
    for (int r = 0; r < nslabs && rc == DEFF_OK; ++r) {
        const int a = (int)((long long)NY * r / nslabs), b = (int)((long long)NY * (r + 1) / nslabs);
        deff_ctx *c = nullptr;
        rc = slab_create_ctx(devices ? devices[r] : 0, nx, NY, nslabs, a, b - a, &c);
        if (rc != DEFF_OK) break;
        g->ctx.push_back(c); g->g0.push_back(a); g->own.push_back(b - a);
        hipEvent_t ev = nullptr, eb = nullptr, eh = nullptr;
        hipStream_t xs = nullptr;
        if (hipSetDevice(c->device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&eh, hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithFlags(&xs, hipStreamNonBlocking) != hipSuccess)
            rc = fail(DEFF_EHIP, "event / stream creation failed");
        g->done.push_back(ev); g->bnd.push_back(eb); g->halo.push_back(eh); g->xs.push_back(xs);
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
        (void)hipSetDevice(g->ctx[r]->device);
        if (r < g->xs.size() && g->xs[r]) { (void)hipStreamSynchronize(g->xs[r]); (void)hipStreamDestroy(g->xs[r]); }
        if (r < g->done.size() && g->done[r]) (void)hipEventDestroy(g->done[r]);
        if (r < g->bnd.size() && g->bnd[r]) (void)hipEventDestroy(g->bnd[r]);
        if (r < g->halo.size() && g->halo[r]) (void)hipEventDestroy(g->halo[r]);
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
        TRY(image_shape(c, c->nxt, c->ny, 1, 1));
        HIP_TRY(hipMemsetAsync(c->pix, 0, (size_t)c->nxt * c->rows, c->stream));
        int m0, a0, cnt;
        slab_window(g, r, &m0, &a0, &cnt);
        HIP_TRY(hipMemcpyAsync(c->pix + (size_t)a0 * c->nxt, pix + (size_t)m0 * c->nxt, (size_t)cnt * c->nxt,
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
        TRY(image_shape(c, c->nxt, c->ny, 1, 1));
        HIP_TRY(hipMemsetAsync(c->pix, 0, (size_t)c->nxt * c->rows, c->stream));
        int m0, a0, cnt;
        slab_window(g, r, &m0, &a0, &cnt);
        // the generator's key is seed*K + img*NY*nx + global cell index: start it at mesh row m0
        const uint64_t base_img_cells = img * (uint64_t)g->NY * (uint64_t)g->nxt + (uint64_t)m0 * (uint64_t)g->nxt;
        hipLaunchKernelGGL(k_synth_mask_at, dim3(grid_for((size_t)cnt * c->nxt)), dim3(256), 0, c->stream,
                           c->pix + (size_t)a0 * c->nxt, (size_t)cnt * c->nxt, seed, base_img_cells);
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

// 3-phase system over the slabs (deff_assemble_3phase per slab): Grid is the whole image's flood-fill
// result (NY x nx, may be NULL); every slab takes the rows of its array window.  The explicit planes
// are harvested into a row dictionary per slab at the first sweep, so the slabs still run on the
// temporally blocked kernel; a system with too many distinct rows is refused by the sweep planner.
extern "C" int deff_slab_group_assemble_3phase(deff_slab_group *g, double Ds, double Df, double Dg,
                                               const unsigned int *Grid, double CL, double CR)
try {
    if (!g) return fail(DEFF_EINVAL, "group is NULL");
    std::vector<unsigned int> win;
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        if (Grid) {
            int m0, a0, cnt;
            slab_window(g, r, &m0, &a0, &cnt);
            win.assign((size_t)c->rows * c->nxt, 0u);
            memcpy(&win[(size_t)a0 * c->nxt], Grid + (size_t)m0 * c->nxt, sizeof(unsigned int) * (size_t)cnt * c->nxt);
        }
        TRY(deff_assemble_3phase(c, Ds, Df, Dg, Grid ? win.data() : nullptr, CL, CR));
    }
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
        TRY(rows_h2d(c, c->x[c->cur] + (size_t)a0 * c->nx, x + (size_t)m0 * c->nxt, (size_t)cnt));
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
        TRY(rows_d2h(c, x + (size_t)g->g0[r] * c->nxt, (const double *)(c->x[c->cur] + (size_t)c->own_lo * c->nx),
                     (size_t)g->own[r]));
    }
    for (int r = 0; r < g->n; ++r) { TRY(use_device(g->ctx[r])); HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream)); }
    return DEFF_OK;
}
DEFF_API_CATCH

// One pass over all slabs.  Per slab r, on its own stream: wait until its halo rows are valid (`halo[r]`, recorded by the
// previous pass's copies), sweep the two bands its neighbours wait for, record `bnd[r]`, sweep the interior.  On the slab's
// copy stream xs[r]: wait for the neighbours' `bnd`, copy their new boundary rows into this slab's halo rows of the NEW
// field (hipMemcpyPeerAsync: xGMI between devices), record `halo[r]`.  So the copies of pass p run while the interiors of
// pass p are still being swept, and pass p+1 of a slab starts as soon as ITS halos are in -- no global synchronisation.
// Buffer reuse is safe by transitivity: a slab overwrites the rows a neighbour copied from two passes later, and it cannot
// get there before that neighbour recorded the `bnd` of the pass in between, which it does only after its copies finished.
static int slab_pass(deff_slab_group *g, std::vector<SlabPass> &sp)
{
    const size_t blk = (size_t)SLAB_HALO * g->nx;                      // doubles per halo block
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        HIP_TRY(hipStreamWaitEvent(c->stream, g->halo[r], 0));         // never recorded yet: returns at once
        TRY(slab_enqueue_pass(c, sp[r], g->bnd[r]));
    }
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        hipStream_t xs = g->overlap ? g->xs[r] : c->stream;
        if (r > 0) {                                                   // top halo <- last own rows of slab r-1
            deff_ctx *u = g->ctx[r - 1];
            HIP_TRY(hipStreamWaitEvent(xs, g->bnd[r - 1], 0));
            const double *src = u->x[u->cur] + (size_t)(u->own_lo + u->own_h - SLAB_HALO) * g->nx;
            HIP_TRY(hipMemcpyPeerAsync(c->x[c->cur], c->device, src, u->device, sizeof(double) * blk, xs));
        }
        if (r + 1 < g->n) {                                            // bottom halo <- first own rows of slab r+1
            deff_ctx *d = g->ctx[r + 1];
            HIP_TRY(hipStreamWaitEvent(xs, g->bnd[r + 1], 0));
            const double *src = d->x[d->cur] + (size_t)d->own_lo * g->nx;
            HIP_TRY(hipMemcpyPeerAsync(c->x[c->cur] + (size_t)(c->own_lo + c->own_h) * g->nx, c->device, src,
                                       d->device, sizeof(double) * blk, xs));
        }
        HIP_TRY(hipEventRecord(g->halo[r], xs));
    }
    return DEFF_OK;
}

// n sweeps on every slab: blocked passes of T, remainder as T = 1 passes, one exchange per pass.
static int slab_sweeps(deff_slab_group *g, std::vector<SlabPass> &plT, std::vector<SlabPass> &pl1, int64_t n)
{
    const int T = plT[0].whole.T;
    while (n > 0) {
        const bool big = n >= T;
        TRY(slab_pass(g, big ? plT : pl1));
        n -= big ? T : 1;
    }
    // whatever comes next on a slab's stream (fluxes, field download, another solve) sees complete halos
    for (int r = 0; r < g->n; ++r) {
        TRY(use_device(g->ctx[r]));
        HIP_TRY(hipStreamWaitEvent(g->ctx[r]->stream, g->halo[r], 0));
    }
    return DEFF_OK;
}

static int slab_plans(deff_slab_group *g, double omega, std::vector<SlabPass> &plT, std::vector<SlabPass> &pl1)
{
    plT.assign(g->n, SlabPass()); pl1.assign(g->n, SlabPass());
    for (int r = 0; r < g->n; ++r) {
        deff_ctx *c = g->ctx[r];
        TRY(use_device(c));
        if (c->tb_T > SLAB_HALO) return fail(DEFF_EINVAL, "tb_T exceeds the slab halo depth %d", SLAB_HALO);
        TRY(slab_pass_plans(c, omega, 0, g->overlap, &plT[r]));
        TRY(slab_pass_plans(c, omega, 1, g->overlap, &pl1[r]));
        c->last_launches = 0;
        // all slabs of one image advance in lock-step: one T, one exchange per pass
        if (plT[r].whole.T != plT[0].whole.T)
            return fail(DEFF_ESTATE, "slab %d plans %d sweeps per pass, slab 0 plans %d: set tb_T on the group, not per slab", r,
                        plT[r].whole.T, plT[0].whole.T);
    }
    return DEFF_OK;
}

extern "C" int deff_slab_group_sweeps(deff_slab_group *g, int64_t n, double omega, float *ms)
try {
    if (!g || n < 0) return fail(DEFF_EINVAL, "bad arguments");
    std::vector<SlabPass> plT, pl1;
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
    std::vector<SlabPass> plT, pl1;
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
    if (key && !strcmp(key, "slab_overlap")) { g->overlap = value > 2 ? 2 : value; return DEFF_OK; }
    for (int r = 0; r < g->n; ++r) TRY(deff_set_tuning(g->ctx[r], key, value));
    return DEFF_OK;
}
DEFF_API_CATCH

extern "C" int deff_slab_group_get_plan(deff_slab_group *g, int slab, const char *key, int *value)
try {
    if (!g || slab < 0 || slab >= g->n) return fail(DEFF_EINVAL, "bad slab index");
    return deff_get_plan(g->ctx[slab], key, value);
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
    int rank = 0, nranks = 1, nx = 0, nxt = 0, NY = 0, maxown = 0;   // nx: device row width (even), nxt: image width
    std::vector<int> g0, own;
    double *d_pack = nullptr, *d_all = nullptr;      // [2*maxown], [nranks*2*maxown]
    std::vector<double> h_all, mfl, mfr;
    // host-staged custom transport (deff_slab_rank_create_custom): the same loop, the blocks go
    // through host buffers and the caller's callbacks instead of RCCL
    deff_host_exchange_fn xchg = nullptr;
    deff_host_allgather_fn gather = nullptr;
    void *user = nullptr;
    std::vector<double> h_send_up, h_send_dn, h_recv_up, h_recv_dn, h_pack;
    // exchange overlapped with the interior of the pass (see slab_pass): copy / RCCL stream and the two events
    hipStream_t xs = nullptr;
    hipEvent_t bnd = nullptr, halo = nullptr;
    int overlap = 1;
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
    if (s->xs) { (void)hipStreamSynchronize(s->xs); (void)hipStreamDestroy(s->xs); }
    if (s->bnd) (void)hipEventDestroy(s->bnd);
    if (s->halo) (void)hipEventDestroy(s->halo);
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
    if (nx < 2) return fail(DEFF_EINVAL, "row-slab mode needs nx >= 2 (got %d)", nx);
    if (NY / nranks < SLAB_HALO) return fail(DEFF_EINVAL, "%d rows over %d ranks: fewer than %d rows per slab", NY, nranks, SLAB_HALO);
    deff_slab_rank *s = new (std::nothrow) deff_slab_rank();
    if (!s) return fail(DEFF_ENOMEM, "host allocation failed");
    s->rank = rank; s->nranks = nranks; s->nxt = nx; s->nx = (nx + 1) & ~1; s->NY = NY;
    for (int r = 0; r < nranks; ++r) {
        const int a = (int)((long long)NY * r / nranks), b = (int)((long long)NY * (r + 1) / nranks);
        s->g0.push_back(a); s->own.push_back(b - a);
        if (b - a > s->maxown) s->maxown = b - a;
    }
    s->mfl.assign(NY, 0.0); s->mfr.assign(NY, 0.0);
    s->h_all.assign((size_t)nranks * 2 * s->maxown, 0.0);
    s->xchg = xchg; s->gather = gather; s->user = user;
    // The grouped ncclSend/ncclRecv on a second stream, concurrent with the interior launch, has never run between two
    // ranks (no multi-GPU box in this pipeline): until it has, the RCCL transport exchanges on the solver's stream unless
    // the caller asks for the overlap (slab_overlap 1 / 2).  The host-staged transport is verified and keeps it.
    s->overlap = xchg ? 1 : 0;
    const size_t blk = (size_t)SLAB_HALO * s->nx;
    if (xchg) {
        s->h_send_up.assign(blk, 0.0); s->h_send_dn.assign(blk, 0.0);
        s->h_recv_up.assign(blk, 0.0); s->h_recv_dn.assign(blk, 0.0);
        s->h_pack.assign((size_t)2 * s->maxown, 0.0);
    }
    int rc = slab_create_ctx(device, nx, NY, nranks, s->g0[rank], s->own[rank], &s->ctx);
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
        else if ((he = hipStreamCreateWithFlags(&s->xs, hipStreamNonBlocking)) != hipSuccess ||
                 (he = hipEventCreateWithFlags(&s->bnd, hipEventDisableTiming)) != hipSuccess ||
                 (he = hipEventCreateWithFlags(&s->halo, hipEventDisableTiming)) != hipSuccess)
            rc = fail(DEFF_EHIP, "stream / event creation failed: %s", hipGetErrorString(he));
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

extern "C" int deff_slab_rank_set_tuning(deff_slab_rank *s, const char *key, int value)
try {
    if (!s || !key) return fail(DEFF_EINVAL, "NULL argument");
    if (!strcmp(key, "slab_overlap")) { s->overlap = value > 2 ? 2 : value; return DEFF_OK; }
    return deff_set_tuning(s->ctx, key, value);
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
    TRY(image_shape(c, c->nxt, c->ny, 1, 1));
    int a = 0, cnt = 0;
    TRY(deff_slab_rank_window(s, &a, &cnt));
    HIP_TRY(hipMemsetAsync(c->pix, 0, (size_t)c->nxt * c->rows, c->stream));
    HIP_TRY(hipMemcpyAsync(c->pix + (size_t)(a + c->dom_lo) * c->nxt, pix_window, (size_t)cnt * c->nxt,
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_image = true; c->have_matfree = false;
    return DEFF_OK;
}
DEFF_API_CATCH

// 3-phase system of this rank's slab; Grid_window = the rows deff_slab_rank_window() names of the
// image's flood-fill result (or NULL)
extern "C" int deff_slab_rank_assemble_3phase(deff_slab_rank *s, double Ds, double Df, double Dg,
                                              const unsigned int *Grid_window, double CL, double CR)
try {
    if (!s) return fail(DEFF_EINVAL, "slab is NULL");
    deff_ctx *c = s->ctx;
    std::vector<unsigned int> win;
    if (Grid_window) {
        int a = 0, cnt = 0;
        TRY(deff_slab_rank_window(s, &a, &cnt));
        win.assign((size_t)c->rows * c->nxt, 0u);
        memcpy(&win[(size_t)(a + c->dom_lo) * c->nxt], Grid_window, sizeof(unsigned int) * (size_t)cnt * c->nxt);
    }
    return deff_assemble_3phase(c, Ds, Df, Dg, Grid_window ? win.data() : nullptr, CL, CR);
}
DEFF_API_CATCH

extern "C" int deff_slab_rank_synth_image(deff_slab_rank *s, uint64_t seed, uint64_t img)
try {
    if (!s) return fail(DEFF_EINVAL, "slab is NULL");
    deff_ctx *c = s->ctx;
    TRY(use_device(c));
    TRY(image_shape(c, c->nxt, c->ny, 1, 1));
    int a = 0, cnt = 0;
    TRY(deff_slab_rank_window(s, &a, &cnt));
    HIP_TRY(hipMemsetAsync(c->pix, 0, (size_t)c->nxt * c->rows, c->stream));
    const uint64_t first = img * (uint64_t)s->NY * (uint64_t)s->nxt + (uint64_t)a * (uint64_t)s->nxt;
    hipLaunchKernelGGL(k_synth_mask_at, dim3(grid_for((size_t)cnt * c->nxt)), dim3(256), 0, c->stream,
                       c->pix + (size_t)(a + c->dom_lo) * c->nxt, (size_t)cnt * c->nxt, seed, first);
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
    TRY(rows_d2h(c, x_own, (const double *)(c->x[c->cur] + (size_t)c->own_lo * c->nx), (size_t)c->own_h));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DEFF_OK;
}
DEFF_API_CATCH

// The exchange of one pass, on stream `xs` (the copy stream when overlapping, else the context's): the first / last
// SLAB_HALO owned rows of the NEW field go to the neighbours, theirs come into the halo rows.  The caller has made `xs`
// wait for the band kernels that write those rows.
static int rank_exchange(deff_slab_rank *s, hipStream_t xs)
{
    deff_ctx *c = s->ctx;
    const size_t blk = (size_t)SLAB_HALO * s->nx;
    double *x = c->x[c->cur];
    if (s->xchg) {                                               // host-staged custom transport
        const bool up = s->rank > 0, dn = s->rank + 1 < s->nranks;
        double *top_own = x + (size_t)c->own_lo * s->nx, *bot_own = x + (size_t)(c->own_lo + c->own_h - SLAB_HALO) * s->nx;
        if (up) HIP_TRY(hipMemcpyAsync(s->h_send_up.data(), top_own, sizeof(double) * blk, hipMemcpyDeviceToHost, xs));
        if (dn) HIP_TRY(hipMemcpyAsync(s->h_send_dn.data(), bot_own, sizeof(double) * blk, hipMemcpyDeviceToHost, xs));
        HIP_TRY(hipStreamSynchronize(xs));                       // the interior of the pass keeps running on the other stream
        if (s->xchg(s->user, up ? s->h_send_up.data() : nullptr, up ? s->h_recv_up.data() : nullptr,
                    dn ? s->h_send_dn.data() : nullptr, dn ? s->h_recv_dn.data() : nullptr, blk) != 0)
            return fail(DEFF_ECOMM, "custom halo exchange failed");
        if (up) HIP_TRY(hipMemcpyAsync(x, s->h_recv_up.data(), sizeof(double) * blk, hipMemcpyHostToDevice, xs));
        if (dn) HIP_TRY(hipMemcpyAsync(x + (size_t)(c->own_lo + c->own_h) * s->nx, s->h_recv_dn.data(), sizeof(double) * blk,
                                       hipMemcpyHostToDevice, xs));
        HIP_TRY(hipStreamSynchronize(xs));                       // the host buffers are reused by the next pass
        return DEFF_OK;
    }
    // a failure between GroupStart and GroupEnd must still close the group, or every later collective
    // on this communicator is poisoned: remember the first error, always call ncclGroupEnd, then report
    NCCL_TRY(ncclGroupStart());
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    auto step = [&](ncclResult_t r, const char *name) { if (first == ncclSuccess && r != ncclSuccess) { first = r; what = name; } };
    if (s->rank > 0) {
        step(ncclSend(x + (size_t)c->own_lo * s->nx, blk, ncclDouble, s->rank - 1, s->comm, xs), "ncclSend(up)");
        step(ncclRecv(x, blk, ncclDouble, s->rank - 1, s->comm, xs), "ncclRecv(up)");
    }
    if (s->rank + 1 < s->nranks) {
        step(ncclSend(x + (size_t)(c->own_lo + c->own_h - SLAB_HALO) * s->nx, blk, ncclDouble, s->rank + 1, s->comm, xs),
             "ncclSend(down)");
        step(ncclRecv(x + (size_t)(c->own_lo + c->own_h) * s->nx, blk, ncclDouble, s->rank + 1, s->comm, xs), "ncclRecv(down)");
    }
    const ncclResult_t end = ncclGroupEnd();
    if (first != ncclSuccess) return fail(DEFF_ECOMM, "%s failed: %s", what, ncclGetErrorString(first));
    if (end != ncclSuccess) return fail(DEFF_ECOMM, "ncclGroupEnd failed: %s", ncclGetErrorString(end));
    return DEFF_OK;
}

// n sweeps of this rank's slab: per pass the two boundary bands, then -- while the interior is swept on the context's
// stream -- the exchange on the copy stream; the next pass waits for the halos only (see slab_pass for the argument).
static int rank_sweeps(deff_slab_rank *s, const SlabPass &plT, const SlabPass &pl1, int64_t n)
{
    deff_ctx *c = s->ctx;
    hipStream_t xs = s->overlap ? s->xs : c->stream;
    while (n > 0) {
        const bool big = n >= plT.whole.T;
        HIP_TRY(hipStreamWaitEvent(c->stream, s->halo, 0));
        TRY(slab_enqueue_pass(c, big ? plT : pl1, s->bnd));
        if (s->nranks > 1) {
            HIP_TRY(hipStreamWaitEvent(xs, s->bnd, 0));
            TRY(rank_exchange(s, xs));
            HIP_TRY(hipEventRecord(s->halo, xs));
        }
        n -= big ? plT.whole.T : 1;
    }
    HIP_TRY(hipStreamWaitEvent(c->stream, s->halo, 0));
    return DEFF_OK;
}

static int rank_plans(deff_slab_rank *s, double omega, SlabPass *plT, SlabPass *pl1)
{
    deff_ctx *c = s->ctx;
    TRY(use_device(c));
    if (c->tb_T > SLAB_HALO) return fail(DEFF_EINVAL, "tb_T exceeds the slab halo depth %d", SLAB_HALO);
    TRY(slab_pass_plans(c, omega, 0, s->nranks > 1 ? s->overlap : 0, plT));
    TRY(slab_pass_plans(c, omega, 1, s->nranks > 1 ? s->overlap : 0, pl1));
    c->last_launches = 0;
    return DEFF_OK;
}

extern "C" int deff_slab_rank_sweeps(deff_slab_rank *s, int64_t n, double omega, float *ms)
try {
    if (!s || n < 0) return fail(DEFF_EINVAL, "bad arguments");
    SlabPass plT, pl1;
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

static __global__ void k_pack_own_flux(const double *__restrict__ mf, int rows, int own_lo, int own_h, int maxown,
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
                       c->nx, c->nxt, c->rows, c->dx, c->CL, c->CR, c->mf);
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
    SlabPass plT, pl1;
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
