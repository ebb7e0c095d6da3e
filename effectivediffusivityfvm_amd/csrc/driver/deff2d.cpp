// deff2d -- command-line driver with the reference's front end and outputs.
//
// Reads the reference's input.txt (same keys), loads the same grayscale JPEGs (decoded to the
// same bytes as stb_image, see jpeg_gray.hpp), runs the four modes of the reference's main
// (Deff2DGPU/Deff2D.cu:17-50) -- {2,3} phases x {single image, numbered batch} -- on the native
// GPU path of libdeff_amd (pixels up, everything else on the device), and writes the
// reference's CSV rows (cuh:177-232) and concentration maps (cuh:497-554).  Because those
// formats keep only 4-7 digits, --json also writes every result at full precision and
// --field-bin dumps the FP64 concentration field.
//
//   deff2d [input.txt] [--device N | --devices 0,1,..] [--json results.json] [--field-bin prefix] [--batch-size B]
//          [--progress file] [--arith reference|contracted] [--precond-maxiter N] [--prefetch-threads K]
// --arith contracted: products fused into adds the way a compiler contracts the reference's expressions
// (kernels_sweep.hpp); default is the reference's written operation order.
//
// --devices with RunBatch 0: the one image (2 or 3 phases) is split into row slabs over the listed GPUs.
// --devices: batch mode over several GPUs from one process -- one host thread and one solver context
// per listed device, images handed out from a shared counter, no inter-GPU communication.
//
// --batch-size B (2-phase batch mode): images of equal size are solved B at a time in one stacked
// context (deff_create_batch) -- each image still stops by its own convergence rule, the numbers are
// those of the one-at-a-time run, but small images fill the GPU.  Default: as many images as bring
// the stack to ~16 Mi cells (1 for images that large).
//
// --progress file: checkpoint/resume for long batches (the reference keeps all results in memory
// and loses them if interrupted, doc section 3.6): every finished image appends one line to `file`;
// on start, images already listed there are not solved again, their rows are taken from the file.
//
// This is host-side orchestration only; all arithmetic of the hot path happens behind the C ABI.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/deff_amd.h"
#include "input_file.hpp"
#include "jpeg_gray.hpp"

using deff::Options;

struct Image {
    std::vector<uint8_t> pix;
    int W = 0, H = 0, nChannels = 0;
};

struct Row {                       // one output row (2-phase: porosity; 3-phase: SVF/LVF)
    std::string name;
    double porosity = 0, SVF = 0, LVF = 0, deff = NAN, seconds = 0, conv = NAN;
    double residual = NAN;         // Residual() cuh:451-494 of the final field (--json only; the reference never prints it)
    int path = 0, nElements = 0;
    long iters = 0;
    std::vector<long> stages;
};

#define CK(expr)                                                                              \
    do {                                                                                      \
        int rc_ = (expr);                                                                     \
        if (rc_ != DEFF_OK) {                                                                 \
            std::fprintf(stderr, "deff2d: %s: %s (%s)\n", #expr, deff_last_error(), deff_error_string(rc_)); \
            return false;                                                                     \
        }                                                                                     \
    } while (0)

static bool load_image(const std::string &path, Image *img)
{
    std::string err;
    if (!deff::jpeg::load_gray(path.c_str(), img->pix, img->W, img->H, img->nChannels, err)) {
        if (img->nChannels > 1)     // the reference's message, cuh:1665-1667
            std::printf("Error: please enter a grascale image with 1 channel.\n Current number of channels = %d\n",
                        img->nChannels);
        else
            std::fprintf(stderr, "deff2d: %s: %s\n", path.c_str(), err.c_str());
        return false;
    }
    return true;
}

static double porosity_of(const Image &im)                         // calcPorosity, cuh:383-408
{
    const double total = (double)im.H * im.W;
    double p = 0;
    for (size_t k = 0; k < im.pix.size(); ++k)
        if (im.pix[k] < 150) p += 1.0 / total;
    return p;
}

// Grid over the mesh for FloodFill.  (The reference indexes the image with mesh-sized loop
// bounds here, cuh:1921-1926, which is only well defined for MeshAmp 1; the amplified lookup
// below is what it evidently means.)
static std::vector<unsigned int> grid_of(const Image &im, const Options &o, int threshold)
{
    const int nx = im.W * o.MeshIncreaseX, ny = im.H * o.MeshIncreaseY;
    std::vector<unsigned int> g((size_t)nx * ny);
    for (int i = 0; i < ny; ++i)
        for (int j = 0; j < nx; ++j)
            g[(size_t)i * nx + j] = im.pix[(size_t)(i / o.MeshIncreaseY) * im.W + j / o.MeshIncreaseX] > threshold ? 1u : 0u;
    return g;
}

static int g_contracted = 0;       // --arith contracted: deff_set_tuning(ctx, "fma", 1) on every context
// MAX_ITER of the 3-phase continuation stages: the reference's literal 1e6 (cuh:1503).  --precond-maxiter N replaces it so
// that a capped run of a large image fits a CPU oracle run (tests/golden/make_img00042_golden.py); not an input.txt key.
static int64_t g_precond_maxiter = 1000000;
// threads that prepare images ahead of each worker's solver (2-phase batch mode), see Prefetcher
static int g_prefetch_threads = 2;

struct Session {                   // one solver context, re-created only when the mesh / batch size changes
    deff_ctx *ctx = nullptr;
    int nx = 0, ny = 0, nimg = 0, device = 0;
    ~Session() { deff_destroy(ctx); }
    bool prepare(int nx_, int ny_, int nimg_ = 1)
    {
        if (ctx && nx == nx_ && ny == ny_ && nimg == nimg_) return true;
        deff_destroy(ctx);
        ctx = nullptr;
        CK(deff_create_batch(device, nx_, ny_, nimg_, &ctx));
        CK(deff_set_tuning(ctx, "fma", g_contracted));
        nx = nx_; ny = ny_; nimg = nimg_;
        return true;
    }
};

static void progress(int64_t iter, double deff, double change, void *user)
{
    std::printf("Iteration = %d, Deff = %1.3e, Deff Change = %1.3e\n", (int)iter, deff / *(double *)user, change);   // cuh:1270
}

// Where a 2-phase image is solved: one context on one GPU, or -- a single image with --devices
// a,b,... -- row slabs over several GPUs (deff_slab_group_*, one exchange of 8 halo rows per
// temporally blocked pass; results identical to the one-GPU run).
struct OneGpu {
    Session &S;
    bool open(const Image &im, const Options &o, int nx, int ny)
    {
        if (!S.prepare(nx, ny)) return false;
        CK(deff_set_image(S.ctx, im.pix.data(), im.W, im.H, o.MeshIncreaseX, o.MeshIncreaseY));
        CK(deff_init_linear(S.ctx, o.CLeft, o.CRight));
        return true;
    }
    bool assemble(const Options &o, double DCF) { CK(deff_assemble_2phase(S.ctx, o.DCsolid, DCF, o.CLeft, o.CRight)); return true; }
    bool solve(const Options &o, double *scale, deff_result *r)
    {
        if (o.verbose == 1 && !o.BatchFlag) deff_set_progress(S.ctx, progress, scale);
        const int rc = deff_solve(S.ctx, 2.0 / 3.0, o.ConvergeCriteria, o.MAX_ITER, 10000, r, nullptr, nullptr);
        deff_set_progress(S.ctx, nullptr, nullptr);
        CK(rc);
        return true;
    }
    bool get_field(double *x) { CK(deff_get_field(S.ctx, x)); return true; }
    bool residual(double *r) { CK(deff_residual(S.ctx, r, nullptr)); return true; }
    bool assemble3(const Options &o, double DCG, const unsigned int *grid)
    {
        CK(deff_assemble_3phase(S.ctx, o.DCsolid, o.DCfluid, DCG, grid, o.CLeft, o.CRight));
        return true;
    }
    bool solve_with(const Options &o, double tol, int64_t max_iter, bool show, double *scale, deff_result *r)
    {
        if (show && o.verbose == 1 && !o.BatchFlag) deff_set_progress(S.ctx, progress, scale);
        const int rc = deff_solve(S.ctx, 2.0 / 3.0, tol, max_iter, 10000, r, nullptr, nullptr);
        deff_set_progress(S.ctx, nullptr, nullptr);
        CK(rc);
        return true;
    }
};

struct Slabs {
    const std::vector<int> &devices;
    deff_slab_group *g = nullptr;
    ~Slabs() { deff_slab_group_destroy(g); }
    bool open(const Image &im, const Options &o, int nx, int ny)
    {
        CK(deff_slab_group_create((int)devices.size(), devices.data(), nx, ny, &g));
        CK(deff_slab_group_set_tuning(g, "fma", g_contracted));
        std::vector<uint8_t> mesh_pix((size_t)nx * ny);              // the slabs take the image at mesh resolution
        for (int i = 0; i < ny; ++i)
            for (int j = 0; j < nx; ++j)
                mesh_pix[(size_t)i * nx + j] = im.pix[(size_t)(i / o.MeshIncreaseY) * im.W + j / o.MeshIncreaseX];
        CK(deff_slab_group_set_image(g, mesh_pix.data()));
        CK(deff_slab_group_init_linear(g, o.CLeft, o.CRight));
        if (o.verbose == 1) std::printf("Row slabs over %zu GPUs\n", devices.size());
        return true;
    }
    bool assemble(const Options &o, double DCF) { CK(deff_slab_group_assemble_2phase(g, o.DCsolid, DCF, o.CLeft, o.CRight)); return true; }
    bool solve(const Options &o, double *, deff_result *r)
    {
        CK(deff_slab_group_solve(g, 2.0 / 3.0, o.ConvergeCriteria, o.MAX_ITER, 10000, r, nullptr, nullptr));
        return true;
    }
    bool get_field(double *x) { CK(deff_slab_group_get_field(g, x)); return true; }
    bool residual(double *r) { *r = NAN; return true; }              // (the rows of the image live on several devices: not offered)
    bool assemble3(const Options &o, double DCG, const unsigned int *grid)
    {
        CK(deff_slab_group_assemble_3phase(g, o.DCsolid, o.DCfluid, DCG, grid, o.CLeft, o.CRight));
        return true;
    }
    bool solve_with(const Options &, double tol, int64_t max_iter, bool, double *, deff_result *r)
    {
        CK(deff_slab_group_solve(g, 2.0 / 3.0, tol, max_iter, 10000, r, nullptr, nullptr));
        return true;
    }
};

// 2-phase image.  single = SingleSim's DCF ramp 100, 1e4, ... up to Df (cuh:1759-1817);
// batch = BatchSim: one solve with Df (cuh:1939-2017).
template <class Target>
static bool solve_2phase(Target &&T, const Image &im, const Options &o, bool single, Row *row, std::vector<double> *field)
{
    const int nx = im.W * o.MeshIncreaseX, ny = im.H * o.MeshIncreaseY;
    row->nElements = nx * ny;
    row->porosity = porosity_of(im);
    std::vector<unsigned int> grid = grid_of(im, o, 150);
    CK(deff_flood_fill(grid.data(), nx, ny, &row->path));
    if (!T.open(im, o, nx, ny)) return false;
    double ms_total = 0;
    auto stage = [&](double DCF) -> bool {
        if (!T.assemble(o, DCF)) return false;
        deff_result r;
        double scale = DCF;
        if (!T.solve(o, &scale, &r)) return false;
        if (o.verbose == 1) std::printf("Iterations taken = %ld\n", (long)r.iters);
        row->deff = r.deff_raw / DCF;                                 // cuh:1802 / cuh:2017
        row->conv = r.conv;
        row->iters = (long)r.iters;
        row->stages.push_back((long)r.iters);
        ms_total += r.loop_ms;
        if (o.verbose == 1) std::printf("DCF = %g, Deff %g\n", DCF, row->deff);
        return true;
    };
    if (single) {
        const double DCF_Max = o.DCfluid;
        double DCF = 10.0;
        int count = 1, ran = 0;
        while (DCF <= DCF_Max) {                                     // cuh:1761
            DCF = std::pow(100, count);
            if (DCF >= DCF_Max) DCF = DCF_Max;
            if (!stage(DCF)) return false;
            ++ran;
            if (DCF == DCF_Max) break;
            ++count;
        }
        if (!ran) {
            // The reference's loop does not execute for Df < 10 and its CSV row then holds
            // uninitialised memory.  Solve once with Df instead, and say so.
            std::fprintf(stderr, "deff2d: note: Df = %g < 10: the reference's single-image ramp runs no solve here; "
                                 "solving directly with Df (as RunBatch: 1 does)\n", o.DCfluid);
            if (!stage(o.DCfluid)) return false;
        }
    } else {
        if (!stage(o.DCfluid)) return false;
    }
    row->seconds = ms_total / 1000.0;
    if (!T.residual(&row->residual)) return false;
    if (field) { field->resize((size_t)nx * ny); if (!T.get_field(field->data())) return false; }
    return true;
}

// 3-phase image: FloodFill on pixels > 200, DCG continuation (cuh:1443-1597).
template <class Target>
static bool solve_3phase(Target &&T, const Image &im, const Options &o, Row *row, std::vector<double> *field)
{
    const int nx = im.W * o.MeshIncreaseX, ny = im.H * o.MeshIncreaseY;
    row->nElements = nx * ny;
    std::vector<unsigned int> grid = grid_of(im, o, 200);
    CK(deff_flood_fill(grid.data(), nx, ny, &row->path));
    if (!T.open(im, o, nx, ny)) return false;
    const double DCF = o.DCfluid, DCG = o.DCgas, DCS = o.DCsolid;
    int stage_no = 1;
    for (double g = 10; g < DCG; g *= 10, ++stage_no) {              // JacobiGPUPreCond stages, cuh:1492-1549
        if (o.verbose == 1) std::printf("Pre-Cond Stage %d: DCG = %1.3e\n", stage_no, g);
        if (!T.assemble3(o, g, grid.data())) return false;
        deff_result r;
        if (!T.solve_with(o, o.ConvergeCriteria * 10, g_precond_maxiter, false, nullptr, &r)) return false;
        if (o.verbose == 1) std::printf("Iterations taken = %ld\n", (long)r.iters);
        row->stages.push_back((long)r.iters);
    }
    // volume fractions by exact comparison of D with the phase values (calcFracts3D, cuh:411-448)
    {
        const double total = (double)nx * ny;
        double s = 0, l = 0;
        for (int i = 0; i < ny; ++i)
            for (int j = 0; j < nx; ++j) {
                const uint8_t v = im.pix[(size_t)(i / o.MeshIncreaseY) * im.W + j / o.MeshIncreaseX];
                const double D = v > 200 ? DCS : (v < 50 ? DCG : DCF);
                if (D == DCS) s += 1.0 / total;
                else if (D == DCF) l += 1.0 / total;
            }
        row->SVF = s; row->LVF = l;
    }
    if (!T.assemble3(o, DCG, grid.data())) return false;
    deff_result r;
    double scale = DCF;
    if (!T.solve_with(o, o.ConvergeCriteria, o.MAX_ITER, true, &scale, &r)) return false;
    if (o.verbose == 1) std::printf("Iterations taken = %ld\n", (long)r.iters);
    row->stages.push_back((long)r.iters);
    row->iters = (long)r.iters;
    row->deff = r.deff_raw / DCF;                                     // cuh:1601
    row->conv = r.conv;
    row->seconds = r.loop_ms / 1000.0;                                // JacobiGPUPreCond does not add to gpuTime, cuh:1147
    if (o.verbose == 1) std::printf("DCF = %g, Deff %g\n", DCF, row->deff);
    if (!T.residual(&row->residual)) return false;
    if (field) { field->resize((size_t)nx * ny); if (!T.get_field(field->data())) return false; }
    return true;
}

// BatchSim3Phase over a group of equally sized images in ONE stacked context (cuh:2056-2419 per
// image).  All images go through a continuation stage together -- each stops by its own rule
// inside the stage (deff_solve_batch) -- and the next stage starts when the last one is done; the
// stage systems are harvested into a row dictionary and swept by the temporally blocked kernel.
static bool solve_3phase_group(Session &S, const std::vector<Image> &ims, const Options &o, Row *rows,
                               std::vector<double> *fields)
{
    const int B = (int)ims.size();
    const int nx = ims[0].W * o.MeshIncreaseX, ny = ims[0].H * o.MeshIncreaseY;
    if (!S.prepare(nx, ny, B)) return false;
    const size_t npix = (size_t)ims[0].W * ims[0].H, ncell = (size_t)nx * ny;
    std::vector<uint8_t> stack(npix * B);
    std::vector<unsigned int> grid(ncell * B);
    const double DCF = o.DCfluid, DCG = o.DCgas, DCS = o.DCsolid;
    for (int k = 0; k < B; ++k) {
        std::memcpy(&stack[npix * k], ims[k].pix.data(), npix);
        rows[k].nElements = nx * ny;
        std::vector<unsigned int> g = grid_of(ims[k], o, 200);
        CK(deff_flood_fill(g.data(), nx, ny, &rows[k].path));
        std::memcpy(&grid[ncell * k], g.data(), sizeof(unsigned int) * ncell);
        const double total = (double)ncell;                          // calcFracts3D, cuh:411-448
        double s = 0, l = 0;
        for (int i = 0; i < ny; ++i)
            for (int j = 0; j < nx; ++j) {
                const uint8_t v = ims[k].pix[(size_t)(i / o.MeshIncreaseY) * ims[k].W + j / o.MeshIncreaseX];
                const double D = v > 200 ? DCS : (v < 50 ? DCG : DCF);
                if (D == DCS) s += 1.0 / total;
                else if (D == DCF) l += 1.0 / total;
            }
        rows[k].SVF = s; rows[k].LVF = l;
    }
    CK(deff_set_image(S.ctx, stack.data(), ims[0].W, ims[0].H, o.MeshIncreaseX, o.MeshIncreaseY));
    CK(deff_init_linear(S.ctx, o.CLeft, o.CRight));
    std::vector<deff_result> res(B);
    int stage_no = 1;
    for (double g = 10; g < DCG; g *= 10, ++stage_no) {              // JacobiGPUPreCond stages, cuh:2184-2326
        if (o.verbose == 1) std::printf("Pre-Cond Stage %d: DCG = %1.3e\n", stage_no, g);
        CK(deff_assemble_3phase(S.ctx, DCS, DCF, g, grid.data(), o.CLeft, o.CRight));
        CK(deff_solve_batch(S.ctx, 2.0 / 3.0, o.ConvergeCriteria * 10, g_precond_maxiter, 10000, res.data(), nullptr, nullptr));
        for (int k = 0; k < B; ++k) rows[k].stages.push_back((long)res[k].iters);
    }
    CK(deff_assemble_3phase(S.ctx, DCS, DCF, DCG, grid.data(), o.CLeft, o.CRight));
    CK(deff_solve_batch(S.ctx, 2.0 / 3.0, o.ConvergeCriteria, o.MAX_ITER, 10000, res.data(), nullptr, nullptr));
    for (int k = 0; k < B; ++k) {
        rows[k].stages.push_back((long)res[k].iters);
        rows[k].iters = (long)res[k].iters;
        rows[k].deff = res[k].deff_raw / DCF;                        // cuh:2370
        rows[k].conv = res[k].conv;
        rows[k].seconds = res[k].loop_ms / 1000.0 / B;               // final stage only (cuh:1147), shared by the group
        if (o.verbose == 1) std::printf("Number%dDCF = %g, Deff %g\n", k, DCF, rows[k].deff);
    }
    std::vector<double> rr(B);
    CK(deff_residual(S.ctx, rr.data(), nullptr));
    for (int k = 0; k < B; ++k) rows[k].residual = rr[k];
    if (fields) { fields->resize(ncell * B); CK(deff_get_field(S.ctx, fields->data())); }
    return true;
}

static void write_cmap(const std::string &name, const std::vector<double> &x, int nx, int ny)   // createCMAP, cuh:497-524
{
    FILE *f = std::fopen(name.c_str(), "w+");
    if (!f) { std::fprintf(stderr, "deff2d: cannot write %s\n", name.c_str()); return; }
    std::fprintf(f, "X,Y,C\n");
    for (int i = 0; i < ny; ++i)
        for (int j = 0; j < nx; ++j) std::fprintf(f, "%d,%d,%1.3e\n", j, i, x[(size_t)i * nx + j]);
    std::fclose(f);
}

static void write_csv(const Options &o, const std::vector<Row> &rows)
{
    FILE *f = std::fopen(o.outputFilename.c_str(), "a+");           // append, header every run: cuh:182-183
    if (!f) { std::fprintf(stderr, "deff2d: cannot write %s\n", o.outputFilename.c_str()); return; }
    if (o.nPhase == 2) {
        std::fprintf(f, "imgNum,porosity,PathFlag,Deff,Time,nElements,converge,ds,df\n");
        for (size_t k = 0; k < rows.size(); ++k) {
            const Row &r = rows[k];
            if (o.BatchFlag) std::fprintf(f, "%d,", (int)k); else std::fprintf(f, "%s,", r.name.c_str());
            std::fprintf(f, "%f,%d,%f,%f,%d,%f,%f,%f\n", r.porosity, r.path, r.deff, r.seconds, r.nElements, r.conv,
                         o.DCsolid, o.DCfluid);
        }
    } else {
        std::fprintf(f, "imgNum,SVF,LVF,PathFlag,Deff,Time,nElements,converge,ds,df,dg\n");
        for (size_t k = 0; k < rows.size(); ++k) {
            const Row &r = rows[k];
            if (o.BatchFlag)                                         // cuh:228-229
                std::fprintf(f, "%d,%f,%f,%d,%1.5e,%f,%d,%1.5e,%1.5e,%1.5e,%1.5e\n", (int)k, r.SVF, r.LVF, r.path, r.deff,
                             r.seconds, r.nElements, r.conv, o.DCsolid, o.DCfluid, o.DCgas);
            else                                                     // cuh:198-199
                std::fprintf(f, "%s,%f,%f,%d,%1.3e,%f,%d,%1.3e,%1.3e,%1.3e,%1.3e\n", r.name.c_str(), r.SVF, r.LVF, r.path,
                             r.deff, r.seconds, r.nElements, r.conv, o.DCsolid, o.DCfluid, o.DCgas);
        }
    }
    std::fclose(f);
}

static void write_json(const std::string &path, const Options &o, const std::vector<Row> &rows)
{
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) { std::fprintf(stderr, "deff2d: cannot write %s\n", path.c_str()); return; }
    std::fprintf(f, "{\"phases\": %d, \"results\": [\n", o.nPhase);
    for (size_t k = 0; k < rows.size(); ++k) {
        const Row &r = rows[k];
        std::fprintf(f, "  {\"image\": \"%s\", \"porosity\": %.17g, \"SVF\": %.17g, \"LVF\": %.17g, \"PathFlag\": %d, ", r.name.c_str(),
                     r.porosity, r.SVF, r.LVF, r.path);
        if (std::isfinite(r.deff)) std::fprintf(f, "\"Deff\": %.17g, ", r.deff); else std::fprintf(f, "\"Deff\": null, ");
        if (std::isfinite(r.conv)) std::fprintf(f, "\"converge\": %.17g, ", r.conv); else std::fprintf(f, "\"converge\": null, ");
        if (std::isfinite(r.residual)) std::fprintf(f, "\"residual\": %.17g, ", r.residual); else std::fprintf(f, "\"residual\": null, ");
        std::fprintf(f, "\"iterations\": %ld, \"stage_iterations\": [", r.iters);
        for (size_t q = 0; q < r.stages.size(); ++q) std::fprintf(f, "%s%ld", q ? ", " : "", r.stages[q]);
        std::fprintf(f, "], \"Time\": %.9g, \"nElements\": %d}%s\n", r.seconds, r.nElements, k + 1 < rows.size() ? "," : "");
    }
    std::fprintf(f, "]}\n");
    std::fclose(f);
}

// ---- 2-phase batch mode as a stream (deff_solve_stream) -----------------------------------------
// Every worker keeps the slots of one stacked context full: its image source hands out the next
// unsolved image index from the shared counter, decodes it, computes porosity / PathFlag, and the
// solver reports each image as it stops.  An image of a different size ends the worker's current
// stream (the slots drain) and opens the next one.
static void progress_append(const std::string &path, int k, const Row &r);

struct Shared {
    const Options *o;
    std::vector<Row> *rows;
    const std::vector<char> *done;
    std::atomic<int> *next_index;
    std::atomic<bool> *failed;
    int count;
    bool want_field;
    const std::function<void(int, const double *, int, int)> *emit;
    const std::string *progress_path;
    std::function<std::string(int)> name;
};

// An image made ready for the solver on the host: decoded, porosity and PathFlag computed.
struct Prepared {
    int k = -1;                    // image index; -1 = the source is exhausted (or failed)
    Image im;
    double porosity = 0;
    int path = 0;
};

// One per worker: host threads that take the next unsolved image indices from the shared counter and prepare them (file
// read, JPEG decode, porosity, flood fill) a few images AHEAD of the solver, so that the GPU does not wait for the host
// between two launches when a slot is refilled (12 288 images of 128^2: ~0.2 ms of host work per image, ~5 images per
// check interval of 25 ms).  The look-ahead is short, so that several workers still share the end of a dataset evenly.
// One thread prepares ~140 images of 1024^2 per second (7 ms each: decode, porosity, flood fill); BASELINE config #5 needs
// ~126 per second and worker to keep a GPU busy, so large images get `threads` > 1 (--prefetch-threads; images may then
// reach the solver out of index order, which the results table does not care about: rows land by index).
class Prefetcher {
public:
    Prefetcher(Shared *sh, size_t depth, int threads = 1) : sh_(sh), depth_(depth), running_(threads < 1 ? 1 : threads)
    {
        // the count is fixed before the first thread exists: a thread that finds no work decrements running_ (leave())
        // while its siblings are still being started
        const int n = running_;
        for (int t = 0; t < n; ++t) th_.emplace_back([this] { run(); });
    }
    ~Prefetcher()
    {
        { std::lock_guard<std::mutex> lock(mu_); stop_ = true; }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    Prepared pop()
    {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [this] { return !q_.empty(); });
        Prepared p = std::move(q_.front());
        if (p.k >= 0) q_.pop_front();                            // the end marker stays
        lock.unlock();
        cv_.notify_all();
        return p;
    }

private:
    // the last thread to run out of work appends the end marker: every prepared image is in the queue before it
    void leave()
    {
        std::lock_guard<std::mutex> lock(mu_);
        if (--running_ == 0) q_.push_back(Prepared());
        cv_.notify_all();
    }
    void run()
    {
        const Options &o = *sh_->o;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [this] { return stop_ || q_.size() < depth_; });
                if (stop_) return;
            }
            Prepared p;
            int k;
            do { k = sh_->next_index->fetch_add(1); } while (k < sh_->count && (*sh_->done)[(size_t)k]);
            if (k >= sh_->count || sh_->failed->load()) return leave();
            if (!load_image(sh_->name(k), &p.im)) {
                *sh_->failed = true;
                return leave();
            }
            p.k = k;
            p.porosity = porosity_of(p.im);
            std::vector<unsigned int> grid = grid_of(p.im, o, 150);
            if (deff_flood_fill(grid.data(), p.im.W * o.MeshIncreaseX, p.im.H * o.MeshIncreaseY, &p.path) != DEFF_OK) {
                *sh_->failed = true;
                return leave();
            }
            { std::lock_guard<std::mutex> lock(mu_); q_.push_back(std::move(p)); }
            cv_.notify_all();
        }
    }
    Shared *sh_;
    size_t depth_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Prepared> q_;
    bool stop_ = false;
    int running_;
    std::vector<std::thread> th_;                                // last member: started when everything else exists
};

struct StreamState {
    Shared *sh = nullptr;
    deff_ctx *ctx = nullptr;
    Prefetcher *source = nullptr;
    int W = 0, H = 0;
    Prepared pending;              // image that did not fit the running stream / first image of the next one
    bool have_pending = false, serve_pending = false;
};

// fills the row's image statistics and hands the pixels to the solver
static int stream_emit(StreamState &st, const Prepared &p, uint8_t *pix, int64_t *id)
{
    const Options &o = *st.sh->o;
    Row &row = (*st.sh->rows)[(size_t)p.k];
    row = Row();
    row.name = st.sh->name(p.k);
    row.nElements = p.im.W * o.MeshIncreaseX * p.im.H * o.MeshIncreaseY;
    row.porosity = p.porosity;
    row.path = p.path;
    if (o.verbose == 1) std::printf("Width = %d Height = %d Channel = %d\nPorosity = %g\n", p.im.W, p.im.H, p.im.nChannels, row.porosity);
    std::memcpy(pix, p.im.pix.data(), p.im.pix.size());
    *id = p.k;
    return 1;
}

static int stream_next(void *user, int /*slot*/, uint8_t *pix, int64_t *id)
{
    StreamState &st = *(StreamState *)user;
    if (st.have_pending) {
        if (!st.serve_pending) return 0;                         // wrong size for this stream: let it drain
        st.have_pending = st.serve_pending = false;
        return stream_emit(st, st.pending, pix, id);
    }
    Prepared p = st.source->pop();                               // prepared ahead by the worker's prefetch thread
    if (p.k < 0) return st.sh->failed->load() ? -1 : 0;
    if (p.im.W != st.W || p.im.H != st.H) {
        st.pending = std::move(p);
        st.have_pending = true;
        st.serve_pending = false;
        return 0;
    }
    return stream_emit(st, p, pix, id);
}

static void stream_done(void *user, int64_t id, int slot, const deff_result *r)
{
    StreamState &st = *(StreamState *)user;
    const Options &o = *st.sh->o;
    Row &row = (*st.sh->rows)[(size_t)id];
    row.deff = r->deff_raw / o.DCfluid;                          // cuh:2017
    row.conv = r->conv;
    row.iters = (long)r->iters;
    row.stages.push_back((long)r->iters);
    row.seconds = r->loop_ms / 1000.0;                           // stream time when the image stopped
    if (o.verbose == 1) std::printf("Number%dDCF = %g, Deff %g\n", (int)id, o.DCfluid, row.deff);
    if (deff_residual_slot(st.ctx, slot, &row.residual) != DEFF_OK) row.residual = NAN;
    progress_append(*st.sh->progress_path, (int)id, row);
    if (st.sh->want_field) {
        const int nx = st.W * o.MeshIncreaseX, ny = st.H * o.MeshIncreaseY;
        std::vector<double> x((size_t)nx * ny);
        if (deff_get_slot_field(st.ctx, slot, x.data()) == DEFF_OK) (*st.sh->emit)((int)id, x.data(), nx, ny);
    }
}

// One line per finished image: index and every field of its row, doubles as hex floats (exact).
static void progress_append(const std::string &path, int k, const Row &r)
{
    if (path.empty()) return;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    FILE *f = std::fopen(path.c_str(), "a");
    if (!f) return;
    std::fprintf(f, "%d %a %a %a %a %a %a %d %d %ld %zu", k, r.porosity, r.SVF, r.LVF, r.deff, r.seconds, r.conv, r.path,
                 r.nElements, r.iters, r.stages.size());
    for (long st : r.stages) std::fprintf(f, " %ld", st);
    std::fprintf(f, " %a\n", r.residual);
    std::fclose(f);
}

static void progress_load(const std::string &path, std::vector<Row> &rows, std::vector<char> &done)
{
    if (path.empty()) return;
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return;
    char line[4096];
    while (std::fgets(line, sizeof line, f)) {
        int k = -1, consumed = 0;
        Row r;
        size_t ns = 0;
        if (std::sscanf(line, "%d %la %la %la %la %la %la %d %d %ld %zu%n", &k, &r.porosity, &r.SVF, &r.LVF, &r.deff, &r.seconds,
                        &r.conv, &r.path, &r.nElements, &r.iters, &ns, &consumed) != 11)
            continue;                                            // torn last line of an interrupted run
        if (k < 0 || k >= (int)rows.size()) continue;
        const char *p = line + consumed;
        bool ok = true;
        for (size_t q = 0; q < ns; ++q) {
            char *e = nullptr;
            const long v = std::strtol(p, &e, 10);
            if (e == p) { ok = false; break; }
            r.stages.push_back(v);
            p = e;
        }
        if (!ok) continue;
        {
            char *e = nullptr;                                   // the residual trails the stages (absent in older files: NaN)
            const double v = std::strtod(p, &e);
            if (e != p) r.residual = v;
        }
        rows[(size_t)k] = r;
        done[(size_t)k] = 1;
    }
    std::fclose(f);
}

int main(int argc, char **argv)
{
    std::string input = "input.txt", json, field_prefix;            // fixed name in the reference, Deff2D.cu:13
    int device = 0, batch_size = 0;
    std::vector<int> devices;
    std::string progress_path;
    for (int a = 1; a < argc; ++a) {
        const std::string s = argv[a];
        if (s == "--device" && a + 1 < argc) device = std::atoi(argv[++a]);
        else if (s == "--devices" && a + 1 < argc) {                 // e.g. 0,1,2,3,4,5,6,7: one worker per entry
            for (const char *p = argv[++a]; *p;) {
                devices.push_back((int)std::strtol(p, const_cast<char **>(&p), 10));
                if (*p == ',') ++p;
            }
        }
        else if (s == "--json" && a + 1 < argc) json = argv[++a];
        else if (s == "--field-bin" && a + 1 < argc) field_prefix = argv[++a];
        else if (s == "--batch-size" && a + 1 < argc) batch_size = std::atoi(argv[++a]);
        else if (s == "--progress" && a + 1 < argc) progress_path = argv[++a];
        else if (s == "--arith" && a + 1 < argc) {
            const std::string v = argv[++a];
            if (v == "contracted" || v == "fma") g_contracted = 1;
            else if (v == "reference" || v == "plain") g_contracted = 0;
            else { std::fprintf(stderr, "deff2d: --arith takes 'reference' or 'contracted'\n"); return 2; }
        }
        else if (s == "--prefetch-threads" && a + 1 < argc) g_prefetch_threads = std::atoi(argv[++a]);
        else if (s == "--precond-maxiter" && a + 1 < argc) {
            g_precond_maxiter = (int64_t)std::strtod(argv[++a], nullptr);
            if (g_precond_maxiter < 1) { std::fprintf(stderr, "deff2d: --precond-maxiter must be >= 1\n"); return 2; }
        }
        else if (s == "-h" || s == "--help") {
            std::printf("usage: deff2d [input.txt] [--device N] [--json results.json] [--field-bin prefix] [--batch-size B] [--devices 0,1,...] [--progress file] [--arith reference|contracted] [--precond-maxiter N] [--prefetch-threads K]\n");
            return 0;
        } else if (!s.empty() && s[0] != '-') input = s;
        else { std::fprintf(stderr, "deff2d: unknown argument %s\n", s.c_str()); return 2; }
    }
    Options o;
    std::string err;
    if (!deff::read_input_file(input.c_str(), &o, &err)) { std::fprintf(stderr, "deff2d: %s\n", err.c_str()); return 1; }
    if (o.verbose == 1) deff::print_options(o);

    const int count = o.BatchFlag ? o.NumImg : 1;
    std::vector<Row> rows((size_t)count);
    const bool want_field = o.printCmap == 1 || !field_prefix.empty();
    const std::function<void(int, const double *, int, int)> emit_field = [&](int k, const double *x, int nx, int ny) {
        if (o.printCmap == 1) {
            char cm[32];
            std::snprintf(cm, sizeof cm, "CMAP_%05d.csv", k);        // batch naming, cuh:2387
            write_cmap(o.BatchFlag ? std::string(cm) : o.CMapName, std::vector<double>(x, x + (size_t)nx * ny), nx, ny);
        }
        if (!field_prefix.empty()) {
            char fn[512];
            std::snprintf(fn, sizeof fn, "%s_%05d_%dx%d.f64", field_prefix.c_str(), k, nx, ny);
            if (FILE *f = std::fopen(fn, "wb")) { std::fwrite(x, sizeof(double), (size_t)nx * ny, f); std::fclose(f); }
        }
    };
    const std::function<std::string(int)> image_name = [&](int k) {
        char numbered[32];
        std::snprintf(numbered, sizeof numbered, "%05d.jpg", k);     // cuh:1876
        return o.BatchFlag ? std::string(numbered) : o.inputFilename;
    };

    // Work distribution: one host thread per device takes image indices from a shared counter --
    // whole images per GPU, no communication; JPEG decoding, flood fill and uploads
    // of one worker overlap the other devices' solves.  Rows land in the table by image index, so
    // the output does not depend on the number of devices or slots.
    std::vector<char> done((size_t)count, 0);
    progress_load(progress_path, rows, done);
    for (int k = 0; k < count; ++k)
        if (done[(size_t)k]) rows[(size_t)k].name = image_name(k);
    std::atomic<int> next_index{0};
    std::atomic<bool> failed{false};

    // ---- 2-phase batch mode: streaming slots (deff_solve_stream) --------------------------------
    Shared shared{&o, &rows, &done, &next_index, &failed, count, want_field, &emit_field, &progress_path, image_name};
    auto stream_worker = [&](int dev) {
        StreamState st;
        st.sh = &shared;
        Prefetcher source(&shared, 8, g_prefetch_threads);
        st.source = &source;
        for (;;) {
            if (!st.have_pending) {                              // first image of the next stream fixes its size
                st.pending = source.pop();
                if (st.pending.k < 0 || failed.load()) return;
                st.have_pending = true;
            }
            st.W = st.pending.im.W; st.H = st.pending.im.H;
            st.serve_pending = true;
            const int nx = st.W * o.MeshIncreaseX, ny = st.H * o.MeshIncreaseY;
            // default: enough slots for a stack of ~16 Mi cells (a 128^2 image is 1 strip x few chunks: the
            // chip needs about a thousand of them in flight); for long runs of small images ~64 Mi cells,
            // where every wave sweeps a whole image with nothing recomputed (128^2: 1 347 against
            // 1 113 G cells*iter/s) -- only when the images outnumber the slots several times, or the
            // stream would spend its time draining
            // (since the tall resident tiles: an image that fits ONE workgroup tile gets one slot per CU instead -- the
            // rule lives in the library: deff_recommended_batch)
            const long long per_dev = (count + (long long)std::max<size_t>(1, devices.size()) - 1) / (long long)std::max<size_t>(1, devices.size());
            int slots = 1;
            if (batch_size > 0) {
                slots = (int)std::min<long long>(std::min<long long>(batch_size, std::max<long long>(1, per_dev)), 4096);
            } else if (deff_recommended_batch(dev, nx, ny, std::max<long long>(1, per_dev), &slots) != DEFF_OK) {
                std::fprintf(stderr, "deff2d: %s\n", deff_last_error());
                failed = true;
                return;
            }
            if (deff_create_batch(dev, nx, ny, slots, &st.ctx) != DEFF_OK ||
                deff_set_tuning(st.ctx, "fma", g_contracted) != DEFF_OK) {
                std::fprintf(stderr, "deff2d: %s\n", deff_last_error());
                failed = true;
                return;
            }
            const int rc = deff_solve_stream(st.ctx, st.W, st.H, o.MeshIncreaseX, o.MeshIncreaseY, o.DCsolid, o.DCfluid,
                                             o.CLeft, o.CRight, 2.0 / 3.0, o.ConvergeCriteria, o.MAX_ITER, 10000, stream_next,
                                             stream_done, &st);
            deff_destroy(st.ctx);
            st.ctx = nullptr;
            if (rc != DEFF_OK) {
                std::fprintf(stderr, "deff2d: %s\n", deff_last_error());
                failed = true;
                return;
            }
            if (!st.have_pending) return;                        // the source ran dry (no odd-sized image waiting)
        }
    };

    // ---- every other mode: one image at a time ----------------------------------------------------
    // 3-phase batch mode: work items are runs of `group3` consecutive images, solved together
    // when equally sized
    std::atomic<int> next_item{0};
    auto worker3 = [&](int dev) {
        Session S;
        S.device = dev;
        int group3 = batch_size;
        for (;;) {
            if (group3 <= 0) {                                   // default: a stack of ~16 Mi cells
                Image first;
                if (!load_image(image_name(0), &first)) { failed = true; return; }
                group3 = (int)std::max<long long>(1, std::min<long long>(4096, (16ll << 20) /
                          ((long long)first.W * o.MeshIncreaseX * first.H * o.MeshIncreaseY)));
            }
            const int w = next_item.fetch_add(1);
            const int k0 = w * group3, k1 = std::min(count, k0 + group3);
            if (k0 >= count || failed.load()) break;
            bool all_done = true;
            for (int k = k0; k < k1; ++k) all_done = all_done && done[(size_t)k];
            if (all_done) continue;
            std::vector<Image> ims((size_t)(k1 - k0));
            for (int k = k0; k < k1; ++k)
                if (!load_image(image_name(k), &ims[(size_t)(k - k0)])) { failed = true; return; }
            for (int k = k0; k < k1;) {
                int e = k + 1;
                while (e < k1 && ims[(size_t)(e - k0)].W == ims[(size_t)(k - k0)].W && ims[(size_t)(e - k0)].H == ims[(size_t)(k - k0)].H) ++e;
                const int nx = ims[(size_t)(k - k0)].W * o.MeshIncreaseX, ny = ims[(size_t)(k - k0)].H * o.MeshIncreaseY;
                for (int q = k; q < e; ++q) { rows[(size_t)q] = Row(); rows[(size_t)q].name = image_name(q); }
                std::vector<Image> run(ims.begin() + (k - k0), ims.begin() + (e - k0));
                std::vector<double> fields;
                if (!solve_3phase_group(S, run, o, &rows[(size_t)k], want_field ? &fields : nullptr)) { failed = true; return; }
                for (int q = k; q < e; ++q) {
                    progress_append(progress_path, q, rows[(size_t)q]);
                    if (want_field) emit_field(q, fields.data() + (size_t)(q - k) * nx * ny, nx, ny);
                }
                k = e;
            }
        }
    };
    auto worker = [&](int dev) {
        Session S;
        S.device = dev;
        for (;;) {
            const int k = next_index.fetch_add(1);
            if (k >= count || failed.load()) break;
            if (done[(size_t)k]) continue;                       // resumed run
            Image im;
            if (!load_image(image_name(k), &im)) { failed = true; return; }
            if (o.verbose == 1) std::printf("Width = %d Height = %d Channel = %d\n", im.W, im.H, im.nChannels);
            const int nx = im.W * o.MeshIncreaseX, ny = im.H * o.MeshIncreaseY;
            rows[(size_t)k] = Row();
            rows[(size_t)k].name = image_name(k);
            std::vector<double> field;
            const bool slabs = !o.BatchFlag && devices.size() >= 2;             // one image over several GPUs
            std::vector<double> *fp = want_field ? &field : nullptr;
            Row *rw = &rows[(size_t)k];
            const bool ok = (o.nPhase == 2) ? (slabs ? solve_2phase(Slabs{devices}, im, o, true, rw, fp)
                                                     : solve_2phase(OneGpu{S}, im, o, !o.BatchFlag, rw, fp))
                                            : (slabs ? solve_3phase(Slabs{devices}, im, o, rw, fp)
                                                     : solve_3phase(OneGpu{S}, im, o, rw, fp));
            if (!ok) { failed = true; return; }
            if (o.verbose == 1 && o.nPhase == 2) std::printf("Porosity = %g\n", rows[(size_t)k].porosity);
            progress_append(progress_path, k, rows[(size_t)k]);
            if (want_field) emit_field(k, field.data(), nx, ny);
        }
    };
    const bool streaming = o.BatchFlag && o.nPhase == 2;
    const bool grouped3 = o.BatchFlag && o.nPhase == 3;
    auto run = [&](int dev) { if (streaming) stream_worker(dev); else if (grouped3) worker3(dev); else worker(dev); };
    if (devices.size() <= 1 || count <= 1) {
        run(devices.empty() ? device : devices[0]);
    } else {
        std::vector<std::thread> pool;
        for (int dev : devices) pool.emplace_back(run, dev);
        for (std::thread &t : pool) t.join();
    }
    if (failed.load()) return 1;
    write_csv(o, rows);                                              // after ALL images, like the reference (cuh:2051)
    if (!json.empty()) write_json(json, o, rows);
    return 0;
}
