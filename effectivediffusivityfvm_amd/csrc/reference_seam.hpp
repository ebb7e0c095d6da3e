// reference_seam.hpp -- the reference's function-level seam, re-hosted on libdeff_amd.
//
// The reference (adama-wzr/EffectiveDiffusivityFVM, Deff2DGPU/Deff2D.cuh) has no
// plugin or FFI interface: its four drivers call a handful of free functions in
// the same translation unit.  This header declares those functions with the
// reference's names, argument lists, ownership rules and return conventions,
// implemented as thin inline calls into the C ABI (include/deff_amd.h), so a
// driver written against Deff2D.cuh compiles against this header unchanged:
//
//   reference (cuh:line)                          here
//   ------------------------------------------   -----------------------------------------
//   options / simulationInfo / meshInfo 18-61     same field names and order
//   WeightedHarmonicMean            cuh:347-360   same arithmetic (host)
//   DiscretizeMatrix2D              cuh:815-902   device assembly, A/b returned to the host
//   DiscretizeMatrix2D_ImpSolid     cuh:715-812   idem with Grid
//   initializeGPU                   cuh:904-981   creates a solver context; 1 = ok, 0 = failure
//   unInitializeGPU                 cuh:983-1021  destroys it (no device reset)
//   JacobiGPU                       cuh:1163-1314 upload A,b,x; solve; x_vec/deff/conv/gpuTime out
//   JacobiGPUPreCond                cuh:1024-1160 same loop, does not touch myImg
//
// The four "device pointers" the drivers carry between initializeGPU and
// unInitializeGPU are opaque to them (never dereferenced on the host); here
// *d_x_vec holds the context handle and the other three are left NULL.
//
// This is the compatibility surface.  New code should use the native entry
// points (deff_set_image + deff_assemble_2phase + deff_solve): they keep the
// coefficients on the device instead of shipping 40 B/cell over PCIe.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/deff_amd.h"

namespace deff_seam {

struct options {                 // cuh:18-37
    double DCsolid;
    double DCfluid;
    double DCgas;
    int MeshIncreaseX;
    int MeshIncreaseY;
    double CLeft;
    double CRight;
    long int MAX_ITER;
    double ConvergeCriteria;
    char *inputFilename;
    char *outputFilename;
    int printCmap;
    char *CMapName;
    int verbose;
    int BatchFlag;
    int NumImg;
    int nPhase;
};

struct simulationInfo {          // cuh:39-52
    int Width;
    int Height;
    int nChannels;
    double porosity;
    double SVF;
    double LVF;
    double gpuTime;
    unsigned char *target_data;
    double deff;
    bool PathFlag;
    double conv;
};

struct meshInfo {                // cuh:54-61
    int numCellsX;
    int numCellsY;
    int nElements;
    double dx;
    double dy;
};

inline double WeightedHarmonicMean(double w1, double w2, double x1, double x2)   // cuh:347-360
{
    return (w1 + w2) / (w1 / x1 + w2 / x2);
}

namespace detail {
inline void report(const char *where, int rc)
{
    // the reference prints CUDA errors to stderr and carries on (cuh:1204-1209); it never throws
    if (rc != DEFF_OK) std::fprintf(stderr, "%s: %s (%s)\n", where, deff_last_error(), deff_error_string(rc));
}

// One cached context per thread for the assembly-only entry points, which the
// reference calls without any handle (DiscretizeMatrix2D takes no device pointer).
inline deff_ctx *scratch_ctx(int nx, int ny)
{
    thread_local deff_ctx *ctx = nullptr;
    thread_local int cx = 0, cy = 0;
    if (ctx && (cx != nx || cy != ny)) { deff_destroy(ctx); ctx = nullptr; }
    if (!ctx) {
        int rc = deff_create(0, nx, ny, &ctx);                // device 0, cuh:908
        if (rc != DEFF_OK) { report("DiscretizeMatrix2D", rc); ctx = nullptr; return nullptr; }
        cx = nx; cy = ny;
    }
    return ctx;
}

struct Verbose {
    const options *opts;
};
inline void print_check(int64_t iter, double deff, double change, void *user)
{
    const options *o = static_cast<const Verbose *>(user)->opts;
    std::printf("Iteration = %d, Deff = %1.3e, Deff Change = %1.3e\n", (int)iter, deff / o->DCfluid, change);   // cuh:1270
}

inline int jacobi(double *arr, double *sol, double *x_vec, double *temp_x_vec, const options &opts,
                  double *d_x_vec, double *MFL, double *MFR, double *D, const meshInfo &mesh,
                  simulationInfo *out)
{
    deff_ctx *ctx = reinterpret_cast<deff_ctx *>(d_x_vec);
    if (!ctx) { std::fprintf(stderr, "JacobiGPU: initializeGPU was not called\n"); return 0; }
    for (int i = 0; i < mesh.nElements; i++) temp_x_vec[i] = x_vec[i];                   // cuh:1190-1193
    int rc = deff_set_system(ctx, arr, sol, D, opts.CLeft, opts.CRight);                  // cuh:1210-1217
    report("JacobiGPU: upload of A, b", rc);
    if (rc == DEFF_OK) { rc = deff_set_field(ctx, x_vec); report("JacobiGPU: upload of x", rc); }   // cuh:1203
    deff_result res;
    std::memset(&res, 0, sizeof res);
    if (rc == DEFF_OK) {
        Verbose v{&opts};
        if (opts.verbose == 1 && opts.BatchFlag == 0) deff_set_progress(ctx, print_check, &v);   // cuh:1267
        // updateX_SOR with its literal w = 2/3 (cuh:72) is the kernel the reference launches (cuh:1237)
        rc = deff_solve(ctx, 2.0 / 3.0, opts.ConvergeCriteria, (int64_t)opts.MAX_ITER, 10000, &res, MFL, MFR);
        deff_set_progress(ctx, nullptr, nullptr);
        report("JacobiGPU: solve", rc);
    }
    if (rc == DEFF_OK) { rc = deff_get_field(ctx, x_vec); report("JacobiGPU: download of x", rc); }   // cuh:1300
    if (out && rc == DEFF_OK) {
        out->deff = res.deff_raw;                                                         // cuh:1309
        out->conv = res.conv;                                                             // cuh:1275
        out->gpuTime += res.loop_ms;                                                      // cuh:1311
    }
    return (int)res.iters;
}
}  // namespace detail

inline int DiscretizeMatrix2D(double *D, double *A, double *b, meshInfo mesh, options opts)   // cuh:815-902
{
    deff_ctx *ctx = detail::scratch_ctx(mesh.numCellsX, mesh.numCellsY);
    if (!ctx) return 0;
    int rc = deff_assemble_from_D(ctx, D, nullptr, opts.CLeft, opts.CRight);
    if (rc == DEFF_OK) rc = deff_get_system(ctx, A, b);
    detail::report("DiscretizeMatrix2D", rc);
    return 0;                                                                             // always 0, cuh:901
}

inline int DiscretizeMatrix2D_ImpSolid(double *D, double *A, double *b, meshInfo mesh, options opts,
                                       unsigned int *Grid)                                // cuh:715-812
{
    deff_ctx *ctx = detail::scratch_ctx(mesh.numCellsX, mesh.numCellsY);
    if (!ctx) return 0;
    int rc = deff_assemble_from_D(ctx, D, Grid, opts.CLeft, opts.CRight);
    if (rc == DEFF_OK) rc = deff_get_system(ctx, A, b);
    detail::report("DiscretizeMatrix2D_ImpSolid", rc);
    return 0;
}

// 1 = success, 0 = failure (the reference's inverted convention, cuh:978-980); never waits on stdin.
inline int initializeGPU(double **d_x_vec, double **d_temp_x_vec, double **d_RHS, double **d_Coeff,
                         meshInfo mesh)                                                   // cuh:904-981
{
    deff_ctx *ctx = nullptr;
    int rc = deff_create(0, mesh.numCellsX, mesh.numCellsY, &ctx);
    *d_x_vec = reinterpret_cast<double *>(ctx);
    *d_temp_x_vec = nullptr;
    *d_RHS = nullptr;
    *d_Coeff = nullptr;
    if (rc != DEFF_OK) { detail::report("initializeGPU", rc); return 0; }
    return 1;
}

inline void unInitializeGPU(double **d_x_vec, double **d_temp_x_vec, double **d_RHS, double **d_Coeff)   // cuh:983-1021
{
    if (d_x_vec && *d_x_vec) deff_destroy(reinterpret_cast<deff_ctx *>(*d_x_vec));
    if (d_x_vec) *d_x_vec = nullptr;
    if (d_temp_x_vec) *d_temp_x_vec = nullptr;
    if (d_RHS) *d_RHS = nullptr;
    if (d_Coeff) *d_Coeff = nullptr;
}

inline int JacobiGPU(double *arr, double *sol, double *x_vec, double *temp_x_vec, options opts,
                     double *d_x_vec, double *d_temp_x_vec, double *d_Coeff, double *d_RHS, double *MFL,
                     double *MFR, double *D, meshInfo mesh, simulationInfo *myImg)        // cuh:1163-1314
{
    (void)d_temp_x_vec; (void)d_Coeff; (void)d_RHS;
    return detail::jacobi(arr, sol, x_vec, temp_x_vec, opts, d_x_vec, MFL, MFR, D, mesh, myImg);
}

inline int JacobiGPUPreCond(double *arr, double *sol, double *x_vec, double *temp_x_vec, options opts,
                            double *d_x_vec, double *d_temp_x_vec, double *d_Coeff, double *d_RHS, double *MFL,
                            double *MFR, double *D, meshInfo mesh, simulationInfo *myImg)  // cuh:1024-1160
{
    (void)d_temp_x_vec; (void)d_Coeff; (void)d_RHS; (void)myImg;   // does not write deff/conv/gpuTime, cuh:1127-1128
    return detail::jacobi(arr, sol, x_vec, temp_x_vec, opts, d_x_vec, MFL, MFR, D, mesh, nullptr);
}

}  // namespace deff_seam
