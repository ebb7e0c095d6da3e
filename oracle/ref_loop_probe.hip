/* ref_loop_probe -- the reference's host loop JacobiGPU (Deff2DGPU/Deff2D.cuh:1163-1314) with its own LOGIC lines.
 * TEST INFRASTRUCTURE (oracle/_ref).  JacobiGPU interleaves plain C++ -- its local declarations with their literals (deffOld = 5,
 * percentChange = 100, iterToCheck = 10000, 160 threads), the while condition, the check condition, the wall-flux / Deff /
 * change evaluation, the counter, the outputs -- with calls into the CUDA runtime API that only move data (cudaMemcpy,
 * cudaDeviceSynchronize, cudaEvent*).  This file, which is ours, is a function of the same signature whose body is
 *     the reference's lines, included verbatim as fragments cut at build time from the file where it lies (oracle/Makefile):
 *         ref_loop_decl.inc    cuh:1167-1193   declarations, literals, the copy x_vec -> temp_x_vec
 *         ref_loop_while.inc   cuh:1232-1233   while (iterCount < opts.MAX_ITER && opts.ConvergeCriteria < fabs(percentChange)) {
 *         ref_loop_launch.inc  cuh:1237        updateX_SOR<<<numBlocks, threads_per_block>>>(...)   -- valid HIP as it stands
 *         ref_loop_if.inc      cuh:1243-1244   if (iterCount % iterToCheck == 0) {
 *         ref_loop_deff.inc    cuh:1252-1276   Q1, Q2, MFL, MFR, qAvg, deffNew, percentChange, deffOld, myImg->conv, }
 *         ref_loop_count.inc   cuh:1288-1290   iterCount++; }
 *         ref_loop_out.inc     cuh:1309        myImg->deff = deffNew;
 *     and, where the reference calls the CUDA runtime to move data, the same movement through HIP, written here and marked "ours".
 * No CUDA header or library is stood in for: nothing named cuda* exists in this build.  The kernels come from
 * ref_kernel_part.hpp (cuh:17-118, see ref_kernel_probe.hip).
 *
 *   ref_loop in.bin out.bin [precond]          (precond: JacobiGPUPreCond, cuh:1024-1160, built the same way -- see below)
 *     in : int nx, ny; long MAX_ITER; double tol, CL, CR, DCfluid; double A[n*5], b[n], x[n], D[n]
 *     out: long iters; double deff, conv, gpu_ms; double x[n], MFL[ny], MFR[ny] */
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>
#include "ref_kernel_part.hpp"

static int JacobiGPU(double *arr, double *sol, double *x_vec, double *temp_x_vec, options opts,
                     double *d_x_vec, double *d_temp_x_vec, double *d_Coeff, double *d_RHS, double *MFL, double *MFR, double *D,
                     meshInfo mesh, simulationInfo *myImg)                                   /* signature: cuh:1163-1164 */
{
#include "ref_loop_decl.inc"
    (void)Res; (void)dy; (void)str;
    hipMemcpy(d_temp_x_vec, temp_x_vec, sizeof(double) * nRows, hipMemcpyHostToDevice);          /* ours, for cuh:1203 */
    hipMemcpy(d_RHS, sol, sizeof(double) * nRows, hipMemcpyHostToDevice);                        /* ours, for cuh:1210 */
    hipMemcpy(d_Coeff, arr, sizeof(double) * nRows * nCols, hipMemcpyHostToDevice);              /* ours, for cuh:1217 */
    hipEvent_t start, stop;                                                                      /* ours, for cuh:1226-1230 */
    hipEventCreate(&start);
    hipEventCreate(&stop);
    hipEventRecord(start, 0);
#include "ref_loop_while.inc"
#include "ref_loop_launch.inc"
        hipDeviceSynchronize();                                                                  /* ours, for cuh:1239 */
#include "ref_loop_if.inc"
            hipMemcpy(x_vec, d_x_vec, sizeof(double) * nRows, hipMemcpyDeviceToHost);            /* ours, for cuh:1245 */
#include "ref_loop_deff.inc"
        hipMemcpy(d_temp_x_vec, d_x_vec, sizeof(double) * nRows, hipMemcpyDeviceToDevice);       /* ours, for cuh:1281 */
#include "ref_loop_count.inc"
    hipEventRecord(stop, 0);                                                                     /* ours, for cuh:1294-1298 */
    hipEventSynchronize(stop);
    float elapsedTime;
    hipEventElapsedTime(&elapsedTime, start, stop);
    hipMemcpy(x_vec, d_x_vec, sizeof(double) * nRows, hipMemcpyDeviceToHost);                    /* ours, for cuh:1300 */
#include "ref_loop_out.inc"
    myImg->gpuTime += elapsedTime;                                                               /* cuh:1311 */
    return iterCount;                                                                            /* cuh:1313 */
}

/* JacobiGPUPreCond (cuh:1024-1160), the loop of the 3-phase continuation stages: the same construction from its own lines
 *     ref_pre_decl.inc cuh:1028-1054, ref_pre_while.inc cuh:1087-1088, ref_pre_launch.inc cuh:1092, ref_pre_if.inc cuh:1098-1099,
 *     ref_pre_deff.inc cuh:1107-1128, ref_pre_count.inc cuh:1140-1142  (it writes nothing into myImg and returns iterCount) */
static int JacobiGPUPreCond(double *arr, double *sol, double *x_vec, double *temp_x_vec, options opts,
                            double *d_x_vec, double *d_temp_x_vec, double *d_Coeff, double *d_RHS, double *MFL, double *MFR, double *D,
                            meshInfo mesh, simulationInfo *myImg)                            /* signature: cuh:1024-1025 */
{
#include "ref_pre_decl.inc"
    (void)Res; (void)dy; (void)str; (void)myImg;
    hipMemcpy(d_temp_x_vec, temp_x_vec, sizeof(double) * nRows, hipMemcpyHostToDevice);          /* ours, for cuh:1058 */
    hipMemcpy(d_RHS, sol, sizeof(double) * nRows, hipMemcpyHostToDevice);                        /* ours, for cuh:1065 */
    hipMemcpy(d_Coeff, arr, sizeof(double) * nRows * nCols, hipMemcpyHostToDevice);              /* ours, for cuh:1072 */
#include "ref_pre_while.inc"
#include "ref_pre_launch.inc"
        hipDeviceSynchronize();                                                                  /* ours, for cuh:1094 */
#include "ref_pre_if.inc"
            hipMemcpy(x_vec, d_x_vec, sizeof(double) * nRows, hipMemcpyDeviceToHost);            /* ours, for cuh:1100 */
#include "ref_pre_deff.inc"
        hipMemcpy(d_temp_x_vec, d_x_vec, sizeof(double) * nRows, hipMemcpyDeviceToDevice);       /* ours, for cuh:1133 */
#include "ref_pre_count.inc"
    hipMemcpy(x_vec, d_x_vec, sizeof(double) * nRows, hipMemcpyDeviceToHost);                    /* ours, for cuh:1150 */
    return iterCount;                                                                            /* cuh:1159 */
}

int main(int argc, char **argv)
{
    const bool precond = argc == 4 && !strcmp(argv[3], "precond");
    if (argc != 3 && !precond) { fprintf(stderr, "usage: ref_loop in.bin out.bin [precond]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    int dims[2];
    long max_iter;
    double par[4];
    if (!f || fread(dims, 4, 2, f) != 2 || fread(&max_iter, sizeof(long), 1, f) != 1 || fread(par, 8, 4, f) != 4) return 2;
    const int nx = dims[0], ny = dims[1];
    const size_t n = (size_t)nx * ny;
    std::vector<double> A(n * 5), b(n), x(n), D(n), tmp(n), MFL(ny), MFR(ny);
    if (fread(A.data(), 8, n * 5, f) != n * 5 || fread(b.data(), 8, n, f) != n || fread(x.data(), 8, n, f) != n || fread(D.data(), 8, n, f) != n) return 2;
    fclose(f);
    options opts;
    memset(&opts, 0, sizeof opts);
    opts.MAX_ITER = max_iter; opts.ConvergeCriteria = par[0]; opts.CLeft = par[1]; opts.CRight = par[2]; opts.DCfluid = par[3];
    opts.verbose = 0; opts.BatchFlag = 1;
    meshInfo mesh;
    mesh.numCellsX = nx; mesh.numCellsY = ny; mesh.nElements = nx * ny; mesh.dx = 1.0 / nx; mesh.dy = 1.0 / ny;
    simulationInfo img;
    memset(&img, 0, sizeof img);
    double *d_x, *d_tmp, *d_A, *d_b;
    if (hipMalloc((void **)&d_x, n * 8) != hipSuccess || hipMalloc((void **)&d_tmp, n * 8) != hipSuccess ||
        hipMalloc((void **)&d_A, n * 40) != hipSuccess || hipMalloc((void **)&d_b, n * 8) != hipSuccess) return 3;
    hipMemset(d_x, 0, n * 8);                                                                    /* initializeGPU zero-fills, cuh:946-973 */
    const long iters = precond
        ? JacobiGPUPreCond(A.data(), b.data(), x.data(), tmp.data(), opts, d_x, d_tmp, d_A, d_b, MFL.data(), MFR.data(), D.data(), mesh, &img)
        : JacobiGPU(A.data(), b.data(), x.data(), tmp.data(), opts, d_x, d_tmp, d_A, d_b, MFL.data(), MFR.data(), D.data(), mesh, &img);
    FILE *g = fopen(argv[2], "wb");
    if (!g) return 2;
    const double head[3] = {img.deff, img.conv, img.gpuTime};
    fwrite(&iters, sizeof(long), 1, g);
    fwrite(head, 8, 3, g);
    fwrite(x.data(), 8, n, g);
    fwrite(MFL.data(), 8, ny, g);
    fwrite(MFR.data(), 8, ny, g);
    fclose(g);
    return 0;
}
