/* ref_kernel_probe -- the reference's OWN sweep kernels, run on the MI355X.
 * TEST INFRASTRUCTURE (oracle/_ref).  updateX_SOR and updateX_V1 (Deff2DGPU/Deff2D.cuh:69-118) are written in the CUDA kernel
 * language, which is also HIP's: hipcc compiles their text for gfx950 as it lies.  "ref_kernel_part.hpp" is produced at build
 * time (oracle/Makefile, target `ref`) from the reference where it lies: cuh:17-68 (the structs; the kernels take meshInfo by
 * value) and cuh:69-118 (the two kernels), verbatim, in a mktemp directory for the duration of the compile.  Nothing is written
 * in place of anything: the host side of the reference (cudaMalloc, cudaMemcpy, cudaEvent...: cuh:904-1314) is NOT built --
 * this file, which is ours, allocates with HIP, and launches the kernel the way the reference's loop does:
 *     grid n/160 + 1 blocks of 160 threads (cuh:1169-1170), one launch per sweep, then x <- xNew (cuh:1237-1281).
 * Two binaries: ref_kernel (-ffp-contract=off: the written operation order) and ref_kernel_fma (hipcc's default contraction,
 * the counterpart of nvcc's default -fmad=true).
 *
 *   ref_kernel in.bin out.bin
 *     in : int nx, ny, nsweeps, which (0 = updateX_SOR, 1 = updateX_V1); double A[n*5], b[n], x[n]
 *     out: double x[n] after nsweeps sweeps; stdout: "loop_ms <ms> sweeps <n>" (HIP events around the loop) */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>
#include "ref_kernel_part.hpp"

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: ref_kernel in.bin out.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    int hdr[4];
    if (!f || fread(hdr, 4, 4, f) != 4) return 2;
    const int nx = hdr[0], ny = hdr[1], nsweeps = hdr[2], which = hdr[3];
    const size_t n = (size_t)nx * ny;
    std::vector<double> A(n * 5), b(n), x(n);
    if (fread(A.data(), 8, n * 5, f) != n * 5 || fread(b.data(), 8, n, f) != n || fread(x.data(), 8, n, f) != n) return 2;
    fclose(f);
    meshInfo mesh;
    mesh.numCellsX = nx; mesh.numCellsY = ny; mesh.nElements = nx * ny;
    mesh.dx = 1.0 / nx; mesh.dy = 1.0 / ny;
    double *d_A, *d_b, *d_x, *d_xNew;
    CK(hipMalloc((void **)&d_A, n * 5 * 8)); CK(hipMalloc((void **)&d_b, n * 8));
    CK(hipMalloc((void **)&d_x, n * 8)); CK(hipMalloc((void **)&d_xNew, n * 8));
    CK(hipMemcpy(d_A, A.data(), n * 5 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, b.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, x.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_xNew, 0, n * 8));
    const int threads_per_block = 160;                               // cuh:1169
    const int numBlocks = mesh.nElements / threads_per_block + 1;    // cuh:1170
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < nsweeps; ++it) {
        if (which == 0) updateX_SOR<<<numBlocks, threads_per_block>>>(d_A, d_x, d_b, d_xNew, mesh);   // cuh:1237
        else updateX_V1<<<numBlocks, threads_per_block>>>(d_A, d_x, d_b, d_xNew, mesh);               // cuh:1236 (commented out there)
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());                                                                   // cuh:1239
        CK(hipMemcpy(d_x, d_xNew, n * 8, hipMemcpyDeviceToDevice));                                   // cuh:1281
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("loop_ms %.6f sweeps %d\n", ms, nsweeps);                // the reference's loop shape: launch + sync + D2D copy per sweep
    CK(hipMemcpy(x.data(), d_x, n * 8, hipMemcpyDeviceToHost));
    FILE *g = fopen(argv[2], "wb");
    if (!g) return 2;
    fwrite(x.data(), 8, n, g);
    fclose(g);
    return 0;
}
