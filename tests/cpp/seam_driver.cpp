// seam_driver.cpp -- a driver written against the reference's own seam
// (DiscretizeMatrix2D / initializeGPU / JacobiGPU / unInitializeGPU with the
// reference's argument lists), in the call order of BatchSim
// (Deff2DGPU/Deff2D.cuh:1843-2054), compiled against reference_seam.hpp.
// Proves the drop-in: the only thing that differs from a driver written for
// Deff2D.cuh is the header it includes.  Used by tests/test_seam_cpp.py.
//
// usage: seam_driver pix.raw W H Ds Df CL CR tol max_iter out.bin
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../effectivediffusivityfvm_amd/csrc/reference_seam.hpp"

using namespace deff_seam;

int main(int argc, char **argv)
{
    if (argc != 11) { std::fprintf(stderr, "usage: %s pix.raw W H Ds Df CL CR tol max_iter out.bin\n", argv[0]); return 2; }
    options opts{};
    simulationInfo myImg{};
    meshInfo mesh{};
    myImg.Width = std::atoi(argv[2]);
    myImg.Height = std::atoi(argv[3]);
    opts.DCsolid = std::atof(argv[4]);
    opts.DCfluid = std::atof(argv[5]);
    opts.CLeft = std::atof(argv[6]);
    opts.CRight = std::atof(argv[7]);
    opts.ConvergeCriteria = std::atof(argv[8]);
    opts.MAX_ITER = (long)std::atof(argv[9]);
    opts.MeshIncreaseX = opts.MeshIncreaseY = 1;
    opts.verbose = 0;
    opts.BatchFlag = 1;

    std::vector<unsigned char> pix((size_t)myImg.Width * myImg.Height);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(pix.data(), 1, pix.size(), f) != pix.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::fclose(f);
    myImg.target_data = pix.data();

    mesh.numCellsX = myImg.Width * opts.MeshIncreaseX;
    mesh.numCellsY = myImg.Height * opts.MeshIncreaseY;
    mesh.nElements = mesh.numCellsX * mesh.numCellsY;
    mesh.dx = 1.0 / mesh.numCellsX;
    mesh.dy = 1.0 / mesh.numCellsY;
    const int n = mesh.nElements;

    std::vector<double> D(n), MFL(mesh.numCellsY), MFR(mesh.numCellsY), A((size_t)n * 5), RHS(n), x(n), tmp(n);
    for (int i = 0; i < mesh.numCellsY; i++)
        for (int j = 0; j < mesh.numCellsX; j++)
            x[(size_t)i * mesh.numCellsX + j] = (double)j / mesh.numCellsX * (opts.CRight - opts.CLeft) + opts.CLeft;
    myImg.gpuTime = 0;

    double *d_x = nullptr, *d_tmp = nullptr, *d_A = nullptr, *d_b = nullptr;
    if (!initializeGPU(&d_x, &d_tmp, &d_b, &d_A, mesh)) {
        std::printf("\n Error when allocating space in GPU");
        unInitializeGPU(&d_x, &d_tmp, &d_b, &d_A);
        return 1;
    }
    for (int i = 0; i < mesh.numCellsY; i++)
        for (int j = 0; j < mesh.numCellsX; j++)
            D[(size_t)i * mesh.numCellsX + j] =
                (pix[(size_t)(i / opts.MeshIncreaseY) * myImg.Width + j / opts.MeshIncreaseX] < 150) ? opts.DCfluid : opts.DCsolid;

    DiscretizeMatrix2D(D.data(), A.data(), RHS.data(), mesh, opts);
    int iters = JacobiGPU(A.data(), RHS.data(), x.data(), tmp.data(), opts, d_x, d_tmp, d_A, d_b, MFL.data(),
                          MFR.data(), D.data(), mesh, &myImg);
    myImg.deff = myImg.deff / opts.DCfluid;
    unInitializeGPU(&d_x, &d_tmp, &d_b, &d_A);

    FILE *o = std::fopen(argv[10], "wb");
    if (!o) return 2;
    double head[4] = {(double)iters, myImg.deff, myImg.conv, myImg.gpuTime};
    std::fwrite(head, sizeof(double), 4, o);
    std::fwrite(x.data(), sizeof(double), n, o);
    std::fwrite(A.data(), sizeof(double), (size_t)n * 5, o);
    std::fwrite(RHS.data(), sizeof(double), n, o);
    std::fclose(o);
    std::printf("iters=%d deff=%.17g conv=%.17g loop_ms=%.3f\n", iters, myImg.deff, myImg.conv, myImg.gpuTime);
    return 0;
}
