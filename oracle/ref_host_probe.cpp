/* ref_host_probe -- the HOST-ONLY functions of the reference, called as a program.
 * TEST INFRASTRUCTURE (oracle/_ref): the only source here that is ours is this file.  "ref_host_part.hpp" is produced at build
 * time (oracle/Makefile, target `ref`) from /root/reference/Deff2DGPU/Deff2D.cuh where it lies, by cutting OUT the lines that
 * need CUDA and keeping the rest verbatim -- nothing is written in their place, no header is stood in for:
 *     kept    cuh:1-14     the standard / stb_image includes
 *             cuh:17-68    options, simulationInfo, meshInfo, coordPair
 *             cuh:119-903  printOptions, the CSV writers, readInputFile, readImage, WeightedHarmonicMean, calcPorosity,
 *                          calcFracts3D, Residual, createCMAP, FloodFill, DiscretizeMatrix2D_ImpSolid, DiscretizeMatrix2D
 *             + three loops of the drivers as fragments (the D fills and the linear guess: see `fill` / `guess` below)
 *     dropped cuh:15-16    #include "cuda_runtime.h" / "cuda.h"
 *             cuh:69-118   the two __global__ kernels
 *             cuh:904-end  initializeGPU, unInitializeGPU, the Jacobi loops (CUDA calls throughout) and the four drivers
 * The temporary file lives in a mktemp directory for the duration of the compile; only the binary stays (oracle/_ref/, git-ignored,
 * travels to the GPU box).  What this gives: the reference's OWN code for rows a1, a4, a5, a6 of SURVEY.md section 8 (and
 * Residual, FloodFill, the volume fractions, the input parser) as a checker -- for the oracle on the CPU and for the HIP
 * assembly on the GPU box.  The sweep kernels and the stopping rule are CUDA and stay out of reach.
 *
 *   ref_host assemble  in.bin out.bin     in: int nx, ny, has_grid; double CL, CR; D[n]; (unsigned Grid[n])   out: A[n*5], b[n]
 *   ref_host residual  in.bin             in: int nx, ny; double CL, CR; x[n]; D[n]                           prints %.17g
 *   ref_host floodfill in.bin out.bin     in: int nx, ny; unsigned Grid[n]                                    out: Grid[n]
 *   ref_host fractions in.bin             in: int nx, ny; double Ds, Df; uint8 pix[n]; D[n]                   prints porosity SVF LVF
 *   ref_host fill      in.bin out.bin     in: int W, H; int ampX, ampY, phases; double DCS, DCF, DCG; uint8 pix[W*H]   out: D[n]
 *                                         (the drivers' own loops, included as FRAGMENTS: BatchSim's 2-phase fill cuh:1988-2000,
 *                                         SingleSim3Phase's 3-class fill cuh:1510-1531 -- mesh amplification included)
 *   ref_host guess     in.bin out.bin     in: int nx, ny; double CL, CR                                      out: x[n]  (cuh:1955-1959)
 *   ref_host whm w1 w2 x1 x2                                                                                   prints %.17g
 *   ref_host input file                   readInputFile(file): prints the 17 options, one per line */
#include "ref_host_part.hpp"

static bool rd(FILE *f, void *p, size_t bytes) { return fread(p, 1, bytes, f) == bytes; }

// The drivers keep these loops inline, between file handling and CUDA calls; the loops themselves are plain C++ over local
// variables.  Each function below declares those variables under the drivers' names and #includes the reference's lines.
static void driver_fill_2phase(double *D, double *MFL, double *MFR, meshInfo mesh, options opts, simulationInfo myImg, double DCF, double DCS)
{
#include "ref_frag_fill2.inc"        /* cuh:1988-2000 */
}
static void driver_fill_3phase(double *D, double *MFL, double *MFR, meshInfo mesh, options opts, simulationInfo myImg, double DCF, double DCS,
                               double DCG_Temp)
{
#include "ref_frag_fill3.inc"        /* cuh:1510-1531 */
}
static void driver_linear_guess(double *ConcentrationDist, meshInfo mesh, options opts)
{
#include "ref_frag_guess.inc"        /* cuh:1955-1959 */
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_host assemble|residual|floodfill|fractions|whm|input ...\n"); return 2; }
    const std::string cmd = argv[1];
    if (cmd == "whm" && argc == 6) {
        printf("%.17g\n", WeightedHarmonicMean(atof(argv[2]), atof(argv[3]), atof(argv[4]), atof(argv[5])));
        return 0;
    }
    if (cmd == "input" && argc == 3) {
        options o;
        memset(&o, 0, sizeof o);
        readInputFile(argv[2], &o);
        printf("%.17g\n%.17g\n%.17g\n%d\n%d\n%.17g\n%.17g\n%ld\n%.17g\n%s\n%s\n%d\n%s\n%d\n%d\n%d\n%d\n", o.DCsolid, o.DCfluid, o.DCgas,
               o.MeshIncreaseX, o.MeshIncreaseY, o.CLeft, o.CRight, o.MAX_ITER, o.ConvergeCriteria, o.inputFilename, o.outputFilename,
               o.printCmap, o.CMapName, o.verbose, o.BatchFlag, o.NumImg, o.nPhase);
        return 0;
    }
    if (argc < 3) return 2;
    FILE *f = fopen(argv[2], "rb");
    if (!f) return 2;
    if (cmd == "fill" && argc == 4) {
        int hdr[5];
        double dc[3];
        if (!rd(f, hdr, 20) || !rd(f, dc, 24)) return 2;
        const int W = hdr[0], H = hdr[1];
        options o;
        memset(&o, 0, sizeof o);
        o.MeshIncreaseX = hdr[2]; o.MeshIncreaseY = hdr[3];
        std::vector<unsigned char> pix((size_t)W * H);
        if (!rd(f, pix.data(), pix.size())) return 2;
        simulationInfo img;
        memset(&img, 0, sizeof img);
        img.Width = W; img.Height = H; img.target_data = pix.data();
        meshInfo mesh;
        mesh.numCellsX = W * o.MeshIncreaseX; mesh.numCellsY = H * o.MeshIncreaseY;          // cuh:1907-1908
        mesh.nElements = mesh.numCellsX * mesh.numCellsY;
        std::vector<double> D((size_t)mesh.nElements), MFL(mesh.numCellsY), MFR(mesh.numCellsY);
        if (hdr[4] == 2) driver_fill_2phase(D.data(), MFL.data(), MFR.data(), mesh, o, img, dc[1], dc[0]);
        else driver_fill_3phase(D.data(), MFL.data(), MFR.data(), mesh, o, img, dc[1], dc[0], dc[2]);
        FILE *g = fopen(argv[3], "wb");
        if (!g) return 2;
        fwrite(D.data(), 8, D.size(), g);
        fclose(g);
        return 0;
    }
    int nx = 0, ny = 0;
    if (!rd(f, &nx, 4) || !rd(f, &ny, 4) || nx < 1 || ny < 1) return 2;
    const size_t n = (size_t)nx * ny;
    meshInfo mesh;
    mesh.numCellsX = nx; mesh.numCellsY = ny; mesh.nElements = nx * ny;
    mesh.dx = 1.0 / nx; mesh.dy = 1.0 / ny;                          // cuh:1910-1911
    options o;
    memset(&o, 0, sizeof o);
    if (cmd == "assemble" && argc == 4) {
        int has_grid = 0;
        if (!rd(f, &has_grid, 4) || !rd(f, &o.CLeft, 8) || !rd(f, &o.CRight, 8)) return 2;
        std::vector<double> D(n), A(n * 5), b(n);
        std::vector<unsigned int> G(n);
        if (!rd(f, D.data(), n * 8) || (has_grid && !rd(f, G.data(), n * 4))) return 2;
        if (has_grid) DiscretizeMatrix2D_ImpSolid(D.data(), A.data(), b.data(), mesh, o, G.data());
        else DiscretizeMatrix2D(D.data(), A.data(), b.data(), mesh, o);
        FILE *g = fopen(argv[3], "wb");
        if (!g) return 2;
        fwrite(A.data(), 8, n * 5, g);
        fwrite(b.data(), 8, n, g);
        fclose(g);
        return 0;
    }
    if (cmd == "guess" && argc == 4) {
        if (!rd(f, &o.CLeft, 8) || !rd(f, &o.CRight, 8)) return 2;
        std::vector<double> x(n);
        driver_linear_guess(x.data(), mesh, o);
        FILE *g = fopen(argv[3], "wb");
        if (!g) return 2;
        fwrite(x.data(), 8, n, g);
        fclose(g);
        return 0;
    }
    if (cmd == "residual") {
        std::vector<double> x(n), D(n);
        if (!rd(f, &o.CLeft, 8) || !rd(f, &o.CRight, 8) || !rd(f, x.data(), n * 8) || !rd(f, D.data(), n * 8)) return 2;
        printf("%.17g\n", Residual(ny, nx, &o, x.data(), D.data()));
        return 0;
    }
    if (cmd == "floodfill" && argc == 4) {
        std::vector<unsigned int> G(n);
        if (!rd(f, G.data(), n * 4)) return 2;
        simulationInfo info;
        memset(&info, 0, sizeof info);
        FloodFill(G.data(), &mesh, &info);
        FILE *g = fopen(argv[3], "wb");
        if (!g) return 2;
        fwrite(G.data(), 4, n, g);
        fclose(g);
        return 0;
    }
    if (cmd == "fractions") {
        std::vector<unsigned char> pix(n);
        std::vector<double> D(n);
        if (!rd(f, &o.DCsolid, 8) || !rd(f, &o.DCfluid, 8) || !rd(f, pix.data(), n) || !rd(f, D.data(), n * 8)) return 2;
        simulationInfo info;
        memset(&info, 0, sizeof info);
        calcFracts3D(&info, D.data(), &mesh, &o);
        printf("%.17g %.17g %.17g\n", calcPorosity(pix.data(), nx, ny), info.SVF, info.LVF);
        return 0;
    }
    return 2;
}
