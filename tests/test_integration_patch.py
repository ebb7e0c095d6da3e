"""INTEGRATION.md section A applied to the REAL header.

Where the reference is mounted (the authoring container; never on the GPU box), a temporary copy of Deff2D.cu / Deff2D.cuh
gets exactly the patch section A describes -- the CUDA includes replaced by reference_seam.hpp, the ranges that hold the
structs, the kernels, WeightedHarmonicMean, the two assemblies, initializeGPU / unInitializeGPU and the two Jacobi loops
deleted -- and must then compile with the host compiler alone and link against libdeff_amd.so: the reference's four drivers,
readInputFile, FloodFill, Residual, the CSV writers and main, unchanged, on the seam's signatures.  Any drift between
reference_seam.hpp and the reference's call sites breaks this test instead of a reader.  Nothing of the reference is
committed or travels: the copy lives in pytest's tmp_path.

Cross-checks riding along (they pin nothing by the pipeline's rules -- the build only exists through the patch, and
WeightedHarmonicMean inside it is the seam's): the patched translation unit still holds the reference's own text of its
host-only functions.  Small probes of ours call them: Residual() (cuh:451-494) -- the oracle's restatement must return the
same double, bit for bit (same serial order); FloodFill() (cuh:557-713), calcPorosity() (cuh:383-408) and calcFracts3D()
(cuh:411-448) -- the oracle's and the library's flood fill must mark the same cells (including the right-column seeding
quirk of cuh:601, with and without a solid top-left cell, and the top <-> bottom wrap) and the volume fractions must be
the same doubles."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

REF = "/root/reference/Deff2DGPU"
CSRC = os.path.join(ROOT, "effectivediffusivityfvm_amd", "csrc")
LIBDIR = os.path.join(ROOT, "effectivediffusivityfvm_amd")
# cuda includes; options / simulationInfo / meshInfo; updateX_SOR / updateX_V1;
# WeightedHarmonicMean; DiscretizeMatrix2D_ImpSolid ... JacobiGPU (assemblies, initializeGPU, unInitializeGPU, both loops)
DELETED = [(15, 16), (18, 61), (69, 118), (347, 360), (715, 1314)]

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Deff2D.cuh")), reason="reference not mounted")


def apply_section_A(tmp_path):
    src = open(os.path.join(REF, "Deff2D.cuh"), errors="replace").read().split("\n")
    # the ranges must still be what section A says they are (a moved function would silently survive the deletion)
    assert '#include "cuda_runtime.h"' in src[14] and '#include "cuda.h"' in src[15]
    assert src[17].startswith("typedef struct") and "updateX_SOR" in src[68] and "WeightedHarmonicMean" in src[346]
    assert "DiscretizeMatrix2D_ImpSolid" in src[714] and "JacobiGPU" in src[1162]
    assert src[1313].strip() == "}" and "SingleSim3Phase" in "\n".join(src[1314:1320])
    out = []
    for i, line in enumerate(src, start=1):
        if i == 15:
            out += ['#include "reference_seam.hpp"', "using namespace deff_seam;"]
        if any(a <= i <= b for a, b in DELETED):
            continue
        out.append(line)
    text = "\n".join(out)
    assert "cuda" not in text.lower().replace("cudamemcpy", "")      # nothing of CUDA is left outside the deleted ranges
    (tmp_path / "Deff2D.cuh").write_text(text)
    (tmp_path / "Deff2D.cpp").write_text(open(os.path.join(REF, "Deff2D.cu"), errors="replace").read())


def test_section_A_patch_compiles_and_links(tmp_path):
    apply_section_A(tmp_path)
    flags = ["-std=c++17", "-w", f"-I{REF}", f"-I{CSRC}"]          # -I REF: stb_image.h where it lies
    r = subprocess.run(["g++", "-fsyntax-only"] + flags + ["Deff2D.cpp"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    if not os.path.exists(os.path.join(LIBDIR, "libdeff_amd.so")):
        pytest.skip("libdeff_amd.so not built: syntax check only")
    r = subprocess.run(["g++", "-O1"] + flags + ["Deff2D.cpp", "-o", "deff_ref_on_seam", f"-L{LIBDIR}", "-ldeff_amd",
                        f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    # the seam's entry points are what the reference's drivers ended up calling
    syms = subprocess.run(["nm", "-u", "deff_ref_on_seam"], cwd=tmp_path, capture_output=True, text=True).stdout
    for name in ("deff_create", "deff_destroy", "deff_assemble_from_D", "deff_get_system", "deff_set_system", "deff_set_field",
                 "deff_solve", "deff_get_field"):
        assert f" U {name}" in syms, name


PROBE = r"""
#include "Deff2D.cuh"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    int dims[2]; double walls[2];
    if (!f || fread(dims, sizeof(int), 2, f) != 2 || fread(walls, sizeof(double), 2, f) != 2) return 2;
    const size_t n = (size_t)dims[0] * dims[1];
    std::vector<double> x(n), D(n);
    if (fread(x.data(), 8, n, f) != n || fread(D.data(), 8, n, f) != n) return 2;
    options o; memset(&o, 0, sizeof o); o.CLeft = walls[0]; o.CRight = walls[1];
    printf("%.17g\n", Residual(dims[1], dims[0], &o, x.data(), D.data()));
    return 0;
}
"""


def test_reference_residual_text_agrees_with_the_oracle(tmp_path, oracle):
    if not os.path.exists(os.path.join(LIBDIR, "libdeff_amd.so")):
        pytest.skip("libdeff_amd.so not built")
    apply_section_A(tmp_path)
    (tmp_path / "probe.cpp").write_text(PROBE)
    r = subprocess.run(["g++", "-O1", "-ffp-contract=off", "-std=c++17", "-w", f"-I{REF}", f"-I{CSRC}", "probe.cpp", "-o", "probe",
                        f"-L{LIBDIR}", "-ldeff_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    rng = np.random.default_rng(2)
    for nx, ny, CL, CR in ((33, 17, 0.0, 1.0), (16, 16, 0.25, 0.75), (2, 2, 0.0, 1.0), (128, 96, 0.0, 1.0)):
        pix = np.where(rng.random((ny, nx)) < 0.5, 0, 255).astype(np.uint8)
        D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
        A, b = oracle.discretize(D, CL, CR)
        x = oracle.sweeps(A, b, oracle.linear_guess(nx, ny, CL, CR), 13)
        with open(tmp_path / "case.bin", "wb") as f:
            f.write(np.array([nx, ny], dtype=np.int32).tobytes() + np.array([CL, CR]).tobytes() + x.tobytes() + D.tobytes())
        r = subprocess.run([str(tmp_path / "probe"), "case.bin"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        assert float(r.stdout) == oracle.residual(x, D, CL, CR), (nx, ny)


PROBE_FILL = r"""
#include "Deff2D.cuh"
// in: nx ny | ny*nx uint8 pixels | threshold.  out (text): porosity, SVF, LVF, then the flood-filled Grid, one digit per cell
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    int dims[3];
    if (!f || fread(dims, sizeof(int), 3, f) != 3) return 2;
    const int nx = dims[0], ny = dims[1], thr = dims[2];
    std::vector<unsigned char> pix((size_t)nx * ny);
    if (fread(pix.data(), 1, pix.size(), f) != pix.size()) return 2;
    meshInfo mesh; mesh.numCellsX = nx; mesh.numCellsY = ny; mesh.nElements = nx * ny; mesh.dx = 1.0 / nx; mesh.dy = 1.0 / ny;
    simulationInfo info; memset(&info, 0, sizeof info);
    options o; memset(&o, 0, sizeof o); o.DCsolid = 0.5; o.DCfluid = 1.0; o.DCgas = 30.0;
    std::vector<unsigned int> Grid((size_t)nx * ny);
    std::vector<double> D((size_t)nx * ny);
    for (size_t p = 0; p < Grid.size(); ++p) {
        Grid[p] = pix[p] > thr ? 1 : 0;
        D[p] = pix[p] > 200 ? o.DCsolid : (pix[p] < 50 ? o.DCgas : o.DCfluid);      // cuh:1518-1529
    }
    FloodFill(Grid.data(), &mesh, &info);
    calcFracts3D(&info, D.data(), &mesh, &o);
    printf("%.17g %.17g %.17g\n", calcPorosity(pix.data(), nx, ny), info.SVF, info.LVF);
    for (size_t p = 0; p < Grid.size(); ++p) putchar('0' + (int)Grid[p]);
    putchar('\n');
    return 0;
}
"""


def test_reference_flood_fill_and_fractions_agree_with_ours(tmp_path, oracle):
    if not os.path.exists(os.path.join(LIBDIR, "libdeff_amd.so")):
        pytest.skip("libdeff_amd.so not built")
    import effectivediffusivityfvm_amd as pkg
    apply_section_A(tmp_path)
    (tmp_path / "probe_fill.cpp").write_text(PROBE_FILL)
    r = subprocess.run(["g++", "-O1", "-ffp-contract=off", "-std=c++17", "-w", f"-I{REF}", f"-I{CSRC}", "probe_fill.cpp", "-o", "probe_fill",
                        f"-L{LIBDIR}", "-ldeff_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    rng = np.random.default_rng(4)
    vals = np.array([0, 30, 120, 149, 150, 199, 200, 201, 255], dtype=np.uint8)
    for case in range(60):
        nx, ny = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        pix = rng.choice(vals, size=(ny, nx), p=None)
        if case % 3 == 0:
            pix[0, 0] = 255                                  # solid top-left: the right column is seeded (cuh:601)
        if case % 3 == 1:
            pix[0, 0] = 0
        thr = 150 if case % 2 else 200
        with open(tmp_path / "fill.bin", "wb") as f:
            f.write(np.array([nx, ny, thr], dtype=np.int32).tobytes() + pix.tobytes())
        r = subprocess.run([str(tmp_path / "probe_fill"), "fill.bin"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        head, cells = r.stdout.split("\n")[:2]
        por, svf, lvf = (float(v) for v in head.split())
        ref_grid = np.frombuffer(cells.encode(), dtype=np.uint8).reshape(ny, nx) - ord("0")
        g0 = (pix > thr).astype(np.uint32)
        mine, _ = oracle.floodfill(g0)
        lib, _ = pkg.flood_fill(g0)
        assert np.array_equal(ref_grid, mine) and np.array_equal(ref_grid, lib), (case, nx, ny, thr)
        assert por == oracle.porosity(pix)
        D = oracle.fill_D_3phase(pix, 1.0, 0.5, 30.0)
        assert (svf, lvf) == oracle.fracts_3d(D, 0.5, 1.0)
