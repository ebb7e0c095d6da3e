// kernels_setup.hpp -- once-per-image device kernels (gfx950): image -> phases,
// coefficient assembly, layout import/export, initial guess, wall fluxes.
// None of these is on the per-sweep path; they are written for exactness and
// coalescing, not for the last percent.
//
// Batches: a context may hold `nimg` images of the same size stacked on top of
// each other (image k = rows [k*ny, (k+1)*ny) of one tall array).  The zero-flux
// top/bottom boundary of every image means no coefficient ever links two
// images, so the stack is swept as a single domain; only the position class of
// a row (first / last / interior row OF ITS IMAGE) has to be taken modulo ny.
// Below `rows` = nimg*ny is the stacked height and `ny` the height of one image.
//
// Row pitch: device arrays are `nx` wide, the mesh `nxt` <= nx: an odd mesh width is padded by
// one column so that rows stay 16-byte aligned for the two-cells-per-lane kernels.  Pad cells are
// outside the mesh: code 0 (the zero row) / an identity row in explicit planes, field 0, linked
// to nothing (the last mesh column has no east link, the first none to the west), so they never
// change and never contribute.  Geometry (wall columns, dx = 1/nxt, the linear guess) uses nxt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fvm_row.hpp"

namespace deff {

// ---------------------------------------------------------------- image ---

// Synthetic two-phase mask of SURVEY.md 8d (splitmix64 of a per-pixel key).
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// `ny` rows per image, `rows` >= ny stacked rows: the stack holds images img, img+1, ... (the key of
// a pixel is img*nx*ny + its index inside the image, and the stack just keeps counting).
static __global__ void k_synth_mask(uint8_t *__restrict__ pix, int nx, int ny, int rows, uint64_t seed, uint64_t img)
{
    const size_t n = (size_t)nx * rows;
    const uint64_t base = seed * 0x100000001B3ull + img * ((uint64_t)nx * (uint64_t)ny);
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (size_t)gridDim.x * blockDim.x)
        pix[p] = (splitmix64(base + p) >> 63) ? 255 : 0;
}

// Same generator for `count` consecutive cells starting at global cell index `first` of the
// image sequence (row slabs generate only their own window of the image).
static __global__ void k_synth_mask_at(uint8_t *__restrict__ pix, size_t count, uint64_t seed, uint64_t first)
{
    const uint64_t base = seed * 0x100000001B3ull + first;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < count;
         p += (size_t)gridDim.x * blockDim.x)
        pix[p] = (splitmix64(base + p) >> 63) ? 255 : 0;
}

// Pixel of mesh cell (i, j) under nearest-neighbour amplification, cuh:1992-1994.
// (i is the stacked row: image i / ny, row i % ny of that image; H = ny / ampY.)
__device__ __forceinline__ uint8_t cell_pixel(const uint8_t *pix, int W, int ampX, int ampY, int ny,
                                              int i, int j)
{
    const int img = i / ny, li = i - img * ny;
    const int H = ny / ampY;
    return pix[((size_t)img * H + (li / ampY)) * W + (j / ampX)];
}

// 2-phase D fill, cuh:1988-2000: pixel < 150 -> fluid.
static __global__ void k_fill_D_2phase(const uint8_t *__restrict__ pix, int W, int ampX, int ampY,
                                int nx, int nxt, int ny, int rows, double DCF, double DCS, double *__restrict__ D)
{
    const size_t n = (size_t)nx * rows;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(p / nx), j = (int)(p % nx);
        D[p] = j >= nxt ? 0.0 : ((cell_pixel(pix, W, ampX, ampY, ny, i, j) < 150) ? DCF : DCS);
    }
}

// 3-phase D fill, cuh:1518-1529: pixel > 200 -> solid, < 50 -> gas, otherwise fluid.
static __global__ void k_fill_D_3phase(const uint8_t *__restrict__ pix, int W, int ampX, int ampY,
                                int nx, int nxt, int ny, int rows, double DCF, double DCS, double DCG,
                                double *__restrict__ D)
{
    const size_t n = (size_t)nx * rows;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(p / nx), j = (int)(p % nx);
        if (j >= nxt) { D[p] = 0.0; continue; }
        const uint8_t v = cell_pixel(pix, W, ampX, ampY, ny, i, j);
        D[p] = (v > 200) ? DCS : (v < 50 ? DCG : DCF);
    }
}

// Diffusivity of the first and last cell of every row, for the wall fluxes
// (cuh:1256-1257 read D[j*nx] and D[(j+1)*nx-1]).
static __global__ void k_wall_D_2phase(const uint8_t *__restrict__ pix, int W, int ampX, int ampY,
                                int nxt, int ny, int rows, double DCF, double DCS,
                                double *__restrict__ Dl, double *__restrict__ Dr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    Dl[i] = (cell_pixel(pix, W, ampX, ampY, ny, i, 0) < 150) ? DCF : DCS;
    Dr[i] = (cell_pixel(pix, W, ampX, ampY, ny, i, nxt - 1) < 150) ? DCF : DCS;
}

static __global__ void k_wall_D_from_D(const double *__restrict__ D, int nx, int nxt, int rows,
                                double *__restrict__ Dl, double *__restrict__ Dr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    Dl[i] = D[(size_t)i * nx];
    Dr[i] = D[(size_t)i * nx + nxt - 1];
}

// Matrix-free code, 16 bits per cell: the BYTE OFFSET (row index x 8) of the cell's matrix
// row in the lookup tables (lut_layout.hpp).  Row 0 is the all-zero row (cells outside the
// mesh); for the native 2-phase system row 1 + (ycls*3 + xcls)*32 + c5 holds the row of a cell
// of position class (ycls, xcls) whose own / W / E / S(row+1) / N(row-1) phases are the bits
// 0..4 of c5 (1 = solid, i.e. pixel >= 150).  Neighbours outside the mesh read the clamped
// cell; their bits do not matter because the position class already removes those links.
// dom_lo / mesh_ny: array row li of an image is mesh row li - dom_lo of a mesh_ny-row mesh
// (dom_lo = 0, mesh_ny = ny except for a row slab, whose array is a window with halo rows).
static __global__ void k_phase_codes(const uint8_t *__restrict__ pix, int W, int ampX, int ampY,
                              int nx, int nxt, int ny, int rows, int dom_lo, int mesh_ny,
                              uint16_t *__restrict__ code)
{
    const size_t n = (size_t)nx * rows;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(p / nx), j = (int)(p % nx);
        const int li = i % ny;                       // row inside its image's array
        const int gi = li - dom_lo;                  // mesh row
        if (gi < 0 || gi >= mesh_ny || j >= nxt) { code[p] = 0; continue; }
        int jw = j > 0 ? j - 1 : j, je = j < nxt - 1 ? j + 1 : j;
        int is = (gi < mesh_ny - 1 && li < ny - 1) ? i + 1 : i;
        int in = (gi > 0 && li > 0) ? i - 1 : i;
        unsigned c = (cell_pixel(pix, W, ampX, ampY, ny, i, j) >= 150) ? 1u : 0u;
        c |= (cell_pixel(pix, W, ampX, ampY, ny, i, jw) >= 150) ? 2u : 0u;
        c |= (cell_pixel(pix, W, ampX, ampY, ny, i, je) >= 150) ? 4u : 0u;
        c |= (cell_pixel(pix, W, ampX, ampY, ny, is, j) >= 150) ? 8u : 0u;
        c |= (cell_pixel(pix, W, ampX, ampY, ny, in, j) >= 150) ? 16u : 0u;
        const unsigned cls = (unsigned)(pos_class(gi, mesh_ny) * 3 + pos_class(j, nxt));
        code[p] = (uint16_t)((1u + cls * 32u + c) * 8u);
    }
}

// ------------------------------------------------------------- assembly ---

struct CoefSoA {
    double *a0, *aW, *aE, *aS, *aN, *b;
};

// General assembly from a per-cell D array (DiscretizeMatrix2D cuh:815-902;
// with Grid != nullptr, DiscretizeMatrix2D_ImpSolid cuh:715-812: Grid 1 or 2
// gets the identity row, cuh:750-752).
// dom_lo / mesh_ny as in k_phase_codes: array row li of an image is mesh row li - dom_lo of a
// mesh_ny-row mesh (a row slab's array is a window with halo rows; rows outside the mesh and the
// pad column get the identity row and stay 0).
static __global__ void k_assemble_from_D(const double *__restrict__ D, const unsigned int *__restrict__ Grid,
                                  int nx, int nxt, int ny, int rows, int dom_lo, int mesh_ny, double dx, double dy,
                                  double CL, double CR, CoefSoA c)
{
    const size_t n = (size_t)nx * rows;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (size_t)gridDim.x * blockDim.x) {
        const int li = (int)(p / nx) % ny, j = (int)(p % nx);   // li: row inside its image's array
        const int gi = li - dom_lo;                            // mesh row
        FvmRow r;
        if (j >= nxt || gi < 0 || gi >= mesh_ny || (Grid != nullptr && (Grid[p] == 1 || Grid[p] == 2))) {
            r.a0 = 1; r.aW = 0; r.aE = 0; r.aS = 0; r.aN = 0; r.b = 0;
        } else {
            // clamped neighbour reads; values at clamped positions are unused
            double Dp = D[p];
            double Dw = D[j > 0 ? p - 1 : p];
            double De = D[j < nxt - 1 ? p + 1 : p];
            double Ds = D[(gi < mesh_ny - 1 && li < ny - 1) ? p + nx : p];
            double Dn = D[(gi > 0 && li > 0) ? p - nx : p];
            r = fvm_row(Dp, Dw, De, Ds, Dn, pos_class(j, nxt), pos_class(gi, mesh_ny), dx, dy, CL, CR);
        }
        c.a0[p] = r.a0; c.aW[p] = r.aW; c.aE[p] = r.aE; c.aS[p] = r.aS; c.aN[p] = r.aN; c.b[p] = r.b;
    }
}

// AoS [cells][5] chunk -> SoA planes (import of a host-assembled matrix).  `first`, `count` count
// mesh cells (rows of nxt); the planes are nx wide.
__device__ __forceinline__ size_t padded_index(size_t cell, int nx, int nxt)
{
    return nx == nxt ? cell : (cell / (size_t)nxt) * (size_t)nx + cell % (size_t)nxt;
}

static __global__ void k_import_aos(const double *__restrict__ A, size_t first, size_t count, int nx, int nxt, CoefSoA c)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < count;
         q += (size_t)gridDim.x * blockDim.x) {
        const double *a = A + q * 5;
        size_t p = padded_index(first + q, nx, nxt);
        c.a0[p] = a[0]; c.aW[p] = a[1]; c.aE[p] = a[2]; c.aS[p] = a[3]; c.aN[p] = a[4];
    }
}

static __global__ void k_export_aos(double *__restrict__ A, size_t first, size_t count, int nx, int nxt, CoefSoA c)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < count;
         q += (size_t)gridDim.x * blockDim.x) {
        double *a = A + q * 5;
        size_t p = padded_index(first + q, nx, nxt);
        a[0] = c.a0[p]; a[1] = c.aW[p]; a[2] = c.aE[p]; a[3] = c.aS[p]; a[4] = c.aN[p];
    }
}

// identity rows in the pad column of explicit planes (imported systems)
static __global__ void k_pad_identity(int nx, int nxt, int rows, CoefSoA c)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    for (int j = nxt; j < nx; ++j) {
        const size_t p = (size_t)i * nx + j;
        c.a0[p] = 1; c.aW[p] = 0; c.aE[p] = 0; c.aS[p] = 0; c.aN[p] = 0; c.b[p] = 0;
    }
}

// c0 = w / A0: the reference evaluates w/A[p*5+0] first and multiplies the
// result by (b - sigma) (cuh:89, C precedence), so hoisting the division out
// of the sweep keeps every bit.
static __global__ void k_make_c0(const double *__restrict__ a0, double w, double *__restrict__ c0, size_t n)
{
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (size_t)gridDim.x * blockDim.x)
        c0[p] = w / a0[p];
}

// ---------------------------------------------------------------- field ---

// Linear ramp, cuh:1955-1959: (double)j/nx*(CR-CL)+CL; `contracted` = the product fused into the
// add, as in the contracted build of the reference's expression (see kernels_sweep.hpp, FMA).
static __global__ void k_init_linear(double *__restrict__ x, int nx, int nxt, int rows, double CL, double CR, int contracted)
{
    const size_t n = (size_t)nx * rows;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (size_t)gridDim.x * blockDim.x) {
        int j = (int)(p % nx);
        if (j >= nxt) { x[p] = 0.0; continue; }
        x[p] = contracted ? __builtin_fma((double)j / nxt, (CR - CL), CL) : (double)j / nxt * (CR - CL) + CL;
    }
}

// Wall fluxes of every row, cuh:1256-1257.  The host adds them up in row order
// (cuh:1258-1259) so Deff has the reference's summation order; the transfer is
// 16*ny bytes per check instead of the reference's whole field (cuh:1245).
static __global__ void k_wall_flux(const double *__restrict__ x, const double *__restrict__ Dl,
                            const double *__restrict__ Dr, int nx, int nxt, int rows, double dx,
                            double CL, double CR, double *__restrict__ mf)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    mf[i] = Dl[i] * (x[(size_t)i * nx] - CL) / (dx / 2.0);
    mf[rows + i] = Dr[i] * (CR - x[(size_t)i * nx + nxt - 1]) / (dx / 2.0);
}

// Device-side sums of the wall fluxes (cuh:1258-1259), one WAVE per image, q[img] = {Q1, Q2}.
//   TREE = false: lane 0 adds the image's ny values in row order -- the reference's order, bit-identical to the host sum;
//   TREE = true : lane l adds rows l, l+64, ..., then a fixed butterfly over the 64 partial sums (wavefront-level
//                 reduction): deterministic, ~1e-16 relative from the serial order, ny/64 dependent adds instead of ny.
template <bool TREE>
static __global__ void k_flux_sum(const double *__restrict__ mf, int rows, int ny, int nimg, double *__restrict__ q)
{
    const int img = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (img >= nimg) return;
    const double *L = mf + (size_t)img * ny, *R = mf + rows + (size_t)img * ny;
    double q1 = 0, q2 = 0;
    if constexpr (TREE) {
        for (int j = lane; j < ny; j += 64) { q1 += L[j]; q2 += R[j]; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            q1 += __shfl_xor(q1, off, 64);
            q2 += __shfl_xor(q2, off, 64);
        }
    } else {
        if (lane != 0) return;
        for (int j = 0; j < ny; ++j) { q1 += L[j]; q2 += R[j]; }
    }
    if (lane == 0) { q[2 * img] = q1; q[2 * img + 1] = q2; }
}

}  // namespace deff
