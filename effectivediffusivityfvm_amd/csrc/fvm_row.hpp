// fvm_row.hpp -- one row of the 5-point FVM matrix, host + device.
//
// Follows DiscretizeMatrix2D (reference Deff2DGPU/Deff2D.cuh:815-902) for a
// single cell: the x-direction block cuh:849-873 (left wall / right wall /
// interior) and the y-direction block cuh:875-897 (top / bottom / interior).
// Left and right walls are Dirichlet at half a cell (conductance D*dy/(dx/2),
// RHS C*D*dy/(dx/2)); top and bottom are zero-flux.  The expression order is
// the reference's, so with FP contraction off (this whole library is built
// with -ffp-contract=off) each coefficient is the same IEEE-754 double.
//
// The same routine feeds (a) the general device assembly from a D array and
// (b) the host-side builder of the matrix-free lookup tables, which is why it
// takes the five diffusivities as scalars instead of indexing D itself.
#pragma once
#include <hip/hip_runtime.h>

namespace deff {

// position classes: 0 interior, 1 first (left / top), 2 last (right / bottom)
enum : int { POS_INTERIOR = 0, POS_FIRST = 1, POS_LAST = 2 };

// WeightedHarmonicMean, cuh:347-360.  x == 0 gives w/0 = +inf and H = 0.
__host__ __device__ inline double whm(double w1, double w2, double x1, double x2)
{
    return (w1 + w2) / (w1 / x1 + w2 / x2);
}

struct FvmRow {
    double a0, aW, aE, aS, aN, b;
};

__host__ __device__ inline FvmRow fvm_row(double Dp, double Dw, double De, double Ds, double Dn,
                                          int xpos, int ypos, double dx, double dy,
                                          double CL, double CR)
{
    FvmRow r;
    r.a0 = 0; r.aW = 0; r.aE = 0; r.aS = 0; r.aN = 0; r.b = 0;
    double dxw, dxe, dys, dyn, kw, ke, ks, kn;
    if (xpos == POS_FIRST) {                       // cuh:849-856
        dxe = dx;
        ke = whm(dxe / 2, dxe / 2, Dp, De);
        dxw = dx / 2;
        kw = Dp;
        r.aE = -ke * dy / dxe;
        r.a0 += (ke * dy / dxe + kw * dy / dxw);
        r.b += CL * kw * dy / dxw;
    } else if (xpos == POS_LAST) {                 // cuh:857-864
        dxw = dx;
        kw = whm(dxw / 2, dxw / 2, Dp, Dw);
        dxe = dx / 2;
        ke = Dp;
        r.aW = -kw * dy / dxw;
        r.a0 += (ke * dy / dxe + kw * dy / dxw);
        r.b += CR * ke * dy / dxe;
    } else {                                       // cuh:865-873
        dxw = dx;
        kw = whm(dxw / 2, dxw / 2, Dp, Dw);
        dxe = dx;
        ke = whm(dxe / 2, dxe / 2, Dp, De);
        r.aW = -kw * dy / dxw;
        r.aE = -ke * dy / dxe;
        r.a0 += (ke * dy / dxe + kw * dy / dxw);
    }
    if (ypos == POS_FIRST) {                       // cuh:875-881
        dys = dy;
        ks = whm(dys / 2, dys / 2, Ds, Dp);
        r.aS = -ks * dx / dys;
        r.a0 += (ks * dx / dys);
    } else if (ypos == POS_LAST) {                 // cuh:882-888
        dyn = dy;
        kn = whm(dyn / 2, dyn / 2, Dp, Dn);
        r.aN = -kn * dx / dyn;
        r.a0 += kn * dx / dyn;
    } else {                                       // cuh:889-897
        dyn = dy;
        kn = whm(dyn / 2, dyn / 2, Dp, Dn);
        dys = dy;
        ks = whm(dys / 2, dys / 2, Ds, Dp);
        r.aS = -ks * dx / dys;
        r.aN = -kn * dx / dyn;
        r.a0 += (kn * dx / dyn + ks * dx / dys);
    }
    return r;
}

__host__ __device__ inline int pos_class(int idx, int count)
{
    return idx == 0 ? POS_FIRST : (idx == count - 1 ? POS_LAST : POS_INTERIOR);
}

}  // namespace deff
