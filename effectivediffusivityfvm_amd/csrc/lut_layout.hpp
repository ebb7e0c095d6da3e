// lut_layout.hpp -- the row dictionary of the matrix-free kernels.
//
// A matrix-free system is (a) a table of the DISTINCT matrix rows that occur in the image,
// (b) one 16-bit code per cell naming its row.  For piecewise-constant diffusivity the number of
// distinct rows is tiny (2 phases: 32 neighbourhood patterns x 9 position classes), so the table
// lives in LDS and a sweep moves x 8 + code 2 + xNew 8 bytes per cell instead of the 64 bytes
// of explicit coefficients.  The table is either enumerated a priori (native 2-phase assembly)
// or harvested from explicit coefficient planes (kernels_dict.hpp: 3-phase / ImpSolid systems,
// host-assembled systems passed through the reference's seam).
//
// Device layout: 6 planes (c0 = w/A0, aW, aE, aS, aN, b) of LUT_PLANE_STRIDE doubles; a cell's
// code is the BYTE offset of its row inside a plane (row index x 8).  Row 0 is all zeros: cells
// outside the mesh carry code 0 and stay exactly 0.
//
// The stride is 520, not 512, on purpose: with a stride that is a multiple of 512 B hipcc fuses
// the per-cell lookups into ds_read2st64_b64, which banks on 32 banks (2-way conflicts on a
// 32-entry x 8-B group, 16 LDS cycles per instruction -- measured as THE bottleneck of the first
// temporally blocked kernel); with 4160 B neither ds_read2 form can encode the offset, the lookups
// stay plain ds_read_b64 (64 banks: 32 consecutive rows are conflict-free, 2 cycles each).
#pragma once

namespace deff {

constexpr int LUT_PLANES = 6;
constexpr int LUT_MAX_ROWS = 512;                               // including the zero row
constexpr int LUT_PLANE_STRIDE = LUT_MAX_ROWS + 8;              // doubles
constexpr int LUT_DOUBLES = LUT_PLANES * LUT_PLANE_STRIDE;      // 3120 doubles = 24.4 KiB
constexpr int LUT_NATIVE_ROWS = 1 + 9 * 32;                     // native 2-phase dictionary

}  // namespace deff
