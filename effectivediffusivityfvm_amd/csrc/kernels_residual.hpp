// kernels_residual.hpp -- the reference's Residual() (Deff2DGPU/Deff2D.cuh:451-494) on the device: the L1 norm of the
// cells' flux imbalance |qW - qE + qN - qS|, one pass over the field, wavefront-level reduction (gfx950, wave64, FP64).
//
// Per cell the arithmetic is the reference's, operation for operation (so every cell's |...| is the oracle's double):
//   interior face   q = (dy/dx * H(dx/2, dx/2, D[p], D[q])) * (c[hi] - c[lo])      cuh:469-474, 480-489
//   left wall       qW = (dy/(dx/2) * D[p]) * (c[p] - CL)                           cuh:466
//   right wall      qE = (dy/(dx/2) * D[p]) * (CR - c[p])                           cuh:470
//   top / bottom    qN = 0 / qS = 0                                                 cuh:476, 480
// (vertical faces carry the same dy/dx and dx/2 weights as horizontal ones: that is the reference's text).  H is symmetric
// in its two diffusivities bit for bit (w1 = w2, and IEEE addition commutes), so a face has ONE value: the kernel
// evaluates every face once (see k_residual_classes).
//
// Two sources of D:
//   k_residual_classes  the native systems (2 or 3 pixel classes): a cell's class comes from its pixel, the conductance
//                       (dy/dx * H) of a face from a 3 x 8 table in LDS indexed by the two classes (column 3 = wall)
//                       built on the host with the same expressions: no D plane, no division,
//                       9 B per cell of HBM (x 8 + pixel 1);
//   k_residual_plane    any D plane the caller supplies (deff_assemble_from_D / the reference's own call shape
//                       Residual(rows, cols, &opts, x, D)): one cell per thread, H evaluated in place.
// What differs from the reference is only the ORDER of the final sum (the reference adds the cells serially, row-major):
// a lane adds its cells top to bottom (first cell of its pair, then the second), the 64 lane sums of a wave are combined by a
// fixed DPP tree (row_shr 1, 2, 3, 4, 8, then row_bcast 15 and 31: lane 63 holds the wave's sum), and k_residual_final
// (one workgroup of 16 waves per image) has thread t add the waves' partial sums t, t + 1024, ... in that order, the same tree
// per wave, and thread 0 the 16 wave sums in wave order.  The order depends on the launch geometry only: same bits on every
// call.  Of the two orders the serial one is the less accurate (it drifts from the exactly added terms by up to n * 2^-53; a
// tree loses ~log2 n ulps): tests compare with the oracle's per-cell doubles added in long double (1e-13) and with its serial
// sum within that sum's own error bound (the tests' assert_residual).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fvm_row.hpp"
#include "kernels_sweep.hpp"

namespace deff {

constexpr int RES_ROWS = 8;                      // rows per wave tile (k_residual_classes)
constexpr int RES_COLS = 128;                    // columns per wave tile: 2 per lane
template <bool V> struct ResTag { static constexpr bool value = V; };

// conductances by class pair, built on the host (api_residual.hip: residual_table) with the reference's expressions
struct ResTable {
    double g[3][8];                              // g[k][q]: q < 3 face to a cell of class q; 3: wall dy/(dx/2)*D[k]; 4..7 unused
};

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);     // lanes without a source read 0
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}
// lane i <- lane i - 1 (wave_shr:1) / lane i + 1 (wave_shl:1); the lane without a source (0 / 63) keeps `edge`
template <int CTRL>
__device__ __forceinline__ double dpp_f64_keep(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32_keep(unsigned v, unsigned edge)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, CTRL, 0xf, 0xf, false);
}

// Sum of the 64 lanes' values; valid in lane 63.  Fixed order: within a row of 16 lanes prefix sums by row_shr 1, 2, 3, then
// 4 and 8; rows 1 and 3 add lane 15 of the row below (row_bcast:15), rows 2 and 3 add lane 31 (row_bcast:31).
__device__ __forceinline__ double wave_sum_to_lane63(double v)
{
    double t = v + dpp_f64<0x111, 0xf, 0xf>(v);                  // row_shr:1
    t = t + dpp_f64<0x112, 0xf, 0xf>(v);                         // row_shr:2
    t = t + dpp_f64<0x113, 0xf, 0xf>(v);                         // row_shr:3   -> t[i] = v[i-3..i]
    t = t + dpp_f64<0x114, 0xf, 0xe>(t);                         // row_shr:4, banks 1-3
    t = t + dpp_f64<0x118, 0xf, 0xc>(t);                         // row_shr:8, banks 2-3 -> lane 15 of a row = the row's sum
    t = t + dpp_f64<0x142, 0xa, 0xf>(t);                         // row_bcast:15 into rows 1 and 3
    t = t + dpp_f64<0x143, 0xc, 0xf>(t);                         // row_bcast:31 into rows 2 and 3
    return t;
}

// pixel -> class CODE = 8 x class index (0 fluid, 1 solid, 2 gas; 24 = "wall"): the byte offset of the class in a row of the
// table, and << 3 the byte offset of its own row.  PHASES = 2: cuh:1988-2000 (< 150 fluid); 3: cuh:1518-1529.
constexpr unsigned RES_WALL = 24u;
template <int PHASES>
__device__ __forceinline__ unsigned pixel_code(unsigned v)
{
    if constexpr (PHASES == 2) return v < 150u ? 0u : 8u;
    else return v > 200u ? 8u : (v < 50u ? 16u : 0u);
}

typedef unsigned int res_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int res_u2 __attribute__((ext_vector_type(2)));

// One wave per work item = `kt` consecutive tiles of RES_ROWS rows x 128 columns of one image (a column strip streamed top to
// bottom), 4 work items (vertically adjacent) per workgroup; every lane holds two cells of each row.  What keeps the
// instruction count down (the first version spent 146 VALU instructions per lane and row, 35 of them selects, and was
// issue-bound at 52 us for 4096^2):
//  - a face is evaluated ONCE: per row a lane computes W of its first cell, the face between its two cells, E of its second
//    cell and the two S faces; N is the S face of the row above, carried down the strip;
//  - rows are clamped into the image instead of being tested: the row "above" the first row is the first row itself, so its
//    face is g * (c - c) = 0 exactly as the reference's literal 0 (likewise below the last row) -- no top / bottom cases;
//  - walls are a class: the cell beyond a wall has the wall's concentration and the code RES_WALL, whose table column holds
//    dy/(dx/2) * D; lanes 0 and 63 take their outer neighbour (the next strip's cell or the wall) from a register filled
//    once per tile, through the `old` operand of the lane shift -- no per-row selects;
//  - only a strip that sticks out of the mesh on the right (EDGE) masks lanes, and that is a wave-uniform property;
//  - rows are addressed as buffer base (the image) + lane offset (one VGPR) + row offset (an SGPR): no 64-bit address
//    arithmetic in the VALU.  (An image's field therefore has to stay below 4 GiB: 512 Mi cells, checked on the host.)
// FAST = no mesh amplification and an even image width: the two pixels of a lane are one aligned 16-bit load.
// partial[img * per_img + tx * cpi + chunk] = the work item's sum (cpi = work items per strip and image).
template <int PHASES, bool FAST>
__global__ __launch_bounds__(256) void k_residual_classes(const double *__restrict__ x, const uint8_t *__restrict__ pix,
                                                          int W, int ampX, int ampY, int nx, int nxt, int ny, int nimg,
                                                          int ntx, int cpi, int kt, double CL, double CR, ResTable tab,
                                                          double *__restrict__ partial)
{
    __shared__ double g[24];                     // g[k * 8 + q]
    if (threadIdx.x < 24) g[threadIdx.x] = tab.g[threadIdx.x >> 3][threadIdx.x & 7];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gy = nimg * cpi;
    const unsigned wt = blockIdx.x * 4u + (unsigned)wave;
    if (wt >= (unsigned)ntx * (unsigned)gy) return;
    const int tx = (int)(wt / (unsigned)gy), ty = (int)(wt % (unsigned)gy);
    const int img = ty / cpi, chunk = ty - img * cpi;
    const int l_begin = chunk * RES_ROWS * kt;                   // rows [l_begin, l_end) of the image
    const int l_end = min(l_begin + RES_ROWS * kt, ny);
    const int col = tx * RES_COLS + 2 * lane;
    const int H = ny / ampY;
    const char *gb = reinterpret_cast<const char *>(g);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(x) + (size_t)img * ny * nx, 0, (int)((unsigned)ny * (unsigned)nx * 8u), 0x00020000);
    const uint8_t *pimg = pix + (size_t)img * H * W;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(pimg), 0, (int)((unsigned)H * (unsigned)W), 0x00020000);

    auto code_at = [&](int lr, int j) -> unsigned {              // class code of mesh cell (lr, j) of this image, any amplification
        return pixel_code<PHASES>(pimg[(size_t)(lr / ampY) * W + (j / ampX)]);
    };
    auto run = [&](auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const bool v0 = !EDGE || col < nxt, v1 = !EDGE || col + 1 < nxt;
        const int colx = v0 ? col : 0;
        const unsigned vox = (unsigned)colx * 8u;
        double2 xr[RES_ROWS + 2];
        unsigned e0[RES_ROWS + 2], e1[RES_ROWS + 2];
        auto load_row = [&](int li, double2 &c, unsigned &k0, unsigned &k1) __attribute__((always_inline)) {
            const int lr = min(max(li, 0), ny - 1);              // wave-uniform clamp
            const res_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vox, (int)((unsigned)lr * (unsigned)nx * 8u), 0);
            __builtin_memcpy(&c, &v, 16);
            if constexpr (FAST) {
                const unsigned two = __builtin_amdgcn_raw_buffer_load_b16(rp, colx, (int)((unsigned)lr * (unsigned)W), 0);
                k0 = pixel_code<PHASES>(two & 0xFFu);
                k1 = pixel_code<PHASES>((two >> 8) & 0xFFu);
            } else {
                k0 = code_at(lr, colx);
                k1 = code_at(lr, v1 ? colx + 1 : colx);
            }
            if constexpr (EDGE) {                                // beyond the right wall: the wall's concentration and code
                if (!v0) { c.x = CR; k0 = RES_WALL; }
                if (!v1) { c.y = CR; k1 = RES_WALL; }
            }
        };
        // lane 0's outer neighbour is cell col - 1, lane 63's cell col + 2: the next strip's cell, or the wall
        const bool west = lane == 0;
        const int jh = west ? col - 1 : col + 2;
        const bool hwall = west ? col == 0 : jh >= nxt;
        const bool hload = (lane == 0 || lane == 63) && !hwall;
        const double cwall = west ? CL : CR;
        auto face = [&](unsigned own, unsigned other, double hi, double lo) __attribute__((always_inline)) {
            return *reinterpret_cast<const double *>(gb + ((own << 3) + other)) * (hi - lo);
        };
        load_row(l_begin - 1, xr[0], e0[0], e1[0]);
        load_row(l_begin, xr[1], e0[1], e1[1]);
        double acc = 0.0;
        double fN0 = 0.0, fN1 = 0.0;
        bool first = true;
#pragma unroll 1
        for (int l = l_begin; l < l_end; l += RES_ROWS) {
#pragma unroll
            for (int q = 0; q < RES_ROWS; ++q) load_row(l + 1 + q, xr[q + 2], e0[q + 2], e1[q + 2]);
            double hx[RES_ROWS];
            unsigned hk[RES_ROWS];
#pragma unroll
            for (int q = 0; q < RES_ROWS; ++q) { hx[q] = cwall; hk[q] = RES_WALL; }
            if (hload) {
#pragma unroll
                for (int q = 0; q < RES_ROWS; ++q) {
                    const int lr = min(l + q, ny - 1);
                    const res_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rx, jh * 8, (int)((unsigned)lr * (unsigned)nx * 8u), 0);
                    __builtin_memcpy(&hx[q], &v, 8);
                    if constexpr (FAST) hk[q] = pixel_code<PHASES>(__builtin_amdgcn_raw_buffer_load_b8(rp, jh, (int)((unsigned)lr * (unsigned)W), 0) & 0xFFu);
                    else hk[q] = code_at(lr, jh);
                }
            }
            if (first) {                                         // the face above the item's first row (0 at the image's top)
                fN0 = face(e0[1], e0[0], xr[1].x, xr[0].x);
                fN1 = face(e1[1], e1[0], xr[1].y, xr[0].y);
                first = false;
            }
#pragma unroll
            for (int q = 0; q < RES_ROWS; ++q) {
                if (l + q >= l_end) break;                       // wave-uniform: the item's (or the image's) last rows
                const double2 c = xr[q + 1], cS = xr[q + 2];
                const unsigned k0 = e0[q + 1], k1 = e1[q + 1];
                const double cW = dpp_f64_keep<0x138>(c.y, hx[q]);   // wave_shr:1, lane 0 keeps its outer neighbour
                const unsigned kW = dpp_u32_keep<0x138>(k1, hk[q]);
                const double cE = dpp_f64_keep<0x130>(c.x, hx[q]);   // wave_shl:1, lane 63 keeps its outer neighbour
                const unsigned kE = dpp_u32_keep<0x130>(k0, hk[q]);
                const double fW = face(k0, kW, c.x, cW);             // cuh:466 / 472
                const double fM = face(k0, k1, c.y, c.x);            // E of the first cell = W of the second (cuh:467 / 472)
                const double fE = face(k1, kE, cE, c.y);             // cuh:470 / 473
                const double fS0 = face(k0, e0[q + 2], cS.x, c.x);   // cuh:477 / 483 (0 at the image's bottom)
                const double fS1 = face(k1, e1[q + 2], cS.y, c.y);
                const double r0 = __builtin_fabs(fW - fM + fN0 - fS0);   // cuh:488: qW - qE + qN - qS, left to right
                const double r1 = __builtin_fabs(fM - fE + fN1 - fS1);
                if (v0) acc += r0;
                if (v1) acc += r1;
                fN0 = fS0;
                fN1 = fS1;
            }
            xr[0] = xr[RES_ROWS]; e0[0] = e0[RES_ROWS]; e1[0] = e1[RES_ROWS];
            xr[1] = xr[RES_ROWS + 1]; e0[1] = e0[RES_ROWS + 1]; e1[1] = e1[RES_ROWS + 1];
        }
        const double s = wave_sum_to_lane63(acc);
        if (lane == 63) partial[(size_t)img * ((size_t)ntx * cpi) + (size_t)tx * cpi + chunk] = s;
    };
    if ((tx + 1) * RES_COLS > nxt) run(ResTag<true>{});
    else run(ResTag<false>{});
}

// Any D plane (device, nx wide like the field): one cell per thread, the reference's expressions in place.
// 256 cells of one row segment per workgroup -> 4 wave sums -> partial[img * per_img + row_in_image * segs + seg] (the four
// waves of a workgroup are added in wave order by thread 0).
__global__ __launch_bounds__(256) void k_residual_plane(const double *__restrict__ x, const double *__restrict__ D, int nx,
                                                        int nxt, int ny, int nimg, int segs, double dx, double dy, double CL,
                                                        double CR, double *__restrict__ partial)
{
    __shared__ double ws[4];
    const int seg = blockIdx.x % segs;
    const int srow = blockIdx.x / segs;                          // stacked row
    const int img = srow / ny, li = srow - img * ny;
    const int col = seg * 256 + (int)threadIdx.x;
    double r = 0.0;
    if (col < nxt) {
        const size_t p = (size_t)srow * nx + col;
        double qW, qE, qN, qS;
        if (col == 0) {
            qW = dy / (dx / 2) * D[p] * (x[p] - CL);
            qE = dy / (dx) * whm(dx / 2, dx / 2, D[p], D[p + 1]) * (x[p + 1] - x[p]);
        } else if (col == nxt - 1) {
            qW = dy / (dx) * whm(dx / 2, dx / 2, D[p], D[p - 1]) * (x[p] - x[p - 1]);
            qE = dy / (dx / 2) * D[p] * (CR - x[p]);
        } else {
            qW = dy / (dx) * whm(dx / 2, dx / 2, D[p], D[p - 1]) * (x[p] - x[p - 1]);
            qE = dy / (dx) * whm(dx / 2, dx / 2, D[p], D[p + 1]) * (x[p + 1] - x[p]);
        }
        if (li == 0) {
            qN = 0;
            qS = dy / dx * whm(dx / 2, dx / 2, D[p + nx], D[p]) * (x[p + nx] - x[p]);
        } else if (li == ny - 1) {
            qS = 0;
            qN = dy / dx * whm(dx / 2, dx / 2, D[p - nx], D[p]) * (x[p] - x[p - nx]);
        } else {
            qS = dy / dx * whm(dx / 2, dx / 2, D[p + nx], D[p]) * (x[p + nx] - x[p]);
            qN = dy / dx * whm(dx / 2, dx / 2, D[p - nx], D[p]) * (x[p] - x[p - nx]);
        }
        r = __builtin_fabs(qW - qE + qN - qS);
    }
    const double s = wave_sum_to_lane63(r);
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)img * ((size_t)ny * segs) + (size_t)li * segs + seg] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}

// (Measured and dropped: finishing the reduction inside the class kernel -- every workgroup delivers its sum through a
// device-coherent store, counts itself in with an agent-scope atomic once the store is acknowledged, and the last one of an
// image adds them up in this same fixed order: 33.6 against 32.0 us at 4096^2, 21.7 against 14.0 at 2048^2, 126.8 against 114 at
// 8192^2.  The acknowledged store and the atomic's round trip keep every workgroup's slot on its CU ~2 us longer, which costs
// more than the second launch saves.)
// One workgroup of 16 waves per image: thread t adds the partial sums t, t + 1024, ... of its image in that order, the wave
// tree combines the 64 thread sums of a wave, thread 0 adds the 16 wave sums in wave order.  out[img] = sum.
__global__ __launch_bounds__(1024) void k_residual_final(const double *__restrict__ partial, size_t per_img,
                                                         double *__restrict__ out)
{
    __shared__ double ws[16];
    const double *p = partial + (size_t)blockIdx.x * per_img;
    double acc = 0.0;
    size_t i = threadIdx.x;
    for (; i + 3 * 1024 < per_img; i += 4 * 1024) {               // four loads in flight, added in index order
        const double a = p[i], b = p[i + 1024], c = p[i + 2048], d = p[i + 3072];
        acc += a; acc += b; acc += c; acc += d;
    }
    for (; i < per_img; i += 1024) acc += p[i];
    const double s = wave_sum_to_lane63(acc);
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = ws[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) t += ws[w];
        out[blockIdx.x] = t;
    }
}

}  // namespace deff
