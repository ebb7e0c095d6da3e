// ubench.hip -- instruction-rate microbenchmarks on gfx950 (diagnostics for DESIGN.md's kernel model):
// how many clocks a wave64 FP64 add / mul / fma, a DPP move and a ds_read_b64 occupy their unit.
// Each kernel runs `iters` iterations of 8 independent chains per lane; all waves of the chip busy.
// Build (the box has hipcc; the binary travels with the snapshot):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench/ubench tools/ubench/ubench.hip
// (-ffp-contract=off matters for k_level: contracted, a level is 14 FP64 instructions instead of 22 and costs 71-86 clocks
// instead of 109-219; DESIGN.md quotes the uncontracted figures).   ./ubench      instruction rates
//                                                                   ./ubench mem  HBM streams by read / write mix
//                                                                   ./ubench clk  shader clock under FP64 / LDS load
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_valu(double *out, int iters, double a, double b)
{
    double v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (OP == 0) { v0 += a; v1 += a; v2 += a; v3 += a; v4 += a; v5 += a; v6 += a; v7 += a; }
            if (OP == 1) { v0 *= b; v1 *= b; v2 *= b; v3 *= b; v4 *= b; v5 *= b; v6 *= b; v7 *= b; }
            if (OP == 2) {
                v0 = __builtin_fma(v0, b, a); v1 = __builtin_fma(v1, b, a); v2 = __builtin_fma(v2, b, a); v3 = __builtin_fma(v3, b, a);
                v4 = __builtin_fma(v4, b, a); v5 = __builtin_fma(v5, b, a); v6 = __builtin_fma(v6, b, a); v7 = __builtin_fma(v7, b, a);
            }
            if (OP == 3) {   // one dependent chain only: exposes the latency of a dependent FP64 op
                v0 += a; v0 *= b; v0 += a; v0 *= b; v0 += a; v0 *= b; v0 += a; v0 *= b;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

__global__ __launch_bounds__(256) void k_dpp(int *out, int iters)
{
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v0 = __builtin_amdgcn_update_dpp(0, v0, 0x138, 0xf, 0xf, true); v1 = __builtin_amdgcn_update_dpp(0, v1, 0x130, 0xf, 0xf, true);
            v2 = __builtin_amdgcn_update_dpp(0, v2, 0x138, 0xf, 0xf, true); v3 = __builtin_amdgcn_update_dpp(0, v3, 0x130, 0xf, 0xf, true);
            v4 = __builtin_amdgcn_update_dpp(0, v4, 0x138, 0xf, 0xf, true); v5 = __builtin_amdgcn_update_dpp(0, v5, 0x130, 0xf, 0xf, true);
            v6 = __builtin_amdgcn_update_dpp(0, v6, 0x138, 0xf, 0xf, true); v7 = __builtin_amdgcn_update_dpp(0, v7, 0x130, 0xf, 0xf, true);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

// 8 ds_read_b64 per unrolled step at lane-dependent addresses (MODE 0: consecutive 8-B words, MODE 1:
// a pseudo-random row of a 512-row table, like the row dictionary; MODE 2: all lanes the same word)
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(double *out, int iters)
{
    __shared__ double tab[8 * 520];
    for (int i = threadIdx.x; i < 8 * 520; i += 256) tab[i] = i;
    __syncthreads();
    unsigned idx = MODE == 0 ? (threadIdx.x & 63) : (MODE == 1 ? (threadIdx.x * 2654435761u >> 23) & 511 : 7);
    double s = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double *p = tab + idx;
            double a0 = p[0], a1 = p[520], a2 = p[1040], a3 = p[1560], a4 = p[2080], a5 = p[2600], a6 = p[3120], a7 = p[3640];
            s += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
            idx = (idx + (unsigned)(s == 12345.0)) & 511;      // keeps the loads in the loop, never true
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// VGPR bank test: v_mul_f64 / v_fma_f64 with explicit registers.  MODE 0: the two (three) source pairs
// sit in different register banks (pair index mod 2 differs: v[8:9] x v[10:11]); MODE 1: all sources
// in the same banks (v[8:9] x v[12:13]); destinations rotate over v[40..55].
template <int MODE, int FMA3>
__global__ __launch_bounds__(256) void k_banks(double *out, int iters)
{
    double acc = 0;
    asm volatile("v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n"
                 "v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"
                 "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n"
                 ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21");
    for (int i = 0; i < iters; ++i) {
        if (FMA3 == 0 && MODE == 0)
            asm volatile("v_mul_f64 v[40:41], v[8:9], v[10:11]\n v_mul_f64 v[42:43], v[12:13], v[14:15]\n"
                         "v_mul_f64 v[44:45], v[8:9], v[14:15]\n v_mul_f64 v[46:47], v[12:13], v[10:11]\n"
                         "v_mul_f64 v[48:49], v[8:9], v[10:11]\n v_mul_f64 v[50:51], v[12:13], v[14:15]\n"
                         "v_mul_f64 v[52:53], v[8:9], v[14:15]\n v_mul_f64 v[54:55], v[12:13], v[10:11]\n"
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (FMA3 == 0 && MODE == 1)
            asm volatile("v_mul_f64 v[40:41], v[8:9], v[12:13]\n v_mul_f64 v[42:43], v[12:13], v[16:17]\n"
                         "v_mul_f64 v[44:45], v[8:9], v[16:17]\n v_mul_f64 v[46:47], v[12:13], v[20:21]\n"
                         "v_mul_f64 v[48:49], v[8:9], v[12:13]\n v_mul_f64 v[50:51], v[12:13], v[16:17]\n"
                         "v_mul_f64 v[52:53], v[8:9], v[16:17]\n v_mul_f64 v[54:55], v[12:13], v[20:21]\n"
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (FMA3 == 1 && MODE == 0)   // three sources, banks (0,1) (2,3) (0,1)
            asm volatile("v_fma_f64 v[40:41], v[8:9], v[10:11], v[12:13]\n v_fma_f64 v[42:43], v[12:13], v[14:15], v[8:9]\n"
                         "v_fma_f64 v[44:45], v[8:9], v[14:15], v[16:17]\n v_fma_f64 v[46:47], v[12:13], v[10:11], v[20:21]\n"
                         "v_fma_f64 v[48:49], v[8:9], v[10:11], v[12:13]\n v_fma_f64 v[50:51], v[12:13], v[14:15], v[8:9]\n"
                         "v_fma_f64 v[52:53], v[8:9], v[14:15], v[16:17]\n v_fma_f64 v[54:55], v[12:13], v[10:11], v[20:21]\n"
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (FMA3 == 1 && MODE == 1)   // three sources, all in banks (0,1)
            asm volatile("v_fma_f64 v[40:41], v[8:9], v[12:13], v[16:17]\n v_fma_f64 v[42:43], v[12:13], v[16:17], v[20:21]\n"
                         "v_fma_f64 v[44:45], v[8:9], v[16:17], v[20:21]\n v_fma_f64 v[46:47], v[12:13], v[20:21], v[8:9]\n"
                         "v_fma_f64 v[48:49], v[8:9], v[12:13], v[16:17]\n v_fma_f64 v[50:51], v[12:13], v[16:17], v[20:21]\n"
                         "v_fma_f64 v[52:53], v[8:9], v[16:17], v[20:21]\n v_fma_f64 v[54:55], v[12:13], v[20:21], v[8:9]\n"
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    }
    asm volatile("v_mov_b32 %0, v40" : "=v"(*(int *)&acc));
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// One sweep level of the temporally blocked kernel, in isolation: per lane two cells, each
//   sigma = aW*xw; sigma += aE*xe; sigma += aS*xs; sigma += aN*xn; o = omw*xc + c0*(0 - sigma)
// (11 FP64 instructions, one dependent chain per cell) + 4 DPP moves for the W/E neighbours, no
// global memory.  LUT = 0: coefficients in registers; LUT = 1: the 10 coefficients come from LDS by
// per-lane row offsets that change every level (32 rows, conflict-free like the commonest rows of
// the dictionary).  8 levels per iteration, each level feeding the next like the register pipeline.
template <int LUT>
__global__ __launch_bounds__(256) void k_level(double *out, int iters, double omw)
{
    __shared__ double tab[6 * 520];
    for (int i = threadIdx.x; i < 6 * 520; i += 256) tab[i] = 1.0 / (1.0 + (i % 97));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned cw = (unsigned)(lane * 2654435761u);
    double2 vN = make_double2(0.3 + lane, 0.7), vC = make_double2(0.5, 0.25 + lane), vS = make_double2(0.1, 0.9);
    const double c0r = 0.61, aWr = -0.21, aEr = -0.23, aSr = -0.27, aNr = -0.29;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            int lo = __double2loint(vC.y), hi = __double2hiint(vC.y);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
            const double xw0 = __hiloint2double(hi, lo);
            lo = __double2loint(vC.x); hi = __double2hiint(vC.x);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
            const double xe1 = __hiloint2double(hi, lo);
            double c00 = c0r, aW0 = aWr, aE0 = aEr, aS0 = aSr, aN0 = aNr, c01 = c0r, aW1 = aWr, aE1 = aEr, aS1 = aSr, aN1 = aNr;
            if (LUT) {
                cw = cw * 1664525u + 1013904223u;
                const unsigned off0 = ((cw >> 8) & 31u) * 8u, off1 = ((cw >> 20) & 31u) * 8u;
                const char *b0 = reinterpret_cast<const char *>(tab) + off0, *b1 = reinterpret_cast<const char *>(tab) + off1;
                c00 = *reinterpret_cast<const double *>(b0);            c01 = *reinterpret_cast<const double *>(b1);
                aW0 = *reinterpret_cast<const double *>(b0 + 4160);     aW1 = *reinterpret_cast<const double *>(b1 + 4160);
                aE0 = *reinterpret_cast<const double *>(b0 + 2 * 4160); aE1 = *reinterpret_cast<const double *>(b1 + 2 * 4160);
                aS0 = *reinterpret_cast<const double *>(b0 + 3 * 4160); aS1 = *reinterpret_cast<const double *>(b1 + 3 * 4160);
                aN0 = *reinterpret_cast<const double *>(b0 + 4 * 4160); aN1 = *reinterpret_cast<const double *>(b1 + 4 * 4160);
            }
            double s0 = aW0 * xw0, s1 = aW1 * vC.x;
            s0 += aE0 * vC.y; s1 += aE1 * xe1;
            s0 += aS0 * vS.x; s1 += aS1 * vS.y;
            s0 += aN0 * vN.x; s1 += aN1 * vN.y;
            double2 o;
            o.x = omw * vC.x + c00 * (0.0 - s0);
            o.y = omw * vC.y + c01 * (0.0 - s1);
            vN = vC; vC = vS; vS = o;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = vS.x + vS.y + vC.x + vN.y + (double)cw;
}

// ---- HBM streams by read / write mix (./ubench mem): what the memory system gives a kernel that reads NR 8-byte streams
// and writes NW, 16 bytes per lane per access, grid-stride -- the ceiling a sweep kernel of the same mix can be held
// against (the explicit sweep reads 7 and writes 1, the matrix-free one reads 1.25 and writes 1).
template <int NR, int NW>
__global__ __launch_bounds__(256) void k_stream(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n2, size_t plane2)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        double2 a = make_double2(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double2 v = in[(size_t)r * plane2 + i];
            a.x += v.x; a.y += v.y;
        }
        if constexpr (NW > 0) {
#pragma unroll
            for (int w = 0; w < NW; ++w) out[(size_t)w * plane2 + i] = a;
        } else {
            if (a.x == 1.2345e300) out[i] = a;                      // never: keeps the loads
        }
    }
}

template <class F>
static double time_ms(F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

// ---- shader clock under load (./ubench clk): clock64() (shader cycles) against wall_clock64() (100 MHz) around a long
// FP64 / LDS loop -- do the SIMDs really run at the 2.4 GHz that "clocks per instruction" figures assume?
template <int MODE>
__global__ __launch_bounds__(256) void k_clk(double *out, unsigned long long *stamps, int iters, double a)
{
    __shared__ double tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = 1.0 / (1.0 + i);
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE >= 1) {                                            // FP64 adds / muls on VGPR pairs, 8 chains
            x0 = x0 * a + x4; x1 = x1 * a + x5; x2 = x2 * a + x6; x3 = x3 * a + x7;
            x4 = x4 * a + x0; x5 = x5 * a + x1; x6 = x6 * a + x2; x7 = x7 * a + x3;
        }
        if (MODE == 2) {                                            // + LDS reads
            x0 += tab[(threadIdx.x * 7 + i) & 2047]; x1 += tab[(threadIdx.x * 13 + i) & 2047];
        }
        if (MODE == 0) __builtin_amdgcn_s_sleep(8);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

static int clk_main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out; unsigned long long *st;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
    CHECK(hipMalloc(&st, sizeof(unsigned long long) * 2 * cus * 8));
    std::vector<unsigned long long> h(2 * cus * 8);
    printf("%s: %d CUs, clockRate %.0f MHz; shader MHz = clock64 delta / (wall_clock64 delta / 100 MHz)\n", prop.gcnArchName, cus, prop.clockRate * 1e-3);
    auto run = [&](const char *name, auto kern, int wg_per_cu, int iters) {
        const int blocks = cus * wg_per_cu;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0000001);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
        double lo = 1e9, hi = 0, sum = 0, us = 0;
        for (int b = 0; b < blocks; ++b) {
            const double mhz = (double)h[2 * b] / ((double)h[2 * b + 1] / 100.0);
            lo = mhz < lo ? mhz : lo; hi = mhz > hi ? mhz : hi; sum += mhz; us += h[2 * b + 1] / 100.0;
        }
        printf("  %-26s %d wave/SIMD: %7.0f MHz mean (%.0f .. %.0f), %.0f us per workgroup\n", name, wg_per_cu, sum / blocks, lo, hi, us / blocks);
    };
    for (int w : {1, 2, 4}) {
        run("idle (s_sleep)", k_clk<0>, w, 20000);
        run("FP64 mul+add", k_clk<1>, w, 400000);
        run("FP64 mul+add + LDS reads", k_clk<2>, w, 400000);
    }
    CHECK(hipFree(out)); CHECK(hipFree(st));
    return 0;
}

static int mem_main(int side)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const size_t n = (size_t)side * side, n2 = n / 2;
    double2 *in, *out;
    CHECK(hipMalloc(&in, sizeof(double) * n * 7));
    CHECK(hipMalloc(&out, sizeof(double) * n * 2));
    CHECK(hipMemset(in, 0, sizeof(double) * n * 7));
    CHECK(hipMemset(out, 0, sizeof(double) * n * 2));
    printf("%s: HBM streams over %d^2 doubles per plane (%.0f MB), 16 B per lane per access\n", prop.gcnArchName, side, n * 8e-6);
    auto run = [&](const char *name, auto kern, int nr, int nw) {
        for (int blocks : {4096, 16384}) {
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, n2, n2); });
                if (ms < best) best = ms;
            }
            printf("  %-28s %5d blocks: %8.2f us  %7.0f GB/s\n", name, blocks, best * 1e3, (nr + nw) * 8.0 * n / (best * 1e-3) / 1e9);
        }
    };
    run("read 1", k_stream<1, 0>, 1, 0);
    run("read 7", k_stream<7, 0>, 7, 0);
    run("read 1 write 1 (copy)", k_stream<1, 1>, 1, 1);
    run("read 2 write 1", k_stream<2, 1>, 2, 1);
    run("read 7 write 1 (explicit)", k_stream<7, 1>, 7, 1);
    run("read 1 write 2", k_stream<1, 2>, 1, 2);
    CHECK(hipFree(in)); CHECK(hipFree(out));
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "mem")) return mem_main(argc > 2 ? atoi(argv[2]) : 4096);
    if (argc > 1 && !strcmp(argv[1], "clk")) return clk_main();
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", prop.gcnArchName, cus, ghz);
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
    const int iters = 20000;
    for (int wg_per_cu : {1, 2, 3, 4}) {                       // 4 / 8 / 16 waves per CU = 1 / 2 / 4 per SIMD
        const int blocks = cus * wg_per_cu;
        const double waves_per_simd = wg_per_cu;            // 256 threads = 4 waves = 1 per SIMD
        auto report = [&](const char *name, double ms, double instr_per_iter) {
            // wave-instructions issued per SIMD = waves_per_simd * iters * instr_per_iter
            const double clks = ms * 1e-3 * ghz * 1e9;
            printf("  %-34s %d wave/SIMD: %8.3f ms  %6.2f clocks per wave-instruction per SIMD\n", name, wg_per_cu, ms,
                   clks / (waves_per_simd * iters * instr_per_iter));
        };
        report("v_add_f64 (8 chains)", time_ms([&] { hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5, 1.0000001); }), 32);
        report("v_mul_f64 (8 chains)", time_ms([&] { hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5, 1.0000001); }), 32);
        report("v_fma_f64 (8 chains)", time_ms([&] { hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5, 1.0000001); }), 32);
        report("add/mul f64, ONE dependent chain", time_ms([&] { hipLaunchKernelGGL(k_valu<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5, 1.0000001); }), 32);
        report("v_mul_f64 2 VGPR sources, banks differ", time_ms([&] { hipLaunchKernelGGL((k_banks<0, 0>), dim3(blocks), dim3(256), 0, 0, out, iters * 4); }), 32);
        report("v_mul_f64 2 VGPR sources, same banks", time_ms([&] { hipLaunchKernelGGL((k_banks<1, 0>), dim3(blocks), dim3(256), 0, 0, out, iters * 4); }), 32);
        report("v_fma_f64 3 VGPR sources, 2 banks", time_ms([&] { hipLaunchKernelGGL((k_banks<0, 1>), dim3(blocks), dim3(256), 0, 0, out, iters * 4); }), 32);
        report("v_fma_f64 3 VGPR sources, same banks", time_ms([&] { hipLaunchKernelGGL((k_banks<1, 1>), dim3(blocks), dim3(256), 0, 0, out, iters * 4); }), 32);
        report("TB sweep level, coefficients in registers (clocks per level)", time_ms([&] { hipLaunchKernelGGL(k_level<0>, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 1.0 / 3.0); }), 2);
        report("TB sweep level, 10 LDS lookups          (clocks per level)", time_ms([&] { hipLaunchKernelGGL(k_level<1>, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 1.0 / 3.0); }), 2);
        report("v_mov_b32_dpp wave_shr/shl", time_ms([&] { hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(256), 0, 0, (int *)out, iters); }), 32);
        report("ds_read_b64 consecutive", time_ms([&] { hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 0, 0, out, iters); }), 32);
        report("ds_read_b64 random rows", time_ms([&] { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 0, 0, out, iters); }), 32);
        report("ds_read_b64 broadcast", time_ms([&] { hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(256), 0, 0, out, iters); }), 32);
    }
    CHECK(hipFree(out));
    return 0;
}
