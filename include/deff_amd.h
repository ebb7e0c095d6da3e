/*
 * deff_amd.h -- C ABI of the MI355X-native effective-diffusivity hot path.
 *
 * One shared library, libdeff_amd.so (HIP, gfx950).  Plain pointers and sizes
 * only; no C++ or torch types cross this boundary.  The reference has no FFI:
 * its seam is function-level inside one translation unit, so
 * each entry point below cites the reference function (Deff2DGPU/Deff2D.cuh,
 * "cuh:line") whose work it takes over.  A C++ mirror with the reference's
 * own names and argument lists (DiscretizeMatrix2D, initializeGPU, JacobiGPU,
 * ...) sits on top of this ABI in
 * effectivediffusivityfvm_amd/csrc/reference_seam.hpp; see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns DEFF_OK (0) or a negative DEFF_E* code and never
 *     blocks on stdin (the reference calls getchar() on failure, cuh:912-916);
 *     deff_last_error() gives the message of the calling thread's last failure;
 *   - a context is bound to one device and one (nx, ny) mesh, owns every device
 *     buffer and one HIP stream, and is reused across images (the reference
 *     re-allocates and resets the device per image, cuh:2038, cuh:1015);
 *   - a context is not thread-safe; use one per host thread / per GPU;
 *   - host arrays are row-major, cell p = i*nx + j, exactly as in the
 *     reference; "AoS" coefficient arrays are [n][5] = P,W,E,S(row+1),N(row-1).
 */
#ifndef DEFF_AMD_H
#define DEFF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEFF_OK            0
#define DEFF_EINVAL       -1   /* bad argument / call order */
#define DEFF_EHIP         -2   /* a HIP runtime call failed */
#define DEFF_ENOMEM       -3   /* device or host allocation failed */
#define DEFF_ENODEV       -4   /* no usable gfx950 device */
#define DEFF_ESTATE       -5   /* system / field not set before solve */
#define DEFF_ECOMM        -6   /* RCCL failure (row-slab mode) */

/* sweep kernel selection, deff_set_kernel() */
#define DEFF_KERNEL_AUTO       0
#define DEFF_KERNEL_EXPLICIT   1   /* SoA coefficient streams, 64 B/cell/sweep */
#define DEFF_KERNEL_SCALAR     2   /* 1 cell/thread, any nx; correctness fallback */
#define DEFF_KERNEL_MATFREE    3   /* rows from the dictionary by a 16-bit code, 18 B/cell/sweep */
#define DEFF_KERNEL_MATFREE_TB 4   /* matrix-free, several sweeps per HBM pass */

typedef struct deff_ctx deff_ctx;

typedef struct deff_result {
    int64_t iters;      /* sweeps executed: 10000k+1 or max_iter (cuh:1289, return value) */
    int64_t checks;     /* convergence checks performed */
    double  deff_raw;   /* Deff at the LAST CHECK, not normalised by Df (cuh:1309) */
    double  conv;       /* last signed relative change (cuh:1275) */
    double  loop_ms;    /* hipEvent time of the sweep loop, same window as cuh:1230-1298 */
} deff_result;

/* ---- library ---------------------------------------------------------- */
const char *deff_version(void);
const char *deff_last_error(void);
const char *deff_error_string(int code);
int deff_device_count(int *count);

/* ---- lifecycle: replaces initializeGPU cuh:904-981 / unInitializeGPU cuh:983-1021 */
int deff_create(int device, int nx, int ny, deff_ctx **out);
/* dataset-generation mode (BatchSim cuh:1843-2054): `nimg` images of the same nx x ny mesh are
 * held and swept together as one stacked domain (image k = rows [k*ny, (k+1)*ny)); every
 * array argument of the calls below then covers the whole stack, image after image.  The
 * zero-flux top/bottom boundaries keep the images uncoupled, so each image's numbers are
 * those of a one-at-a-time run. */
int deff_create_batch(int device, int nx, int ny, int nimg, deff_ctx **out);
int deff_batch_size(const deff_ctx *ctx, int *nimg);
/* how many slots a stack context for `images` images of an nx x ny mesh should have on `device` (dataset generation:
 * small images that fit one workgroup tile get one slot per CU and stay resident; otherwise ~16-64 Mi cells per stack) */
int deff_recommended_batch(int device, int nx, int ny, int64_t images, int *slots);
int deff_destroy(deff_ctx *ctx);
int deff_mesh(const deff_ctx *ctx, int *nx, int *ny, double *dx, double *dy);   /* meshInfo cuh:54-61 */
int deff_set_kernel(deff_ctx *ctx, int kernel);
int deff_get_kernel(const deff_ctx *ctx, int *kernel_in_use);
/* tuning knob; 0 restores the default.  Keys: "rows_explicit", "rows_matfree", "wg_matfree",
 * "nt_explicit", "serpentine", "tb_T" (sweeps per pass: 1,2,4,6,8), "tb_LY" (rows per chunk), "tb_wg",
 * "tb_xmajor", "tb_wall_halo", "tb_ranked" (streaming kernel: chunk heights by the service order of a SIMD's waves, 1 default;
 * weights "tb_rank_w0", "tb_rank_w1", "tb_rank_w2", "tb_rank_wall" per mille), "tb_tall_deal" / "tb_sym_age" (resident tiles: rows
 * dealt by the waves' age, 1 default), "dict" (harvest a row dictionary from explicit systems: 1 default),
 * "tb_impl" (1 streaming, 2 workgroup tiles), "tb_R", "tb_NW" (8 / 12 / 16 waves per
 *   tile: 12 = link-symmetric matrix rows in registers, 16 = tall resident tiles),
 * "tb_launch" (workgroup tiles whose tiles all fit the chip
 *   run every pass between two checks in ONE launch, neighbouring tiles synchronised by flags: 1 = one launch per
 *   pass instead; a resident launch that cannot make progress -- another process holds part of the GPU -- gives up
 *   after a bounded wait, the interval is redone with one launch per pass and the context stays in that mode:
 *   deff_get_plan "tb_fallbacks"), "flux_reduce", and
 *   "fma" = 1: contracted arithmetic -- the reference's expressions (cuh:74-89, cuh:1957) with each
 *   product fused into the following add, as nvcc's default -fmad=true / gcc -ffp-contract=fast compile
 *   them; bit-identical to the oracle's fma build, not to the default (written-order) arithmetic.
 *   Set it before deff_init_linear(); it applies to every sweep kernel. */
int deff_set_tuning(deff_ctx *ctx, const char *key, int value);
/* what the last launch plan of the temporally blocked kernel chose: "tb_T", "tb_LY" (rows per chunk),
 * "tb_strips", "tb_chunks_per_image", "tb_blocks" (workgroups launched), "tb_impl", "tb_R", "tb_resident" (1: the
 * passes of a batch run as one resident launch), "tb_fallbacks" (resident intervals that gave up and were redone with
 * one launch per pass); 0 before any sweep */
int deff_get_plan(deff_ctx *ctx, const char *key, int *value);

/* ---- image -> phases: replaces the mask->D loops cuh:1988-2000 (2-phase),
 *      cuh:1518-1529 (3-phase) and the synthetic generator of SURVEY.md 8d */
int deff_set_image(deff_ctx *ctx, const uint8_t *pix, int W, int H, int ampX, int ampY);
int deff_synth_image(deff_ctx *ctx, uint64_t seed, uint64_t img);   /* generated on the device; a stack holds images img, img+1, ... */
int deff_get_image(deff_ctx *ctx, uint8_t *pix);                    /* W*H bytes back */

/* grayscale JPEG file -> bytes: replaces readImage cuh:327-345 (stbi_load(..., 1)); decodes to the
 * same bytes as stb_image v2.26 for one-component files, Huffman-coded baseline, extended sequential
 * and progressive (images up to 2^28 pixels; PNG / BMP get a message); *pix is malloc'ed, release it
 * with deff_free(); host code, needs no context */
int deff_load_jpeg_gray(const char *path, uint8_t **pix, int *W, int *H, int *nChannels);
void deff_free(void *p);

/* ---- assembly: replaces DiscretizeMatrix2D cuh:815-902 (+ WeightedHarmonicMean cuh:347-360) */
/* native 2-phase path: image already on the device, coefficients never leave it */
int deff_assemble_2phase(deff_ctx *ctx, double Ds, double Df, double CL, double CR);
/* drop-in path: caller supplies the per-cell diffusivity D[n] (host); optional
 * Grid[n] selects DiscretizeMatrix2D_ImpSolid cuh:715-812 semantics (NULL = plain) */
int deff_assemble_from_D(deff_ctx *ctx, const double *D, const unsigned int *Grid,
                         double CL, double CR);
/* 3-phase path (SingleSim3Phase cuh:1509-1586): D from pixels (> 200 solid, < 50 gas, else
 * fluid, cuh:1518-1529) on the device + DiscretizeMatrix2D_ImpSolid with Grid (NULL = plain) */
int deff_assemble_3phase(deff_ctx *ctx, double Ds, double Df, double Dg, const unsigned int *Grid,
                         double CL, double CR);
/* FloodFill cuh:557-713 on the host, linear time: Grid[n] (1 = solid) gets 2 where the pore
 * space is not connected to the left wall; *path_flag = PathFlag.  Needs no context. */
int deff_flood_fill(unsigned int *Grid, int nx, int ny, int *path_flag);
/* host-assembled system as the reference passes it to JacobiGPU (cuh:1163):
 * A[n*5] AoS, b[n], D[n] (only its first and last column are read, cuh:1256-1257) */
int deff_set_system(deff_ctx *ctx, const double *A, const double *b, const double *D,
                    double CL, double CR);
/* assembled coefficients back in the reference's AoS layout (parity tests, drop-in) */
int deff_get_system(deff_ctx *ctx, double *A, double *b);

/* ---- field ------------------------------------------------------------- */
int deff_init_linear(deff_ctx *ctx, double CL, double CR);          /* cuh:1955-1959 */
int deff_set_field(deff_ctx *ctx, const double *x);                 /* H2D of the guess, cuh:1203 */
int deff_get_field(deff_ctx *ctx, double *x);                       /* final D2H, cuh:1300 */

/* ---- solve: replaces JacobiGPU cuh:1163-1314 / JacobiGPUPreCond cuh:1024-1160
 * and the kernels updateX_SOR cuh:69-92 (omega = 2/3) / updateX_V1 cuh:96-118 (omega = 1).
 * Stopping rule exactly as cuh:1232-1290: check when iter % check_every == 0
 * (including iter 0), deffOld seeded with 5, stop when |change| <= tol or
 * iter == max_iter.  MFL/MFR (ny doubles each, may be NULL) receive the wall
 * fluxes of the last check (cuh:1256-1257). */
int deff_solve(deff_ctx *ctx, double omega, double tol, int64_t max_iter, int64_t check_every,
               deff_result *out, double *MFL, double *MFR);
/* same loop for a batch context: out[nimg]; each image stops by its own rule (its sweeps
 * end, its field is frozen) while the others continue; MFL/MFR hold nimg*ny values */
int deff_solve_batch(deff_ctx *ctx, double omega, double tol, int64_t max_iter, int64_t check_every,
                     deff_result *out, double *MFL, double *MFR);
/* streaming batch (dataset generation): the nimg slots of a batch context are kept full -- when a
 * slot's image stops (its own rule), its result is reported and the slot is refilled with the next
 * image, which enters one sweep before a check of the running ones so that every image keeps the
 * reference's schedule (checks after its own sweeps 1, C+1, 2C+1, ...).  2-phase native system.
 *   next(user, slot, pix, &image_id) fills W*H bytes: 1 = image provided, 0 = no more, <0 = error
 *   done(user, image_id, slot, result): called once per image; deff_get_slot_field(ctx, slot, x)
 *   may be called from inside it to fetch the image's final field */
typedef int (*deff_next_image_fn)(void *user, int slot, uint8_t *pix, int64_t *image_id);
typedef void (*deff_image_done_fn)(void *user, int64_t image_id, int slot, const deff_result *res);
int deff_solve_stream(deff_ctx *ctx, int W, int H, int ampX, int ampY, double Ds, double Df, double CL,
                      double CR, double omega, double tol, int64_t max_iter, int64_t check_every,
                      deff_next_image_fn next, deff_image_done_fn done, void *user);
int deff_get_slot_field(deff_ctx *ctx, int slot, double *x /* nx*ny */);
/* optional observer called on the host after every convergence check with
 * (iter of the checked sweep, Deff, signed change): what the reference prints under
 * Verbose (cuh:1267-1271).  NULL removes it. */
typedef void (*deff_progress_fn)(int64_t iter, double deff_raw, double change, void *user);
int deff_set_progress(deff_ctx *ctx, deff_progress_fn fn, void *user);
/* building blocks, also used by bench.py: n sweeps without a check (ms = hipEvent
 * time on the context's stream), and one flux / Deff evaluation (cuh:1252-1263) */
int deff_sweeps(deff_ctx *ctx, int64_t n, double omega, float *ms);
int deff_flux(deff_ctx *ctx, double *deff_raw /* [nimg] */, double *MFL, double *MFR);
/* Residual() cuh:451-494 of the current field: r[k] = mean over the cells of image k of |qW - qE + qN - qS| (the reference
 * defines it and leaves its two call sites, cuh:1121 and cuh:1266, commented out).  Every cell's term is the reference's
 * arithmetic; the sum is a wavefront-level reduction in a fixed order (deterministic; ~1e-16 relative from the reference's
 * serial row-major order).  deff_residual: systems assembled from the image (deff_assemble_2phase / _3phase; no D plane is
 * read: 9 B per cell); deff_residual_D: any diffusivity plane D[ny*nx] (host), the reference's own call shape.  Both may be
 * called from a deff_set_progress() callback.  *ms (may be NULL) = device time of the reduction.  Not for slab contexts. */
int deff_residual(deff_ctx *ctx, double *r /* [nimg] */, float *ms);
/* ... of ONE image of a stack, wherever its newest field lives: callable from the deff_image_done_fn callback of a stream */
int deff_residual_slot(deff_ctx *ctx, int slot, double *r);
int deff_residual_D(deff_ctx *ctx, const double *D, double CL, double CR, double *r /* [nimg] */, float *ms);
/* sweep-kernel launches issued by the last deff_sweeps()/deff_solve() and the sweeps one
 * temporally blocked launch performs (1 for the single-sweep kernels) */
int deff_last_launches(const deff_ctx *ctx, int64_t *launches, int *sweeps_per_pass);

/* ---- row slabs: ONE image split over several GPUs (BASELINE config #4; nothing like it in the
 * reference, which is pinned to device 0, cuh:908).  Slab r owns a contiguous block of rows and
 * keeps 8 halo rows on each side; one neighbour exchange per temporally blocked pass keeps them
 * valid; results are bit-identical to the one-GPU path.  This group form drives all slabs from
 * one process (devices[r] may repeat: N slabs on one GPU is how the path is tested on one GPU). */
typedef struct deff_slab_group deff_slab_group;
int deff_slab_group_create(int nslabs, const int *devices, int nx, int NY, deff_slab_group **out);
int deff_slab_group_destroy(deff_slab_group *g);
int deff_slab_group_layout(const deff_slab_group *g, int *first_row, int *row_count);
int deff_slab_group_set_tuning(deff_slab_group *g, const char *key, int value);
/* deff_get_plan() of slab `slab`: every slab of an image plans the same sweeps-per-pass ("tb_T") */
int deff_slab_group_get_plan(deff_slab_group *g, int slab, const char *key, int *value);
int deff_slab_group_set_image(deff_slab_group *g, const uint8_t *pix /* NY*nx */);
int deff_slab_group_synth_image(deff_slab_group *g, uint64_t seed, uint64_t img);
int deff_slab_group_assemble_2phase(deff_slab_group *g, double Ds, double Df, double CL, double CR);
/* 3-phase system (deff_assemble_3phase per slab); Grid = flood-fill result of the whole image, NY*nx, or NULL */
int deff_slab_group_assemble_3phase(deff_slab_group *g, double Ds, double Df, double Dg, const unsigned int *Grid,
                                    double CL, double CR);
int deff_slab_group_init_linear(deff_slab_group *g, double CL, double CR);
int deff_slab_group_set_field(deff_slab_group *g, const double *x /* NY*nx */);
int deff_slab_group_get_field(deff_slab_group *g, double *x /* NY*nx */);
int deff_slab_group_sweeps(deff_slab_group *g, int64_t n, double omega, float *ms);
int deff_slab_group_flux(deff_slab_group *g, double *deff_raw, double *MFL, double *MFR);
int deff_slab_group_solve(deff_slab_group *g, double omega, double tol, int64_t max_iter,
                          int64_t check_every, deff_result *out, double *MFL, double *MFR);

/* ---- row slabs, one process per GPU: same slabs, RCCL transport (grouped ncclSend/ncclRecv of the
 * 8-row halo blocks between neighbour ranks once per blocked pass; ncclAllGather of the per-row
 * wall fluxes at a check).  Rank 0 makes the 128-byte id with deff_rccl_unique_id() and hands it
 * to the others (torch.distributed broadcast, a file, MPI ...).  solve/sweeps are collective. */
typedef struct deff_slab_rank deff_slab_rank;
int deff_rccl_unique_id(char *id128);
int deff_slab_rank_create(int device, int nx, int NY, int rank, int nranks, const char *id128,
                          deff_slab_rank **out);
/* the same slab with a caller-supplied, host-staged transport instead of RCCL (testing, portability):
 *   exchange(user, send_up, recv_up, send_down, recv_down, count): swap `count` doubles with the
 *     rank above (NULL pointers on the first rank) and below (NULL on the last); 0 = ok
 *   allgather(user, mine, all, count): all[r*count ..] = rank r's `mine`; 0 = ok */
typedef int (*deff_host_exchange_fn)(void *user, const double *send_up, double *recv_up, const double *send_down,
                                     double *recv_down, size_t count);
typedef int (*deff_host_allgather_fn)(void *user, const double *mine, double *all, size_t count);
int deff_slab_rank_create_custom(int device, int nx, int NY, int rank, int nranks, deff_host_exchange_fn exchange,
                                 deff_host_allgather_fn allgather, void *user, deff_slab_rank **out);
int deff_slab_rank_destroy(deff_slab_rank *s);
int deff_slab_rank_layout(const deff_slab_rank *s, int *first_row, int *row_count);   /* rows it owns */
int deff_slab_rank_window(const deff_slab_rank *s, int *first_row, int *row_count);   /* rows it holds */
int deff_slab_rank_context(deff_slab_rank *s, deff_ctx **ctx);   /* for set_tuning / assemble_2phase / init_linear */
/* deff_set_tuning() of the slab's context, plus "slab_overlap": the halo exchange of a pass runs on a second stream while
 * the interior of the slab is still being swept -- 0 = never (pass, exchange, pass on one stream), 1 = default: for slabs
 * of 16 Mi cells and more, 2 = always */
int deff_slab_rank_set_tuning(deff_slab_rank *s, const char *key, int value);
int deff_slab_rank_set_image_window(deff_slab_rank *s, const uint8_t *pix_window);
/* 3-phase system of this rank's slab; Grid_window = the rows deff_slab_rank_window() names, or NULL */
int deff_slab_rank_assemble_3phase(deff_slab_rank *s, double Ds, double Df, double Dg, const unsigned int *Grid_window,
                                   double CL, double CR);
int deff_slab_rank_synth_image(deff_slab_rank *s, uint64_t seed, uint64_t img);
int deff_slab_rank_get_field(deff_slab_rank *s, double *x_own);
int deff_slab_rank_sweeps(deff_slab_rank *s, int64_t n, double omega, float *ms);
int deff_slab_rank_solve(deff_slab_rank *s, double omega, double tol, int64_t max_iter,
                         int64_t check_every, deff_result *out, double *MFL, double *MFR);

/* diagnostics: per wave tile of one temporally blocked pass of the streaming kernel two words -- the wall-clock (100 MHz) start,
 * and duration (low 32 bits) | HW_ID[15:0] << 32 | XCC_ID << 48 (where it ran); resident tiles: 12 wall-clock stamps per tile */
int deff_debug_tb_stamps(deff_ctx *ctx, double omega, unsigned long long *out, int *ntiles);

/* raw device pointers for zero-copy interop (torch tensors, RCCL): current field, and the byte
 * pitch between rows -- nx*8 for an even nx, (nx+1)*8 for an odd one (device rows are padded to an
 * even number of cells; the pad cell holds 0 and is not part of the mesh) */
int deff_device_field(deff_ctx *ctx, void **d_x, size_t *row_pitch_bytes);
int deff_synchronize(deff_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* DEFF_AMD_H */
