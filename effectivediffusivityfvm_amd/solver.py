"""Host-side wrapper of one solver context (one GPU, one mesh).

Thin: every method is one call into the C ABI (include/deff_amd.h).  Array
conventions are the reference's: row-major (ny, nx) fields, A as (n, 5) =
P, W, E, S(row+1), N(row-1) (Deff2D.cuh:815-902).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import KERNEL_NAMES, DeffError, Result, check

OMEGA_REFERENCE = 2.0 / 3.0     # updateX_SOR, Deff2D.cuh:72


class SolveResult:
    __slots__ = ("iters", "checks", "deff_raw", "conv", "loop_ms", "MFL", "MFR", "field")

    def __repr__(self):
        return (f"SolveResult(iters={self.iters}, checks={self.checks}, deff_raw={self.deff_raw!r}, "
                f"conv={self.conv!r}, loop_ms={self.loop_ms:.3f})")


def recommended_batch(nx, ny, images, device=0):
    """Slots a stack Solver(nx, ny, nimg=...) should have to solve `images` images (deff_recommended_batch)."""
    n = C.c_int()
    check(_capi.load().deff_recommended_batch(int(device), int(nx), int(ny), int(images), C.byref(n)))
    return n.value


class Solver:
    """One context: `nimg` images of an nx x ny mesh (nimg > 1 = dataset-generation batch,
    arrays are then stacked image after image: shape (nimg*ny, nx))."""

    def __init__(self, nx, ny, device=0, kernel="auto", nimg=1):
        self._L = _capi.load()
        self._ctx = C.c_void_p()
        self.nx, self.ny, self.nimg = int(nx), int(ny), int(nimg)
        self.rows = self.ny * self.nimg
        check(self._L.deff_create_batch(int(device), self.nx, self.ny, self.nimg, C.byref(self._ctx)))
        if kernel != "auto":
            self.set_kernel(kernel)

    # -- lifecycle --------------------------------------------------------
    def close(self):
        if self._ctx:
            self._L.deff_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_kernel(self, kernel):
        k = KERNEL_NAMES[kernel] if isinstance(kernel, str) else int(kernel)
        check(self._L.deff_set_kernel(self._ctx, k))

    def kernel_in_use(self):
        k = C.c_int()
        check(self._L.deff_get_kernel(self._ctx, C.byref(k)))
        return {v: n for n, v in KERNEL_NAMES.items()}[k.value]

    def set_tuning(self, key, value):
        check(self._L.deff_set_tuning(self._ctx, key.encode(), int(value)))

    def plan(self):
        """What the temporally blocked kernel's last launch plan chose (zeros before any sweep)."""
        out = {}
        for key in ("tb_T", "tb_LY", "tb_strips", "tb_chunks_per_image", "tb_blocks", "tb_impl", "tb_R", "tb_NW", "tb_resident", "tb_sym", "tb_ranked", "tb_aged"):
            v = C.c_int()
            check(self._L.deff_get_plan(self._ctx, key.encode(), C.byref(v)))
            out[key] = v.value
        return out

    def plan_value(self, key):
        """One key of deff_get_plan, e.g. "tb_fallbacks" (resident intervals that were redone with one launch per pass)."""
        v = C.c_int()
        check(self._L.deff_get_plan(self._ctx, key.encode(), C.byref(v)))
        return v.value

    # -- image / assembly -------------------------------------------------
    def set_image(self, pix, ampX=1, ampY=1):
        pix = np.ascontiguousarray(pix, dtype=np.uint8)
        if pix.ndim == 3:                                   # (nimg, H, W)
            assert pix.shape[0] == self.nimg
            H, W = pix.shape[1:]
        else:                                               # (nimg*H, W)
            assert pix.shape[0] % self.nimg == 0
            H, W = pix.shape[0] // self.nimg, pix.shape[1]
        check(self._L.deff_set_image(self._ctx, pix.reshape(-1, W), W, H, ampX, ampY))

    def synth_image(self, seed=12345, img=0):
        check(self._L.deff_synth_image(self._ctx, seed, img))

    def get_image(self):
        out = np.empty((self.rows, self.nx), dtype=np.uint8)   # only valid for amp 1
        check(self._L.deff_get_image(self._ctx, out))
        return out

    def assemble_2phase(self, Ds, Df, CL, CR):
        check(self._L.deff_assemble_2phase(self._ctx, Ds, Df, CL, CR))

    def assemble_3phase(self, Ds, Df, Dg, CL, CR, grid=None):
        g = None
        if grid is not None:
            grid = np.ascontiguousarray(grid, dtype=np.uint32)
            assert grid.size == self.nx * self.rows
            g = grid.ctypes.data_as(C.c_void_p)
        check(self._L.deff_assemble_3phase(self._ctx, Ds, Df, Dg, g, CL, CR))

    def assemble_from_D(self, D, CL, CR, grid=None):
        D = np.ascontiguousarray(D, dtype=np.float64)
        g = None
        if grid is not None:
            grid = np.ascontiguousarray(grid, dtype=np.uint32)
            g = grid.ctypes.data_as(C.c_void_p)
        check(self._L.deff_assemble_from_D(self._ctx, D, g, CL, CR))

    def set_system(self, A, b, D, CL, CR):
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        d = None
        if D is not None:
            D = np.ascontiguousarray(D, dtype=np.float64)
            d = D.ctypes.data_as(C.c_void_p)
        check(self._L.deff_set_system(self._ctx, A, b, d, CL, CR))

    def get_system(self):
        n = self.nx * self.rows
        A = np.empty((n, 5), dtype=np.float64)
        b = np.empty(n, dtype=np.float64)
        check(self._L.deff_get_system(self._ctx, A, b))
        return A, b

    # -- field ------------------------------------------------------------
    def init_linear(self, CL, CR):
        check(self._L.deff_init_linear(self._ctx, CL, CR))

    def set_field(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.nx * self.rows
        check(self._L.deff_set_field(self._ctx, x))

    def get_field(self):
        x = np.empty((self.rows, self.nx), dtype=np.float64)
        check(self._L.deff_get_field(self._ctx, x))
        return x

    # -- solve ------------------------------------------------------------
    def solve(self, tol, max_iter, omega=OMEGA_REFERENCE, check_every=10000, fluxes=True):
        """One image: a SolveResult.  Batch: a list with one SolveResult per image (MFL/MFR
        are that image's rows; fluxes=False passes NULL for them, as a caller that only wants Deff does)."""
        res = (Result * self.nimg)()
        MFL = np.zeros(self.rows)
        MFR = np.zeros(self.rows)
        check(self._L.deff_solve_batch(self._ctx, omega, tol, int(max_iter), int(check_every), res,
                                       MFL.ctypes.data_as(C.c_void_p) if fluxes else None,
                                       MFR.ctypes.data_as(C.c_void_p) if fluxes else None))
        outs = []
        for k in range(self.nimg):
            out = SolveResult()
            out.iters, out.checks = res[k].iters, res[k].checks
            out.deff_raw, out.conv, out.loop_ms = res[k].deff_raw, res[k].conv, res[k].loop_ms
            out.MFL = MFL[k * self.ny:(k + 1) * self.ny]
            out.MFR = MFR[k * self.ny:(k + 1) * self.ny]
            outs.append(out)
        return outs[0] if self.nimg == 1 else outs

    def solve_stream(self, images, Ds, Df, CL, CR, tol, max_iter, omega=OMEGA_REFERENCE, check_every=10000,
                     want_fields=False, ampX=1, ampY=1):
        """Dataset generation: `images` is any iterable of uint8 (H, W) arrays of one size; the nimg
        slots of this context are kept full (a finished image's slot is refilled with the next one).
        Returns a list with one SolveResult per image, in input order (plus .field when want_fields)."""
        it = iter(images)
        H, W = self.ny // ampY, self.nx // ampX
        results = {}
        counter = [0]
        errors = []

        def _next(_user, _slot, pix_ptr, id_ptr):
            try:
                img = next(it)
            except StopIteration:
                return 0
            except Exception as e:          # noqa: BLE001 - reported to the C side as an error code
                errors.append(e)
                return -1
            a = np.ascontiguousarray(img, dtype=np.uint8)
            if a.shape != (H, W):
                errors.append(ValueError(f"image {counter[0]} is {a.shape}, expected {(H, W)}"))
                return -1
            C.memmove(pix_ptr, a.ctypes.data, a.size)
            id_ptr[0] = counter[0]
            counter[0] += 1
            return 1

        def _done(_user, image_id, slot, res_ptr):
            # an exception raised inside a ctypes callback is swallowed: record it, re-raise after the call
            try:
                r = res_ptr[0]
                out = SolveResult()
                out.iters, out.checks, out.deff_raw, out.conv, out.loop_ms = r.iters, r.checks, r.deff_raw, r.conv, r.loop_ms
                out.MFL = out.MFR = None
                if want_fields:
                    x = np.empty((self.ny, self.nx), dtype=np.float64)
                    check(self._L.deff_get_slot_field(self._ctx, slot, x))
                    out.field = x
                results[int(image_id)] = out
            except Exception as e:          # noqa: BLE001
                errors.append(e)

        nxt, dn = _capi.NEXT_IMAGE_FN(_next), _capi.IMAGE_DONE_FN(_done)
        rc = self._L.deff_solve_stream(self._ctx, W, H, ampX, ampY, Ds, Df, CL, CR, omega, tol, int(max_iter),
                                       int(check_every), nxt, dn, None)
        if errors:
            raise errors[0]
        check(rc)
        return [results[k] for k in range(counter[0])]

    def sweeps(self, n, omega=OMEGA_REFERENCE):
        ms = C.c_float()
        check(self._L.deff_sweeps(self._ctx, int(n), omega, C.byref(ms)))
        return ms.value

    def flux(self):
        d = (C.c_double * self.nimg)()
        MFL = np.zeros(self.rows)
        MFR = np.zeros(self.rows)
        check(self._L.deff_flux(self._ctx, d, MFL.ctypes.data_as(C.c_void_p), MFR.ctypes.data_as(C.c_void_p)))
        return (d[0] if self.nimg == 1 else np.array(d[:])), MFL, MFR

    def residual(self, D=None, CL=None, CR=None, timing=False):
        """Residual() cuh:451-494 of the current field (mean |qW - qE + qN - qS| per cell): a float, or one per image.
        D = None: the system assembled from the image; otherwise the diffusivity plane (rows, nx) with the wall values."""
        r = (C.c_double * self.nimg)()
        ms = C.c_float()
        if D is None:
            check(self._L.deff_residual(self._ctx, r, C.byref(ms)))
        else:
            D = np.ascontiguousarray(D, dtype=np.float64)
            assert D.size == self.nx * self.rows
            check(self._L.deff_residual_D(self._ctx, D, float(CL), float(CR), r, C.byref(ms)))
        out = r[0] if self.nimg == 1 else np.array(r[:])
        return (out, ms.value) if timing else out

    def set_progress(self, fn):
        """fn(iter, deff_raw, change) after every convergence check, or None."""
        if fn is None:
            self._progress = _capi.PROGRESS_FN(0)
        else:
            self._progress = _capi.PROGRESS_FN(lambda it, d, ch, _u: fn(it, d, ch))
        check(self._L.deff_set_progress(self._ctx, self._progress, None))

    def last_launches(self):
        """(kernel launches of the last sweeps()/solve(), sweeps per temporally blocked launch)."""
        n = C.c_int64()
        t = C.c_int()
        check(self._L.deff_last_launches(self._ctx, C.byref(n), C.byref(t)))
        return n.value, t.value

    def device_field_ptr(self):
        p = C.c_void_p()
        pitch = C.c_size_t()
        check(self._L.deff_device_field(self._ctx, C.byref(p), C.byref(pitch)))
        return p.value, pitch.value

    def synchronize(self):
        check(self._L.deff_synchronize(self._ctx))


class SlabGroup:
    """One image split into row slabs over `devices` (one slab per entry; entries may repeat,
    which is how the multi-GPU path is exercised on a single GPU).  Same call sequence as Solver."""

    def __init__(self, nx, NY, devices):
        self._L = _capi.load()
        self._g = C.c_void_p()
        self.nx, self.ny, self.nslabs = int(nx), int(NY), len(devices)
        dev = (C.c_int * len(devices))(*devices)
        check(self._L.deff_slab_group_create(len(devices), dev, self.nx, self.ny, C.byref(self._g)))

    def close(self):
        if self._g:
            self._L.deff_slab_group_destroy(self._g)
            self._g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def layout(self):
        a = (C.c_int * self.nslabs)()
        b = (C.c_int * self.nslabs)()
        check(self._L.deff_slab_group_layout(self._g, a, b))
        return list(a), list(b)

    def set_tuning(self, key, value):
        check(self._L.deff_slab_group_set_tuning(self._g, key.encode(), int(value)))

    def plans(self):
        """Every slab's last temporally blocked launch plan (Solver.plan() per slab)."""
        out = []
        for r in range(self.nslabs):
            d = {}
            for key in ("tb_T", "tb_LY", "tb_strips", "tb_chunks_per_image", "tb_blocks", "tb_impl", "tb_R", "tb_NW", "tb_resident", "tb_sym"):
                v = C.c_int()
                check(self._L.deff_slab_group_get_plan(self._g, r, key.encode(), C.byref(v)))
                d[key] = v.value
            out.append(d)
        return out

    def set_image(self, pix):
        pix = np.ascontiguousarray(pix, dtype=np.uint8)
        assert pix.shape == (self.ny, self.nx)
        check(self._L.deff_slab_group_set_image(self._g, pix))

    def synth_image(self, seed=12345, img=0):
        check(self._L.deff_slab_group_synth_image(self._g, seed, img))

    def assemble_2phase(self, Ds, Df, CL, CR):
        check(self._L.deff_slab_group_assemble_2phase(self._g, Ds, Df, CL, CR))

    def assemble_3phase(self, Ds, Df, Dg, CL, CR, grid=None):
        """grid: the whole image's flood-fill result (NY, nx) uint32, or None."""
        g = None
        if grid is not None:
            grid = np.ascontiguousarray(grid, dtype=np.uint32)
            assert grid.shape == (self.ny, self.nx)
            g = grid.ctypes.data_as(C.c_void_p)
        check(self._L.deff_slab_group_assemble_3phase(self._g, Ds, Df, Dg, g, CL, CR))

    def init_linear(self, CL, CR):
        check(self._L.deff_slab_group_init_linear(self._g, CL, CR))

    def set_field(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.nx * self.ny
        check(self._L.deff_slab_group_set_field(self._g, x))

    def get_field(self):
        x = np.empty((self.ny, self.nx), dtype=np.float64)
        check(self._L.deff_slab_group_get_field(self._g, x))
        return x

    def sweeps(self, n, omega=OMEGA_REFERENCE):
        ms = C.c_float()
        check(self._L.deff_slab_group_sweeps(self._g, int(n), omega, C.byref(ms)))
        return ms.value

    def flux(self):
        d = C.c_double()
        MFL = np.zeros(self.ny)
        MFR = np.zeros(self.ny)
        check(self._L.deff_slab_group_flux(self._g, C.byref(d), MFL.ctypes.data_as(C.c_void_p),
                                           MFR.ctypes.data_as(C.c_void_p)))
        return d.value, MFL, MFR

    def solve(self, tol, max_iter, omega=OMEGA_REFERENCE, check_every=10000):
        res = Result()
        MFL = np.zeros(self.ny)
        MFR = np.zeros(self.ny)
        check(self._L.deff_slab_group_solve(self._g, omega, tol, int(max_iter), int(check_every), C.byref(res),
                                            MFL.ctypes.data_as(C.c_void_p), MFR.ctypes.data_as(C.c_void_p)))
        out = SolveResult()
        out.iters, out.checks = res.iters, res.checks
        out.deff_raw, out.conv, out.loop_ms = res.deff_raw, res.conv, res.loop_ms
        out.MFL, out.MFR = MFL, MFR
        return out


def rccl_unique_id():
    """128-byte RCCL id made on rank 0; hand it to every rank's SlabRank."""
    buf = C.create_string_buffer(128)
    check(_capi.load().deff_rccl_unique_id(buf))
    return buf.raw


class TorchDistTransport:
    """Host-staged slab transport over a torch.distributed group of CPU tensors (gloo): the halo
    blocks and the flux vectors arrive as host buffers, neighbours swap them with isend/irecv.
    For tests and for boxes without a working RCCL path between the ranks; RCCL is the fast one."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.rank, self.nranks = dist.get_rank(group), dist.get_world_size(group)

    def _peer(self, r):
        return r if self._group is None else self._dist.get_global_rank(self._group, r)

    def exchange(self, send_up, recv_up, send_down, recv_down):
        t, d = self._torch, self._dist
        ops, keep = [], []
        for send, recv, peer in ((send_up, recv_up, self.rank - 1), (send_down, recv_down, self.rank + 1)):
            if send is None:
                continue
            st, rt = t.from_numpy(send.copy()), t.from_numpy(recv)
            keep += [st, rt]
            ops += [d.P2POp(d.isend, st, self._peer(peer), self._group), d.P2POp(d.irecv, rt, self._peer(peer), self._group)]
        for w in (d.batch_isend_irecv(ops) if ops else []):
            w.wait()

    def allgather(self, mine, out):
        t, d = self._torch, self._dist
        parts = [t.from_numpy(out[r]) for r in range(self.nranks)]
        d.all_gather(parts, t.from_numpy(mine.copy()), group=self._group)


class SlabRank:
    """This process's slab of one image split over `nranks` processes (one GPU each); halo
    exchange and flux all-gather go through RCCL (`unique_id`), or through `transport` (an object
    with exchange()/allgather() like TorchDistTransport) when unique_id is None.  solve() and
    sweeps() are collective."""

    def __init__(self, nx, NY, rank, nranks, unique_id=None, device=0, transport=None):
        self._L = _capi.load()
        self._s = C.c_void_p()
        self.nx, self.ny, self.rank, self.nranks = int(nx), int(NY), int(rank), int(nranks)
        if unique_id is not None:
            check(self._L.deff_slab_rank_create(int(device), self.nx, self.ny, self.rank, self.nranks, unique_id,
                                                C.byref(self._s)))
        else:
            if transport is None:
                raise ValueError("SlabRank needs an RCCL unique_id or a transport")
            self._transport, self._cb_error = transport, None

            def as_np(p, n):
                return np.ctypeslib.as_array(p, shape=(n,)) if p else None

            def xchg(_user, su, ru, sd, rd, count):
                try:
                    transport.exchange(as_np(su, count), as_np(ru, count), as_np(sd, count), as_np(rd, count))
                    return 0
                except Exception as e:                         # noqa: BLE001 -- reported through the C error path
                    self._cb_error = e
                    return 1

            def gather(_user, mine, out, count):
                try:
                    transport.allgather(as_np(mine, count), np.ctypeslib.as_array(out, shape=(self.nranks, count)))
                    return 0
                except Exception as e:                         # noqa: BLE001
                    self._cb_error = e
                    return 1

            self._cbs = (_capi.HOST_EXCHANGE_FN(xchg), _capi.HOST_ALLGATHER_FN(gather))   # keep alive
            check(self._L.deff_slab_rank_create_custom(int(device), self.nx, self.ny, self.rank, self.nranks,
                                                       self._cbs[0], self._cbs[1], None, C.byref(self._s)))
        self._ctx = C.c_void_p()
        check(self._L.deff_slab_rank_context(self._s, C.byref(self._ctx)))

    def close(self):
        if self._s:
            self._L.deff_slab_rank_destroy(self._s)
            self._s = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _pair(self, fn):
        a, b = C.c_int(), C.c_int()
        check(fn(self._s, C.byref(a), C.byref(b)))
        return a.value, b.value

    def layout(self):
        """(first owned row, owned rows)"""
        return self._pair(self._L.deff_slab_rank_layout)

    def window(self):
        """(first row held incl. halo, rows held): the slice of the image set_image() takes"""
        return self._pair(self._L.deff_slab_rank_window)

    def set_tuning(self, key, value):
        check(self._L.deff_slab_rank_set_tuning(self._s, key.encode(), int(value)))

    def set_image(self, pix_full):
        a, n = self.window()
        win = np.ascontiguousarray(pix_full[a:a + n], dtype=np.uint8)
        check(self._L.deff_slab_rank_set_image_window(self._s, win))

    def synth_image(self, seed=12345, img=0):
        check(self._L.deff_slab_rank_synth_image(self._s, seed, img))

    def assemble_2phase(self, Ds, Df, CL, CR):
        check(self._L.deff_assemble_2phase(self._ctx, Ds, Df, CL, CR))

    def assemble_3phase(self, Ds, Df, Dg, CL, CR, grid_full=None):
        """grid_full: the whole image's flood-fill result (NY, nx); this rank passes its window of it."""
        g = None
        if grid_full is not None:
            a, n = self.window()
            win = np.ascontiguousarray(grid_full[a:a + n], dtype=np.uint32)
            g = win.ctypes.data_as(C.c_void_p)
        check(self._L.deff_slab_rank_assemble_3phase(self._s, Ds, Df, Dg, g, CL, CR))

    def init_linear(self, CL, CR):
        check(self._L.deff_init_linear(self._ctx, CL, CR))

    def get_field(self):
        _, own = self.layout()
        x = np.empty((own, self.nx), dtype=np.float64)
        check(self._L.deff_slab_rank_get_field(self._s, x))
        return x

    def sweeps(self, n, omega=OMEGA_REFERENCE):
        ms = C.c_float()
        check(self._L.deff_slab_rank_sweeps(self._s, int(n), omega, C.byref(ms)))
        return ms.value

    def solve(self, tol, max_iter, omega=OMEGA_REFERENCE, check_every=10000):
        res = Result()
        MFL = np.zeros(self.ny)
        MFR = np.zeros(self.ny)
        check(self._L.deff_slab_rank_solve(self._s, omega, tol, int(max_iter), int(check_every), C.byref(res),
                                           MFL.ctypes.data_as(C.c_void_p), MFR.ctypes.data_as(C.c_void_p)))
        out = SolveResult()
        out.iters, out.checks = res.iters, res.checks
        out.deff_raw, out.conv, out.loop_ms = res.deff_raw, res.conv, res.loop_ms
        out.MFL, out.MFR = MFL, MFR
        return out


def load_jpeg_gray(path):
    """Grayscale JPEG -> uint8 (H, W), decoded like the reference's stbi_load(..., 1)."""
    L = _capi.load()
    p = C.c_void_p()
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    check(L.deff_load_jpeg_gray(str(path).encode(), C.byref(p), C.byref(w), C.byref(h), C.byref(n)))
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value)).copy()
    finally:
        L.deff_free(p)


def flood_fill(grid):
    """FloodFill (Deff2D.cuh:557-713): grid (ny, nx) with 1 = solid -> (grid with unreachable
    pore cells = 2, PathFlag).  Host function of the library; needs no GPU."""
    g = np.array(grid, dtype=np.uint32, order="C", copy=True)
    ny, nx = g.shape
    flag = C.c_int()
    check(_capi.load().deff_flood_fill(g, nx, ny, C.byref(flag)))
    return g, bool(flag.value)


__all__ = ["Solver", "SlabGroup", "SlabRank", "rccl_unique_id", "flood_fill", "load_jpeg_gray", "SolveResult", "DeffError", "OMEGA_REFERENCE"]
