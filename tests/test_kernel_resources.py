"""Register / scratch / occupancy budget of the hot kernels, checked on the ISA hipcc emits for gfx950 (no GPU needed:
`-Rpass-analysis=kernel-resource-usage`).  These numbers ARE the performance model of DESIGN.md section 4 -- waves per SIMD
decide the FP64 issue rate -- and they move with innocent-looking edits (a `blockDim.x` in the dictionary load cost the
streaming kernel 22 VGPRs and one wave per SIMD; workgroup tiles with 8 rows per wave spill into the sweep loop), so they
are pinned here."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "effectivediffusivityfvm_amd", "csrc")


def remarks_of(unit):
    """The compiler's resource remarks of one translation unit: from the in-tree build if it is up to date (the Makefile keeps
    them as build/<unit>.usage.txt), else from a compile of its own."""
    kept = os.path.join(CSRC, "build", unit + ".usage.txt")
    sources = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    if os.path.exists(kept) and os.path.getsize(kept) > 0 and os.path.getmtime(kept) >= max(os.path.getmtime(f) for f in sources):
        return open(kept, errors="replace").read()
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-fno-fast-math", "-fvisibility=hidden", "-Wno-unused-function",
                        "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, unit + ".hip"],
                       cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


@pytest.fixture(scope="module")
def usage():
    out, cur = {}, None
    for line in remarks_of("api_solve").splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return out


@pytest.fixture(scope="module")
def usage_residual():
    out, cur = {}, None
    for line in remarks_of("api_residual").splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return out


def kernels(usage, prefix):
    got = {k: v for k, v in usage.items() if k.startswith(prefix)}
    assert got, prefix
    return got


def test_streaming_blocked_kernel_keeps_its_waves(usage):
    """k_sweep_matfree_tb<T, FMA, GUARD>: T = 8 and 6 at 3 waves per SIMD (<= 168 VGPRs), T = 4 at 4 (<= 128), no AGPR,
    next to no scratch (the 24 B of T = 8 sit outside the row loop)."""
    for name, u in kernels(usage, "_ZN4deff18k_sweep_matfree_tbILi8E").items():
        assert u["Occupancy"] >= 3 and u["VGPRs"] <= 168 and u["AGPRs"] == 0 and u["ScratchSize"] <= 32, (name, u)
    for name, u in kernels(usage, "_ZN4deff18k_sweep_matfree_tbILi6E").items():
        assert u["Occupancy"] >= 3 and u["ScratchSize"] == 0, (name, u)
    for name, u in kernels(usage, "_ZN4deff18k_sweep_matfree_tbILi4E").items():
        assert u["Occupancy"] >= 4 and u["VGPRs"] <= 128 and u["ScratchSize"] == 0, (name, u)
    for name, u in kernels(usage, "_ZN4deff18k_sweep_matfree_tbI").items():
        assert u["LDS"] <= 25 * 1024, (name, u)                 # dictionary only: 3-4 workgroups per CU


def test_experiment_kernels_are_not_in_the_library(usage):
    """The paired-wave form and the A/B branches of round 3 live in tools/experiments/, not in the shipped translation unit."""
    assert not [k for k in usage if "matfree_tb2" in k]


def test_workgroup_tile_kernel_fits_two_waves_per_simd(usage):
    """k_sweep_wgtile<T, R, FMA, GUARD>: 8 waves per workgroup = 2 per SIMD = 256 VGPRs; R = 4 and 6 without scratch,
    R = 7 with at most a few spilled registers (outside the sweep loop); one workgroup's LDS (dictionary + mailbox)."""
    for R, scratch in ((4, 0), (6, 0), (7, 128)):
        for T in (4, 8):
            for name, u in kernels(usage, f"_ZN4deff14k_sweep_wgtileILi{T}ELi{R}E").items():
                assert u["Occupancy"] >= 2 and u["VGPRs"] <= 256 and u["AGPRs"] == 0 and u["ScratchSize"] <= scratch, (name, u)
                assert u["LDS"] <= 64 * 1024, (name, u)


def test_resident_kernel_fits_two_waves_per_simd(usage):
    """k_sweep_wgres<T, R, FMA, GUARD> (the resident form of the same tiles): same budget; the few spilled registers of
    R = 7 are the codes during the one-time lookups and six values around the pass loop's end, not in the sweeps."""
    for R, scratch in ((4, 0), (6, 0), (7, 64)):
        for T in (4, 8):
            for name, u in kernels(usage, f"_ZN4deff13k_sweep_wgresILi{T}ELi{R}E").items():
                m = re.search(r"ELb[01]ELb([01])ELb([01])ELb[01]EEE", name)    # <.., FMA, GUARD, TALL, SYM>
                if m.group(2) == "1":
                    continue                                         # the tall form, checked below
                guard = m.group(1) == "1"                            # zero-diffusivity variant: SGPR-heavy branches
                assert u["Occupancy"] >= 2 and u["VGPRs"] <= 256 and u["AGPRs"] == 0, (name, u)
                assert u["ScratchSize"] <= (max(scratch, 64) if guard else scratch), (name, u)
                assert u["LDS"] <= 64 * 1024, (name, u)


def test_tall_resident_kernel_fits_four_waves_per_simd(usage):
    """k_sweep_wgres<8, R, FMA, GUARD, TALL=true>: 16 waves per workgroup = 4 per SIMD = 128 VGPRs.  R = 6 without scratch;
    the larger tiles spill a few field rows around the pass loop's exchange (not in the sweeps -- that is what the row
    lambda's empty asm and the codes in LDS are for: without them R = 12 needed 1 KiB of scratch and ran 12x slower);
    LDS = dictionary + 64 KiB mailbox + 256 B of codes per tile row, under the CU's 160 KiB."""
    budget = {4: 0, 5: 0, 6: 0, 7: 0, 8: 64, 9: 128, 10: 192, 11: 256, 12: 288, 13: 320, 14: 400}
    seen = set()
    for name, u in usage.items():
        m = re.match(r"_ZN4deff13k_sweep_wgresILi8ELi(\d+)ELb[01]ELb[01]ELb1ELb[01]EEE", name)
        if not m:
            continue
        R = int(m.group(1))
        seen.add(R)
        assert u["Occupancy"] >= 4 and u["VGPRs"] <= 128 and u["AGPRs"] == 0, (name, u)
        assert u["ScratchSize"] <= budget[R], (name, u)
        assert u["LDS"] <= 160 * 1024 and u["LDS"] >= 24 * 1024 + 64 * 1024 + 16 * R * 256, (name, u)
    assert seen == set(budget)


def test_aged_tall_kernels_fit_four_waves_per_simd(usage):
    """k_sweep_wgage<8, RA, RB, RC, RD, FMA, false, SYM>: four pass loops in one kernel, one per wave age; the registers and the
    scratch are those of the largest (RA rows), the LDS that of the uniform tile of (RA + RB + RC + RD) / 4 rows per wave."""
    seen = set()
    for name, u in usage.items():
        m = re.match(r"_ZN4deff13k_sweep_wgageILi8ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb[01]ELb0ELb[01]EEE", name)
        if not m:
            continue
        rows = tuple(int(v) for v in m.groups())
        seen.add(rows)
        assert u["Occupancy"] >= 4 and u["VGPRs"] <= 128 and u["AGPRs"] == 0 and u["ScratchSize"] <= 400, (name, u)
        assert u["LDS"] <= 160 * 1024 and u["LDS"] >= 24 * 1024 + 64 * 1024 + sum(rows) * 4 * 256, (name, u)
    assert seen == {(6, 6, 5, 3), (8, 8, 5, 3), (8, 8, 8, 4), (9, 9, 9, 5), (10, 9, 9, 8), (12, 12, 10, 6), (13, 13, 11, 7), (13, 13, 13, 9)}


def test_symmetric_tile_kernel_fits_three_waves_per_simd(usage):
    """k_sweep_wgsym<8, R, FMA>: 12 waves per workgroup = 3 per SIMD = 168 VGPRs, symmetric matrix rows in registers (14 VGPRs
    per tile row); R = 4, 5 without any scratch (R = 6 would spill inside the sweep loop and is not instantiated);
    LDS = dictionary + 48 KiB mailbox."""
    seen = set()
    for name, u in usage.items():
        m = re.match(r"_ZN4deff13k_sweep_wgsymILi([468])ELi(\d+)ELb[01]EEE", name)
        if not m:
            continue
        seen.add((int(m.group(1)), int(m.group(2))))
        assert u["Occupancy"] >= 3 and u["VGPRs"] <= 168 and u["AGPRs"] == 0 and u["ScratchSize"] == 0, (name, u)
        assert u["LDS"] <= 80 * 1024, (name, u)
    assert seen == {(8, 4), (8, 5), (6, 4), (6, 5), (4, 4), (4, 5)}          # passes of 8 sweeps and (round 4) of 6 and 4


def test_aged_symmetric_tile_kernels_fit_three_waves_per_simd(usage):
    """k_sweep_wgsage<8, a, b, c, FMA>: three pass loops in one kernel (one per wave age), 168 VGPRs, no scratch."""
    seen = set()
    for name, u in usage.items():
        m = re.match(r"_ZN4deff14k_sweep_wgsageILi8ELi(\d+)ELi(\d+)ELi(\d+)ELb[01]EEE", name)
        if not m:
            continue
        seen.add(tuple(int(v) for v in m.groups()))
        assert u["Occupancy"] >= 3 and u["VGPRs"] <= 168 and u["AGPRs"] == 0 and u["ScratchSize"] == 0, (name, u)
        assert u["LDS"] <= 80 * 1024, (name, u)
    assert seen == {(4, 4, 3), (5, 4, 4), (5, 5, 4)}


def test_single_sweep_kernels_are_light(usage):
    for prefix in ("_ZN4deff16k_sweep_explicitI", "_ZN4deff15k_sweep_matfreeI", "_ZN4deff14k_sweep_scalarI"):
        for name, u in kernels(usage, prefix).items():
            assert u["ScratchSize"] == 0 and u["Occupancy"] >= 4, (name, u)


def test_residual_kernels_are_light(usage_residual):
    """k_residual_classes<PHASES, FAST>: a streaming reduction that must hide HBM latency with occupancy -- at least 4 waves per
    SIMD (<= 128 VGPRs), no scratch, a 192-byte table in LDS; the plane and final kernels likewise."""
    seen = 0
    for name, u in usage_residual.items():
        if "k_residual_classes" in name:
            seen += 1
            assert u["Occupancy"] >= 4 and u["VGPRs"] <= 128 and u["ScratchSize"] == 0 and u["LDS"] <= 256, (name, u)
        if "k_residual_plane" in name or "k_residual_final" in name:
            assert u["ScratchSize"] == 0 and u["Occupancy"] >= 4, (name, u)
    assert seen == 4
