"""Front end either side of the hot path: JPEG decoding (the reference decodes with stb_image),
input.txt parsing and the command-line driver's outputs."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

EXE = os.path.join(ROOT, "effectivediffusivityfvm_amd", "deff2d")


@pytest.fixture(scope="module")
def built():
    # build only what is missing: libdeff_amd.so may already be loaded by this process
    pkg_dir = os.path.join(ROOT, "effectivediffusivityfvm_amd")
    if not (os.path.exists(os.path.join(pkg_dir, "libdeff_amd.so")) and os.path.exists(EXE)):
        subprocess.run(["make", "-s", "-C", os.path.join(pkg_dir, "csrc")], check=True)
    assert os.access(EXE, os.X_OK)
    return EXE


def test_jpeg_decoder_byte_equal_to_stb_image(built, img00000, stb_recorded):
    """The reference decodes with stbi_load(name,&w,&h,&n,1) (Deff2D.cuh:342; stb_image.h:1306).
    tests/golden/img00000_pix_stb.npy holds the bytes the reference's own stb_image.h v2.26 yields for
    00000.jpg (compiled from /root/reference as it lies by tests/golden/make_stb_fixture.py): the decoder
    here must give exactly those bytes."""
    import hashlib
    import effectivediffusivityfvm_amd as pkg
    mine = pkg.load_jpeg_gray(os.path.join(GOLDEN, "00000.jpg"))
    assert mine.shape == (128, 128) and mine.dtype == np.uint8
    assert np.array_equal(mine, img00000)
    rec = stb_recorded["00000.jpg"]
    assert hashlib.sha256(mine.tobytes()).hexdigest() == rec["sha256"]
    assert float((mine < 150).mean()) == rec["porosity_lt150"] == 0.3460693359375
    # and it is NOT libjpeg's decode: the two differ by +-1 on 41 pixels (none across a threshold)
    from PIL import Image
    ref = np.array(Image.open(os.path.join(GOLDEN, "00000.jpg")), dtype=np.uint8)
    d = mine.astype(int) - ref.astype(int)
    assert np.abs(d).max() == 1 and int((d != 0).sum()) == 41
    for thr in (50, 150, 200):
        assert np.array_equal(mine < thr, ref < thr)


def test_jpeg_decoder_on_second_reference_image(built, stb_recorded):
    """00042.jpg (1002x2007, the image the shipped input.txt names; 771 KB, not committed): the decode must
    hash to the SHA-256 of the reference's stb_image decode (stb_decode_recorded.json) -- on this image
    libjpeg would flip 364 pixels across the 150 threshold.  Only runs where the reference is mounted."""
    path = "/root/reference/Deff2DGPU/00042.jpg"
    if not os.path.exists(path):
        pytest.skip("reference not mounted")
    import hashlib
    import effectivediffusivityfvm_amd as pkg
    from PIL import Image
    mine = pkg.load_jpeg_gray(path)
    rec = stb_recorded["00042.jpg"]
    assert list(mine.shape) == rec["shape"] == [2007, 1002]
    assert hashlib.sha256(mine.tobytes()).hexdigest() == rec["sha256"]
    assert int((mine < 50).sum()) == rec["count_lt50"] and int((mine > 200).sum()) == rec["count_gt200"]
    assert int((mine < 150).sum()) == rec["count_lt150"]
    ref = np.array(Image.open(path), dtype=np.uint8)
    d = mine.astype(int) - ref.astype(int)
    assert np.abs(d).max() == 1 and int((d != 0).sum()) == 9809
    assert 364 <= int(((mine < 150) != (ref < 150)).sum()) <= 378


def test_stb_fixture_regenerates_from_the_reference(img00000, stb_recorded, tmp_path):
    """Where /root/reference is mounted: compile the reference's stb_image.h as it lies (plain C, gcc, a
    12-line program of ours -- tests/golden/make_stb_fixture.py) and check that the committed pixel fixture and
    hashes are what it yields today.  The GPU box has no reference: skipped there."""
    if not os.path.exists("/root/reference/Deff2DGPU/stb_image.h"):
        pytest.skip("reference not mounted")
    import hashlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_stb_fixture", os.path.join(GOLDEN, "make_stb_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    exe = mk.build_decoder(str(tmp_path))
    pix, n = mk.stb_decode(exe, "/root/reference/Deff2DGPU/00000.jpg", str(tmp_path))
    assert n == 1 and np.array_equal(pix, img00000)
    pix42, n42 = mk.stb_decode(exe, "/root/reference/Deff2DGPU/00042.jpg", str(tmp_path))
    assert n42 == 1 and hashlib.sha256(pix42.tobytes()).hexdigest() == stb_recorded["00042.jpg"]["sha256"]
    # the committed copy of 00000.jpg is the reference's file
    assert open(os.path.join(GOLDEN, "00000.jpg"), "rb").read() == open("/root/reference/Deff2DGPU/00000.jpg", "rb").read()


def test_jpeg_decoder_byte_equal_to_stb_image_on_random_files(built, tmp_path):
    """oracle/_ref/stb_dump is the reference's image read -- stbi_load(name,&w,&h,&n,1), Deff2D.cuh:342 -- built from the
    reference's own stb_image.h (oracle/Makefile, target `ref`; the binary travels to the GPU box).  60 one-component baseline
    JPEGs of assorted sizes (partial MCUs), contents (two-level, three-level, smooth, noise) and qualities, with and
    without restart markers, must decode to exactly the same bytes with csrc/driver/jpeg_gray.hpp."""
    exe = os.path.join(ROOT, "oracle", "_ref", "stb_dump")
    if not os.access(exe, os.X_OK):
        pytest.skip("oracle/_ref/stb_dump not built (needs /root/reference at build time)")
    import effectivediffusivityfvm_amd as pkg
    from PIL import Image
    rng = np.random.default_rng(20241022)
    checked = 0
    for k in range(60):
        W, H = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        kind = k % 4
        if kind == 0:
            a = np.where(rng.random((H, W)) < 0.5, 0, 255)
        elif kind == 1:
            f = np.kron(rng.random((H // 8 + 1, W // 8 + 1)), np.ones((8, 8)))[:H, :W]
            a = np.where(f < 0.3, 0, np.where(f < 0.6, 150, 255))
        elif kind == 2:
            yy, xx = np.mgrid[0:H, 0:W]
            a = 127.5 + 127.5 * np.sin(xx / 7.0) * np.cos(yy / 11.0)
        else:
            a = rng.random((H, W)) * 255
        path = tmp_path / f"r{k}.jpg"
        opts = dict(quality=int(rng.integers(30, 101)))
        if k % 5 == 0:
            opts["restart_marker_blocks"] = 3
        Image.fromarray(a.astype(np.uint8)).save(path, **opts)
        raw = tmp_path / "out.raw"
        r = subprocess.run([exe, str(path), str(raw)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        w, h, n = map(int, r.stdout.split())
        want = np.fromfile(raw, dtype=np.uint8).reshape(h, w)
        mine = pkg.load_jpeg_gray(path)
        assert n == 1 and mine.shape == (H, W) == (h, w)
        assert np.array_equal(mine, want), (k, W, H, opts, int((mine != want).sum()))
        checked += 1
    assert checked == 60


def test_progressive_jpeg_byte_equal_to_stb_image(built, tmp_path):
    """stbi_load also reads progressive JPEG (SOF2: spectral selection + successive approximation, stb_image.h:2194-2340,
    3006-3025), so the front end does: 48 one-component progressive files -- the same four kinds of content, sizes with
    partial blocks down to 1 x 1, qualities 30..100, libjpeg's default scan script (DC first + refinement, AC bands with
    first and refinement passes, end-of-band runs), optimised tables, with and without restart markers -- must decode to
    exactly the bytes of oracle/_ref/stb_dump (the reference's own header)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "stb_dump")
    if not os.access(exe, os.X_OK):
        pytest.skip("oracle/_ref/stb_dump not built (needs /root/reference at build time)")
    import effectivediffusivityfvm_amd as pkg
    from PIL import Image
    rng = np.random.default_rng(3008)
    for k in range(48):
        W, H = (1, 1) if k == 0 else (int(rng.integers(1, 260)), int(rng.integers(1, 260)))
        kind = k % 4
        if kind == 0:
            a = np.where(rng.random((H, W)) < 0.5, 0, 255)
        elif kind == 1:
            f = np.kron(rng.random((H // 8 + 1, W // 8 + 1)), np.ones((8, 8)))[:H, :W]
            a = np.where(f < 0.3, 0, np.where(f < 0.6, 150, 255))
        elif kind == 2:
            yy, xx = np.mgrid[0:H, 0:W]
            a = 127.5 + 127.5 * np.sin(xx / 7.0) * np.cos(yy / 11.0)
        else:
            a = rng.random((H, W)) * 255
        path = tmp_path / f"p{k}.jpg"
        opts = dict(quality=int(rng.integers(30, 101)), progressive=True)
        if k % 5 == 0:
            opts["restart_marker_blocks"] = 3
        if k % 3 == 0:
            opts["optimize"] = True
        Image.fromarray(a.astype(np.uint8)).save(path, **opts)
        assert b"\xff\xc2" in open(path, "rb").read()             # really a progressive frame
        raw = tmp_path / "out.raw"
        r = subprocess.run([exe, str(path), str(raw)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        w, h, n = map(int, r.stdout.split())
        want = np.fromfile(raw, dtype=np.uint8).reshape(h, w)
        mine = pkg.load_jpeg_gray(path)
        assert n == 1 and mine.shape == (H, W) == (h, w)
        assert np.array_equal(mine, want), (k, W, H, opts, int((mine != want).sum()))
    # the reference's own image re-encoded progressively (three grey levels, 2 M pixels)
    big = tmp_path / "big.jpg"
    Image.open(os.path.join(GOLDEN, "00042.jpg")).save(big, quality=92, progressive=True)
    raw = tmp_path / "big.raw"
    r = subprocess.run([exe, str(big), str(raw)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    w, h, n = map(int, r.stdout.split())
    assert np.array_equal(pkg.load_jpeg_gray(big), np.fromfile(raw, dtype=np.uint8).reshape(h, w))


def test_png_and_bmp_get_a_message(built, tmp_path):
    """stbi_load reads 1-channel PNG and BMP too (stb_image.h:1094-1097); this front end reads JPEG only and says so."""
    import effectivediffusivityfvm_amd as pkg
    from PIL import Image
    img = Image.fromarray(np.zeros((8, 8), dtype=np.uint8))
    for ext, word in (("png", "PNG input"), ("bmp", "BMP input")):
        path = tmp_path / f"x.{ext}"
        img.save(path)
        with pytest.raises(pkg.DeffError, match=word):
            pkg.load_jpeg_gray(path)


def test_jpeg_decoder_rejects_what_the_reference_rejects(built, tmp_path):
    import effectivediffusivityfvm_amd as pkg
    from PIL import Image
    rgb = tmp_path / "rgb.jpg"
    Image.fromarray(np.zeros((16, 16, 3), dtype=np.uint8)).save(rgb)
    with pytest.raises(pkg.DeffError, match="single-channel"):
        pkg.load_jpeg_gray(rgb)
    with pytest.raises(pkg.DeffError):
        pkg.load_jpeg_gray(tmp_path / "missing.jpg")
    # odd sizes (partial MCUs) and restart markers decode like libjpeg up to the IDCT rounding
    rng = np.random.default_rng(0)
    img = (rng.random((37, 53)) * 255).astype(np.uint8)
    p = tmp_path / "odd.jpg"
    Image.fromarray(img).save(p, quality=90)
    mine = pkg.load_jpeg_gray(p)
    ref = np.array(Image.open(p), dtype=np.uint8)
    assert mine.shape == (37, 53) and np.abs(mine.astype(int) - ref.astype(int)).max() <= 1


def test_jpeg_header_that_promises_a_huge_image_is_refused_before_allocating(built, tmp_path):
    """a progressive file keeps 128 B per block over all scans, so a small crafted file whose frame header names
    65535 x 65535 pixels must be refused by size (2^28 pixels is the cap), not by an allocation."""
    import effectivediffusivityfvm_amd as pkg
    from PIL import Image
    p = tmp_path / "small.jpg"
    Image.fromarray(np.zeros((16, 16), dtype=np.uint8)).save(p, progressive=True)
    raw = bytearray(p.read_bytes())
    i = raw.index(b"\xff\xc2")                        # SOF2: length(2) precision(1) height(2) width(2)
    raw[i + 5:i + 9] = b"\xff\xff\xff\xff"
    q = tmp_path / "huge.jpg"
    q.write_bytes(bytes(raw))
    with pytest.raises(pkg.DeffError, match="larger than 2\\^28 pixels"):
        pkg.load_jpeg_gray(q)


def test_input_file_parsing_matches_reference_conventions(built, tmp_path):
    """Keys with their colon, any order, numeric values through double (MaxIter: 5e5), unknown
    lines ignored (the decorative 'Input File:' header), file names as second token; defaults for
    what is missing.  Checked through the driver's Verbose echo (it fails later for lack of an image)."""
    open(tmp_path / "in.txt", "w").write(
        "Input File:\nVerbose: 1\nMaxIter: 5e5\nsomething else entirely\nDf: 2.5\nPhases: 3\nDg: 1237500\n"
        "InputName: nothere.jpg  trailing words\nConvergence: 1e-5\nMeshAmpX: 2\n")
    r = subprocess.run([EXE, str(tmp_path / "in.txt")], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "cannot open nothere.jpg" in r.stderr
    out = r.stdout
    assert "Phases = 3" in out and "Df = 2.5, Dg = 1.2375e+06" in out and "MaxIter = 500000" in out
    assert "Mesh amplification = 2 x 1" in out and "Convergence = 1e-05" in out and "Input = nothere.jpg" in out


def _write_input(path, **kv):
    lines = ["Input File:"] + [f"{k}: {v}" for k, v in kv.items()]
    open(path, "w").write("\n".join(lines) + "\n")


def test_driver_input_errors_without_gpu(built, tmp_path):
    r = subprocess.run([EXE, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage" in r.stdout
    r = subprocess.run([EXE, str(tmp_path / "nope.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    _write_input(tmp_path / "bad.txt", Phases=4)
    r = subprocess.run([EXE, str(tmp_path / "bad.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "Phases" in r.stderr
    _write_input(tmp_path / "amp.txt", Phases=2, MeshAmpX=0)
    r = subprocess.run([EXE, str(tmp_path / "amp.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "MeshIncrease" in r.stderr
    r = subprocess.run([EXE, "--precond-maxiter", "0"], capture_output=True, text=True)
    assert r.returncode == 2 and "--precond-maxiter" in r.stderr
    r = subprocess.run([EXE, "--help"], capture_output=True, text=True)
    assert "--prefetch-threads" in r.stdout and "--precond-maxiter" in r.stdout


def test_front_end_survives_damaged_input_under_sanitizers(tmp_path):
    """JPEG decoder and input-file parser on 4 000 truncated / bit-flipped / spliced / random inputs,
    built with AddressSanitizer + UBSan (CPU build): nothing may read out of bounds, overflow or
    leak, and every rejection carries a message."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    exe = str(tmp_path / "frontend_fuzz")
    subprocess.run(["make", "-s", "-C", cpp, "frontend_fuzz"], check=True)
    shutil.copy(os.path.join(cpp, "frontend_fuzz"), exe)
    img = str(tmp_path / "good.jpg")
    shutil.copy(os.path.join(GOLDEN, "00000.jpg"), img)
    r = subprocess.run([exe, img, "4000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "accepted" in r.stdout and "rejected" in r.stdout
    # the same with a PROGRESSIVE seed (several scans, refinement passes, end-of-band runs, restart markers): the multi-scan
    # path of the decoder under the same damage
    from PIL import Image
    prog = str(tmp_path / "good_progressive.jpg")
    Image.open(img).save(prog, quality=85, progressive=True, restart_marker_blocks=5)
    assert b"\xff\xc2" in open(prog, "rb").read()
    r = subprocess.run([exe, prog, "4000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "accepted" in r.stdout and "rejected" in r.stdout


@pytest.mark.gpu
def test_driver_2phase_batch_config1(built, tmp_path, recorded, oracle, img00000):
    """Config #1 through the command line: the reference's input.txt keys, 00000.jpg, RunBatch 1."""
    shutil.copy(os.path.join(GOLDEN, "00000.jpg"), tmp_path / "00000.jpg")
    _write_input(tmp_path / "input.txt", Phases=2, Ds="1e-3", Df=1, MeshAmpX=1, MeshAmpY=1, CR=1, CL=0,
                 OutputName="out.csv", printCMap=0, Convergence="1e-6", MaxIter="5e5", Verbose=0, RunBatch=1,
                 NumImages=1)
    r = subprocess.run([EXE, "input.txt", "--json", "res.json", "--field-bin", "field"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    res = json.load(open(tmp_path / "res.json"))["results"][0]
    rec = recorded["img00000_2phase_batch"]
    assert res["iterations"] == rec["iters"] and res["porosity"] == rec["porosity"] and res["PathFlag"] == 1
    assert res["Deff"] == rec["deff_build_b"]              # the reference's written operation order
    # --arith contracted: the survey's other host build of the reference, Deff and conv to the last digit
    r = subprocess.run([EXE, "input.txt", "--json", "res_c.json", "--arith", "contracted"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    res_c = json.load(open(tmp_path / "res_c.json"))["results"][0]
    assert res_c["iterations"] == rec["iters"] and res_c["Deff"] == rec["deff_build_a"]
    assert res_c["converge"] == rec["conv_build_a"]
    rows = open(tmp_path / "out.csv").read().splitlines()
    assert rows[0] == "imgNum,porosity,PathFlag,Deff,Time,nElements,converge,ds,df"
    cols = rows[1].split(",")
    assert cols[0] == "0" and cols[1] == "0.346069" and cols[2] == "1" and cols[3] == "0.182862" and cols[5] == "16384"
    field = np.fromfile(tmp_path / "field_00000_128x128.f64").reshape(128, 128)
    gold = np.load(os.path.join(GOLDEN, "img00000_field.npy"))
    assert np.linalg.norm(field - gold) / np.linalg.norm(gold) <= 1e-6 and np.array_equal(field, gold)
    # --json also carries Residual() (cuh:451-494) of the final field: the oracle's value up to the order of the sum
    oracle.assert_residual(res["residual"], gold, oracle.fill_D_2phase(img00000, 1.0, 1e-3), 0.0, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("Df,want_dcf", [("1e4", [100.0, 1e4]), ("250", [100.0, 250.0]), ("100", [100.0]),
                                         ("1e6", [100.0, 1e4, 1e6])])
def test_driver_single_sim_dcf_ramp(built, tmp_path, oracle, img00000, Df, want_dcf):
    """RunBatch 0, 2 phases: SingleSim's DCF continuation (Deff2D.cuh:1759-1817) -- the fluid diffusivity ramped
    100, 1e4, ... up to Df, every stage re-assembled and warm-started from the previous field, deff /= DCF per
    stage.  Per-stage iteration counts, the final Deff / conv and the final field must equal the oracle's
    restatement of the ramp (tests/oracle_binding.solve_single_2phase_ramp) on the stb-decoded 00000.jpg."""
    shutil.copy(os.path.join(GOLDEN, "00000.jpg"), tmp_path / "00000.jpg")
    _write_input(tmp_path / "input.txt", Phases=2, Ds="1e-3", Df=Df, MeshAmpX=1, MeshAmpY=1, InputName="00000.jpg",
                 CR=1, CL=0, OutputName="out.csv", printCMap=0, Convergence="1e-6", MaxIter="3e5", Verbose=1,
                 RunBatch=0, NumImages=1)
    r = subprocess.run([EXE, "input.txt", "--json", "res.json", "--field-bin", "f"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    res = json.load(open(tmp_path / "res.json"))["results"][0]
    want = oracle.solve_single_2phase_ramp(img00000, 1e-3, float(Df), 0.0, 1.0, 1e-6, 300000)
    assert [st[0] for st in want["stages"]] == want_dcf
    assert res["stage_iterations"] == [st[1] for st in want["stages"]]
    assert res["iterations"] == want["stages"][-1][1]
    assert res["Deff"] == want["stages"][-1][2] and res["converge"] == want["stages"][-1][3]
    # the per-stage lines the reference prints under Verbose: 1 (cuh:1796-1808), one per stage
    for dcf, it, deff, _ in want["stages"]:
        assert f"Iterations taken = {it}\n" in r.stdout
        assert f"DCF = {dcf:g}, Deff {deff:g}\n" in r.stdout
    field = np.fromfile(tmp_path / "f_00000_128x128.f64").reshape(128, 128)
    assert np.linalg.norm(field - want["field"]) / np.linalg.norm(want["field"]) <= 1e-6
    assert np.array_equal(field, want["field"])
    cols = open(tmp_path / "out.csv").read().splitlines()[1].split(",")
    assert cols[0] == "00000.jpg" or cols[0] == "0"
    assert cols[3] == f"{want['stages'][-1][2]:f}"                  # outputSingle prints Deff with %f, cuh:184


@pytest.mark.gpu
def test_driver_single_sim_below_ten_is_flagged(built, tmp_path, oracle, img00000):
    """Df < 10: the reference's ramp loop `while (DCF <= DCF_Max)` starts from DCF = 10 (cuh:1714, :1761), runs no
    solve and writes uninitialised memory to the CSV.  The oracle restatement shows the no-op; deff2d says so on
    stderr and solves once with Df, i.e. gives the BatchSim value."""
    assert oracle.solve_single_2phase_ramp(img00000, 1e-3, 5.0, 0.0, 1.0, 1e-6, 1000)["stages"] == []
    shutil.copy(os.path.join(GOLDEN, "00000.jpg"), tmp_path / "00000.jpg")
    _write_input(tmp_path / "input.txt", Phases=2, Ds="1e-3", Df=5, MeshAmpX=1, MeshAmpY=1, InputName="00000.jpg",
                 CR=1, CL=0, OutputName="out.csv", printCMap=0, Convergence="1e-6", MaxIter="3e5", Verbose=0,
                 RunBatch=0, NumImages=1)
    r = subprocess.run([EXE, "input.txt", "--json", "res.json"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "runs no solve" in r.stderr
    res = json.load(open(tmp_path / "res.json"))["results"][0]
    D = oracle.fill_D_2phase(img00000, 5.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, _, _, _ = oracle.jacobi(A, b, oracle.linear_guess(128, 128, 0.0, 1.0), D, 0.0, 1.0, 1e-6, 300000)
    assert (res["iterations"], res["Deff"], res["converge"]) == (it, deff / 5.0, conv)


@pytest.mark.gpu
def test_driver_single_image_over_row_slabs(built, tmp_path, recorded):
    """RunBatch 0 with --devices a,b,c: the one image is solved as three row slabs (here all on GPU 0);
    iterations, Deff, conv and the field are those of the one-GPU run."""
    shutil.copy(os.path.join(GOLDEN, "00000.jpg"), tmp_path / "00000.jpg")
    _write_input(tmp_path / "input.txt", Phases=2, Ds="1e-3", Df=1, MeshAmpX=1, MeshAmpY=1, InputName="00000.jpg", CR=1,
                 CL=0, OutputName="out.csv", printCMap=0, Convergence="1e-6", MaxIter="5e5", Verbose=1, RunBatch=0,
                 NumImages=1)
    outs = {}
    for tag, extra in (("one", []), ("slabs", ["--devices", "0,0,0"])):
        r = subprocess.run([EXE, "input.txt", "--json", f"{tag}.json", "--field-bin", tag] + extra, cwd=tmp_path,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        outs[tag] = (json.load(open(tmp_path / f"{tag}.json"))["results"][0],
                     np.fromfile(tmp_path / f"{tag}_00000_128x128.f64"), r.stdout)
    assert "Row slabs over 3 GPUs" in outs["slabs"][2]
    rec = recorded["img00000_2phase_batch"]
    for tag in outs:
        res = outs[tag][0]
        assert res["iterations"] == rec["iters"] and res["Deff"] == rec["deff_build_b"]
    assert outs["one"][0]["converge"] == outs["slabs"][0]["converge"]
    assert np.array_equal(outs["one"][1], outs["slabs"][1])


@pytest.mark.gpu
def test_driver_3phase_as_shipped(built, tmp_path, recorded):
    """The reference's shipped input.txt, pointed at 00000.jpg: 3 phases, DCG continuation."""
    shutil.copy(os.path.join(GOLDEN, "00000.jpg"), tmp_path / "00000.jpg")
    _write_input(tmp_path / "input.txt", Phases=3, Ds=0, Df=1, Dg=1237500, MeshAmpX=1, MeshAmpY=1,
                 InputName="00000.jpg", CR=1, CL=0, OutputName="singleTest.csv", printCMap=1, CMapName="CMAP.csv",
                 Convergence="1e-5", MaxIter="5e5", Verbose=1, RunBatch=0, NumImages=500)
    r = subprocess.run([EXE, "--json", "res.json"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    rec = recorded["img00000_3phase_as_shipped"]
    res = json.load(open(tmp_path / "res.json"))["results"][0]
    assert res["stage_iterations"] == rec["stage_sweeps"]
    assert res["Deff"] == rec["deff"] and res["converge"] == rec["conv"] and res["SVF"] == rec["SVF"]
    assert "Pre-Cond Stage 6: DCG = 1.000e+06" in r.stdout and "Iteration = 0, Deff = " in r.stdout
    # the same input over three row slabs (--devices): identical numbers
    r2 = subprocess.run([EXE, "--json", "res_slabs.json", "--devices", "0,0,0"], cwd=tmp_path, capture_output=True,
                        text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr + r2.stdout
    res2 = json.load(open(tmp_path / "res_slabs.json"))["results"][0]
    assert "Row slabs over 3 GPUs" in r2.stdout
    assert res2["stage_iterations"] == rec["stage_sweeps"] and res2["Deff"] == rec["deff"] and res2["converge"] == rec["conv"]
    rows = open(tmp_path / "singleTest.csv").read().splitlines()
    assert rows[0] == "imgNum,SVF,LVF,PathFlag,Deff,Time,nElements,converge,ds,df,dg"
    assert rows[1].startswith("00000.jpg,0.653931,0.000000,1,2.247e+05,")
    cmap = open(tmp_path / "CMAP.csv").read().splitlines()
    assert cmap[0] == "X,Y,C" and len(cmap) == 1 + 128 * 128 and cmap[1].startswith("0,0,")


@pytest.mark.gpu
def test_driver_3phase_as_shipped_on_its_own_image(built, tmp_path):
    """The reference's shipped input.txt on the image it names, 00042.jpg (1002 x 2007, three grey levels, ~329 k pixels
    exactly at 150; input.txt:2-18, SingleSim3Phase cuh:1316-1633): flood fill with the seeded right column, six DCG
    continuation stages and the final solve.  The oracle ran this flow ONCE in the authoring container with every stage
    capped (tests/golden/make_img00042_golden.py -> img00042_3phase_capped.json; the uncapped run is hours of one core);
    deff2d must reproduce stage sweeps, Deff, conv, volume fractions, PathFlag and the FP64 field bit for bit."""
    import hashlib
    gold = json.load(open(os.path.join(GOLDEN, "img00042_3phase_capped.json")))
    cap = gold["options"]["MaxIter"]
    shutil.copy(os.path.join(GOLDEN, "00042.jpg"), tmp_path / "00042.jpg")
    _write_input(tmp_path / "input.txt", Phases=3, Ds=0, Df=1, Dg=1237500, MeshAmpX=1, MeshAmpY=1,
                 InputName="00042.jpg", CR=1, CL=0, OutputName="singleTest.csv", printCMap=0, CMapName="CMAP_00042.csv",
                 Convergence="1e-5", MaxIter=cap, Verbose=1, RunBatch=0, NumImages=500)
    r = subprocess.run([EXE, "--json", "res.json", "--field-bin", "field", "--precond-maxiter", str(gold["options"]["precond_maxiter"])],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    res = json.load(open(tmp_path / "res.json"))["results"][0]
    assert res["stage_iterations"] == gold["stage_sweeps"]
    assert res["Deff"] == gold["deff"] and res["converge"] == gold["conv"]
    assert res["SVF"] == gold["SVF"] and res["LVF"] == gold["LVF"] and bool(res["PathFlag"]) == gold["path"]
    H, W = gold["shape"]
    assert res["nElements"] == H * W
    x = np.fromfile(tmp_path / f"field_00000_{W}x{H}.f64", dtype=np.float64).reshape(H, W)
    for key, v in gold["field_probe"].items():
        i, j = map(int, key.split(","))
        assert x[i, j] == v, (key, x[i, j], v)
    assert hashlib.sha256(x.tobytes()).hexdigest() == gold["field_sha256"]
    assert "Pre-Cond Stage 6: DCG = 1.000e+06" in r.stdout


@pytest.mark.gpu
def test_driver_batch_groups_images(built, tmp_path, oracle):
    """RunBatch over 7 images of 96x64 solved in stacked groups of 3: every row equals the oracle's
    one-image result, whatever the grouping."""
    from PIL import Image
    rng = np.random.default_rng(123)
    pixs = []
    for k in range(7):
        a = np.where(rng.random((64, 96)) < 0.35 + 0.05 * k, 0, 255).astype(np.uint8)
        Image.fromarray(a).save(tmp_path / f"{k:05d}.jpg", quality=95)
    import effectivediffusivityfvm_amd as pkg
    for k in range(7):
        pixs.append(pkg.load_jpeg_gray(tmp_path / f"{k:05d}.jpg"))
    _write_input(tmp_path / "input.txt", Phases=2, Ds="1e-2", Df=1, MeshAmpX=1, MeshAmpY=1, CR=1, CL=0,
                 OutputName="out.csv", printCMap=0, Convergence="1e-4", MaxIter="2e5", Verbose=0, RunBatch=1,
                 NumImages=7)
    # image 5 is stored as a PROGRESSIVE JPEG of the same pixels' source (its own decode is what its row must match)
    a5 = np.where(np.random.default_rng(5).random((64, 96)) < 0.6, 0, 255).astype(np.uint8)
    Image.fromarray(a5).save(tmp_path / "00005.jpg", quality=90, progressive=True)
    pixs[5] = pkg.load_jpeg_gray(tmp_path / "00005.jpg")
    results = {}
    # 2: three worker threads on one GPU, writing a progress file; 3 with one / 1 with three prefetch threads per worker
    # (images then reach the solver out of index order: rows still land by index)
    for bs, extra in ((3, ["--prefetch-threads", "1"]), (1, ["--prefetch-threads", "3"]),
                      (2, ["--devices", "0,0,0", "--progress", "prog.txt"])):
        r = subprocess.run([EXE, "input.txt", "--json", f"res{bs}.json", "--batch-size", str(bs)] + extra,
                           cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        results[bs] = json.load(open(tmp_path / f"res{bs}.json"))["results"]
    # resume: drop the last two lines of the progress file (an "interrupted" run) plus a torn line;
    # only those images are solved again and the result is the same table
    lines = open(tmp_path / "prog.txt").read().splitlines()
    assert len(lines) == 7
    open(tmp_path / "prog.txt", "w").write("\n".join(lines[:5]) + "\n3 0x1.8p+0 torn")
    r = subprocess.run([EXE, "input.txt", "--json", "res_resume.json", "--batch-size", "2", "--progress", "prog.txt"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    resumed = json.load(open(tmp_path / "res_resume.json"))["results"]
    strip = lambda rows: [{k: v for k, v in row.items() if k != "Time"} for row in rows]
    assert strip(resumed) == strip(results[2])
    assert len(open(tmp_path / "prog.txt").read().splitlines()) >= 7
    for k in range(7):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, xk, _, _ = oracle.jacobi(A, b, oracle.linear_guess(96, 64, 0.0, 1.0), D, 0.0, 1.0, 1e-4, 200000)
        for bs in (3, 1, 2):
            res = results[bs][k]
            assert res["image"] == f"{k:05d}.jpg"
            assert (res["iterations"], res["Deff"], res["converge"]) == (it, deff, conv), (bs, k)
            oracle.assert_residual(res["residual"], xk, D, 0.0, 1.0)     # per slot of a stream (deff_residual_slot)
            assert res["PathFlag"] == int(oracle.floodfill((pixs[k] > 150).astype(np.uint32))[1])


@pytest.mark.gpu
def test_driver_3phase_batch_groups(built, tmp_path, oracle):
    """RunBatch with 3 phases: equally sized images go through the DCG continuation together in one
    stacked context; every image's stage iteration counts, Deff and conv equal the oracle's one-image flow."""
    from PIL import Image
    import effectivediffusivityfvm_amd as pkg
    rng = np.random.default_rng(77)
    pixs = []
    for k in range(5):
        f = np.kron(rng.random((6, 8)), np.ones((8, 8)))
        a = np.where(f < 0.3, 0, np.where(f < 0.65, 150, 255)).astype(np.uint8)
        Image.fromarray(a).save(tmp_path / f"{k:05d}.jpg", quality=100)
        pixs.append(pkg.load_jpeg_gray(tmp_path / f"{k:05d}.jpg"))
    _write_input(tmp_path / "input.txt", Phases=3, Ds=0, Df=1, Dg=2500, MeshAmpX=1, MeshAmpY=1, CR=1, CL=0,
                 OutputName="out.csv", printCMap=0, Convergence="1e-4", MaxIter="3e5", Verbose=0, RunBatch=1,
                 NumImages=5)
    out = {}
    for bs in (3, 1):
        r = subprocess.run([EXE, "input.txt", "--json", f"res{bs}.json", "--batch-size", str(bs)], cwd=tmp_path,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        out[bs] = json.load(open(tmp_path / f"res{bs}.json"))["results"]
    for k in range(5):
        with np.errstate(all="ignore"):
            want = oracle.solve_3phase(pixs[k], 0.0, 1.0, 2500.0, 0.0, 1.0, 1e-4, 300000)
        for bs in (3, 1):
            res = out[bs][k]
            assert res["stage_iterations"] == want["stage_sweeps"], (bs, k)
            assert res["Deff"] == want["deff"] and res["converge"] == want["conv"]
            assert res["SVF"] == want["SVF"] and res["LVF"] == want["LVF"] and res["PathFlag"] == int(want["path"])
            if np.isfinite(want["field"]).all():               # 3 pixel classes, Ds = 0: stacks (deff_residual) and single images
                oracle.assert_residual(res["residual"], want["field"], oracle.fill_D_3phase(pixs[k], 1.0, 0.0, 2500.0), 0.0, 1.0)
            else:
                assert res["residual"] is None
    rows = open(tmp_path / "out.csv").read().splitlines()
    assert rows[0] == "imgNum,SVF,LVF,PathFlag,Deff,Time,nElements,converge,ds,df,dg" and rows[1].startswith("0,")
