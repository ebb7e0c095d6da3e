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


def test_jpeg_decoder_matches_stb_image_statistics(built):
    """SURVEY.md section 5 measured stb_image (the reference's decoder) against libjpeg on this
    file: they differ by +-1 on exactly 41 pixels, none across a phase threshold.  The decoder
    here must show exactly that signature against PIL's libjpeg, and the recorded porosity."""
    import effectivediffusivityfvm_amd as pkg
    from PIL import Image
    mine = pkg.load_jpeg_gray(os.path.join(GOLDEN, "00000.jpg"))
    ref = np.array(Image.open(os.path.join(GOLDEN, "00000.jpg")), dtype=np.uint8)
    assert mine.shape == (128, 128)
    d = mine.astype(int) - ref.astype(int)
    assert np.abs(d).max() == 1 and int((d != 0).sum()) == 41
    for thr in (50, 150, 200):
        assert np.array_equal(mine < thr, ref < thr)
    assert float((mine < 150).mean()) == 0.3460693359375
    # the committed pixel fixture (PIL-decoded) gives the same phases
    assert np.array_equal(np.load(os.path.join(GOLDEN, "img00000_pix.npy")) < 150, mine < 150)


def test_jpeg_decoder_on_second_reference_image(built):
    """00042.jpg (1002x2007, not committed: 771 KB): stb_image vs libjpeg differ on 9 809 pixels and
    364-378 of them cross the 150 threshold.  Only runs where the reference is mounted."""
    path = "/root/reference/Deff2DGPU/00042.jpg"
    if not os.path.exists(path):
        pytest.skip("reference not mounted")
    import effectivediffusivityfvm_amd as pkg
    from PIL import Image
    mine = pkg.load_jpeg_gray(path)
    ref = np.array(Image.open(path), dtype=np.uint8)
    assert mine.shape == (2007, 1002)
    d = mine.astype(int) - ref.astype(int)
    assert np.abs(d).max() == 1 and int((d != 0).sum()) == 9809
    assert 364 <= int(((mine < 150) != (ref < 150)).sum()) <= 378
    assert np.array_equal(mine < 50, ref < 50) and np.array_equal(mine > 200, ref > 200)


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


@pytest.mark.gpu
def test_driver_2phase_batch_config1(built, tmp_path, recorded):
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
    results = {}
    # 2: three worker threads on one GPU, writing a progress file
    for bs, extra in ((3, []), (1, []), (2, ["--devices", "0,0,0", "--progress", "prog.txt"])):
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
        it, deff, conv, _, _, _ = oracle.jacobi(A, b, oracle.linear_guess(96, 64, 0.0, 1.0), D, 0.0, 1.0, 1e-4, 200000)
        for bs in (3, 1, 2):
            res = results[bs][k]
            assert res["image"] == f"{k:05d}.jpg"
            assert (res["iterations"], res["Deff"], res["converge"]) == (it, deff, conv), (bs, k)
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
    rows = open(tmp_path / "out.csv").read().splitlines()
    assert rows[0] == "imgNum,SVF,LVF,PathFlag,Deff,Time,nElements,converge,ds,df,dg" and rows[1].startswith("0,")
