"""CPU tests: the oracle against the reference (golden vectors, the reference-built
barycentric function, the reference's committed render) and against itself."""
import os

import numpy as np
import pytest

from util import assert_bit_equal, sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _scene(name):
    from cython3dmodelrenderer_amd import scenes
    return scenes.load_fixture(name)


def test_projection_matrix_bits(oracle, golden):
    # SURVEY.md section 3.3 [probe]: reference proj_mat for fov=45, h=w
    P = oracle.projection_matrix(45.0, 0.1, 1000.0, 1024, 1024).view(np.uint32)
    pin = golden["survey_pins"]["proj_fov45_square"]
    assert (hex(P[0, 0]), hex(P[1, 1]), hex(P[2, 2]), hex(P[3, 2])) == \
        (pin["P00"], pin["P00"], pin["P22"], pin["P32"])
    assert P[2, 3] == np.float32(1).view(np.uint32)
    assert np.count_nonzero(P) == 5


@pytest.mark.parametrize("name,fixture,res", [("cube256", "cube_inputs.npz", 256),
                                              ("trex256", "trex_inputs.npz", 256),
                                              ("trex1024", "trex_inputs.npz", 1024)])
def test_oracle_matches_reference_run(oracle, golden, name, fixture, res):
    """Buffer hashes / covered pixels / work counts of the REFERENCE (compiled and run by the
    survey session) for the same input arrays."""
    tri, col, nrm = _scene(fixture)
    f = oracle.OracleFiller(res, res, fov=45)
    f.render_arrays(tri, col, nrm)
    pin = golden["survey_pins"][name]
    assert sha(f.z_buffer)[:16] == pin["z"]
    assert sha(f.color_buffer)[:16] == pin["c"]
    assert sha(f.normals_buffer)[:16] == pin["n"]
    assert int((f.z_buffer < 1e6).sum()) == pin["covered"]
    if "stats" in pin:
        assert f.stats.as_dict() == pin["stats"]


def test_input_fixture_hashes(golden):
    tri, col, nrm = _scene("trex_inputs.npz")
    pin = golden["survey_pins"]["inputs_trex"]
    assert (sha(tri)[:16], sha(col)[:16], sha(nrm)[:16]) == (pin["tri"], pin["col"], pin["nrm"])


@pytest.mark.parametrize("name", ["cube64", "trex128"])
def test_oracle_matches_committed_buffers(oracle, name):
    exp = np.load(os.path.join(GOLD, f"{name}_expected.npz"))
    fixture = "cube_inputs.npz" if name.startswith("cube") else "trex_inputs.npz"
    tri, col, nrm = _scene(fixture)
    res = exp["z"].shape[0]
    f = oracle.OracleFiller(res, res, fov=45)
    f.render_arrays(tri, col, nrm)
    assert_bit_equal(f.projected, exp["projected"], "projected")
    assert_bit_equal(f.z_buffer, exp["z"], "z")
    assert_bit_equal(f.color_buffer, exp["color"], "color")
    assert_bit_equal(f.normals_buffer, exp["normals"], "normals")
    assert_bit_equal(f.winner, exp["winner"], "winner")


def test_bar_against_reference_build(oracle):
    """oracle_bar vs the reference's own math_utils.pyx compiled from where it lies
    (oracle/build_ref.sh -> oracle/_ref/math_utils.so)."""
    from oracle import ref_bar
    if not ref_bar.available():
        pytest.skip("oracle/_ref/math_utils.so not built (needs /root/reference)")
    rng = np.random.default_rng(7)
    for i in range(4000):
        tri = rng.uniform(-50, 300, 9).astype(np.float32)
        if i % 7 == 0:
            tri[6:8] = tri[0:2]            # zero-area: division by zero -> inf / NaN
        if i % 11 == 0:
            tri = np.round(tri)            # vertices on pixel centres: exact zeros
        if i % 13 == 0:
            tri[3:5] = tri[0:2] + np.float32(1e-30)
        x, y = (int(v) for v in rng.integers(-5, 300, 2))
        a, b = ref_bar.bar(tri, x, y), oracle.bar(tri, x, y)
        an, bn = np.isnan(a), np.isnan(b)
        assert (an == bn).all(), (tri, x, y, a, b)
        assert a[~an].tobytes() == b[~bn].tobytes(), (tri, x, y, a, b)


def test_omp_shape_equals_serial_single_thread(oracle):
    tri, col, nrm = _scene("trex_inputs.npz")
    a = oracle.OracleFiller(256, 256, fov=45)
    b = oracle.OracleFiller(256, 256, fov=45, mode="omp", n_threads=1)
    a.render_arrays(tri, col, nrm)
    b.render_arrays(tri, col, nrm)
    assert_bit_equal(a.z_buffer, b.z_buffer, "z")
    assert_bit_equal(a.color_buffer, b.color_buffer, "color")
    assert_bit_equal(a.normals_buffer, b.normals_buffer, "normals")


def test_omp_multithread_close_to_serial(oracle):
    """The Version-C shape with threads has the reference's own z race; it
    must still agree with the serial order on all but a handful of pixels."""
    tri, col, nrm = _scene("trex_inputs.npz")
    a = oracle.OracleFiller(512, 512, fov=45)
    b = oracle.OracleFiller(512, 512, fov=45, mode="omp", n_threads=4)
    a.render_arrays(tri, col, nrm)
    b.render_arrays(tri, col, nrm)
    assert (a.z_buffer != b.z_buffer).sum() <= 64


def test_row_strips_compose(oracle):
    tri, col, nrm = _scene("trex_inputs.npz")
    full = oracle.OracleFiller(200, 200, fov=45)
    full.render_arrays(tri, col, nrm)
    parts = oracle.OracleFiller(200, 200, fov=45)
    for y0, y1 in ((0, 37), (37, 100), (100, 200)):
        parts.render_arrays(tri, col, nrm, y0=y0, y1=y1)
    assert_bit_equal(parts.z_buffer, full.z_buffer, "z")
    assert_bit_equal(parts.color_buffer, full.color_buffer, "color")
    assert_bit_equal(parts.winner, full.winner, "winner")


def test_closed_form_min_z_highest_index(oracle):
    """the serial result is min z, ties -> highest index,
    independent of the order triangles are submitted in (checked by permuting)."""
    from util import random_soup
    rng = np.random.default_rng(3)
    tri, col, nrm = random_soup(rng, 600, 96, size_px=(2, 30), frac_backface=0.0)
    tri[300:] = tri[:300]                  # exact duplicates -> exact z ties
    f = oracle.OracleFiller(96, 96, fov=45)
    f.render_arrays(tri, col, nrm)
    assert (f.winner[f.winner >= 0] >= 300).all()   # the later duplicate wins every tie
    # per pixel: winner's z is the minimum over all triangles drawn alone
    zmin = np.full((96, 96), 1e6, np.float32)
    for t in rng.choice(600, 60, replace=False):
        g = oracle.OracleFiller(96, 96, fov=45)
        g.render_arrays(tri[t:t + 1], col[t:t + 1], nrm[t:t + 1])
        zmin = np.minimum(zmin, g.z_buffer)
    assert (f.z_buffer <= zmin).all()


def test_reference_render_png_end_to_end(oracle):
    """The reference's only committed output (output/T-Rex.png, produced by its run.py:
    Model -> filler(1024, fov 45, 8 threads) -> GuroIllumination([0,0,1]) -> flip -> uint8).
    SURVEY.md section 4: the reference's own 8-thread race / compiler leave ~116 pixels undecided."""
    from PIL import Image
    tri, col, nrm = _scene("trex_inputs.npz")
    f = oracle.OracleFiller(1024, 1024, fov=45)
    f.render_arrays(tri, col, nrm)
    oracle.guro(f.color_buffer, f.normals_buffer, [0, 0, 1])          # (the oracle's own restatement)
    mine_bgr = f.color_buffer[::-1].astype("uint8")
    ref_rgb = np.asarray(Image.open(os.path.join(GOLD, "reference_output_T-Rex.png")).convert("RGB"))
    differing = (mine_bgr[:, :, ::-1] != ref_rgb).any(axis=-1).sum()
    assert differing <= 200, differing


def test_oracle_guro_matches_numpy_statements(oracle):
    """The C restatement of guro_illumination.py:20-27 against the reference's own four numpy
    statements, evaluated by numpy here (written out, not imported from the product package): bit
    for bit on random, degenerate (zero normals), huge, tiny and NaN inputs."""
    rng = np.random.default_rng(17)
    n = (rng.standard_normal((64, 96, 3)) * 10 ** rng.uniform(-3, 3, (64, 96, 1))).astype(np.float32)
    n[0, :8] = 0.0
    n[1, :4] = np.nan
    n[2, :4] = [1e30, -1e30, 1e30]
    n[3, :4] = 1e-30
    col = rng.uniform(0, 255, (64, 96, 3)).astype(np.float32)
    for light in ([0, 0, 1], [0.3, -0.2, 1], [-1, 2, 0.5]):
        want = col.copy()
        light_direction = -np.asarray(light, dtype="float32")                      # :15-18
        light_direction = light_direction / np.linalg.norm(light_direction)
        with np.errstate(invalid="ignore", over="ignore"):
            scalar_product = np.sum(n * light_direction, axis=-1, keepdims=True)   # :23
            norm = np.linalg.norm(n, axis=-1, keepdims=True)                       # :24
            shadow_coeff = scalar_product / (norm + 1e-6)                          # :25
            shadow_coeff = np.clip(shadow_coeff, 0, 1)                             # :26
            want *= shadow_coeff                                                   # :27
        got = oracle.guro(col.copy(), n, light)
        nan = np.isnan(want)
        assert (np.isnan(got) == nan).all()
        assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32)), light


def test_back_face_test_divides_in_double(oracle, tmp_path):
    """.pyx:202 culls on `(n0z + n1z + n2z) / 3 >= 0.0` with float operands and an int literal.  The
    oracle (and the HIP `backface`) take the division to be a DOUBLE one — then the test equals
    `sum >= 0` — which matters for exactly one input: a sum of -1.4e-45 (the smallest negative
    denormal) divided by 3 in FLOAT would round to -0, and -0 >= 0 would cull the triangle.
    Pinned on the reference's OWN text: the filler's .pyx is cythonized to C here (code generation
    only — nothing of it is compiled, imported or run, and nothing of it is stored: the C file lives
    in the test's temporary directory) and the statement Cython emits for line 202 must divide by the
    double literal `3.0`.  Build container only: the checkout never travels (skipped where absent)."""
    import re
    import shutil
    import subprocess
    # 1. the oracle: the smallest negative denormal sum is drawn, +denormal / +0 / -0 sums are culled
    tri = np.array([[[-0.3, -0.3, 1.0], [0.3, -0.3, 1.0], [0.0, 0.3, 1.0]]], np.float32)
    col = np.full((1, 3, 3), 200.0, np.float32)
    for zs, drawn in (([-1e-45, 0.0, 0.0], True), ([1e-45, -1e-45, -1e-45], True), ([1e-45, 0.0, 0.0], False),
                      ([0.0, 0.0, 0.0], False), ([-0.0, -0.0, -0.0], False), ([-1e-45, 1e-45, 0.0], False)):
        nrm = np.zeros((1, 3, 3), np.float32)
        nrm[0, :, 2] = np.array(zs, np.float32)
        f = oracle.OracleFiller(64, 64, fov=45)
        f.render_arrays(tri, col, nrm)
        assert bool((f.z_buffer < 1e6).any()) == drawn, zs
    # 2. what Cython emits for the reference's line 202 itself
    pyx = "/root/reference/crender/cy/pixel_buffer_filler/advanced_pixel_buffer_filler.pyx"
    cython = shutil.which("cython")
    if cython is None or not os.path.exists(pyx):
        pytest.skip("no cython / no reference checkout here: the division's type is pinned in the build container")
    out = tmp_path / "filler.c"
    subprocess.check_call([cython, "-3", "-X", "legacy_implicit_noexcept=True", pyx, "-o", str(out)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    # the C statement(s) generated under the source-line marker of .pyx:202
    marks = [i for i, l in enumerate(lines) if 'advanced_pixel_buffer_filler.pyx":202' in l]
    assert marks, "no code generated for .pyx:202"
    stmt = [l for i in marks for l in lines[i:i + 80] if ">= 0.0)" in l and "__pyx_v_normals.data" in l]
    assert stmt, "the cull test of .pyx:202 was not found in the generated C"
    for l in stmt:
        assert re.search(r"\)\s*/\s*3\.0\)\s*>=\s*0\.0\)", l), l[-80:]
        assert "(float)3" not in l and "3.0f" not in l
