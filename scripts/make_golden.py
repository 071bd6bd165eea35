#!/usr/bin/env python3
"""Generates tests/golden/ (run in the build container, where /root/reference exists).

Inputs  : the reference's .obj / texture ASSETS (data files) are parsed by this
          repository's own data_structures.Model — the reference's Python package is not
          imported (its import chain needs cv2, which the image lacks; DESIGN.md "Oracle").
Outputs : produced by the CPU oracle (oracle/crender_oracle.c).  Before anything is
          written the oracle is checked against the reference-run results recorded in
          SURVEY.md section 8c / 8a-a10 (buffer hashes, covered-pixel and work counts, input
          hashes); a mismatch aborts.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cython3dmodelrenderer_amd.data_structures import Model          # noqa: E402
from cython3dmodelrenderer_amd.scenes import fit_model               # noqa: E402
from oracle import oracle as O                                        # noqa: E402

REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def h16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def full(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# Reference-run values recorded by the survey session (SURVEY.md section 8c "This session's
# hashes", section 8a row a10 probe counts, section 3.3 projection-matrix bits).
SURVEY_PINS = {
    "inputs_trex": {"tri": "a127569779db0498", "col": "4339587371e49c8c", "nrm": "ce26d9cff663ce70"},
    "cube256": {"z": "67c15d0d7aa14afa", "c": "196f3aea08ded5d1", "n": "ee1084a589590021",
                "covered": 65536,
                "stats": {"culled": 6, "empty": 0, "drawn": 6, "bbox_samples": 145920,
                          "inside": 67034, "writes": 65792}},
    "trex256": {"z": "4938c7846db2fab4", "c": "aa17af6bb2e2f5aa", "n": "c2f9ab75e2fc2ba9",
                "covered": 15801},
    "trex1024": {"z": "ce156226ddd3283a", "c": "b1a7b831aa204926", "n": "1d87b7b6a2dbd0b5",
                 "covered": 252539,
                 "stats": {"culled": 6120, "empty": 749, "drawn": 6945, "bbox_samples": 964916,
                           "inside": 316636, "writes": 280328}},
    "bunny4096": {"covered": 14819106,
                  "stats": {"culled": 15720, "empty": 10227, "drawn": 4391,
                            "bbox_samples": 41985294, "inside": 15393831, "writes": 15300857}},
    "trex8192": {"covered": 16157266,
                 "stats": {"culled": 6120, "empty": 351, "drawn": 7343, "bbox_samples": 61744805,
                           "inside": 20260426, "writes": 17935055}},
    "proj_fov45_square": {"P00": "0x401a827a", "P22": "0x3f800347", "P32": "0xbdccd20b"},
}


def render(tri, col, nrm, res, fov=45.0):
    f = O.OracleFiller(res, res, fov=fov)
    f.render_arrays(tri, col, nrm)
    return f


def record(f):
    return {"z": full(f.z_buffer), "c": full(f.color_buffer), "n": full(f.normals_buffer),
            "winner": full(f.winner), "covered": int((f.z_buffer < 1e6).sum()),
            "stats": f.stats.as_dict()}


def check_pin(name, rec):
    pin = SURVEY_PINS[name]
    for k in ("z", "c", "n"):
        if k in pin:
            assert rec[k][:16] == pin[k], (name, k, rec[k][:16], pin[k])
    assert rec["covered"] == pin["covered"], (name, rec["covered"], pin["covered"])
    if "stats" in pin:
        assert rec["stats"] == pin["stats"], (name, rec["stats"], pin["stats"])
    print("pinned against the reference run:", name)


def synth10m_record():
    """configs[4] at full size: 10 M synthetic triangles at 4096 x 4096 (scenes.synthetic_triangles,
    the recipe of SURVEY.md section 8d) through the serial oracle — hashes only."""
    from cython3dmodelrenderer_amd import scenes
    tri, col, nrm = scenes.synthetic_triangles(10_000_000, res=4096)
    f = render(tri, col, nrm, 4096)
    rec = record(f)
    rec["res"] = 4096
    rec["proj"] = full(f.projected)
    print("synth10m", rec["covered"], rec["stats"])
    return rec


def main():
    if "--only-synth10m" in sys.argv:       # add / refresh that one entry (needs no reference checkout)
        path = os.path.join(OUT, "golden.json")
        with open(path) as fh:
            golden = json.load(fh)
        golden["scenes"]["synth10m"] = synth10m_record()
        with open(path, "w") as fh:
            json.dump(golden, fh, indent=1, sort_keys=True)
        return
    os.makedirs(OUT, exist_ok=True)
    # ---- inputs ------------------------------------------------------------------
    trex = Model.read_model(os.path.join(REF, "objects", "T-Rex.obj"))
    # the mesh as parsed, before any transform: what a device-resident Model starts from (the test
    # that takes it through rotate x 2 + fit on the GPU must arrive at trex_inputs.npz, bit for bit)
    np.savez_compressed(os.path.join(OUT, "trex_mesh.npz"), vertices=trex._vertices.astype(np.float32),
                        faces=np.asarray(trex._triangles_vertices, dtype=np.int32))
    if "--only-trex-mesh" in sys.argv:
        return
    trex.rotate([-90, 180, 0])
    trex.rotate([10, -80, 0])
    fit_model(trex)
    t_tri, t_col, t_nrm = trex._vertices_by_triangles, trex._colors_by_triangles, trex._normals_by_triangles
    pin = SURVEY_PINS["inputs_trex"]
    assert (h16(t_tri), h16(t_col), h16(t_nrm)) == (pin["tri"], pin["col"], pin["nrm"]), \
        "Model does not reproduce the reference's T-Rex input arrays"
    print("T-Rex input arrays equal the reference Model's (SURVEY 8c input hashes)")
    np.savez_compressed(os.path.join(OUT, "trex_inputs.npz"), tri=t_tri, col=t_col, nrm=t_nrm)

    cube = Model.read_model(os.path.join(REF, "objects", "cube.obj"))
    fit_model(cube)
    cube.set_uniform_color()
    c_tri, c_col, c_nrm = cube._vertices_by_triangles, cube._colors_by_triangles, cube._normals_by_triangles
    np.savez_compressed(os.path.join(OUT, "cube_inputs.npz"), tri=c_tri, col=c_col, nrm=c_nrm)

    bunny = Model.read_model(os.path.join(REF, "objects", "bunny.obj"))
    fit_model(bunny)
    bunny.set_uniform_color()
    b_tri, b_col, b_nrm = bunny._vertices_by_triangles, bunny._colors_by_triangles, bunny._normals_by_triangles
    np.savez_compressed(os.path.join(OUT, "bunny_inputs.npz"), tri=b_tri, nrm=b_nrm)  # colours = 255

    # ---- oracle outputs, pinned first ----------------------------------------------
    P = O.projection_matrix(45.0, 0.1, 1000.0, 1024, 1024).view(np.uint32)
    pp = SURVEY_PINS["proj_fov45_square"]
    assert (hex(P[0, 0]), hex(P[2, 2]), hex(P[3, 2])) == (pp["P00"], pp["P22"], pp["P32"])

    golden = {"_comment": "sha256 of the CPU oracle's buffers (n_threads=1 order); the entries named "
                          "in SURVEY_PINS were first checked against the reference run",
              "survey_pins": SURVEY_PINS, "scenes": {}}
    scenes = [("cube64", c_tri, c_col, c_nrm, 64), ("cube256", c_tri, c_col, c_nrm, 256),
              ("trex128", t_tri, t_col, t_nrm, 128), ("trex256", t_tri, t_col, t_nrm, 256),
              ("trex1024", t_tri, t_col, t_nrm, 1024), ("bunny512", b_tri, b_col, b_nrm, 512),
              ("bunny4096", b_tri, b_col, b_nrm, 4096), ("trex8192", t_tri, t_col, t_nrm, 8192)]
    for name, tri, col, nrm, res in scenes:
        f = render(tri, col, nrm, res)
        rec = record(f)
        rec["res"] = res
        rec["proj"] = full(f.projected)
        if name in SURVEY_PINS:
            check_pin(name, rec)
        golden["scenes"][name] = rec
        if name in ("cube64", "trex128"):
            np.savez_compressed(os.path.join(OUT, f"{name}_expected.npz"), z=f.z_buffer,
                                color=f.color_buffer, normals=f.normals_buffer, winner=f.winner,
                                projected=f.projected)
        print(name, rec["covered"], rec["stats"])
    golden["scenes"]["synth10m"] = synth10m_record()
    with open(os.path.join(OUT, "golden.json"), "w") as fh:
        json.dump(golden, fh, indent=1, sort_keys=True)

    # the reference's only committed render (a data file): end-to-end golden for
    # Model -> filler -> GuroIllumination -> flip -> uint8 (reference: run.py:20-26)
    shutil.copyfile(os.path.join(REF, "output", "T-Rex.png"),
                    os.path.join(OUT, "reference_output_T-Rex.png"))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
