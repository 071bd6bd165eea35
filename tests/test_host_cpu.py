"""CPU tests of the host side: the C ABI library loads and exports every symbol that
include/crender_hip.h declares (no compute calls: there is no GPU here), the argument
checks that need no device, the Model / illumination host code, and the two-rank strip
layout over gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from util import assert_bit_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from cython3dmodelrenderer_amd import _build, _capi
    _build.build()           # hipcc cross-compiles for gfx950 without a GPU
    return _capi


def test_library_exports_every_declared_symbol(capi):
    header = open(os.path.join(ROOT, "include", "crender_hip.h")).read()
    declared = set(re.findall(r"CRENDER_API[^;(]*?\b(crender_\w+)\s*\(", header))
    assert len(declared) >= 14
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    L = capi.load()
    for name in declared:
        assert getattr(L, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.lib_path()], text=True)
    exported = set(re.findall(r" T (crender_\w+)", out))
    assert declared <= exported
    assert L.crender_abi_version() == capi.ABI_VERSION


def test_torch_extension_is_built_and_bound():
    """The torch C++ extension over the C ABI (csrc/crender_torch.cpp): builds in-tree without a
    GPU, loads, links the same libcrender_hip.so, and exposes the per-frame entry points."""
    from cython3dmodelrenderer_amd import _capi, _torch_ext
    if _torch_ext.needs_build():
        _torch_ext.build()
    m = _torch_ext.load()
    assert m.abi_version() == _capi.ABI_VERSION
    for name in ("render_model", "pipeline_bind", "pipeline_submit", "pipeline_join"):
        assert callable(getattr(m, name))
    with open("/proc/self/maps") as fh:
        maps = fh.read()
    assert "crender_torch" in maps and "libcrender_hip.so" in maps
    # argument checks run before anything touches a device
    import torch
    z = torch.zeros(4)
    with pytest.raises(RuntimeError):
        m.render_model(0, z, z, z, torch.zeros(16), z, z, z, None, 0)


def test_library_contains_gfx950_code_object(capi):
    data = open(capi.lib_path(), "rb").read()
    assert b"gfx950" in data


def test_host_only_entry_points(capi, oracle):
    import ctypes as C
    L = capi.load()
    P = (C.c_float * 16)()
    for fov, zn, zf, h, w in [(45, 0.1, 1000, 1024, 1024), (90, 0.1, 1000, 512, 512),
                              (60, 0.5, 50, 300, 500), (33.3, 0.01, 10, 4096, 2048)]:
        assert L.crender_projection_matrix(fov, zn, zf, h, w, P) == capi.OK
        assert_bit_equal(np.array(P[:], np.float32).reshape(4, 4),
                         oracle.projection_matrix(fov, zn, zf, h, w), "proj_mat")
    assert L.crender_projection_matrix(45, 0.1, 1000, 0, 10, P) == capi.EINVAL
    assert L.crender_last_error()
    # workspace sizing is pure host arithmetic
    a = L.crender_plan_workspace_bytes(1024, 1024, 0, 1024, 13814, 0, 0)
    b = L.crender_plan_workspace_bytes(1024, 1024, 0, 1024, 13814, 10**6, 0)
    assert 0 < a < b and a % 256 == 0
    assert L.crender_plan_workspace_bytes(1024, 1024, 512, 256, 100, 0, 0) == 0     # y0 > y1
    assert L.crender_plan_workspace_bytes(70000, 64, 0, 64, 100, 0, 0) == 0         # H > 65535
    assert L.crender_atomic_scratch_bytes(1024, 512) == 8 * 1024 * 512
    plan = C.c_void_p()
    assert L.crender_plan_create(C.byref(plan), 64, 64, 0, 64, 10, 0, 0, None, 0, None) == capi.EINVAL
    assert not plan.value


def test_filler_fails_loudly_without_gpu(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
    with pytest.raises(capi.CrenderError, match="no CPU fallback"):
        AdvancedPixelBufferFiller(64, 64)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import or load it."""
    pkg = os.path.join(ROOT, "cython3dmodelrenderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "liboracle" not in text and "from oracle" not in text \
                    and "import oracle" not in text, os.path.join(dirpath, fn)


# ---- Model (input producer) ------------------------------------------------------------
CUBE_OBJ = """# unit cube, quads
v 0 1 1
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 0
v 0 0 0
v 1 0 0
v 1 1 0
f 1 2 3 4
f 8 7 6 5
f 4 3 7 8
f 5 1 4 8
f 5 6 2 1
f 2 6 7 3
"""


def test_model_parses_and_fan_triangulates(tmp_path):
    from cython3dmodelrenderer_amd.data_structures import Model
    from cython3dmodelrenderer_amd.scenes import fit_model, load_fixture
    p = tmp_path / "cube.obj"
    p.write_text(CUBE_OBJ)
    m = Model.read_model(str(p))
    assert m.n_vertices() == 8 and m.n_triangles() == 12
    assert m._colors_by_triangles is None            # untextured (reference: model.py:135-141)
    assert m._vertices_by_triangles.dtype == np.float32
    assert m._vertices_by_triangles.shape == (12, 3, 3)
    assert (m._triangles_vertices[0] == [0, 1, 2]).all() and (m._triangles_vertices[1] == [0, 2, 3]).all()
    fit_model(m)
    m.set_uniform_color()
    tri, col, nrm = load_fixture("cube_inputs.npz")   # made from the reference's cube.obj asset
    assert_bit_equal(m._vertices_by_triangles, tri, "cube vertices")
    assert_bit_equal(m._normals_by_triangles, nrm, "cube normals")
    assert_bit_equal(m._colors_by_triangles, col, "cube colours")
    np.testing.assert_allclose(np.linalg.norm(m._normals, axis=-1), 1, atol=1e-6)


def test_model_texture_negative_indices_and_transforms(tmp_path):
    from PIL import Image
    from cython3dmodelrenderer_amd.data_structures import Model
    img = np.zeros((4, 4, 4), np.uint8)
    img[..., 0] = np.arange(16).reshape(4, 4) * 10      # R
    img[..., 1] = 7                                     # G
    img[..., 2] = 200                                   # B
    img[..., 3] = 128                                   # alpha is dropped
    Image.fromarray(img, "RGBA").save(tmp_path / "t.png")
    (tmp_path / "m.mtl").write_text("newmtl a\nmap_Kd t.png\n")
    (tmp_path / "m.obj").write_text(
        "mtllib m.mtl\nv 0 0 1\nv 1 0 1\nv 0 1 1\nv 1 1 2\n"
        "vt 0.1 0.9\nvt 0.9 0.9\nvt 0.1 0.1\nvt 0.99 0.01\n"
        "f 1/1 2/2 3/3\nf -3/2 -1/4 -2/3\n")
    m = Model.read_model(str(tmp_path / "m.obj"))
    assert m.n_triangles() == 2
    assert (m._triangles_vertices[1] == [-3, -1, -2]).all()          # kept negative, numpy resolves
    assert_bit_equal(m._vertices_by_triangles[1], m._vertices[[1, 3, 2]], "negative indices")
    c = m._colors_by_triangles
    assert c.shape == (2, 3, 3) and c.dtype == np.float32
    assert tuple(c[0, 0]) == (200.0, 7.0, 0.0)          # BGR; (u, v) = (.1, .9) -> texel row 0, col 0
    assert tuple(c[1, 1]) == (200.0, 7.0, 150.0)        # (.99, .01) -> row 3, col 3 -> R = 150
    before = m._vertices.copy()
    m.rotate([0, 0, 90])
    np.testing.assert_allclose(m._vertices[:, 0], before[:, 1], atol=1e-6)
    np.testing.assert_allclose(m._vertices[:, 1], -before[:, 0], atol=1e-6)
    m.shift([1, 2, 3])
    np.testing.assert_allclose(m._vertices[:, 2], before[:, 2] + 3, atol=1e-6)
    span = m.get_max_span()
    m.scale(2.0)
    np.testing.assert_allclose(m.get_max_span(), 2 * span, rtol=1e-5)


def test_guro_illumination_numpy_form():
    from cython3dmodelrenderer_amd.illumination import GuroIllumination, NoIllumination
    rng = np.random.default_rng(0)
    n = rng.standard_normal((8, 9, 3)).astype(np.float32)
    n[0, 0] = 0
    c = rng.uniform(0, 255, (8, 9, 3)).astype(np.float32)
    light = GuroIllumination([0, 0, 1])
    assert light.light_direction.dtype == np.float32 and tuple(light.light_direction) == (0, 0, -1)
    want = c * np.clip(-n[..., 2:3] / (np.linalg.norm(n, axis=-1, keepdims=True) + np.float32(1e-6)), 0, 1)
    got = c.copy()
    light.draw_illumination(got, n)
    np.testing.assert_allclose(got, want, rtol=1e-6)
    assert (got[0, 0] == 0).all()
    keep = c.copy()
    NoIllumination().draw_illumination(keep, n)
    assert (keep == c).all()


def test_numpy_three_element_sum_order():
    """The HIP Guro kernel hard-codes numpy's float32 reduction order for a length-3 last
    axis: (a0 + a1) + a2.  If a numpy upgrade changes it, this fails first."""
    rng = np.random.default_rng(1)
    a = (rng.standard_normal((4096, 3)) * 10 ** rng.uniform(-3, 3, (4096, 3))).astype(np.float32)
    s = np.sum(a, axis=-1)
    alt1 = a[:, 0] + (a[:, 1] + a[:, 2])
    alt2 = (a[:, 0] + a[:, 1]) + a[:, 2]
    assert (alt1 != alt2).any()
    assert (s == alt2).all()
    n = np.linalg.norm(a, axis=-1)
    q = a * a
    assert (n == np.sqrt((q[:, 0] + q[:, 1]) + q[:, 2])).all()


# ---- two-rank row strips over gloo -----------------------------------------------------
_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from cython3dmodelrenderer_amd import distributed as D
from cython3dmodelrenderer_amd import scenes
from oracle import oracle as O
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
H = W = 96
tri, col, nrm = scenes.load_fixture("trex_inputs.npz")
y0, y1 = D.strip_rows(H, 2, rank)
assert (y0, y1) == ((0, 48) if rank == 0 else (48, 96))
# each rank renders only its strip (here with the CPU oracle standing in for the GPU filler:
# this test covers the sharding + all-gather logic, not the kernels)
f = O.OracleFiller(H, W, fov=45)
f.render_arrays(tri, col, nrm, y0=y0, y1=y1)
bufs = [torch.from_numpy(b) for b in (f.z_buffer, f.color_buffer, f.normals_buffer)]
D.all_gather_strips(bufs, H, rank, 2)
full = O.OracleFiller(H, W, fov=45)
full.render_arrays(tri, col, nrm)
for got, want in zip(bufs, (full.z_buffer, full.color_buffer, full.normals_buffer)):
    assert np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_row_strips_gloo(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {r} ok" in out


_RAGGED_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from cython3dmodelrenderer_amd import distributed as D
from cython3dmodelrenderer_amd import scenes
from oracle import oracle as O
world = 3
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=world)
rank = dist.get_rank()
tri, col, nrm = scenes.load_fixture("trex_inputs.npz")
# ragged frames with sub-strips: ranks own different numbers of rows and of NON-EMPTY sub-strips, yet
# every rank must take part in every sub-strip's collective (the oracle stands in for the GPU filler)
for H, W, chunks in ((50, 64, 4), (5, 40, 4), (2, 16, 3), (96, 96, 1), (97, 33, 2)):
    n = D.chunk_count(H, world, chunks)
    assert n == max(1, min(chunks, D.strip_height(H, world)))
    f = O.OracleFiller(H, W, fov=45)
    mine = 0
    for k in range(n):
        a, b = D.substrip_rows(H, world, rank, k, chunks)
        if b > a:
            f.render_arrays(tri, col, nrm, y0=a, y1=b)
            mine += b - a
    y0, y1 = D.strip_rows(H, world, rank)
    assert mine == y1 - y0
    bufs = [torch.from_numpy(b) for b in (f.z_buffer, f.color_buffer, f.normals_buffer)]
    for k in range(n):                       # exactly n collectives per plane on EVERY rank
        D.all_gather_substrips(bufs, H, rank, world, k, chunks)
    full = O.OracleFiller(H, W, fov=45)
    full.render_arrays(tri, col, nrm)
    for got, want in zip(bufs, (full.z_buffer, full.color_buffer, full.normals_buffer)):
        assert np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32)), (H, W, chunks)
# north_star's broadcast-of-projected-vertices variant, host side of it: rank 0 projects, everybody
# receives the 36 T bytes and rasterizes its strip from them
H = W = 64
f = O.OracleFiller(H, W, fov=45)
proj = torch.from_numpy(O.project(tri, f.proj_mat, W, H) if rank == 0 else np.zeros_like(tri))
dist.broadcast(proj, src=0)
assert np.array_equal(proj.numpy().view(np.uint32), O.project(tri, f.proj_mat, W, H).view(np.uint32))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_three_rank_ragged_substrips_gloo(tmp_path):
    """Every rank issues the same number of collectives whatever its share of rows (ragged last
    strip, empty sub-strips, more sub-strips asked for than a strip has rows)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "ragged_worker.py"
    script.write_text(_RAGGED_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(3)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {r} ok" in out


_EIGHT_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from cython3dmodelrenderer_amd import distributed as D
from cython3dmodelrenderer_amd import scenes
from oracle import oracle as O
world = 8
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=world)
rank = dist.get_rank()
torch.set_num_threads(1)
tri, col, nrm = scenes.load_fixture("trex_inputs.npz")
# the SCALE run's shape in small: eight equal strips (H a multiple of 8), every exchange choice, through
# the calls StripRenderer.render_frame makes (the oracle stands in for the GPU filler).  Equal strips
# of CPU tensors take Gatherer.gather_blocks' in-place branch — what RCCL runs on device tensors
H, W = 64, 48
y0, y1 = D.strip_rows(H, world, rank)
assert y1 - y0 == 8
f = O.OracleFiller(H, W, fov=45)
f.render_arrays(tri, col, nrm, y0=y0, y1=y1)
full = O.OracleFiller(H, W, fov=45)
full.render_arrays(tri, col, nrm)
g = D.Gatherer(None)
taken = []
real = dist.all_gather_into_tensor
def spy(out, inp, group=None):
    taken.append(out.data_ptr())
    return real(out, inp, group=group)
dist.all_gather_into_tensor = spy
rows = lambda r: D.strip_rows(H, world, r)
# exchange = planes: z, colour, normal in place
planes = [torch.from_numpy(b.copy()) for b in (f.z_buffer, f.color_buffer, f.normals_buffer)]
for p in planes:
    g.gather_blocks(p, rank, world, D.strip_height(H, world), rows)
assert taken == [p.data_ptr() for p in planes], "equal strips gather straight into the planes"
for got, want in zip(planes, (full.z_buffer, full.color_buffer, full.normals_buffer)):
    assert np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32))
# exchange = color
c = torch.from_numpy(f.color_buffer.copy())
g.gather_blocks(c, rank, world, D.strip_height(H, world), rows)
assert np.array_equal(c.numpy().view(np.uint32), full.color_buffer.view(np.uint32))
# exchange = present: the flipped uint8 image, blocks in descending row order (staged)
image = torch.zeros((H, W, 3), dtype=torch.uint8)
image[H - y1: H - y0] = torch.from_numpy(f.color_buffer[y0:y1][::-1].astype("uint8").copy())
n_before = len(taken)
D.gather_present(g, image, H, rank, world)
assert taken[n_before] != image.data_ptr()
assert np.array_equal(image.numpy(), full.color_buffer[::-1].astype("uint8"))
assert D.exchange_bytes_received("planes", H, W, world, rank) == 28 * (H - 8) * W
assert D.exchange_bytes_received("present", H, W, world, rank) == 3 * (H - 8) * W
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_eight_rank_equal_strips_every_exchange_gloo(tmp_path):
    """world_size 8, eight equal strips — the shape of the driver's SCALE run — for planes / color /
    present; the planes and colour exchanges must take the in-place branch of Gatherer.gather_blocks."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "eight_worker.py"
    script.write_text(_EIGHT_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(8)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {r} ok" in out


def test_profile_figures_are_quoted_only_for_the_sources_they_were_measured_on(tmp_path, monkeypatch):
    """bench.py's roofline.traffic and committed rocprofv3 average come from profiles/*.json; they carry the
    fingerprint of the kernel sources they were measured on and are dropped (null) for any other."""
    import json
    import bench
    from cython3dmodelrenderer_amd import _build
    sha = _build.source_sha16()
    assert len(sha) == 16 and sha == bench.csrc_sha16()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    for stamp, want in ((sha, True), ("0" * 16, False), (None, False)):
        doc_t = {"trex1024": {"k_raster": {"hbm_bytes_per_launch": 123.0}, "k_frame_owners": {"hbm_bytes_per_launch": 456.0}}}
        doc_k = {"trex1024": {"k_raster": 11000.0, "k_frame_owners_one_stream": 13000.0}}
        if stamp is not None:
            doc_t["csrc_sha16"] = doc_k["csrc_sha16"] = stamp
        (prof / "traffic.json").write_text(json.dumps(doc_t))
        (prof / "kernel_avg.json").write_text(json.dumps(doc_k))
        # (each raster kernel has its own figures: a line quotes those of ITS roofline.kernel)
        assert bench.load_traffic("trex1024", "k_raster<16,true,0>") == (123.0 if want else None)
        assert bench.load_traffic("trex1024", "k_frame<32,true,1>") == (456.0 if want else None)
        assert bench.load_traffic("trex1024", "k_frame<32,true,0>") is None
        assert bench.load_rocprof_avg_ms("trex1024", bench.profile_key("k_raster<32,true,0>")) == (0.011 if want else None)
        assert bench.load_rocprof_avg_ms("trex1024", bench.profile_key("k_frame<32,true,1>") + "_one_stream") == (0.013 if want else None)
    # the fingerprint follows the sources: one more byte in a translation unit changes it
    monkeypatch.setattr(_build, "HIPCC_FLAGS", _build.HIPCC_FLAGS + ["-DX"])
    assert _build.source_sha16() != sha


def test_strip_rows_partition():
    from cython3dmodelrenderer_amd.distributed import strip_rows
    for H, n in [(8192, 8), (1024, 8), (1000, 3), (7, 8), (64, 1)]:
        rows = [strip_rows(H, n, r) for r in range(n)]
        assert rows[0][0] == 0 and rows[-1][1] == H
        assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        assert len({y1 - y0 for y0, y1 in rows}) == 1 or H % n != 0


def test_renderer_takes_an_overridden_illumination_not_the_inherited_device_form():
    """The reference's only shading hook is draw_illumination (cy/renderer.py:47-49).  A subclass of
    GuroIllumination that overrides it alone must get ITS shading in the default mode, not the
    parent's device kernel (advisor, round 3)."""
    from cython3dmodelrenderer_amd.illumination import GuroIllumination
    from cython3dmodelrenderer_amd.illumination.illumination_drawer import NoIllumination
    from cython3dmodelrenderer_amd.renderer import Renderer, _device_form_is_the_same_shading

    class Half(GuroIllumination):
        def draw_illumination(self, color_buffer, n_buffer):
            color_buffer *= np.float32(0.5)

    class Both(GuroIllumination):
        def draw_illumination(self, color_buffer, n_buffer):
            color_buffer *= np.float32(0.25)

        def draw_illumination_device(self, filler):
            filler.device_calls += 1
            return True

    assert _device_form_is_the_same_shading(GuroIllumination())
    assert _device_form_is_the_same_shading(NoIllumination())
    assert not _device_form_is_the_same_shading(Half())
    assert _device_form_is_the_same_shading(Both())

    class FakeFiller:
        def __init__(self):
            self.color = np.full((2, 2, 3), 8.0, np.float32)
            self.normals = np.zeros((2, 2, 3), np.float32)
            self.device_calls = 0

        def render_model(self, model, **kw):
            pass

        def get_color_buffer(self):
            return self.color

        def get_normals_buffer(self):
            return self.normals

        def get_color_tensor(self):
            return self.color

    f = FakeFiller()
    out = Renderer(f, Half()).render(object())
    assert np.all(out == 4.0) and f.device_calls == 0
    f = FakeFiller()
    Renderer(f, Both()).render(object())
    assert f.device_calls == 1 and np.all(f.color == 8.0)
    f = FakeFiller()
    Renderer(f, Both(), on_device=False).render(object())
    assert f.device_calls == 0 and np.all(f.color == 2.0)


def test_numpy_dot_of_3_vectors():
    """What the device's vertex-normal kernels assume of numpy (row f2): np.dot of two float32
    3-vectors — and np.linalg.norm through it — forms float32 products and adds them in a double
    accumulator, rounding once (OpenBLAS sdot).  If a numpy build ever differs, this fails before
    the GPU tests do."""
    rng = np.random.default_rng(0)
    n = 20000
    a = rng.standard_normal((n, 3)).astype(np.float32)
    b = rng.standard_normal((n, 3)).astype(np.float32)
    a[: n // 2] /= np.linalg.norm(a[: n // 2], axis=1, keepdims=True)
    b[: n // 2] = a[: n // 2] + rng.standard_normal((n // 2, 3)).astype(np.float32) * np.float32(1e-4)
    ref = np.array([np.dot(a[i], b[i]) for i in range(n)], dtype=np.float32)
    p = a * b
    mine = (p[:, 0].astype(np.float64) + p[:, 1].astype(np.float64) + p[:, 2].astype(np.float64)).astype(np.float32)
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))
    norms = np.array([np.linalg.norm(a[i]) for i in range(2000)], dtype=np.float32)
    sq = a[:2000] * a[:2000]
    mine = np.sqrt((sq[:, 0].astype(np.float64) + sq[:, 1].astype(np.float64) + sq[:, 2].astype(np.float64)).astype(np.float32))
    assert np.array_equal(mine.view(np.uint32), norms.view(np.uint32))
