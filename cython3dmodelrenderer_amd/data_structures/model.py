"""Triangle-mesh container that feeds the rasterizer.

Host-side mirror of the reference's ``Model`` (reference:
crender/cy/data_structures/model.py:5-329).  The rasterizer reads exactly three
attributes off it (reference: advanced_pixel_buffer_filler.pyx:94-96):

    _vertices_by_triangles  float32 [T, 3, 3]   xyz per corner
    _colors_by_triangles    float32 [T, 3, 3]   BGR 0..255 per corner, or None
    _normals_by_triangles   float32 [T, 3, 3]   vertex normal per corner

This module is an input producer: it runs once per
model, not per frame.  Its arithmetic follows the reference's numpy call sequence
(float32 cross products, per-vector ``np.linalg.norm``/``np.dot``, float64 rotation
matrices applied with ``np.matmul``) so that the arrays it hands the rasterizer are
bit-identical to the reference's for the same .obj file and numpy build.

The texture is decoded with PIL into the channel order the reference's loader yields
(BGR, alpha dropped; reference: model.py:114-116 uses ``cv2.imread``); PNG decoding is
lossless, so the sampled colours are the same bytes.
"""
from __future__ import annotations

import os

import numpy as np


def _load_texture_bgr(path):
    """uint8 [h, w, 3] in BGR order, or None if the file cannot be decoded."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"))
    except Exception:
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _obj_index(token):
    # .obj indices are 1-based; negative ones count from the end and are kept as they
    # are, which numpy's negative indexing then resolves (reference: model.py:281-285).
    i = int(token)
    return i - 1 if i > 0 else i


def _parse_face(data):
    """Fan-triangulate one ``f`` record (reference: model.py:287-316).

    Returns three lists of index triples (vertex, texture, normal); a triple is None
    when any of its corners lacks that index.
    """
    corners = data.split()
    vs, vts, vns = [], [], []
    for k in range(len(corners) - 2):
        tv, tt, tn = [], [], []
        for corner in (corners[0], corners[k + 1], corners[k + 2]):
            v, vt, vn = (corner + "//").split("/")[:3]
            tv.append(_obj_index(v))
            if tt is not None:
                tt = None if vt == "" else tt + [_obj_index(vt)]
            if tn is not None:
                tn = None if vn == "" else tn + [_obj_index(vn)]
        vs.append(tv)
        vts.append(tt)
        vns.append(tn)
    return vs, vts, vns


def _split_record(line):
    if not line or line[0] == "#":
        return None
    parts = line.split(" ", 1)
    return parts if len(parts) == 2 else None


def _texture_name_from_mtl(path, origin):
    """Last ``map_Kd`` entry of a material file (reference: model.py:79-112)."""
    name = None
    try:
        with open(path.strip(), "r") as fh:
            for line in fh:
                rec = _split_record(line)
                if rec and rec[0] == "map_Kd":
                    name = rec[1]
    except Exception as exc:  # missing .mtl: keep going without a texture
        print(f"Error occurred while parsing material file of object file '{origin}':")
        print(exc)
        print("Material info will be ignored")
    return name


def _dir_prefix(filename):
    head = filename.rsplit("/", 1)
    return head[0] + "/" if len(head) == 2 else ""


def _unit(n):
    length = np.linalg.norm(n)
    return n if length == 0 else n / length


def _vertex_normals(vertices, faces):
    """Per-vertex normals: normalised mean of the distinct unit face normals.

    Same accumulation as the reference (model.py:175-208): faces visited in index
    order, a face normal joins a vertex's set unless an already collected one has a
    float32 dot product >= 1 with it, and the set is averaged then normalised.
    """
    tri = vertices[faces]                                     # [T, 3, 3]
    raw = -np.cross(tri[:, 1] - tri[:, 0], tri[:, 1] - tri[:, 2])
    collected = [[] for _ in range(len(vertices))]
    for t in range(len(faces)):
        n = _unit(raw[t])
        for v in faces[t]:
            bucket = collected[v]
            if not any(np.dot(m, n) >= 1 for m in bucket):
                bucket.append(n)
    out = np.zeros((len(vertices), 3), np.float32)
    for v, bucket in enumerate(collected):
        if bucket:
            out[v] = _unit(np.mean(np.stack(bucket), axis=0))
    return out


def _rot2(angle_deg):
    a = angle_deg * (np.pi / 180)
    return np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])


class Model:
    def __init__(self, vertices, triangles_vertices,
                 texture_coords=None, triangles_texture_coords=None, texture=None,
                 normals=None, triangles_normals=None,
                 recalculate_normals=True, invert_calculated_normals=False):
        given_n = given_tn = None
        if normals is not None and triangles_normals is not None:
            given_n = np.array(normals, dtype=np.float32)
            given_tn = np.array(triangles_normals, dtype=np.int32)
        self._set_geometry(np.array(vertices, dtype=np.float32),
                           np.array(triangles_vertices, dtype=np.int32),
                           given_n, given_tn, recalculate_normals, invert_calculated_normals)

        self._texture_coords = self._triangles_texture_coords = self._texture = None
        self._colors = self._colors_by_triangles = None
        if texture_coords is not None and triangles_texture_coords is not None \
                and texture is not None:
            # one colour per texture coordinate, nearest texel, v axis flipped
            # (reference: model.py:143-151)
            self._texture_coords = np.array(texture_coords, dtype=np.float32)
            self._triangles_texture_coords = np.array(triangles_texture_coords, dtype=np.int32)
            self._texture = np.array(texture)
            th, tw, _ = self._texture.shape
            uv = self._texture_coords
            row = np.clip(((1 - uv[:, 1]) * th).astype("int32"), 0, th - 1)
            colm = np.clip((uv[:, 0] * tw).astype("int32"), 0, tw - 1)
            self._colors = self._texture[row, colm].astype("float32")
            self._colors_by_triangles = self._colors[self._triangles_texture_coords]

    # -- construction from files ---------------------------------------------------
    @staticmethod
    def read_model(filename, silent=True, external_texture_filename=None,
                   recalculate_normals=True, invert_calculated_normals=False):
        """Parse a Wavefront .obj (+ .mtl / texture) — reference: model.py:7-77."""
        filename = filename.strip()
        v, vt, vn = [], [], []
        f_v, f_vt, f_vn = [], [], []
        texture = None
        if external_texture_filename is not None:
            texture = _load_texture_bgr(external_texture_filename.strip())
        with open(filename, "r") as fh:
            for lineno, line in enumerate(fh, 1):
                try:
                    rec = _split_record(line)
                    if rec is None:
                        continue
                    key, data = rec
                    if key == "v":
                        xyz = [float(t) for t in data.split()]
                        assert len(xyz) >= 3
                        v.append(xyz[:3])
                    elif key == "vt":
                        vt.append([float(t) for t in data.split()])
                    elif key == "vn":
                        xyz = [float(t) for t in data.split()]
                        assert len(xyz) == 3
                        vn.append(xyz)
                    elif key == "f":
                        tv, tt, tn = _parse_face(data)
                        f_v.extend(tv)
                        # one face without texture (normal) indices disables them for
                        # the whole model, for good
                        if tt.count(None) > 0:
                            f_vt = None
                        if f_vt is not None:
                            f_vt.extend(tt)
                        if tn.count(None) > 0:
                            f_vn = None
                        if f_vn is not None:
                            f_vn.extend(tn)
                    elif key == "mtllib" and texture is None:
                        base = _dir_prefix(filename)
                        mtl = (base if data[0] != "/" else "") + data
                        image = _texture_name_from_mtl(mtl, filename)
                        if image is not None:
                            image = (base if image[0] != "/" else "") + image
                            texture = _load_texture_bgr(image.strip())
                except Exception as exc:
                    if not silent:
                        raise RuntimeError(
                            f'Error occurred while parsing line #{lineno} of "{filename}"') from exc
        return Model(v, f_v, vt, f_vt, texture, vn, f_vn,
                     recalculate_normals, invert_calculated_normals)

    # -- geometry state ------------------------------------------------------------
    def _set_geometry(self, vertices, faces, normals, faces_n, recalc=True, invert=False):
        """reference: model.py:153-173."""
        self._vertices = vertices.astype("float32")
        self._triangles_vertices = faces
        self._vertices_by_triangles = self._vertices[faces]
        self._mean_vertex = self._vertices.mean(axis=0)
        self._max_span = np.max(np.linalg.norm(self._vertices - self._mean_vertex, axis=-1))
        if normals is not None and faces_n is not None and not recalc:
            self._normals = normals.astype("float32")
            self._triangles_normals = faces_n
        else:
            self._normals = _vertex_normals(self._vertices, faces)
            self._triangles_normals = faces
            if invert:
                self._normals *= -1
        self._normals_by_triangles = self._normals[self._triangles_normals]

    def shift(self, shift):
        self._set_geometry(self._vertices + shift, self._triangles_vertices,
                           self._normals, self._triangles_normals, recalc=False)

    def scale(self, scale_coef, keep_position=True):
        vtx = self._vertices            # scaled in place, in float32, like the reference
        if keep_position:
            vtx -= self._mean_vertex
            vtx *= scale_coef
            vtx += self._mean_vertex
        else:
            vtx *= scale_coef
        self._set_geometry(vtx, self._triangles_vertices,
                           self._normals, self._triangles_normals, recalc=False)

    def rotate(self, angles):
        """Rotate about x, then y, then z (degrees); normals are recomputed.
        reference: model.py:238-256 (float64 matrices, float32 vertices)."""
        assert len(angles) == 3
        ax, ay, az = angles
        rx, ry, rz = np.eye(3), np.eye(3), np.eye(3)
        rx[1:, 1:] = _rot2(ax)
        ry[::2, ::2] = _rot2(ay)
        rz[:2, :2] = _rot2(az)
        rot = np.matmul(np.matmul(rx, ry), rz)
        self._set_geometry(np.matmul(self._vertices, np.transpose(rot)),
                           self._triangles_vertices, None, None, recalc=True)

    # -- accessors -----------------------------------------------------------------
    def get_vertex(self, index):
        return (self._vertices[index],
                self._colors[index] if self._colors is not None else None,
                self._normals[index])

    def get_triangle(self, index):
        return (self._vertices_by_triangles[index],
                self._colors_by_triangles[index] if self._colors_by_triangles is not None else None,
                self._normals_by_triangles[index])

    def n_triangles(self):
        return len(self._triangles_vertices)

    def n_vertices(self):
        return len(self._vertices)

    def get_mean_vertex(self):
        return self._mean_vertex

    def get_max_span(self):
        return self._max_span

    def set_uniform_color(self, bgr=(255.0, 255.0, 255.0)):
        """Give an untextured model one colour (the reference leaves
        ``_colors_by_triangles`` None for such models and its filler then raises;
        its pure-Python renderer paints them white, py/renderer.py:54)."""
        T = self.n_triangles()
        self._colors_by_triangles = np.broadcast_to(
            np.asarray(bgr, np.float32), (T, 3, 3)).copy()
