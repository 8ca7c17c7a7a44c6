"""Scene input for the hot path: OBJ/MTL reader and the synthetic `blocks_restir` stand-in.

* `load_obj` restates what common/loader.hpp:11-66 gets from tinyobjloader v1.0.6
  (libs/tiny_obj_loader/tiny_obj_loader.h:890-951): polygons are triangulated as a FAN around
  the first vertex, one Triangle per fan triangle, in file order, with the face's `usemtl`
  material giving color = Kd and emissive = Ke. Triangle order = primID = light order.
* `make_blocks_restir` builds the documented stand-in for the reference's missing
  `assets/blocks_restir.obj`. Deterministic: integer LCG (seed 2024), only
  IEEE + - * / in binary64, one final rounding to binary32, no libm. The material table is
  the data of the reference's `assets/blocks_restir.mtl` (MIT, (c) the CEDEC-2024-RT
  authors): 41 emissive + 7 diffuse materials.
"""
import hashlib
import os

import numpy as np

from .types import TRIANGLE

# (name, Kd, Ke) — data of assets/blocks_restir.mtl
BLOCKS_RESTIR_MATERIALS = [
    ("Back", (0.053973, 0.053973, 0.053973), (0.0, 0.0, 0.0)),
    ("Black", (0.018829, 0.018829, 0.018829), (0.0, 0.0, 0.0)),
    ("Brown", (0.100000, 0.046767, 0.021787), (0.0, 0.0, 0.0)),
    ("Emmisive", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.004", (0.8, 0.8, 0.8), (20.027411, 120.0, 68.444496)),
    ("Emmisive.005", (0.8, 0.8, 0.8), (120.0, 12.442083, 107.111855)),
    ("Emmisive.006", (0.8, 0.8, 0.8), (36.973640, 120.0, 33.652977)),
    ("Emmisive.007", (0.8, 0.8, 0.8), (6.597945, 16.341084, 120.0)),
    ("Emmisive.008", (0.8, 0.8, 0.8), (120.0, 40.437397, 22.585363)),
    ("Emmisive.009", (0.8, 0.8, 0.8), (120.0, 40.437397, 22.585363)),
    ("Emmisive.010", (0.8, 0.8, 0.8), (62.007309, 24.863846, 120.0)),
    ("Emmisive.011", (0.8, 0.8, 0.8), (120.0, 40.437397, 22.585363)),
    ("Emmisive.012", (0.8, 0.8, 0.8), (20.027411, 120.0, 68.444496)),
    ("Emmisive.013", (0.8, 0.8, 0.8), (120.0, 12.442083, 107.111855)),
    ("Emmisive.014", (0.8, 0.8, 0.8), (120.0, 12.442083, 107.111855)),
    ("Emmisive.015", (0.8, 0.8, 0.8), (120.0, 15.857393, 1.082641)),
    ("Emmisive.016", (0.8, 0.8, 0.8), (42.537949, 50.0, 2.450186)),
    ("Emmisive.017", (0.8, 0.8, 0.8), (104.086250, 83.048935, 120.0)),
    ("Emmisive.018", (0.8, 0.8, 0.8), (120.0, 108.947151, 14.899497)),
    ("Emmisive.019", (0.8, 0.8, 0.8), (34.875523, 10.179805, 120.0)),
    ("Emmisive.020", (0.8, 0.8, 0.8), (7.090583, 120.0, 3.423895)),
    ("Emmisive.021", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.022", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.023", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.024", (0.8, 0.8, 0.8), (117.972893, 120.0, 115.965866)),
    ("Emmisive.025", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.026", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.027", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.029", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.031", (0.8, 0.8, 0.8), (114.499733, 120.0, 118.940681)),
    ("Emmisive.033", (0.8, 0.8, 0.8), (120.0, 120.0, 120.0)),
    ("Emmisive.035", (0.8, 0.8, 0.8), (120.0, 2.742580, 1.2)),
    ("Emmisive.036", (0.8, 0.8, 0.8), (120.0, 1.2, 1.2)),
    ("Emmisive.037", (0.8, 0.8, 0.8), (120.0, 1.2, 1.2)),
    ("Emmisive.038", (0.8, 0.8, 0.8), (120.0, 1.2, 2.220691)),
    ("Emmisive.039", (0.8, 0.8, 0.8), (120.0, 1.2, 1.074211)),
    ("Emmisive.040", (0.8, 0.8, 0.8), (120.0, 1.2, 1.2)),
    ("Emmisive.041", (0.8, 0.8, 0.8), (120.0, 2.053821, 6.980591)),
    ("Emmisive.042", (0.8, 0.8, 0.8), (2.022431, 1.2, 120.0)),
    ("Emmisive.043", (0.8, 0.8, 0.8), (1.2, 7.758688, 120.0)),
    ("Emmisive.044", (0.8, 0.8, 0.8), (1.637471, 1.2, 120.0)),
    ("Emmisive.045", (0.8, 0.8, 0.8), (1.2, 1.2, 120.0)),
    ("Emmisive.046", (0.8, 0.8, 0.8), (120.0, 42.767715, 117.070190)),
    ("FloorMaterial", (0.538017, 0.523392, 0.489841), (0.0, 0.0, 0.0)),
    ("Green", (0.057167, 0.136217, 0.024592), (0.0, 0.0, 0.0)),
    ("WeakLight", (0.8, 0.8, 0.8), (0.5, 0.5, 0.5)),
    ("Yellow", (0.617207, 0.419342, 0.179603), (0.0, 0.0, 0.0)),
    ("white", (0.546776, 0.508329, 0.527966), (0.0, 0.0, 0.0)),
]
_MAT = {m[0]: i for i, m in enumerate(BLOCKS_RESTIR_MATERIALS)}
_EMISSIVE_IDS = [i for i, m in enumerate(BLOCKS_RESTIR_MATERIALS) if m[0].startswith("Emmisive")]
assert len(BLOCKS_RESTIR_MATERIALS) == 48 and len(_EMISSIVE_IDS) == 40  # + WeakLight = 41 emissive

# camera "blocks_restir.obj 1" of examples/10_restir_di/10_restir_di.cpp:188-189
BLOCKS_RESTIR_EYE = (-0.579885, 22.194597, -6.567105)
BLOCKS_RESTIR_LOOKAT = (5.224952, 20.847435, 1.431192)
# common/misc.hpp:217-218 (default CameraControl, used by 04_ao on cornellbox1)
DEFAULT_EYE = (8.0, 8.0, 8.0)
DEFAULT_LOOKAT = (0.0, 0.0, 0.0)
# chosen camera for 07_pt on cornellbox2 (the reference has none)
CORNELLBOX_EYE = (0.0, 2.7, 8.0)
CORNELLBOX_LOOKAT = (0.0, 2.7, -2.8)


# --------------------------------------------------------------------------- OBJ / MTL
def load_mtl(path):
    mats = {}
    cur = None
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "newmtl":
                cur = " ".join(t[1:])
                mats[cur] = {"Kd": (0.0, 0.0, 0.0), "Ke": (0.0, 0.0, 0.0)}
            elif cur is not None and t[0] in ("Kd", "Ke") and len(t) >= 4:
                mats[cur][t[0]] = (float(t[1]), float(t[2]), float(t[3]))
    return mats


def load_obj(path, mtl_basedir=None):
    """OBJ -> array of TRIANGLE, triangle order as common/loader.hpp:11-66 produces it."""
    base = mtl_basedir if mtl_basedir is not None else os.path.dirname(path)
    verts = []
    mats = {}
    cur_mat = None
    tris = []
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                verts.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "mtllib":
                for name in t[1:]:
                    p = os.path.join(base, name)
                    if os.path.exists(p):
                        mats.update(load_mtl(p))
            elif t[0] == "usemtl":
                cur_mat = " ".join(t[1:])
            elif t[0] == "f":
                idx = []
                for w in t[1:]:
                    vi = int(w.split("/")[0])
                    idx.append(vi - 1 if vi > 0 else len(verts) + vi)
                m = mats.get(cur_mat, {"Kd": (0.0, 0.0, 0.0), "Ke": (0.0, 0.0, 0.0)})
                # triangle fan around the first vertex (tiny_obj_loader.h:908-931)
                for k in range(2, len(idx)):
                    tris.append((idx[0], idx[k - 1], idx[k], m["Kd"], m["Ke"]))
    out = np.zeros(len(tris), dtype=TRIANGLE)
    v = np.asarray(verts, dtype=np.float32)
    for n, (a, b, c, kd, ke) in enumerate(tris):
        out["v"][n, 0] = v[a]
        out["v"][n, 1] = v[b]
        out["v"][n, 2] = v[c]
        out["color"][n] = kd
        out["emissive"][n] = ke
    return out


def light_indices(triangles):
    """examples/10_restir_di/10_restir_di.cpp:196-205."""
    e = triangles["emissive"]
    return np.nonzero((e[:, 0] > 0) | (e[:, 1] > 0) | (e[:, 2] > 0))[0].astype(np.uint32)


def scene_sha256(triangles):
    return hashlib.sha256(np.ascontiguousarray(triangles).tobytes()).hexdigest()


# ------------------------------------------------------------- synthetic blocks_restir
class _LCG:
    def __init__(self, seed):
        self.x = seed & 0xFFFFFFFF

    def next(self):
        self.x = (self.x * 1664525 + 1013904223) & 0xFFFFFFFF
        return self.x >> 8  # 24 bits

    def below(self, n):
        return self.next() % n


_FONT = {
    "C": ["01111", "10000", "10000", "10000", "10000", "10000", "01111"],
    "E": ["11111", "10000", "10000", "11110", "10000", "10000", "11111"],
    "D": ["11110", "10001", "10001", "10001", "10001", "10001", "11110"],
    "2": ["11110", "00001", "00001", "01110", "10000", "10000", "11111"],
    "0": ["01110", "10001", "10001", "10001", "10001", "10001", "01110"],
    "4": ["10001", "10001", "10001", "11111", "00001", "00001", "00001"],
}

_NGON = 16
# cos/sin of k*22.5 degrees as binary64 literals (no libm at run time)
_C = [1.0, 0.9238795325112867, 0.7071067811865476, 0.3826834323650898, 0.0]
_COS = [_C[0], _C[1], _C[2], _C[3], _C[4], -_C[3], -_C[2], -_C[1], -_C[0], -_C[1], -_C[2], -_C[3], -_C[4], _C[3], _C[2], _C[1]]
_SIN = [_C[4], _C[3], _C[2], _C[1], _C[0], _C[1], _C[2], _C[3], _C[4], -_C[3], -_C[2], -_C[1], -_C[0], -_C[1], -_C[2], -_C[3]]


class _Builder:
    def __init__(self):
        self.tris = []  # list of (3x3 float64 array, material id)

    def tri(self, a, b, c, mat):
        self.tris.append((a, b, c, mat))

    def quad(self, p0, p1, p2, p3, mat):
        self.tri(p0, p1, p2, mat)
        self.tri(p0, p2, p3, mat)

    def patch(self, fn, nu, nv, mat):
        """fn(i, j) -> point on an (nu+1) x (nv+1) grid."""
        pts = [[fn(i, j) for j in range(nv + 1)] for i in range(nu + 1)]
        for i in range(nu):
            for j in range(nv):
                self.quad(pts[i][j], pts[i + 1][j], pts[i + 1][j + 1], pts[i][j + 1], mat)

    def box(self, lo, hi, mat):
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        p = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0),
             (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
        for f in ((0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)):
            self.quad(p[f[0]], p[f[1]], p[f[2]], p[f[3]], mat)

    def stud(self, base, axis, r, h, mat):
        """16-gon prism on `base`; axis 'y' (up) or 'z' (towards -z). Side + cap."""
        bx, by, bz = base
        ring0, ring1 = [], []
        for k in range(_NGON):
            c, s = r * _COS[k], r * _SIN[k]
            if axis == "y":
                ring0.append((bx + c, by, bz + s))
                ring1.append((bx + c, by + h, bz + s))
            else:
                ring0.append((bx + c, by + s, bz))
                ring1.append((bx + c, by + s, bz - h))
        for k in range(_NGON):
            k1 = (k + 1) % _NGON
            self.quad(ring0[k], ring0[k1], ring1[k1], ring1[k], mat)
        for k in range(2, _NGON):
            self.tri(ring1[0], ring1[k - 1], ring1[k], mat)

    def brick(self, x0, y0, z0, nx, nz, mat, height=1.2):
        self.box((x0, y0, z0), (x0 + nx, y0 + height, z0 + nz), mat)
        for i in range(nx):
            for j in range(nz):
                self.stud((x0 + i + 0.5, y0 + height, z0 + j + 0.5), "y", 0.3, 0.2, mat)


def make_blocks_restir(detail=1.0):
    """Synthetic stand-in for assets/blocks_restir.obj. `detail` < 1 thins the foliage (tests)."""
    b = _Builder()
    rng = _LCG(2024)
    M = _MAT

    # floor (y = 0) and curved backdrop, local frame: camera at (0, 22.19, 0) looking along +z
    b.patch(lambda i, j: (-100.0 + 200.0 * i / 64.0, 0.0, -30.0 + 130.0 * j / 64.0), 64, 64, M["FloorMaterial"])

    def arc(i, j):
        t = j / 32.0  # rational quarter circle: no libm
        d = 1.0 + t * t
        return (-100.0 + 200.0 * i / 32.0, 30.0 - 30.0 * (1.0 - t * t) / d, 100.0 + 30.0 * (2.0 * t) / d)

    b.patch(arc, 32, 32, M["Back"])
    b.patch(lambda i, j: (-100.0 + 200.0 * i / 32.0, 30.0 + 60.0 * j / 8.0, 130.0), 32, 8, M["Back"])
    # weak ceiling light
    b.patch(lambda i, j: (-40.0 + 80.0 * i / 8.0, 75.0, 0.0 + 80.0 * j / 8.0), 8, 8, M["WeakLight"])

    # two stud panels at z = 70 facing the camera (-z); stud pitch 1.5
    def panel(x_left, text, mats_for_glyph, plate_mat, stud_mat):
        cols, rows, pitch, zf = 30, 12, 1.5, 70.0
        y0 = 16.0
        b.box((x_left - cols * pitch, y0, zf), (x_left, y0 + rows * pitch, zf + 1.5), plate_mat)
        lit = {}
        col = 1
        for gi, ch in enumerate(text):
            glyph = _FONT[ch]
            for r in range(7):
                for c in range(5):
                    if glyph[r][c] == "1":
                        lit[(col + c, rows - 3 - r)] = mats_for_glyph(gi, c, r)
            col += 6
        hp = 0.5 * pitch
        for i in range(cols):  # i grows towards -x = camera right, so the text reads correctly
            for j in range(rows):
                cx = x_left - (i + 0.5) * pitch
                cy = y0 + (j + 0.5) * pitch
                m = lit.get((i, j))
                if m is None:
                    b.stud((cx, cy, zf), "z", 0.45, 0.3, stud_mat)
                else:
                    zt = zf - 0.015
                    b.quad((cx - hp, cy - hp, zt), (cx + hp, cy - hp, zt), (cx + hp, cy + hp, zt), (cx - hp, cy + hp, zt), m)
                    b.stud((cx, cy, zt), "z", 0.45, 0.3, m)

    rainbow = lambda gi, c, r: _EMISSIVE_IDS[(gi * 7 + c * 3 + r * 5) % 40]
    digits = [M["Emmisive"], M["Emmisive.021"], M["Emmisive.036"], M["Emmisive.045"]]
    panel(49.5, "CEDEC", rainbow, M["Black"], M["Black"])
    panel(-4.5, "2024", lambda gi, c, r: digits[gi], M["white"], M["white"])

    # bonsai: pot, trunk, foliage pads (the camera looks over the crown towards the panels)
    for i in range(-3, 3):
        for j in range(0, 3):
            b.brick(4 * i, 0.0 + 1.2 * (j % 2), 15.0 + 4.0 * j, 4, 4, M["Black"])
    tx, tz = 0, 20
    for k in range(9):
        b.brick(tx - 1, 2.4 + 1.2 * k, tz - 1, 2, 2, M["Brown"])
        tx += rng.below(3) - 1
        tz += rng.below(3) - 1
        tx = max(-3, min(3, tx))
        tz = max(17, min(23, tz))
    pads = [(-6, 6.0, 16, 8), (6, 7.2, 24, 8), (-2, 8.4, 27, 7), (5, 9.6, 15, 7), (-7, 10.8, 22, 7),
            (1, 12.0, 20, 8), (8, 13.2, 28, 6), (-4, 14.4, 14, 6), (3, 15.6, 24, 6), (-1, 16.8, 18, 5)]
    keep = int(256 * min(1.0, max(0.0, detail)))
    for (cx, cy, cz, rad) in pads:
        for layer in range(3):
            r = rad - layer
            nx, nz = (4, 2) if layer % 2 == 0 else (2, 4)
            for ix in range(-r, r, nx):
                for iz in range(-r, r, nz):
                    mx, mz = ix + nx * 0.5, iz + nz * 0.5
                    u = rng.below(256)
                    if mx * mx + mz * mz > r * r:
                        continue
                    if u >= keep or u >= 232:
                        continue
                    mat = M["Green"] if rng.below(16) else M["Yellow"]
                    b.brick(cx + ix, cy + 1.2 * layer, cz + iz, nx, nz, mat)

    # local -> world: rotate about y so local +z is the camera heading, translate to the eye's xz
    c, s = 0.80932, 0.58737
    ex, ez = BLOCKS_RESTIR_EYE[0], BLOCKS_RESTIR_EYE[2]
    n = len(b.tris)
    P = np.zeros((n, 3, 3), dtype=np.float64)
    mat_ids = np.zeros(n, dtype=np.int64)
    for k, (p0, p1, p2, m) in enumerate(b.tris):
        P[k, 0], P[k, 1], P[k, 2] = p0, p1, p2
        mat_ids[k] = m
    W = np.empty_like(P)
    W[..., 0] = c * P[..., 0] + s * P[..., 2] + ex
    W[..., 1] = P[..., 1]
    W[..., 2] = -s * P[..., 0] + c * P[..., 2] + ez
    out = np.zeros(n, dtype=TRIANGLE)
    out["v"] = W.astype(np.float32)
    kd = np.asarray([m[1] for m in BLOCKS_RESTIR_MATERIALS], dtype=np.float32)
    ke = np.asarray([m[2] for m in BLOCKS_RESTIR_MATERIALS], dtype=np.float32)
    out["color"] = kd[mat_ids]
    out["emissive"] = ke[mat_ids]
    return out


# ---- blocks_pt stand-in (assets/blocks_pt.obj is a missing blob; its material file is present) ----
# (name, Kd, Ke) exactly as assets/blocks_pt.mtl lists them: 8 materials, 2 emissive
BLOCKS_PT_MATERIALS = [
    ("Black", (0.018829, 0.018829, 0.018829), (0.0, 0.0, 0.0)),
    ("Brown", (0.100000, 0.046767, 0.021787), (0.0, 0.0, 0.0)),
    ("Emmisive", (0.800000, 0.800000, 0.800000), (120.000015, 116.813454, 99.692398)),
    ("FloorMaterial", (0.538017, 0.523392, 0.489841), (0.0, 0.0, 0.0)),
    ("Green", (0.057167, 0.136217, 0.024592), (0.0, 0.0, 0.0)),
    ("WeakLight", (0.800000, 0.800000, 0.800000), (0.500000, 0.500000, 0.500000)),
    ("Yellow", (0.617207, 0.419342, 0.179603), (0.0, 0.0, 0.0)),
    ("white", (0.546776, 0.508329, 0.527966), (0.0, 0.0, 0.0)),
]
# camera "blocks_pt.obj" of examples/07_pt/07_pt.cpp:139-140 (also 08_nee)
BLOCKS_PT_EYE = (5.983407, 13.970583, -28.553869)
BLOCKS_PT_LOOKAT = (-5.354514, 4.815835, -2.047728)


def make_blocks_pt(detail=1.0):
    """Synthetic stand-in for assets/blocks_pt.obj (the scene of 07_pt / 08_nee and of BASELINE config #3's text):
    the 8 materials of assets/blocks_pt.mtl, one small bright lamp (Emmisive, Ke 120) and a large weak ceiling
    light (WeakLight, Ke 0.5), a studded-brick tree on a brick base over a tiled floor, seen from the camera of
    07_pt.cpp:139-140. Integer LCG (seed 2025), no libm; triangle order = file order = primID = light order."""
    b = _Builder()
    rng = _LCG(2025)
    M = {m[0]: i for i, m in enumerate(BLOCKS_PT_MATERIALS)}
    # floor and a back wall behind the look-at point (the camera looks towards +z and slightly -x, down)
    b.patch(lambda i, j: (-70.0 + 140.0 * i / 48.0, 0.0, -50.0 + 120.0 * j / 48.0), 48, 48, M["FloorMaterial"])
    b.patch(lambda i, j: (-70.0 + 140.0 * i / 24.0, 50.0 * j / 12.0, 70.0), 24, 12, M["white"])
    b.patch(lambda i, j: (-70.0, 50.0 * j / 12.0, -50.0 + 120.0 * i / 24.0), 24, 12, M["white"])
    # weak area light under the ceiling, and the ceiling itself
    b.patch(lambda i, j: (-30.0 + 60.0 * i / 8.0, 44.0, -20.0 + 60.0 * j / 8.0), 8, 8, M["WeakLight"])
    # the lamp: a 2x2 emissive brick on a black post, to the left of the tree
    for k in range(8):
        b.brick(-18, 1.2 * k, 2, 1, 1, M["Black"])
    b.brick(-19, 9.6, 1, 2, 2, M["Emmisive"])
    # brick base under the tree
    for i in range(-4, 3):
        for j in range(-3, 3):
            b.brick(4 * i - 5, 1.2 * ((i + j) % 2), 4 * j - 2, 4, 4, M["Black"] if (i + j) % 3 else M["Brown"])
    # trunk and foliage pads around the look-at point (-5.35, 4.8, -2.05)
    tx, tz = -6, -3
    for k in range(7):
        b.brick(tx, 2.4 + 1.2 * k, tz, 2, 2, M["Brown"])
        tx += rng.below(3) - 1
        tz += rng.below(3) - 1
        tx = max(-9, min(-3, tx))
        tz = max(-6, min(0, tz))
    pads = [(-11, 4.8, -6, 7), (0, 6.0, 2, 7), (-7, 7.2, 5, 6), (-1, 8.4, -7, 6), (-12, 9.6, 0, 6),
            (-5, 10.8, -2, 7), (2, 12.0, 4, 5), (-9, 13.2, -8, 5), (-3, 14.4, 2, 5), (-6, 15.6, -3, 4)]
    keep = int(256 * min(1.0, max(0.0, detail)))
    for (cx, cy, cz, rad) in pads:
        for layer in range(3):
            r = rad - layer
            nx, nz = (4, 2) if layer % 2 == 0 else (2, 4)
            for ix in range(-r, r, nx):
                for iz in range(-r, r, nz):
                    mx, mz = ix + nx * 0.5, iz + nz * 0.5
                    u = rng.below(256)
                    if mx * mx + mz * mz > r * r:
                        continue
                    if u >= keep or u >= 236:
                        continue
                    mat = M["Green"] if rng.below(12) else M["Yellow"]
                    b.brick(cx + ix, cy + 1.2 * layer, cz + iz, nx, nz, mat)
    n = len(b.tris)
    out = np.zeros(n, dtype=TRIANGLE)
    P = np.zeros((n, 3, 3), dtype=np.float64)
    mat_ids = np.zeros(n, dtype=np.int64)
    for k, (p0, p1, p2, m) in enumerate(b.tris):
        P[k, 0], P[k, 1], P[k, 2] = p0, p1, p2
        mat_ids[k] = m
    out["v"] = P.astype(np.float32)
    out["color"] = np.asarray([m[1] for m in BLOCKS_PT_MATERIALS], dtype=np.float32)[mat_ids]
    out["emissive"] = np.asarray([m[2] for m in BLOCKS_PT_MATERIALS], dtype=np.float32)[mat_ids]
    return out


def make_quad_room(n_lights=8, seed=7):
    """Tiny deterministic test scene: a floor, a back wall, a box and a few small emissive
    quads. For unit tests where cornellbox fixtures are too large or too regular."""
    b = _Builder()
    rng = _LCG(seed)
    b.patch(lambda i, j: (-4.0 + 8.0 * i / 4.0, 0.0, -4.0 + 8.0 * j / 4.0), 4, 4, 0)
    b.patch(lambda i, j: (-4.0 + 8.0 * i / 4.0, 0.0 + 6.0 * j / 4.0, -4.0), 4, 4, 1)
    b.box((-1.0, 0.0, -1.5), (0.5, 1.5, 0.0), 2)
    kd = [(0.7, 0.7, 0.7), (0.6, 0.3, 0.2), (0.2, 0.5, 0.7)]
    ke = [(0.0, 0.0, 0.0)] * 3
    for k in range(n_lights):
        x = -3.0 + 6.0 * rng.below(1024) / 1024.0
        y = 2.0 + 3.0 * rng.below(1024) / 1024.0
        z = -3.0 + 4.0 * rng.below(1024) / 1024.0
        b.quad((x, y, z), (x + 0.4, y, z), (x + 0.4, y, z + 0.4), (x, y, z + 0.4), 3 + k)
        kd.append((0.8, 0.8, 0.8))
        ke.append((4.0 + rng.below(16), 4.0 + rng.below(16), 4.0 + rng.below(16)))
    n = len(b.tris)
    out = np.zeros(n, dtype=TRIANGLE)
    for k, (p0, p1, p2, m) in enumerate(b.tris):
        out["v"][k] = np.asarray([p0, p1, p2], dtype=np.float64).astype(np.float32)
        out["color"][k] = kd[m]
        out["emissive"][k] = ke[m]
    return out
