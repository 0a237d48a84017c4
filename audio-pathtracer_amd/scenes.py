"""Procedural acoustic scenes for the BASELINE.json configs.

Every asset under the reference's Content/ is a git-LFS pointer, so the three scenes
the configs name are synthesised here with matched triangle counts, fixed seeds, units = cm
(Unreal units), float32:

  shoebox()       12 triangles      cfg1  "Shoebox room (12 tris)"
  starter_room()  5 000 triangles   cfg2  "StarterContent room (~5k tris)"
  old_mine()      100 000 triangles cfg3-5 "Scene_OldMine (~100k tris)"

A scene is what RegisterGeometry + UAcousticMaterial hand the subsystem (reference
Public/AudioRayTracingSubsystem.h:99-100, Public/AcousticMaterial.h:22-33): a triangle soup
[T][3][3], one material id per triangle, and an absorption table [M][B] (the BDPT path reads
Absorption only, AudioRayTracingSubsystem.cpp:385).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

NO_MATERIAL = 0xFFFF


@dataclass
class Scene:
    name: str
    triangles: np.ndarray  # [T,3,3] float32, cm
    material_ids: np.ndarray  # [T] uint16
    absorption: np.ndarray  # [M,B] float32
    source: np.ndarray  # [3]
    listener: np.ndarray  # [3]
    material_names: list = field(default_factory=list)
    extra_sources: np.ndarray | None = None  # [S,3] for the multi-source config
    object_ids: np.ndarray | None = None     # [T] uint32 actor per triangle (legacy tracer: ignored / passed-through actors)

    @property
    def num_triangles(self) -> int:
        return int(self.triangles.shape[0])

    @property
    def num_bands(self) -> int:
        return int(self.absorption.shape[1])


# ------------------------------------------------------------------------------------------------
# mesh helpers
# ------------------------------------------------------------------------------------------------
class _Mesh:
    def __init__(self):
        self.tris = []
        self.mats = []
        self.objs = []
        self.next_obj = 0

    def add(self, tris, mat, obj=None):
        """one call = one actor (a mesh with its own collision), unless obj names an existing actor"""
        tris = np.asarray(tris, dtype=np.float64).reshape(-1, 3, 3)
        if obj is None:
            obj = self.next_obj
            self.next_obj += 1
        self.tris.append(tris)
        self.mats.append(np.full(tris.shape[0], mat, dtype=np.uint16))
        self.objs.append(np.full(tris.shape[0], obj, dtype=np.uint32))

    def count(self):
        return int(sum(t.shape[0] for t in self.tris))

    def finish(self):
        t = np.concatenate(self.tris, axis=0).astype(np.float32)
        m = np.concatenate(self.mats, axis=0)
        self.object_ids = np.ascontiguousarray(np.concatenate(self.objs, axis=0))
        return np.ascontiguousarray(t), np.ascontiguousarray(m)


def _grid_quad(origin, u, v, nu, nv, keep=None):
    """(nu x nv) quads spanning origin + a*u + b*v, a,b in [0,1]; keep(i,j,ca,cb) filters cells."""
    origin, u, v = (np.asarray(x, dtype=np.float64) for x in (origin, u, v))
    out = []
    for i in range(nu):
        for j in range(nv):
            a0, a1 = i / nu, (i + 1) / nu
            b0, b1 = j / nv, (j + 1) / nv
            if keep is not None and not keep(i, j, 0.5 * (a0 + a1), 0.5 * (b0 + b1)):
                continue
            p00 = origin + a0 * u + b0 * v
            p10 = origin + a1 * u + b0 * v
            p11 = origin + a1 * u + b1 * v
            p01 = origin + a0 * u + b1 * v
            out.append([p00, p10, p11])
            out.append([p00, p11, p01])
    return np.asarray(out, dtype=np.float64).reshape(-1, 3, 3)


def _box(lo, hi, n=1):
    """closed axis-aligned box, each face n x n quads -> 12 n^2 triangles"""
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    faces = [
        (lo, ex, ey), (lo + ez, ex, ey),
        (lo, ex, ez), (lo + ey, ex, ez),
        (lo, ey, ez), (lo + ex, ey, ez),
    ]
    return np.concatenate([_grid_quad(o, u, v, n, n) for o, u, v in faces], axis=0)


def _uv_sphere(center, radius, nu, nv, squash=(1.0, 1.0, 1.0)):
    """closed UV sphere: 2*nu*(nv-1) triangles"""
    c = np.asarray(center, dtype=np.float64)
    out = []

    def pt(i, j):
        th = 2.0 * np.pi * (i % nu) / nu
        ph = np.pi * j / nv
        return c + radius * np.array([squash[0] * np.sin(ph) * np.cos(th), squash[1] * np.sin(ph) * np.sin(th),
                                      squash[2] * np.cos(ph)])

    for i in range(nu):
        for j in range(nv):
            p00, p10, p01, p11 = pt(i, j), pt(i + 1, j), pt(i, j + 1), pt(i + 1, j + 1)
            if j > 0:
                out.append([p00, p10, p11])
            if j < nv - 1:
                out.append([p00, p11, p01])
    return np.asarray(out, dtype=np.float64)


def _cylinder(base, radius, height, nseg):
    """closed z-axis cylinder: 4*nseg triangles"""
    b = np.asarray(base, dtype=np.float64)
    out = []
    top = b + np.array([0, 0, height])
    for i in range(nseg):
        a0, a1 = 2 * np.pi * i / nseg, 2 * np.pi * (i + 1) / nseg
        r0 = radius * np.array([np.cos(a0), np.sin(a0), 0.0])
        r1 = radius * np.array([np.cos(a1), np.sin(a1), 0.0])
        out.append([b + r0, b + r1, top + r1])
        out.append([b + r0, top + r1, top + r0])
        out.append([b, b + r1, b + r0])
        out.append([top, top + r0, top + r1])
    return np.asarray(out, dtype=np.float64)


def _tetra(center, size):
    c = np.asarray(center, dtype=np.float64)
    v = c + size * np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=np.float64)
    return np.asarray([[v[0], v[1], v[2]], [v[0], v[3], v[1]], [v[0], v[2], v[3]], [v[1], v[3], v[2]]])


def _materials(rng, M, B):
    """band coefficients uniform in [0.05, 0.9] from the scene seed"""
    return rng.uniform(0.05, 0.9, size=(M, B)).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# cfg1: shoebox
# ------------------------------------------------------------------------------------------------
def shoebox(bands: int = 1, reflectivity: float = 0.5) -> Scene:
    """Shoebox 1000 x 800 x 300 cm, 6 quads = 12 triangles, one material rho = 0.5 in every band."""
    m = _Mesh()
    m.add(_box([0, 0, 0], [1000, 800, 300], 1), 0)
    t, ids = m.finish()
    assert t.shape[0] == 12
    return Scene("shoebox", t, ids, np.full((1, bands), reflectivity, dtype=np.float32),
                 np.array([250, 200, 150], dtype=np.float32), np.array([750, 600, 120], dtype=np.float32),
                 ["plaster"], None, m.object_ids)


# ------------------------------------------------------------------------------------------------
# cfg2: "StarterContent room"
# ------------------------------------------------------------------------------------------------
def starter_room(bands: int = 4, seed: int = 0x57A7, target_tris: int = 5000) -> Scene:
    """2000 x 1600 x 400 cm shell with a door and a half-glazed window (open parts let rays escape,
    exercising the miss branch of GeneratePath), pillars, crates, a table and round props;
    4 materials named after Content/StarterContent/AudioMaterials (carpet, concrete, glass, wood)."""
    rng = np.random.default_rng(seed)
    CARPET, CONCRETE, GLASS, WOOD = 0, 1, 2, 3
    W, D, H = 2000.0, 1600.0, 400.0
    m = _Mesh()
    m.add(_grid_quad([0, 0, 0], [W, 0, 0], [0, D, 0], 20, 16), CARPET)      # floor   640
    m.add(_grid_quad([0, 0, H], [W, 0, 0], [0, D, 0], 20, 16), CONCRETE)    # ceiling 640

    def door(i, j, a, b):   # opening in the y=0 wall: x in [0.40,0.50] W, z below 0.55 H
        return not (0.40 < a < 0.50 and b < 0.55)

    def window(i, j, a, b):  # opening in the x=W wall: y in [0.3,0.7] D, z in [0.3,0.7] H
        return not (0.30 < a < 0.70 and 0.30 < b < 0.70)

    m.add(_grid_quad([0, 0, 0], [W, 0, 0], [0, 0, H], 20, 8, door), CONCRETE)     # y=0 wall
    m.add(_grid_quad([0, D, 0], [W, 0, 0], [0, 0, H], 20, 8), CONCRETE)           # y=D wall
    m.add(_grid_quad([0, 0, 0], [0, D, 0], [0, 0, H], 16, 8), CONCRETE)           # x=0 wall
    m.add(_grid_quad([W, 0, 0], [0, D, 0], [0, 0, H], 16, 8, window), CONCRETE)   # x=W wall
    # glass pane over the lower half of the window; the upper half stays open
    m.add(_grid_quad([W, 0.3 * D, 0.3 * H], [0, 0.4 * D, 0], [0, 0, 0.2 * H], 4, 2), GLASS)
    # pillars
    for px, py in ((500, 400), (1500, 400), (500, 1200), (1500, 1200)):
        m.add(_box([px - 30, py - 30, 0], [px + 30, py + 30, H], 3), CONCRETE)
    # table (wood): top + 4 legs
    m.add(_box([900, 700, 70], [1100, 900, 78], 4), WOOD)
    for lx, ly in ((905, 705), (1089, 705), (905, 889), (1089, 889)):
        m.add(_box([lx, ly, 0], [lx + 6, ly + 6, 70], 1), WOOD)
    # crates
    for _ in range(10):
        s = rng.uniform(40, 90)
        x, y = rng.uniform(100, W - 200), rng.uniform(100, D - 200)
        m.add(_box([x, y, 0], [x + s, y + s, s], 2), WOOD)
    # round props (tessellated)
    for _ in range(6):
        r = rng.uniform(25, 60)
        x, y = rng.uniform(150, W - 150), rng.uniform(150, D - 150)
        m.add(_uv_sphere([x, y, r], r, 12, 8), GLASS if rng.random() < 0.3 else WOOD)
    # filler: small debris tetrahedra on the floor until the exact count is met
    left = target_tris - m.count()
    assert left >= 0 and left % 4 == 0, (m.count(), left)
    for _ in range(left // 4):
        m.add(_tetra([rng.uniform(50, W - 50), rng.uniform(50, D - 50), 6.0], 4.0), CONCRETE)
    t, ids = m.finish()
    assert t.shape[0] == target_tris
    return Scene("starter_room", t, ids, _materials(rng, 4, bands),
                 np.array([300, 300, 150], dtype=np.float32), np.array([1100, 1250, 160], dtype=np.float32),
                 ["carpet", "concrete", "glass", "wood"], None, m.object_ids)


# ------------------------------------------------------------------------------------------------
# cfg3-5: "Scene_OldMine"
# ------------------------------------------------------------------------------------------------
def _hash_u32(a):
    a = np.asarray(a, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    a = (a ^ (a >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    a = (a ^ (a >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    return a ^ (a >> np.uint64(16))


def _lattice_noise(i, j, k, seed):
    """deterministic value in [-1,1) per integer lattice point"""
    h = _hash_u32(np.uint64(seed) + _hash_u32(i) * np.uint64(3) + _hash_u32(j + 0x9E37) * np.uint64(5)
                  + _hash_u32(k + 0x85EB) * np.uint64(7))
    return (h.astype(np.float64) / 2147483648.0) - 1.0


def _fbm2(x, y, seed, octaves=4, base=400.0):
    """value-noise fBm on a plane, [-1,1]"""
    out = np.zeros_like(x, dtype=np.float64)
    amp, freq, norm = 1.0, 1.0 / base, 0.0
    for o in range(octaves):
        xs, ys = x * freq, y * freq
        x0, y0 = np.floor(xs), np.floor(ys)
        fx, fy = xs - x0, ys - y0
        fx, fy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
        xi, yi = x0.astype(np.int64) + 4096, y0.astype(np.int64) + 4096
        z = np.zeros_like(xi) + o
        v00 = _lattice_noise(xi, yi, z, seed)
        v10 = _lattice_noise(xi + 1, yi, z, seed)
        v01 = _lattice_noise(xi, yi + 1, z, seed)
        v11 = _lattice_noise(xi + 1, yi + 1, z, seed)
        out += amp * ((v00 * (1 - fx) + v10 * fx) * (1 - fy) + (v01 * (1 - fx) + v11 * fx) * fy)
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return out / norm


def _dist_to_segments(px, py, segs):
    d = np.full(px.shape, np.inf)
    for (ax, ay), (bx, by) in segs:
        vx, vy = bx - ax, by - ay
        t = np.clip(((px - ax) * vx + (py - ay) * vy) / (vx * vx + vy * vy), 0.0, 1.0)
        d = np.minimum(d, np.hypot(px - (ax + t * vx), py - (ay + t * vy)))
    return d


def old_mine(bands: int = 8, seed: int = 0x01D, target_tris: int = 100000, cell: float = 22.0,
             levels: int = 14) -> Scene:
    """Seeded fBm-displaced tunnel network: three branches (~130 m) meeting at one junction, built
    on a displaced lattice so the shell is watertight (every lattice vertex has one position shared
    by floor, ceiling and wall triangles); mine props (beams, crates, barrels, boulders) and rubble
    bring the count to exactly `target_tris`.  8 materials.  Source and listener sit in different
    branches with no direct line of sight."""
    assert levels % 2 == 0
    rng = np.random.default_rng(seed)
    DIRT, ROCK_A, ROCK_B, ROCK_C, CEIL, WOOD, METAL, GRAVEL = range(8)
    junction = (5000.0, 3000.0)
    ends = [(500.0, 3000.0), (8500.0, 6000.0), (8000.0, 500.0)]
    segs = [(e, junction) for e in ends]
    NX, NY = int(9200 // cell) + 2, int(6800 // cell) + 2
    ci, cj = np.meshgrid(np.arange(NX), np.arange(NY), indexing="ij")
    cx, cy = (ci + 0.5) * cell, (cj + 0.5) * cell
    halfw = 200.0 + 60.0 * _fbm2(cx, cy, seed + 11, octaves=3, base=900.0)
    open_ = _dist_to_segments(cx, cy, segs) < halfw
    open_[0, :] = open_[-1, :] = False
    open_[:, 0] = open_[:, -1] = False

    # lattice vertex positions P[i, j, k]
    vi, vj = np.meshgrid(np.arange(NX + 1), np.arange(NY + 1), indexing="ij")
    vx0, vy0 = vi * cell, vj * cell
    zf = 25.0 * _fbm2(vx0, vy0, seed + 1, octaves=4, base=500.0)
    zc = 300.0 + 45.0 * _fbm2(vx0, vy0, seed + 2, octaves=4, base=350.0)
    K = levels
    P = np.zeros((NX + 1, NY + 1, K + 1, 3), dtype=np.float64)
    for k in range(K + 1):
        kk = np.zeros_like(vi) + k
        amp = 0.28 * cell
        P[:, :, k, 0] = vx0 + amp * _lattice_noise(vi, vj, kk, seed + 21)
        P[:, :, k, 1] = vy0 + amp * _lattice_noise(vi, vj, kk, seed + 22)
        P[:, :, k, 2] = zf + (zc - zf) * (k / K)

    tris, mats = [], []
    oi, oj = np.nonzero(open_)
    # floor + ceiling
    for k, mat in ((0, DIRT), (K, CEIL)):
        p00, p10 = P[oi, oj, k], P[oi + 1, oj, k]
        p11, p01 = P[oi + 1, oj + 1, k], P[oi, oj + 1, k]
        tris.append(np.stack([p00, p10, p11], axis=1))
        tris.append(np.stack([p00, p11, p01], axis=1))
        mats.append(np.full(2 * oi.shape[0], mat, dtype=np.uint16))
    # walls: every edge between an open cell and a solid neighbour, K quads each
    rock = (ROCK_A, ROCK_B, ROCK_C)
    for di, dj, (a, b) in ((1, 0, ((1, 0), (1, 1))), (-1, 0, ((0, 0), (0, 1))),
                           (0, 1, ((0, 1), (1, 1))), (0, -1, ((0, 0), (1, 0)))):
        nb_solid = ~open_[np.clip(oi + di, 0, NX - 1), np.clip(oj + dj, 0, NY - 1)]
        wi, wj = oi[nb_solid], oj[nb_solid]
        band = (_fbm2(wi * cell, wj * cell, seed + 31, octaves=2, base=700.0) * 1.5 + 1.5).astype(np.int64) % 3
        wall_mat = np.asarray(rock, dtype=np.uint16)[band]
        for k in range(K):
            q00, q10 = P[wi + a[0], wj + a[1], k], P[wi + b[0], wj + b[1], k]
            q11, q01 = P[wi + b[0], wj + b[1], k + 1], P[wi + a[0], wj + a[1], k + 1]
            tris.append(np.stack([q00, q10, q11], axis=1))
            tris.append(np.stack([q00, q11, q01], axis=1))
            mats.append(np.concatenate([wall_mat, wall_mat]))
    m = _Mesh()
    m.tris = [np.concatenate(tris, axis=0)]
    m.mats = [np.concatenate(mats, axis=0)]
    m.objs = [np.zeros(m.tris[0].shape[0], dtype=np.uint32)]   # the tunnel shell is one landscape actor
    m.next_obj = 1
    lattice = m.count()
    assert lattice % 4 == 0 and lattice < target_tris - 2000, lattice

    # props along the tunnels
    def along(seg_idx, s, lateral):
        (ax, ay), (bx, by) = segs[seg_idx]
        vx, vy = bx - ax, by - ay
        ln = np.hypot(vx, vy)
        return ax + s * vx - lateral * vy / ln, ay + s * vy + lateral * vx / ln

    for si in range(3):                                  # timber sets every ~6 m: two posts + a cap
        (ax, ay), (bx, by) = segs[si]
        ln = np.hypot(bx - ax, by - ay)
        for q in range(1, int(ln // 600)):
            s = q * 600.0 / ln
            for lat in (-120.0, 120.0):
                x, y = along(si, s, lat)
                m.add(_box([x - 10, y - 10, -20], [x + 10, y + 10, 250], 2), WOOD)
            x, y = along(si, s, 0.0)
            m.add(_box([x - 130, y - 12, 238], [x + 130, y + 12, 262], 2), WOOD)
    for _ in range(40):                                  # crates, barrels, boulders
        si = int(rng.integers(0, 3))
        x, y = along(si, rng.uniform(0.08, 0.92), rng.uniform(-110, 110))
        kind = rng.integers(0, 3)
        if kind == 0:
            s = rng.uniform(40, 80)
            m.add(_box([x, y, 0], [x + s, y + s, s], 2), WOOD)
        elif kind == 1:
            m.add(_cylinder([x, y, 0], rng.uniform(25, 35), rng.uniform(70, 95), 16), METAL)
        else:
            r = rng.uniform(30, 70)
            m.add(_uv_sphere([x, y, 0.4 * r], r, 14, 10, squash=(1.0, rng.uniform(0.7, 1.3), 0.7)), ROCK_B)
    left = target_tris - m.count()
    assert left >= 0 and left % 4 == 0, (m.count(), left)
    for _ in range(left // 4):                           # rubble
        si = int(rng.integers(0, 3))
        x, y = along(si, rng.uniform(0.03, 0.97), rng.uniform(-150, 150))
        m.add(_tetra([x, y, 8.0], rng.uniform(3.0, 7.0)), GRAVEL)
    t, ids = m.finish()
    assert t.shape[0] == target_tris
    sx, sy = along(0, 0.80, 20.0)      # 9 m before the junction in branch 0
    lx, ly = along(1, 0.88, -30.0)     # 5.5 m past the junction in branch 1: no direct line of sight
    # cfg5: 8 sources on a ~15 m spacing along the three branches
    spots = [(0, 0.95), (0, 0.62), (0, 0.29), (1, 0.90), (1, 0.57), (1, 0.25), (2, 0.85), (2, 0.47)]
    extra = np.array([[*along(b, s_, 0.0), 150.0] for b, s_ in spots], dtype=np.float32)
    return Scene("old_mine", t, ids, _materials(rng, 8, bands),
                 np.array([sx, sy, 150.0], dtype=np.float32), np.array([lx, ly, 140.0], dtype=np.float32),
                 ["dirt", "rock_a", "rock_b", "rock_c", "ceiling", "wood", "metal", "gravel"], extra, m.object_ids)


def by_name(name: str, bands: int | None = None) -> Scene:
    if name == "shoebox":
        return shoebox(bands or 1)
    if name == "starter_room":
        return starter_room(bands or 4)
    if name == "old_mine":
        return old_mine(bands or 8)
    raise KeyError(name)


def material_lobes(scene: Scene, seed: int = 0x10BE, max_transmission: float = 0.3):
    """Synthetic UAcousticMaterial::Transmission / Scattering arrays (MAT.h:26-30) for a scene's materials, [M][B]
    float32 each: transmission uniform in [0, max_transmission], scattering uniform in [0.1, 0.9] (seeded).  The
    generated scenes carry absorption only; these feed FS_FLAG_MATERIAL_LOBES in tests and bench."""
    rng = np.random.default_rng(seed)
    shape = np.asarray(scene.absorption).shape
    transmission = rng.uniform(0.0, max_transmission, shape).astype(np.float32)
    scattering = rng.uniform(0.1, 0.9, shape).astype(np.float32)
    return transmission, scattering
