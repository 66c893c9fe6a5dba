"""A procedural triangle scene for the whitted path (the reference's own input for it is a glTF file whose loader,
sutil/Scene.cpp + tinygltf, is out of scope): a UV sphere with smooth vertex normals, a faceted box without normals,
a ground quad, three materials (dielectric, rough metal, smooth metal), two point lights."""
import numpy as np


def build(n_lat=10, n_lon=16, seed=0):
    pos, nrm, idx, tmat = [], [], [], []

    def add_mesh(p, n, tris, mat):
        base = len(pos)
        pos.extend(p)
        nrm.extend(n)
        idx.extend([[a + base, b + base, c + base] for a, b, c in tris])
        tmat.extend([mat] * len(tris))

    # sphere, radius 1.2 at (-1.2, 1.2, 0): smooth normals
    c, r = np.array([-1.2, 1.2, 0.0]), 1.2
    p, n, tris = [], [], []
    for i in range(n_lat + 1):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * np.pi * j / n_lon
            d = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            p.append(c + r * d)
            n.append(d)
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
            cc, dd = a + n_lon, b + n_lon
            if i > 0:
                tris.append((a, b, cc))
            if i < n_lat - 1:
                tris.append((b, dd, cc))
    add_mesh(p, n, tris, 1)
    # box, rotated: flat shading comes from per-face vertices whose normals equal the face normal
    rng = np.random.RandomState(seed)
    ang = 0.6
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    h = np.array([0.8, 1.0, 0.6])
    centre = np.array([1.6, 1.0, 0.4])
    for axis in range(3):
        for sgn in (-1, 1):
            u, v = [a for a in range(3) if a != axis]
            nn = np.zeros(3); nn[axis] = sgn
            quad = []
            for su, sv in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                q = np.zeros(3); q[axis] = sgn * h[axis]; q[u] = su * h[u]; q[v] = sv * h[v]
                quad.append(centre + R @ q)
            order = [(0, 1, 2), (0, 2, 3)] if sgn * (1 if axis != 1 else -1) > 0 else [(0, 2, 1), (0, 3, 2)]
            add_mesh(quad, [R @ nn] * 4, order, 2)
    # ground
    g = 6.0
    add_mesh([[-g, 0, -g], [g, 0, -g], [g, 0, g], [-g, 0, g]], [[0, 1, 0]] * 4, [(0, 2, 1), (0, 3, 2)], 0)
    materials = np.array([[0.8, 0.8, 0.75, 1.0, 0.0, 0.9],      # ground: dielectric, rough
                          [0.9, 0.25, 0.2, 1.0, 0.1, 0.35],     # sphere: mostly dielectric, glossy
                          [0.95, 0.8, 0.3, 1.0, 1.0, 0.25]],    # box: metal
                         dtype=np.float32)
    lights = np.zeros((2, 8), dtype=np.float32)
    lights[0] = [1.0, 0.95, 0.9, 2.5, 4.0, 6.0, 3.0, 0]
    lights[1] = [0.6, 0.7, 1.0, 1.2, -5.0, 4.0, -2.0, 0]
    lights[:, 7] = np.array([2, 2], dtype=np.int32).view(np.float32)   # Light::Falloff::QUADRATIC (never read)
    return {"positions": np.array(pos, dtype=np.float32), "normals": np.array(nrm, dtype=np.float32),
            "indices": np.array(idx, dtype=np.uint32), "tri_material": np.array(tmat, dtype=np.uint32),
            "materials": materials, "lights": lights, "miss": np.array([0.1, 0.15, 0.25], dtype=np.float32)}


def camera(oracle, width, height, eye=(0.5, 3.0, 7.0), lookat=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0), fov=45.0):
    U, V, W = [np.zeros(3, dtype=np.float32) for _ in range(3)]
    e, l, u = oracle.f32(eye), oracle.f32(lookat), oracle.f32(up)
    oracle.lib().oracle_camera_uvw(oracle.fptr(e), oracle.fptr(l), oracle.fptr(u), fov, np.float32(np.float32(width) / np.float32(height)),
                                   oracle.fptr(U), oracle.fptr(V), oracle.fptr(W))
    return np.concatenate([e, U, V, W]).astype(np.float32)


def waterbottle():
    """the reference's data/WaterBottle as tests/golden/waterbottle/make_fixture.py stored it (mesh in world space, three 256 x 256
    RGBA8 images), as a mesh dict for oracle.whitted_render / the C ABI: one material with the glTF default factors and all three textures"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "waterbottle", "waterbottle.npz"))
    lights = np.zeros((2, 8), dtype=np.float32)
    lights[0] = [1.0, 0.97, 0.9, 1.2, 0.25, 0.3, 0.35, 0]     # (intensity scaled to a 26 cm bottle: whitted.cu has no distance falloff)
    lights[1] = [0.7, 0.8, 1.0, 0.7, -0.3, 0.1, 0.15, 0]
    return {"positions": z["positions"], "normals": z["normals"], "texcoords": z["texcoords"], "indices": z["indices"], "tri_material": None,
            "materials": np.array([[1.0, 1.0, 1.0, 1.0, 1.0, 1.0]], dtype=np.float32),
            "textures": {0: (z["base_color_tex"], z["metallic_roughness_tex"], z["normal_tex"])},
            "lights": lights, "miss": np.array([0.05, 0.06, 0.08], dtype=np.float32)}


def textured_quad(tex_w=7, tex_h=5, seed=3, with_texcoords=True):
    """two quads (four triangles) under procedural textures of odd size, texture coordinates that leave [0, 1] on both sides (wrap
    addressing), one material with only a base-colour texture, one with only a normal map, one with none"""
    rng = np.random.RandomState(seed)
    pos = np.array([[-2, 0, -1], [0, 0, -1], [0, 2, -1], [-2, 2, -1], [0.2, 0, -1.2], [2.2, 0, -0.6], [2.2, 2, -0.6], [0.2, 2, -1.2],
                    [-3, -0.01, -3], [3, -0.01, -3], [3, -0.01, 3], [-3, -0.01, 3]], dtype=np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 10, 9], [8, 11, 10]], dtype=np.uint32)
    uv = np.array([[-0.7, -0.3], [1.6, -0.3], [1.6, 2.4], [-0.7, 2.4], [0.1, 0.2], [0.9, 0.2], [0.9, 0.8], [0.1, 0.8],
                   [0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    t0 = rng.randint(0, 256, (tex_h, tex_w, 4)).astype(np.uint8)
    t1 = rng.randint(0, 256, (tex_w, tex_h, 4)).astype(np.uint8)
    nm = np.zeros((4, 4, 4), np.uint8)
    nm[..., 0:2] = rng.randint(96, 160, (4, 4, 2))
    nm[..., 2] = 230
    nm[..., 3] = 255
    lights = np.zeros((1, 8), dtype=np.float32)
    lights[0] = [1.0, 1.0, 1.0, 3.0, 0.5, 2.0, 3.0, 0]
    return {"positions": pos, "normals": None, "texcoords": uv if with_texcoords else None, "indices": idx,
            "tri_material": np.array([0, 0, 1, 1, 2, 2], dtype=np.uint32),
            "materials": np.array([[1.0, 0.9, 0.8, 1.0, 0.6, 0.8], [0.7, 0.7, 0.9, 1.0, 0.2, 0.5], [0.5, 0.5, 0.5, 1.0, 0.0, 0.9]], dtype=np.float32),
            "textures": {0: (t0, t1, None), 1: (None, None, nm)},
            "lights": lights, "miss": np.array([0.1, 0.1, 0.12], dtype=np.float32)}
