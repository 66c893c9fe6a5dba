"""Build-container script: the reference's only triangle asset, data/WaterBottle (glTF 2.0 sample; its LICENSE.txt: CC0), as a data fixture
for the whitted path's texture / UV / normal-map code (tests/test_gpu_parity.py::test_whitted_waterbottle, tests/test_oracle_whitted.py).

    python tests/golden/waterbottle/make_fixture.py        ->  waterbottle.npz, waterbottle.json

What sutil::Scene's loader (sutil/Scene.cpp:256-1314, through support/tinygltf) would hand to whitted.cu, restated with json + numpy +
PIL and nothing of tinygltf: the one mesh primitive's POSITION / NORMAL / TEXCOORD_0 accessors and its 16-bit indices (widened to 32
bits: the ABI takes uint32), with the node's rotation (quaternion (0, 1, 0, 0): half a turn about y) applied to positions and normals --
the ABI takes meshes in world space, instance transforms are the caller's; the material's three images (base colour, occlusion-
roughness-metallic, normal) as RGBA8, row 0 first.  The images are 2048 x 2048 in the asset (16 MB each as texels): the fixture holds
them box-filtered down to 256 x 256, which is what keeps it under a megabyte -- the code under test does not care about the size, and
tests/test_gpu_parity.py also runs a procedural texture at odd sizes.  Factors: glTF defaults (base colour (1,1,1,1), metallic 1,
roughness 1: the material names none).  The emissive image is not stored: whitted.cu never reads an emissive term.
"""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/data/WaterBottle"
TEX = 256


def main():
    from PIL import Image
    g = json.load(open(os.path.join(SRC, "WaterBottle.gltf")))
    raw = open(os.path.join(SRC, g["buffers"][0]["uri"]), "rb").read()
    comp = {5126: np.float32, 5123: np.uint16, 5125: np.uint32}
    width = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}

    def accessor(i):
        a = g["accessors"][i]
        v = g["bufferViews"][a["bufferView"]]
        off = v.get("byteOffset", 0) + a.get("byteOffset", 0)
        n = a["count"] * width[a["type"]]
        assert "byteStride" not in v
        return np.frombuffer(raw, dtype=comp[a["componentType"]], count=n, offset=off).reshape(a["count"], width[a["type"]]).copy()

    prim = g["meshes"][0]["primitives"][0]
    pos, nrm, uv = accessor(prim["attributes"]["POSITION"]), accessor(prim["attributes"]["NORMAL"]), accessor(prim["attributes"]["TEXCOORD_0"])
    idx = accessor(prim["indices"]).astype(np.uint32).reshape(-1, 3)
    assert g["nodes"][0]["rotation"] == [0.0, 1.0, 0.0, 0.0]    # half a turn about y: (x, y, z) -> (-x, y, -z), exactly
    turn = np.array([-1.0, 1.0, -1.0], dtype=np.float32)
    pos, nrm = (pos * turn).astype(np.float32), (nrm * turn).astype(np.float32)
    mat = g["materials"][prim["material"]]
    pbr = mat["pbrMetallicRoughness"]
    images = {"base_color_tex": g["textures"][pbr["baseColorTexture"]["index"]]["source"],
              "metallic_roughness_tex": g["textures"][pbr["metallicRoughnessTexture"]["index"]]["source"],
              "normal_tex": g["textures"][mat["normalTexture"]["index"]]["source"]}
    out = {"positions": pos, "normals": nrm, "texcoords": uv.astype(np.float32), "indices": idx}
    meta = {"source": "data/WaterBottle/WaterBottle.gltf", "vertices": int(len(pos)), "triangles": int(len(idx)),
            "node_rotation_applied": [0.0, 1.0, 0.0, 0.0], "texture_size_in_fixture": TEX, "files": {},
            "factors": {"base_color": pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0]), "metallic": pbr.get("metallicFactor", 1.0), "roughness": pbr.get("roughnessFactor", 1.0)}}
    for name in ("WaterBottle.gltf", g["buffers"][0]["uri"]):
        meta["files"][name] = hashlib.sha256(open(os.path.join(SRC, name), "rb").read()).hexdigest()
    for key, src in images.items():
        uri = g["images"][src]["uri"]
        meta["files"][uri] = hashlib.sha256(open(os.path.join(SRC, uri), "rb").read()).hexdigest()
        im = Image.open(os.path.join(SRC, uri)).convert("RGBA")
        meta.setdefault("texture_size_in_asset", list(im.size))
        out[key] = np.asarray(im.resize((TEX, TEX), Image.BOX), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "waterbottle.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "waterbottle.json"), "w"), indent=1, sort_keys=True)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "waterbottle.npz")))


if __name__ == "__main__":
    main()
