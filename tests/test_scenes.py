"""Scene ingest: OBJ reader semantics (fan triangulation, materials, order) and the synthetic
benchmark scene's determinism."""
import os

import numpy as np


def test_fixture_scene_counts(golden_dir):
    """Triangle / light counts of the reference's assets as its loader orders them."""
    from cedec_2024_rt_amd import scenes

    g = np.load(os.path.join(golden_dir, "scenes.npz"))
    assert len(g["cornellbox1"]) == 36 and len(scenes.light_indices(g["cornellbox1"])) == 2
    assert len(g["cornellbox2"]) == 3470 and len(scenes.light_indices(g["cornellbox2"])) == 2


def test_obj_reader_fan_and_materials(tmp_path):
    from cedec_2024_rt_amd import scenes

    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nKe 0 0 0\nnewmtl b\nKd 1 1 1\nKe 5 6 7\n")
    (tmp_path / "s.obj").write_text(
        "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\nusemtl a\nf 1 2 3 4 5\nusemtl b\nf -5//1 -4//1 -3//1\n")
    t = scenes.load_obj(str(tmp_path / "s.obj"))
    assert len(t) == 4
    # pentagon -> fan (1,2,3) (1,3,4) (1,4,5)
    assert np.array_equal(t["v"][0], np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0]]))
    assert np.array_equal(t["v"][1], np.float32([[0, 0, 0], [1, 1, 0], [0, 1, 0]]))
    assert np.array_equal(t["v"][2], np.float32([[0, 0, 0], [0, 1, 0], [0.5, 2, 0]]))
    assert np.allclose(t["color"][0], [0.1, 0.2, 0.3]) and np.all(t["emissive"][:3] == 0)
    # negative indices are relative to the end of the vertex list
    assert np.array_equal(t["v"][3], np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0]]))
    assert np.allclose(t["emissive"][3], [5, 6, 7])
    assert list(scenes.light_indices(t)) == [3]


def test_blocks_restir_stand_in_is_deterministic():
    from cedec_2024_rt_amd import scenes

    t = scenes.make_blocks_restir()
    assert len(t) == 211916 and len(scenes.light_indices(t)) == 6944
    assert scenes.scene_sha256(t).startswith("a5a236e72415e48e")
    e = t["emissive"][scenes.light_indices(t)]
    assert len(np.unique(e, axis=0)) >= 20 and e.max() == np.float32(120.0)
    # no degenerate triangles (they would put NaNs into normals / pdfs)
    v = t["v"].astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1)
    assert area.min() > 1e-6


def test_blocks_pt_stand_in():
    """The stand-in for the missing assets/blocks_pt.obj (07_pt / 08_nee scene, BASELINE config #3's text): fixed
    triangle count and hash, the 8 materials of assets/blocks_pt.mtl with its 2 emissive ones, no degenerate triangle."""
    from cedec_2024_rt_amd import scenes

    t = scenes.make_blocks_pt()
    assert len(t) == 151816 and len(scenes.light_indices(t)) == 324
    assert scenes.scene_sha256(t).startswith("886d2264a483a8c9")
    e = np.unique(t["emissive"][scenes.light_indices(t)], axis=0)
    assert len(e) == 2 and e.max() == np.float32(120.000015) and e.min() == np.float32(0.5)
    v = t["v"].astype(np.float64)
    assert (0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1)).min() > 1e-6
    mtl = "/root/reference/assets/blocks_pt.mtl"
    if os.path.exists(mtl):  # build container: the table in scenes.py is the file
        m = scenes.load_mtl(mtl)
        assert sorted(m) == sorted(n for n, _, _ in scenes.BLOCKS_PT_MATERIALS)
    assert scenes.BLOCKS_PT_EYE == (5.983407, 13.970583, -28.553869)  # 07_pt.cpp:139


def test_obj_readers_match_tinyobj_fixture(tmp_path):
    """Runs everywhere (GPU box included): both OBJ readers on the committed copies of two scene files the reference
    ships (tests/golden/assets, MIT) against the triangle arrays the reference's vendored tinyobjloader v1.0.6 +
    the loop of common/loader.hpp:25-64 produced for them (tests/golden/ref_tinyobj.npz, made by make_golden.py)."""
    import subprocess

    from cedec_2024_rt_amd import scenes
    from cedec_2024_rt_amd.types import TRIANGLE

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = np.load(os.path.join(root, "tests", "golden", "ref_tinyobj.npz"))
    assets = os.path.join(root, "tests", "golden", "assets")
    app = os.path.join(root, "app", "restir_app")
    for name, count, lights in (("cornellbox1", 36, 2), ("blocks_ao", 3034, 0)):
        ref = gold[name]
        assert len(ref) == count
        mine = scenes.load_obj(os.path.join(assets, name + ".obj"))
        assert mine.tobytes() == ref.tobytes(), name
        assert len(scenes.light_indices(mine)) == lights
        if os.path.exists(app):
            out2 = str(tmp_path / (name + "_app.tris"))
            subprocess.check_call([app, "--obj", os.path.join(assets, name + ".obj"), "--dump-tris", out2], stdout=subprocess.DEVNULL)
            assert np.fromfile(out2, dtype=TRIANGLE).tobytes() == ref.tobytes(), name + " (C++ app)"


def test_obj_readers_match_the_references_tinyobj_loader(tmp_path):
    """Both OBJ readers (scenes.load_obj and app/restir_main.cpp) produce, byte for byte, the
    triangle array that the reference's vendored tinyobjloader v1.0.6 + the loop of
    common/loader.hpp:25-64 produce (oracle/_ref/ref_tinyobj) on every OBJ the reference ships."""
    import subprocess

    import pytest

    from cedec_2024_rt_amd import scenes
    from cedec_2024_rt_amd.types import TRIANGLE

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_bin = os.path.join(root, "oracle", "_ref", "ref_tinyobj")
    assets = "/root/reference/assets"
    if not (os.path.exists(ref_bin) and os.path.isdir(assets)):
        pytest.skip("needs the reference checkout and oracle/_ref/ref_tinyobj (build container only)")
    app = os.path.join(root, "app", "restir_app")
    for name, count in (("cornellbox1", 36), ("cornellbox2", 3470), ("blocks_ao", 3034)):
        out = str(tmp_path / (name + ".tris"))
        subprocess.check_call([ref_bin, f"{assets}/{name}.obj", assets + "/", out], stdout=subprocess.DEVNULL)
        ref = np.fromfile(out, dtype=TRIANGLE)
        assert len(ref) == count
        assert scenes.load_obj(f"{assets}/{name}.obj").tobytes() == ref.tobytes(), name
        if os.path.exists(app):
            out2 = str(tmp_path / (name + "_app.tris"))
            subprocess.check_call([app, "--obj", f"{assets}/{name}.obj", "--dump-tris", out2], stdout=subprocess.DEVNULL)
            assert np.fromfile(out2, dtype=TRIANGLE).tobytes() == ref.tobytes(), name + " (C++ app)"
