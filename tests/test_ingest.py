"""Scene text ingestion (pbrt trianglemesh blocks, .obj) and the film writer: the reference's own scene
files where they ship triangle meshes as text, and its ML test scenes."""
import os

import numpy as np
import pytest

from nn_bvh_amd import scene
from nn_bvh_amd.film import pixel_rgb, write_pfm

REF = "/root/reference"


def test_pbrt_trianglemesh_text(tmp_path):
    text = '''
# a comment with "integer indices" [ 9 9 9 ] inside must be ignored
Shape "trianglemesh"
  "integer indices" [ 0 1 2
      2 1 3 ]   # trailing comment
  "point3 P" [ 0 0 0   1 0 0
               0 1 0   1 1 0.5 ]
AttributeBegin
Shape "trianglemesh" "point P" [ -1 -1 -1  1 -1 -1  0 1 -1 ] "integer indices" [0 2 1]
AttributeEnd
'''
    p = tmp_path / "t.pbrt"
    p.write_text(text)
    v, t, m = scene.read_pbrt_trianglemeshes(str(p))
    assert v.shape == (7, 3) and t.tolist() == [[0, 1, 2], [2, 1, 3], [4, 6, 5]] and m.tolist() == [0, 0, 1]
    assert v[3].tolist() == [1, 1, 0.5] and v[4].tolist() == [-1, -1, -1]
    (tmp_path / "bad.pbrt").write_text('"integer indices" [0 1 5] "point3 P" [0 0 0 1 0 0 0 1 0]')
    with pytest.raises(ValueError, match="out of range"):
        scene.read_pbrt_trianglemeshes(str(tmp_path / "bad.pbrt"))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference scenes only exist in the build container")
def test_reference_scene_files_parse():
    # killeroo-simple.pbrt carries its ground quads as trianglemesh text (killeroo-simple.pbrt:32-46)
    v, t, m = scene.read_pbrt_trianglemeshes(os.path.join(REF, "scenes", "killeroos", "killeroo-simple.pbrt"))
    assert len(t) >= 4 and t.max() < len(v) and len(np.unique(m)) >= 2
    # the ML test scenes: triangle counts recorded in SURVEY.md §8c
    for name, n in (("bedroom_LowPoly_test.obj", 8646), ("chaos_test.obj", 26721)):
        path = os.path.join(REF, "machine_learning", "test_scenes", name)
        if os.path.exists(path):
            v, t, m = scene.read_obj(path)
            assert len(t) == n and t.max() < len(v)


def test_obj_reader(tmp_path):
    lines = ["v 0 0 0", "v 1 0 0", "v 0 1 0", "v 0 0 1", "g one", "f 1 2 3", "g two", "f 1/1/1 3/2/2 4/3/3", ""]
    (tmp_path / "a.obj").write_text(chr(10).join(lines))
    v, t, m = scene.read_obj(str(tmp_path / "a.obj"))
    assert v.shape == (4, 3) and t.tolist() == [[0, 1, 2], [0, 2, 3]] and m.tolist() == [1, 2]


def test_film_pixel_rgb_and_pfm(tmp_path):
    pix = np.array([[2.0, 4.0, 6.0, 2.0], [1.0, 1.0, 1.0, 0.0], [0.3, 0.6, 0.9, 3.0], [0, 0, 0, 0]])
    rgb = pixel_rgb(pix)
    assert rgb[0].tolist() == [1.0, 2.0, 3.0] and rgb[1].tolist() == [1.0, 1.0, 1.0]
    assert rgb[2].tolist() == [np.float32(0.3) / np.float32(3), np.float32(0.6) / np.float32(3), np.float32(0.9) / np.float32(3)]
    path = tmp_path / "f.pfm"
    write_pfm(str(path), rgb, 2, 2)
    raw = open(path, "rb").read()
    head = ("PF" + chr(10) + "2 2" + chr(10) + "-1.0" + chr(10)).encode()
    assert raw.startswith(head)
    data = np.frombuffer(raw[len(head):], "<f4").reshape(2, 2, 3)
    assert np.array_equal(data[::-1].reshape(4, 3), rgb)
