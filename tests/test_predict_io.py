"""Predict-time input/output: ada_mvs_amd.datasets against files and records the
reference's own code produced -- datasets/data_io.py (tests/golden/io/, tools/gen_golden_io.py) and the cv2-free
functions of datasets/preprocess.py / predict_oblique.py (preprocess.npz, tools/gen_golden_preprocess.py: scale_camera,
crop_input, create_cams, center_image).  The one piece without a reference run is cv2.resize (cv2 is absent from the
build container): scale_image restates OpenCV's published 8-bit fixed-point INTER_LINEAR and is held to hand-derived
vectors below -- parity-unpinned against a cv2 run, and said so here."""
import argparse
import json
import os

import numpy as np
import pytest
from PIL import Image

import ada_mvs_amd  # noqa: F401
from ada_mvs_amd.datasets import data_io, find_dataset_def, preprocess

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io")
EXPECTED = json.load(open(os.path.join(GOLD, "expected.json")))


def test_camera_and_image_records_match_reference():
    cams = data_io.read_cameras_text(os.path.join(GOLD, "camera_info.txt"))
    assert sorted(map(str, cams)) == sorted(EXPECTED["cameras"])
    for k, c in cams.items():
        e = EXPECTED["cameras"][str(k)]
        assert (c.camera_id, c.size, c.pixelsize) == (e["camera_id"], e["size"], e["pixelsize"])
        assert [float(x) for x in c.focallength] == e["focallength"] and [float(x) for x in c.x0y0] == e["x0y0"]
        assert [float(x) for x in c.distortion] == e["distortion"]
    imgs = data_io.read_images_text(os.path.join(GOLD, "image_info.txt"))
    assert sorted(map(str, imgs)) == sorted(EXPECTED["images"])
    for k, p in imgs.items():
        e = EXPECTED["images"][str(k)]
        assert (p.image_id, p.camera_id, p.name) == (e["image_id"], e["camera_id"], e["name"])
        assert p.rotation_matrix.tolist() == e["rotation_matrix"] and p.rotation_matrix.shape == (3, 3)
        assert p.project_center.tolist() == e["project_center"] and p.depth.tolist() == e["depth"]
        assert (p.camera_coordinate_type, p.rotation_type, p.translation_type) == \
               (e["camera_coordinate_type"], e["rotation_type"], e["translation_type"])


def test_path_and_view_pair_lists_match_reference(capsys):
    paths, names = data_io.read_images_path_text(os.path.join(GOLD, "image_path.txt"))
    assert {str(k): v for k, v in paths.items()} == EXPECTED["paths"]
    assert {str(k): v for k, v in names.items()} == EXPECTED["names"]
    for n, want in EXPECTED["view_pairs"].items():
        assert data_io.read_view_pair_text(os.path.join(GOLD, "viewpair.txt"), int(n)) == want
    assert "< num_views:" in capsys.readouterr().out          # the padding notice of the reference
    got = data_io.read_view_pair_text(os.path.join(GOLD, "viewpair.txt"), 5)
    assert all(g[0] != 12 for g in got)                       # the viewpoint without sources is dropped
    assert got[1] == [11, 10, 12, 10, 10, 10]                 # padded with the first source up to view_num entries


@pytest.mark.parametrize("name,scale", [("depth", 1), ("color", 2), ("column", 1)])
def test_save_pfm_bytes_match_reference(tmp_path, name, scale):
    arr = np.load(os.path.join(GOLD, "pfm_inputs.npz"))[name]
    out = str(tmp_path / (name + ".pfm"))
    data_io.save_pfm(out, arr, scale=scale)
    assert open(out, "rb").read() == open(os.path.join(GOLD, name + ".pfm"), "rb").read()
    back, s = data_io.read_pfm(os.path.join(GOLD, name + ".pfm"))
    assert s == float(scale) and back.dtype == np.float32
    assert np.array_equal(back, arr.reshape(back.shape))


def test_pfm_error_behaviour(tmp_path):
    with pytest.raises(Exception, match="float32"):
        data_io.save_pfm(str(tmp_path / "a.pfm"), np.zeros((2, 2), dtype=np.float64))
    with pytest.raises(Exception, match="dimensions"):
        data_io.save_pfm(str(tmp_path / "a.pfm"), np.zeros((2, 2, 2), dtype=np.float32))
    bad = tmp_path / "bad.pfm"
    bad.write_bytes(b"P6\n2 2\n-1.0\n" + bytes(16))
    with pytest.raises(Exception, match="Not a PFM"):
        data_io.read_pfm(str(bad))
    bad.write_bytes(b"Pf\n2x2\n-1.0\n" + bytes(16))
    with pytest.raises(Exception, match="Malformed"):
        data_io.read_pfm(str(bad))
    big = tmp_path / "big.pfm"                                # big-endian files (positive scale) are read too
    big.write_bytes(b"Pf\n2 1\n1.0\n" + np.array([1.5, -2.0], dtype=">f4").tobytes())
    data, s = data_io.read_pfm(str(big))
    assert s == 1.0 and data.tolist() == [[1.5, -2.0]]


def test_write_red_cam_text_matches_reference(tmp_path):
    cam = np.load(os.path.join(GOLD, "cam_input.npy"))
    out = str(tmp_path / "cam.txt")
    data_io.write_red_cam(out, cam, "/data/whu/view0/000.jpg")
    assert open(out).read() == open(os.path.join(GOLD, "cam.txt")).read()


def test_scale_and_crop_cameras():
    cam = np.zeros((2, 4, 4), dtype=np.float32)
    cam[1, :3, :3] = [[100, 0.5, 40], [0.25, 110, 30], [0, 0, 1]]
    cam[1, 3] = [400, 1, 192, 600]
    half = preprocess.scale_camera(cam, 0.5)
    assert half is not cam and half[1, 0, 0] == 50 and half[1, 1, 1] == 55 and half[1, 0, 2] == 20 and half[1, 1, 2] == 15
    assert half[1, 0, 1] == 0.5 and half[1, 1, 0] == 0.25 and np.array_equal(half[1, 3], cam[1, 3])     # skew, depth row kept
    cams = preprocess.scale_mvs_camera([cam.copy(), cam.copy()], 2)
    assert cams[1][1, 0, 0] == 200
    img = np.arange(70 * 100 * 3, dtype=np.uint8).reshape(70, 100, 3)
    out, c2 = preprocess.crop_input(img, cam.copy(), max_h=64, max_w=96)          # larger than the limits: cut
    assert out.shape == (64, 96, 3) and np.array_equal(out, img[:64, :96]) and c2[1, 0, 2] == 40 and c2[1, 1, 2] == 30
    out, _ = preprocess.crop_input(img, cam.copy(), max_h=128, max_w=128)         # below: rounded up, slice ends at the border
    assert out.shape == (70, 100, 3)
    out, _ = preprocess.crop_input(img, cam.copy(), max_h=128, max_w=96, resize_scale=0.5)   # limits scale with resize_scale
    assert out.shape == (64, 48, 3)
    out, _, d = preprocess.crop_input(img, cam.copy(), depth_image=img[..., 0], max_h=64, max_w=96)
    assert d.shape == (64, 96)


PRE = np.load(os.path.join(GOLD, "preprocess.npz"))


def test_scale_camera_and_crop_input_match_reference_run():
    """scale_camera / crop_input against what the reference's own functions returned (tools/gen_golden_preprocess.py),
    including sides below the limit that are no multiple of 32 (the slice ends at the image border) and limits
    scaled by resize_scale."""
    cam = PRE["create_cams_192"][0]
    for k in range(4):
        got = preprocess.scale_camera(cam, float(PRE["scale_camera_%d_scale" % k]))
        assert got.dtype == PRE["scale_camera_%d" % k].dtype and np.array_equal(got, PRE["scale_camera_%d" % k])
    for k, (h, w, mh, mw, rs) in enumerate(PRE["crop_cases"]):
        h, w, mh, mw = int(h), int(w), int(mh), int(mw)
        rs = int(rs) if rs == int(rs) else float(rs)
        image = (np.arange(h * w * 3) % 251).astype(np.uint8).reshape(h, w, 3)
        depth = np.arange(h * w, dtype=np.float32).reshape(h, w)
        im2, c2, d2 = preprocess.crop_input(image, cam.copy(), depth_image=depth, max_h=mh, max_w=mw, resize_scale=rs)
        assert tuple(im2.shape) == tuple(PRE["crop_%d_shape" % k]) and tuple(d2.shape) == tuple(PRE["crop_%d_depth_shape" % k])
        assert np.array_equal(c2, PRE["crop_%d_cam" % k])
        assert int(im2.astype(np.int64).sum()) == int(PRE["crop_%d_checksum" % k][0])
        assert float(d2.astype(np.float64).sum()) == float(PRE["crop_%d_checksum" % k][1])


def test_create_cams_and_center_image_match_reference_run():
    """MVSDataset.create_cams (axis flip, float32 pose inverse, depth row) and center_image, bit for bit against the
    reference's methods run on the records of tests/golden/io."""
    ds_cls = find_dataset_def("predict_oblique")
    cams = data_io.read_cameras_text(os.path.join(GOLD, "camera_info.txt"))
    imgs = data_io.read_images_text(os.path.join(GOLD, "image_info.txt"))
    for nd in (192, 384):
        for row, image_id in enumerate(PRE["create_cams_ids"]):
            got = ds_cls.create_cams(None, imgs[int(image_id)], cams, nd, 0.1)
            want = PRE["create_cams_%d" % nd][row]
            assert got.dtype == want.dtype == np.float32
            assert np.array_equal(got[1], want[1])                                  # intrinsics and depth row: exact
            assert np.allclose(got[0], want[0], rtol=0, atol=1e-6 * np.abs(want[0]).max())     # LAPACK inverse, same float32 input
    img = PRE["center_input"]
    assert np.array_equal(ds_cls.center_image(None, img, mode="mean"), PRE["center_mean"])
    assert np.array_equal(ds_cls.center_image(None, img, mode="standard"), PRE["center_standard"])
    assert np.array_equal(preprocess.center_image(img), PRE["center_mean"])


def test_scale_image_fixed_point_hand_vector():
    """cv2.resize(INTER_LINEAR) on uint8, derived by hand from OpenCV's arithmetic (11-bit weights, int32 rows,
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2):

    row [0, 100, 200, 50], fx = fy = 1.5 -> 2 x 6.  scale_x = 2/3; sample centres -0.167, 0.5, 1.167, 1.833, 2.5, 3.167:
      x weights (2048, 0) on pixel 0 | (1024, 1024) on 0, 1 | (1707, 341) on 1, 2 | (341, 1707) on 1, 2 | (1024, 1024) on 2, 3 |
      (2048, 0) on pixel 3   ->   horizontal sums S = 0, 102400, 238900, 375500, 256000, 102400
    y: centre -0.167 -> rows (0, 0) with weights (341, 1707); centre 0.5 -> rows (0, 0) with weights (1024, 1024)
      S = 238900: S >> 4 = 14931;  (341 * 14931) >> 16 = 77, (1707 * 14931) >> 16 = 388 -> (465 + 2) >> 2 = 116
                                    (1024 * 14931) >> 16 = 233, twice                    -> (466 + 2) >> 2 = 117
      (the exact bilinear value is 116.65: float arithmetic would give 117 in both rows)
      S = 375500 -> 183, 183;  S = 256000 -> 125, 125;  S = 102400 -> 50, 50."""
    img = np.array([[0, 100, 200, 50]], dtype=np.uint8)
    out = preprocess.scale_image(img, 1.5)
    assert out.dtype == np.uint8 and out.shape == (2, 6)
    assert out.tolist() == [[0, 50, 116, 183, 125, 50], [0, 50, 117, 183, 125, 50]]
    rgb = np.stack((img, img[:, ::-1], np.full_like(img, 9)), -1)                     # channels are filtered independently
    out3 = preprocess.scale_image(rgb, 1.5)
    assert out3[..., 0].tolist() == out.tolist() and out3[..., 1].tolist() == [r[::-1] for r in out.tolist()]
    assert np.all(out3[..., 2] == 9)                                                  # a constant stays constant


def test_scale_image_geometry_and_special_cases():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(8, 12, 3)).astype(np.uint8)
    assert preprocess.scale_image(img, 1) is img
    half = preprocess.scale_image(img, 0.5)                                         # exactly 1/2: OpenCV switches to INTER_AREA
    box = img.reshape(4, 2, 6, 2, 3).astype(np.int64).sum(axis=(1, 3))
    assert half.shape == (4, 6, 3) and half.dtype == np.uint8 and np.array_equal(half, ((box + 2) >> 2).astype(np.uint8))
    odd = preprocess.scale_image(np.arange(35, dtype=np.uint8).reshape(5, 7), 0.5)   # cvRound: 2.5 -> 2, 3.5 -> 4
    assert odd.shape == (2, 4)
    assert odd[0, 0] == (0 + 1 + 7 + 8 + 2) >> 2 and odd[1, 2] == (18 + 19 + 25 + 26 + 2) >> 2
    assert odd[0, 3] == 10 and odd[1, 3] == 24                                       # block cut by the border: mean of (6, 13), of (20, 27): 9.5 -> 10 (even), 23.5 -> 24
    for scale in (0.3, 0.75, 1.25, 2.0, 3.1):                                        # constants survive the fixed point; sizes are cvRound
        c = preprocess.scale_image(np.full((9, 14), 255, dtype=np.uint8), scale)
        assert c.shape == (int(np.rint(9 * scale)), int(np.rint(14 * scale))) and np.all(c == 255)
    smooth = (np.add.outer(np.arange(20), np.arange(30)) * 3).astype(np.uint8)       # a ramp: fixed point within one grey level of float bilinear
    up = preprocess.scale_image(smooth, 2.0)
    ref = preprocess.scale_image(smooth.astype(np.float32), 2.0)
    assert up.shape == (40, 60) and np.abs(up.astype(np.float32) - ref).max() <= 1.0
    f = preprocess.scale_image(img[..., 0].astype(np.float32), 2.0)                  # float images: plain bilinear, border clamp
    assert f.shape == (16, 24) and f[0, 0] == img[0, 0, 0] and f[-1, -1] == img[-1, -1, 0]
    assert np.isclose(f[1, 1], (0.75 * 0.75 * img[0, 0, 0] + 0.75 * 0.25 * (float(img[0, 1, 0]) + img[1, 0, 0]) + 0.0625 * img[1, 1, 0]))
    s16 = preprocess.scale_image(img[..., 0].astype(np.int16) - 100, 0.5)            # signed 16-bit: the general path (no unsigned box)
    assert s16.dtype == np.int16 and s16.shape == (4, 6)
    near = preprocess.scale_image(img, 0.5, interpolation="biculic")                 # nearest neighbour, as the reference maps it
    assert np.array_equal(near, img[::2, ::2])
    im2, cam2 = preprocess.scale_input(img, np.ones((2, 4, 4), dtype=np.float32), scale=0.5)
    assert im2.shape == (4, 6, 3) and cam2[1, 0, 0] == 0.5


def test_center_image_statistics():
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, size=(16, 20, 3)).astype(np.uint8)
    out = preprocess.center_image(img)
    assert out.dtype == np.float32 and out.shape == img.shape
    assert np.allclose(out.mean(axis=(0, 1)), 0, atol=1e-5) and np.allclose(out.std(axis=(0, 1)), 1, atol=1e-5)
    flat = preprocess.center_image(np.full((4, 4, 3), 7, dtype=np.uint8))            # zero variance: the 1e-8 guard
    assert np.all(flat == 0)


def _write_scene(folder, n_img=4, h=128, w=192):
    """A synthetic whu-omvs predict folder: n_img nadir-ish views (camera Y up, Z back) over flat ground."""
    rng = np.random.RandomState(3)
    os.makedirs(os.path.join(folder, "images", "view0"), exist_ok=True)
    fx, fy, cx, cy = 400.0, 402.0, w / 2 - 0.5, h / 2 - 0.5
    with open(os.path.join(folder, "camera_info.txt"), "w") as f:
        f.write("# CAMERA_ID WIDTH HEIGHT PIXELSIZE PARAMS DISTORTION\n0 %d %d 0.005 %r %r %r %r 0 0 0 0 0\n" % (w, h, fx, fy, cx, cy))
    lines, paths = [], []
    for i in range(n_img):
        a = 0.02 * i
        # world Z up; the camera looks down: camera axes (X right, Y up, Z back) = world (X, Y, Z) rotated a little
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        t = np.array([10.0 * i, 3.0 * i, 500.0])
        name = "view0/%03d.png" % i
        path = os.path.join(folder, "images", name)
        Image.fromarray(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(path)
        lines.append("%d 0 %s %s 400.0 600.0 %s" % (i, " ".join(repr(float(v)) for v in R.reshape(-1)),
                                                    " ".join(repr(float(v)) for v in t), name))
        paths.append("%d %03d.png %s" % (i, i, path))
    open(os.path.join(folder, "image_info.txt"), "w").write("# header\n" + "\n".join(lines) + "\n")
    open(os.path.join(folder, "image_path.txt"), "w").write("%d\n%s\n" % (n_img, "\n".join(paths)))
    with open(os.path.join(folder, "viewpair.txt"), "w") as f:
        f.write("%d\n" % n_img)
        for i in range(n_img):
            src = [j for j in range(n_img) if j != i]
            f.write("%d\n%d %s\n" % (i, len(src), " ".join("%d %.1f" % (j, 1.0) for j in src)))
    return (fx, fy, cx, cy)


def _args(**kw):
    d = dict(min_interval=0.1, interval_scale=1, numdepth=16, resize_scale=0.5, sample_scale=1, max_h=5504, max_w=3712,
             normalize="mean")
    d.update(kw)
    return argparse.Namespace(**d)


def test_create_cams_converts_axes_and_pose(tmp_path):
    fx, fy, cx, cy = _write_scene(str(tmp_path))
    ds = find_dataset_def("predict_oblique")(str(tmp_path), 3, _args())
    photo = ds.image_params_dict[2]
    cam = ds.create_cams(photo, ds.cam_params_dict, num_depth=16, min_interval=0.1)
    assert cam.dtype == np.float32 and cam.shape == (2, 4, 4)
    rng = np.random.RandomState(5)
    X = rng.rand(20, 3) * [200, 200, 50]                                       # ground points below the camera
    y_up = (X - photo.project_center) @ photo.rotation_matrix                  # Rwc^T (X - C): camera axes X right, Y up, Z back
    want = y_up * [1, -1, -1]                                                  # X right, Y down, Z forward
    got = X @ cam[0, :3, :3].T.astype(np.float64) + cam[0, :3, 3]
    assert np.allclose(got, want, atol=2e-3) and np.all(want[:, 2] > 0)        # in front of the converted camera
    assert np.allclose(cam[0, 3], [0, 0, 0, 1])
    assert np.allclose(cam[1, :3, :3], [[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    assert np.allclose(cam[1, 3], [400.0, 200.0 / 16, 16, 600.0])


def test_dataset_sample_layout(tmp_path):
    fx, fy, cx, cy = _write_scene(str(tmp_path))
    ds = find_dataset_def("predict_oblique")(str(tmp_path), 3, _args())
    assert len(ds) == 4
    s = ds[1]
    assert s["imgs"].shape == (3, 3, 64, 96) and s["imgs"].dtype == np.float32      # 128x192 at resize_scale 0.5
    assert np.allclose(s["imgs"].mean(axis=(2, 3)), 0, atol=1e-4)
    assert s["outimage"].shape == (64, 96, 3) and s["outimage"].dtype == np.uint8
    assert s["depth_values"].tolist() == [400.0, 600.0] and s["depth_values"].dtype == np.float32
    assert s["out_name"] == "001" and s["out_view"] == "view0" and s["ref_image_path"].endswith("view0/001.png")
    K = s["outcam"][1, :3, :3]
    assert np.allclose(K, [[fx / 2, 0, cx / 2], [0, fy / 2, cy / 2], [0, 0, 1]])
    p3 = s["proj_matrices"]["stage3"]
    assert p3.shape == (3, 4, 4) and p3.dtype == np.float32
    assert np.allclose(p3[0, :3], K @ s["outcam"][0, :3], rtol=1e-5, atol=1e-3) and np.allclose(p3[:, 3], [0, 0, 0, 1])
    p2, p1 = s["proj_matrices"]["stage2"], s["proj_matrices"]["stage1"]
    assert np.array_equal(p2[:, :2], p3[:, :2] / 2) and np.array_equal(p1[:, :2], p3[:, :2] / 4)
    assert np.array_equal(p2[:, 2:], p3[:, 2:]) and np.array_equal(p1[:, 2:], p3[:, 2:])
    # view order = viewpair order: reference view 1, sources 0 and 2
    first = np.array(Image.open(ds.image_paths[1]))
    want = preprocess.center_image(preprocess.scale_image(first, 0.5)).transpose(2, 0, 1)
    assert np.allclose(s["imgs"][0], want)
    # a ground point projects to the same pixel in the half-resolution image as the y-up pinhole model says
    photo = ds.image_params_dict[1]
    X = np.array([40.0, 30.0, 20.0])
    yup = (X - photo.project_center) @ photo.rotation_matrix
    u = fx / 2 * (yup[0] / -yup[2]) + cx / 2
    v = fy / 2 * (-yup[1] / -yup[2]) + cy / 2
    q = p3[0].astype(np.float64) @ np.append(X, 1.0)
    assert np.allclose([q[0] / q[2], q[1] / q[2]], [u, v], atol=1e-2)


def test_predict_cli_options_and_model_errors():
    from ada_mvs_amd import predict
    a = predict.build_parser().parse_args(["--data_folder", "d", "--output_folder", "o"])
    assert (a.model, a.dataset, a.view_num, a.numdepth, a.max_w, a.max_h) == ("adamvs", "predict_oblique", 5, 192, 3712, 5504)
    assert (a.resize_scale, a.sample_scale, a.ndepths, a.depth_inter_r, a.cr_base_chs) == (0.5, 1, "48,32,8", "4,2,1", "8,8,8")
    assert a.display is True and a.batch_size == 1
    a.model = "other"
    with pytest.raises(Exception, match="Not implemented"):
        predict.build_model(a, "cpu")
    for name in ("adamvs", "msrednet"):
        a.model = name
        with pytest.raises(Exception, match="loadckpt"):
            predict.build_model(a, "cpu")
    with pytest.raises(RuntimeError, match="MI355X"):          # no GPU in the CPU test run: loud, no fallback
        import torch
        if torch.cuda.is_available():
            raise RuntimeError("MI355X present: nothing to check")
        predict.predict_depth(a)


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["adamvs", "msrednet"])
def test_predict_end_to_end_writes_reference_layout(tmp_path, model_name):
    """The whole predict_whu.py chain on a synthetic folder: files, formats, and values equal to a direct model call."""
    import torch
    from ada_mvs_amd import predict, synth
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    from ada_mvs_amd.models.msrednet import Infer_CascadeREDNet
    src, out = tmp_path / "src", tmp_path / "out"
    _write_scene(str(src))
    argv = ["--data_folder", str(src), "--output_folder", str(out), "--view_num", "3", "--numdepth", "16", "--model", model_name,
            "--ndepths", "16,8,4", "--seeded_weights", "0", "--batch_size", "2", "--num_workers", "0"]
    assert predict.main(argv) == 4
    if model_name == "adamvs":        # the default replays one captured graph per shape (graphed.py); launched eagerly: the same bytes
        eager = tmp_path / "eager"
        assert predict.main(argv[:3] + [str(eager)] + argv[4:] + ["--graph", "0"]) == 4
        for i in range(4):
            for f in ("%03d_init.pfm" % i, "%03d_prob.pfm" % i):
                assert open(str(out / "view0" / f), "rb").read() == open(str(eager / "view0" / f), "rb").read(), f
    ds = find_dataset_def("predict_oblique")(str(src), 3, predict.build_parser().parse_args(argv))
    cls = Infer_AdaMVSNet if model_name == "adamvs" else Infer_CascadeREDNet
    model = cls(16, [16, 8, 4], [4.0, 2.0, 1.0], False, [8, 8, 8])
    model.load_state_dict(synth.seeded_state_dict(model, seed=0))
    model = model.cuda().eval()
    for i in range(4):
        s = ds[i]
        folder = out / "view0"
        depth, scale = data_io.read_pfm(str(folder / ("%03d_init.pfm" % i)))
        prob, _ = data_io.read_pfm(str(folder / ("%03d_prob.pfm" % i)))
        assert scale == 1.0 and depth.shape == (64, 96) and prob.shape == (64, 96)
        with torch.no_grad():
            o = model(torch.from_numpy(s["imgs"])[None].cuda(),
                      {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()},
                      torch.from_numpy(s["depth_values"])[None].cuda())
        assert np.allclose(depth, o["depth"][0].cpu().numpy(), rtol=1e-5, atol=1e-3)
        assert np.allclose(prob, o["photometric_confidence"][0].cpu().numpy(), atol=1e-4)
        assert np.isfinite(depth).all() and 300.0 < depth.min() and depth.max() < 700.0         # later stages may leave [min, max]
        assert 0 <= prob.min() and prob.max() <= 1 + 1e-5
        saved = np.array(Image.open(str(folder / ("%03d.jpg" % i))))                           # RGBA PNG bytes, as plt.imsave writes them
        assert saved.shape[-1] == 4 and np.array_equal(saved[..., :3], s["outimage"]) and np.all(saved[..., 3] == 255)
        cam_txt = open(str(folder / ("%03d.txt" % i))).read().splitlines()
        assert cam_txt[0] == "extrinsic: XrightYdown, [Rcw|tcw]" and cam_txt[-1] == s["ref_image_path"]
        assert cam_txt[11].split() == [str(v) for v in s["outcam"][1, 3]]
        assert (folder / "color" / ("%03d_init.png" % i)).exists() and (folder / "color" / ("%03d_prob.png" % i)).exists()
