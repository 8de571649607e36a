"""Golden fixtures for the predict-time file formats, made by the reference itself.

    python tools/gen_golden_io.py          (in the build container; needs /root/reference)

Writes tests/golden/io/: four small input files in the whu-omvs text formats (content invented here), and what the
reference's own datasets/data_io.py makes of them: expected.json (the parsed records), depth.pfm / color.pfm /
column.pfm (its save_pfm on seeded arrays, inputs in pfm_inputs.npz), cam.txt (its write_red_cam).
datasets/data_io.py imports only numpy; datasets/preprocess.py and predict_oblique.py need cv2 / imageio, which this
image lacks, so resize / crop / camera conversion have no reference-run fixture (tests check them against formulas).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "io")
sys.path.insert(0, "/root/reference")
from datasets import data_io as ref  # noqa: E402


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(7)
    with open(os.path.join(OUT, "camera_info.txt"), "w") as f:
        f.write("# CAMERA_ID WIDTH HEIGHT PIXELSIZE fx fy cx cy k1 k2 k3 p1 p2\n\n")
        f.write("0 3712 5504 0.0046 7800.5 7801.25 1856.0 2752.5 1e-3 -2e-4 0 1.5e-5 0\n")
        f.write("3 160 128 0.006 210.0 208.0 79.5 63.5 0 0 0 0 0\n")
    with open(os.path.join(OUT, "image_info.txt"), "w") as f:
        f.write("# IMAGE_ID CAMERA_ID Rwc[9] twc[3] MINDEPTH MAXDEPTH NAME\n")
        for i in range(6):
            R = rot(0.05 * i, -0.03 * i + 3.0, 0.4 + 0.01 * i)
            t = np.array([500.0 + 31.5 * i, -200.0 + 7.25 * i, 650.0 - 2.0 * i])
            vals = list(R.reshape(-1)) + list(t) + [400.0 + i, 600.0 + 2 * i]
            f.write("%d %d %s view%d/%03d.jpg\n" % (10 + i, 3 if i % 2 else 0, " ".join(repr(float(v)) for v in vals), i % 3, i))
    with open(os.path.join(OUT, "image_path.txt"), "w") as f:
        f.write("6\n")
        for i in range(6):
            f.write("%d %03d.jpg /data/whu/view%d/%03d.jpg\n" % (10 + i, i, i % 3, i))
    with open(os.path.join(OUT, "viewpair.txt"), "w") as f:
        f.write("4\n")
        f.write("10\n5 11 0.9 12 0.8 13 0.7 14 0.6 15 0.5\n")
        f.write("11\n2 10 0.9 12 0.5\n")               # padded with the first source
        f.write("12\n0\n")                             # no source views: dropped
        f.write("13\n4 12 3.5 14 2.5 15 1.5 10 0.5\n")

    cams = ref.read_cameras_text(os.path.join(OUT, "camera_info.txt"))
    imgs = ref.read_images_text(os.path.join(OUT, "image_info.txt"))
    paths, names = ref.read_images_path_text(os.path.join(OUT, "image_path.txt"))
    expected = {
        "cameras": {str(k): {"camera_id": c.camera_id, "size": c.size, "pixelsize": c.pixelsize,
                             "focallength": [float(x) for x in c.focallength], "x0y0": [float(x) for x in c.x0y0],
                             "distortion": [float(x) for x in c.distortion]} for k, c in cams.items()},
        "images": {str(k): {"image_id": p.image_id, "camera_id": p.camera_id,
                            "rotation_matrix": p.rotation_matrix.tolist(), "project_center": p.project_center.tolist(),
                            "depth": p.depth.tolist(), "name": p.name,
                            "camera_coordinate_type": p.camera_coordinate_type, "rotation_type": p.rotation_type,
                            "translation_type": p.translation_type} for k, p in imgs.items()},
        "paths": {str(k): v for k, v in paths.items()},
        "names": {str(k): v for k, v in names.items()},
        "view_pairs": {str(n): ref.read_view_pair_text(os.path.join(OUT, "viewpair.txt"), n) for n in (3, 5, 7)},
    }
    depth = (rng.rand(7, 5) * 200 + 400).astype(np.float32)
    color = rng.randn(4, 6, 3).astype(np.float32)
    column = rng.randn(5, 3, 1).astype(np.float32)
    np.savez(os.path.join(OUT, "pfm_inputs.npz"), depth=depth, color=color, column=column)
    ref.save_pfm(os.path.join(OUT, "depth.pfm"), depth)
    ref.save_pfm(os.path.join(OUT, "color.pfm"), color, scale=2)
    ref.save_pfm(os.path.join(OUT, "column.pfm"), column)
    back, scale = ref.read_pfm(os.path.join(OUT, "depth.pfm"))
    expected["read_pfm_depth"] = {"scale": scale, "equal_to_input": bool(np.array_equal(back, depth))}
    cam = np.zeros((2, 4, 4), dtype=np.float32)
    cam[0] = np.linalg.inv(np.block([[rot(0.1, 3.0, 0.4), np.array([[500.0], [-200.0], [650.0]])],
                                     [np.zeros((1, 3)), np.ones((1, 1))]])).astype(np.float32)
    cam[1, :3, :3] = [[3900.25, 0, 928.0], [0, 3900.625, 1376.25], [0, 0, 1]]
    cam[1, 3] = [400.0, (600.0 - 400.0) / 192, 192, 600.0]
    np.save(os.path.join(OUT, "cam_input.npy"), cam)
    ref.write_red_cam(os.path.join(OUT, "cam.txt"), cam, "/data/whu/view0/000.jpg")
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
