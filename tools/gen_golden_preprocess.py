"""Golden fixtures for the cv2-free part of the predict-time preprocessing, made by running
the reference's own functions.

    python tools/gen_golden_preprocess.py          (in the build container; needs /root/reference)

datasets/preprocess.py imports cv2 and datasets/predict_oblique.py imports imageio at module level; neither is installed
here.  Harness-side shims let the modules IMPORT: `cv2` and `imageio` become stub modules whose every attribute access
raises (so nothing below can silently depend on them), and `np.float` -- an alias numpy removed in 1.24 that
predict_oblique.py:81 still uses -- is set to the builtin float it used to name.  What is then executed is the
reference's code, unmodified: scale_camera, crop_input (preprocess.py:22-34, 68-99), MVSDataset.create_cams and
MVSDataset.center_image (predict_oblique.py:53-111), fed with the parsed records of tests/golden/io (the reference's
own data_io.py parses them).  scale_image (preprocess.py:44-54) is cv2.resize and stays unpinned against a reference run.

Writes tests/golden/io/preprocess.npz.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "io")


class _Absent(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise RuntimeError("%s.%s: %s is not installed; the fixture generator only runs code that does not need it" % (self.__name__, name, self.__name__))


for _name in ("cv2", "imageio"):
    sys.modules[_name] = _Absent(_name)
# `from imageio import imread, imsave, imwrite` needs the names to exist
for _fn in ("imread", "imsave", "imwrite"):
    def _raise(*a, _n=_fn, **k):
        raise RuntimeError("imageio.%s is not available" % _n)
    object.__setattr__(sys.modules["imageio"], _fn, _raise)
if not hasattr(np, "float"):
    np.float = float
sys.path.insert(0, "/root/reference")
from datasets import data_io as ref_io  # noqa: E402
from datasets import preprocess as ref_pre  # noqa: E402
from datasets import predict_oblique as ref_ds  # noqa: E402


def main():
    cams = ref_io.read_cameras_text(os.path.join(OUT, "camera_info.txt"))
    imgs = ref_io.read_images_text(os.path.join(OUT, "image_info.txt"))
    out = {}
    # ---- create_cams for every image record, two hypothesis counts
    ids = sorted(imgs)
    out["create_cams_ids"] = np.array(ids)
    for nd in (192, 384):
        out["create_cams_%d" % nd] = np.stack([ref_ds.MVSDataset.create_cams(None, imgs[i], cams, nd, 0.1) for i in ids])
    # ---- scale_camera
    cam = out["create_cams_192"][0].copy()
    for k, s in enumerate((0.5, 0.25, 1.0, 1.7)):
        out["scale_camera_%d" % k] = ref_pre.scale_camera(cam, scale=s)
        out["scale_camera_%d_scale" % k] = np.float64(s)
    # ---- crop_input: (h, w, max_h, max_w, resize_scale) incl. sides below the limit that are no multiple of 32
    cases = [(50, 70, 384, 768, 1), (400, 800, 384, 768, 1), (400, 100, 384, 768, 1), (2752, 1856, 5504, 3712, 0.5),
             (300, 900, 5504, 3712, 0.1), (64, 96, 64, 96, 1), (65, 97, 64, 96, 1.5)]
    out["crop_cases"] = np.array(cases, dtype=np.float64)
    for k, (h, w, mh, mw, rs) in enumerate(cases):
        image = (np.arange(h * w * 3) % 251).astype(np.uint8).reshape(h, w, 3)
        depth = np.arange(h * w, dtype=np.float32).reshape(h, w)
        c = cam.copy()
        im2, c2, d2 = ref_pre.crop_input(image, c, depth_image=depth, max_h=mh, max_w=mw, resize_scale=rs)
        out["crop_%d_shape" % k] = np.array(im2.shape)
        out["crop_%d_cam" % k] = c2
        out["crop_%d_depth_shape" % k] = np.array(d2.shape)
        out["crop_%d_checksum" % k] = np.array([int(im2.astype(np.int64).sum()), float(d2.astype(np.float64).sum())])
    # ---- center_image ('mean' and 'standard')
    rng = np.random.RandomState(11)
    img = rng.randint(0, 256, size=(12, 20, 3)).astype(np.uint8)
    out["center_input"] = img
    out["center_mean"] = ref_ds.MVSDataset.center_image(None, img, mode="mean")
    out["center_standard"] = ref_ds.MVSDataset.center_image(None, img, mode="standard")
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **out)
    print("wrote preprocess.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
