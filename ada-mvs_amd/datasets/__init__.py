"""Predict-time input/output of the reference: the whu-omvs text formats, the
camera conversion, resize / crop / normalise, PFM and camera writers.  Pure numpy + PIL (the reference needs
cv2 and imageio).  `find_dataset_def` mirrors reference datasets/__init__.py:4-8."""
import importlib


def find_dataset_def(dataset_name):
    """`predict_oblique` -> ada_mvs_amd.datasets.predict_oblique.MVSDataset (reference datasets/__init__.py:4-8)."""
    module = importlib.import_module("%s.%s" % (__name__, dataset_name))
    return getattr(module, "MVSDataset")
