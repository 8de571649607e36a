"""Predict-time input/output of the reference: the whu-omvs text formats, the
camera conversion, resize / crop / normalise, PFM and camera writers.  Pure numpy + PIL (the reference needs
cv2 and imageio)."""
from importlib import import_module


def find_dataset_def(dataset_name):
    """Dataset class by module name, as reference datasets/__init__.py:4-8 resolves `--dataset`:
    'predict_oblique' -> ada_mvs_amd.datasets.predict_oblique.MVSDataset."""
    return import_module("." + dataset_name, package=__name__).MVSDataset
