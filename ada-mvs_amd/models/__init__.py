"""Host-side mirror of the reference's `models` package for the inference path
(reference models/adamvs.py, models/module.py): same class names, constructor
arguments, forward() signatures, output dicts and state-dict keys; the hot path
runs in libadamvs_hip.so."""
