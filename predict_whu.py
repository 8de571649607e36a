"""Entry point with the reference's name and options (reference predict_whu.py): see ada_mvs_amd/predict.py."""
import ada_mvs_amd  # noqa: F401  (registers the package directory `ada-mvs_amd`)
from ada_mvs_amd.predict import main

if __name__ == "__main__":
    main()
