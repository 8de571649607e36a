#!/usr/bin/env python3
"""Build an experimental variant of the library next to the shipped one (A/B runs on the GPU box):

    python tools/build_variant.py <name> -DADAMVS_EXP_FOO [-D...]   ->  ada-mvs_amd/libadamvs_hip.<name>.so
    ADAMVS_LIB_PATH=ada-mvs_amd/libadamvs_hip.<name>.so python bench.py ...

Objects go to ada-mvs_amd/csrc/_build_<name>/; the shipped library and its objects are untouched.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import build as B  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    obj_dir = os.path.join(B.CSRC, "_build_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    lib = os.path.join(B.HERE, "libadamvs_hip.%s.so" % name)

    def cc(src):
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = [B._hipcc()] + B.FLAGS + B.SOURCE_FLAGS.get(src, []) + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-4000:])
        return obj

    with ThreadPoolExecutor(max_workers=5) as ex:
        objs = list(ex.map(cc, B.SOURCES))
    subprocess.run([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", lib] + objs, check=True)
    print("built", lib)


if __name__ == "__main__":
    main()
