"""Build libadamvs_hip.so (gfx950) in-tree with hipcc; no torch, no cmake.

    python -m ada_mvs_amd.build            (or __graft_entry__.build())

Objects go to ada-mvs_amd/csrc/_build/, the library next to this file.  hipcc
cross-compiles without a GPU.  Up-to-date objects are reused (mtime check
against every source and header).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libadamvs_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCES = ["api.hip", "geometry.hip", "planesweep.hip", "sweep.hip", "costreg2d.hip", "costreg2d_wino.hip", "costreg2d_bf16x3.hip", "slice_red.hip", "slice_red_bf16x3.hip", "recurrence.hip", "featnet.hip", "msred.hip"]
ARCH = "gfx950"
# Per-source flags.  slice_red.hip: the IR load/store vectorizer fuses the three horizontally adjacent taps of an
# MFMA B-fragment into one ds_read_b96 at a 4-byte-aligned address; gfx950 executes those with "unaligned" stalls
# (SQ_LDS_UNALIGNED_STALL ~ the MFMA busy time in the two-row conv1).  Without it the reads stay dword pairs
# (ds_read2_b32, formed later by the machine pass); every intended wide access in that file is an explicit vector type.
SOURCE_FLAGS = {"slice_red.hip": ["-mllvm", "-amdgpu-load-store-vectorizer=0"],
                "recurrence.hip": ["-mllvm", "-amdgpu-load-store-vectorizer=0"],
                "featnet.hip": ["-mllvm", "-amdgpu-load-store-vectorizer=0"]}
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-I", INCLUDE, "-I", CSRC]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newest_dep():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(INCLUDE, "adamvs_hip.h"))
    return max(os.path.getmtime(d) for d in deps)


def _compile(src, extra):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_time = max(hdr_time, os.path.getmtime(os.path.join(INCLUDE, "adamvs_hip.h")))
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(os.path.join(CSRC, src)), hdr_time) and not extra:
        return obj, False
    cmd = [_hipcc()] + FLAGS + SOURCE_FLAGS.get(src, []) + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout[-4000:], r.stderr[-8000:]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, extra_flags=(), verbose=True):
    """extra_flags: remarks and temporaries only (-R..., -save-temps...).  Anything that changes the generated code (-D, -O, -m)
    would land in the PRODUCT's object directory and library, and the next plain build() would call that "up to date":
    variants belong to tools/build_variant.py, which builds into _build_<name>/ and libadamvs_hip.<name>.so."""
    bad = [f for f in extra_flags if not (f.startswith("-R") or f.startswith("-save-temps"))]
    if bad:
        raise ValueError("build(extra_flags=%r): only -R* / -save-temps* here; build variants with tools/build_variant.py" % (bad,))
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, extra_flags), SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout[-4000:], r.stderr[-8000:]))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra_flags=tuple(a for a in sys.argv[1:] if a.startswith("-R") or a.startswith("-save")))
