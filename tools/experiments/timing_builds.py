#!/usr/bin/env python3
"""Timing builds WITHOUT touching the product sources: csrc/ is copied to a scratch directory, the textual patches of a variant are
applied to the copy, and the copy is built into ada-mvs_amd/libadamvs_hip.<variant>.so (most of these libraries give wrong
results by construction; they answer "what does this part of the kernel cost").  Use one with ADAMVS_LIB_PATH=<that .so>.

    python tools/experiments/timing_builds.py [variant ...]                  (CPU box: hipcc cross-compiles; default: all)
    python tools/experiments/bx3_costreg_timing/time_variants.py             (GPU box: the bxc_* set, k_conv_dd_bx3)

bxc_*: k_conv_dd_bx3 (CostRegNet2D in the bf16x3 mode, reference models/adamvs.py:229-238); profiles/r06_bx3_costreg_timing.txt.
swp_*: k_sweep_blend (the weighted aggregation, reference models/adamvs.py:495-512); profiles/r06_sweep_timing.txt.
"""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import build as B  # noqa: E402

MFMA3 = '''#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][r] = mfma_bf16(wh[mt], bh[r], acc[mt][r]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][r] = mfma_bf16(wh[mt], bl[r], acc[mt][r]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][r] = mfma_bf16(wl[mt], bh[r], acc[mt][r]);'''
KEEP = '''#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { asm volatile("" ::"v"(wh[mt]), "v"(wl[mt])); }
      asm volatile("" ::"v"(bh[r]), "v"(bl[r]));'''
ONE = '''#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][r] = mfma_bf16(wh[mt], bh[r], acc[mt][r]);
      asm volatile("" ::"v"(bl[r]));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { asm volatile("" ::"v"(wl[mt])); }'''

VARIANTS = {
    # name: [(old, new), ...]
    "bxc_nomfma": [(MFMA3, KEEP)],                                     # everything but the matrix instructions
    "bxc_onemfma": [(MFMA3, ONE)],                                     # one of the three products (what plain bf16 would issue)
    "bxc_noweights": [("        if (t + 1 < NTAP) load_w(w0h, w0l, kb, t + 1);", "        if (t + 1 < NTAP && kb == 0 && t == 0) load_w(w0h, w0l, kb, t + 1);"),
                      ("        if (t + 1 < NTAP) load_w(w1h, w1l, kb, t + 1);", "        if (t + 1 < NTAP && kb == 0 && t == 0) load_w(w1h, w1l, kb, t + 1);")],
    "bxc_nofill": [("    if (kb + 1 < KBT) load_x(xs, (kb + 1) * BX_KB);", "    if (false) load_x(xs, (kb + 1) * BX_KB);"),
                   ("    store_x(xs);\n    load_w(w0h, w0l, kb, 0);", "    if (kb == 0) store_x(xs);\n    load_w(w0h, w0l, kb, 0);")],
    # candidates (results stay right): the next chunk's window requested AFTER the first tap's weight request, so that the wait for tap
    # 1's weights (vmcnt retires in order) does not include it -- it then has two taps instead of one to arrive
    "bxc_xlate": [("    if (kb + 1 < KBT) load_x(xs, (kb + 1) * BX_KB);               // in flight during the MFMAs, stored at the next top\n", ""),
                  ("        if (t + 1 < NTAP) load_w(w1h, w1l, kb, t + 1);\n        tap(w0h, w0l, t);",
                   "        if (t + 1 < NTAP) load_w(w1h, w1l, kb, t + 1);\n        if (t == 0 && kb + 1 < KBT) load_x(xs, (kb + 1) * BX_KB);\n        tap(w0h, w0l, t);")],
    # ... and the B fragments two rows ahead of their MFMAs instead of one
    "bxc_xlate_b2": [("    if (kb + 1 < KBT) load_x(xs, (kb + 1) * BX_KB);               // in flight during the MFMAs, stored at the next top\n", ""),
                     ("        if (t + 1 < NTAP) load_w(w1h, w1l, kb, t + 1);\n        tap(w0h, w0l, t);",
                      "        if (t + 1 < NTAP) load_w(w1h, w1l, kb, t + 1);\n        if (t == 0 && kb + 1 < KBT) load_x(xs, (kb + 1) * BX_KB);\n        tap(w0h, w0l, t);"),
                     ("""      bh[0] = *(const bf16x8*)at;
      bl[0] = *(const bf16x8*)(at + LOB);
    }""", """      bh[0] = *(const bf16x8*)at;
      bl[0] = *(const bf16x8*)(at + LOB);
      if (NTR > 1) {
        const char* at1 = (const char*)lds + boff + ((STR + ty) * LC + tx) * (BX_PIX * 2);
        bh[1] = *(const bf16x8*)at1;
        bl[1] = *(const bf16x8*)(at1 + LOB);
      }
    }"""),
                     ("""      if (r + 1 < NTR) {
        const char* at = (const char*)lds + boff + (((r + 1) * STR + ty) * LC + tx) * (BX_PIX * 2);
        bh[r + 1] = *(const bf16x8*)at;
        bl[r + 1] = *(const bf16x8*)(at + LOB);
      }""", """      if (r + 2 < NTR) {
        const char* at = (const char*)lds + boff + (((r + 2) * STR + ty) * LC + tx) * (BX_PIX * 2);
        bh[r + 2] = *(const bf16x8*)at;
        bl[r + 2] = *(const bf16x8*)(at + LOB);
      }""")],
    # ---- k_sweep_blend (sweep.hip), HALVES = 1
    "swp_nostore": [("sweep.hip", """        if (accum) v += buf_load4(ro, ooff);
        buf_store4(ro, ooff, v);
      }
    } else {""", """        if (accum) v += buf_load4(ro, ooff);
        asm volatile("" ::"v"(v));
      }
    } else {""")],                                                                       # the flush's global stores
    "swp_noreload": [("sweep.hip", "        if (cell != -1 && cell != cc) {                            // entered another source cell",
                      "        if (cell != -1 && cell != cc && cc == -1) {")],            # taps loaded once per launch, never reloaded
    "swp_nowait": [("sweep.hip", "      wait_vmem_all();\n    }\n    const int nd_ = min(PK, d1 - dg);", "    }\n    const int nd_ = min(PK, d1 - dg);")],
    "swp_nostore_noreload": [("sweep.hip", """        if (accum) v += buf_load4(ro, ooff);
        buf_store4(ro, ooff, v);
      }
    } else {""", """        if (accum) v += buf_load4(ro, ooff);
        asm volatile("" ::"v"(v));
      }
    } else {"""), ("sweep.hip", "        if (cell != -1 && cell != cc) {                            // entered another source cell",
                      "        if (cell != -1 && cell != cc && cc == -1) {")],
    # candidate (results stay right: the compiler's own wait insertion guards every use): the explicit wait at the end of a plane only
    # when some lane of the wave reloaded in that plane -- otherwise the flush's stores keep draining behind the next planes
    "swp_condwait": [("sweep.hip", "        if (cell != -1 && cell != cc) {                            // entered another source cell\n          cc = cell;",
                      "        if (cell != -1 && cell != cc) {                            // entered another source cell\n          cc = cell; reloaded = true;"),
                     ("sweep.hip", "      // ---- phase 1: the cells of this plane, every reload issued\n", "      bool reloaded = false;\n"),
                     ("sweep.hip", "      wait_vmem_all();\n    }\n    const int nd_ = min(PK, d1 - dg);", "      if (__builtin_amdgcn_ballot_w64(reloaded) != 0) wait_vmem_all();\n    }\n    const int nd_ = min(PK, d1 - dg);")],
    # candidate: the two workgroups of a CU at different issue priorities (the one whose LDS allocation starts at 0 gets priority 3), so
    # that their MFMA phases cannot run in lock-step: kernel time = "everything else" + the matrix time of BOTH waves of a SIMD today
    "bxc_prio": [("  __bf16* lhi = lds;                       // [NPIX][BX_PIX]\n",
                  "  __bf16* lhi = lds;                       // [NPIX][BX_PIX]\n  if (__builtin_amdgcn_s_getreg((7 << 11) | 6) == 0) __builtin_amdgcn_s_setprio(3);\n")],
    "bxc_sleep": [("  __bf16* lhi = lds;                       // [NPIX][BX_PIX]\n",
                   "  __bf16* lhi = lds;                       // [NPIX][BX_PIX]\n  if ((blockIdx.x + blockIdx.y + blockIdx.z) & 1) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }\n")],
    "bxc_prio2": [("  __bf16* lhi = lds;                       // [NPIX][BX_PIX]\n",
                   "  __bf16* lhi = lds;                       // [NPIX][BX_PIX]\n  if ((blockIdx.x + blockIdx.y + blockIdx.z) & 1) __builtin_amdgcn_s_setprio(3);\n")],
    # debug: what the two workgroups of a CU read from HW_REG_LDS_ALLOC / HW_ID (written over the first output pixels' channel 0..1)
    "bxc_hwregs": [("  if (SOFTMAX) {                         // the last layer inside a stage: scores reduced over D here, never stored",
                    "  if (!SOFTMAX && threadIdx.x == 0) { float* dbg = a.out + ((size_t)n * a.ho * a.wo + (size_t)(by * gridDim.x + bx)) * a.D * 0 + ((size_t)((n * gridDim.y + by) * gridDim.x + bx)) * 4; dbg[0] = (float)__builtin_amdgcn_s_getreg((15 << 11) | 6); dbg[1] = (float)__builtin_amdgcn_s_getreg((31 << 11) | 4); return; }\n  if (SOFTMAX) {                         // the last layer inside a stage: scores reduced over D here, never stored")],
    "bxc_nobarrier": [("    __syncthreads();                     // previous chunk's readers are done\n", "    if (kb == 0) __syncthreads();\n"),
                      ("    load_w(w0h, w0l, kb, 0);\n    __syncthreads();\n", "    load_w(w0h, w0l, kb, 0);\n    if (kb == 0) __syncthreads();\n")],
}


def build(name, patches):
    tmp = tempfile.mkdtemp(prefix="adamvs_" + name + "_")
    src_dir = os.path.join(tmp, "pkg", "csrc")                  # api.hip includes ../../include/adamvs_hip.h
    shutil.copytree(B.CSRC, src_dir, ignore=shutil.ignore_patterns("_build*"))
    shutil.copytree(B.INCLUDE, os.path.join(tmp, "include"))
    for patch in patches:
        fname, old, new = patch if len(patch) == 3 else ("costreg2d_bf16x3.hip",) + tuple(patch)
        p = os.path.join(src_dir, fname)
        s = open(p).read()
        assert s.count(old) == 1, (name, old[:60], s.count(old))
        open(p, "w").write(s.replace(old, new))
    flags = [f if f != B.CSRC else src_dir for f in B.FLAGS]

    def cc(src):
        obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
        r = subprocess.run([B._hipcc()] + flags + B.SOURCE_FLAGS.get(src, []) + ["-c", os.path.join(src_dir, src), "-o", obj], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-3000:])
        return obj
    with ThreadPoolExecutor(max_workers=5) as ex:
        objs = list(ex.map(cc, B.SOURCES))
    lib = os.path.join(B.HERE, "libadamvs_hip.%s.so" % name)
    subprocess.run([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", lib] + objs, check=True)
    shutil.rmtree(tmp, ignore_errors=True)
    print("built", lib, flush=True)


if __name__ == "__main__":
    for name, patches in VARIANTS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        build(name, patches)
