"""Instruction mix per basic block of one kernel (device assembly from hipcc -S).

    python tools/asm_mix.py ada-mvs_amd/csrc/slice_red.hip k_conv_smallILi8ELi8ELi1ELi1ELi1E [min_block_size]

fp32 MFMA and the vector ALU share lanes on gfx950, so the VALU count of a loop body is what to minimise.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd.build import SOURCE_FLAGS  # noqa: E402


def main():
    src, pat = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    out = "/tmp/asm_mix.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-I",
                    os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ada-mvs_amd", "csrc"), "-S",
                    "--cuda-device-only", src, "-o", out] + SOURCE_FLAGS.get(os.path.basename(src), []), check=True,
                   stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
    print(lines[start])
    blocks, cur = [], ("entry", [])
    for l in lines[start + 1:]:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
            continue
        s = l.strip()
        if s.startswith("s_endpgm"):
            break
        if s and not s.startswith(";") and not s.startswith("."):
            cur[1].append(s.split()[0])
    blocks.append(cur)
    for name, ins in blocks:
        c = collections.Counter()
        for i in ins:
            if i.startswith("v_mfma"): k = "mfma"
            elif i.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")): k = "trans"
            elif i.startswith("v_accvgpr"): k = "accmov"
            elif i.startswith("v_"): k = "valu"
            elif i.startswith("s_waitcnt"): k = "wait"
            elif i.startswith("s_barrier"): k = "barrier"
            elif i.startswith("s_"): k = "salu"
            elif i.startswith("ds_"): k = "ds"
            elif i.startswith(("global_", "buffer_", "scratch_", "flat_")): k = "vmem"
            else: k = i
            c[k] += 1
        if len(ins) >= minsz:
            print("%-12s %4d  %s" % (name, len(ins), dict(c)))
    for l in lines:
        if pat in l and (".num_vgpr" in l or ".num_agpr" in l or "private_seg_size" in l) and l.strip().startswith(".set"):
            print(l.strip().split(".")[-1])


if __name__ == "__main__":
    main()
