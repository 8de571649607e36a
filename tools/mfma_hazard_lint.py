#!/usr/bin/env python3
"""Check the gfx950 code of libadamvs_hip.so for reads of MFMA results that come too early on SOME control-flow path.

    python tools/mfma_hazard_lint.py [path/to/libadamvs_hip.so]      (exit code 1 when a hazard is found)

gfx950 does not interlock an MFMA result: an instruction that reads the destination registers of a
v_mfma_* must issue at least N wait states after it (10 for the 8-pass fp32 16x16x4, 7 for the 4-pass bf16 16x16x32;
the compiler fills in s_nop).  clang 22 (ROCm 7.2) was seen to count those wait states along ONE predecessor path when
the first read sits behind a branch -- `s_nop 0` where `s_nop 8` was due on the path that skips an LDS refill --
and the last tile of every workgroup silently lost its final MFMA in accumulator rows 8-15 (csrc/common.h, drain()).

What is checked, per kernel: for every instruction that reads an AGPR/VGPR range written by an MFMA (other than an
MFMA accumulating into the same registers, which the matrix pipe forwards), the minimum number of wait states over
ALL backward paths to such an MFMA; every instruction counts 1, `s_nop n` counts n + 1.  The search stops at
REQUIRED wait states, so it is cheap.  Conservative in one direction only: elapsed cycles of waits and barriers
are ignored (they can be zero).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

# wait states between an MFMA and the first read of its result, per opcode: what the compiler itself leaves in
# straight-line code of this library (fp32 16x16x4: 8 passes + 2; bf16 16x16x32: 4 passes + 3, one spare observed)
NEEDED = {"v_mfma_f32_16x16x4_f32": 10, "v_mfma_f32_16x16x32_bf16": 8}
REQUIRED = 11          # search horizon and the requirement for any other MFMA shape (8-pass XDL)
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def disassemble(lib):
    """-> {kernel name: [instruction text]} of every gfx950 code object bundled in `lib`."""
    tmp = tempfile.mkdtemp(prefix="adamvs_lint_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True,
                                 text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = kernels.setdefault(m.group(1), [])
                    continue
                if cur is None or not line.startswith("\t") or "//" not in line:
                    continue
                ins, comment = line.split("//", 1)
                m = re.match(r"\s*([0-9A-Fa-f]+):", comment)
                t = re.search(r"<.*\+0x([0-9a-fA-F]+)>\s*$", comment)
                cur.append((int(m.group(1), 16), ins.strip(), int(t.group(1), 16) if t else None))
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check_kernel(name, code):
    """code: [(address, text)].  -> list of (index, text, wait states found)."""
    n = len(code)
    addr_index = {a: i for i, (a, _, _) in enumerate(code)}
    base = code[0][0]
    preds = [[] for _ in range(n)]
    for i, (a, ins, toff) in enumerate(code):
        op = ins.split()[0]
        if op.startswith(("s_cbranch", "s_branch")):
            target = addr_index.get(base + toff) if toff is not None else None
            if target is None:
                raise RuntimeError("%s: branch target of `%s` not found" % (name, ins))
            preds[target].append(i)
        if i + 1 < n and op not in ("s_branch", "s_endpgm", "s_setpc_b64"):
            preds[i + 1].append(i)
    code = [(a, ins) for a, ins, _ in code]
    is_mfma = [ins.startswith("v_mfma") for _, ins in code]
    dst = [regs(ins.split(",")[0]) if is_mfma[i] else set() for i, (_, ins) in enumerate(code)]
    mfma_regs = set().union(*dst) if dst else set()
    problems = []
    for i, (_, ins) in enumerate(code):
        op = ins.split()[0]
        parts = ins.split(None, 1)
        if len(parts) < 2:
            continue
        operands = parts[1]
        if op.startswith(("buffer_load", "global_load", "ds_read", "s_")):
            continue
        if is_mfma[i]:
            srcs = regs(",".join(operands.split(",")[1:3]))      # A and B operands; the C operand is forwarded in the pipe
        elif op.startswith(("buffer_store", "global_store", "ds_write")):
            srcs = regs(operands)
        else:
            srcs = regs(",".join(operands.split(",")[1:])) if "," in operands else set()
        srcs &= mfma_regs
        if not srcs:
            continue
        # backward search: (instruction index, wait states accumulated, registers still looked for)
        best = None
        stack = [(p, 0, frozenset(srcs)) for p in preds[i]]
        seen = {}
        while stack:
            j, w, want = stack.pop()
            if w >= REQUIRED:
                continue
            key = (j, want)
            if key in seen and seen[key] <= w:
                continue
            seen[key] = w
            text = code[j][1]
            if is_mfma[j] and dst[j] & want:
                if w < NEEDED.get(text.split()[0], REQUIRED):
                    best = w if best is None else min(best, w)
                continue
            opj = text.split()[0]
            # a register overwritten by something else is no longer the MFMA's result
            if not is_mfma[j] and not opj.startswith(("s_", "buffer_store", "global_store", "ds_write")):
                want = want - regs(text.split(",")[0].split(None, 1)[1] if len(text.split(None, 1)) > 1 else "")
                if not want:
                    continue
            step = 1
            if opj == "s_nop":
                step = int(text.split()[1], 0) + 1
            for p in preds[j]:
                stack.append((p, w + step, want))
        if best is not None:
            problems.append((i, ins, best))
    return problems


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "ada-mvs_amd", "libadamvs_hip.so")
    kernels = disassemble(lib)
    bad = 0
    n_mfma = 0
    for name, code in sorted(kernels.items()):
        if not any(t.startswith("v_mfma") for _, t, _ in code):
            continue
        n_mfma += 1
        for i, ins, w in check_kernel(name, code):
            bad += 1
            print("HAZARD %s: instruction %d `%s` reads an MFMA result after only %d wait states on some path" % (name[:90], i, ins, w))
    print("%d kernels with MFMAs checked, %d hazards" % (n_mfma, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
