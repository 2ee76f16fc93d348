#!/usr/bin/env python3
"""The gfx950 fault of profiles/r05_miscompile.md, as a build check: a 64-bit shift (v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64) whose
32-bit shift-amount operand is the LAST vector register the wavefront was allocated gives wrong results now and then (reproducer:
tools/microbench/shift_last).  The compiler is free to emit that shape in any kernel, so every gfx950 code object of the library is
disassembled and the shape refused.  Run by __graft_entry__.build() after every build (fails the BUILD) and by the CPU suite.

    python tools/check_shift_last.py [path/to/librb_hip.so]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def shifts_with_their_amount_in_the_last_vgpr(code_object_path):
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", code_object_path], capture_output=True, text=True).stdout
    used = {}
    for blk in notes.split("- .agpr_count")[1:]:
        nm, vg = re.search(r"\.name:\s+(\S+)", blk), re.search(r"\.vgpr_count:\s+(\d+)", blk)
        if nm and vg:
            used[nm.group(1)] = int(vg.group(1))
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", code_object_path], capture_output=True, text=True).stdout
    bad, n_kernels, n_shifts, cur = [], 0, 0, None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1); n_kernels += 1
            continue
        m = re.search(r"\b(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+)\s*,", line)
        if m and cur in used:
            n_shifts += 1
            allocated = (used[cur] + 7) // 8 * 8                 # registers are handed out in blocks of 8
            if int(m.group(2)) == allocated - 1:
                bad.append((cur, line.split("//")[0].strip(), used[cur]))
    return bad, n_kernels, n_shifts


def check_library(path):
    """(offending shifts, kernels seen, shifts with a vector-register amount seen) over every gfx950 code object bundled in the library"""
    data = open(path, "rb").read()
    at, bad, kernels, shifts = 0, [], 0, 0
    while True:
        at = data.find(b"__CLANG_OFFLOAD_BUNDLE__", at)
        if at < 0:
            break
        n = struct.unpack_from("<Q", data, at + 24)[0]
        o = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o); o += 24
            triple = data[o:o + tl]; o += tl
            if b"gfx950" in triple and size:
                with tempfile.NamedTemporaryFile(suffix=".co") as f:
                    f.write(data[at + off: at + off + size]); f.flush()
                    b, k, sh = shifts_with_their_amount_in_the_last_vgpr(f.name)
                    bad += b; kernels += k; shifts += sh
        at += 24
    return bad, kernels, shifts


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rna-bloom_amd", "lib", "librb_hip.so")
    if not os.path.exists(LLVM + "llvm-objdump"):
        print("check_shift_last: no llvm-objdump here, nothing checked")
        return 0
    bad, kernels, shifts = check_library(path)
    print("check_shift_last: %d kernels, %d 64-bit shifts by a vector register, %d with the amount in the last allocated VGPR" % (kernels, shifts, len(bad)))
    for b in bad[:10]:
        print("  ", b)
    return 1 if bad or kernels < 100 else 0


if __name__ == "__main__":
    sys.exit(main())
