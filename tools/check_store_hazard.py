#!/usr/bin/env python3
"""Static check for the store-data hazard met in round 6 (csrc/gemm_nt.hip): a `buffer_store_dwordx3 / x4 ... sN offen` (data wider than
64 bits, SGPR offset) whose data registers are overwritten by the VERY NEXT VALU instruction -- hipcc inserts no wait state for that form and
lanes 12-15 of every 16 were seen to store the new value of the first register on gfx950 / ROCm 7.2.  Disassembles every translation unit
(hipcc -S) and reports the places; exit status 1 if any.  usage: python tools/check_store_hazard.py [file.hip ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "heal_swin_amd", "csrc")
STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0|\S+)\s")
VDST = re.compile(r"^\s*(v_\w+|ds_read\w*|ds_load\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))")


def dst_range(line):
    m = VDST.match(line)
    if not m:
        return None
    if m.group(3) is not None:
        return int(m.group(3)), int(m.group(4))
    return int(m.group(5)), int(m.group(5))


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for f in files:
        out = f"/tmp/hazard_{os.path.basename(f)}.s"
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-x", "hip", "-S",
                            "--cuda-device-only", f, "-o", out, f"-I{os.path.join(ROOT, 'include')}"], capture_output=True, text=True)
        if r.returncode:
            print(f"{f}: hipcc failed\n{r.stderr[-500:]}")
            bad += 1
            continue
        kernel, stores = "?", 0
        lines = open(out).read().split("\n")
        for i, ln in enumerate(lines):
            if ln and not ln[0].isspace() and ln.endswith(":") and not ln.startswith("."):
                kernel = ln[:-1]
            m = STORE.match(ln)
            if not m:
                continue
            soff = m.group(4).rstrip(",")
            if not re.fullmatch(r"s\d+|m0", soff):
                continue  # an inline-constant offset: the compiler knows that hazard
            stores += 1
            a, b = int(m.group(2)), int(m.group(3))
            j = i + 1
            while j < len(lines) and (not lines[j].strip() or lines[j].lstrip().startswith((";", "."))):
                j += 1
            d = dst_range(lines[j]) if j < len(lines) else None
            if d and not (d[1] < a or d[0] > b) and lines[j].lstrip().startswith("v_"):
                bad += 1
                print(f"{os.path.basename(f)}: {kernel[:90]}\n    {ln.strip()}\n    {lines[j].strip()}")
        print(f"{os.path.basename(f)}: checked")
    print(f"{bad} hazardous place(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
