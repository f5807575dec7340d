#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py uzu_amd/csrc/k_decode.hip [name-filter]"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src, "-o", "/tmp/_resource_usage.o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    keys = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"TotalSGPRs"), ("occ", r"Occupancy \[waves/SIMD\]"), ("scratch", r"ScratchSize \[bytes/lane\]"),
            ("lds", r"LDS Size \[bytes/block\]")]
    for b in blocks:
        name = b.split("\n")[0].split(" [-R")[0].strip()
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt not in dn:
            continue
        vals = []
        for label, pat in keys:
            m = re.search(pat + r": (\d+)", b)
            vals.append(f"{label} {m.group(1) if m else '?'}")
        print(f"{dn[:100]:100s} " + " ".join(vals))


if __name__ == "__main__":
    main()
