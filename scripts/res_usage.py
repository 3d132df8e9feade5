#!/usr/bin/env python3
"""Print VGPR / scratch / occupancy / LDS of the kernels in one .hip file whose mangled name matches a regex.
usage: scripts/res_usage.py <file.hip> <regex>"""
import re, subprocess, sys, os
src, pat = sys.argv[1], sys.argv[2]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-c", src, "-o", "/tmp/_res.o",
                    "-I", os.path.join(os.path.dirname(os.path.abspath(src)), "../../include"),
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
for b in re.split(r"Function Name: ", r.stderr)[1:]:
    name = b.split()[0]
    if not re.search(pat, name):
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print(name[:72], "vgpr", g(r" VGPRs"), "agpr", g(r"AGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "occ", g(r"Occupancy \[waves/SIMD\]"), "lds", g(r"LDS Size \[bytes/block\]"))
