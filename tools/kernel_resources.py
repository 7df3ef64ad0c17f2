#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (VGPR/SGPR/scratch/occupancy per kernel)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
rows = []
cur = None
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
    for key in ("VGPRs:", "AGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:", "SGPRs:",
                "LDS Size [bytes/block]:"):
        if key in line and cur is not None and "Function Name" not in line:
            cur[key] = line.split(key)[1].strip()
for r in rows:
    try:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        name = r["name"]
    name = name.replace("ude::", "").replace("void ", "")
    name = re.sub(r"NetCfg<IntList<([\d, ]+)>, IntList<([\d, ]+)>\s*>",
                  lambda m: "Net[%s|%s]" % (m.group(1).replace(" ", ""), m.group(2).replace(" ", "")), name)
    name = re.sub(r"\(KParams\)", "", name)
    print(name[:88].ljust(88), "VGPR", r.get("VGPRs:"), "AGPR", r.get("AGPRs:"), "SGPR", r.get("SGPRs:"), "scratch",
          r.get("ScratchSize [bytes/lane]:"), "occ", r.get("Occupancy [waves/SIMD]:"))
