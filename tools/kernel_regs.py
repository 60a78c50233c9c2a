#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel in a -save-temps assembly file (hipcc ... -save-temps=obj): tools/kernel_regs.py file.s [filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
for l in s.split("\n"):
    m = re.match(r"^(_Z\S+):", l)
    if m: cur = m.group(1); vals = {}
    m = re.match(r"^; (NumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", l)
    if m and cur:
        vals[m.group(1)] = int(m.group(2))
        if m.group(1) == "LDSByteSize":
            name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip().replace("meso::", "")
            if flt in name:
                print("%-70s sgpr %3d vgpr %3d scratch %4d occ %d" % (name[:70], vals.get("NumSgprs", -1), vals.get("NumVgprs", -1), vals.get("ScratchSize", -1), vals.get("Occupancy", -1)))
            cur = None
