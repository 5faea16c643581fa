#!/usr/bin/env python3
"""Summarises hipcc's -Rpass-analysis=kernel-resource-usage remarks (one block per kernel) into a table:
registers, scratch bytes per lane, occupancy, LDS -- and lists every kernel that uses scratch.
usage: make -C basic_dsp_amd/csrc resources   (writes build_res.log, then calls this)  |  kernel_resources.py <log> [out]"""
import re
import subprocess
import sys

log = open(sys.argv[1]).read()
pat = re.compile(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)
rows = {}
for name, sg, vg, ag, scr, occ, lds in pat.findall(log):
    rows[name] = (int(vg), int(ag), int(scr), int(occ), int(lds))
names = sorted(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
spill = [(d, rows[n]) for n, d in zip(names, dem) if rows[n][2] > 0]
out.write("# %d kernels; %d use scratch\n" % (len(rows), len(spill)))
for d, r in spill:
    out.write("SCRATCH %3d B/lane  vgpr %3d agpr %3d occ %d  %s\n" % (r[2], r[0], r[1], r[3], d))
out.write("# vgpr agpr scratch occ lds  kernel\n")
for n, d in zip(names, dem):
    r = rows[n]
    out.write("%3d %3d %3d %d %6d  %s\n" % (r[0], r[1], r[2], r[3], r[4], d))
