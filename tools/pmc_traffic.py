#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh into profiles/<name>_hbm_traffic.json.
usage: tools/pmc_traffic.py <dir with *counter_collection.csv> <out.json>
FETCH_SIZE is doubled: on gfx950 rocprofv3 tallies the 128-byte requests of a wide coalesced read at
64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported (both in KiB units of 1024... the
tool reports KB; bench.py only needs bytes per launch)."""
import collections, csv, glob, json, os, sys
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            name = row["Kernel_Name"]
            if "bdsp::" not in name:
                continue
            short = name.split("bdsp::", 1)[1].split("(")[0]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {"note": "rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE (tools/pmc.sh), bench.py --steps 3; "
               "FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (it tallies 128-B requests at 64 B)",
       "kernels": {}}
for k, c in acc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    res["kernels"][k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_read_bytes_corrected": f * 1024 * 2,
                         "hbm_write_bytes": w * 1024, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
