"""Per-kernel sums of rocprofv3 --pmc counters: python tools/pmc_summary.py <dir>  (reads *counter_collection.csv)."""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], f) not in seen:
            seen.add((r["Dispatch_Id"], f))
            calls[k] += 1
for k, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if c.get("SQ_WAVE_CYCLES", 0) < 1e6:
        continue
    w = c["SQ_WAVE_CYCLES"]
    out = ["%-50s n=%d" % (k[-50:], calls[k])]
    for name, v in sorted(c.items()):
        if name == "SQ_WAVE_CYCLES":
            continue
        if name == "SQ_VALU_MFMA_BUSY_CYCLES":
            # MFMA-busy cycles are per SIMD, wave cycles are quad-cycles per wave: busy fraction over GRBM_GUI_ACTIVE when present
            out.append("%s=%.3g" % (name, v))
        else:
            out.append("%s/WAVE=%.3f" % (name.replace("SQ_", ""), v / w))
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        out.append("mfma_busy=%.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 1024 / 8)))
    print("  ".join(out))
