"""profiles/hbm_traffic.json from a tools/prof_round.sh PMC summary.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB,
and on gfx950 FETCH_SIZE counts 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM
section), so the read side is doubled -- an upper bound for this path, whose reads are a
mix of 4..16 B per lane.  bench.py copies the dominant kernel's figure into
roofline.traffic."""
import json
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
summ = json.load(open(src))
out = {"_source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes)",
       "_formula": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024"}
for k, d in summ.items():
    name = k.split("::")[-1]
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[name] = {"hbm_bytes_per_launch": int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024),
                     "fetch_size_kib": round(d["FETCH_SIZE"], 1), "write_size_kib": round(d["WRITE_SIZE"], 1)}
        if "SQ_INSTS_VALU" in d:  # wave-instructions per launch (SQ counters of the same summary)
            out[name]["valu_insts_per_launch"] = int(d["SQ_INSTS_VALU"])
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
