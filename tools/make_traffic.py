"""profiles/hbm_traffic.json from a tools/prof_round.sh PMC summary.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB,
and on gfx950 FETCH_SIZE counts 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM
section), so the read side is doubled -- an upper bound for this path, whose reads are a
mix of 4..16 B per lane.  bench.py copies the dominant kernel's figure into
roofline.traffic."""
import hashlib
import re
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["pm_bin.hip", "pm_bin_rows.h", "pm_fine.hip", "pm_fine_tile.h", "pm_coarse.hip", "pm_coarse_tile.h", "pm_frame.hip", "pm_frame_row.h",
                  "pm_kernels_common.h", "pm_device.h", "gfx950/pm_pin.h", "gfx950/pm_params.h"]


def _code_only(text):
    """The source without comments and white space: what the compiler sees (a reworded comment does not
    make a profile stale)."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def kernel_sources_sha16(root=ROOT):
    """What the PMC figures were measured on: a digest of the kernel sources, comments and white space
    aside (bench.py recomputes it and says so when the committed figures are older than the kernels it runs)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(root, "piet_metal_amd", "csrc", f), "r", encoding="utf-8") as fh:
            h.update(_code_only(fh.read()).encode())
    return h.hexdigest()[:16]


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    summ = json.load(open(src))
    commit = sys.argv[4] if len(sys.argv) > 4 else None  # (the commit the passes ran on, when it is not HEAD)
    if commit is None:
        try:
            commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
        except Exception:
            commit = None
    out = {"_source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes)",
           "_formula": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "_kernel_sources_sha16": kernel_sources_sha16(), "_commit": commit}
    for k, d in summ.items():
        name = k.split("::")[-1]
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            out[name] = {"hbm_bytes_per_launch": int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024),
                         "fetch_size_kib": round(d["FETCH_SIZE"], 1), "write_size_kib": round(d["WRITE_SIZE"], 1)}
            if "SQ_INSTS_VALU" in d:  # wave-instructions per launch (SQ counters of the same summary)
                out[name]["valu_insts_per_launch"] = int(d["SQ_INSTS_VALU"])
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
