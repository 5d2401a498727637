#!/bin/bash
# Developer probe (round 6): rocprofv3 PC sampling of the tile kernel.  NOT SUPPORTED on this stack: rocprofv3 answers "Given PC sampling
# configuration is not supported on any of the agents" (host_trap as well as stochastic; `rocprofv3-avail info --pc-sampling` lists no agent).
# Kept as the record of the attempt; the leave-one-part-out timings (tools/probes/skipab.sh) did the job instead.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
{ echo "== tile timeline config4"; PM_TL_WORKLOAD=config4 timeout 200 python tools/tile_timeline.py 2>&1 | grep -v "amdgpu.ids\|^  slot [0-9]* tile" | head -80; } > gpurun_out/tl4.log 2>&1
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
for wl in config4 config3; do
PM_LIB_DEV=1 PM_LIB_VARIANT=dbg timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace --output-format csv -d gpurun_out/pcs_$wl -- python bench.py --workload $wl --steps 300 --warmup 20 --no-cpu-baseline --no-config5 > gpurun_out/pcs_$wl.log 2>&1
echo "pcs $wl exit $?" >> gpurun_out/tl4.log
done
ls -laR gpurun_out/pcs_config4 | head -20 >> gpurun_out/tl4.log
# keep what comes back small: aggregate samples by instruction + comment
python - <<'PY' >> gpurun_out/tl4.log 2>&1
import glob, csv, collections, os
for wl in ("config4", "config3"):
    for f in glob.glob(f"gpurun_out/pcs_{wl}/**/*pc_sampling*.csv", recursive=True):
        c = collections.Counter(); n = 0
        with open(f) as fh:
            rd = csv.DictReader(fh)
            for row in rd:
                n += 1
                c[(row.get("Instruction", ""), row.get("Instruction_Comment", ""), row.get("Dispatch_Id", "") and "")] += 1
        out = f"gpurun_out/pcs_{wl}_agg.csv"
        with open(out, "w") as o:
            for (ins, com, _), v in c.most_common(): o.write(f"{v}\t{ins}\t{com}\n")
        print(wl, f, "samples", n, "distinct", len(c))
        os.remove(f)
PY
du -sh gpurun_out/pcs_* >> gpurun_out/tl4.log
cat gpurun_out/tl4.log | tail -120
