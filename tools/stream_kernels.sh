#!/bin/bash
# kernels of one streaming chunk (C = 2): rocprofv3 kernel trace of tools/stream_probe.py, condensed into a text table on stdout
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stream
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stream -- python3 tools/stream_probe.py 2 > gpurun_out/prof_stream.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_stream/*/*kernel_stats.csv")
assert f, "no kernel_stats.csv"
rows = list(csv.DictReader(open(f[0])))
steps = 320.0
tot_l = tot_t = 0
for r in rows:
    calls, tns = int(r["Calls"]), float(r["TotalDurationNs"])
    if calls < 100:
        continue
    print("%-72s launches/step %5.2f  us/step %7.2f  avg %7.2f us" % (r["Name"][:72], calls / steps, tns / 1e3 / steps, float(r["AverageNs"]) / 1e3))
    tot_l += calls / steps; tot_t += tns / 1e3 / steps
print("sum: %.1f launches, %.1f us of kernel time per step" % (tot_l, tot_t))
PY
tail -1 gpurun_out/prof_stream.log
