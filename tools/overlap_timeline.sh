#!/bin/bash
# Timeline of one benchmark step from a rocprofv3 kernel trace: start / end of every kernel relative to the step's first kernel, with the
# queue it ran on - shows the second band block beside the first time-axis launch and the mask chain beside the second (overlapped dual path).
#   bash tools/overlap_timeline.sh TAG [ENV=VALUE ...]      -> gpurun_out/TAG_timeline.txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-train-step --no-in-flight > gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/prof_%s/*/*kernel_trace.csv" % tag)
assert f, "no kernel_trace.csv"
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# every stft_kernel starts a step: bench.py runs 64 pre-warm + 2 warm-up steps, then the 12 timed ones (later steps belong to its serial
# second context and the other sub-measurements)
starts = [i for i, r in enumerate(rows) if "stft_kernel" in r["Kernel_Name"]]
out = open("gpurun_out/%s_timeline.txt" % tag, "w")
def emit(s):
    print(s); out.write(s + "\n")
for which in (70, 71):
    i0 = starts[which]; i1 = starts[which + 1]
    t0 = int(rows[i0]["Start_Timestamp"])
    emit("step starting at kernel #%d (%d kernels): start us / end us / duration us / queue / kernel" % (i0, i1 - i0))
    for r in rows[i0:i1]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        emit("%9.1f %9.1f %8.1f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
    emit("next step starts at %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
PY
