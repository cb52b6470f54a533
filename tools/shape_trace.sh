#!/bin/bash
# kernel-trace stats of REPS forwards of one batch shape (bash tools/shape_trace.sh B S [reps]) -> gpurun_out/shape_B_S/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$1; S=$2; N=${3:-10}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/shape_${B}_${S} -- python3 $R/tools/one_shape.py $B $S $N > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/shape_${B}_${S}/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "kr::" in r["Name"])
print(f"shape $B x $S: {tot / $N / 1e6:.2f} ms of kr:: kernels per forward")
for r in rows:
    if "kr::" in r["Name"]:
        print(f"  {r['Name'][:70]:70s} calls {int(r['Calls']) // $N:4d}/fwd  avg {float(r['AverageNs']) / 1e3:8.1f} us  {100 * float(r['TotalDurationNs']) / tot:5.1f} %")
PY
