#!/bin/bash
# kernel breakdown of the fused forward next to the LayerNorm path (1000 x 32 and 1024 x 128 tokens)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c32; mkdir -p $OUT
export KIRAG_AMD_LIB=$R/tools/bin/libkirag_exp.so
for f in 0 1; do
  export KIRAG_AMD_FUSED_LN=$f
  rocprofv3 --kernel-trace -d $OUT/kt_$f -o t --output-format csv -- python3 $R/tools/one_shape.py 1000 32 8 > $OUT/log_$f.txt 2>&1 || exit 1
  echo "== fused=$f 1000 x 32" >> $OUT/breakdown.txt
  python3 - $OUT/kt_$f >> $OUT/breakdown.txt <<'PY'
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
seq = sorted(((r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Start_Timestamp'])) for r in rows), key=lambda x: x[2])
pools = [i for i, s in enumerate(seq) if 'k_pool' in s[0]]
a, b = pools[-2] + 1, pools[-1]          # the last forward
agg = collections.OrderedDict(); pos = collections.Counter()
for s in seq[a:b + 1]:
    n = s[0].split('(')[0][:60]
    if 'k_proj<1' in n or 'k_proj<4' in n or 'k_ln' in n or 'k_row_stats' in n:
        base = n; n += ' #%d' % (pos[base] % 2); pos[base] += 1
    agg.setdefault(n, []).append(s[1])
print('  kernel time %.2f ms, wall %.2f ms' % (sum(s[1] for s in seq[a:b + 1]) / 1e6, (seq[b][2] + seq[b][1] - seq[a][2]) / 1e6))
for n, v in agg.items(): print('   %-66s n=%3d avg %8.1f us  tot %7.2f ms' % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
PY
  rm -rf $OUT/kt_$f
done
cat $OUT/breakdown.txt
