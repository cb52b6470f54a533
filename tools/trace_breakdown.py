#!/usr/bin/env python3
"""Per-forward kernel breakdown from a rocprofv3 --kernel-trace CSV (tools/trace_breakdown.py <dir>): one block per distinct encoder-forward
shape found in the trace (forwards are grouped by total kernel time within 4 %; the last forward of each group is shown)."""
import csv, glob, collections, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
seq = sorted(((r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Start_Timestamp'])) for r in rows), key=lambda x: x[2])
starts = [i for i, s in enumerate(seq) if 'k_embed_ln' in s[0]]
forwards = []
for which in starts:
    end = [i for i, s in enumerate(seq) if 'k_pool' in s[0] and i > which][0]
    forwards.append((which, end, sum(s[1] for s in seq[which:end + 1])))
groups = []   # (representative total, last forward)
for f in forwards:
    for g in groups:
        if abs(f[2] - g[0]) <= 0.04 * g[0]:
            g[1] = f; g[2] += 1
            break
    else:
        groups.append([f[2], f, 1])
for tot0, (which, end, tot), count in groups:
    agg = collections.OrderedDict()
    pos = collections.Counter()
    for s in seq[which:end + 1]:
        n = s[0].split('(')[0][:48]
        if 'k_proj<1' in n or 'k_ln' in n:      # out-proj / FF2 and LN1 / LN2 alternate
            n += ' #%d' % (pos[n] % 2); pos[n.rsplit(' #', 1)[0]] += 1
        agg.setdefault(n, []).append(s[1])
    print('forward (%d in the trace): kernel time %.2f ms, wall %.2f ms' % (count, tot / 1e6, (seq[end][2] + seq[end][1] - seq[which][2]) / 1e6))
    for n, v in agg.items():
        print('   %-54s n=%3d avg %8.1f us  tot %7.2f ms' % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
