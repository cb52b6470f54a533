#!/usr/bin/env python3
"""Per-forward kernel breakdown from a rocprofv3 --kernel-trace CSV (tools/trace_breakdown.py <dir>)."""
import csv, glob, collections, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
seq = sorted(((r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Start_Timestamp'])) for r in rows), key=lambda x: x[2])
idx = [i for i, s in enumerate(seq) if 'k_embed_ln' in s[0]]
for which, name in ((idx[-1], 'last forward'), (idx[0], 'first forward')):
    end = [i for i, s in enumerate(seq) if 'k_pool' in s[0] and i > which][0]
    agg = collections.OrderedDict()
    per_layer_pos = collections.Counter()
    for s in seq[which:end + 1]:
        n = s[0].split('(')[0][:48]
        if 'k_proj<1' in n:   # Wo and W2 alternate
            n += ' #%d' % (per_layer_pos[n] % 2); per_layer_pos[n.rsplit(' #', 1)[0]] += 1
        if 'k_ln' in n:
            n += ' #%d' % (per_layer_pos[n] % 2); per_layer_pos[n.rsplit(' #', 1)[0]] += 1
        agg.setdefault(n, []).append(s[1])
    tot = sum(sum(v) for v in agg.values())
    print(name, 'kernel time %.2f ms' % (tot / 1e6), 'wall %.2f ms' % ((seq[end][2] + seq[end][1] - seq[which][2]) / 1e6))
    for n, v in agg.items():
        print('   %-54s n=%3d avg %8.1f us  tot %7.2f ms' % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
