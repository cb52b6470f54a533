#!/usr/bin/env python3
"""Small-batch forward latency, wall vs kernels (python tools/small_timeline.py [trace_dir]).
Without an argument: back-to-back and one-at-a-time wall time per forward of the reference's real batch shapes (KiRAG hop 1-2 x 256, e5.py helpers 4 x 64,
compute_corpus_embeddings 8 x 128, one query 1 x 32, a triple batch 125 x 32).  With a rocprofv3 --kernel-trace output directory: per shape (forwards are
delimited by k_seq_len / k_pack_small .. k_pool) the kernel-time sum, the first-start-to-last-end span and the idle time between kernels of the LAST forward of each group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = ((1, 32), (1, 256), (2, 256), (4, 64), (8, 128), (125, 32))
if len(sys.argv) > 1:
    import csv, glob
    rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
    seq = sorted(((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows), key=lambda x: x[1])
    starts = [i for i, s in enumerate(seq) if 'k_seq_len' in s[0] or 'k_pack_small' in s[0]]
    fw = []
    for a in starts:
        ends = [i for i, s in enumerate(seq) if 'k_pool' in s[0] and i > a]
        if ends: fw.append((a, ends[0]))
    groups = {}
    for a, b in fw:
        n = b - a + 1
        ksum = sum(s[2] - s[1] for s in seq[a:b + 1]); span = seq[b][2] - seq[a][1]
        groups.setdefault((n, round(ksum / 2e4)), []).append((ksum, span, a, b))
    for key, v in groups.items():
        ksum, span, a, b = v[-1]
        gaps = [seq[i + 1][1] - seq[i][2] for i in range(a, b)]
        print(f"forward of {key[0]} launches ({len(v)} in trace): kernels {ksum / 1e3:.1f} us, span {span / 1e3:.1f} us, idle {(span - ksum) / 1e3:.1f} us "
              f"(mean gap {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us, mean kernel {ksum / key[0] / 1e3:.2f} us)")
    sys.exit(0)
import numpy as np, torch
from kirag_amd import bench_support as BS
dev = torch.device("cuda:0")
enc = BS.make_hip_encoder(dev)
reps = int(os.environ.get("REPS", "40"))
for B, S in SHAPES:
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    for _ in range(5): enc.forward(ids, mask, 0)
    torch.cuda.synchronize()
    b2b, one = [], []
    for rnd in range(5):
        t0 = time.perf_counter()
        for _ in range(reps): enc.forward(ids, mask, 0)
        torch.cuda.synchronize(); b2b.append((time.perf_counter() - t0) / reps * 1e3)
        t0 = time.perf_counter()
        for _ in range(reps): enc.forward(ids, mask, 0); torch.cuda.synchronize()
        one.append((time.perf_counter() - t0) / reps * 1e3)
    # host-side enqueue cost alone: time to return from forward() calls (queue kept short by a sync every 4)
    t0 = time.perf_counter()
    for i in range(reps): enc.forward(ids, mask, 0)
    t_enq = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    print(f"{B:4d} x {S:3d}: back-to-back {np.median(b2b):.3f} ms, one at a time {np.median(one):.3f} ms, host enqueue {t_enq:.3f} ms per forward", flush=True)
