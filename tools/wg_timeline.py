#!/usr/bin/env python3
"""Timeline of the compaction kernel's workgroups, from a library built with -DDPOMP_TRACE_WG (experimental build;
every workgroup records entry, start of the event loop, the moments its 4th/3rd/2nd wave emptied, its end and where it
ran).  Prints per launch: span, workgroups resident and waves live over time.

    DPOMP_HIP_LIB=build_variants/libT.so python tools/wg_timeline.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
os.environ.setdefault("DPOMP_TRACE_WG_FILE", "/tmp/dpomp_wg_trace.bin")
import dpomp_amd as dp  # noqa: E402

P = 1 << 20
pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), dp.get_observations("tests/golden/pooley.csv"))
th = np.array([[0.003], [0.1]])
for i in range(25):
    pf.loglik(th, P, seed=i)
raw = np.fromfile(os.environ["DPOMP_TRACE_WG_FILE"], dtype=np.uint64)
n, per = int(raw[0]), int(raw[1])
rec = raw[2:].reshape(n, per // 8, 8).astype(np.int64)
TICK = 0.01  # us
for k in range(n):
    r = rec[k]
    t0, t1, x3, x2, x1, t5 = (r[:, j] for j in range(6))
    base = t0.min()
    span = (t5.max() - base) * TICK
    # a wave that never emptied before the end counts as live until the end
    x3 = np.where(x3 > 0, x3, t5); x2 = np.where(x2 > 0, x2, t5); x1 = np.where(x1 > 0, x1, t5)
    nb = 12
    edges = base + np.linspace(0, t5.max() - base, nb + 1)
    res, live = [], []
    for a, b in zip(edges[:-1], edges[1:]):
        w = float(b - a)
        ov = lambda s, e: np.clip(np.minimum(e, b) - np.maximum(s, a), 0, None).sum() / w
        res.append(ov(t0, t5))
        live.append(ov(t1, x3) + ov(t1, x2) + ov(t1, x1) + ov(t1, t5))
    dur = (t5 - t0) * TICK
    print(f"launch {k}: span {span:7.1f} us, workgroup duration {dur.mean():6.1f} +- {dur.std():5.1f} (min {dur.min():.1f} max {dur.max():.1f}),"
          f" prologue {((t1 - t0) * TICK).mean():5.1f} us, last start at {((t0.max() - base) * TICK):6.1f} us,"
          f" mean resident {np.mean(res):6.0f} of 2048, mean live waves {np.mean(live):6.0f} of 8192")
    cu = ((r[:, 6] >> 32) & 0xF) * 4096 + ((r[:, 6] >> 8) & 0xFF)  # XCC, then SE / SH / CU of HW_ID
    ends = np.array([(t5[cu == c].max() - base) * TICK for c in np.unique(cu)])
    per_cu = np.array([(cu == c).sum() for c in np.unique(cu)])
    print(f"   {len(ends)} CUs: workgroups per CU {per_cu.min()}..{per_cu.max()}, CU finishes at {ends.mean():6.1f} +- {ends.std():4.1f} us"
          f" (first {ends.min():.1f}, last {ends.max():.1f}): idle CU time before the launch ends {100 * (1 - ends.mean() / span):.1f} %")
    first = (t0 - base) * TICK < 20.0  # workgroups dispatched when the launch begins
    pro = (t1 - t0)[first] * TICK
    print(f"   launch start: {first.sum()} workgroups dispatched within {((t0[first] - base) * TICK).max():.1f} us, their prologue "
          f"{pro.mean():.1f} +- {pro.std():.1f} us (min {pro.min():.1f}, max {pro.max():.1f}); first event loop entered at "
          f"{((t1 - base) * TICK).min():.1f} us, half of them by {np.median((t1[first] - base) * TICK):.1f} us")
    # time-weighted histogram, over all CUs, of the number of live waves on a CU (32 wave slots, 4 SIMDs)
    hist = np.zeros(34)
    for c in np.unique(cu):
        m = cu == c
        ev = [(t, +1) for t in t1[m]] * 1
        ev = []
        for s_, e_ in ((t1[m], x3[m]), (t1[m], x2[m]), (t1[m], x1[m]), (t1[m], t5[m])):
            ev += [(int(t), 1) for t in s_] + [(int(t), -1) for t in e_]
        ev.sort()
        cur, last = 0, int(base)
        for t, d in ev:
            hist[min(cur, 33)] += t - last
            cur, last = cur + d, t
        hist[0] += int(t5.max()) - last
    hist /= hist.sum()
    print("   share of CU time with n live waves on the CU: 0: %.3f  1-4: %.3f  5-8: %.3f  9-12: %.3f  13-16: %.3f  17+: %.3f"
          % (hist[0], hist[1:5].sum(), hist[5:9].sum(), hist[9:13].sum(), hist[13:17].sum(), hist[17:].sum()))
    print("   resident:", " ".join(f"{v:5.0f}" for v in res))
    print("   live    :", " ".join(f"{v:5.0f}" for v in live))
