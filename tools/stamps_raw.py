#!/usr/bin/env python3
"""Dump per-workgroup timing of one S288c-sized launch: when each workgroup entered, how long it
lived, which XCD / CU ran it."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
wl = sys.argv[2] if len(sys.argv) > 2 else "S288c"
if wl == "384":
    ctgs = synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)
elif wl == "Atha":
    ctgs = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
else:
    ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
tw = int(sys.argv[1]) if len(sys.argv) > 1 else 0
step = int(sys.argv[3]) if len(sys.argv) > 3 else 10
plan = engine.WavePlan(eng, ss, 100, step, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS, tile_windows=tw)
for _ in range(5):
    plan.run()
eng.sync()
eng.check(eng.lib.gams_wave_plan_set_stamps(eng.h, plan.p, 1))
plan.run()
eng.sync()
nt = 120000
buf = np.zeros(nt * 16, np.uint64)
eng.check(eng.lib.gams_wave_stamps_raw(eng.h, plan.p, buf.ctypes.data, buf.size))
st = buf.reshape(-1, 16)
st = st[st[:, 10] != 0]
t0 = st[:, 10].min()
entry = (st[:, 10] - t0) / 100.0
s0 = (st[:, 8] - t0) / 100.0
end = (st[:, 9] - t0) / 100.0
xcc = (st[:, 11] >> np.uint64(32)).astype(int)
hw = (st[:, 11] & np.uint64(0xffffffff)).astype(int)
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 1
se = (hw >> 13) & 0x7
print("workgroups", len(st), "span us", end.max())
print("entry percentiles us", np.percentile(entry, [0, 10, 50, 90, 99, 100]).round(2))
print("entry->stamp0 us", np.percentile(s0 - entry, [0, 50, 90, 100]).round(2))
print("stamp0->end us", np.percentile(end - s0, [0, 50, 90, 100]).round(2))
print("per-xcc count", np.bincount(xcc, minlength=8))
key = xcc * 1000 + se * 100 + sh * 16 + cu
uk, cnt = np.unique(key, return_counts=True)
print("distinct CUs used", len(uk), "wg per CU min/median/max", cnt.min(), np.median(cnt), cnt.max())
order = np.argsort(entry)
print("first 12 entries", entry[order][:12].round(2), "\nlast 12", entry[order][-12:].round(2))
# time line: number of resident workgroups every microsecond
for t in np.arange(0, end.max(), max(1.0, round(end.max() / 40))):
    print(f"t={t:5.1f} resident={int(((entry <= t) & (end > t)).sum())}")
# phases of the slowest workgroups (shader cycles)
dur = (st[:, 1:7].astype(np.int64) - st[:, 0:6].astype(np.int64))
tot = dur.sum(axis=1)
worst = np.argsort(tot)[-8:]
print("median phase cycles [load, (prefix), counts, z, exact, outputs]:", np.median(dur, axis=0).tolist(),
      "mean", dur.mean(axis=0).round(0).tolist(), "lifetime us median", round(float(np.median(end - entry)), 2))
print("phase cycles [load, (prefix), counts, z, exact, outputs] of the 8 slowest workgroups:")
for w in worst:
    print("  ", dur[w].tolist(), "total", int(tot[w]))
ex = dur[:, 4]
print("exact phase: median", int(np.median(ex)), "n>1000:", int((ex > 1000).sum()), "mean of those", int(ex[ex > 1000].mean()) if (ex > 1000).any() else 0)
# the workgroups that finish last: when they started, their phases, whether a wave of theirs ran the exact path
last = np.argsort(end)[-16:]
print("last 16 workgroups to finish: entry us, end us, phase cycles")
for w in last:
    print(f"   entry {entry[w]:5.2f}  end {end[w]:5.2f}  phases {dur[w].tolist()}")
print("end-time percentiles us", np.percentile(end, [50, 90, 95, 99, 100]).round(2))
# is the workgroup -> XCD map round robin?  (stamps rows are indexed by blockIdx.x)
full = buf.reshape(-1, 16)
ids = np.flatnonzero(full[:, 10] != 0)
x = (full[ids, 11] >> np.uint64(32)).astype(int)
print("blockIdx % 8 == XCC_ID for", int((ids % 8 == x).sum()), "of", ids.size, "workgroups;",
      "distinct (blockIdx % 8 -> xcc) pairs:", sorted(set(zip((ids % 8).tolist(), x.tolist())))[:16])
# which blockIdx values share a CU (first round): dispatch order within a CU
full_ids = ids
cu_key = key  # xcc*1000 + se*100 + sh*16 + cu of the workgroups in `st` (same order as ids)
first_round = full_ids < 2048
by_cu = {}
for b, k in zip(full_ids[first_round].tolist(), cu_key[first_round].tolist()):
    by_cu.setdefault(k, []).append(b)
shown = 0
for k in sorted(by_cu)[:6]:
    print("CU", k, "first-round blockIdx:", sorted(by_cu[k]))
ks = np.array([len(v) for v in by_cu.values()])
print("first-round workgroups per CU: min/median/max", ks.min(), np.median(ks), ks.max())
# is slot k = blockIdx // 256 ?  (every CU would hold exactly one workgroup of every 256-block)
ok = sum(1 for v in by_cu.values() if sorted(x // 256 for x in v) == list(range(len(v))))
print("CUs whose first-round workgroups are one per 256-block of blockIdx:", ok, "of", len(by_cu))
