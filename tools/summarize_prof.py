#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel trace / pmc passes) into the summaries kept under profiles/.

    python tools/summarize_prof.py kernel <dir> <out.md>            # per-kernel time table from *_kernel_trace.csv
    python tools/summarize_prof.py steady <dir> <out.md> <nsteps>   # the same over the last nsteps bench steps only
    python tools/summarize_prof.py pmc <fetch_dir> <write_dir> <out.json>   # HBM bytes per launch per kernel
    python tools/summarize_prof.py share <dir> <out.md> <nsteps>    # machine time per kernel with streams overlapping: every
                                                                    #   instant is split evenly between the kernels resident then
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    base = m.group(1) if m else name.split("(")[0][-60:]
    return base


def find(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not fs:
        raise SystemExit(f"no {pat} under {d}")
    return fs


def kernel(d, out):
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    total = 0.0
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            k = short(r["Kernel_Name"])
            a = agg[k]; a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
            total += dur
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out, "w") as fo:
        fo.write("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for k, (n, t, mn, mx) in rows:
            fo.write(f"| {k} | {n} | {t:.1f} | {t / n:.2f} | {mn:.2f} | {mx:.2f} | {100 * t / total:.1f} |\n")
    print(open(out).read())


def share(d, out, nsteps):
    """who holds the machine when several streams overlap: over the last nsteps periods, every instant is divided evenly
    between the kernels resident at that instant; also the distribution of the number of resident kernels."""
    rows = []
    for f in find(d, "*kernel_trace.csv"):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "org_count_kernel" in r["Kernel_Name"]]
    if len(ends) < nsteps + 1:
        raise SystemExit("not enough steps in the trace")
    seg = rows[ends[-nsteps - 1]: ends[-1]]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["Start_Timestamp"])
    ev = []
    for r in rows:
        a, b = max(int(r["Start_Timestamp"]), t0), min(int(r["End_Timestamp"]), t1)
        if a < b:
            k = f'{short(r["Kernel_Name"])} [grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}]' if "Grid_Size_X" in r else short(r["Kernel_Name"])
            ev.append((a, 1, k)); ev.append((b, -1, k))
    ev.sort(key=lambda e: (e[0], e[1]))
    live = defaultdict(int)
    acc = defaultdict(float)
    conc = defaultdict(float)
    prev = t0
    for t, d_, k in ev:
        n = sum(live.values())
        if t > prev:
            conc[n] += t - prev
            if n:
                for kk, c in live.items():
                    if c:
                        acc[kk] += (t - prev) * c / n
        prev = t
        live[k] += d_
    tot = (t1 - t0) / 1e3
    with open(out, "w") as fo:
        fo.write(f"last {nsteps} periods, {tot:.1f} us of wall time = {tot / nsteps:.1f} us per period; resident kernels: " +
                 ", ".join(f"{n}: {100 * v / (t1 - t0):.1f} %" for n, v in sorted(conc.items())) + "\n\n")
        fo.write("| kernel [launch geometry] | share of the wall time, us per period | % |\n|---|---:|---:|\n")
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:40]:
            fo.write(f"| {k} | {v / 1e3 / nsteps:.1f} | {100 * v / (t1 - t0):.1f} |\n")
    print(open(out).read())


def steady(d, out, nsteps):
    """per-kernel table over the LAST nsteps bench steps only (a period starts with org_count_kernel), so that
    set-up work (map generation, torch kernels) does not dilute the averages.  Kernels of the same name but
    different launch geometry (ring / scan / map voxel batches) are kept apart by their grid size."""
    rows = []
    for f in find(d, "*kernel_trace.csv"):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a scan's scan-side chain starts with org_count_kernel: N + 1 of them delimit N whole periods of the steady state
    ends = [i for i, r in enumerate(rows) if "org_count_kernel" in r["Kernel_Name"]]
    if len(ends) < nsteps + 1:
        raise SystemExit("not enough steps in the trace")
    seg = rows[ends[-nsteps - 1]: ends[-1]]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in seg:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = f'{short(r["Kernel_Name"])} [grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r.get("Grid_Size_Z", "1")}, wg {r["Workgroup_Size_X"]}]'
        a = agg[k]; a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(v[1] for v in agg.values())
    # union of the kernel intervals = time with at least one kernel resident (the rest is launch gaps / idle)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > ce:
            union += ce - cs; cs, ce = a, b
        else:
            ce = max(ce, b)
    union += ce - cs
    with open(out, "w") as fo:
        fo.write(f"steady state: last {nsteps} steps, wall {1e-3 * (t1 - t0) / nsteps:.1f} us/step, kernel-busy sum {total / nsteps:.1f} us/step, "
                 f"at least one kernel resident {1e-3 * union / nsteps:.1f} us/step (streams overlap), {len(seg) / nsteps:.0f} launches/step\n\n")
        fo.write("| kernel [launch geometry] | calls/step | avg us | min us | max us | us/step | % of kernel time |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for k, (n, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fo.write(f"| {k} | {n / nsteps:.1f} | {t / n:.2f} | {mn:.2f} | {mx:.2f} | {t / nsteps:.1f} | {100 * t / total:.1f} |\n")
    print(open(out).read())


def pmc(fd, wd, out, nsteps=3):
    """HBM bytes per launch per kernel AND launch geometry, over the last nsteps bench steps (a period starts with org_count_kernel).  rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE are separate passes (TCC slots)."""
    def per_kernel(d, counter):
        rows = []
        for f in find(d, "*counter_collection.csv"):
            rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        ends = [i for i, r in enumerate(rows) if "org_count_kernel" in r["Kernel_Name"]]
        seg = rows[ends[-nsteps - 1]: ends[-1]] if len(ends) > nsteps else rows
        agg = defaultdict(lambda: [0, 0.0])
        for r in seg:
            a = agg[f'{short(r["Kernel_Name"])} [grid {r["Grid_Size"]}, wg {r["Workgroup_Size"]}]']
            a[0] += 1; a[1] += float(r["Counter_Value"])
        return {k: (v[1] / v[0], v[0] / nsteps) for k, v in agg.items() if v[0]}
    fetch = per_kernel(fd, "FETCH_SIZE")
    write = per_kernel(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0)); w, _ = write.get(k, (0.0, 0))
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of
        # the bytes of a wide coalesced streaming read → doubled; WRITE_SIZE is exact for 16-B/lane streaming stores.
        res[k] = dict(launches_per_step=nf, fetch_kib_raw=round(f, 1), write_kib_raw=round(w, 1), hbm_bytes_per_launch=round((2.0 * f + w) * 1024.0),
                      note="read side = 2 x FETCH_SIZE (gfx950 correction, calibrated for wide coalesced streams only); Infinity-Cache hits are counted")
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:20]:
        print(f"{k:70s} {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch  x{v['launches_per_step']:.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "steady":
        steady(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    elif sys.argv[1] == "share":
        share(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 3)
