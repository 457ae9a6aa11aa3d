#!/usr/bin/env python3
"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into bytes per launch per kernel."""
import argparse, collections, csv, glob, json, os, sys
out = sys.argv[1]
ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--width", type=int, default=1920)
args, _ = ap.parse_known_args(sys.argv[2:])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(set))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"]][c] += float(r["Counter_Value"])
                n[r["Kernel_Name"]][c].add(r["Dispatch_Id"])
res = {}
print("%-12s %8s %16s %16s %16s" % ("kernel", "launches", "read B/launch(x2)", "write B/launch", "total B/launch"))
for k in sorted(acc):
    if not k.startswith("k_"):
        continue
    nl = max(1, len(n[k]["FETCH_SIZE"]))
    rd = acc[k]["FETCH_SIZE"] * 1024 * 2 / nl           # KiB -> B, gfx950 half-count correction
    wr = acc[k]["WRITE_SIZE"] * 1024 / max(1, len(n[k]["WRITE_SIZE"]))
    res[k] = round(rd + wr)
    print("%-12s %8d %16.0f %16.0f %16.0f" % (k, nl, rd, wr, rd + wr))
json.dump({"streams": args.streams, "frames": args.frames, "width": args.width, "bytes_per_launch": res,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) in separate passes; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); "
                   "averaged over all launches of the kernel in one bench pass (+ the profiled passes)"},
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json"), "w"), indent=1)
