"""Development aid: events/s of the exact direct path at config 3's shape against the number of occupied haplotypes per
population at the start (bench.py's spread-occupancy leg at other list lengths): python tools/sweep_occupancy.py"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

rows = []
for occ in (16, 64, 256, 1024, 4096, 16384):
    events = max(1000, min(20000, 2500 * 4096 // occ))
    r = bench.spread_leg(0, "exact", replicates=8192, events=events, occupied=occ)
    rows.append({"occupied": occ, "events_per_replicate": events, "events_per_s": r["value"], "kernel_ms": r["kernel_ms_per_launch"],
                 "algorithmic_GBps": r["roofline"]["achieved"], "chain_frac": r["chain_bound"]["frac"]})
    print(json.dumps(rows[-1]), flush=True)
