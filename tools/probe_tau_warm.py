"""Development aid: bench.tau_warm_start alone (index case, 6000 tau steps of warm-up, 100 timed steps at natural occupancy):
python tools/probe_tau_warm.py [warm_steps] [timed_steps]   (VGX_TAU_DENSE_DRIFT=1: the dense drift pass)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
t = int(sys.argv[2]) if len(sys.argv) > 2 else 100
out = bench.tau_warm_start(0, warm_steps=w, timed_steps=t)
print(json.dumps({k: out[k] for k in ("start", "occupied_compartments", "infected_at_start", "steps", "ms_per_step", "wall_ms_per_step", "events_drawn", "epidemic_time_at_start")}))
