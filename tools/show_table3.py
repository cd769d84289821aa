"""Prints the cells of a bench.py table3 leg (JSON file given on the command line)."""
import json
import sys

d = json.load(open(sys.argv[1]))
d = d.get("table3", d)
for k, c in d["cells"].items():
    cpu = c.get("cpu_baseline", {}).get("value", float("nan"))
    print(k, 'pub %.1fs' % c['published_s_per_1e8'],
          'single %.3g ev/s (%.0f s/1e8)' % (c['single_trajectory']['events_per_s'], c['single_trajectory']['s_per_1e8_iterations']),
          'ens R=%d %.3g ev/s %.0f ms rej %.4f x%.0f' % (c['ensemble']['replicates'], c['ensemble']['events_per_s'], c['ensemble']['kernel_ms_per_launch'],
                                               c['ensemble']['rejected_migration_share'], c['ensemble']['vs_baseline']), 'cpu %.3g' % cpu)
