"""Line-level overlap of this package's Python files with the reference's sources (development aid, run where the
reference checkout is mounted): share of significant lines (>= 25 non-blank characters, whitespace-normalised) that
also occur in a reference file, and the longest run of consecutive matched lines."""
import glob
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(line):
    return "".join(line.split())


ref = set()
for f in glob.glob(REF + "/src/*") + glob.glob(REF + "/*.py") + glob.glob(REF + "/testing/*.py"):
    if os.path.isfile(f):
        for ln in open(f, errors="ignore"):
            n = norm(ln)
            if len(n) >= 25:
                ref.add(n)
for f in sorted(glob.glob(ROOT + "/vgsim_amd/*.py") + glob.glob(ROOT + "/tests/*.py")):
    sig = [norm(ln) for ln in open(f)]
    sig = [n for n in sig if len(n) >= 25]
    if not sig:
        continue
    hit = [n in ref for n in sig]
    run = best = 0
    for h in hit:
        run = run + 1 if h else 0
        best = max(best, run)
    print("%-45s %4d lines  %5.1f %% matched  longest run %d" % (os.path.relpath(f, ROOT), len(sig), 100.0 * sum(hit) / len(sig), best))
