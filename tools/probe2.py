import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from probe import run
for args in [(0, 1, 1, 100000), (8, 1, 1, 100000), (0, 16, 1, 50000), (0, 64, 1, 50000), (0, 256, 1, 20000), (8, 64, 1, 20000),
             (8, 64, 2048, 20000), (8, 64, 3072, 20000), (0, 1, 1024, 100000), (0, 1, 3072, 100000)]:
    run(*args)
