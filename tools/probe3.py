import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from probe import run
for args in [(8, 64, 4096, 20000), (8, 64, 4096, 200000), (8, 64, 8192, 50000), (8, 64, 2048, 1000000)]:
    run(*args)
