#!/usr/bin/env python3
"""Records what the REFERENCE's settings-file readers (src/IO.py:4-142) return for the reference's own example
data files (testing/cmd_example/example.{rt,pp,mg,su,st}; copied as data to tests/golden/cmd_example/).

Usage (development container only):
    tests/golden/build_reference.sh /tmp/vgsim_ref_build
    PYTHONPATH=/tmp/vgsim_ref_build:/tmp/vgsim_ref_build/stubs python3 tests/golden/make_io_golden.py
Writes tests/golden/io_readers.json (data only).
"""
import json
import os

from VGsim import IO  # the reference build

HERE = os.path.dirname(os.path.abspath(__file__))
D = os.path.join(HERE, "cmd_example")
out = {
    "read_rates": IO.read_rates(os.path.join(D, "example.rt")),
    "read_populations": IO.read_populations(os.path.join(D, "example.pp")),
    "read_matrix_mg": IO.read_matrix(os.path.join(D, "example.mg")),
    "read_susceptibility": IO.read_susceptibility(os.path.join(D, "example.su")),
    "read_matrix_st": IO.read_matrix(os.path.join(D, "example.st")),
}
with open(os.path.join(HERE, "io_readers.json"), "w") as f:
    json.dump(out, f, indent=0)
print("wrote io_readers.json")
