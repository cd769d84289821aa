"""Writes vgsim_amd/csrc/vgx_logfact.h: ln(k!) for k < 126, correctly rounded (60-digit decimal arithmetic), and
vgsim_amd/csrc/vgx_tau_lf.h: ln(n!) for n <= 256 in binary32 (the tau front pass's bound)."""
import os
from decimal import Decimal, getcontext

getcontext().prec = 60
vals, f = [], 1
for k in range(126):
    if k > 0:
        f *= k
    vals.append(float(Decimal(f).ln()))
lines = ["    " + ", ".join(repr(v) for v in vals[i:i + 3]) + "," for i in range(0, 126, 3)]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgsim_amd", "csrc", "vgx_logfact.h")
open(out, "w").write(
    "// ln(k!) for k = 0..125, correctly rounded to binary64 (tools/gen_logfact.py, 60-digit decimal arithmetic): the\n"
    "// table part of numpy's logfactorial() used by its hypergeometric sampler (logfactorial.c).\n"
    "#pragma once\nstatic const double vgx_logfact_table[126] = {\n" + "\n".join(lines) + "\n};\n")

import numpy as np
f, vals32 = 1, []
for n in range(257):
    if n > 0:
        f *= n
    vals32.append(repr(float(np.float32(float(Decimal(f).ln())))) + "f")
rows = ["    " + ", ".join(vals32[i:i + 8]) for i in range(0, 257, 8)]
out = os.path.join(os.path.dirname(out), "vgx_tau_lf.h")
open(out, "w").write(
    "// ln(n!) for n = 0..256 in binary32 (tools/gen_logfact.py): the front pass's bound of a try (vgx_tau_front_kernel,\n"
    "// vgx_tau_listscan_kernel) - a table instead of 257 lgammaf() calls in every block's prologue.  The bound carries a slack of\n"
    "// 0.05, far above any rounding difference; what it lists is a superset of the failures either way.\n"
    "#pragma once\nstatic __device__ const float vgx_tau_logfact_f[257] = {\n" + ",\n".join(rows) + "\n};\n")
