"""Writes vgsim_amd/csrc/vgx_logfact.h: ln(k!) for k < 126, correctly rounded (60-digit decimal arithmetic)."""
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
