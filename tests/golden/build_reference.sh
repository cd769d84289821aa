#!/usr/bin/env bash
# Builds the *reference* (Genomics-HSE/VGsim, mounted read-only at /root/reference) into a
# scratch directory OUTSIDE the repository, so that tests/golden/make_golden.py can import
# it and record golden vectors.  Nothing produced here is committed or shipped: only the
# vectors written by make_golden.py are.
#
# The reference imports a third-party module that is neither vendored nor installed:
#   mc_lib.rndm.RndmWrapper  (pyproject.toml:9,33  ->  mc_lib @ ev-br/mc_lib v0.4.1)
# This script writes a *reconstruction* of that wrapper (numpy PCG64 seeded with
# SeedSequence(entropy, spawn_key=(k,)); uniform() = bitgen.next_double).  Whether that
# reconstruction is bit-identical to upstream v0.4.1 cannot be verified offline, therefore
# the "seed -> uniform stream" mapping is PARITY-UNPINNED (oracle header);
# everything downstream of the uniform stream is pinned by the compiled reference itself.
set -euo pipefail
REF=${REF:-/root/reference}
OUT=${1:-/tmp/vgsim_ref_build}
rm -rf "$OUT"; mkdir -p "$OUT"/{mc_lib,VGsim,stubs/tskit,stubs/prettytable}
cd "$OUT"
touch mc_lib/__init__.py stubs/tskit/__init__.py
cat > stubs/prettytable/__init__.py <<'PY'
class PrettyTable:
    def __init__(self, *a, **k): self.field_names = []; self.rows = []
    def add_row(self, r): self.rows.append(r)
    def __str__(self): return "\n".join(str(r) for r in [self.field_names] + self.rows)
PY
cat > mc_lib/rndm.pxd <<'PY'
from numpy.random cimport bitgen_t
cdef class RndmWrapper():
    cdef:
        bitgen_t *rng
        object py_gen
    cdef double uniform(self) nogil
PY
cat > mc_lib/rndm.pyx <<'PY'
# cython: language_level=3
from cpython.pycapsule cimport PyCapsule_GetPointer
from numpy.random cimport bitgen_t
from numpy.random import PCG64, SeedSequence
cdef class RndmWrapper():
    def __init__(self, seed=(1234, 0), bitgen_kind=None):
        entropy, num = seed
        py_gen = (bitgen_kind or PCG64)(SeedSequence(entropy, spawn_key=(num,)))
        self.py_gen = py_gen
        self.rng = <bitgen_t *>PyCapsule_GetPointer(py_gen.capsule, "BitGenerator")
    cdef double uniform(self) nogil:
        return self.rng.next_double(self.rng.state)
PY
# the reference sources are used where they lie; the copies below live only in $OUT
cp "$REF"/src/*.pyx "$REF"/src/*.pxi "$REF"/src/*.py VGsim/
cat > setup.py <<'PY'
import os, numpy
from setuptools import setup, Extension
from Cython.Build import cythonize
npdir = os.path.dirname(numpy.__file__)
exts = [
    Extension('mc_lib.rndm', ['mc_lib/rndm.pyx'], include_dirs=[numpy.get_include()]),
    Extension('VGsim._BirthDeath', ['VGsim/_BirthDeath.pyx'], language='c++',
              include_dirs=[numpy.get_include()],
              library_dirs=[os.path.join(npdir, '_core', 'lib'), os.path.join(npdir, 'random', 'lib')],
              libraries=['npyrandom', 'npymath'], extra_compile_args=['-O2']),
]
setup(ext_modules=cythonize(exts, include_path=['.'], compiler_directives={'language_level': 3}))
PY
python3 setup.py build_ext --inplace > build.log 2>&1 || { tail -30 build.log; exit 1; }
echo "reference built in $OUT (PYTHONPATH=$OUT:$OUT/stubs)"
