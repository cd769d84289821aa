"""Settings-file readers and the text writers under the reference's module name (``VGsim.IO``).

Formats (reference ``src/IO.py:4-142``, written by ``export_settings``, pyx:1853-1907, and read by ``VGsim_cmd.py``):

``.rt``  rates: a version line, a header ``[H] B D S|SP M0 M1 ...``, then one row per haplotype in haplotype order.
         A mutation column is ``rate`` or ``rate,w0,w1,w2`` (weights of the three derived states, 1/3 each by
         default).  With an ``SP`` header the third column is a sampling *probability*: ``d*(1-sp)``, ``d*sp``.
``.su``  susceptibility: version line, header ``[H] T S0 S1 ...``, rows ``[hap] type s0 s1 ...``.
``.pp``  populations: version line, header, rows ``id size contactDensity [a] [b]`` where a lone number is the
         sampling multiplier and a triple ``x,y,z`` is (contact density at lockdown, start, end), in either order.
``.mg`` / ``.st``  square matrices (migration probabilities, immunity transition rates): version line, then rows.

Lines beginning with ``#`` after the header are skipped (the reference means to: IO.py:29-30 is a no-op ``next``).
"""
import math
import sys

from ._writers import write_mutations as writeMutations  # noqa: F401  (IO.py:144)
from ._writers import write_newick as writeGenomeNewick  # noqa: F401  (IO.py:225)


def _rows(f):
    for line in f:
        if not line.strip() or line[0] == "#":
            continue
        yield line.rstrip().split(" ")


def calculate_allele(haplotype, site, sites):  # IO.py:63-67
    allele = 0
    for _ in range(sites - site):
        allele = haplotype % 4
        haplotype = haplotype // 4
    return allele


def update_mRate(mRate):  # IO.py:53-61: the own allele gets weight 0, giving [rate, wA, wT, wC, wG]
    if math.log(len(mRate), 4) != int(math.log(len(mRate), 4)):
        print("Error!")
        sys.exit(1)
    for i in range(len(mRate)):
        for j in range(len(mRate[0])):
            mRate[i][j].insert(calculate_allele(i, j, len(mRate[0])) + 1, 0)
    return mRate


def read_rates(fn):  # IO.py:4-51
    bRate, dRate, sRate, mRate = [], [], [], []
    with open(fn) as f:
        next(f)  # version line
        header = next(f).rstrip().split(" ")
        shift = int(header[0] == "H")
        dim = len(header) - shift
        if dim < 3:
            print("At least three rates (B, D, S) are expected")
            sys.exit(1)
        as_probability = header[2 + shift] == "SP"
        for line in _rows(f):
            line = line[shift:]
            bRate.append(float(line[0]))
            if not as_probability:
                dRate.append(float(line[1]))
                sRate.append(float(line[2]))
            else:
                dRate.append(float(line[1]) * (1 - float(line[2])))
                sRate.append(float(line[1]) * float(line[2]))
            mRate.append([])
            for mut in line[3:]:
                a = mut.split(',')
                if len(a) == 1:
                    mRate[-1].append([float(a[0]), 1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0])
                elif len(a) == 4:
                    mRate[-1].append([float(a[0]), float(a[1]), float(a[2]), float(a[3])])
                else:
                    print("Error in mutations!!!")
                    sys.exit(1)
    return bRate, dRate, sRate, update_mRate(mRate)


def read_susceptibility(fn):  # IO.py:69-86 (values stay strings, as upstream)
    susceptibility, sType = [], []
    with open(fn) as f:
        next(f)
        header = next(f).rstrip().split(" ")
        shift = int(header[0] == "H")
        for line in _rows(f):
            line = line[shift:]
            susceptibility.append(line[1:])
            sType.append(int(line[0]))
    return susceptibility, sType


def read_populations(fn):  # IO.py:88-128
    sizes, contactDensity, contactAfter, startLD, endLD, samplingMultiplier = [], [], [], [], [], []
    with open(fn) as f:
        next(f)
        next(f)
        for line in _rows(f):
            sizes.append(int(line[1]))
            contactDensity.append(float(line[2]))
            if len(line) == 4:
                part = line[3].split(",")
                if len(part) == 1:
                    contactAfter.append(0)
                    startLD.append(1.0)
                    endLD.append(1.0)
                    samplingMultiplier.append(float(part[0]))
                elif len(part) == 3:
                    contactAfter.append(float(part[0]))
                    startLD.append(float(part[1]))
                    endLD.append(float(part[2]))
                    samplingMultiplier.append(1)
            elif len(line) == 5:
                part1, part2 = line[3].split(","), line[4].split(",")
                if len(part1) == 1:
                    part1, part2 = part2, part1
                if len(part1) == 3:
                    samplingMultiplier.append(float(part2[0]))
                    contactAfter.append(float(part1[0]))
                    startLD.append(float(part1[1]))
                    endLD.append(float(part1[2]))
    return sizes, contactDensity, contactAfter, startLD, endLD, samplingMultiplier


def read_matrix(fn):  # IO.py:130-142
    with open(fn) as f:
        next(f)
        return [[float(v) for v in line] for line in _rows(f)]
