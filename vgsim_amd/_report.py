"""Printouts, settings files and timeline logs of the model object (host side, no device work).

Mirrors the reference's ``print_*`` family (``src/_BirthDeath.pyx:1003-1184``), ``Debug`` (pyx:2070-2282),
``PrintPropensities`` (pyx:2615-2649), ``export_settings`` (pyx:1853-1907) and ``output_epidemiology_timelines``
(pyx:1765-1847).  The rate caches the reference keeps on its object live on the device here, so the two debugging
printouts recompute them on the host from the current state with the formulas of ``UpdateAllRates`` (pyx:279-351)
and ``Propensities`` (pyx:2351-2417); that is a display aid (vectorised, summation order free), not the hot path.

Upstream defects that are not reproduced (each would raise before doing anything): ``export_settings`` calls an
undefined ``calculate_string`` (pyx:1870), ``output_epidemiology_timelines`` reads an undefined ``susceptible_num``
(pyx:1767), ``Simulator.export_state`` calls undefined ``output_*`` methods (if:580-581).  ``set_settings`` is an
empty placeholder upstream (pyx:1721-1722); here it reads back what ``export_settings`` wrote.
"""
import os

import numpy as np

BIRTH, DEATH, SAMPLING, MUTATION, SUSCCHANGE, MIGRATION, MULTITYPE = range(7)  # events.pxi:2-8

try:  # the reference prints through prettytable; it is optional here
    from prettytable import PrettyTable
except ImportError:  # pragma: no cover - depends on the image
    class PrettyTable:
        """Minimal stand-in with prettytable's default look (centred cells, +---+ borders, multi-line cells)."""

        def __init__(self):
            self.field_names = []
            self._rows = []

        def add_row(self, row):
            self._rows.append(list(row))

        def __str__(self):
            import re
            ansi = re.compile(r"\033\[[0-9;]*m")
            cells = [[str(c).split("\n") for c in row] for row in [self.field_names] + self._rows]
            ncol = len(self.field_names)
            width = [max((len(ansi.sub("", ln)) for row in cells for ln in row[k]), default=0) for k in range(ncol)]
            bar = "+" + "+".join("-" * (w + 2) for w in width) + "+"

            def fmt(row):
                lines = []
                for i in range(max(len(c) for c in row)):
                    parts = []
                    for k in range(ncol):
                        s = row[k][i] if i < len(row[k]) else ""
                        pad = width[k] - len(ansi.sub("", s))
                        parts.append(" " + " " * (pad // 2) + s + " " * (pad - pad // 2) + " ")
                    lines.append("|" + "|".join(parts) + "|")
                return lines
            out = [bar] + fmt(cells[0]) + [bar]
            for row in cells[1:]:
                out += fmt(row)
            return "\n".join(out + [bar])


def host_rates(m):
    """Rate caches of ``UpdateAllRates`` (pyx:279-351) recomputed from the model's current parameters and state."""
    P, H, S = m.popNum, m.hapNum, m.susNum
    mig = m.migrationRates.copy()
    np.fill_diagonal(mig, 0.0)
    np.fill_diagonal(mig, 1.0 - mig.sum(axis=1))
    sizes = m.sizes.astype(float)
    actual = mig.T @ sizes
    cd = m.contactDensity
    r = {"migrationRates": mig, "actualSizes": actual}
    r["suscepCumulTransition"] = m.suscepTransition.sum(axis=1)
    r["tmRate"] = m.mRate.sum(axis=1) if m.sites else np.zeros(H)
    w = mig * mig * (cd / actual)[None, :]                      # [pi, pn]
    # BirthRate (pyx:382-392): b[h] * sum_sn sum_pn S[pi,sn]*sigma[h,sn]*m[pi,pn]^2*cd[pn]/as[pn]
    sus = m.susceptible.astype(float)
    r["susceptHapPopRate"] = sus[:, None, :] * m.susceptibility[None, :, :] * w.sum(axis=1)[:, None, None]
    ev = np.zeros((P, H, 4))
    ev[:, :, 0] = m.bRate[None, :] * r["susceptHapPopRate"].sum(axis=2)
    ev[:, :, 1] = m.dRate[None, :]
    ev[:, :, 2] = m.sRate[None, :] * m.samplingMultiplier[:, None]
    ev[:, :, 3] = r["tmRate"][None, :]
    r["eventHapPopRate"] = ev
    r["tEventHapPopRate"] = ev.sum(axis=2)
    r["hapPopRate"] = r["tEventHapPopRate"] * m.infectious
    r["infectPopRate"] = r["hapPopRate"].sum(axis=1)
    r["immuneSourcePopRate"] = sus * r["suscepCumulTransition"][None, :]
    r["immunePopRate"] = r["immuneSourcePopRate"].sum(axis=1)
    r["popRate"] = r["infectPopRate"] + r["immunePopRate"]
    r["totalRate"] = float(r["popRate"].sum())
    eff = np.einsum("ak,bk,k->ab", mig, mig, cd / actual)       # pyx:327-338
    np.fill_diagonal(eff, 0.0)
    r["effectiveMigration"] = eff
    max_birth = float((m.bRate[:, None] * m.susceptibility).max()) if H else 0.0
    r["maxEffectiveBirthMigration"] = eff.max(axis=0) * max_birth
    ti, ts = m.infectious.sum(axis=1), m.susceptible.sum(axis=1)
    r["migPopRate"] = r["maxEffectiveBirthMigration"] * ts * (ti.sum() - ti)
    r["totalMigrationRate"] = float(r["migPopRate"].sum())
    return r


def host_propensities(m):
    """Tau-leaping channel propensities (pyx:2351-2417) of the model's current state, as dense arrays."""
    r = host_rates(m)
    mig, actual, cd = r["migrationRates"], r["actualSizes"], m.contactDensity
    Sx, I = m.susceptible.astype(float), m.infectious.astype(float)
    out = {}
    # [spn, tpn, sn, hn]  (pyx:2366-2367)
    out["PropensitiesMigr"] = (r["effectiveMigration"].T[:, :, None, None] * Sx[None, :, :, None] * I[:, None, None, :]
                               * (m.bRate[None, :] * m.susceptibility.T)[None, None, :, :]
                               * np.diag(mig)[:, None, None, None])
    out["PropensitiesSuscep"] = m.suscepTransition[None, :, :] * Sx[:, :, None]
    out["PropensitiesRecovery"] = m.dRate[None, :] * I
    out["PropensitiesSampling"] = m.sRate[None, :] * I * m.samplingMultiplier[:, None]
    if m.sites:
        wsum = m.hapMutType.sum(axis=2)
        out["PropensitiesMutatations"] = (m.mRate[:, :, None] * m.hapMutType / wsum[:, :, None])[None] * I[:, :, None, None]
    else:
        out["PropensitiesMutatations"] = np.zeros((m.popNum, m.hapNum, 0, 3))
    coef = (mig * mig * (cd / actual)[None, :]).sum(axis=1)      # [tpn]
    out["PropensitiesTransmission"] = (m.bRate[None, :, None] * m.susceptibility[None, :, :] * coef[:, None, None]
                                       * Sx[:, None, :] * I[:, :, None])
    return out


class Reporting:
    """Mixed into ``BirthDeathModel``."""

    # ------------------------------------------------------------------ tables (pyx:1003-1174)
    def calculate_colored_haplotype(self, haplotype, site):  # pyx:1220-1241
        hap = self.calculate_string_from_haplotype(haplotype)
        variants = [hap[:site] + a + hap[site + 1:] for a in "ATCG"]
        variants.remove(hap)
        variants.append(hap)

        def colour(h):
            return h[:site] + "\033[31m{}\033[0m".format(h[site:site + 1]) + h[site + 1:]
        return "".join(colour(hap) + "->" + colour(variants[i]) + ": " + str(self.hapMutType[haplotype, site, i]) + "\n"
                       for i in range(3))

    def print_basic_parameters(self):
        table = PrettyTable()
        field = ["H", "TR", "RR", "SR", "ST"]
        for s in range(self.sites):
            field += [f"M{s}", f"MW{s}"]
        table.field_names = field
        for hn in range(self.hapNum):
            row = ["\n" + self.calculate_string_from_haplotype(hn), f"\n{self.bRate[hn]}", f"\n{self.dRate[hn]}",
                   f"\n{self.sRate[hn]}", f"\n{self.suscType[hn]}"]
            for s in range(self.sites):
                row += [f"\n{self.mRate[hn, s]}", self.calculate_colored_haplotype(hn, s)]
            table.add_row(row)
        print(table)
        print("Legend:")
        print("H - haplotype")
        print("TR - transmission rate")
        print("RR - recovery rate")
        print("SR - sampling rate")
        print("ST - susceptibility type")
        for s in range(self.sites):
            print(f"M{s} - {s} mutation rate")
            print(f"MW{s} - {s} mutation weights")
        print()

    def GetCurrentIndividuals(self):
        """pyx:1030-1059: compartments rebuilt by replaying the event log from the initial state (vectorised; like
        upstream, MULTITYPE rows fall into the last branch)."""
        sus = self.initial_susceptible.copy()
        inf = self.initial_infectious.copy()
        n = self.events.ptr
        t = self.events.types[:n]
        hap, pop = self.events.haplotypes[:n], self.events.populations[:n]
        nh, npop = self.events.newHaplotypes[:n], self.events.newPopulations[:n]
        for code, ds, di in ((BIRTH, -1, +1), (DEATH, +1, -1), (SAMPLING, +1, -1)):
            k = t == code
            np.add.at(sus, (pop[k], nh[k]), ds)
            np.add.at(inf, (pop[k], hap[k]), di)
        k = t == MUTATION
        np.add.at(inf, (pop[k], hap[k]), -1)
        np.add.at(inf, (pop[k], nh[k]), +1)
        k = t == SUSCCHANGE
        np.add.at(sus, (pop[k], hap[k]), -1)
        np.add.at(sus, (pop[k], nh[k]), +1)
        k = t == MIGRATION
        np.add.at(sus, (npop[k], nh[k]), -1)
        np.add.at(inf, (npop[k], hap[k]), +1)
        return [sus.tolist(), inf.tolist()]

    def print_populations(self, population, susceptibles, infectious, migration):
        if susceptibles or infectious:
            current_susceptible, current_infectious = self.GetCurrentIndividuals()
        if population:
            self._compute_actual_sizes()
            table = PrettyTable()
            table.field_names = ["ID", "Size", 'Actual size', "CD", 'CDBLC', "CDALD", "SLD", "ELD", "SM"]
            for pn in range(self.popNum):
                table.add_row([pn, self.sizes[pn], self.actualSizes[pn], self.contactDensity[pn],
                               self.contactDensityBeforeLockdown[pn], self.contactDensityAfterLockdown[pn],
                               self.startLD[pn], self.endLD[pn], self.samplingMultiplier[pn]])
            print(table)
            print("Legend:")
            print("ID - number of population")
            print("Size - size of population")
            print("Actual size - actual size of population")
            print("CD - contact density")
            print("CDBLD - contact density without lockdown")
            print("CDALD - contact density at lockdown")
            print("SLD - start of lockdown")
            print("ELD - end of lockdown")
            print("SM - sampling multiplier")
            print()
        if susceptibles:
            table = PrettyTable()
            table.field_names = ["ST\\ID"] + list(range(self.popNum))
            for sn in range(self.susNum):
                table.add_row([sn] + [current_susceptible[pn][sn] for pn in range(self.popNum)])
            print(table)
            print("Legend:")
            print("ID - ID population")
            print("ST - susceptibility type")
            print()
        if infectious:
            table = PrettyTable()
            table.field_names = ["H\\ID"] + list(range(self.popNum))
            for hn in range(self.hapNum):
                table.add_row([self.calculate_string_from_haplotype(hn)] +
                              [current_infectious[pn][hn] for pn in range(self.popNum)])
            print(table)
            print("Legend:")
            print("ID - ID population")
            print("H - haplotype")
            print()
        if migration:
            table = PrettyTable()
            table.field_names = ["S\\T"] + list(range(self.popNum))
            for pn1 in range(self.popNum):
                table.add_row([pn1] + [self.migrationRates[pn1, pn2] for pn2 in range(self.popNum)])
            print(table)
            print("Legend:")
            print("S - ID source population")
            print("T - ID target population")
            print()

    def print_immunity_model(self, immunity, transition):
        if immunity:
            table = PrettyTable()
            table.field_names = ["H\\ST"] + [f"S{sn}" for sn in range(self.susNum)]
            for hn in range(self.hapNum):
                table.add_row([self.calculate_string_from_haplotype(hn)] +
                              [self.susceptibility[hn, sn] for sn in range(self.susNum)])
            print(table)
            print("Legend:")
            print("H - haplotype")
            print("ST - susceptibility type")
            print()
        if transition:
            table = PrettyTable()
            table.field_names = ["ID"] + list(range(self.susNum))
            for sn1 in range(self.susNum):
                table.add_row([sn1] + [self.suscepTransition[sn1, sn2] for sn2 in range(self.susNum)])
            print(table)
            print("Legend:")
            print("ID - ID susceptibility type")
            print()

    def print_chain(self):
        """``Simulator.print_chain`` (if:842-843) has no counterpart on the reference's model; prints the event log."""
        print("time\ttype\thaplotype\tpopulation\tnewHaplotype\tnewPopulation")
        ev = self.events
        for i in range(ev.ptr):
            print(ev.times[i], ev.types[i], ev.haplotypes[i], ev.populations[i], ev.newHaplotypes[i], ev.newPopulations[i],
                  sep="\t")

    def print_tree(self):
        """``Simulator.print_tree`` (if:845-846): parent array and node times of the genealogy."""
        self._need_tree()
        print("node\tparent\ttime\tpopulation")
        for i in range(len(self.tree)):
            print(i, self.tree[i], self.times[i], self.tree_pop[i], sep="\t")

    def print_recomb(self, left, right):  # pyx:607-612
        for i in range(left, right):
            print('hi(', self.calculate_string_from_haplotype(self.rec.his[i]), ') = ', self.rec.his[i],
                  ', hi2(', self.calculate_string_from_haplotype(self.rec.hi2s[i]), ') = ', self.rec.hi2s[i],
                  ', nhi(', self.calculate_string_from_haplotype(self.rec.nhis[i]), ') = ', self.rec.nhis[i],
                  ', pos = ', self.rec.posRecombs[i], sep='')

    # ------------------------------------------------------------------ Debug / PrintPropensities
    def Debug(self):
        """pyx:2070-2282: scalars, parameters and (recomputed) rate caches."""
        r = host_rates(self)

        def row(label, values):
            print(label, end="")
            for v in values:
                print(v, end=" ")
            print()

        def block(label, a):
            print(label)
            a = np.asarray(a)
            for i in range(a.shape[0]):
                if a.ndim == 2:
                    row("", a[i])
                else:
                    for j in range(a.shape[1]):
                        row("", a[i, j])
                    print()
            print()
        # scalar dump of pyx:2071-2094 (labels are the reference's output text); None = blank line, r[...] = recomputed cache
        print("Parameters")
        scalars = (("first_simulation", "mutable", ": "), ("sampling_probability", "const", ": "),
                   ("memory_optimization", "const", ": "), None, ("sites", "const", ": "), ("hapNum", "const", ": "),
                   ("currentHapNum", "mutable", ": "), ("maxHapNum", "mutable", ": "), ("popNum", "const", ": "),
                   ("susNum", "const", ": "), ("bCounter", "mutable", ": "), ("dCounter", "mutable", ": "),
                   ("sCounter", "mutable", ": "), ("mCounter", "mutable", ": "), ("iCounter", "mutable", ":"),
                   ("swapLockdown", "mutable", ": "), ("migPlus", "mutable", ": "), ("migNonPlus", "mutable", ": "),
                   ("globalInfectious", "mutable", ": "), None, ("currentTime", "mutable", ": "))
        for item in scalars:
            if item is None:
                print()
            else:
                name, kind, sep = item
                print("%s(%s)%s" % (name, kind, sep), getattr(self, name))
        for label, key in (("totalRate", "totalRate"), ("totalMigrationTate", "totalMigrationRate")):   # (upstream's spelling)
            print("%s(mutable): " % label, r[key])
        print()
        row("suscType(const): ", self.suscType)
        print()
        row("hapToNum(mutable): ", self.hapToNum)
        row("numToHap(mutable): ", self.numToHap)
        row("sizes(const): ", self.sizes)
        row("totalSusceptible(mutable): ", self.totalSusceptible)
        row("totalInfectious(mutable): ", self.totalInfectious)
        row("lockdownON(mutable): ", self.lockdownON)
        print()
        block("susceptible(mutable)----", self.susceptible)
        block("infectious(mutable)----", self.infectious)
        row("Birth rate(const): ", self.bRate)
        row("Death rate(const): ", self.dRate)
        row("Sampling rate(const): ", self.sRate)
        row("tmRate(const): ", r["tmRate"])
        row("maxEffectiveBirthMigration(const): ", r["maxEffectiveBirthMigration"])
        row("suscepCumulTransition(const): ", r["suscepCumulTransition"])
        row("immunePopRate(mutable): ", r["immunePopRate"])
        row("infectPopRate(mutable): ", r["infectPopRate"])
        row("popRate(mutable): ", r["popRate"])
        row("migPopRate(mutable): ", r["migPopRate"])
        row("actualSizes(const): ", r["actualSizes"])
        row("contactDensity(const): ", self.contactDensity)
        row("contactDensityBeforeLockdown(const): ", self.contactDensityBeforeLockdown)
        row("contactDensityAfterLockdown(const): ", self.contactDensityAfterLockdown)
        row("startLD(const): ", self.startLD)
        row("endLD(const): ", self.endLD)
        row("samplingMultiplier(const): ", self.samplingMultiplier)
        print()
        block("mRate(const)----", self.mRate)
        block("susceptibility(const)----", self.susceptibility)
        block("tEventHapPopRate(mutable)----", r["tEventHapPopRate"])
        block("suscepTransition(const)----", self.suscepTransition)
        block("immuneSourcePopRate(mutable)----", r["immuneSourcePopRate"])
        block("hapPopRate(mutable)----", r["hapPopRate"])
        block("migrationRates(const)----", r["migrationRates"])
        block("effectiveMigration(const)----", r["effectiveMigration"])
        block("hapMutType(const)----", self.hapMutType)
        block("eventHapPopRate(mutable)----", r["eventHapPopRate"])
        block("susceptHapPopRate(mutable)----", r["susceptHapPopRate"])

    def PrintPropensities(self):  # pyx:2615-2649
        p = host_propensities(self)
        print("Migrations")
        for s in range(self.popNum):
            for r in range(self.popNum):
                if s == r:
                    continue
                for i in range(self.susNum):
                    for h in range(self.hapNum):
                        print(s, r, i, h, p["PropensitiesMigr"][s, r, i, h])
        for s in range(self.popNum):
            print("Susceptibility transition")
            for i in range(self.susNum):
                for j in range(self.susNum):
                    if i == j:
                        continue
                    print(s, i, j, p["PropensitiesSuscep"][s, i, j])
            for h in range(self.hapNum):
                print("Recovery ", s, h, self.suscType[h], p["PropensitiesRecovery"][s, h])
                print("Sampling ", s, h, self.suscType[h], p["PropensitiesSampling"][s, h])
                for site in range(self.sites):
                    for i in range(3):
                        print("Mutation", s, h, site, i, p["PropensitiesMutatations"][s, h, site, i])
                for i in range(self.susNum):
                    print("Transmission", s, h, i, p["PropensitiesTransmission"][s, h, i])

    # ------------------------------------------------------------------ settings files (pyx:1853-1907, IO.py:4-142)
    def export_settings(self, file_template):
        """Writes ``<t>/<t>.rt .pp .mg .su .st`` exactly in the layout of pyx:1853-1907 and prints the command line
        that would load them.  (The working directory is left untouched; upstream ``chdir``s in and out.)"""
        if not os.path.isdir(file_template):
            os.mkdir(file_template)
        base = os.path.join(file_template, file_template)
        comand = 'Command line command: '
        with open(base + ".rt", "w") as file:
            file.write("#Rates_format_version 0.0.1\nH B D S")
            for s in range(self.sites):
                file.write(" M" + str(s))
            file.write("\n")
            for hn in range(self.hapNum):
                file.write(self.calculate_string_from_haplotype(hn) + " " + str(self.bRate[hn]) + " " + str(self.dRate[hn]) +
                           " " + str(self.sRate[hn]) + ' ')
                for s in range(self.sites):
                    file.write(str(self.mRate[hn, s]) + "," + str(self.hapMutType[hn, s, 0]) + "," +
                               str(self.hapMutType[hn, s, 1]) + "," + str(self.hapMutType[hn, s, 2]) + " ")
                file.write("\n")
            comand += (file_template + '/' + file_template + '.rt ')
        with open(base + ".pp", "w") as file:
            file.write("#Population_format_version 0.0.1\nid size contactDensity conDenAfterLD startLD endLD samplingMulriplier\n")
            for pn in range(self.popNum):
                file.write(str(pn) + " " + str(self.sizes[pn]) + " " + str(self.contactDensity[pn]) + " " +
                           str(self.contactDensityAfterLockdown[pn]) + "," + str(self.startLD[pn]) + "," + str(self.endLD[pn]) +
                           " " + str(self.samplingMultiplier[pn]) + "\n")
            comand += ('-pm ' + file_template + '/' + file_template + '.pp ')
        with open(base + ".mg", "w") as file:
            file.write("#Migration_format_version 0.0.1\n")
            for pn1 in range(self.popNum):
                for pn2 in range(self.popNum):
                    file.write(str(self.migrationRates[pn1, pn2]) + " ")
                file.write("\n")
            comand += (file_template + '/' + file_template + '.mg ')
        with open(base + ".su", "w") as file:
            file.write("#Susceptibility_format_version 0.0.1\nH T")
            for sn in range(self.susNum):
                file.write(" S" + str(sn))
            file.write("\n")
            for hn in range(self.hapNum):
                file.write(self.calculate_string_from_haplotype(hn) + " " + str(self.suscType[hn]))
                for sn in range(self.susNum):
                    file.write(" " + str(self.susceptibility[hn, sn]))
                file.write("\n")
            comand += ('-su ' + file_template + '/' + file_template + '.su ')
        with open(base + ".st", "w") as file:
            file.write("#Susceptibility_format_version 0.0.1\n")
            for sn1 in range(self.susNum):
                for sn2 in range(self.susNum):
                    file.write(str(self.suscepTransition[sn1, sn2]) + " ")
                file.write("\n")
            comand += ('-st ' + file_template + '/' + file_template + '.st ')
        print(comand)

    def set_settings(self, file_template):
        """Loads ``<t>/<t>.rt .pp .mg .su .st`` (the files ``export_settings`` writes) into this model through the
        ordinary setters, the way ``VGsim_cmd.py:113-142`` applies them.  The model's dimensions must match the files."""
        from . import IO
        base = os.path.join(file_template, file_template)
        bRate, dRate, sRate, mRate = IO.read_rates(base + ".rt")
        sizes, contactDensity, contactAfter, startLD, endLD, samplingMultiplier = IO.read_populations(base + ".pp")
        migrationRates = IO.read_matrix(base + ".mg")
        susceptible, susType = IO.read_susceptibility(base + ".su")
        suscepTransition = IO.read_matrix(base + ".st")
        if len(bRate) != self.hapNum or len(sizes) != self.popNum or len(susceptible[0]) != self.susNum:
            raise ValueError('Incorrect settings files: %d haplotypes, %d populations, %d susceptible groups; the model '
                             'has %d, %d, %d.' % (len(bRate), len(sizes), len(susceptible[0]), self.hapNum, self.popNum,
                                                  self.susNum))
        apply_settings(self, bRate, dRate, sRate, mRate, sizes, contactDensity, contactAfter, startLD, endLD,
                       samplingMultiplier, migrationRates, susceptible, susType, suscepTransition)

    # ------------------------------------------------------------------ timelines (pyx:1765-1847)
    def output_epidemiology_timelines(self, step_num, output_file):
        """Compartment sizes of every population at ``step_num + 1`` time points, rebuilt from the event log; a dict
        ``{"time": [...], "P0": {"S0": [...], "H0": [...]}, ...}`` or, with ``output_file``, ``logs/PID<p>.log``.
        Follows pyx:1765-1847 as written: start state = everybody in group 0 with one index case of haplotype 0 in
        population 0; a row is emitted after the first event at or past each time point (at most one row per event);
        MULTITYPE events are skipped (upstream TODO)."""
        P, S, H = self.popNum, self.susNum, self.hapNum
        time_points = [i * self.currentTime / step_num for i in range(step_num + 1)]
        suscepDate = np.zeros((P, S), dtype=np.int64)
        hapDate = np.zeros((P, H), dtype=np.int64)
        suscepDate[:, 0] = self.sizes
        hapDate[0, 0] += 1
        suscepDate[0, 0] -= 1
        ev = self.events
        rows_t, rows_s, rows_h = [], [], []
        point = 0
        for j in range(ev.ptr):
            t, hap, pop, nh, npop = ev.types[j], ev.haplotypes[j], ev.populations[j], ev.newHaplotypes[j], ev.newPopulations[j]
            if t == BIRTH:
                hapDate[pop, hap] += 1
                suscepDate[pop, nh] -= 1
            elif t == DEATH or t == SAMPLING:
                hapDate[pop, hap] -= 1
                suscepDate[pop, nh] += 1
            elif t == MUTATION:
                hapDate[pop, hap] -= 1
                hapDate[pop, nh] += 1
            elif t == SUSCCHANGE:
                suscepDate[pop, hap] -= 1
                suscepDate[pop, nh] += 1
            elif t == MIGRATION:
                suscepDate[npop, nh] -= 1
                hapDate[npop, hap] += 1
            if point <= step_num and time_points[point] <= ev.times[j]:
                rows_t.append(time_points[point])
                rows_s.append(suscepDate.copy())
                rows_h.append(hapDate.copy())
                point += 1
        if output_file == True:  # noqa: E712 (reference semantics)
            if not os.path.isdir("logs"):
                os.mkdir("logs")
            for i in range(P):
                with open('logs/PID' + str(i) + '.log', 'w') as f:
                    f.write("time" + "".join(" S" + str(sn) for sn in range(S)) + "".join(" H" + str(hn) for hn in range(H)) + "\n")
                    for k in range(len(rows_t)):
                        f.write(str(rows_t[k]) + " " + "".join(str(v) + " " for v in rows_s[k][i]) +
                                "".join(str(v) + " " for v in rows_h[k][i]) + "\n")
            return None
        log = {"time": rows_t}
        for i in range(P):
            log["P" + str(i)] = {}
            for j in range(S):
                log["P" + str(i)]["S" + str(j)] = [r[i, j] for r in rows_s]
            for j in range(H):
                log["P" + str(i)]["H" + str(j)] = [r[i, j] for r in rows_h]
        return log


def apply_settings(sim, bRate, dRate, sRate, mRate, sizes, contactDensity, contactAfter, startLD, endLD,
                   samplingMultiplier, migrationRates, susceptible, susType, suscepTransition):
    """The setter sequence of ``VGsim_cmd.py:113-142`` on a ``Simulator`` or a ``BirthDeathModel`` (its setters
    take the same leading arguments; the optional ones are passed explicitly)."""
    for i in range(len(bRate)):
        sim.set_transmission_rate(bRate[i], i)
        sim.set_recovery_rate(dRate[i], i)
        sim.set_sampling_rate(sRate[i], i)
        for j in range(len(mRate[0])):
            sim.set_mutation_rate(mRate[i][j][0], i, j)
            sim.set_mutation_probabilities([mRate[i][j][1], mRate[i][j][2], mRate[i][j][3], mRate[i][j][4]], i, j)
    for i in range(len(sizes)):
        sim.set_population_size(sizes[i], i)
        sim.set_contact_density(contactDensity[i], i)
        sim.set_npi([contactAfter[i], startLD[i], endLD[i]], i)
        sim.set_sampling_multiplier(samplingMultiplier[i], i)
        for j in range(len(sizes)):
            if i != j:
                sim.set_migration_probability(migrationRates[i][j], i, j)
    for i in range(len(susceptible)):
        for j in range(len(susceptible[i])):
            sim.set_susceptibility(float(susceptible[i][j]), i, j)
    for i in range(len(susType)):
        sim.set_susceptibility_type(susType[i], i)
    for i in range(len(suscepTransition)):
        for j in range(len(suscepTransition[i])):
            if i != j:
                sim.set_immunity_transition(suscepTransition[i][j], i, j)
