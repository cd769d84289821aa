"""TEST INFRASTRUCTURE.  Literal restatement of the reference's log replays get_data_infectious /
get_data_susceptible (src/_BirthDeath.pyx:1967-2045), loop for loop including the operator precedence of
pyx:1982 / 1995 (``A or B or C and D``: recoveries and samplings of EVERY compartment decrement the series).
Pinned on tests/golden/timeline_*.npz (recorded from the reference).  ``mev`` is the multievent log the MULTITYPE
events index (dict of arrays: num, types, haplotypes, populations, newHaplotypes, newPopulations) or None."""
import numpy as np

BIRTH, DEATH, SAMPLING, MUTATION, SUSCCHANGE, MIGRATION, MULTITYPE = range(7)


def _lockdowns(m, pop):
    return [[m.loc.states[i], m.loc.times[i]] for i in range(len(m.loc.times)) if m.loc.populationsId[i] == pop]


def get_data_infectious(m, mev, pop, hap, step_num):
    ev = m.events
    time_points = [i * m.currentTime / step_num for i in range(step_num + 1)]
    Data = np.zeros(step_num + 1)
    Sample = np.zeros(step_num + 1)
    Data[0] = m.initial_infectious[pop, hap]
    point = 0
    for i in range(ev.ptr):
        while point != step_num and time_points[point] < ev.times[i]:
            Data[point + 1] = Data[point]
            Sample[point + 1] = Sample[point]
            point += 1
        t = ev.types[i]
        if t == BIRTH and ev.populations[i] == pop and ev.haplotypes[i] == hap:
            Data[point] += 1
        elif t == DEATH or t == SAMPLING or t == MUTATION and ev.populations[i] == pop and ev.haplotypes[i] == hap:
            Data[point] -= 1
            if t == SAMPLING:
                Sample[point] += 1
        elif t == MUTATION and ev.newHaplotypes[i] == hap and ev.populations[i] == pop:
            Data[point] += 1
        elif t == MIGRATION and ev.newPopulations[i] == pop and ev.haplotypes[i] == hap:
            Data[point] += 1
        elif t == MULTITYPE:
            for j in range(ev.haplotypes[i], ev.populations[i]):
                mt = mev["types"][j]
                if mt == BIRTH and mev["haplotypes"][j] == hap and mev["populations"][j] == pop:
                    Data[point] += mev["num"][j]
                elif mt == DEATH or mt == SAMPLING or mt == MUTATION and mev["haplotypes"][j] == hap and mev["populations"][j] == pop:
                    Data[point] -= mev["num"][j]
                    if mt == SAMPLING:
                        Sample[point] += mev["num"][j]
                elif mt == MUTATION and mev["newHaplotypes"][j] == hap and mev["populations"][j] == pop:
                    Data[point] += mev["num"][j]
                elif mt == MIGRATION and mev["newPopulations"][j] == pop and mev["haplotypes"][j] == hap:
                    Data[point] += mev["num"][j]
    return Data, Sample, time_points, _lockdowns(m, pop)


def get_data_susceptible(m, mev, pop, sus, step_num):
    ev = m.events
    time_points = [i * m.currentTime / step_num for i in range(step_num + 1)]
    Data = np.zeros(step_num + 1)
    Data[0] = m.initial_susceptible[pop, sus]
    point = 0
    for i in range(ev.ptr):
        while point != step_num and time_points[point] < ev.times[i]:
            Data[point + 1] = Data[point]
            point += 1
        t = ev.types[i]
        if t == BIRTH and ev.populations[i] == pop and ev.newHaplotypes[i] == sus:
            Data[point] -= 1
        elif (t == DEATH or t == SAMPLING or t == SUSCCHANGE) and ev.populations[i] == pop and ev.newHaplotypes[i] == sus:
            Data[point] += 1
        elif t == SUSCCHANGE and ev.haplotypes[i] == sus and ev.populations[i] == pop:
            Data[point] -= 1
        elif t == MIGRATION and ev.newPopulations[i] == pop and ev.newHaplotypes[i] == sus:
            Data[point] -= 1
        elif t == MULTITYPE:
            for j in range(ev.haplotypes[i], ev.populations[i]):
                mt = mev["types"][j]
                if mt == BIRTH and mev["newHaplotypes"][j] == sus and mev["populations"][j] == pop:
                    Data[point] -= mev["num"][j]
                elif (mt == DEATH or mt == SAMPLING or mt == SUSCCHANGE) and mev["newHaplotypes"][j] == sus and mev["populations"][j] == pop:
                    Data[point] += mev["num"][j]
                elif mt == SUSCCHANGE and mev["haplotypes"][j] == sus and mev["populations"][j] == pop:
                    Data[point] -= mev["num"][j]
                elif mt == MIGRATION and mev["newPopulations"][j] == pop and mev["haplotypes"][j] == sus:
                    Data[point] -= mev["num"][j]
    return Data, time_points, _lockdowns(m, pop)
