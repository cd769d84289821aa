"""Trajectory plots of the facade (reference ``src/_interface.py:639-797``): same method names, arguments and labels.
matplotlib is imported when the first plot is requested, so the simulator itself does not depend on it."""


def _plt():
    import matplotlib.pyplot as plt
    return plt


def _shade_lockdowns(plt, time_points, series, lockdowns):
    """Shades the stretches between a lockdown's start and its end (if:689-704): entries come in (on, off) pairs of
    ``[state, time]``; an unpaired last entry shades to the end of the series."""
    point = 0
    pointEnd = 0
    for ld in range(0, len(lockdowns), 2):
        while time_points[point] < lockdowns[ld][1]:
            point += 1
        if ld + 1 == len(lockdowns):
            plt.fill_between(time_points[point:], series[point:], alpha=0.2)
            continue
        while time_points[pointEnd] < lockdowns[ld + 1][1]:
            pointEnd += 1
        if pointEnd != point:
            plt.fill_between(time_points[point:pointEnd + 1], series[point:pointEnd + 1], alpha=0.2)


class PlotMixin:
    def _axes(self):
        if self.fig is None:
            self.fig, self.ax = _plt().subplots(figsize=(8, 6))
            self.ax.set_ylabel('Number of samples')
            self.ax.set_xlabel('Time')
            self.ax_2 = self.ax.twinx()
            self.ax_2.set_ylabel('Number of individuals')

    def add_plot_infectious(self, population, haplotype, step_num=100, label_infectious=None, label_samples=None):
        self._axes()
        if isinstance(haplotype, int):
            self.plot_infectious(population, haplotype, step_num, label_infectious, label_samples)
        elif isinstance(haplotype, str):
            for hi in sorted(self.simulation.calculate_indexes(haplotype, self.simulation.hapNum)):
                self.plot_infectious(population, hi, step_num, label_infectious, label_samples)
        else:
            print("Incorrect type of haplotype. Type should be int or str.")   # upstream prints "#TODO" (if:663)

    def plot_infectious(self, population, haplotype, step_num, label_infectious, label_samples):
        infections, sample, time_points, lockdowns = self.simulation.get_data_infectious(population, haplotype, step_num)
        name = 'pop:' + str(population) + ' hap:' + self.simulation.calculate_string_from_haplotype(haplotype)
        if label_infectious is None:
            self.ax_2.plot(time_points, infections, label='Infectious ' + name)
        elif isinstance(label_infectious, str):
            self.ax_2.plot(time_points, infections, label=label_infectious)
        else:
            print("Incorrect type of label. Type should be str or None.")   # upstream prints "#TODO"
        if label_samples is None:
            self.ax.plot(time_points, sample, "--", label='Samples ' + name)
        elif isinstance(label_samples, str):
            self.ax.plot(time_points, sample, "--", label=label_samples)
        else:
            print("Incorrect type of label. Type should be str or None.")   # upstream prints "#TODO"
        if len(lockdowns) != 0:
            _shade_lockdowns(_plt(), time_points, infections, lockdowns)

    def add_plot_susceptible(self, population, susceptibility_type, step_num=100, label_susceptible=None):
        self._axes()
        susceptible, time_points, lockdowns = self.simulation.get_data_susceptible(population, susceptibility_type, step_num)
        if label_susceptible is None:
            self.ax_2.plot(time_points, susceptible, label='Susceptible pop:' + str(population) + ' sus:' + str(susceptibility_type))
        elif isinstance(label_susceptible, str):
            self.ax_2.plot(time_points, susceptible, label=label_susceptible)
        else:
            print("Incorrect type of label. Type should be str or None.")   # upstream prints "#TODO"
        if len(lockdowns) != 0:
            _shade_lockdowns(_plt(), time_points, susceptible, lockdowns)

    def add_legend(self):
        lines_1, labels_1 = self.ax.get_legend_handles_labels()
        lines_2, labels_2 = self.ax_2.get_legend_handles_labels()
        self.ax.legend(lines_1 + lines_2, labels_1 + labels_2, loc=0)

    def add_title(self, name="Plot"):
        self.ax.set_title(name)

    def plot(self, name_file=None):
        if name_file:
            _plt().savefig(name_file)
        else:
            _plt().show()
        self.fig = None
