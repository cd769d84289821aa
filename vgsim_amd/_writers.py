"""Text writers for the genealogy: Newick tree, sample populations, mutations (reference src/IO.py:144-255, called by
``Simulator.export_newick`` / ``export_mutations``, src/_interface.py:498-522).  Same bytes as the reference writes;
the tree is walked with an explicit stack (the reference recurses once per tree level and fails on deep trees)."""


def find_children(pruferSeq, times):  # IO.py:207-217
    children = {}
    for index in range(len(pruferSeq)):
        children.setdefault(pruferSeq[index], []).append([index, times[index]])
    return children


def newick_and_populations(pruferSeq, times, populations):
    """(Newick string without the closing ';', sample-population table) of IO.py:169-205 / 225-250."""
    children = find_children(pruferSeq, times)
    root, root_time = children[-1][0][0], children[-1][0][1]
    # post-order for the Newick text, pre-order for the population table
    text = {}
    pops = []
    stack = [(root, root_time, root_time, False)]   # node, its time, base time of its branch, expanded?
    while stack:
        node, t, base, expanded = stack.pop()
        if node not in children:                         # Leaf (IO.py:192-204)
            text[node] = '{0}:{1}'.format(node, t - base)
            pops.append('{0}\t{1}\n'.format(node, populations[t]))
            continue
        (left, lt), (right, rt) = children[node][0], children[node][1]
        if not expanded:
            pops.append('{0}\t{1}\n'.format(node, populations[t]))
            stack.append((node, t, base, True))
            stack.append((right, rt, t, False))          # popped after the left subtree: pre-order left, then right
            stack.append((left, lt, t, False))
        else:
            text[node] = '({0},{1}){2}:{3}'.format(text.pop(left), text.pop(right), node, t - base)
    return text[root], ''.join(pops)


def write_newick(pruferSeq, times, populations, name_file, file_path):  # IO.py:225-255
    tree, pops = newick_and_populations(pruferSeq, times, populations)
    if file_path is not None:
        nwk, pop = file_path + '/' + name_file + '_tree.nwk', file_path + '/' + name_file + '_sample_population.tsv'
    elif name_file is not None:
        nwk, pop = name_file + '_tree.nwk', name_file + '_sample_population.tsv'
    else:
        nwk, pop = 'tree.nwk', 'sample_population.tsv'
    with open(nwk, 'w') as f:
        f.write(tree)
        f.write(';')
    with open(pop, 'w') as f:
        f.write(pops)


def mutation_lines(mut, len_prufer):
    """IO.py:144-167 including its lookup by first occurrence (``mut[0].index(nodeId)``): a node that carries several
    mutations is written with its FIRST mutation repeated."""
    alleles = ["A", "T", "C", "G"]
    AS = [alleles[v] for v in mut[1]]
    DS = [alleles[v] for v in mut[3]]
    first = {}
    for k, nodeId in enumerate(mut[0]):
        first.setdefault(nodeId, k)
    per_node = {}
    for nodeId in mut[0]:
        k = first[nodeId]
        per_node[nodeId] = per_node.get(nodeId, '') + str(AS[k]) + str(mut[2][k]) + str(DS[k]) + ','
    return ''.join(str(i) + '\t' + per_node[i][:-1] + '\n' for i in range(len_prufer) if i in per_node)


def write_mutations(mut, len_prufer, name_file, file_path):
    path = (file_path + '/' + name_file + '.tsv') if file_path is not None else (name_file + '.tsv')
    with open(path, 'w') as f:
        f.write(mutation_lines(mut, len_prufer))
