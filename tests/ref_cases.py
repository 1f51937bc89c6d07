"""Known-answer vectors held by the reference's own unit tests, restated as data.

Each case gives the index records exactly as the test's Index::add_record calls build them, the
kmer graphs the test wires by hand (for KmerGraph::min_path_length) and the expectation the test
asserts.  Sources: test/utils_test.cpp:140-276 (addReadHits), :313-454
(simpleInferLocalPRGOrderForRead), :456-693 (biggerInferLocalPRGOrderForRead), :708-944
(pangraphFromReadFile).
"""
import numpy as np

import oracle_lib as O


def record(kmer, prg, path, knode=0, k=3):
    """index->add_record(min(kh.first, kh.second), prg, path, knode, kh.first < kh.second)."""
    f, r = O.kmerhash(kmer, k)
    return (min(f, r), prg, [(s, e - s) for s, e in path], knode, f < r)


def min_path_length(paths, edges):
    """KmerGraph::min_path_length (src/kmergraph.cpp:664-684) for a hand-built graph: `paths[i]` is
    the path of node id i as [(start, end), ...], `edges` are (from, to) in add_edge order."""
    n = len(paths)
    outs = [[] for _ in range(n)]
    for a, b in edges:
        if b not in outs[a]:
            outs[a].append(b)
    key = [[(s, e - s) for s, e in p] for p in paths]
    order = sorted(range(n), key=lambda i: key[i])     # sort_topologically: prg::Path::operator<
    ln = [0] * n
    for j in range(n - 1, 0, -1):
        for o in outs[order[j - 1]]:
            if ln[o] + 1 > ln[j - 1]:
                ln[j - 1] = ln[o] + 1
    return ln[0]


def gfa_text(paths, edges):
    """KmerGraph::save format (src/kmergraph.cpp:686-714)."""
    lines = ["H\tVN:Z:1.0\tbn:Z:--linear --singlearr"]
    for i, p in enumerate(paths):
        ptxt = "%d{%s}" % (len(p), "".join("[%d, %d)" % (s, e) for s, e in p))
        lines.append("S\t%d\t%s\tFC:i:0\t\tRC:i:0" % (i, ptxt))
        for a, b in edges:
            if a == i:
                lines.append("L\t%d\t+\t%d\t+\t0M" % (a, b))
    return "\n".join(lines) + "\n"


# ---- test/utils_test.cpp:140-276 addReadHits -------------------------------------------------
ADD_READ_HITS_RECORDS = [
    record("AGC", 1, [(0, 3)]),
    record("GCT", 1, [(1, 4)]),
    record("AGC", 3, [(0, 1), (4, 5), (8, 9)]),
    record("AGT", 3, [(0, 1), (4, 5), (12, 13)]),
    record("AGT", 3, [(0, 1), (19, 20), (23, 24)]),
    record("GCT", 3, [(4, 5), (8, 9), (16, 16), (23, 24)]),
    record("GTT", 3, [(4, 5), (12, 13), (16, 16), (23, 24)]),
]
_P1, _P2 = ((0, 3),), ((1, 3),)
_P3, _P4, _P5 = ((0, 1), (4, 1), (8, 1)), ((0, 1), (4, 1), (12, 1)), ((0, 1), (19, 1), (23, 1))
_P6, _P7 = ((4, 1), (8, 1), (16, 0), (23, 1)), ((4, 1), (12, 1), (16, 0), (23, 1))
# expected hits as (read_start, prg, path as (start, length) tuples, is_forward)
_EXP1 = {(0, 1, _P1, 1), (0, 1, _P2, 0), (0, 3, _P3, 1), (0, 3, _P6, 0)}
_EXP2 = {(1, 1, _P1, 0), (1, 1, _P2, 1), (1, 3, _P3, 0), (1, 3, _P6, 1)}
_EXP3 = {(0, 3, _P4, 1), (0, 3, _P5, 1), (1, 3, _P7, 1)}
_EXP4 = {(1, 3, _P7, 1)}
# (read, w, expected hit set)
ADD_READ_HITS_CASES = [
    ("AGC", 1, _EXP1),
    ("AGTT", 2, _EXP4),
    ("AGTT", 1, _EXP3),
    ("AGCT", 1, _EXP1 | _EXP2),
    ("AGCT", 2, _EXP1 | _EXP2),
]

# ---- test/utils_test.cpp:313-454 simpleInferLocalPRGOrderForRead ----------------------------
_LP1_SIMPLE = ([[(0, 0)], [(0, 3)], [(1, 4)], [(4, 4)]], [(0, 1), (1, 2), (2, 3)])
_LP3_SIMPLE = ([[(0, 0)], [(0, 1), (4, 5), (8, 9)], [(0, 1), (4, 5), (12, 13)], [(0, 1), (19, 20), (23, 24)],
                [(4, 5), (8, 9), (16, 16), (23, 24)], [(4, 5), (12, 13), (16, 16), (23, 24)],
                [(12, 13), (16, 16), (23, 25)], [(23, 26)], [(24, 27)], [(27, 27)]],
               [(0, 1), (0, 2), (0, 3), (1, 4), (2, 5), (5, 6), (3, 7), (4, 7), (6, 7), (7, 8), (8, 9)])
SIMPLE_INFER = {
    "records": [
        record("TAC", 1, [(0, 3)]), record("ACG", 1, [(1, 4)]),
        record("AGC", 0, [(0, 1), (4, 5), (8, 9)]), record("AGT", 0, [(0, 1), (4, 5), (12, 13)]),
        record("ATT", 0, [(0, 1), (19, 20), (23, 24)]), record("GCT", 0, [(4, 5), (8, 9), (16, 16), (23, 24)]),
        record("GTT", 0, [(4, 5), (12, 13), (16, 16), (23, 24)]), record("TTA", 0, [(12, 13), (16, 16), (23, 25)]),
        record("TAA", 0, [(23, 26)]), record("AAG", 0, [(24, 27)]),
    ],
    "graphs": {0: _LP3_SIMPLE, 1: _LP1_SIMPLE},     # prgs[0] = lp3 (id 0), prgs[1] = lp1 (id 1)
    "read": "AGTTAAGTACG", "w": 1, "k": 3, "max_diff": 1, "fraction": 0.1, "min_cluster_size": 1,
    "expected_nodes": {0, 1},
}

# ---- test/utils_test.cpp:456-693 and :708-944 --------------------------------------------------
_LP1 = ([[(0, 0)], [(0, 3)], [(1, 4)], [(2, 5)], [(3, 6)], [(4, 7)], [(7, 7)]],
        [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6)])
_LP3 = ([[(0, 0)], [(0, 1), (4, 5), (8, 9)], [(0, 1), (4, 5), (12, 13)], [(0, 1), (19, 20), (23, 24)],
         [(4, 5), (8, 9), (16, 16), (23, 24)], [(4, 5), (12, 13), (16, 16), (23, 24)], [(12, 13), (16, 16), (23, 25)],
         [(23, 26)], [(24, 27)], [(27, 27)]],
        [(0, 1), (0, 2), (0, 3), (1, 4), (2, 5), (5, 6), (3, 7), (4, 7), (6, 7), (7, 8), (8, 9)])
_LP4 = ([[(8, 8)], [(8, 11)], [(9, 12)], [(12, 12)]], [(0, 1), (1, 2), (2, 3)])
_LP2 = ([[(0, 0)], [(0, 3)], [(1, 4)], [(2, 5)], [(5, 5)]], [(0, 1), (1, 2), (2, 3), (3, 4)])
BIGGER_INFER = {
    "records": [
        record("TAC", 1, [(0, 3)]), record("ACG", 1, [(1, 4)]), record("CGG", 1, [(2, 5)]),
        record("GGT", 1, [(3, 6)]), record("GTA", 1, [(4, 7)]),
        record("ACC", 3, [(0, 1), (4, 5), (8, 9)]), record("AGT", 3, [(0, 1), (4, 5), (12, 13)]),
        record("ATT", 3, [(0, 1), (19, 20), (23, 24)]), record("GCT", 3, [(4, 5), (8, 9), (16, 16), (23, 24)]),
        record("GTT", 3, [(4, 5), (12, 13), (16, 16), (23, 24)]), record("TTA", 3, [(12, 13), (16, 16), (23, 25)]),
        record("TAT", 3, [(23, 26)]), record("ATG", 3, [(24, 27)]),
        record("CTA", 0, [(8, 11)]), record("TAG", 0, [(9, 12)]),
        record("CTA", 2, [(0, 3)]), record("TAC", 2, [(1, 4)]), record("ACT", 2, [(2, 5)]),
    ],
    "graphs": {0: _LP4, 1: _LP1, 2: _LP2, 3: _LP3},
    "read": "AGTTATGCTAGCTACTTACGGTA", "w": 1, "k": 3, "max_diff": 1, "fraction": 0.1, "min_cluster_size": 1,
    "expected_nodes": {0, 1, 2, 3},
}
# pangraph_from_read_file(read2.fa / read2.fq, ..., w=1, k=3, max_diff=1, e_rate=0.1, min_cluster_size=1)
PANGRAPH_FROM_READ_FILE = dict(BIGGER_INFER, e_rate=0.1, files=["read2.fa", "read2.fq"])


def flat_index(case_records):
    return O.FlatIndex.from_records(case_records)


def case_mpl(case):
    n_prg = max(case["graphs"]) + 1
    mpl = np.zeros(n_prg, np.uint32)
    for p, (paths, edges) in case["graphs"].items():
        mpl[p] = min_path_length(paths, edges)
    return mpl
