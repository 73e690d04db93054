#!/usr/bin/env python
"""Generates opencv_amd/csrc/median_net.h: compare-exchange networks for the 5 x 5 median.

MI355_MEDIAN25_NET   the median of 25 unordered values on wire 12: Batcher's odd-even merge sort on 32 wires (the 7 extra wires hold +infinity,
                     so every exchange touching them is dropped), pruned to the exchanges wire 12 depends on.  113 exchanges.
The other networks work on COLUMNS SORTED ONCE PER POSITION (a column of the window is shared by the five outputs that see it):
MI355_SORT5          sorts 5 values (the optimal 9 exchanges)
MI355_MERGE55        two sorted 5-lists (wires 0..4, 5..9) -> the sorted 10-list, on the wires MI355_MERGE55_OUT names in ascending order
MI355_MID6           two sorted 10-lists (wires 0..9, 10..19) -> ranks 7..12 of the 20 on the wires MI355_MID6_OUT (ascending): of four sorted
                     columns only these six can still be the median of 25 (an element with >= 13 others of the 20 on one side cannot)
MI355_RANK5          a sorted 6-list (wires 0..5) and a sorted 5-list (6..10) -> rank 5 of the 11 on wire MI355_RANK5_OUT: with the seven
                     smallest of the 20 discarded that is rank 12 of the 25
Each is built from Batcher's odd-even merge for arbitrary lengths (Knuth 5.3.4), exchanges that never swap on inputs with the stated
precondition are dropped, exchanges the outputs do not depend on are pruned, and a greedy pass removes whatever else the outputs survive.
Verification is exhaustive over all 0/1 inputs that satisfy the precondition (min / max commute with thresholding, so the zero-one
principle holds under a monotone precondition); the unordered network is checked on random vectors and sampled 0/1 vectors."""
import itertools
import os
import random
import sys

import numpy as np

N, PAD = 25, 32


def batcher(n):
    pairs = []
    t = n.bit_length() - 1
    p = 1 << (t - 1)
    while p > 0:
        q, r, d = 1 << (t - 1), 0, p
        while d > 0:
            for i in range(n - d):
                if (i & p) == r:
                    pairs.append((i, i + d))
            d, q, r = q - p, q >> 1, p
        p >>= 1
    return pairs


def prune(pairs, out_wires):
    need = set(out_wires)
    kept = []
    for a, b in reversed(pairs):
        if a in need or b in need:
            kept.append((a, b))
            need.update((a, b))
    return kept[::-1]


def run(pairs, v):
    v = list(v)
    for a, b in pairs:
        if v[a] > v[b]:
            v[a], v[b] = v[b], v[a]
    return v


def oe_merge(a, b, net):
    """Batcher's odd-even merge of the sorted wire lists a and b (any lengths); returns the wires in ascending order of their contents"""
    if not a:
        return list(b)
    if not b:
        return list(a)
    if len(a) == 1 and len(b) == 1:
        net.append((a[0], b[0]))
        return [a[0], b[0]]
    c = oe_merge(a[0::2], b[0::2], net)
    d = oe_merge(a[1::2], b[1::2], net)
    out = [c[0]]
    i = 0
    while i < len(d) and i + 1 < len(c):
        net.append((d[i], c[i + 1]))
        out += [d[i], c[i + 1]]
        i += 1
    out += d[i:] if i < len(d) else c[i + 1:]
    return out


def states(groups, n):
    """all 0/1 inputs on n wires whose groups (wire lists in ascending order of content) are sorted; one column per input"""
    cols = []
    for combo in itertools.product(*[range(len(g) + 1) for g in groups]):
        v = [0] * n
        for g, k in zip(groups, combo):
            for w in g[len(g) - k:]:
                v[w] = 1
        cols.append(v)
    return np.array(cols, dtype=bool).T


def run01(net, S):
    S = S.copy()
    for a, b in net:
        lo, hi = S[a] & S[b], S[a] | S[b]
        S[a], S[b] = lo, hi
    return S


def correct(net, S, outs, ranks):
    R, ones, n = run01(net, S), S.sum(axis=0), S.shape[0]
    return all(np.array_equal(R[w], ones >= n - r) for w, r in zip(outs, ranks))


def drop_dead(net, S):
    S = S.copy()
    kept = []
    for a, b in net:
        if (S[a] & ~S[b]).any():
            kept.append((a, b))
            lo, hi = S[a] & S[b], S[a] | S[b]
            S[a], S[b] = lo, hi
    return kept


def greedy(net, S, outs, ranks, seed):
    rng = random.Random(seed)
    best = list(net)
    changed = True
    while changed:
        changed = False
        idx = list(range(len(best)))
        rng.shuffle(idx)
        for i in sorted(idx, reverse=True):
            cand = best[:i] + best[i + 1:]
            if correct(cand, S, outs, ranks):
                best, changed = cand, True
    return best


def build(groups, n, ranks):
    """merge the groups left to right, keep what the wanted ranks need; returns (network, wires holding those ranks)"""
    net = []
    out = list(groups[0])
    for g in groups[1:]:
        out = oe_merge(out, list(g), net)
    S = states(groups, n)
    outs = [out[r] for r in ranks]
    assert correct(net, S, outs, ranks)
    net = prune(drop_dead(net, S), outs)
    net = min((greedy(net, S, outs, ranks, s) for s in range(8)), key=len)
    assert correct(net, S, outs, ranks)
    return net, outs


def emit(f, name, net):
    f.write("#define %s(CE) \\\n" % name)
    for i in range(0, len(net), 8):
        f.write("    " + " ".join("CE(%d,%d)" % p for p in net[i:i + 8]) + (" \\\n" if i + 8 < len(net) else "\n"))


def main():
    random.seed(1)
    pairs = [(a, b) for a, b in batcher(PAD) if a < N and b < N]      # wires >= 25 hold +inf: exchanges with them never swap
    for _ in range(2000):
        v = [random.randrange(256) for _ in range(N)]
        assert run(pairs, v) == sorted(v)
    net = prune(pairs, [N // 2])
    for _ in range(20000):
        v = [random.randrange(256) for _ in range(N)]
        assert run(net, v)[N // 2] == sorted(v)[N // 2]
    for ones in (12, 13):                                               # 2 x C(25,12) = 10.4 M vectors is too slow in Python: sample 300k
        for _ in range(150000):
            idx = set(random.sample(range(N), ones))
            v = [1 if i in idx else 0 for i in range(N)]
            assert run(net, v)[N // 2] == sorted(v)[N // 2]

    sort5 = [(0, 1), (3, 4), (2, 4), (2, 3), (0, 3), (0, 2), (1, 4), (1, 3), (1, 2)]
    for v in itertools.product((0, 1), repeat=5):
        assert run(sort5, v) == sorted(v)
    merge55, m55_out = build([range(0, 5), range(5, 10)], 10, list(range(10)))
    mid6, mid6_out = build([range(0, 10), range(10, 20)], 20, list(range(7, 13)))
    rank5, rank5_out = build([range(0, 6), range(6, 11)], 11, [5])
    # the whole chain on random windows: columns sorted, pairs merged, middle six of four columns, rank 5 with the fifth column
    for _ in range(20000):
        w = [[random.randrange(256) if random.random() < 0.7 else random.choice((0, 7, 255)) for _ in range(5)] for _ in range(5)]
        cols = [run(sort5, c) for c in w]
        p = []
        for a, b in ((0, 1), (2, 3)):
            r = run(merge55, cols[a] + cols[b])
            p.append([r[i] for i in m55_out])
        r = run(mid6, p[0] + p[1])
        q = [r[i] for i in mid6_out]
        assert run(rank5, q + cols[4])[rank5_out[0]] == sorted(sum(w, []))[12]

    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opencv_amd", "csrc", "median_net.h")
    with open(out, "w") as f:
        f.write("// median_net.h -- GENERATED by tools/gen_median_net.py (which also verifies every network); do not edit.\n")
        f.write("// CE(a, b): wire a <- min, wire b <- max.\n")
        f.write("// MI355_MEDIAN25_NET: %d exchanges that leave the median of 25 unordered values on wire 12 (Batcher's odd-even merge sort, pruned).\n" % len(net))
        f.write("// Networks on columns sorted once per position: MI355_SORT5 (%d), MI355_MERGE55 (%d: sorted wires 0..4 + 5..9 -> sorted on MI355_MERGE55_OUT),\n" % (len(sort5), len(merge55)))
        f.write("// MI355_MID6 (%d: sorted 0..9 + 10..19 -> ranks 7..12 of the 20 on MI355_MID6_OUT), MI355_RANK5 (%d: sorted 0..5 + 6..10 -> rank 5 on MI355_RANK5_OUT).\n" % (len(mid6), len(rank5)))
        f.write("#pragma once\n")
        emit(f, "MI355_MEDIAN25_NET", net)
        emit(f, "MI355_SORT5", sort5)
        emit(f, "MI355_MERGE55", merge55)
        f.write("#define MI355_MERGE55_OUT {%s}\n" % ", ".join(map(str, m55_out)))
        emit(f, "MI355_MID6", mid6)
        f.write("#define MI355_MID6_OUT {%s}\n" % ", ".join(map(str, mid6_out)))
        emit(f, "MI355_RANK5", rank5)
        f.write("#define MI355_RANK5_OUT %d\n" % rank5_out[0])
    print(len(pairs), "->", len(net), "exchanges; sort5", len(sort5), "merge55", len(merge55), m55_out, "mid6", len(mid6), mid6_out, "rank5", len(rank5), rank5_out, ";", out)


if __name__ == "__main__":
    main()
