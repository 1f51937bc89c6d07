"""Parity soak on a GPU: many random small worlds (mostly the default w = 14, k <= 15) through every sketch
kernel and sort variant, each compared with the oracle at every stage (tests/parity.py).

    python tests/soak_parity.py [first seed] [cases] [w_max: w uniform in 1..w_max instead of mostly 14]

(not collected by pytest: a longer run than the suite's random tests, started by hand)
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import pandora_amd
from pandora_amd import api, synth
from parity import compare_batch
pandora_amd.init(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.time()
n = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    rng = np.random.default_rng(seed0 + case)
    k = int(rng.choice([15, 15, 15, 14, 12, 9, 6]))
    w = 14 if rng.random() < 0.8 else int(rng.integers(1, 40))
    if len(sys.argv) > 3:
        w = int(rng.integers(1, int(sys.argv[3]) + 1))
    loci = [synth.random_dna(int(rng.integers(200, 6000)), rng) for _ in range(int(rng.integers(1, 8)))]
    if rng.random() < 0.5:
        loci.append(loci[0].copy())
    si = synth.build_index(loci, w, k, rng=rng, split_paths=float(rng.choice([0.0, 0.3])))
    reads = []
    for _ in range(int(rng.integers(1, 60))):
        kind = int(rng.integers(0, 8))
        if kind == 0:
            r = bytes(synth.random_dna(int(rng.integers(0, w + k + 3)), rng))
        elif kind == 1:
            r = bytes(synth.random_dna(int(rng.choice([480, 992, 1024, 2016])) + int(rng.integers(-20, 21)), rng))
        else:
            l = loci[int(rng.integers(0, len(loci)))]
            a = int(rng.integers(0, max(1, len(l) - 30)))
            r = bytes(l[a:a + int(rng.integers(30, 5000))])
            if kind == 2:
                r = bytes(synth.revcomp(np.frombuffer(r, np.uint8)))
            if kind == 3 and len(r) > 5:
                p = int(rng.integers(0, len(r)))
                r = r[:p] + b"N" + r[p + 1:]
            if kind == 4:
                r = r * int(rng.integers(2, 5))
            if kind == 5:
                # point mutations
                arr = np.frombuffer(r, np.uint8).copy()
                m = rng.random(len(arr)) < 0.05
                arr[m] = synth.random_dna(int(m.sum()), rng)
                r = bytes(arr)
        reads.append(r)
    bases, off = O.pack_reads(reads)
    # the presence filter in front of the table: the library's choice, none / hashed only, or the key-range bitmap
    os.environ.pop("PMAP_FILTER_DIRECT", None)
    mode = int(rng.integers(0, 3))
    if mode:
        os.environ["PMAP_FILTER_DIRECT"] = str(mode - 1)
    compare_batch(si, bases, off, w, k, int(rng.integers(1, 300)), api.fraction(float(rng.random()) * 0.2, k),
                  min_cluster_size=int(rng.integers(0, 4)), first_read_id=int(rng.integers(0, 1000)))
    n += 1
    if n % 20 == 0:
        print("cases", n, "%.1f s" % (time.time() - t0), flush=True)
print("soak ok:", n, "cases")
