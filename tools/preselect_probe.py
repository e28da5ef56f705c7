"""What index.preselect(maxmums) saves on the Python-callback path: the same alignment
(align() with Python callbacks that start like schemes.graphmumpicker: filter, sort, cap at maxmums, then take the
longest) with and without it.  Prints wall time, matches that crossed into Python, and checks the anchors agree.
usage: python tools/preselect_probe.py [L=1000000] [genomes=2] [maxmums=1000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reveal_amd import rem, reveallib, synth       # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    maxmums = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    seqs = [g.decode() for g in synth.genomes(L, G)]
    out = {}
    for on in (False, True, False, True):
        idx = reveallib.index()
        rem.add_sequences(idx, seqs)
        idx.construct()
        if on:
            idx.preselect(maxmums)
        crossed, anchors = [0], []

        def pick(mums, i, precomputed=False, minlength=0):
            crossed[0] += len(mums)
            mm = [m for m in mums if m[1] == i.nsamples]           # schemes.py:227
            mm.sort(key=lambda m: m[0], reverse=True)              # schemes.py:240
            mm.sort(key=lambda m: (m[1], m[0]))                    # schemes.py:246
            mm = mm[-maxmums:]                                     # schemes.py:287-289
            return rem.bench_mumpicker(mm, i)

        def galign(i, mum):
            r = rem.linear_graphalign(i, mum)
            if r is not None:
                anchors.append(mum)
            return r
        t0 = time.time()
        idx.align(pick, galign, minl=20, minn=2)
        dt = time.time() - t0
        out[on] = (dt, crossed[0], sorted(anchors))
        print("preselect %-5s  align %.3f s  %d matches handed to Python, %d anchors" % (on, dt, crossed[0], len(anchors)), flush=True)
    assert out[False][2] == out[True][2]
    print("same anchors; %d x %d bp, maxmums %d: %.2fx" % (G, L, maxmums, out[False][0] / out[True][0]))


if __name__ == "__main__":
    main()
