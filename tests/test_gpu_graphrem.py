"""The Python-3 `reveal rem` graph driver on top of reveal_amd's own index (HIP path): same bodies as
tests/test_cpu_graphrem.py runs with the reference's index -- BASELINE config 1's figures, graphs feeding graphs level by
level (reveal/align.py:27-54), multi-genome and multi-contig inputs -- plus a synthetic hierarchical run."""
import numpy as np
import pytest

import graphrem_cases as C
from helpers import synth
from reveal_amd import align, rem

pytestmark = pytest.mark.gpu


def test_config1_figures(tmp_path):
    C.config1(tmp_path, None)


def test_hierarchical_plan(tmp_path):
    C.hierarchical(tmp_path, None)


def test_multi_and_multicontig(tmp_path):
    C.multi_fasta(tmp_path, None)


def test_64bit_library(tmp_path):
    files = C.fasta_files(tmp_path, ["1a", "1b"])
    G, idx, fn = rem.graph_rem(files, str(tmp_path / "ab64.gfa"), sa64=True)
    spelled, _ = C.spelled_by_file(fn)
    assert spelled == C.input_sequences(files)
    assert sum(1 for n in G.seq_nodes() if G.aligned[n]) == 553


def test_synthetic_config5_shape(tmp_path):
    """BASELINE config 5 in miniature: 9 genomes of 60 kbp (1 % substitutions each), --order=sequential --chunksize=3:
    3 + 1 jobs; the final graph spells all nine"""
    seqs = synth.genomes(60000, 9, seed=21)
    files = []
    for k, s in enumerate(seqs):
        p = tmp_path / ("g%d.fa" % k)
        p.write_text(">genome%d\n%s\n" % (k, s.decode()))
        files.append(str(p))
    levels = align.sequential_plan(files, 3, output=str(tmp_path / "prg"), tmpdir=str(tmp_path))
    assert [len(j) for j in levels] == [3, 1]
    done = align.run_plan(levels)
    assert len(done) == 4
    spelled, G = C.spelled_by_file(levels[-1][-1][1])
    assert spelled == {"genome%d" % k: s.decode() for k, s in enumerate(seqs)}
    shared = [n for n in G.seq_nodes() if len(G.offsets[n]) == 9]
    assert sum(e - b for b, e in shared) > 0.3 * 60000
