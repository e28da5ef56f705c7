"""rem.fasta_reader's whole-buffer form beside the line-by-line form it replaces for ordinary files (reveal/utils.py:79-160 with its defaults): hand-made odd files and random ones"""
import gzip
import os
import random

from reveal_amd import rem


def line_by_line(fn, toupper=True, keepdash=False):
    name, seq = None, []
    fopen = gzip.open if fn.endswith(".gz") else open
    with fopen(fn, "rt") as ff:
        for line in ff:
            line = line.rstrip()
            if line.startswith(">"):
                if seq:
                    yield name, "".join(seq)
                name, seq = line.replace(">", "").replace("\t", ""), []
            else:
                if toupper:
                    line = line.upper()
                if not keepdash:
                    line = line.replace("-", "")
                seq.append(line)
    if seq:
        yield name, "".join(seq)


def test_fast_reader_equals_the_line_reader(tmp_path):
    rng = random.Random(1)
    cases = [">a\nACGT\nacgt\n>b x\ty\nAC-GT\n\n>c\n>d\nNNNN", ">a\nACGT \nAC\n", ">a\r\nACGT\r\n", "ACGT\n>a\nAC\n", ">a\nAC>GT\n", ">only\n", "", ">a\nACGT",
             ">a\n\n\nAC\n\nGT\n>b\n", ">x\n-\n>y\nAC\n", ">x\n\n>y\n\n", ">x\n\n", ">x", ">\nAC\n", ">a\n>b\n>c\nA\n", ">a\nAC\rGT\n>b\rTT\n", ">a\nAC\x0bGT\n"]
    for _ in range(1500):
        t = ""
        for _r in range(rng.randint(0, 4)):
            t += ">" + "".join(rng.choice("abc xyz\t|>") for _ in range(rng.randint(0, 6))) + "\n"
            for _l in range(rng.randint(0, 4)):
                t += "".join(rng.choice("ACGTacgtN-") for _ in range(rng.randint(0, 12))) + rng.choice(["\n", "\n", "\n", "\n", " \n", "\r\n", "\r", ""])
        cases.append(t)
    for i, t in enumerate(cases):
        fn = str(tmp_path / ("c%d.fa%s" % (i, ".gz" if i % 7 == 0 else "")))
        with (gzip.open if fn.endswith(".gz") else open)(fn, "wt", newline="") as f:
            f.write(t)
        for up in (True, False):
            for kd in (False, True):
                assert list(rem.fasta_reader(fn, up, kd)) == list(line_by_line(fn, up, kd)), (t, up, kd)
        os.remove(fn)
