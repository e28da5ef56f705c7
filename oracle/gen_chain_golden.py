#!/usr/bin/env python3
"""Golden vectors for chain() / gapcost() / trim_overlap() / segment() of the reference's Python picker
(reveal/schemes.py:20-105, 107-126, 160-193; reveal/utils.py:162-183) -- TEST INFRASTRUCTURE, build container only.

The reference is Python 2.  This script converts reveal/schemes.py and reveal/utils.py IN MEMORY with lib2to3 (nothing of
them is written to the repository), pulls the four pure functions out of the converted syntax trees and executes THEM on
seeded random inputs (and on adversarial ones: ties, nested and overlapping matches, more than two paths).  The module-level
imports of those files (intervaltree, matplotlib ...) are never executed, so nothing the image lacks is stood in for.  Two
Python-2 semantics the converted code would otherwise lose are restored explicitly, as SURVEY.md Appendix A lists them:
`/` on ints is floor division (schemes.py:71, utils.py:166-168) and `x > None` is True for any number (schemes.py:84).

    python oracle/gen_chain_golden.py        # rewrites tests/golden/chain_vectors.json
"""
import ast
import json
import logging
import math
import os
import random
import sys
from lib2to3 import refactor

REF = "/root/reference/reveal"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def convert(path):
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    src = open(path).read()
    if not src.endswith("\n"):
        src += "\n"
    return str(tool.refactor_string(src, path))


class FloorDiv(ast.NodeTransformer):
    """Python 2 `/` on the ints these functions divide"""
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            node.op = ast.FloorDiv()
        return node


class NoneCompare(ast.NodeTransformer):
    """`tmpw>w or w==None` (schemes.py:84): Python 2 orders None below every number -> test for None first"""
    def visit_BoolOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Or) and len(node.values) == 2:
            a, b = node.values
            if (isinstance(b, ast.Compare) and isinstance(b.ops[0], ast.Eq) and isinstance(b.comparators[0], ast.Constant)
                    and b.comparators[0].value is None):
                node.values = [b, a]
        return node


def functions(path, names):
    tree = ast.parse(convert(path))
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=body, type_ignores=[])
    mod = ast.fix_missing_locations(NoneCompare().visit(FloorDiv().visit(mod)))
    return compile(mod, path, "exec")


class Args:
    wscore = 1
    wpen = 1


def load():
    logging.trace = lambda *a, **k: None
    env_u = {"logging": logging, "log": math.log}
    exec(functions(os.path.join(REF, "utils.py"), {"gapcost"}), env_u)

    class U:
        gapcost = staticmethod(env_u["gapcost"])
    env_s = {"logging": logging, "utils": U, "args": Args, "math": math}
    exec(functions(os.path.join(REF, "schemes.py"), {"chain", "trim_overlap", "segment"}), env_s)
    return env_u["gapcost"], env_s


def random_case(rng, m, k, span, collinear):
    """m matches over k paths inside (left, right); a collinear backbone plus noise"""
    keys = sorted(rng.sample(range(0, 12), k))
    base = sorted(rng.sample(range(10, span), m))
    mums = []
    used = set()
    for i in range(m):
        l = rng.randint(1, 40)
        crd = {}
        for q, key in enumerate(keys):
            if q == 0 or rng.random() < collinear:
                crd[key] = max(0, base[i] + (rng.randint(-15, 15) if q else 0))
            else:
                crd[key] = rng.randint(5, span)
        if crd[keys[0]] in used:
            continue
        used.add(crd[keys[0]])
        mums.append((l, k, crd))
    left = (0, 0, {key: -1 for key in keys})
    right = (0, 0, {key: span + 60 for key in keys})
    return mums, left, right


def main():
    gapcost, env = load()
    chain, trim_overlap, segment = env["chain"], env["trim_overlap"], env["segment"]
    rng = random.Random(20260930)
    out = {"_generator": "oracle/gen_chain_golden.py: reveal/schemes.py chain/trim_overlap/segment + reveal/utils.py gapcost, converted in memory (lib2to3) and executed",
           "gapcost": [], "chain": [], "trim_overlap": [], "segment": []}
    for _ in range(60):
        k = rng.randint(1, 7)
        a = [rng.randint(0, 1000) for _ in range(k)]
        b = [rng.randint(0, 1000) for _ in range(k)]
        out["gapcost"].append({"a": a, "b": b, "sumofpairs": gapcost(a, b), "star-avg": gapcost(a, b, model="star-avg"), "star-med": gapcost(a, b, model="star-med")})
    for wscore, wpen in ((1, 1), (3, 1), (1, 4)):
        Args.wscore, Args.wpen = wscore, wpen
        for m, k, span, col in ((1, 2, 100, 1.0), (2, 2, 100, 1.0), (6, 2, 300, 0.9), (25, 2, 2000, 0.8), (25, 3, 2000, 0.9), (60, 2, 5000, 0.95),
                                (60, 5, 5000, 0.9), (120, 2, 9000, 0.7), (120, 4, 9000, 0.97), (40, 2, 400, 0.99), (200, 3, 30000, 0.9)):
            for rep in range(3):
                mums, left, right = random_case(rng, m, k, span, col)
                inp = [(l, n, dict(c)) for l, n, c in mums]
                res = chain(list(inp), left, right)
                keys = sorted(left[2])
                out["chain"].append({"wscore": wscore, "wpen": wpen, "keys": keys, "left": [left[2][q] for q in keys], "right": [right[2][q] for q in keys],
                                     "mums": [[l, n] + [c[q] for q in keys] for l, n, c in mums],
                                     "path": [[mm[0], mm[1]] + [mm[2][q] for q in keys] + [sc] for mm, sc in res]})
    Args.wscore, Args.wpen = 1, 1
    for _ in range(40):      # trim_overlap works on index-form matches: (l, n, ((sample, pos), ...))
        k = rng.randint(2, 4)
        m = rng.randint(1, 14)
        mums = []
        at = [rng.randint(0, 50) for _ in range(k)]
        for i in range(m):
            l = rng.randint(3, 30)
            step = rng.randint(1, 25)
            at = [x + step + rng.randint(0, 3) for x in at]
            mums.append((l, k, tuple((q, at[q]) for q in range(k))))
        rng.shuffle(mums)
        res = trim_overlap(list(mums))
        out["trim_overlap"].append({"mums": [[l, n, [list(x) for x in spd]] for l, n, spd in mums], "out": [[l, n, [list(x) for x in spd]] for l, n, spd in res]})
    for _ in range(25):
        mums = []
        for i in range(rng.randint(1, 20)):
            ns = rng.randint(2, 4)
            sm = sorted(rng.sample(range(5), ns))
            rng.shuffle(sm)
            mums.append((rng.randint(1, 60), ns, tuple((s, rng.randint(0, 999)) for s in sm)))
        res = segment(list(mums))
        out["segment"].append({"mums": [[l, n, [list(x) for x in spd]] for l, n, spd in mums], "out": [[l, n, [list(x) for x in spd]] for l, n, spd in res]})
    with open(os.path.join(ROOT, "tests", "golden", "chain_vectors.json"), "w") as f:
        json.dump(out, f, sort_keys=True)
    print("wrote tests/golden/chain_vectors.json: %d gapcost, %d chain, %d trim_overlap, %d segment cases" % (len(out["gapcost"]), len(out["chain"]), len(out["trim_overlap"]), len(out["segment"])))


if __name__ == "__main__":
    main()
