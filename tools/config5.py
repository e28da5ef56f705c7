#!/usr/bin/env python3
"""BASELINE config 5 end to end: N synthetic genomes of L bp, `reveal align --order=sequential --chunksize=C` as a plan of
independent `reveal rem` jobs (reveal/align.py:27-54; 100 genomes, C = 5 -> 20 / 4 / 1), every job through the graph
callbacks (reveal_amd/rem.py graph_rem) on this process' GPU, graphs feeding graphs as GFA files; at the end every input must
be spelled by its path of the final graph (the reference's test15 invariant).  Prints one JSON line with per-level times.

    python tools/config5.py --genomes 100 --L 100000 --chunksize 5 [--jobs-only 0]      # --jobs-only k: only level k
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=100)
    ap.add_argument("--L", type=int, default=100000)
    ap.add_argument("--chunksize", type=int, default=5)
    ap.add_argument("--minl", type=int, default=20)
    ap.add_argument("--minn", type=int, default=2)
    ap.add_argument("--max-jobs", type=int, default=0, help="run at most this many jobs of level 0 and stop (timing of single jobs)")
    ap.add_argument("--dir", default=None)
    ap.add_argument("--procs", type=int, default=1, help="jobs of a level run as this many `python -m reveal_amd.rem` processes at a time on this GPU (they share "
                                                        "nothing but files: reveal/align.py prints them as shell commands); 1 = in this process")
    ap.add_argument("--barriers", action="store_true", help="--procs > 1: a level's jobs start only when the level below is finished (default: a job starts when its inputs exist)")
    a = ap.parse_args()
    from reveal_amd import align, synth
    import graphrem_cases as C
    d = a.dir or tempfile.mkdtemp(prefix="config5_")
    os.makedirs(d, exist_ok=True)
    t0 = time.perf_counter()
    seqs = synth.genomes(a.L, a.genomes, seed=42)
    files = []
    for k, s in enumerate(seqs):
        fn = os.path.join(d, "g%03d.fa" % k)
        with open(fn, "w") as f:
            f.write(">genome%03d\n%s\n" % (k, s.decode()))
        files.append(fn)
    t_gen = time.perf_counter() - t0
    levels = align.sequential_plan(files, a.chunksize, output=os.path.join(d, "prg"), tmpdir=d)
    if a.max_jobs:
        levels = [levels[0][:a.max_jobs]]
    logs = []
    t1 = time.perf_counter()
    if a.procs <= 1:
        done = align.run_plan(levels, minlength=a.minl, minn=a.minn, log=lambda m: (logs.append(m), print(m, file=sys.stderr)))
    else:
        import re
        import subprocess
        done = []
        level_wall = {}
        stage_log = []
        # every job of the plan, started as soon as its inputs exist and a process slot is free (what `make -j` does with the commands reveal/align.py prints level by level;
        # --barriers keeps the levels apart); jobs of lower levels first
        todo = [(lv, j, inputs, out) for lv, jobs in enumerate(levels) for j, (inputs, out) in enumerate(jobs)]
        made_by_plan = {out for _, _, _, out in todo}
        finished = set()
        span = {}      # level -> [first start, last end]
        running = []
        while todo or running:
            lowest = min([lv for lv, _, _, _ in todo] + [x[0] for x in running]) if (todo or running) else 0
            k = 0
            while k < len(todo) and len(running) < a.procs:
                lv, j, inputs, out = todo[k]
                ready = all((f not in made_by_plan) or (f in finished) for f in inputs) and (not a.barriers or lv == lowest)
                if not ready:
                    k += 1
                    continue
                todo.pop(k)
                cmd = [sys.executable, "-m", "reveal_amd.rem"] + list(inputs) + ["-o", out, "-m", str(a.minl), "-n", str(a.minn)]
                now = time.perf_counter()
                span.setdefault(lv, [now, now])
                running.append((lv, j, out, now, subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                                                  env=dict(os.environ, REVEAL_AMD_TIMES="1", RV_GRAPH_TIMES="1"))))
            still = []
            for lv, j, out, ts, pr in running:
                if pr.poll() is None:
                    still.append((lv, j, out, ts, pr))
                    continue
                so, se = pr.communicate()
                if pr.returncode != 0:
                    sys.exit("level %d job %d failed: %s" % (lv, j, se[-2000:]))
                mm = re.search(r"(\d+) nodes, (\d+) paths", so)
                now = time.perf_counter()
                dt = now - ts
                finished.add(out)
                span[lv][1] = now
                done.append((lv, j, dt, int(mm.group(1)) if mm else 0, int(mm.group(2)) if mm else 0))
                print("level %d job %d -> %s  %.2f s (process), %s" % (lv, j, out, dt, so.strip().splitlines()[-1] if so.strip() else ""), file=sys.stderr)
                if lv > 0:
                    for ln in se.splitlines():
                        if ln.startswith(("stages:", "graphalign:", "read_gfa:")):
                            print("    " + ln, file=sys.stderr)
                            stage_log.append((lv, j, ln))
            running = still
            if running:
                time.sleep(0.1)
        level_wall = {str(lv): v[1] - v[0] for lv, v in span.items()}
        level_span = {str(lv): [v[0] - t1, v[1] - t1] for lv, v in span.items()}
    t_run = time.perf_counter() - t1
    per_level = {}
    for lv, j, dt, nodes, paths in done:
        e = per_level.setdefault(lv, dict(jobs=0, seconds=0.0, max_job_s=0.0, nodes=0))
        e["jobs"] += 1; e["seconds"] += dt; e["max_job_s"] = max(e["max_job_s"], dt); e["nodes"] += nodes
    out = dict(genomes=a.genomes, L=a.L, chunksize=a.chunksize, plan=[len(j) for j in levels], processes_per_level=a.procs, t_generate_s=t_gen, t_run_s=t_run,
               levels={str(k): v for k, v in per_level.items()}, bases=a.genomes * a.L)
    if a.procs > 1:
        out["level_wall_s"] = level_wall
        out["level_span_s"] = level_span      # [first start, last end] of a level's jobs, seconds from the start of the run (levels overlap unless --barriers)
        out["barriers"] = bool(a.barriers)
        out["stages_of_graph_jobs"] = ["level %d job %d %s" % x for x in stage_log]
    if not a.max_jobs:
        t2 = time.perf_counter()
        # test15's invariant on the FILE, by a parse of its own (no reader of the package): every P line's segments, joined, spell that input
        seg, visits, ok, npaths = {}, {}, True, 0
        import gzip
        fn = levels[-1][-1][1]
        want = {"genome%03d" % k: s for k, s in enumerate(seqs)}
        with (gzip.open if fn.endswith(".gz") else open)(fn, "rb") as f:
            for line in f:
                if line[:1] == b"S":
                    c = line.rstrip(b"\n").split(b"\t")
                    seg[c[1]] = c[2].upper()
                elif line[:1] == b"P":
                    c = line.rstrip(b"\n").split(b"\t")
                    steps = [x[:-1] for x in c[2].split(b",")] if c[2] else []
                    ok &= all(x[-1:] == b"+" for x in c[2].split(b",")) if c[2] else True
                    ok &= b"".join(seg[x] for x in steps) == want.get(c[1].decode())
                    for x in set(steps):
                        visits[x] = visits.get(x, 0) + 1
                    npaths += 1
        out["paths_spell_inputs"] = bool(ok and npaths == a.genomes)
        out["final_nodes"] = len(seg)
        out["final_nodes_in_all_paths"] = sum(1 for v in visits.values() if v == a.genomes)
        out["t_check_s"] = time.perf_counter() - t2
        out["Mbp_per_s_end_to_end"] = a.genomes * a.L / t_run / 1e6
    print(json.dumps(out))


if __name__ == "__main__":
    main()
