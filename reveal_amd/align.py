"""`reveal align --order=sequential --chunksize=k` (reveal/align.py:27-54): the level plan of a hierarchical alignment.

The reference prints a shell script of `reveal rem` commands: the inputs are cut into chunks of `chunksize`, every chunk
becomes one independent `reveal rem` job that writes a GFA, the GFAs of a level (plus the inputs the division left over)
are the inputs of the next level, until one graph is left.  BASELINE config 5 = 100 genomes, chunksize 5 -> 20 / 4 / 1 jobs.
Jobs of one level share nothing: they are the unit this package spreads over GPUs (one process per GPU, no exchange).
"""
import os
import time


def sequential_plan(inputs, chunksize, output="prg", tmpdir="."):
    """-> list of levels, each a list of jobs (inputs, output file).  Same chunking as align.py:30-54: k, m = divmod(len, n);
    k == 0 -> one job with everything; the m left-over inputs move on to the next level in front of the new graphs; the last
    job writes `output`.gfa, the others temporary graphs."""
    graphs = list(inputs)
    levels = []
    n = int(chunksize)
    serial = 0
    while len(graphs) > 1:
        k, m = divmod(len(graphs), n)
        if k == 0:
            chunks, graphs = [graphs], []
        else:
            chunks = [graphs[i * n:i * n + n] for i in range(k)]
            graphs = graphs[-m:] if m != 0 else []
        jobs = []
        for chunk in chunks:
            if len(chunks) == 1 and graphs == []:
                out = output + ".gfa"
            else:
                out = os.path.join(tmpdir, "level%d_job%d.gfa" % (len(levels), serial))
                serial += 1
            jobs.append((list(chunk), out))
            graphs.append(out)
        levels.append(jobs)
    return levels


def run_plan(levels, minlength=20, minn=2, sa64=False, args=None, rank=0, world=1, barrier=None, log=None, indexmod=None):
    """run the jobs of every level; with world > 1 job j of a level belongs to rank j % world (files are the only exchange,
    `barrier` -- e.g. torch.distributed.barrier -- separates the levels).  -> [(level, job, seconds, nodes, paths)] of this rank"""
    from . import rem
    done = []
    for lv, jobs in enumerate(levels):
        for j, (inputs, out) in enumerate(jobs):
            if j % world != rank:
                continue
            t0 = time.perf_counter()
            G, idx, fn = rem.graph_rem(inputs, out, sa64=sa64, minlength=minlength, minn=minn, args=args, indexmod=indexmod, materialize=False)
            dt = time.perf_counter() - t0
            if isinstance(G, dict):      # (FASTA inputs with reveal_amd's index: graph built, pruned and written behind the ABI)
                done.append((lv, j, dt, G["seq_nodes"], len(G["paths"])))
            else:
                done.append((lv, j, dt, len(G.seq_nodes()), len(G.paths)))
            if log:
                log("level %d job %d: %d inputs -> %s  %.2f s, %d nodes, %d paths" % (lv, j, len(inputs), fn, dt, done[-1][3], done[-1][4]))
            del G, idx
        if barrier is not None:
            barrier()
    return done
