mkdir -p gpurun_out
python - <<'PY' > gpurun_out/native_5x5M.txt 2>&1
import sys, os, time, tempfile, pathlib
sys.path.insert(0, os.getcwd())
from reveal_amd import rem, schemes, synth
tmp = pathlib.Path(tempfile.mkdtemp())
seqs = synth.genomes(5000000, 5, seed=42)
files = []
for k, s in enumerate(seqs):
    p = tmp / ("g%d.fa" % k); p.write_text(">genome%d\n%s\n" % (k, s.decode())); files.append(str(p))
for native in (True, False):
    t0 = time.perf_counter()
    G, idx, fn = rem.graph_rem(files, str(tmp / ("o%d.gfa" % native)), args=schemes.PickerArgs(), native=native, preselect=False)
    dt = time.perf_counter() - t0
    print("5 x 5 Mbp, native=%s: %.1f s, %d nodes, %d aligned, picker calls %s" % (native, dt, len(G.seq_nodes()), sum(1 for n in G.seq_nodes() if G.aligned[n]), idx.picker_info() if native else "-"), flush=True)
a, b = open(tmp / "o1.gfa").read(), open(tmp / "o0.gfa").read()
print("same GFA:", a == b, len(a))
PY
cat gpurun_out/native_5x5M.txt
