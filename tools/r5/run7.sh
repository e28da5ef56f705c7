mkdir -p gpurun_out
python tools/r5/repro.py 5101 5 2>&1 | tail -8
timeout 900 python tools/fuzz.py 200 5102 2>&1 | tail -3
python bench.py --classes all --class-timeout 400 > gpurun_out/classes_c4_a.json 2> gpurun_out/classes_c4_a.err
