mkdir -p gpurun_out
python bench.py --classes all --L 5000000 --class-timeout 200 > gpurun_out/classes_5M.json 2> gpurun_out/classes_5M.err
python bench.py --classes all --L 50000000 --class-timeout 300 > gpurun_out/classes_50M.json 2> gpurun_out/classes_50M.err
