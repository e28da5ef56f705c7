# final collection, part 1: suite + default line + classes
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/final/gputest.txt; cat gpurun_out/final/gputest.txt
python bench.py > gpurun_out/final/bench_c4.json 2> gpurun_out/final/bench_c4.err; tail -2 gpurun_out/final/bench_c4.err
python bench.py --classes all --class-timeout 400 > gpurun_out/final/classes_c4.json 2> gpurun_out/final/classes_c4.err
python bench.py --classes repeats10,tandem2,snp25 --class-timeout 400 > gpurun_out/final/classes_c4_more.json 2> gpurun_out/final/classes_c4_more.err
