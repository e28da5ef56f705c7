mkdir -p gpurun_out
python -m pytest tests/test_gpu_graphrem.py -q -x -k "config5_all_three" 2>&1 | tail -3
timeout 7000 python tools/config5.py --genomes 100 --L 5000000 --chunksize 5 --dir /tmp/c5full > gpurun_out/config5_100x5M.json 2> gpurun_out/config5_100x5M.err
tail -3 gpurun_out/config5_100x5M.err
