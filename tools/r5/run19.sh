mkdir -p gpurun_out
timeout 3300 python tools/config5.py --genomes 100 --L 2000000 --chunksize 5 --dir /tmp/c5full > gpurun_out/config5_100x2M.json 2> gpurun_out/config5_100x2M.err
tail -4 gpurun_out/config5_100x2M.err; cat gpurun_out/config5_100x2M.json | cut -c1-900
