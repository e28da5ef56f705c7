#!/bin/bash
# round 6: BASELINE config 5 at its size, end to end: 100 genomes of 5 Mbp, 20 / 4 / 1 reveal rem jobs, graphs feeding graphs (levels 1-2: GFA inputs)
O=gpurun_out/r6config5; mkdir -p $O
python tools/config5.py --genomes 100 --L 5000000 --procs 8 --dir /tmp/config5_full > $O/config5_full.json 2> $O/config5_full.err
tail -c 1500 $O/config5_full.json; tail -5 $O/config5_full.err
