#!/bin/bash
# round 6: BASELINE config 5 at its size, end to end: 100 genomes of 5 Mbp, 20 / 4 / 1 reveal rem jobs, graphs feeding graphs (levels 1-2: GFA inputs)
O=gpurun_out/r6config5; mkdir -p $O
free -g > $O/box.txt; nproc >> $O/box.txt; cat /sys/fs/cgroup/cpu.max >> $O/box.txt 2>/dev/null; df -h /tmp >> $O/box.txt; cat $O/box.txt
# (a memory guard: the Python graph of level 2 holds ~10^7 nodes with up to 100 path offsets each -- a MemoryError, not a dead box, if that does not fit)
MEMKB=$(awk '/MemAvailable/ {print int($2 * 0.8)}' /proc/meminfo); ulimit -v $MEMKB; echo "ulimit -v $MEMKB" >> $O/box.txt
python tools/config5.py --genomes 100 --L 5000000 --procs 8 --dir /tmp/config5_full > $O/config5_full.json 2> $O/config5_full.err
tail -c 1800 $O/config5_full.json; tail -5 $O/config5_full.err
