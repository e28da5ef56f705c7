O=gpurun_out/r4_soak; mkdir -p $O
python tools/fuzz.py 300 9001 > $O/fuzz_9001.log 2>&1; tail -1 $O/fuzz_9001.log
FUZZ_BIG=1 python tools/fuzz.py 300 9002 > $O/fuzz_big_9002.log 2>&1; tail -1 $O/fuzz_big_9002.log
python tools/fuzz_steps.py 120 9003 > $O/fuzz_steps_9003.log 2>&1; tail -1 $O/fuzz_steps_9003.log
python tools/fuzz_preselect.py 120 9004 > $O/fuzz_preselect_9004.log 2>&1; tail -1 $O/fuzz_preselect_9004.log
