mkdir -p gpurun_out
RV_NO_CASCADE=1 python tools/level_log.py 2 250000000 2> gpurun_out/level_log_c4.txt
for H in 2 4; do python bench.py --config stream --pairs 20 --steps 1 --warmup 1 --stream-handles $H --no-check > gpurun_out/stream20_h$H.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/stream20_h$H.json'));print($H,d['value'],d['ms_per_input'],d['host_thread_ms_per_input'])"; done
