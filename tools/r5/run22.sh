# experiments: the repeat class with other second-attempt thresholds; the level pipeline with a leaf kernel of 4096 ranks
mkdir -p gpurun_out/final
for v in 0 2048 65536; do RV_CASCADE_DANGER_MIN=$v python bench.py --class-one repeats --L 250000000 --no-check 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('danger_min $v', round(d['ms_per_step'],1), d['path'], d['cascade_why'])"; done | tee gpurun_out/final/exp_danger_min.txt
RV_NO_CASCADE=1 python bench.py --class-one snp1 --L 250000000 --no-check 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level pipeline, leaf 2048', round(d['ms_per_step'],1), d['levels'])" | tee gpurun_out/final/exp_leaf.txt
RV_LIB_DIR=$PWD/gpurun_ab/l4096 RV_NO_CASCADE=1 python bench.py --class-one snp1 --L 250000000 --no-check 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level pipeline, leaf 4096', round(d['ms_per_step'],1), d['levels'], d['anchors'])" | tee -a gpurun_out/final/exp_leaf.txt
