set -x
O=gpurun_out/r4_run4; mkdir -p $O
export TMPDIR=/tmp
python tools/fuzz.py 120 4242 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
python tools/fuzz.py 60 99 > $O/fuzz2.log 2>&1; tail -3 $O/fuzz2.log
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
