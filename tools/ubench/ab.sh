cd $GRAFT_REPO_ROOT
python tools/scan_probe.py 5000000 50000000 2>&1 | tail -2 | cut -c1-140
