"""BASELINE config 5 at its size: 100 genomes of 5 Mbp, `--order=sequential --chunksize=5` -> the 20 independent level-0 jobs of
reveal/align.py:27-54, all on this GPU (bench.py --config c5): every job's result passes the full-size properties
(reveal_amd/check.py), job 0 equals the CPU path's digests (tests/golden/fullsize.json, C5job_seed42)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config5_level0_all_jobs():
    env = {k: v for k, v in os.environ.items() if not k.startswith("RV_") and k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["config"]["jobs"] == 20 and d["config"]["bases_per_step"] == 500_000_000
    assert d["properties_full_size"]["all"] is True and d["properties_full_size"]["jobs_checked"] == 20, d["properties_full_size"]
    assert d["parity"]["full_size"]["all"] is True, d["parity"]
    assert len(d["jobs_anchors"]) == 20 and min(d["jobs_anchors"].values()) > 50_000
    assert d["value"] > 0 and d["n_gpus"] == 1
