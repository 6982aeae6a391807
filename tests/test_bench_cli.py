"""CPU tests of bench.py's launcher logic (no GPU needed): --gpus N must either run N ranks or fail loudly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env=dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    """No GPU here, so the ranks stop at "bench.py needs an MI355X" (the launcher may end the second rank before it has
    said so: its own failure report shows that ranks were started) and the parent leaves with their failure, not with a
    silent single-process run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT, env=_env())
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.stderr.count("needs an MI355X") >= 2 or ("needs an MI355X" in r.stderr and "ChildFailedError" in r.stderr), r.stderr[-2000:]
