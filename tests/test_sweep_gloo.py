"""N > 1 path of the depth sweep on CPU: two ranks, gloo backend (the RCCL path uses the same
torch.distributed calls with backend "nccl")."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_share_is_block_cyclic_and_complete():
    from remo3d_amd import sweep
    for n in (0, 1, 7, 40):
        for w in (1, 2, 3, 8):
            shares = [list(sweep.my_share(n, r, w)) for r in range(w)]
            assert sorted(i for s in shares for i in s) == list(range(n))
            assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1


def _run_two_ranks(tmp_path, schedule, mode="normal"):
    out = str(tmp_path / "res")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_sweep_worker.py"), out, schedule, mode]
    env = dict(os.environ, OMP_NUM_THREADS="1", REMO_DIST_BACKEND="gloo")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [json.load(open(f"{out}.{k}")) for k in range(2)]


def _check_logs(res):
    # both ranks hold the same, complete logs after the all-reduce
    for tool in res[0]["logs"]:
        a = np.array(res[0]["logs"][tool]); b = np.array(res[1]["logs"][tool])
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.allclose(a, b, equal_nan=True, rtol=0, atol=0)
        good = ~np.isnan(a[:, 1])
        assert np.allclose(a[good, 1], 7.0, rtol=1e-12)        # homogeneous medium: Ra == R (remo3d.py:285-306)
    # the injected failure of batch 3 became NaN for exactly its records (worker.py:135-138)
    n_nan = sum(int(np.isnan(np.array(v)[:, 1]).sum()) for v in res[0]["logs"].values())
    assert 0 < n_nan <= 4 * 3


def test_two_rank_sweep_with_pull_scheduling(tmp_path):
    """schedule="dynamic" (the reference's pull scheduling, remo3d.py:843-860): every batch is drawn exactly once from
    the shared counter; the rank that is made slow ends up with fewer batches; logs identical to the static sweep."""
    res = _run_two_ranks(tmp_path, "dynamic")
    n = res[0]["n_batches"]
    assert res[0]["taken"] + res[1]["taken"] == n
    assert res[0]["calls"] + res[1]["calls"] == n
    slow = res[0] if res[0]["rank"] == 0 else res[1]
    fast = res[1] if res[0]["rank"] == 0 else res[0]
    assert slow["taken"] < fast["taken"], (slow["taken"], fast["taken"])
    _check_logs(res)
    assert res[0]["timing"]["failed_batches"] + res[1]["timing"]["failed_batches"] == 1


def test_two_rank_sweep_combines_logs(tmp_path):
    """Model.initialize_workers joins the process group by itself (the worker never calls sweep.init_from_env)."""
    res = _run_two_ranks(tmp_path, "static")
    assert res[0]["world"] == 2 and {res[0]["rank"], res[1]["rank"]} == {0, 1}
    assert res[0]["timing"]["world_size"] == 2
    # disjoint block-cyclic shares that cover all batches; every rank only solved its own share
    all_b = sorted(res[0]["share"] + res[1]["share"])
    assert all_b == list(range(len(all_b))) and res[0]["share"][0] == 0 and res[1]["share"][0] == 1
    assert res[0]["calls"] == len(res[0]["share"]) - (1 if 3 in res[0]["share"] else 0) or res[0]["calls"] == len(res[0]["share"])
    _check_logs(res)
    # the failed batch is on record (not only NaN): rank 1 owns batch 3
    assert res[1]["timing"]["failed_batches"] == 1 and "injected failure" in res[1]["first_error"]
    assert res[0]["timing"]["failed_batches"] == 0
    assert len(res[0]["timing"]["busy_s_per_rank"]) == 2


def test_rank_that_pulls_nothing_still_joins_the_collectives(tmp_path):
    """A sweep of ONE batch under the pull schedule: one rank draws it, the other draws nothing and must still take part in
    the completeness check, the all-reduce of the logs and the gather of the busy times (no hang, complete logs on both)."""
    res = _run_two_ranks(tmp_path, "dynamic", "one_batch")
    assert res[0]["n_batches"] == 1
    assert sorted(r["taken"] for r in res) == [0, 1]
    for tool in res[0]["logs"]:
        a = np.array(res[0]["logs"][tool]); b = np.array(res[1]["logs"][tool])
        assert np.array_equal(a, b) and np.allclose(a[:, 1], 7.0, rtol=1e-12)


def test_programming_error_on_one_rank_does_not_hang_the_others(tmp_path):
    """A TypeError inside one batch (here: from the mesh provider) is recorded as a failed batch, the rank still reaches the
    collectives, and only then raises it (a rank that raised at once left the others in the all-reduce)."""
    res = _run_two_ranks(tmp_path, "static", "type_error")
    raised = [r["raised"] for r in sorted(res, key=lambda r: r["rank"])]
    assert raised == [None, "TypeError"]          # batch 3 belongs to rank 1
    n_nan = 0
    for tool in res[0]["logs"]:                   # the logs were combined before the error was raised
        a = np.array(res[0]["logs"][tool]); b = np.array(res[1]["logs"][tool])
        assert np.array_equal(np.isnan(a), np.isnan(b))
        good = ~np.isnan(a[:, 1])
        n_nan += int((~good).sum())
        assert np.allclose(a[good, 1], 7.0, rtol=1e-12) and np.allclose(b[good, 1], 7.0, rtol=1e-12)
    assert 0 < n_nan <= 4 * 3                     # exactly the records of the failed batch
