"""Every collective of the multi-GPU paths through RCCL on ONE MI355X.

The builder has one GPU at a time, so nothing here can show scaling; what it shows is that the RCCL calls of
parallel.ShardedPipeline / ShardedTiles / predict_sharded / run_gp_sharded / NNTrainer work on the device -- tensors on
the right device, the collective ordered with the kernels around it on the stream the caller names -- and that the results
have the bits of the no-group path.  The work is done in a FRESH child process (tests/rccl_world1_child.py): the process
group must be created before anything else touches the GPU, and a process that holds the GPU is never re-executed."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_every_collective_on_rccl_at_world_one(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "report.json"
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DIG_FORCE_COLLECTIVES", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py"), str(port), str(out)], env=env,
                         capture_output=True, text=True, timeout=850)
    tail = (res.stdout[-3000:] + "\n" + res.stderr[-3000:])
    assert out.exists(), "the child did not finish:\n" + tail
    report = json.load(open(out))
    print(json.dumps(report))
    assert report["backend"] == "nccl" and report["world"] == 1
    bad = [k for k, v in report["checks"].items() if not v]
    assert not bad and res.returncode == 0, (bad, tail)
    # which way the GP comparison went: on MI355X two fits of the same data in one process give the same bits
    # (the fit has no atomics on its path), so the RCCL path is compared bit for bit -- the spread comparison of the child is the
    # fall-back for a device where that does not hold, and this assertion says it was not needed here
    assert report["gp_fit_is_deterministic"] is True, report.get("gp_forced_vs_plain_max_abs")
    expected = {"ShardedPipeline.step(stream=side)", "ScaleFactorPlan.run_sharded", "ShardedTiles.run", "ShardedTiles.q_values",
                "ShardedTiles.q_values_all",
                "predict_sharded", "run_gp_sharded", "standardisation_stats", "average_gradients", "NNTrainer.train",
                "gather_visiting_order", "broadcast_module_buffers", "all_gather_rows", "gather_to_rank0", "rank_ordered_sum"}
    assert expected <= set(report["checks"])
