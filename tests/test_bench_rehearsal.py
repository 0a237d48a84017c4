"""bench.py's N > 1 control flow rehearsed on ONE GPU: two ranks launched the way the driver launches
them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), torch.distributed on gloo, both ranks on device 0
(FS_BENCH_SAME_DEVICE), the library's collectives through the shared-memory RCCL test double (tests/fake_rccl.cpp, real
RCCL refuses two ranks on one device).  What it protects: the weak-scaled headline path with the library's all-reduce, and
the two extra regions that time BASELINE.json's named multi-GPU configurations (cfg4: one 1 048 576-ray depth-12 frame
sharded over the ranks; cfg5: 8 sources round-robin + the all-gather) — every rank must take the same collective steps, or
a real 8-GPU run hangs."""
import json
import os
import socket
import subprocess
import sys

import pytest

from test_two_ranks_one_gpu import fake_rccl  # noqa: F401  (fixture)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu(fake_rccl, tmp_path):  # noqa: F811
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FS_BENCH_BACKEND="gloo", FS_BENCH_SAME_DEVICE="1", FS_RCCL_LIB=fake_rccl, FAKE_RCCL_TIMEOUT_S="120")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                                       "--prewarm", "6", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, cwd=ROOT))
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=420))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r}:\n{outs[r][1][-3000:]}"
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(line) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # rank 0 alone prints, ONE line
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["steps"] == 6 and res["scaling"] == "weak" and res["value"] > 0
    assert res["config"]["rays_per_frame"] == 2 * 262144 and "all-reduce" in res["config"]["collective"]
    # round 5: the line proves its N by itself (fs_comm_info asks the communicator) and carries the timed region's timeline
    assert res["config"]["rccl_ranks"] == 2 and res["config"]["rccl_rank"] == 0 and res["config"]["collective_kind"] == "ncclAllReduce"
    tl = res["timeline"]
    assert len(tl["step_host_ms"]) == 6 and tl["library"]["publishes_by_word"] + tl["library"]["publishes_by_event"] == 6
    ex = res["extra"]
    assert ex["cfg4_old_mine_1m_d12"]["value"] > 0 and ex["cfg4_old_mine_1m_d12"]["scaling"] == "strong"
    assert ex["cfg5_multi_source"]["value"] > 0 and ex["cfg5_multi_source"]["sources_per_rank"] == 4
    assert "all-gather" in ex["cfg5_multi_source"]["collective"]
