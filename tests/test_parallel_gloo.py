"""CPU, world_size 2, gloo: sharding and the per-lap exchange give identical, deterministic results on all ranks."""
import os
import subprocess

import pytest
import sys

import numpy as np

from tests import common

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from racinglmpc_amd import parallel
from tests.gloo_comm import GlooComm
comm = GlooComm()
rank, world = comm.rank, comm.world
lo, hi = parallel.shard(11, rank, world)
rng = np.random.default_rng(100 + rank)
laps = []
for i in range(lo, hi):
    T = 120 + (7 * i) % 23                       # deterministic lap lengths, some ties across ranks
    laps.append((rng.normal(size=(T, 6)) + i, rng.normal(size=(T, 2)), rng.normal(size=(T, 6)), np.arange(12.0) + i))
best = parallel.exchange_laps(laps, K=4, T_max=160, comm=comm)
rec, ln = parallel.pack_laps(laps, 4, 160)
recs = comm.allgather(rec)
# rows owned by different ranks (K = 4 parent laps continued by global rollouts 0..3: all on rank 0 here; 7..10 on rank 1)
buf = np.zeros((4, 3, 8)); mask = np.zeros(4, dtype=np.int64)
for k in range(4):
    gidx = 2 + 3 * k                              # global rollout index that owns row k: 2, 5 on rank 0; 8 on rank 1; 11 nobody
    if lo <= gidx < hi:
        buf[k] = gidx; mask[k] = 1
rows, owned = parallel.gather_owned_rows(buf, mask, comm)
mx = comm.allreduce_max(float(rank + 1))[0]
np.savez(os.path.join(sys.argv[2], "rank%d.npz" % rank), lens=np.array([b[4] for b in best]), src=np.array([b[3] for b in best]),
         x0=np.array([b[0][0, 0] for b in best]), chk=np.array([b[0].sum() + b[1].sum() + b[2].sum() for b in best]), mx=mx, lo=lo, hi=hi,
         extra=np.array([b[5] for b in best]), recs=recs, rows=rows, owned=owned)
comm.close()
'''


def test_shard_covers_everything():
    from racinglmpc_amd import parallel
    for total in (0, 1, 7, 8, 8192, 8191):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_exchange_single_process():
    from racinglmpc_amd import parallel
    assert parallel.LocalComm().allgather(np.arange(3.0)).shape == (1, 3)
    rng = np.random.default_rng(0)
    laps = [(rng.normal(size=(T, 6)), rng.normal(size=(T, 2)), rng.normal(size=(T, 6))) for T in (50, 40, 45, 40)]
    best = parallel.exchange_laps(laps, K=3, T_max=64)
    assert [b[4] for b in best] == [40, 40, 45]
    assert np.array_equal(best[0][0], laps[1][0]) and np.array_equal(best[1][0], laps[3][0])      # ties: lower local index first


def test_exchange_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", str(script), common.ROOT, str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    for k in ("lens", "src", "x0", "chk", "mx", "extra", "recs", "rows", "owned"):
        assert np.array_equal(r0[k], r1[k]), k
    assert float(r0["mx"]) == 2.0
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 6, 6, 11)
    lens_all = sorted(120 + (7 * i) % 23 for i in range(11))
    assert list(r0["lens"]) == lens_all[:4]
    # record layout: (world, K, T_max + 1, 14); the extra row carries the 12 finish-line values and the local index
    assert r0["recs"].shape == (2, 4, 161, 14)
    assert np.all(r0["extra"][:, 12] == np.round(r0["extra"][:, 12])) and np.all(r0["extra"][:, 13] == 0)
    # owner gather: rows 0, 1 from rank 0, row 2 from rank 1, row 3 owned by nobody
    assert list(r0["owned"]) == [True, True, True, False]
    assert np.all(r0["rows"][0] == 2) and np.all(r0["rows"][1] == 5) and np.all(r0["rows"][2] == 8) and np.all(r0["rows"][3] == 0)


RDZV_WORKER = r'''
import sys
sys.path.insert(0, sys.argv[1])
from racinglmpc_amd import parallel
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
uid = parallel._rendezvous_id(rank, world, "127.0.0.1", port, lambda: bytes(range(128)))
assert uid == bytes(range(128)), uid
print("ok", rank)
'''


@pytest.mark.parametrize("tcp", ["0", "1"])
def test_unique_id_rendezvous_three_processes(tmp_path, tcp):
    """The hand-off of the RCCL unique id (rank 0 creates, the others fetch), without a GPU: three processes, any start order; through the
    same-node file (default for a local MASTER_ADDR) and over TCP (other nodes; forced here with LMPC_RDZV_TCP=1)."""
    script = tmp_path / "rdzv.py"
    script.write_text(RDZV_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), common.ROOT, str(r), "3", "2968%s" % (9 if tcp == "1" else 7)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              env=dict(os.environ, LMPC_RDZV_TCP=tcp)) for r in (2, 1, 0)]
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and out.decode().startswith("ok"), err.decode()


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set),
    both reach the rendezvous, and the line says n_gpus = 2 (--dry-run: launch path only, no GPU); a WORLD_SIZE that contradicts
    --gpus is an error."""
    import json
    out = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--dry-run"], check=True, capture_output=True, timeout=300)
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["ranks_at_rendezvous"] == 2
    assert line["host_wait"] == "file"                     # the ranks without extras sleep on a host-side event, not inside a collective
    assert set(line["multi_rank_keys"]) >= {"per_rank_ms_per_step", "barrier_us", "rccl_ranks", "exchange_seconds"}
    bad = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, timeout=300,
                         env=dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"))
    assert bad.returncode != 0 and b"WORLD_SIZE=1" in bad.stderr


def test_rendezvous_rejects_a_stale_id_file(tmp_path, monkeypatch):
    """A killed launch can leave its id file behind (same parent process, same port).  The file carries its writer's pid: a rank that starts
    before rank 0 has replaced it must not take the dead writer's id (ncclCommInitRank would hang), and it is created 0600 in a 0700 directory."""
    import stat
    import time
    from racinglmpc_amd import parallel
    port = 29685
    monkeypatch.setattr(os, "getppid", os.getpid)                  # the workers' parent is this process
    monkeypatch.delenv("LMPC_RDZV_NONCE", raising=False); monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    path = parallel._rdzv_file(port)
    dead = subprocess.Popen([sys.executable, "-c", "pass"]); dead.wait()
    with open(path, "wb") as f:
        f.write(b"\xff" * 128 + int(dead.pid).to_bytes(8, "little"))
    assert stat.S_IMODE(os.stat(os.path.dirname(path)).st_mode) == 0o700
    script = tmp_path / "rdzv.py"
    script.write_text(RDZV_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("LMPC_RDZV_NONCE", "TORCHELASTIC_RUN_ID")}
    late = [subprocess.Popen([sys.executable, str(script), common.ROOT, "1", "2", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)]
    time.sleep(1.5)                                                # rank 1 is polling the stale file by now
    assert late[0].poll() is None
    late.append(subprocess.Popen([sys.executable, str(script), common.ROOT, "0", "2", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
    for p in late:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and out.decode().startswith("ok"), err.decode()
    assert not os.path.exists(path)
    with pytest.raises(ValueError):
        parallel._rendezvous_id(0, 2, "127.0.0.1", 70000, lambda: bytes(128))
    monkeypatch.setenv("MASTER_PORT", "65500"); monkeypatch.delenv("LMPC_RDZV_PORT", raising=False)
    assert 0 < parallel.env_world()[4] < 65536


def test_host_wait_ignores_a_dead_writers_event(tmp_path, monkeypatch):
    """The host-side event the idle ranks sleep on (bench.py: rank 0 runs its extra configurations alone) carries its writer's pid: an event file
    left behind by a killed launch with the same parent, port and nonce does not release the waiters, and a signaller that died is an error, not
    an hour of sleep."""
    import threading
    import time
    from racinglmpc_amd import parallel
    port = 29687
    monkeypatch.setattr(os, "getppid", os.getpid)
    path = parallel._rdzv_file(port) + ".extras"
    dead = subprocess.Popen([sys.executable, "-c", "pass"]); dead.wait()
    with open(path, "wb") as f:
        f.write(int(dead.pid).to_bytes(8, "little"))
    with pytest.raises(TimeoutError):
        parallel.host_wait("extras", port, timeout=0.5)
    assert os.path.exists(path)                                      # ignored, not removed: a waiter that unlinked it could delete the live writer's fresh signal (host_signal renames it into place)
    with pytest.raises(RuntimeError):
        parallel.host_wait("extras", port, timeout=30.0, writer_pid=dead.pid)
    threading.Timer(0.3, lambda: parallel.host_signal("extras", port)).start()
    t0 = time.time()
    assert parallel.host_wait("extras", port, timeout=30.0) == path and time.time() - t0 < 5.0
    os.remove(path)
