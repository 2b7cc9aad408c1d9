"""Multi-GPU layer: one process per GPU, problems sharded by contiguous blocks, ONE exchange step per lap.

The QPs / rollouts of a batch are independent given a read-only safe set, so the data path needs no
collective; the only exchange is after a lap: every rank contributes its K fastest finished rollouts as fixed-stride
padded records, one all-gather makes the union visible everywhere, and every rank runs the same deterministic top-K
selection and the same addTrajectory inserts, which leaves identical lap stores on all ranks.

The collective is RCCL (ncclAllGather over xGMI) called from liblmpc_hip.so (include/lmpc_hip.h: lmpc_comm_*,
lmpc_rollout_exchange): the records are packed on the device from the rollout session's logs and never staged through the
host.  This module holds what surrounds it: the rendezvous (rank 0 creates the 128-byte RCCL id and hands it to the other
processes over TCP on the launcher's MASTER_ADDR), the sharding rule and the deterministic top-K.  A communicator is any
object with rank / world / allgather / allreduce_max / barrier: `RcclComm` (GPUs), `LocalComm` (one process); the CPU tests
supply a gloo-backed one of their own.
"""
import os
import socket
import time

import numpy as np

REC_COLS = 14            # x (6) | u (2) | x_glob (6)


def shard(total, rank, world):
    """Contiguous block of `total` items owned by `rank` (first `total % world` ranks get one more)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# ---------------------------------------------------------------------------------------------- communicators
class LocalComm:
    """Single process."""
    rank, world, backend = 0, 1, "local"

    def allgather(self, arr):
        return np.array(arr)[None]

    def allreduce_max(self, values):
        return np.atleast_1d(np.asarray(values, dtype=np.float64)).copy()

    def barrier(self):
        pass

    def close(self):
        pass


def _rdzv_file(port):
    """Same-node hand-off file: every rank of one launch has the same parent process (the launcher's agent, or bench.py --gpus N itself).
    It lives in a directory of the user's own (mode 0700: the RCCL id is not world-readable) and carries the launch's nonce if the launcher
    supplies one (LMPC_RDZV_NONCE, or torchrun's TORCHELASTIC_RUN_ID)."""
    import tempfile
    d = os.path.join(tempfile.gettempdir(), "lmpc_rdzv_u%d" % os.getuid())
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
        if os.stat(d).st_uid != os.getuid():
            raise OSError("rendezvous directory %s belongs to another user" % d)
        os.chmod(d, 0o700)
    except OSError:
        d = tempfile.gettempdir()                                   # (no private directory: the file itself is still created 0600 / O_EXCL)
    nonce = os.environ.get("LMPC_RDZV_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or "none"
    nonce = "".join(ch for ch in nonce if ch.isalnum() or ch in "-_")[:48]
    return os.path.join(d, "lmpc_rdzv_%d_%d_%s.id" % (os.getppid(), int(port), nonce))


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    return True


def host_signal(tag, port):
    """Same-node host-side event (a file next to the rendezvous file): ranks that have nothing to do on their GPU while another rank works
    alone wait for it in host_wait -- sleeping -- instead of spinning inside a collective for seconds."""
    path = _rdzv_file(port) + "." + "".join(ch for ch in tag if ch.isalnum())
    tmp = path + ".tmp%d" % os.getpid()                            # written aside and renamed into place: a waiter sees the whole payload or the old file
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(int(os.getpid()).to_bytes(8, "little"))
    os.replace(tmp, path)
    return path


def host_wait(tag, port, timeout=3600.0, poll=0.05, writer_pid=None):
    """Sleep until host_signal(tag, port) has been called by a LIVE process of this launch.  The event file carries its writer's pid: a file
    whose writer is gone -- left behind by a killed launch with the same parent, port and nonce -- is ignored, so it cannot release
    the waiters early; writer_pid (the signalling rank's pid, if the caller knows it) turns a dead signaller into an error instead of a sleep
    until the timeout."""
    path = _rdzv_file(port) + "." + "".join(ch for ch in tag if ch.isalnum())
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                buf = f.read()
            if len(buf) == 8 and _alive(int.from_bytes(buf, "little")):
                return path
            # (a file whose writer is dead is stale: ignored, NOT removed -- a waiter that unlinked it could delete the fresh signal the live writer has
            #  just renamed into place; the signaller of a launch removes left-overs itself before it starts, bench.py)
        except OSError:
            pass
        if writer_pid is not None and not _alive(int(writer_pid)):
            raise RuntimeError("host_wait(%s): the signalling process %d is gone" % (tag, writer_pid))
        if time.time() - t0 > timeout:
            raise TimeoutError("host_wait(%s): no signal within %.0f s" % (tag, timeout))
        time.sleep(poll)


def _is_local(addr):
    if os.environ.get("LMPC_RDZV_TCP") == "1":                      # (tests: force the TCP hand-off)
        return False
    return addr in ("127.0.0.1", "localhost", "::1") or addr == socket.gethostname()


def _rendezvous_id(rank, world, addr, port, make_id, timeout=120.0):
    """Rank 0 creates the RCCL unique id and hands it to the world - 1 other ranks: through a file in the temporary directory when the
    ranks share a node (no port to collide on, nothing to resolve; a rank acknowledges with a file of its own), and over TCP (addr, port)
    for ranks on other nodes or without a shared temporary directory.  Rank 0 returns once every other rank has the id."""
    path = _rdzv_file(port)
    if not 0 < int(port) < 65536:
        raise ValueError("rendezvous port %r is not a TCP port (LMPC_RDZV_PORT / MASTER_PORT + 117, see env_world)" % (port,))
    if rank == 0:
        uid = make_id()
        if world == 1:
            return uid
        for f in [path] + [path + ".ack%d" % r for r in range(1, world)]:     # left-overs of a killed launch with the same parent and port
            try:
                os.remove(f)
            except OSError:
                pass
        try:
            # payload: the 128-byte id + this process's pid -- a reader rejects a file whose writer is gone (a stale id would hang ncclCommInitRank)
            tmp = path + ".tmp%d" % os.getpid()
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
            with os.fdopen(fd, "wb") as f:
                f.write(uid + int(os.getpid()).to_bytes(8, "little"))
            os.replace(tmp, path)                                   # atomic: a reader sees the whole payload or no file
        except OSError:
            pass
        try:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port)); srv.listen(world); srv.settimeout(0.05)
        except (OSError, OverflowError, ValueError):
            srv = None                                              # port taken / not bindable: the file is the hand-off
        served, t0 = set(), time.time()
        try:
            while len(served) < world - 1:
                for r in range(1, world):
                    if os.path.exists(path + ".ack%d" % r):
                        served.add(r)
                if srv is not None and len(served) < world - 1:
                    try:
                        conn, _peer = srv.accept()
                        conn.settimeout(5.0); conn.sendall(uid)
                        try:
                            served.add(int(conn.recv(16).decode() or "-1"))      # the peer answers with its rank
                        except (OSError, ValueError):
                            pass
                        conn.close()
                    except socket.timeout:
                        pass
                elif srv is None:
                    time.sleep(0.02)
                if time.time() - t0 > timeout:
                    raise TimeoutError("rendezvous: %d of %d ranks fetched the RCCL id within %.0f s" % (len(served), world - 1, timeout))
        finally:
            if srv is not None:
                srv.close()
            for f in [path] + [path + ".ack%d" % r for r in range(1, world)]:
                try:
                    os.remove(f)
                except OSError:
                    pass
        return uid
    t0 = time.time()
    local = _is_local(addr)
    while True:
        if local and os.path.exists(path):
            try:
                with open(path, "rb") as f:
                    buf = f.read()
            except OSError:
                buf = b""
            if len(buf) == 136 and _alive(int.from_bytes(buf[128:], "little")):
                fd = os.open(path + ".ack%d" % rank, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
                with os.fdopen(fd, "wb") as f:
                    f.write(b"1")
                return buf[:128]
        if not local or time.time() - t0 > 15.0:                   # another node, or no shared temporary directory: ask rank 0
            try:
                s = socket.create_connection((addr, port), timeout=5.0)
                break
            except OSError:
                pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rendezvous: no RCCL id from rank 0 within %.0f s (file %s, tcp %s:%d)" % (timeout, path, addr, port))
        time.sleep(0.02)
    buf = b""
    while len(buf) < 128:
        chunk = s.recv(128 - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous: rank 0 closed the connection early")
        buf += chunk
    s.sendall(str(rank).encode())
    s.close()
    return buf


class RcclComm:
    """RCCL communicator owned by an lmpc context (one per process / GPU)."""
    backend = "rccl"

    def __init__(self, ctx, rank, world, addr="127.0.0.1", port=29617):
        from . import _capi
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        uid = _rendezvous_id(self.rank, self.world, addr, int(port), _capi.comm_unique_id)
        ctx.comm_init(uid, self.rank, self.world)

    def allgather(self, arr):
        return self.ctx.comm_allgather(arr)

    def allreduce_max(self, values):
        return self.ctx.comm_allreduce_max(values)

    def barrier(self):
        self.ctx.comm_barrier()

    def close(self):
        self.ctx.comm_destroy()


def env_world():
    """(rank, world, local_rank, addr, rendezvous port) from the launcher's environment (the driver's distributed launcher or bench.py --gpus)."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", str(rank)))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("LMPC_RDZV_PORT", "0")) or int(os.environ.get("MASTER_PORT", "29500")) + 117   # MASTER_PORT itself belongs to the launcher's store
    if port > 65535:                                                # MASTER_PORT near the top of the range: stay a valid port, still off MASTER_PORT
        port -= 2 * 117
    return rank, world, local, addr, port


def comm_from_env(ctx, force_rccl=False):
    rank, world, _local, addr, port = env_world()
    if world > 1 or force_rccl:
        return RcclComm(ctx, rank, world, addr, port)
    return LocalComm()


# ---------------------------------------------------------------------------------------------- lap records
def pack_laps(laps, K, T_max, ids=None):
    """laps: list of (x (T,6), u (T,2), x_glob (T,6)[, extra (<=12,)]).  Keeps the K shortest (ties: lower local index),
    pads to T_max rows + 1 row for `extra` (the state right after the finish line) and the local index (ids[i] if given -- the rollout's index
    in the rank's shard, which is what the device-side packing writes -- else the position in `laps`).
    Returns (records float64 [K, T_max + 1, 14], lengths int64 [K]); unused slots have length -1.
    Same layout as the device-side packing of lmpc_rollout_exchange (lmpc_comm.hip.h)."""
    order = sorted(range(len(laps)), key=lambda i: (laps[i][0].shape[0], i))[:K]
    rec = np.zeros((K, T_max + 1, REC_COLS)); ln = -np.ones(K, dtype=np.int64)
    for j, i in enumerate(order):
        x, u, xg = laps[i][0], laps[i][1], laps[i][2]
        T = x.shape[0]
        if T > T_max:
            raise ValueError("lap of %d steps exceeds the exchange record size %d" % (T, T_max))
        rec[j, :T, 0:6] = x; rec[j, :T, 6:8] = u; rec[j, :T, 8:14] = xg; ln[j] = T
        if len(laps[i]) > 3:
            e = np.asarray(laps[i][3], float).reshape(-1)
            rec[j, T_max, :e.shape[0]] = e
        rec[j, T_max, 12] = i if ids is None else ids[i]
    return rec, ln


def top_k(recs, lens, K, T_max):
    """Deterministic global selection from gathered records (world, K, T_max + 1, 14) / lengths (world, K): the K shortest laps,
    ordered by (T, source rank, slot) -- identical on every rank.  Returns [(x, u, x_glob, src_rank, T, extra14)]."""
    cand = [(int(lens[r, j]), r, j) for r in range(recs.shape[0]) for j in range(recs.shape[1]) if lens[r, j] >= 0]
    cand.sort()
    out = []
    for T, r, j in cand[:K]:
        out.append((recs[r, j, :T, 0:6].copy(), recs[r, j, :T, 6:8].copy(), recs[r, j, :T, 8:14].copy(), r, T, recs[r, j, T_max].copy()))
    return out


def exchange_laps(laps, K, T_max, comm=None):
    """All-gather every rank's K fastest laps (host-side lists) and return the global K fastest, identical on every rank."""
    comm = comm or LocalComm()
    rec, ln = pack_laps(laps, K, T_max)
    recs = comm.allgather(rec); lens = comm.allgather(ln)
    return top_k(recs, lens, K, T_max)


def gather_owned_rows(buf, mask, comm=None):
    """buf (K, ...) holds this rank's rows where mask[k] is set; every rank receives the union (each row taken from the rank that
    owns it; a row nobody owns stays zero and is reported in the returned `owned` mask)."""
    comm = comm or LocalComm()
    allb = comm.allgather(np.ascontiguousarray(buf, dtype=np.float64)); allm = comm.allgather(np.ascontiguousarray(mask, dtype=np.int64))
    out = np.zeros_like(allb[0]); owned = np.zeros(allm.shape[1], dtype=bool)
    for r in range(allb.shape[0]):
        for k in np.nonzero(allm[r])[0]:
            if not owned[k]:
                out[k] = allb[r, k]; owned[k] = True
    return out, owned
