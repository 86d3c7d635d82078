"""
File-system rendezvous of the ranks of ONE launch (`shard="env"`: torchrun / mpirun start one process per GPU).

Before a communicator exists the ranks can only meet through files, and files outlive crashed launches: a marker named after
MASTER_PORT alone is found again by the next launch on the same port, whose ranks then map output files that rank 0 is about to
re-create, or hand a dead RCCL id to ncclCommInitRank.  So every launch first agrees on a NONCE no earlier launch
can have produced, and every later marker carries it:

  hello_<r>   rank r's own random token (rewritten by every launch)
  go          written by rank 0: its job nonce J + the tokens it read -- a rank accepts J only if ITS token is in there, i.e. only
              from a rank 0 that has seen THIS process; rank 0 rewrites `go` whenever a hello changes
  ack_<r>     J, once rank r has accepted it; rank 0 returns when every ack says J (a leftover ack holds an older J)

After that `publish / wait / barrier` work on files named `.nellie_<J>_<name>`, invisible to any other launch.  The reference has
no multi-process mode (SURVEY.md 8(e) is this build's design), so there is no reference behaviour to mirror here.
"""
from __future__ import annotations

import glob
import os
import time
import uuid

_CACHE = {}
_POLL_S = 0.01


def _atomic_write(path: str, data: bytes):
    tmp = f"{path}.tmp{os.getpid()}_{uuid.uuid4().hex[:8]}"
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)                  # the file appears complete or not at all


def _read(path: str):
    try:
        with open(path, "rb") as f:
            return f.read()
    except OSError:
        return None


class FileRendezvous:
    def __init__(self, rank: int, world: int, directory: str, tag: str = "", timeout_s: float = 600.0):
        self.rank, self.world, self.dir, self.tag, self.timeout_s = int(rank), int(world), os.path.abspath(directory), str(tag or "job"), timeout_s
        os.makedirs(self.dir, exist_ok=True)
        self.nonce = self._handshake()
        self._mine = []

    # ------------------------------------------------------------------ the handshake
    def _hs(self, name):
        return os.path.join(self.dir, f".nellie_rdv_{self.tag}_{self.world}_{name}")

    def _handshake(self) -> str:
        token = uuid.uuid4().hex
        _atomic_write(self._hs(f"hello_{self.rank}"), token.encode())
        t0 = time.time()
        if self.rank == 0:
            job = uuid.uuid4().hex[:16]
            last = None
            while True:
                tokens = [(_read(self._hs(f"hello_{r}")) or b"").decode() for r in range(self.world)]
                if all(tokens) and tokens != last:
                    _atomic_write(self._hs("go"), "\n".join([job] + tokens).encode())
                    last = tokens
                if last is not None and all((_read(self._hs(f"ack_{r}")) or b"").decode() == job for r in range(1, self.world)):
                    break
                if time.time() - t0 > self.timeout_s:
                    raise TimeoutError(f"rank 0: the other {self.world - 1} ranks did not answer in {self.dir} (tag {self.tag})")
                time.sleep(_POLL_S)
            for r in range(self.world):                        # everybody has read `go` (an ack says so): the handshake files can go
                for n in (f"hello_{r}", f"ack_{r}"):
                    try:
                        os.remove(self._hs(n))
                    except OSError:
                        pass
            try:
                os.remove(self._hs("go"))
            except OSError:
                pass
            # leftovers of launches that died: markers of any other nonce older than a week are litter (a launch that is still running
            # after a week keeps working: its ranks only read markers while a stage starts or ends, minutes after they were written)
            for p in glob.glob(os.path.join(self.dir, ".nellie_*")):
                try:
                    if f"_{job}_" not in p and time.time() - os.path.getmtime(p) > 7 * 86400:
                        os.remove(p)
                except OSError:
                    pass
            return job
        while True:
            lines = (_read(self._hs("go")) or b"").decode().split("\n")
            if len(lines) == self.world + 1 and lines[1 + self.rank] == token:
                _atomic_write(self._hs(f"ack_{self.rank}"), lines[0].encode())
                return lines[0]
            if time.time() - t0 > self.timeout_s:
                raise TimeoutError(f"rank {self.rank}: no answer from rank 0 in {self.dir} (tag {self.tag})")
            time.sleep(_POLL_S)

    # ------------------------------------------------------------------ markers of this launch
    def path(self, name: str) -> str:
        return os.path.join(self.dir, f".nellie_{self.nonce}_{name}")

    def publish(self, name: str, payload: bytes = b"1"):
        _atomic_write(self.path(name), payload)
        self._mine.append(self.path(name))

    def wait(self, name: str, timeout_s: float = None) -> bytes:
        t0, limit = time.time(), self.timeout_s if timeout_s is None else timeout_s
        while True:
            data = _read(self.path(name))
            if data is not None:
                return data
            if time.time() - t0 > limit:
                raise TimeoutError(f"rank {self.rank}: timed out waiting for {self.path(name)}")
            time.sleep(_POLL_S)

    def remove(self, name: str):
        try:
            os.remove(self.path(name))
        except OSError:
            pass

    def barrier(self, name: str):
        """Every rank has reached this point when any rank returns.  Two rounds, so that rank 0 can clean up: the `a` files stay
        until every rank has written its `b` file, i.e. has finished looking at the `a` files."""
        self.publish(f"{name}_a_{self.rank}")
        for r in range(self.world):
            self.wait(f"{name}_a_{r}")
        self.publish(f"{name}_b_{self.rank}")
        if self.rank == 0:
            for r in range(self.world):
                self.wait(f"{name}_b_{r}")
            for r in range(self.world):
                self.remove(f"{name}_a_{r}")
                self.remove(f"{name}_b_{r}")


def rendezvous_for(spec, directory=None) -> FileRendezvous:
    """The rendezvous of this launch (one handshake per process and (directory, tag, world); later stages reuse its nonce)."""
    d = os.path.abspath(os.environ.get("NELLIE_RENDEZVOUS_DIR") or directory or getattr(spec, "rendezvous_dir", None) or os.getcwd())
    key = (d, spec.tag, spec.world, spec.rank)
    if key not in _CACHE:
        _CACHE[key] = FileRendezvous(spec.rank, spec.world, d, spec.tag)
    return _CACHE[key]
