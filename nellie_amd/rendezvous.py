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
import re
import time
import uuid

_CACHE = {}
_POLL_S = 0.01
_OURS = re.compile(r"^\.nellie_(rdv_.+|[0-9a-f]{16}_.+)$")


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
    def __init__(self, rank: int, world: int, directory: str, tag: str = "", timeout_s: float = 600.0, poll_s: float = None):
        self.rank, self.world, self.dir, self.tag, self.timeout_s = int(rank), int(world), os.path.abspath(directory), str(tag or "job"), timeout_s
        self.poll_s = _POLL_S if poll_s is None else float(poll_s)      # (a benchmark that brackets a timed region with barriers polls faster)
        os.makedirs(self.dir, exist_ok=True)
        self.nonce = self._handshake()
        self._mine = []
        # A name may be used again within one launch (a process that runs several files through `run()` reuses "im_info_built",
        # "streamed_done", ...): every use of a name gets the next GENERATION, counted per name on every rank alike -- each rank either
        # publishes or waits for a marker once per use, and takes part in every barrier -- so a marker or barrier file of use k is
        # never mistaken for one of use k+1, and rank 0's late clean-up of use k cannot delete a file of use k+1.
        self._gen = {}
        self._unwaited = set()             # names this rank published in their current generation and has not waited for itself

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
                time.sleep(self.poll_s)
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
            # (only names this module writes: `.nellie_<16 hex>_...` markers and `.nellie_rdv_...` handshake files -- the directory is
            # usually the user's data directory)
            for p in glob.glob(os.path.join(self.dir, ".nellie_*")):
                try:
                    if _OURS.match(os.path.basename(p)) and f"_{job}_" not in p and time.time() - os.path.getmtime(p) > 7 * 86400:
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
            time.sleep(self.poll_s)

    # ------------------------------------------------------------------ markers of this launch
    def path(self, name: str) -> str:
        """The marker file of `name` in its current generation (what the last publish / wait of this rank used)."""
        return self._path_g(name, self._gen.get(name, 0))

    def _path_g(self, name: str, gen: int) -> str:
        return os.path.join(self.dir, f".nellie_{self.nonce}_{name}.g{gen}")

    def _next(self, name: str) -> int:
        self._gen[name] = self._gen.get(name, 0) + 1
        return self._gen[name]

    def publish(self, name: str, payload: bytes = b"1"):
        p = self._path_g(name, self._next(name))
        self._unwaited.add(name)
        _atomic_write(p, payload)
        self._mine.append(p)

    def _wait_path(self, p: str, timeout_s: float = None) -> bytes:
        t0, limit = time.time(), self.timeout_s if timeout_s is None else timeout_s
        while True:
            data = _read(p)
            if data is not None:
                return data
            if time.time() - t0 > limit:
                raise TimeoutError(f"rank {self.rank}: timed out waiting for {p}")
            time.sleep(self.poll_s)

    def wait(self, name: str, timeout_s: float = None) -> bytes:
        """The payload of `name`'s next generation.  One publisher per name (rank 0 in every use of this package).  The generation counter
        moves only when the wait has SUCCEEDED: a TimeoutError followed by a retry polls the same generation again."""
        if name in self._unwaited:         # the publisher reading its own marker: the same use
            data = self._wait_path(self._path_g(name, self._gen[name]), timeout_s)
            self._unwaited.discard(name)
            return data
        gen = self._gen.get(name, 0) + 1
        data = self._wait_path(self._path_g(name, gen), timeout_s)
        self._gen[name] = gen
        return data

    def remove(self, name: str):
        """Removes the marker of `name`'s current generation (the publisher's job, after a barrier)."""
        self._unwaited.discard(name)
        try:
            os.remove(self.path(name))
        except OSError:
            pass

    def allgather(self, name: str, payload: bytes):
        """Every rank's payload, in rank order (each rank publishes `<name>_<rank>` and reads the others'); the files of a use are
        removed by their owners after a barrier, so a name can be used again."""
        gen = self._next("gather:" + name)
        f = lambda r: os.path.join(self.dir, f".nellie_{self.nonce}_{name}.g{gen}_v_{r}")
        _atomic_write(f(self.rank), payload)
        out = [self._wait_path(f(r)) for r in range(self.world)]
        self.barrier("gather_done:" + name)
        try:
            os.remove(f(self.rank))
        except OSError:
            pass
        return out

    def barrier(self, name: str):
        """Every rank has reached this point when any rank returns.  Two rounds, so that rank 0 can clean up: the `a` files stay
        until every rank has written its `b` file, i.e. has finished looking at the `a` files.  The files carry the generation of
        this use of `name`, so the next barrier of the same name starts from files nobody has written yet."""
        gen = self._next("barrier:" + name)
        self._unwaited.clear()             # a marker published before a barrier has been seen by whoever waits for it: never "own, unread" afterwards
        f = lambda ab, r: os.path.join(self.dir, f".nellie_{self.nonce}_{name}.g{gen}_{ab}_{r}")
        _atomic_write(f("a", self.rank), b"1")
        for r in range(self.world):
            self._wait_path(f("a", r))
        _atomic_write(f("b", self.rank), b"1")
        if self.rank == 0:
            for r in range(self.world):
                self._wait_path(f("b", r))
            for r in range(self.world):
                for ab in ("a", "b"):
                    try:
                        os.remove(f(ab, r))
                    except OSError:
                        pass


def rendezvous_for(spec, directory=None) -> FileRendezvous:
    """The rendezvous of this launch (one handshake per process and (directory, tag, world); later stages reuse its nonce)."""
    d = os.path.abspath(os.environ.get("NELLIE_RENDEZVOUS_DIR") or directory or getattr(spec, "rendezvous_dir", None) or os.getcwd())
    key = (d, spec.tag, spec.world, spec.rank)
    if key not in _CACHE:
        _CACHE[key] = FileRendezvous(spec.rank, spec.world, d, spec.tag)
    return _CACHE[key]
