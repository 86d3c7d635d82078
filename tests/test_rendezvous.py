"""The file rendezvous of a multi-process launch (nellie_amd/rendezvous.py) on CPU: ranks agree on a nonce no earlier launch can
have produced, so the leftovers of a launch that died on the same MASTER_PORT -- ready / done markers, a communicator id -- are
never picked up."""
import multiprocessing as mp
import os
import sys
import time

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _rank(rank, world, d, tag, delay, q):
    try:
        sys.path.insert(0, REPO)
        from nellie_amd.rendezvous import FileRendezvous
        time.sleep(delay)
        rdv = FileRendezvous(rank, world, d, tag, timeout_s=60)
        if rank == 0:
            rdv.publish("payload", b"id-of-this-launch")
        data = rdv.wait("payload", 60)
        rdv.barrier("end")
        q.put((rank, rdv.nonce, data))
    except BaseException as exc:  # noqa: BLE001
        q.put((rank, "error", repr(exc)))


def _launch(world, d, tag, delays):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, world, d, tag, delays[r], q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(30)
    return out


def test_ranks_agree_on_a_fresh_nonce_despite_leftovers(tmp_path):
    d, tag, world = str(tmp_path), "29500", 3
    # what a launch that died leaves behind: its handshake files (complete, consistent) and markers under its nonce
    stale = "deadbeefdeadbeef"

    def leave_a_dead_launch_behind():
        for r in range(world):
            open(os.path.join(d, f".nellie_rdv_{tag}_{world}_hello_{r}"), "w").write(f"tok{r}")
            open(os.path.join(d, f".nellie_rdv_{tag}_{world}_ack_{r}"), "w").write(stale)
        open(os.path.join(d, f".nellie_rdv_{tag}_{world}_go"), "w").write("\n".join([stale] + [f"tok{r}" for r in range(world)]))
        open(os.path.join(d, f".nellie_{stale}_payload"), "w").write("id-of-the-dead-launch")
    # rank 0 late: the others meet the stale `go` first; then rank 2 late: rank 0 meets a stale hello first
    for delays in ((0.5, 0.0, 0.0), (0.0, 0.0, 0.5)):
        leave_a_dead_launch_behind()
        out = _launch(world, d, tag, delays)
        assert all(o[1] != "error" for o in out), out
        nonces = {o[1] for o in out}
        assert len(nonces) == 1 and stale not in nonces
        assert all(o[2] == b"id-of-this-launch" for o in out)
    # two launches in a row never share a nonce, and a finished launch leaves no barrier / handshake files behind
    left = [f for f in os.listdir(d) if "_end." in f or f.startswith(f".nellie_rdv_{tag}")]
    assert left == [], left


def test_world_one_needs_nobody(tmp_path):
    from nellie_amd.rendezvous import FileRendezvous
    rdv = FileRendezvous(0, 1, str(tmp_path), "x", timeout_s=5)
    rdv.publish("a", b"z")
    assert rdv.wait("a", 1) == b"z"
    rdv.barrier("b")
    rdv.remove("a")
    assert [f for f in os.listdir(tmp_path) if f.startswith(".nellie")] == []


def test_a_name_can_be_used_again_within_one_launch(tmp_path):
    """a process that runs several files through run() / run_streamed() / Markers.run() reuses the fixed names
    ("im_info_built", "streamed_done", "markers_files_ready", ...).  Non-zero ranks leave a barrier right after their `b` file while
    rank 0 deletes the files later: without a generation per use, rank 1 passes the second barrier on rank 0's stale `a` file and
    rank 0 then deletes rank 1's new files -- both time out.  Also the publish / wait / remove cycle of a marker."""
    import threading
    from nellie_amd.rendezvous import FileRendezvous
    world, errs, made = 3, [], {}

    def rank(r):
        try:
            rdv = FileRendezvous(r, world, str(tmp_path), "again", timeout_s=20)
            made[r] = rdv
            for k in range(25):
                if r == 0:
                    rdv.publish("files_ready", str(k).encode())
                else:
                    assert rdv.wait("files_ready") == str(k).encode(), "a marker of an earlier use was picked up"
                rdv.barrier("same")
                if r == 0:
                    rdv.remove("files_ready")
                rdv.barrier("same")
        except BaseException as exc:  # noqa: BLE001
            errs.append((r, repr(exc)))

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    assert len({m.nonce for m in made.values()}) == 1
    assert [f for f in os.listdir(tmp_path) if f.startswith(".nellie")] == []


def test_the_litter_sweep_only_touches_this_modules_files(tmp_path):
    """Rank 0 deletes week-old leftovers of dead launches -- and nothing else that happens to start with `.nellie_`."""
    from nellie_amd.rendezvous import FileRendezvous
    old = time.time() - 8 * 86400
    names = {".nellie_deadbeefdeadbeef_payload.g1": False, ".nellie_rdv_29500_2_hello_1": False,       # ours, stale: swept
             ".nellie_notes.txt": True, ".nellie_settings": True, ".nellie_DEADBEEFDEADBEEF_x": True}   # somebody else's: kept
    for n in names:
        p = os.path.join(str(tmp_path), n)
        open(p, "w").write("x")
        os.utime(p, (old, old))
    FileRendezvous(0, 1, str(tmp_path), "sweep", timeout_s=5)
    for n, kept in names.items():
        assert os.path.exists(os.path.join(str(tmp_path), n)) == kept, n


def _build_rank(rank, world, src, out_dir, port, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_PORT=str(port))
        from nellie_amd.im_info.verifier import FileInfo
        from nellie_amd.run import _build_im_info
        im = _build_im_info(FileInfo(src, output_dir=out_dir), "env")
        import numpy as np
        q.put((rank, im.im_path, int(np.asarray(im.im).sum()), os.stat(im.im_path).st_ino))
    except BaseException as exc:  # noqa: BLE001
        q.put((rank, "error", repr(exc), 0))


def test_ranks_build_the_im_info_in_order(tmp_path):
    """run(file_info, shard="env"): every rank constructs ImInfo (run.py:49), which re-saves the input as the canonical copy
    (verifier.py:620-695).  Rank 0 goes first and the others reuse its file: one inode, never a half-written header."""
    import numpy as np
    from nellie_amd.im_info import ome_tiff
    vols = np.arange(2 * 6 * 10 * 12, dtype=np.uint16).reshape(2, 6, 10, 12)
    src = str(tmp_path / "stack.ome.tif")
    ome_tiff.create(src, vols.shape, np.uint16, {"X": 0.1, "Y": 0.1, "Z": 0.2, "T": 1.0}, "raw", data=vols)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_build_rank, args=(r, world, src, str(tmp_path / "out"), 29611, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(30)
    assert all(o[1] != "error" for o in out), out
    assert len({o[1] for o in out}) == 1 and len({o[3] for o in out}) == 1, out      # same path, same inode: written once
    assert all(o[2] == int(vols.sum()) for o in out)


def test_a_wait_that_timed_out_can_be_retried(tmp_path):
    """wait() moves a name's generation only when it SUCCEEDED -- a TimeoutError followed by a retry polls the generation it
    timed out on, not the next one."""
    import threading
    from nellie_amd.rendezvous import FileRendezvous
    rdvs = [None, None]

    def make(r):
        rdvs[r] = FileRendezvous(r, 2, str(tmp_path), tag="retry", timeout_s=20.0, poll_s=0.001)
    ts = [threading.Thread(target=make, args=(r,)) for r in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    with pytest.raises(TimeoutError):
        rdvs[1].wait("late", timeout_s=0.05)
    rdvs[0].publish("late", b"now")
    assert rdvs[1].wait("late", timeout_s=5.0) == b"now"
    rdvs[0].remove("late")
    rdvs[0].publish("late", b"again")                      # the second use of the name: the next generation on both sides
    assert rdvs[1].wait("late", timeout_s=5.0) == b"again"
