"""The native backend's rendezvous (hybrid-drt_amd/mapping/dist.py): rank 0's 128-byte RCCL id reaches the other ranks through a
file -- host logic only, no device, no RCCL (a stand-in hands out the id)."""
import os
import threading
import time

import pytest

from hipdrt.mapping import dist


class _Ids:
    calls = 0

    @classmethod
    def comm_unique_id(cls):
        cls.calls += 1
        return bytes((7 * i + cls.calls) % 256 for i in range(128))


def test_world_of_one_needs_no_file(monkeypatch, tmp_path):
    monkeypatch.setenv("HIPDRT_RCCL_ID_FILE", str(tmp_path / "never.id"))
    uid, path = dist._exchange_unique_id(_Ids, 0, 1)
    assert len(uid) == 128 and path is None and not (tmp_path / "never.id").exists()


def test_other_ranks_wait_for_rank_zero_and_read_its_id(monkeypatch, tmp_path):
    path = tmp_path / "launch.id"
    monkeypatch.setenv("HIPDRT_RCCL_ID_FILE", str(path))
    got = {}
    readers = [threading.Thread(target=lambda r=r: got.__setitem__(r, dist._exchange_unique_id(_Ids, r, 3, timeout=20))) for r in (1, 2)]
    for t in readers:
        t.start()
    time.sleep(0.2)                                   # the readers poll a file that is not there yet
    assert not got
    got[0] = dist._exchange_unique_id(_Ids, 0, 3)
    for t in readers:
        t.join()
    assert got[1][0] == got[0][0] == got[2][0] and len(got[0][0]) == 128
    assert got[0][1] == str(path) and oct(os.stat(path).st_mode & 0o777) == "0o600"
    assert not [p for p in os.listdir(tmp_path) if p.endswith(".tmp")]


def test_a_partial_or_planted_file_is_not_an_id(monkeypatch, tmp_path):
    path = tmp_path / "launch.id"
    monkeypatch.setenv("HIPDRT_RCCL_ID_FILE", str(path))
    path.write_bytes(b"short")                        # not 128 bytes: keep waiting, then give up
    with pytest.raises(RuntimeError, match="no RCCL unique id from rank 0"):
        dist._exchange_unique_id(_Ids, 1, 2, timeout=0.3)
    path.unlink()
    target = tmp_path / "elsewhere"
    target.write_bytes(bytes(128))
    os.symlink(target, path)                          # a link under the expected name is not followed
    with pytest.raises(RuntimeError, match="no RCCL unique id from rank 0"):
        dist._exchange_unique_id(_Ids, 1, 2, timeout=0.3)
    # rank 0 replaces the planted link by its own file instead of writing through it
    uid, _ = dist._exchange_unique_id(_Ids, 0, 2)
    assert not os.path.islink(path) and path.read_bytes() == uid and target.read_bytes() == bytes(128)


def test_launcher_key_separates_launches(monkeypatch):
    monkeypatch.setenv("MASTER_PORT", "29501")
    a = dist._launcher_key()
    monkeypatch.setenv("MASTER_PORT", "29502")
    b = dist._launcher_key()
    assert a != b and str(os.getppid()) in a
