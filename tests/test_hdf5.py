"""maniskill_amd.hdf5 (ctypes over the system's libhdf5): the slice of h5py's API that the reference's trajectory code uses
(mani_skill/utils/wrappers/record.py:271,574-700; trajectory/dataset.py:16-41; merge_trajectory.py:30-60), writing the real format.
Checked against the HDF5 project's own ``h5dump`` where it is installed, and through this package's recorder / replayer."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from maniskill_amd import hdf5 as h5

pytestmark = pytest.mark.skipif(not h5.available(), reason="no libhdf5 on this machine")
H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


def test_h5py_surface_round_trip(tmp_path):
    """the calls RecordEpisode / clean_trajectories / merge_trajectory / load_h5_data make"""
    path = tmp_path / "traj.h5"
    f = h5.File(path, "w")
    g = f.create_group("traj_0", track_order=True)
    g.create_dataset("actions", data=np.arange(12, dtype=np.float32).reshape(4, 3), dtype=np.float32)
    g.create_dataset("terminated", data=np.array([False, False, True, False]), dtype=bool)
    g.create_dataset("rgb", data=np.arange(2 * 4 * 4 * 3, dtype=np.uint8).reshape(2, 4, 4, 3), dtype=np.uint8, compression="gzip", compression_opts=5)
    obs = g.create_group("obs", track_order=True)
    obs.create_dataset("agent/qpos", data=np.ones((5, 9)))
    g.create_dataset("zz_first_by_name_last_by_creation", data=np.float32(2.5))
    g.attrs["note"] = "x"; g.attrs["count"] = 3
    assert f.filename == os.fspath(path) and list(f.keys()) == ["traj_0"] and "traj_0/obs/agent/qpos" in f and "traj_1" not in f
    with pytest.raises(ValueError):
        f.create_group("traj_0")
    f.close()
    with h5.File(path, "r") as r:
        assert list(r["traj_0"].keys()) == ["actions", "terminated", "rgb", "obs", "zz_first_by_name_last_by_creation"]    # track_order
        assert isinstance(r["traj_0"]["actions"], h5.Dataset) and isinstance(r["traj_0/obs"], h5.Group)
        a = r["traj_0"]["actions"]
        assert a.shape == (4, 3) and a.dtype == np.float32 and a[2, 1] == 7.0 and a[:].dtype == np.float32 and len(a) == 4
        assert r["traj_0/terminated"][:].dtype == np.bool_ and r["traj_0/terminated"][:].tolist() == [False, False, True, False]
        assert r["traj_0/rgb"].compression == "gzip" and r["traj_0/rgb"][1, 3, 3, 2] == 95
        assert r["traj_0/obs/agent/qpos"][()].sum() == 45 and r["traj_0/obs/agent/qpos"].dtype == np.float64
        assert r["traj_0/zz_first_by_name_last_by_creation"][()] == 2.5 and r["traj_0/zz_first_by_name_last_by_creation"].shape == ()
        assert r["traj_0"].attrs["note"] == "x" and r["traj_0"].attrs["count"] == 3
        with pytest.raises(KeyError):
            r["traj_9"]
        with pytest.raises(OSError):
            r.create_group("nope")
        w = h5.File(tmp_path / "merged.h5", "w")
        r.copy("traj_0", w, "traj_7")
        w.close()
    with h5.File(tmp_path / "merged.h5", "r+") as m:
        assert np.array_equal(m["traj_7/actions"][:], np.arange(12, dtype=np.float32).reshape(4, 3)) and m["traj_7"].attrs["note"] == "x"
        m["traj_0"] = m["traj_7"]        # record.py clean_trajectories: rename by link + delete
        del m["traj_7"]
        assert list(m.keys()) == ["traj_0"] and len(m) == 1 and m["traj_0/rgb"].shape == (2, 4, 4, 3)
    with pytest.raises(OSError):
        (tmp_path / "not.h5").write_bytes(b"0" * 64)
        h5.File(tmp_path / "not.h5", "r")
    with pytest.raises(OSError):
        h5.File(tmp_path / "missing.h5", "r")


@pytest.mark.skipif(H5DUMP is None, reason="h5dump is not installed")
def test_files_are_hdf5_to_the_reference_tool(tmp_path):
    """h5dump (the HDF5 project's own reader) sees the hierarchy, the types h5py would have written, and the values"""
    p = tmp_path / "t.h5"
    with h5.File(p, "w") as f:
        g = f.create_group("traj_0", track_order=True)
        g.create_dataset("actions", data=np.array([[0.5, -1.0], [2.0, 3.25]], np.float32))
        g.create_dataset("success", data=np.array([True, False]))
        g.create_dataset("env_states/actors/cube", data=np.arange(26, dtype=np.float32).reshape(2, 13))
        g.create_dataset("obs/depth", data=np.full((2, 16, 16, 1), 7, np.int16), compression="gzip", compression_opts=5)
    out = subprocess.run([H5DUMP, os.fspath(p)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    txt = out.stdout
    assert 'GROUP "traj_0"' in txt and 'DATASET "actions"' in txt and "H5T_IEEE_F32LE" in txt and "( 2, 2 ) / ( 2, 2 )" in txt
    assert "0.5, -1" in txt and "2, 3.25" in txt
    assert '"FALSE"' in txt and '"TRUE"' in txt and "H5T_STD_I8LE" in txt and "TRUE, FALSE" in txt        # h5py's bool enum
    assert 'GROUP "env_states"' in txt and 'GROUP "actors"' in txt and 'DATASET "cube"' in txt
    hdr = subprocess.run([H5DUMP, "-H", "-p", os.fspath(p)], capture_output=True, text=True).stdout
    assert "COMPRESSION DEFLATE { LEVEL 5 }" in hdr and "H5T_STD_I16LE" in hdr


def test_recorder_and_replay_through_hdf5(tmp_path, oracle_factory):
    """RecordEpisode(container="h5") -> trajectory.h5 (real HDF5) -> replay_trajectory: the recorded states come back exactly"""
    from maniskill_amd.envs.pick_cube import PickCubeEnv
    from maniskill_amd.trajectory import RecordEpisode, load_trajectory, replay_trajectory

    n = 3
    env = RecordEpisode(PickCubeEnv(num_envs=n, px_factory=oracle_factory), str(tmp_path), container="h5")
    env.reset(seed=[5, 6, 7])
    gen = torch.Generator().manual_seed(1)
    for _ in range(6):
        env.step(2 * torch.rand(n, 8, generator=gen) - 1)
    env.close()
    path = str(tmp_path / "trajectory.h5")
    with open(path, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"                     # the format's signature
    meta, arrays = load_trajectory(path)
    assert len(meta["episodes"]) == n and arrays["traj_0"]["actions"].shape == (6, 8) and arrays["traj_0"]["actions"].dtype == np.float32
    assert arrays["traj_1"]["env_states"]["actors"]["cube"].shape == (7, 13) and arrays["traj_2"]["success"].dtype == np.bool_
    res = replay_trajectory(PickCubeEnv(num_envs=n, px_factory=oracle_factory), path)
    assert res.num_replays == n and res.max_state_error == 0.0


def test_unicode_arrays_are_written_as_utf8_and_old_libraries_are_refused(tmp_path):
    """numpy 'U' arrays (UCS-4) used to be written as raw bytes and ended at the first NUL; libhdf5 < 1.10 has a 32-bit hid_t this binding cannot talk to"""
    from maniskill_amd import hdf5
    if not hdf5.available():
        pytest.skip("no libhdf5")
    p = str(tmp_path / "u.h5")
    with hdf5.File(p, "w") as f:
        f.create_dataset("names", data=np.array(["PickCube-v1", "größe"]))
        f.attrs["who"] = np.array(["a", "bc"])
    with hdf5.File(p, "r") as f:
        got = [x.decode("utf-8") if isinstance(x, bytes) else str(x) for x in np.asarray(f["names"][()]).tolist()]
        assert got == ["PickCube-v1", "größe"], got
    assert tuple(int(x) for x in hdf5.version.hdf5_version.split(".")[:2]) >= (1, 10)
