"""Mesh paths of a URDF as the shim's loader resolves them (maniskill_amd/shim/sapien/wrapper/urdf_loader.py): relative to the URDF,
`package://` relative to `package_dir` or to the URDF's directory (what the reference documents, utils/building/urdf_loader.py:68), and --
when that file does not exist -- the ROS reading, where `package://<name>/...` names a directory that is an ancestor of the URDF
(mani_skill/assets/robots/koch/follower_arm_v1.1.urdf refers to package://koch/meshes/*.stl)."""
import os

import maniskill_amd.shim as shim


def test_package_paths(tmp_path):
    shim.install()
    from sapien.wrapper.urdf_loader import URDFLoader
    robots = tmp_path / "robots"
    (robots / "arm" / "meshes").mkdir(parents=True)
    (robots / "arm" / "meshes" / "link.stl").write_bytes(b"")
    (robots / "arm" / "local.stl").write_bytes(b"")
    urdf_dir = os.fspath(robots / "arm")
    ld = URDFLoader()
    assert ld._resolve("local.stl", urdf_dir) == os.path.join(urdf_dir, "local.stl")
    assert ld._resolve("package://meshes/link.stl", urdf_dir) == os.path.join(urdf_dir, "meshes", "link.stl")
    assert ld._resolve("package://arm/meshes/link.stl", urdf_dir) == os.path.join(os.fspath(robots), "arm", "meshes", "link.stl")     # one level up
    missing = ld._resolve("package://nowhere/x.stl", urdf_dir)
    assert missing == os.path.join(urdf_dir, "nowhere", "x.stl")              # reported where the reference's rule puts it
    ld.package_dir = os.fspath(tmp_path)
    assert ld._resolve("package://robots/arm/local.stl", urdf_dir) == os.path.join(os.fspath(tmp_path), "robots", "arm", "local.stl")
    assert ld._resolve("/abs/file.stl", urdf_dir) == "/abs/file.stl"
