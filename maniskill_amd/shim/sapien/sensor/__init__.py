"""``sapien.sensor``: SAPIEN's simulated active stereo depth sensor (mani_skill/sensors/depth_camera.py:12 imports the two
names at module level; the sensor itself is a ray-traced IR stereo model and is not provided by this backend)."""


class StereoDepthSensorConfig:
    def __init__(self):
        self.rgb_resolution = (1920, 1080)
        self.ir_resolution = (1280, 720)
        self.rgb_intrinsic = None
        self.ir_intrinsic = None
        self.trans_pose_l = self.trans_pose_r = None
        self.light_pattern = None
        self.min_depth, self.max_depth = 0.2, 10.0


class StereoDepthSensor:
    def __init__(self, *a, **k):
        raise NotImplementedError("the stereo depth sensor simulation is not provided by this backend")
