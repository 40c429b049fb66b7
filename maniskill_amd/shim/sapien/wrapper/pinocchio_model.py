"""sapien.wrapper.pinocchio_model.PinocchioModel: CPU kinematics helper of SAPIEN.  ManiSkill uses it for CPU-sim IK
(agents/controllers/utils/kinematics.py:96-140), not on the batched path; not provided."""


class PinocchioModel:
    def __init__(self, urdf_string, gravity):
        raise NotImplementedError("PinocchioModel (CPU inverse kinematics) is not provided by this backend; use the batched "
                                  "GPU kinematics path")

    @classmethod
    def _from_articulation(cls, art):
        raise NotImplementedError("PinocchioModel is not provided by this backend")
