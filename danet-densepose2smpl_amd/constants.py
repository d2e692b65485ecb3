"""Joint tables of the hot path.  Values restated from /root/reference/constants.py:8-100
(JOINT_NAMES order -> JOINT_MAP indices into the 54 joints = 45 smplx joints + 9 extra)."""

FOCAL_LENGTH = 5000.
IMG_RES = 224

# 49 joints = 25 OpenPose + 24 ground-truth joints, as indices into joints54
# (constants.py:12-66 JOINT_NAMES mapped through constants.py:73-93 JOINT_MAP)
JOINT_MAP_49 = [
    24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,   # OpenPose 25
    8, 5, 45, 46, 4, 7, 21, 19, 17, 16, 18, 20, 47, 48, 49, 50, 51, 52, 53, 24, 26, 25, 28, 27,   # GT 24
]
J24_TO_J17 = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 18, 14, 16, 17]
J24_TO_J14 = J24_TO_J17[:14]
J24_TO_J19 = J24_TO_J17[:14] + [19, 20, 21, 22, 23]

# Left/right permutations of the augmentation flips (values restated from /root/reference/constants.py:98-131)
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J14 = H36M_TO_J17[:14]
J24_FLIP_PERM = [5, 4, 3, 2, 1, 0, 11, 10, 9, 8, 7, 6, 12, 13, 14, 15, 16, 17, 18, 19, 21, 20, 23, 22]
J49_FLIP_PERM = [0, 1, 5, 6, 7, 2, 3, 4, 8, 12, 13, 14, 9, 10, 11, 16, 15, 18, 17, 22, 23, 24, 19, 20, 21] + \
    [25 + i for i in J24_FLIP_PERM]
SMPL_JOINTS_FLIP_PERM = [0, 2, 1, 3, 5, 4, 6, 8, 7, 9, 11, 10, 12, 14, 13, 15, 17, 16, 19, 18, 21, 20, 23, 22]
SMPL_POSE_FLIP_PERM = [3 * i + k for i in SMPL_JOINTS_FLIP_PERM for k in range(3)]
IMG_NORM_MEAN = [0.485, 0.456, 0.406]
IMG_NORM_STD = [0.229, 0.224, 0.225]
