"""Constants of the hot-path contract (mirror of reference config.py:3-14,27-32)."""
SMPL_MODEL_DIR = 'additional/smpl'
SMPL_MEAN_PARAMS_PATH = 'additional/neutral_smpl_mean_params_6dpose.npz'
J_REGRESSOR_EXTRA_PATH = 'additional/J_regressor_extra.npy'
COCOPLUS_REGRESSOR_PATH = 'additional/cocoplus_regressor.npy'
H36M_REGRESSOR_PATH = 'additional/J_regressor_h36m.npy'
SMPL_FACES_PATH = 'additional/smpl_faces.npy'
VERTEX_TEXTURE_PATH = 'additional/vertex_texture.npy'
CUBE_PARTS_PATH = 'additional/cube_parts.npy'

FOCAL_LENGTH = 5000.
REGRESSOR_IMG_WH = 256

# 90-joint superset: 0-23 SMPL, 24-44 picked vertices, 45-53 extra, 54-72 cocoplus, 73-89 h36m
ALL_JOINTS_TO_COCO_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]
ALL_JOINTS_TO_H36M_MAP = list(range(73, 90))
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J14 = H36M_TO_J17[:14]
