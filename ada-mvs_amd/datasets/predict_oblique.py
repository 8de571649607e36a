"""Predict-time dataset of the whu-omvs pipeline (reference datasets/predict_oblique.py:12-190).

Reads `viewpair.txt`, `image_info.txt`, `camera_info.txt`, `image_path.txt` from `data_folder`; a sample is one
reference view plus `view_num - 1` source views: normalised images [V,3,H,W], multi-scale projection matrices,
the depth range, and the reference image / camera for the output folder.
"""
import os

import numpy as np
from PIL import Image
from torch.utils.data import Dataset

from .data_io import read_cameras_text, read_images_path_text, read_images_text, read_view_pair_text
from .preprocess import crop_input, scale_camera, scale_input

# camera axes X right / Y up / Z back  ->  X right / Y down / Z forward
_FLIP_YZ = np.diag([1.0, -1.0, -1.0])


class MVSDataset(Dataset):
    def __init__(self, data_folder, view_num, args):
        super().__init__()
        self.data_folder = data_folder
        self.viewpair_path = data_folder + "/viewpair.txt"
        self.image_params_path = data_folder + "/image_info.txt"
        self.cam_params_path = data_folder + "/camera_info.txt"
        self.image_path_path = data_folder + "/image_path.txt"
        self.args = args
        self.view_num = view_num
        self.min_interval = args.min_interval
        self.interval_scale = args.interval_scale
        self.num_depth = args.numdepth
        self.counter = 0
        self.cam_params_dict = read_cameras_text(self.cam_params_path)
        self.image_params_dict = read_images_text(self.image_params_path)
        self.image_paths, _ = read_images_path_text(self.image_path_path)
        self.sample_list = read_view_pair_text(self.viewpair_path, self.view_num)
        self.sample_num = len(self.sample_list)

    def __len__(self):
        return len(self.sample_list)

    def read_img(self, filename):
        return Image.open(filename)

    def center_image(self, img, mode="mean"):
        """'mean': per-channel zero mean / unit variance; 'standard': 0..1 (reference predict_oblique.py:53-68)."""
        if mode == "standard":
            return np.array(img, dtype=np.float32) / 255.
        if mode != "mean":
            raise Exception("{}? Not implemented yet!".format(mode))
        x = np.array(img).astype(np.float32)
        var = np.var(x, axis=(0, 1), keepdims=True)
        mean = np.mean(x, axis=(0, 1), keepdims=True)
        return (x - mean) / (np.sqrt(var) + 0.00000001)

    def create_cams(self, image_params, cam_params_dict, num_depth=384, min_interval=0.1):
        """[Rwc|twc] with camera axes X right / Y up  ->  [2,4,4] float32 camera with [Rcw|tcw], X right / Y down
        (reference predict_oblique.py:71-111).  The depth row is (min, (max - min) / num_depth, num_depth, max);
        `min_interval` is accepted and unused, as in the reference."""
        pose = np.zeros((4, 4), dtype=np.float32)            # Twc; the reference inverts it in float32 as well
        pose[:3, :3] = np.matmul(image_params.rotation_matrix, _FLIP_YZ)
        pose[:3, 3] = image_params.project_center
        pose[3, 3] = 1.0
        cam = np.zeros((2, 4, 4), dtype=np.float32)
        cam[0] = np.linalg.inv(pose)
        lens = cam_params_dict[image_params.camera_id]
        cam[1, 0, 0], cam[1, 1, 1] = lens.focallength[0], lens.focallength[1]
        cam[1, 0, 2], cam[1, 1, 2] = lens.x0y0[0], lens.x0y0[1]
        cam[1, 2, 2] = 1
        near, far = image_params.depth[0], image_params.depth[1]
        cam[1, 3] = (near, (far - near) / num_depth, num_depth, far)
        return cam

    def __getitem__(self, idx):
        views = self.sample_list[idx]
        images, proj = [], []
        out = {}
        for v in range(self.view_num):
            image_idx = views[v]
            image = np.array(self.read_img(self.image_paths[image_idx]))
            params = self.image_params_dict[image_idx]
            cam = self.create_cams(params, self.cam_params_dict, self.num_depth, self.min_interval * self.interval_scale)
            image, cam = scale_input(image, cam, scale=self.args.resize_scale)
            image, cam = crop_input(image, cam, max_h=self.args.max_h, max_w=self.args.max_w,
                                    resize_scale=self.args.resize_scale)
            if v == 0:
                out = {"outimage": image, "outcam": cam, "ref_image_path": self.image_paths[image_idx],
                       "depth_values": np.array([cam[1][3][0], cam[1][3][3]], dtype=np.float32),
                       "out_name": os.path.splitext(os.path.basename(params.name))[0],
                       "out_view": os.path.dirname(params.name).split("/")[-1]}
            sampled = scale_camera(cam, scale=self.args.sample_scale)
            p = sampled[0].copy()                            # K [R|t] in the top three rows
            p[:3, :4] = np.matmul(sampled[1, :3, :3], p[:3, :4])
            proj.append(p)
            images.append(self.center_image(image))            # always 'mean': the reference does not pass --normalize on
        full = np.stack(proj)
        half, quarter = full.copy(), full.copy()
        half[:, :2, :] = full[:, :2, :] / 2
        quarter[:, :2, :] = full[:, :2, :] / 4
        out["imgs"] = np.stack(images).transpose([0, 3, 1, 2])
        out["proj_matrices"] = {"stage1": quarter, "stage2": half, "stage3": full}
        return out
