"""Mirror of DexGanGrasp/data/bps_encoder.py:12-55.  The third-party bps_torch.encode (and the
chamfer_distance CUDA extension under it) is replaced by dgg_bps_encode (HIP, gfx950)."""
import numpy as np
import torch

from .. import _lib, ops


class BPSEncoder():
    def __init__(self, cfg, bps_path, vis=False):
        self.bps_np = np.load(bps_path).astype(np.float32)
        self.device = torch.device('cuda:{}'.format(cfg["gpu_ids"][0])) if torch.cuda.is_available() \
            else torch.device('cpu')
        print("bps encoder device", self.device)
        self._basis = None
        self.VISUALIZE = vis

    def _basis_dev(self):
        if self.device.type != 'cuda':
            raise _lib.DggError("BPSEncoder: no GPU visible; the BPS encode has no CPU path")
        if self._basis is None:
            self._basis = torch.from_numpy(self.bps_np).to(self.device).contiguous()
        return self._basis

    def encode(self, points, return_ids=False):
        """points: (N,3) or (B,N,3) array/tensor (any float dtype) -> {'dists': (B,4096) f32 tensor
        [, 'ids': (B,4096) int32]} — the part of bps_torch.encode()'s result the reference uses."""
        x = torch.as_tensor(np.asarray(points) if not torch.is_tensor(points) else points)
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        d, i = ops.bps_encode(x, self._basis_dev(), return_ids=True)
        out = {'dists': d}
        if return_ids:
            out['ids'] = i
        return out

    def encode_pcd_with_bps(self, obj_pcd):
        """bps_encoder.py:28-55: accepts an Open3D-like cloud (has .points) or an (N,3) array."""
        pts = np.asarray(obj_pcd.points) if hasattr(obj_pcd, "points") else np.asarray(obj_pcd)
        enc_np = self.encode(pts)['dists'].cpu().detach().numpy()
        # if all dists are greater, pcd is too far from bps
        assert enc_np.min() < 0.1, 'The pcd might not be in centered in origin !' + str(enc_np.min())
        return np.squeeze(enc_np)
