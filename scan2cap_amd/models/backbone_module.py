"""PointNet++ backbone (4 SA + 2 FP), drop-in for models/backbone_module.py:11-127.
Same constructor, same `data_dict` keys, same state_dict names (sa1..sa4, fp1, fp2).
"""
import torch
import torch.nn as nn

from ..pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes

# (npoint, radius, nsample, mlp tail)  -- backbone_module.py:28-62
_SA_SPECS = (
    (2048, 0.2, 64, (64, 64, 128)),
    (1024, 0.4, 32, (128, 128, 256)),
    (512, 0.8, 16, (128, 128, 256)),
    (256, 1.2, 16, (128, 128, 256)),
)


class Pointnet2Backbone(nn.Module):
    def __init__(self, input_feature_dim=0):
        super().__init__()
        self.input_feature_dim = input_feature_dim
        cin = input_feature_dim
        for i, (npoint, radius, nsample, tail) in enumerate(_SA_SPECS, 1):
            setattr(self, "sa%d" % i, PointnetSAModuleVotes(
                npoint=npoint, radius=radius, nsample=nsample,
                mlp=[cin] + list(tail), use_xyz=True, normalize_xyz=True))
            cin = tail[-1]
            # sa2..sa4 sample the centres of the stage before them, which are in FPS pick
            # order: their FPS result is 0..npoint-1 and is proven instead of iterated
            # (csrc/s2c_fps_small.hip; backbone_module.py:106,111,115 note the same fact)
            getattr(self, "sa%d" % i).fps_input_in_pick_order = i >= 2
        self.fp1 = PointnetFPModule(mlp=[256 + 256, 256, 256])
        self.fp2 = PointnetFPModule(mlp=[256 + 256, 256, 256])

    def _break_up_pc(self, pc):
        """xyz (B,N,3) contiguous; features as a (B,C,N) VIEW of the point-major
        input -- the reference materialises the transpose (169 MB copy at cfg3,
        backbone_module.py:68-72); the point-major SA path reads the (B,N,3+C)
        rows in place."""
        xyz = pc[..., :3].contiguous()
        features = pc[..., 3:].transpose(1, 2) if pc.size(-1) > 3 else None
        if features is not None and pc.is_contiguous() and pc.dtype == torch.float32:
            # xyz is a copy of the first three columns of these rows and `features` a view of the
            # rest: the first layer's weight gradient reads [xyz | features] as ONE operand, the
            # cloud's own rows (pointnet2/fused.py: GatherSpec.weight_grad)
            xyz._s2c_cloud = pc
        return xyz, features

    def compute_geometry(self, point_clouds):
        """Everything in the backbone that depends on xyz only: the FPS chain, the
        four ball queries and the two 3-NN searches.  No features, no weights, no
        gradient -- so it can run one batch ahead on a side stream (8 CUs busy)
        while the previous batch's GEMMs own the rest of the chip."""
        with torch.no_grad():
            xyz = point_clouds[..., :3].contiguous()
            geo = {}
            for i in (1, 2, 3, 4):
                g = getattr(self, "sa%d" % i).geometry(xyz)
                geo["sa%d" % i] = g
                xyz = g[1]
            geo["fp1"] = PointnetFPModule.geometry(geo["sa3"][1], geo["sa4"][1])
            geo["fp2"] = PointnetFPModule.geometry(geo["sa2"][1], geo["sa3"][1])
        return geo

    def forward(self, data_dict):
        xyz, features = self._break_up_pc(data_dict["point_clouds"])
        geo = data_dict.get("_geometry")
        if geo is None and xyz.is_cuda:
            geo = self.compute_geometry(data_dict["point_clouds"])
        for i in (1, 2, 3, 4):
            xyz, features, inds = getattr(self, "sa%d" % i)(
                xyz, features, geom=None if geo is None else geo["sa%d" % i])
            if i <= 2:
                data_dict["sa%d_inds" % i] = inds
            data_dict["sa%d_xyz" % i] = xyz
            data_dict["sa%d_features" % i] = features
        features = self.fp1(data_dict["sa3_xyz"], data_dict["sa4_xyz"],
                            data_dict["sa3_features"], data_dict["sa4_features"],
                            geom=None if geo is None else geo["fp1"])
        features = self.fp2(data_dict["sa2_xyz"], data_dict["sa3_xyz"],
                            data_dict["sa2_features"], features,
                            geom=None if geo is None else geo["fp2"])
        data_dict["fp2_features"] = features
        data_dict["fp2_xyz"] = data_dict["sa2_xyz"]
        num_seed = data_dict["fp2_xyz"].shape[1]
        # seeds are the first num_seed FPS picks of SA1 (backbone_module.py:125)
        data_dict["fp2_inds"] = data_dict["sa1_inds"][:, 0:num_seed]
        return data_dict
