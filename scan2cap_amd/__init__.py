"""scan2cap_amd -- MI355X-native Scan2Cap hot path.

Layout
  csrc/        hand-written gfx950 HIP kernels + the C ABI (include/s2c_ops.h)
  _C.py        ctypes binding of lib/libs2c_hip.so (no CPU fallback)
  pointnet2/   drop-in for the reference's `pointnet2._ext`, `pointnet2_utils`,
               `pointnet2_modules`, `pytorch_utils`
  models/      CapNet and its sub-modules (same ctor / forward / state_dict)
"""
__version__ = "0.1.0"
