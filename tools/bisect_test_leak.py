import sys, pytest
sys.path.insert(0, "/root/repo")
from scan2cap_amd.pointnet2 import fused
from scan2cap_amd.models import decoder_fused
flag = sys.argv[1]
if flag != "none":
    mod, name = flag.split(".")
    setattr({"fused": fused, "decoder_fused": decoder_fused}[mod], name, False)
rc = pytest.main(["tests/test_fused_gpu.py", "tests/test_capnet_golden.py", "-x", "-q", "-m", "gpu", "-k",
                  "test_capnet_gpu or " + sys.argv[2], "-p", "no:cacheprovider"])
print("FLAG", flag, "RC", rc)
