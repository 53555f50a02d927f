"""Order-dependent test failures: run tests/test_fused_gpu.py (restricted by a -k expression) in front of
the golden comparison in ONE process, optionally with a module-level switch off.  Found in round 5 that
test_geometry_slots_grouped_refill_matches_inline left the persistent GEMM grid resized (another
summation order of the BatchNorm partial sums -> a golden gradient moved by 2e-4): tests/conftest.py now
resets the grids after every test.

    python tools/bisect_test_leak.py none|fused.DW_MULTI|decoder_fused.MFMA_CLASSIFIER|... <k-expression>"""
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
