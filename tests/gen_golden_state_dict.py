"""Generates tests/golden/state_dict_keys.json: the names, shapes and dtypes of the
REFERENCE CapNet's state_dict (models/capnet.py through oracle/ref_harness.py) for the
constructor variants of scripts/train.py -- what `torch.load(model.pth)` /
`load_state_dict(strict=False)` (train.py:100, benchmark/predict.py:104) must find in a
drop-in module.  Names and shapes only, no weights.

    python tests/gen_golden_state_dict.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from tests import golden_common as gc  # noqa: E402
from tests import state_dict_common as sc  # noqa: E402


def main():
    ref = ref_harness.reference_modules()
    vocabulary, embeddings = gc.vocab_and_embeddings(gc.GOLDEN_CFG["V"])
    msa = gc.mean_size_arr()
    out = {}
    for name, kw in sc.VARIANTS.items():
        model = ref.capnet.CapNet(vocabulary=vocabulary, embeddings=embeddings,
                                  mean_size_arr=msa, **kw)
        out[name] = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
        print(name, len(out[name]), "entries")
    path = os.path.join(HERE, "golden", "state_dict_keys.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
