"""Generates tests/golden/caption_plain.npz: the REFERENCE's SceneCaptionModule (plain GRU
captioner, models/caption_module.py:40-200) run through
oracle/ref_harness.py on seeded inputs with deterministic weights -- train (teacher
forcing) and eval (greedy decode of every proposal).  Inputs + outputs only.

    python tests/gen_golden_plain_caption.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from tests import golden_common as gc  # noqa: E402
from tests import plain_caption_common as pc  # noqa: E402


def main():
    torch.manual_seed(0)
    ref = ref_harness.reference_modules()
    vocabulary, embeddings = gc.vocab_and_embeddings(pc.V)
    mod = ref.caption.SceneCaptionModule(vocabulary, embeddings, 300, 128, 512, pc.K)
    sd = mod.state_dict()
    with torch.no_grad():
        gc.det_fill_(sd)
    mod.load_state_dict(sd)
    inputs = pc.make_inputs(embeddings, vocabulary)
    out = {"in/" + k: v for k, v in inputs.items()}
    mod.train()
    dd = mod({k: torch.from_numpy(v) for k, v in inputs.items()}, use_tf=True, is_eval=False)
    out["train/lang_cap"] = dd["lang_cap"].detach().numpy()
    out["train/good_bbox_masks"] = dd["good_bbox_masks"].numpy()
    out["train/pred_ious"] = np.asarray(dd["pred_ious"].detach().numpy(), np.float64)
    mod.eval()
    with torch.no_grad():
        dd = mod({k: torch.from_numpy(v) for k, v in inputs.items()}, use_tf=False,
                 is_eval=True, max_len=pc.EVAL_LEN)
    out["eval/lang_cap"] = dd["lang_cap"].numpy()
    print({k: v.shape for k, v in out.items() if not k.startswith("in/")})
    path = os.path.join(HERE, "golden", "caption_plain.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
