"""Forward recurrence of the teacher-forced decoder at the cfg3 shape (R=8, K=10, H=512, E=300,
F=128, T=30): the persistent kernel (csrc/s2c_decoder_persist.hip) against the 5-launches-per-step
chain, us per decoder step, eager and from a replayed graph."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan2cap_amd.models import decoder_fused
from scan2cap_amd.models.caption_module import TopDownSceneCaptionModule


def main():
    R, K, H, E, F, T = 8, 10, 512, 300, 128, 30
    V = 40
    words = ["w%d" % i for i in range(V)]
    vocab = {"word2idx": {w: i for i, w in enumerate(words)},
             "idx2word": {str(i): w for i, w in enumerate(words)}}
    emb = {w: np.random.randn(E).astype(np.float32) for w in words}
    mod = TopDownSceneCaptionModule(vocab, emb, E, F, H, K, num_locals=K).cuda()
    word_embs = torch.randn(R, 32, E, device="cuda") * 0.3
    obj = torch.randn(R, K, F, device="cuda") * 0.5
    tgt = torch.randn(R, F, device="cuda") * 0.5
    masks = torch.ones(R, K, device="cuda")
    for persist in (True, False, True, False):
        decoder_fused.set_persist(persist)
        with torch.no_grad():
            for _ in range(3):
                decoder_fused.decode(mod, word_embs, tgt, obj, masks, T)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                with torch.cuda.graph(gr, stream=s):
                    decoder_fused.decode(mod, word_embs, tgt, obj, masks, T)
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                gr.replay()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 50
        print("persist=%d: forward %.1f us (%.2f us per decoder step, incl. the hoisted GEMMs)"
              % (persist, ms * 1000, ms * 1000 / T), flush=True)
    decoder_fused.set_persist(True)
    # phase stamps of workgroup 0 (shader cycles; s_memtime), averaged over steps 1..T-1
    decoder_fused.PROF = torch.zeros(8 * T * 16, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        decoder_fused.decode(mod, word_embs, tgt, obj, masks, T)
    torch.cuda.synchronize()
    p = decoder_fused.PROF.cpu().numpy().reshape(8, T, 16).astype(np.float64)
    decoder_fused.PROF = None
    names = {0: "step start", 1: "P1 h2 polled", 2: "P1 x1 published", 3: "P2 operand ready",
             4: "P2 partials", 5: "P2 barrier", 6: "P2 h1 published", 7: "P3 h1 polled",
             8: "P3 q published", 9: "P4 q polled", 10: "P4 x2 published", 11: "P5 operand ready",
             12: "P5 h2 published", 13: "P4 scores done", 14: "P4 scores summed",
             15: "P4 att done"}
    for wv, label in ((0, "wave 0 (h part, polls)"), (4, "wave 4 (x part)"), (2, "wave 2 (h part, LDS)"), (6, "wave 6 (x part)")):
        print(label)
        t0 = p[wv, 1:, 0]
        prev = np.zeros_like(t0)
        for slot in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 14, 15, 10, 11, 12):
            v = p[wv, 1:, slot]
            if (v == 0).all():
                continue
            rel = (v - t0).mean()
            print("   %-20s +%7.0f cycles  (step %7.0f)" % (names[slot], rel, rel - prev.mean()))
            prev = v - t0
    per = (p[0, 2:, 0] - p[0, 1:-1, 0]).mean()
    print("cycles per decoder step: %.0f" % per)
    # the backward kernel (steps run t = T-1 .. 0; stamps of workgroup 0)
    decoder_fused.PROF_BWD = torch.zeros(8 * T * 16, dtype=torch.int64, device="cuda")
    objg = obj.clone().requires_grad_(True)
    out, _ = decoder_fused.decode(mod, word_embs, tgt, objg, masks, T)
    out.sum().backward()
    torch.cuda.synchronize()
    p = decoder_fused.PROF_BWD.cpu().numpy().reshape(8, T, 16).astype(np.float64)
    decoder_fused.PROF_BWD = None
    bn = {0: "step start", 1: "B1 done", 2: "B2 da2 polled", 3: "B2 dq published",
          4: "B3 operands ready", 5: "B3 dh1 published", 6: "B4 done", 7: "B5 da1 ready",
          8: "B5 dh2 published", 9: "B1 operand ready", 10: "B1 dots done", 11: "B1 barrier",
          12: "B4 operand ready", 13: "B4 dots done", 14: "B4 barrier"}
    for wv in (0, 2, 3, 4, 6):
        print("backward, wave %d (job %d)" % (wv, wv & 3))
        sel = slice(1, T - 1)                      # t = T-2 .. 1
        t0 = p[wv, sel, 0]
        prev = np.zeros_like(t0)
        for slot in (0, 9, 10, 11, 1, 2, 3, 4, 5, 12, 13, 14, 6, 7, 8):
            v = p[wv, sel, slot]
            if (v == 0).all():
                continue
            rel = (v - t0).mean()
            print("   %-20s +%7.0f cycles  (step %7.0f)" % (bn[slot], rel, rel - prev.mean()))
            prev = v - t0
    print("cycles per backward step: %.0f" % (p[0, 1:-2, 0] - p[0, 2:-1, 0]).mean())


if __name__ == "__main__":
    main()
