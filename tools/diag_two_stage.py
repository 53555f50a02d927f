import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_train_loop_gpu as T
from scan2cap_amd.graphs import GraphedPair, GraphedCallable
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd.parallel import BucketedGradAllReduce, TwoStageBackward, split_detector_captioner
bench, wl, model, opt, dd, cfg, dev = T._setup()
name = "backbone_net.sa1.mlp_module.layer2.conv.weight"
P = dict(model.named_parameters())
early, late = split_detector_captioner(model)
two = TwoStageBackward(early, late)
mode = sys.argv[1]
def fwd():
    x = model(dict(dd), use_tf=True, is_eval=False)
    return get_scene_cap_loss(x, dev, cfg, None)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    if mode == "eager_two":
        for p in model.parameters(): p.grad = None
        x = fwd(); two.stage1(x); two.stage2()
        g2 = P[name].grad.clone()
    elif mode == "graph_two":
        def first():
            for p in model.parameters(): p.grad = None
            x = fwd(); two.stage1(x); return x["loss"]
        def second():
            two.stage2()
        pair = GraphedPair(first, second).capture()
        pair.replay_first(); pair.replay_second(); torch.cuda.synchronize()
        g2 = P[name].grad.clone()
    elif mode == "graph_one":
        def step():
            for p in model.parameters(): p.grad = None
            x = fwd(); x["loss"].backward(); return x["loss"]
        g = GraphedCallable(step).capture(); g(); torch.cuda.synchronize()
        g2 = P[name].grad.clone()
    for p in model.parameters(): p.grad = None
    x = fwd(); x["loss"].backward()
    g1 = P[name].grad.clone()
torch.cuda.synchronize()
print(mode, "max|ref| %.3f  max|diff| %.4f" % (float(g1.abs().max()), float((g1 - g2).abs().max())))
