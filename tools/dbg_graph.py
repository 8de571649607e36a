import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ada_mvs_amd
from ada_mvs_amd import synth, hip_ops
from ada_mvs_amd.graphed import GraphedForward
from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
m = Infer_AdaMVSNet(16, [16, 8, 4], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
m.load_state_dict(synth.seeded_state_dict(m, seed=0)); m = m.cuda().eval()
cfg = dict(views=3, H=64, W=96, ndepths=[16, 8, 4], num_depth=16)
def inputs(seed, baseline, lo, hi):
    imgs, proj, _ = synth.tile_inputs(cfg, batch=1, seed=seed, baseline=baseline)
    return imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, torch.tensor([[lo, hi]], dtype=torch.float32)
def eager(a):
    with torch.no_grad():
        o = m(a[0], a[1], a[2].cuda())
    return {s: o[s]["depth"].clone() for s in ("stage1", "stage2", "stage3")}
def cmp(tag, got, want):
    torch.cuda.synchronize()
    print(tag, {s: (bool(torch.equal(got[s]["depth"], want[s])), float(torch.isnan(got[s]["depth"]).float().mean())) for s in want}, flush=True)
fwd = GraphedForward(m)
A = inputs(0, 8.0, 400., 600.)
wA = eager(A)
with torch.no_grad(): cmp("A first (capture+replay)", fwd(*A), wA)
with torch.no_grad(): cmp("A again (replay)", fwd(*A), wA)
wA2 = eager(A)
with torch.no_grad(): cmp("A after an eager call", fwd(*A), wA)
for tag, a in (("new imgs", (inputs(1, 8.0, 400., 600.)[0], A[1], A[2])), ("new proj", (A[0], inputs(0, 9.0, 400., 600.)[1], A[2])),
               ("new dv", (A[0], A[1], torch.tensor([[380., 640.]]))), ("all new", inputs(1, 9.0, 380., 640.))):
    w = eager(a)
    with torch.no_grad(): cmp(tag, fwd(*a), w)
# the hot path alone in a graph, features eager
print("--- features outside the graph")
g = torch.cuda.CUDAGraph()
a = A
with torch.no_grad():
    simgs = a[0].clone(); sproj = {k: v.clone() for k, v in a[1].items()}; sdv = a[2].cuda().clone()
    spans = torch.tensor(fwd._spans(a[2]), dtype=torch.float32).cuda()
    table, shared = {}, m._stage_workspace
    m._stage_workspace = table
    m(simgs, sproj, sdv); torch.cuda.synchronize()
    feats, shapes = m.extract_features(simgs)
    with torch.cuda.graph(g):
        out = m.infer_from_features(feats, shapes, sproj, sdv, 0.0, span_dev=spans)
    m._stage_workspace = shared
    for rep in range(3):
        b = inputs(rep, 8.0 + rep, 400. - 10 * rep, 600. + 20 * rep)
        w = eager(b)
        simgs.copy_(b[0]); [sproj[k].copy_(v) for k, v in b[1].items()]; sdv.copy_(b[2]); spans.copy_(torch.tensor(fwd._spans(b[2]), dtype=torch.float32))
        f2, _ = m.extract_features(simgs)
        for x, y in zip(feats, f2): x.copy_(y)
        g.replay()
        cmp("hot path graph, rep %d" % rep, out, w)
