"""Replays of a captured forward against the eager one (GPU box): the evidence behind "a captured stage contains kernel nodes only".

    python tools/dbg_graph.py <tag>                         tag = a free label printed in front of every line; "noeager" in it:
                                                            no eager launch between the capture and the replays
    (round 6, when the stage still zeroed its GRU states with hipMemsetAsync behind option zero_fill_kernel:
     ADAMVS_ZERO_FILL_KERNEL=0 python tools/dbg_graph.py memset  -> wrong maps from the second replay on;
     profiles/r06_graph_memset_node.txt holds the six runs.  The option is gone: every fill is a kernel now, and this script
     is what to run first if a replay ever disagrees with the eager forward again.)

A tiny cascade (64 x 96, 16 / 8 / 4 planes) through ada_mvs_amd.graphed.GraphedForward: capture on input set A, then B, A, B ...;
per call: bit-identity of the three stage depth maps and of stage 1's view weights / pair depths with the eager forward, and the
fraction of NaN pixels; at the end the span_dev path launched eagerly.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ada_mvs_amd
from ada_mvs_amd import synth, hip_ops
from ada_mvs_amd.graphed import GraphedForward
from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
NOEAGER = "noeager" in mode          # every eager reference run happens BEFORE the capture: nothing but replays afterwards
m = Infer_AdaMVSNet(16, [16, 8, 4], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
m.load_state_dict(synth.seeded_state_dict(m, seed=0)); m = m.cuda().eval()
cfg = dict(views=3, H=64, W=96, ndepths=[16, 8, 4], num_depth=16)
def inputs(seed, baseline, lo, hi):
    imgs, proj, _ = synth.tile_inputs(cfg, batch=1, seed=seed, baseline=baseline)
    return imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, torch.tensor([[lo, hi]], dtype=torch.float32)
def eager(a):
    with torch.no_grad():
        o = m(a[0], a[1], a[2].cuda())
    r = {s: o[s]["depth"].clone() for s in ("stage1", "stage2", "stage3")}
    r["vw"] = torch.stack([x.clone() for x in o["stage1"]["pair_confidence"][:2]]); r["pd"] = torch.stack([x.clone() for x in o["stage1"]["pair_result"]])
    return r
def cmp(tag, got, want):
    torch.cuda.synchronize()
    res = {s: (bool(torch.equal(got[s]["depth"], want[s])), round(float(torch.isnan(got[s]["depth"]).float().mean()), 3)) for s in ("stage1", "stage2", "stage3")}
    vw = torch.stack(got["stage1"]["pair_confidence"][:2]); pd = torch.stack(got["stage1"]["pair_result"])
    res["vw"] = bool(torch.equal(vw, want["vw"])); res["pd"] = bool(torch.equal(pd, want["pd"]))
    print(mode, tag, res, flush=True)
fwd = GraphedForward(m)
A = inputs(0, 8.0, 400., 600.)
B = inputs(1, 9.0, 380., 640.)
wA = eager(A)
if NOEAGER: wB = eager(B)
with torch.no_grad(): cmp("A first (capture+replay)", fwd(*A), wA)
if not NOEAGER: wB = eager(B)
if mode == "sync":
    torch.cuda.synchronize()
with torch.no_grad():
    o = fwd(*B)
    if mode == "sync": torch.cuda.synchronize()
    cmp("B (second call)", o, wB)
    g, st, out = list(fwd.cache.values())[0]
    torch.cuda.synchronize()
    print(mode, "static imgs == B", bool(torch.equal(st["imgs"], B[0])), "proj", all(bool(torch.equal(st["proj"][k], B[1][k])) for k in B[1]),
          "dv", st["dv"].cpu().tolist(), "spans", st["spans"].cpu().tolist(), fwd._spans(B[2]))
    g.replay(); cmp("B replayed again", out, wB)
    g.replay(); cmp("B replayed a third time", out, wB)
    cmp("A back", fwd(*A), wA)
    cmp("B back", fwd(*B), wB)
    for rep in range(3):
        cmp("A again %d" % rep, fwd(*A), wA)
        if not NOEAGER: eager(B)
        cmp("B again %d" % rep, fwd(*B), wB)
    # the span_dev path launched eagerly (no graph) on B
    spans = torch.tensor(fwd._spans(B[2]), dtype=torch.float32).cuda()
    feats, shapes = m.extract_features(B[0])
    o2 = m.infer_from_features(feats, shapes, B[1], B[2].cuda(), 0.0, span_dev=spans)
    cmp("eager with span_dev on B", o2, wB)
