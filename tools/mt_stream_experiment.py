import sys, time, threading, torch
sys.path.insert(0, '.')
import ada_mvs_amd
from ada_mvs_amd import synth, _lib
from bench import build_model
dev = torch.device("cuda", 0)
cfg = "cfg2"; c = synth.CONFIGS[cfg]
model, sd = build_model(cfg, dev)
interval = 200.0 / 192
for G, Bg in ((1, 32), (2, 16), (4, 8)):
    groups = []
    with torch.no_grad():
        for g in range(G):
            imgs, proj, dv = synth.tile_inputs(cfg, batch=Bg, seed=g)
            f = model.extract_features(imgs.to(dev))
            groups.append((f, {k: v.to(dev) for k, v in proj.items()}, dv.to(dev)))
    streams = [torch.cuda.Stream() for _ in range(G)]
    def work(g, n):
        with torch.no_grad(), torch.cuda.stream(streams[g]):
            for _ in range(n):
                (fc, sh), pj, dv = groups[g]
                model.infer_from_features(fc, sh, pj, dv, interval, group=g)
    for n in (1, 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(g, n)) for g in range(G)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("G=%d Bg=%d eager threads: %.1f ms per %d tiles -> %.1f maps/s" % (G, Bg, 1e3 * dt / 3, G * Bg, 3 * G * Bg / dt))
