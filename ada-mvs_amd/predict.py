"""Depth inference over a whu-omvs predict folder: the reference's predict_whu.py on the MI355X hot path.

    python predict_whu.py --data_folder <dir> --output_folder <dir> --loadckpt <ckpt> [reference options]

Same options, same input formats and the same files in the output folder as reference predict_whu.py:24-163
(`<view>/<name>_init.pfm`, `_prob.pfm`, `<name>.jpg`, `<name>.txt`, `color/*.png` with --display).  Differences by
design: one process per GPU instead of nn.DataParallel (under torch.distributed.run the samples are dealt
round-robin to the ranks, every rank writes its own files, no collective); `--precision bf16x3` selects the
split-bf16 mode of `--model adamvs`; `--model msrednet` is ada_mvs_amd/models/msrednet.py (fp32).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from .datasets import find_dataset_def
from .datasets.data_io import save_pfm, write_red_cam


def str2bool(v):
    return str(v).lower() not in ("0", "false", "no", "off", "")


def build_parser():
    ap = argparse.ArgumentParser(description="Predict depth for whu-omvs test set")
    ap.add_argument("--model", default="adamvs", help="select model from [msrednet, adamvs]")
    ap.add_argument("--dataset", default="predict_oblique", help="select dataset")
    ap.add_argument("--data_folder", required=True, help="test datapath")
    ap.add_argument("--output_folder", required=True, help="output dir")
    ap.add_argument("--loadckpt", default=None, help="load a specific checkpoint ({'model': state_dict})")
    ap.add_argument("--view_num", type=int, default=5, help="Number of images (1 ref image and view_num - 1 view images).")
    ap.add_argument("--numdepth", type=int, default=192, help="the number of depth values")
    ap.add_argument("--max_w", type=int, default=3712, help="Maximum image width")
    ap.add_argument("--max_h", type=int, default=5504, help="Maximum image height")
    ap.add_argument("--min_interval", type=float, default=0.1, help="min_interval in the bottom stage")
    ap.add_argument("--fext", type=str, default=".jpg", help="Type of images.")
    ap.add_argument("--normalize", type=str, default="mean", help="accepted and unused, as in the reference")
    ap.add_argument("--resize_scale", type=float, default=0.5, help="output scale for depth and image (W and H)")
    ap.add_argument("--sample_scale", type=float, default=1, help="Downsample scale for building cost volume (W and H)")
    ap.add_argument("--interval_scale", type=float, default=1, help="the number of depth values")
    ap.add_argument("--batch_size", type=int, default=1, help="samples per forward pass")
    ap.add_argument("--display", type=str2bool, default=True, help="display depth images")
    ap.add_argument("--share_cr", action="store_true", help="whether share the cost volume regularization")
    ap.add_argument("--ndepths", type=str, default="48,32,8", help="ndepths")
    ap.add_argument("--depth_inter_r", type=str, default="4,2,1", help="depth_intervals_ratio")
    ap.add_argument("--cr_base_chs", type=str, default="8,8,8", help="cost regularization base channels")
    # not in the reference
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"], help="arithmetic of the convolutions")
    ap.add_argument("--num_workers", type=int, default=4, help="loader processes (the reference uses 1); decoding and resizing a view takes ~1 s of CPU")
    ap.add_argument("--graph", type=str2bool, default=True,
                    help="--model adamvs: capture the forward of an input shape into a hipGraph once and replay it for every later sample "
                         "of that shape (ada_mvs_amd/graphed.py); 0: launch eagerly")
    ap.add_argument("--seeded_weights", type=int, default=None,
                    help="no checkpoint: seeded random weights (ada_mvs_amd.synth), for dry runs and tests")
    return ap


def build_model(args, device):
    ndepths = [int(nd) for nd in args.ndepths.split(",") if nd]
    ratios = [float(r) for r in args.depth_inter_r.split(",") if r]
    chs = [int(ch) for ch in args.cr_base_chs.split(",") if ch]
    if args.model == "msrednet":
        from .models.msrednet import Infer_CascadeREDNet
        model = Infer_CascadeREDNet(num_depth=args.numdepth, ndepths=ndepths, depth_interals_ratio=ratios,
                                    share_cr=args.share_cr, cr_base_chs=chs)
    elif args.model == "adamvs":
        from .models.adamvs import Infer_AdaMVSNet
        model = Infer_AdaMVSNet(num_depth=args.numdepth, ndepths=ndepths, depth_intervals_ratio=ratios,
                                share_cr=args.share_cr, cr_base_chs=chs, precision=args.precision)
    else:
        raise Exception("{}? Not implemented yet!".format(args.model))
    if args.loadckpt:
        print("loading model {}".format(args.loadckpt))
        state = torch.load(args.loadckpt, map_location="cpu")["model"]
        # the reference saves nn.DataParallel(model).state_dict(): keys carry a 'module.' prefix
        state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}
        model.load_state_dict(state)
    elif args.seeded_weights is not None:
        from . import synth
        model.load_state_dict(synth.seeded_state_dict(model, seed=args.seeded_weights))
    else:
        raise Exception("--loadckpt is required (or --seeded_weights N for a dry run)")
    return model.to(device).eval()


def _color_maps(folder, name, depth_est, prob):
    """color/<name>_init.png and _prob.png (reference predict_whu.py:118-135); needs matplotlib."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    shown = 36000.0 - depth_est
    for i in range(shown.shape[1]):                 # non-finite values of a column -> its minimum - 1
        col = shown[:, i]
        col[np.isinf(col)] = np.nan
        col[np.isnan(col)] = np.nanmin(col) - 1
    plt.imsave(folder + "/color/%s_init.png" % name, shown, format="png")
    plt.imsave(folder + "/color/%s_prob.png" % name, np.nan_to_num(prob).clip(0, 1), format="png")


def save_outputs(args, output_folder, depth_est, prob, ref_image, ref_cam, ref_path, vid, name):
    """One sample's files (reference predict_whu.py:96-140)."""
    folder = output_folder + ("/%s/" % vid)
    os.makedirs(folder + "/color/", exist_ok=True)
    if args.display:
        _color_maps(folder, name, depth_est, prob)
    save_pfm(folder + ("/%s_init.pfm" % name), depth_est)
    save_pfm(folder + ("/%s_prob.pfm" % name), prob)
    from PIL import Image
    with open(folder + ("/%s.jpg" % name), "wb") as f:      # PNG bytes under a .jpg name, RGBA, as the reference's plt.imsave writes it
        Image.fromarray(np.ascontiguousarray(ref_image)).convert("RGBA").save(f, format="png")
    write_red_cam(folder + ("/%s.txt" % name), ref_cam, ref_path)


def predict_depth(args):
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    # loader processes come from a fork server, and the server is started HERE, before anything below touches the GPU
    # (torch.cuda.is_available() already initialises HIP): a child forked from a process with an initialised HIP runtime
    # must never call into it.  The server is a freshly spawned interpreter; the loader children are forked from IT.
    ctx = None
    if args.num_workers > 0:
        import multiprocessing
        from multiprocessing import forkserver
        ctx = multiprocessing.get_context("forkserver")
        forkserver.ensure_running()
    if not torch.cuda.is_available():
        raise RuntimeError("predict: needs an MI355X (there is no CPU fallback for the Ada-MVS hot path)")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)
    dataset = find_dataset_def(args.dataset)(args.data_folder, args.view_num, args)
    if world > 1:                                    # independent samples: deal them out, no data-path collective
        dataset = Subset(dataset, list(range(rank, len(dataset), world)))
    loader = DataLoader(dataset, args.batch_size, shuffle=False, num_workers=args.num_workers, drop_last=False,
                        multiprocessing_context=ctx)
    model = build_model(args, device)
    forward = model
    if args.graph and args.model == "adamvs":
        from .graphed import GraphedForward
        forward = GraphedForward(model)
    os.makedirs(args.output_folder, exist_ok=True)
    step, t_first = 0, time.time()
    with torch.no_grad():
        for sample in loader:
            t0 = time.time()
            imgs = sample["imgs"].to(device)
            proj = {k: v.to(device) for k, v in sample["proj_matrices"].items()}
            # (the graphed forward reads depth_min / depth_max from the loader's HOST tensor: no device round trip per sample)
            outputs = forward(imgs, proj, sample["depth_values"] if forward is not model else sample["depth_values"].to(device))
            depth = outputs["depth"].float().cpu().numpy()
            conf = outputs["photometric_confidence"].float().cpu().numpy()
            t1 = time.time()
            for b in range(depth.shape[0]):
                save_outputs(args, args.output_folder, np.float32(depth[b]), np.float32(conf[b]),
                             sample["outimage"][b].numpy(), sample["outcam"][b].numpy(), sample["ref_image_path"][b],
                             sample["out_view"][b], sample["out_name"][b])
                step += 1
            print("depth inference {} finished, image {} finished, ({:3f}s and {:3f} sec/step)".format(
                step, sample["out_name"][-1], t1 - t0, time.time() - t1))
    print("final, total_cnt = {}, total_time = {:3f}".format(step, time.time() - t_first))
    return step


def main(argv=None):
    args = build_parser().parse_args(argv)
    print("argv:", sys.argv[1:] if argv is None else argv)
    return predict_depth(args)


if __name__ == "__main__":
    main()
