"""`python basicsr/animate.py -opt options/<file>.yml` -- the dataset-driven animation entry
(reference `basicsr/animate.py:19-82`): parse the yml, build the test loaders and the model, run
`model.generate_video_image` on each.  One process per GPU; with `--launcher pytorch` (torchrun)
the videos of each loader are split round-robin over the ranks, no collective on the data path."""
import argparse
import os
import sys
from os import path as osp

sys.path.insert(0, osp.abspath(osp.join(osp.dirname(osp.abspath(__file__)), osp.pardir)))

import torch  # noqa: E402

from basicsr.data import build_dataloader, build_dataset  # noqa: E402
from basicsr.models import build_model  # noqa: E402
from basicsr.utils.options import parse  # noqa: E402


def parse_options(root_path, is_train=False, argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-opt", type=str, required=True, help="Path to option YAML file.")
    parser.add_argument("--launcher", choices=["none", "pytorch"], default="none", help="job launcher")
    parser.add_argument("--local_rank", type=int, default=0)
    args = parser.parse_args(argv)
    opt = parse(args.opt, root_path, is_train=is_train)
    opt["dist"] = args.launcher != "none"
    opt["rank"], opt["world_size"] = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if opt["dist"]:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", args.local_rank)))
    seed = opt.get("manual_seed") or 0
    opt["manual_seed"] = seed
    torch.manual_seed(seed + opt["rank"])
    return opt


def rank_subset(dataset, rank, world):
    """every world-th item of `dataset` starting at `rank`, selected by INDEX before any loader exists, so a rank only
    reads and decodes its own videos (videos are independent units; no collective on the data path)."""
    sub = torch.utils.data.Subset(dataset, range(rank, len(dataset), world))
    sub.opt = dataset.opt
    return sub


def test_pipeline(root_path, argv=None):
    opt = parse_options(root_path, is_train=False, argv=argv)
    os.makedirs(opt["path"]["results_root"], exist_ok=True)
    loaders = []
    for _, dataset_opt in sorted(opt["datasets"].items()):
        test_set = build_dataset(dataset_opt)
        if opt["world_size"] > 1:
            test_set = rank_subset(test_set, opt["rank"], opt["world_size"])
        loader = build_dataloader(test_set, dataset_opt, num_gpu=opt["num_gpu"], dist=opt["dist"], sampler=None,
                                  seed=opt["manual_seed"])
        loaders.append(loader)
    model = build_model(opt)
    results = {}
    for loader in loaders:
        results[loader.dataset.opt["name"]] = model.generate_video_image(loader, current_iter=opt["name"], tb_logger=None)
    return opt, results


if __name__ == "__main__":
    test_pipeline(osp.abspath(osp.join(__file__, osp.pardir, osp.pardir)))
