"""`python basicsr/test.py -opt options/<file>.yml` -- the frame-pair evaluation entry (reference `basicsr/test.py:19-80`): parse the
yml, build the test loaders (`FramesMotionTransferTestDataset_PairsList`) and the model, run `model.validation` on each (result /
source / driving / visual PNGs + psnr / ssim / l1).  Same option handling as `basicsr/animate.py` (shared `parse_options`)."""
import os
import sys
from os import path as osp

sys.path.insert(0, osp.abspath(osp.join(osp.dirname(osp.abspath(__file__)), osp.pardir)))

from basicsr.animate import parse_options  # noqa: E402
from basicsr.data import build_dataloader, build_dataset  # noqa: E402
from basicsr.models import build_model  # noqa: E402


def test_pipeline(root_path, argv=None):
    opt = parse_options(root_path, is_train=False, argv=argv)
    os.makedirs(opt["path"]["results_root"], exist_ok=True)
    loaders = []
    for _, dataset_opt in sorted(opt["datasets"].items()):
        test_set = build_dataset(dataset_opt)
        loaders.append(build_dataloader(test_set, dataset_opt, num_gpu=opt["num_gpu"], dist=opt["dist"], sampler=None, seed=opt["manual_seed"]))
    model = build_model(opt)
    results = {}
    for loader in loaders:
        results[loader.dataset.opt["name"]] = model.validation(loader, current_iter=opt["name"], tb_logger=None,
                                                               save_img=opt.get("val", {}).get("save_img", False))
    return opt, results


if __name__ == "__main__":
    test_pipeline(osp.abspath(osp.join(__file__, osp.pardir, osp.pardir)))
