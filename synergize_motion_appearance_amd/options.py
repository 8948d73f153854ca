"""yaml -> OrderedDict loader used by demo.py (`basicsr.utils.options.ordered_yaml`,
reference `basicsr/utils/options.py:7-29`)."""
from collections import OrderedDict

import yaml


def ordered_yaml():
    """(Loader, Dumper) that keep mapping order as OrderedDict."""
    try:
        from yaml import CDumper as Dumper
        from yaml import CLoader as Loader
    except ImportError:  # pragma: no cover
        from yaml import Dumper, Loader
    tag = yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG
    Dumper.add_representer(OrderedDict, lambda dumper, data: dumper.represent_dict(data.items()))
    Loader.add_constructor(tag, lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    return Loader, Dumper


def parse(opt_path, root_path, is_train=True):
    """yml -> option dict with the derived test paths (reference `basicsr/utils/options.py:32-88`):
    name prefixed with a time stamp, dataset `phase` from the key, `~` expanded in checkpoint paths,
    `path.results_root / log / visualization` under `<save_path>/results/<name>`.  A yml without a
    `datasets:` section (the shipped test.yml) parses to an empty dataset dict instead of a KeyError."""
    import time
    from os import path as osp
    with open(opt_path, mode="r") as f:
        opt = yaml.load(f, Loader=ordered_yaml()[0])
    if is_train:
        raise NotImplementedError("training options are SURVEY row N2; the MI355X-native build parses test ymls")
    opt["is_train"] = False
    opt["name"] = f"{time.strftime('%Y%m%d_%H%M%S', time.localtime())}_{opt['name']}"
    opt.setdefault("datasets", OrderedDict())
    for phase, dataset in opt["datasets"].items():
        dataset["phase"] = phase.split("_")[0]
        if "scale" in opt:
            dataset["scale"] = opt["scale"]
        if dataset.get("dataroot_gt") is not None:
            dataset["dataroot_gt"] = osp.expanduser(dataset["dataroot_gt"])
    opt.setdefault("path", OrderedDict())
    for key, val in opt["path"].items():
        if val is not None and ("resume_state" in key or "pretrain_network" in key):
            opt["path"][key] = osp.expanduser(val)
    results_root = osp.join(opt["path"].get("save_path", root_path), "results", opt["name"])
    opt["path"]["results_root"] = results_root
    opt["path"]["log"] = results_root
    opt["path"]["visualization"] = osp.join(results_root, "visualization")
    return opt


def dict2str(opt, indent_level=1):
    msg = "\n"
    for k, v in opt.items():
        if isinstance(v, dict):
            msg += " " * (indent_level * 2) + k + ":[" + dict2str(v, indent_level + 1) + " " * (indent_level * 2) + "]\n"
        else:
            msg += " " * (indent_level * 2) + k + ": " + str(v) + "\n"
    return msg
