"""yaml -> OrderedDict loader used by demo.py (`basicsr.utils.options.ordered_yaml`,
reference `basicsr/utils/options.py:7-29`) and the option parser of animate.py / train.py (`parse`, :32-88)."""
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


_CHECKPOINT_KEYS = ("resume_state", "pretrain_network")


def _expand_user_paths(section, wanted):
    """`~` -> home for every non-empty entry of `section` whose key satisfies `wanted`."""
    from os.path import expanduser
    for key in list(section):
        if section[key] is not None and wanted(key):
            section[key] = expanduser(section[key])


def parse(opt_path, root_path, is_train=True):
    """yml -> option dict with the derived entries `animate.py` / `train.py` / the model expect (reference
    `basicsr/utils/options.py:32-88`): `is_train`, a time-stamped `name` (or the experiment folder of `path.resume_state`),
    per-dataset `phase` (+ `scale`), `~` expanded in dataset / checkpoint paths, and under `<path.save_path or root_path>`
    either `results/<name>` with `path.{results_root, log, visualization}` (test) or `experiments/<name>` with
    `path.{experiments_root, models, training_states, log, visualization}` (train).  A yml without `datasets:` / `path:`
    (the shipped test.yml) gets empty sections instead of a KeyError."""
    import time
    from os.path import join
    with open(opt_path, "r") as f:
        opt = yaml.load(f, Loader=ordered_yaml()[0])
    opt["is_train"] = bool(is_train)
    paths = opt.setdefault("path", OrderedDict())
    if paths.get("resume_state"):
        opt["name"] = paths["resume_state"].split("/")[-3]
    else:
        opt["name"] = time.strftime("%Y%m%d_%H%M%S", time.localtime()) + "_" + opt["name"]
    datasets = opt.setdefault("datasets", OrderedDict())
    for key in datasets:
        datasets[key]["phase"] = key.split("_")[0]
        if "scale" in opt:
            datasets[key]["scale"] = opt["scale"]
        _expand_user_paths(datasets[key], lambda k: k == "dataroot_gt")
    _expand_user_paths(paths, lambda k: any(tag in k for tag in _CHECKPOINT_KEYS))
    save = paths.get("save_path") or root_path
    if is_train:
        root = join(save, "experiments", opt["name"])
        paths.update(experiments_root=root, models=join(root, "models"), training_states=join(root, "training_states"), log=root,
                     visualization=join(root, "visualization"))
    else:
        root = join(save, "results", opt["name"])
        paths.update(results_root=root, log=root, visualization=join(root, "visualization"))
    return opt


def dict2str(opt, indent_level=1):
    """nested option dict -> the indented `key: value` / `key:[ ... ]` listing the reference logs
    (`basicsr/utils/options.py:91-109`): two spaces per level, sub-dicts bracketed."""
    pad = "  " * indent_level
    lines = [""]
    for key, value in opt.items():
        if isinstance(value, dict):
            lines.append(f"{pad}{key}:[{dict2str(value, indent_level + 1)}{pad}]")
        else:
            lines.append(f"{pad}{key}: {value}")
    return "\n".join(lines) + "\n"
