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
