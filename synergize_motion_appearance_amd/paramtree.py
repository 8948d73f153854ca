"""Build an nn.Module whose state_dict has exactly the names/shapes of a manifest.

The hot path never calls these sub-modules: they only own parameters/buffers so that
`load_state_dict(strict=True)`, `.cuda()`, `.eval()`, `.state_dict()` behave like the
reference modules' (checkpoint contract)."""
import torch
from torch import nn


class ParamNode(nn.Module):
    """A nameless container; children are attached under the dotted path components."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container, not callable")


def attach(root: nn.Module, manifest, buffers=("running_mean", "running_var", "num_batches_tracked", "down.weight")):
    for name, shape in manifest:
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, ParamNode())
            node = node._modules[p]
        leaf = parts[-1]
        is_buf = leaf in buffers or any(name.endswith(b) for b in buffers if "." in b)
        if leaf == "num_batches_tracked":
            node.register_buffer(leaf, torch.zeros(tuple(shape), dtype=torch.long))
        elif is_buf:
            node.register_buffer(leaf, torch.zeros(tuple(shape)))
        else:
            node.register_parameter(leaf, nn.Parameter(torch.zeros(tuple(shape)), requires_grad=False))
    return root
