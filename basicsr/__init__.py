"""Drop-in alias: `from basicsr.archs import build_network`, `from basicsr.utils import
img2tensor, tensor2img`, `from basicsr.utils.options import ordered_yaml` (what the
reference's demo.py imports, `basicsr/demo.py:19-21`) resolve to the MI355X-native package."""
