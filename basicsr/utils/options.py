from synergize_motion_appearance_amd.options import ordered_yaml, parse, dict2str  # noqa: F401
