from synergize_motion_appearance_amd.options import ordered_yaml  # noqa: F401
