from partner_amd.config import get_downsample_factor  # noqa: F401
