from partner_amd.config import Config, ConfigDict  # noqa: F401
