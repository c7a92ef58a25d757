from . import leo_orbit, sc_attitudes  # noqa: F401
