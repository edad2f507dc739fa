"""Error types mirroring the ones the reference raises on this path."""


class DGLError(Exception):
    """Counterpart of `dgl._ffi.base.DGLError` (reference src/no-sampling/models.py:9, raised at :229-231,
    :336-346, :360-364): same name so `except DGLError` in caller code keeps working."""
