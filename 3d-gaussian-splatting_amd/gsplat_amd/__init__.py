from . import scene  # noqa: F401  (placeholder; HIP surface added below)
