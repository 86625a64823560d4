"""Host-side mirror of the reference's `scene` package (only the modules on the 4DGS hot path)."""
from .gaussian_model import GaussianModel  # noqa: F401
from .synthetic import SyntheticScene  # noqa: F401
