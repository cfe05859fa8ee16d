from .synthetic_data import SyntheticDataGenerator  # noqa: F401

__all__ = ["SyntheticDataGenerator"]
