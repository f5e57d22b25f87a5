"""Drop-in counterpart of the reference's ``predictive_coding`` package (reference __init__.py:1-2)."""
from .pc_layer import PCLayer
from .pc_trainer import PCTrainer

__all__ = ["PCLayer", "PCTrainer"]
