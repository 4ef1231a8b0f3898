"""DALL-E transformer on libdalle_hip: `DALLE` keeps the reference's constructor / forward surface, `DalleEngine` is the
flat-buffer train-step engine underneath."""
from .engine import DalleEngine
from .models import DALLE

__all__ = ["DALLE", "DalleEngine"]
