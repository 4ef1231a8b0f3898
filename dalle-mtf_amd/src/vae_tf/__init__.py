from .models import DiscreteVAE  # noqa: F401
