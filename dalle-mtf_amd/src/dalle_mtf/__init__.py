from .models import DALLE  # noqa: F401
