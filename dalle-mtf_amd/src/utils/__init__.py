from .utils import *  # noqa: F401,F403  (the reference does `from src.utils import *`, train_dalle.py:7)
