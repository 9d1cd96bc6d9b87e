"""Same dict as the reference's models/config.py:1-4 (the operator API for K and the drop probabilities)."""
from dropoutdecoding_amd.config import settings  # noqa: F401
