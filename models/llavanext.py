"""reference models/llavanext.py surface -> MI355X implementation."""
from dropoutdecoding_amd.llavanext import CustomLlavaNextForConditionalGeneration, seed  # noqa: F401
from dropoutdecoding_amd.dropout import select_by_vote  # noqa: F401
