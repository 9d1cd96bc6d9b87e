"""reference models/llava.py surface -> MI355X implementation."""
from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration, seed  # noqa: F401
from dropoutdecoding_amd.dropout import select_by_vote  # noqa: F401
